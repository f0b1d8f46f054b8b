cd /root/repo
timeout 600 python -m pytest tests/test_cascade2.py -q -m gpu -s > gpurun_out/c2_test.log 2>&1
timeout 300 python tools/dbg/cascade2_dbg.py > gpurun_out/c2_dbg.log 2>&1
grep -E "passed|failed|lanes_rc/|lanes_plain/" gpurun_out/c2_test.log | grep -v parity | tail -40; grep -v "^section" gpurun_out/c2_dbg.log
