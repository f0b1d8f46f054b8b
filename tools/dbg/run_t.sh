cd /root/repo; timeout 600 python -m pytest tests/test_spectral.py -q -m gpu -x -k "launch_pair" 2>&1 | tail -40
