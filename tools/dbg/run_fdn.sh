cd /root/repo
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_round3_parity.py tests/test_hip_parity.py tests/test_objectives.py -q -m gpu -x -k "magnitude or e7 or colorless or constant or objectives or mse" 2>&1 | tail -4
python tools/train_colorless_fdn.py --steps 300 --graph --fused-adam 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('colorless:', round(d['ms_per_step'], 4), 'ms per step; losses', d['loss_last'])"
python - <<'PY'
from flamo_amd.processor import dsp
dsp.MAGNITUDE_LAYER = False
import runpy, sys, io, contextlib, json
sys.argv = ["tools/train_colorless_fdn.py", "--steps", "300", "--graph", "--fused-adam"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("tools/train_colorless_fdn.py", run_name="__main__")
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("colorless, magnitude layer as the callable:", round(d["ms_per_step"], 4), "ms per step; losses", d["loss_last"])
PY
