# config 5 with the N = 32 solve variants (0: shuffle kernel for a materialised loop matrix, 3: two rows per lane with DPP broadcasts)
cd /root/repo
for v in 0 3 0 3; do
python - $v <<'PY'
import runpy, sys, io, contextlib, json
v = int(sys.argv[1])
from flamo_amd import _lib
_lib.lib().fl_debug_set_solve_variant(v)
sys.argv = ["tools/bench_fdn.py", "--workload", "config5", "--dtype", "f32", "--steps", "8"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    try:
        runpy.run_path("tools/bench_fdn.py", run_name="__main__")
    except SystemExit:
        pass
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(f"solve variant {v}: {d['f32']['ms_per_step']:.3f} ms eager, {d['f32'].get('graph_ms_per_step', 0):.3f} ms replayed; grad relerr {d['f32'].get('graph_vs_eager_grad_relerr')}")
PY
done
