# bench.py's multi-rank branch with two ranks on GPU 0 over gloo, each rank under faulthandler: a stall dumps both tracebacks
cd /root/repo
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 WORLD_SIZE=2 BENCH_ALLOW_SHARED_GPU=1 BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=0 LOCAL_RANK=0 timeout -s ABRT ${1:-300} python -X faulthandler bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r0.out 2> gpurun_out/r0.err &
RANK=1 LOCAL_RANK=1 timeout -s ABRT ${1:-300} python -X faulthandler bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r1.out 2> gpurun_out/r1.err &
wait
for r in 0 1; do echo "== rank $r"; cut -c1-300 gpurun_out/r$r.out | tail -3; grep -v "amdgpu.ids" gpurun_out/r$r.err | tail -40; done
