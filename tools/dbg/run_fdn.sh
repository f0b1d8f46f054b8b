cd /root/repo
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_round3_parity.py tests/test_hip_parity.py -q -m gpu -x -k "sparsity or constant or colorless" 2>&1 | tail -5
for c in 0 1; do
FLAMO_TORCH_CRITERIA=$c python tools/train_colorless_fdn.py --steps 300 --graph 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('torch criteria $c:', round(d['ms_per_step'], 4), 'ms per step; losses', d['loss_last'])"
done
FLAMO_TORCH_CRITERIA=0 python tools/train_colorless_fdn.py --steps 300 --graph --fused-adam 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('fused adam:', round(d['ms_per_step'], 4), 'ms per step; losses', d['loss_last'])"
