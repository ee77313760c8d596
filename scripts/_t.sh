cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "deep_pipelined_weight" 2>&1 | tail -3
HDY_WGRAD_DEEP_S1=1 python3 scripts/layer_probe.py 'B wgrd +256x +256 k3 s1' 10 2>&1 | grep -v amdgpu
for i in 1 2 3; do
python3 bench.py --steps 40 --warmup 15 --no-infer --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | cut -c90-140
HDY_NO_WGRAD_DEEP=1 python3 bench.py --steps 40 --warmup 15 --no-infer --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | cut -c90-140
done
for v in "m 32 640" "l 16 640"; do set -- $v; python3 bench.py --variant $1 --batch $2 --size $3 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c90-140;  HDY_NO_WGRAD_DEEP=1 python3 bench.py --variant $1 --batch $2 --size $3 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c90-140; done
