cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -x -q -k "deep_pipelined" 2>&1 | tail -3
PAT='F fwd +128-> +128 k3 s1 @40|B dgrd +256<- +256 k1 s1 @40|B dgrd +128<- +128 k3 s1 @40'
for d in 32 96; do HDY_LIB=libhdy_deepdbg.so HDY_DEEP_BN=128 HDY_DEEP_DEBUG=$d python3 scripts/deep_stamps.py "$PAT" 2>&1 | grep -v amdgpu | cut -c1-300; done
for i in 1 2; do
python3 bench.py --steps 40 --warmup 15 --no-infer --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | cut -c90-140
HDY_NO_DEEP=1 python3 bench.py --steps 40 --warmup 15 --no-infer --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | cut -c90-140
done
python3 scripts/bench_infer.py l 128 1024 3 2>/dev/null | tail -1 | cut -c1-200
HDY_DEEP_BN=128 python3 scripts/bench_infer.py l 128 1024 3 2>/dev/null | tail -1 | cut -c1-200
