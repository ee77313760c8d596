cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_model.py tests/test_gpu_dp.py tests/test_gpu_multihead.py tests/test_gpu_mask.py tests/test_gpu_seg.py -x -q 2>&1 | tail -4
for i in 1 2; do
python3 bench.py --steps 40 --warmup 15 --no-infer --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | cut -c90-140
HDY_BATCH_REDUCE=0 python3 bench.py --steps 40 --warmup 15 --no-infer --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | cut -c90-140
done
