cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dp.py tests/test_gpu_entrypoints.py -x -q -k "overlapped or hnet_two or spawns" 2>&1 | tail -15
