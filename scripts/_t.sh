cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "fast_kernels_against" 2>&1 | tail -15
