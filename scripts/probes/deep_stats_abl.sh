cd $GRAFT_REPO_ROOT
for d in 0 512 1024 1536; do echo "HDY_DEEP_DEBUG=$d"; HDY_LIB=libhdy_deepdbg.so HDY_DEEP_DEBUG=$d PYTHONPATH=. python scripts/probes/deep_stats_cost.py 2>&1 | grep -v amdgpu | cut -c1-100; done
