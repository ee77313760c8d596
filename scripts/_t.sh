cd $GRAFT_REPO_ROOT
python3 scripts/bench_hnet.py s 16 1280 4 2>/dev/null | tail -1 | cut -c1-300
HDY_NO_DEEP=1 python3 scripts/bench_hnet.py s 16 1280 4 2>/dev/null | tail -1 | cut -c1-300
python3 scripts/bench_mask.py s 16 1280 4 2>/dev/null | tail -1 | cut -c1-300
HDY_NO_DEEP=1 python3 scripts/bench_mask.py s 16 1280 4 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py --variant m6 --batch 16 --size 1280 --steps 10 --warmup 3 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c90-140
