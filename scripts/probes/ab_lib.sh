#!/bin/bash
# A/B of two builds of the library on one box: layer times (scripts/layer_probe.py) and the train step, alternating.
#   bash scripts/probes/ab_lib.sh libhdy_old.so '<layer regex>' [steps]
cd $GRAFT_REPO_ROOT
OLD=$1; PAT=${2:-'F fwd|B dgrd'}; STEPS=${3:-40}
echo "== layers, new"; python3 scripts/layer_probe.py "$PAT" 10 2>&1 | grep "^[FB] "
echo "== layers, $OLD"; HDY_LIB=$OLD python3 scripts/layer_probe.py "$PAT" 10 2>&1 | grep "^[FB] "
for r in 1 2 3; do
  echo "== step, new";  python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c1-160
  echo "== step, $OLD"; HDY_LIB=$OLD python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c1-160
done
