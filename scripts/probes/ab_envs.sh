#!/bin/bash
# A/B of several environment settings on one box, alternating, ms per train step:   bash scripts/probes/ab_envs.sh "A=1 B=2" "C=3" ... ; STEPS / ROUNDS from the environment
cd $GRAFT_REPO_ROOT
STEPS=${STEPS:-40}; ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for SW in "X_DEFAULT=1" "$@"; do
    echo -n "$SW   "; env $SW python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys, json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
