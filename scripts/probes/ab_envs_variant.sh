#!/bin/bash
# as ab_envs.sh for another bench variant:   V="--variant m --batch 32" bash scripts/probes/ab_envs_variant.sh "A=1" "B=2" ...
cd $GRAFT_REPO_ROOT
STEPS=${STEPS:-30}; ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for SW in "X_DEFAULT=1" "$@"; do
    echo -n "$SW   "; env $SW python3 bench.py $V --steps $STEPS --warmup 5 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | python3 -c "import sys, json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
