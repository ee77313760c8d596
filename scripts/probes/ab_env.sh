#!/bin/bash
# A/B of one environment switch on one box, alternating: train step with the switch unset and set.   bash scripts/probes/ab_env.sh HDY_EXEC=0 [steps] [rounds]
cd $GRAFT_REPO_ROOT
SW=$1; STEPS=${2:-40}; ROUNDS=${3:-3}
for r in $(seq $ROUNDS); do
  echo -n "default   "; python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c1-150
  echo -n "$SW "; env $SW python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-infer --no-roofline 2>/dev/null | tail -1 | cut -c1-150
done
