#!/bin/bash
# layer times (scripts/layer_probe.py, 30 replays each) under several builds of the library on one box:  bash scripts/probes/ab_libs.sh '<regex>' lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
PAT=$1; shift
for L in "$@"; do
  echo "== $L"; HDY_LIB=$L python3 scripts/layer_probe.py "$PAT" 30 2>&1 | grep "^[FB] "
done
