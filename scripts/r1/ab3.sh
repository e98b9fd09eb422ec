#!/bin/bash
# round-robin several builds of librcflow on the same box: scripts/ab3.sh lib1 lib2 ... -- exp10 args
libs=(); while [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
for r in 1 2 3; do for L in "${libs[@]}"; do echo "== $L"; RCFLOW_LIB=$PWD/$L timeout -k 5 120 python scripts/r1/exp10.py "$@" 2>&1 | grep -v amdgpu | tail -1; done; done
