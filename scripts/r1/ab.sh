#!/bin/bash
# A/B two builds of librcflow on the same box: scripts/ab.sh <libA> <libB> [exp10 args...]
A=$1; B=$2; shift 2
for r in 1 2 3; do
  for L in $A $B; do
    echo "== $L"
    RCFLOW_LIB=$PWD/$L timeout -k 5 120 python scripts/r1/exp10.py "$@" 2>&1 | grep -v amdgpu | tail -2
  done
done
