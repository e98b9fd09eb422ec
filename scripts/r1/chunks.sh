#!/bin/bash
# bench.py over chunk sizes (pairs per launch)
for c in "$@"; do
  v=$(timeout -k 5 200 python bench.py --no-cpu-baseline --no-roof --chunk $c 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "chunk $c: $v fps"
done
