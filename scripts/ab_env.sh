#!/bin/bash
# A/B of an environment knob on ONE box: scripts/ab_env.sh VAR "v1 v2 ..." [rounds] [bench args...]
var=$1; vals=$2; rounds=${3:-2}; shift 3
for r in $(seq $rounds); do
  for v in $vals; do
    ms=$(env $var=$v python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 "$@" 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
    echo "$var=$v $ms ms"
  done
done
