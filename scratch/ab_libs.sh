#!/bin/bash
# A/B of library builds on one box: scratch/ab_libs.sh <rounds> <name|default> ...   (tools/lib/lib_<name>.so)
rounds=$1; shift
for r in $(seq $rounds); do
  for n in "$@"; do
    if [ "$n" = default ]; then so=""; else so=$PWD/tools/lib/lib_$n.so; fi
    v=$(ALIGNQ_SO=$so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-shapes --no-dp-probe --no-other-configs 2>/dev/null | tail -1 | python3 -c 'import sys,json; print("%.4f" % json.loads(sys.stdin.read())["ms_per_step"])')
    echo "$n $v"
  done
done
