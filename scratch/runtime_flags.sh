#!/bin/bash
# A/B of HIP runtime switches on the captured CIFAR step (one box, one process per setting).
set -o pipefail
out=gpurun_out/r3/runtime_flags.txt
mkdir -p gpurun_out/r3
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-shapes --no-dp-probe --no-other-configs 2>/dev/null \
    | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"])' >> $out
}
run A=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run AMD_OPT_FLUSH=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run GPU_MAX_HW_QUEUES=1
run A=1
cat $out
