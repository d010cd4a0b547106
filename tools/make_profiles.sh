#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the three rocprofv3 passes behind profiles/ (kernel trace + stats; PMC FETCH_SIZE; PMC
# WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).  Outputs land in gpurun_out/prof/{stats,fetch,write}.
set -e
export ROUND=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o run -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-shapes --no-dp-probe --no-other-configs > gpurun_out/prof/stats.log 2>&1
# the two counter passes run with the filler roles off (a launch's bytes are then its own role's: with them on, the narrow sites'
# launches also read the slabs of earlier sites / convolutions, bytes that the closing reductions read otherwise)
export ALIGNQ_FILL=0
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/fetch -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shapes --no-dp-probe --no-other-configs > gpurun_out/prof/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/write -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shapes --no-dp-probe --no-other-configs > gpurun_out/prof/write.log 2>&1
unset ALIGNQ_FILL
find gpurun_out/prof -name "*.csv" | head -20
# keep the merge-back under the 64 MiB cap: drop the per-dispatch traces of the PMC runs except the counter tables
find gpurun_out/prof/fetch gpurun_out/prof/write -name "*kernel_trace.csv" -delete
du -sh gpurun_out/prof
python3 tools/summarise_profiles.py gpurun_out/prof gpurun_out/profiles
grep "^{\"metric" gpurun_out/prof/stats.log > gpurun_out/profiles/bench_under_rocprof.json || true
rm -rf gpurun_out/prof
python3 bench.py --steps 200 --warmup 20 > gpurun_out/profiles/bench_n1.json 2> gpurun_out/profiles/bench_n1.err
tail -c 600 gpurun_out/profiles/bench_n1.json
