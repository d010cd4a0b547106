#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel statistics of config 5 (ResNet-50-DANN, batch 28 + 28, eager launches so
# every kernel shows by name; MIOpen immediate mode so that no find-mode trial kernels pollute the table) -> gpurun_out/prof_office/office_kernel_stats.csv (top kernels by total time).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_office && mkdir -p gpurun_out/prof_office
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_office/raw -o run -- python3 bench.py --model resnet50_dann --batch 28 --steps 10 --warmup 2 --no-graph --no-miopen-find --no-cpu-baseline --no-kernels > gpurun_out/prof_office/stats.log 2>&1
f=$(find gpurun_out/prof_office/raw -name "*kernel_stats.csv" | head -1)
head -100 "$f" > gpurun_out/prof_office/office_kernel_stats.csv
grep "^{\"metric" gpurun_out/prof_office/stats.log > gpurun_out/prof_office/bench_under_rocprof.json || true
rm -rf gpurun_out/prof_office/raw
cut -c1-200 gpurun_out/prof_office/office_kernel_stats.csv | head -45
