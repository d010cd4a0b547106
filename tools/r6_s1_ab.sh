#!/bin/bash
# Runs ON THE GPU BOX (round 6): the small-batch site tests, then the site kernels at config 5's extreme shapes for the product build and
# the variant libraries given as arguments (tools/build_variant.sh), on one box.
mkdir -p gpurun_out/r6s1
python -m pytest tests/ -q -m gpu -x -k "site or Site or office or config5 or small_batch or round3 or round5" 2>&1 | tail -15 > gpurun_out/r6s1/pytest.txt
cat gpurun_out/r6s1/pytest.txt
bash tools/s1_grid_sweep.sh "$@" > gpurun_out/r6s1/s1_sweep.txt 2>&1
cat gpurun_out/r6s1/s1_sweep.txt
