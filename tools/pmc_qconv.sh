# Runs ON THE GPU BOX: hardware counters of the GEMM convolution kernels (tools/qconv_bench.py, selected layers): where a wave's cycles go
# (SQ_WAIT_ANY = parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_* = issuing) and the L2 hit rate.
#   ONLY="l3.conv" bash tools/pmc_qconv.sh   -> gpurun_out/pmc_qconv.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcq && mkdir -p gpurun_out/pmcq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcq/p0 -o run -- python3 tools/qconv_bench.py --no-ref --i16 --only "${ONLY:-l3.conv}" > gpurun_out/pmcq/log0 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmcq/p1 -o run -- python3 tools/qconv_bench.py --no-ref --i16 --only "${ONLY:-l3.conv}" > gpurun_out/pmcq/log1 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmcq/p2 -o run -- python3 tools/qconv_bench.py --no-ref --i16 --only "${ONLY:-l3.conv}" > gpurun_out/pmcq/log2 2>&1
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/pmcq/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])[:64] + ' g' + r.get('Grid_Size', '')
        if 'qgemm' not in k: continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
with open('gpurun_out/pmc_qconv.txt', 'w') as fo:
    for k in sorted(acc):
        fo.write(k + '\n')
        for c in sorted(acc[k]):
            fo.write(f"    {c:28s} {acc[k][c][1] / acc[k][c][0]:16.1f}  (n={acc[k][c][0]})\n")
print(open('gpurun_out/pmc_qconv.txt').read()[:6000])
PY
rm -rf gpurun_out/pmcq
