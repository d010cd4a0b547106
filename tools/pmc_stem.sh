# Runs ON THE GPU BOX: hardware counters of the Office stem kernel (tools/stem_bench.py) -> gpurun_out/pmc_stem.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcs && mkdir -p gpurun_out/pmcs
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcs/p0 -o run -- python3 tools/stem_bench.py > gpurun_out/pmcs/log0 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmcs/p2 -o run -- python3 tools/stem_bench.py > gpurun_out/pmcs/log2 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmcs/p3 -o run -- python3 tools/stem_bench.py > gpurun_out/pmcs/log3 2>&1 || true
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/pmcs/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])[:64] + ' g' + r.get('Grid_Size', '')
        if 'qstem7' not in k: continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
with open('gpurun_out/pmc_stem.txt', 'w') as fo:
    for k in sorted(acc):
        fo.write(k + '\n')
        for c in sorted(acc[k]):
            fo.write(f"    {c:28s} {acc[k][c][1] / acc[k][c][0]:16.1f}  (n={acc[k][c][0]})\n")
print(open('gpurun_out/pmc_stem.txt').read()[:6000])
PY
rm -rf gpurun_out/pmcs
