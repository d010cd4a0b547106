#!/bin/bash
# Runs ON THE GPU BOX: hardware counters of the kernels whose name contains $1, for the program given after it
#   bash tools/pmc_kernel.sh site1_fwd python3 tools/s1_only.py 256 56 4   -> gpurun_out/pmc_<name>.txt
set -e
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmck && mkdir -p gpurun_out/pmck
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmck/p$i -o run -- "$@" > gpurun_out/pmck/log$i 2>&1 || true
  i=$((i+1))
done
PAT=$pat python3 - <<'PY'
import csv, glob, re, collections, os
pat = os.environ['PAT']
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/pmck/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::|alignq_site::|void ', '', r['Kernel_Name'])[:72] + ' grid ' + r.get('Grid_Size', '')
        if pat not in k: continue
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
out = 'gpurun_out/pmc_%s.txt' % re.sub(r'\W', '_', pat)
with open(out, 'w') as fo:
    for k in sorted(acc):
        fo.write(k + '\n')
        for c in sorted(acc[k]):
            fo.write(f"    {c:28s} {acc[k][c][1] / acc[k][c][0]:18.1f}  (n={acc[k][c][0]})\n")
print(open(out).read()[:8000])
PY
rm -rf gpurun_out/pmck
