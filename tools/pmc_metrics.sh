# Runs ON THE GPU BOX: per-kernel hardware-counter means of the captured step, one rocprofv3 pass per counter group
# (PMC_GROUPS="VALUBusy MfmaUtil|LDSBankConflict MemUnitStalled|MeanOccupancyPerCU" bash tools/pmc_metrics.sh) -> gpurun_out/pmc_metrics.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GROUPS_="${PMC_GROUPS:-VALUBusy MfmaUtil|LDSBankConflict MemUnitStalled|MeanOccupancyPerCU OccupancyPercent}"
rm -rf gpurun_out/pmcm && mkdir -p gpurun_out/pmcm
i=0
IFS='|' read -ra GR <<< "$GROUPS_"
for g in "${GR[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcm/p$i -o run -- python3 ${PMC_PROG:-bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shapes --no-dp-probe --no-kernels} > gpurun_out/pmcm/log$i 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('gpurun_out/pmcm/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::|alignq_site::|void ', '', r['Kernel_Name'])[:58]
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
names = sorted({c for k in acc for c in acc[k]})
with open('gpurun_out/pmc_metrics.txt', 'w') as fo:
    fo.write('kernel'.ljust(60) + ' n ' + ' '.join(n[:18].rjust(18) for n in names) + '\n')
    for k in sorted(acc, key=lambda k: -max(v[0] for v in acc[k].values())):
        n = max(v[0] for v in acc[k].values())
        fo.write(k.ljust(60) + f'{n:4d} ' + ' '.join((f"{acc[k][c][1] / acc[k][c][0]:18.2f}" if c in acc[k] else ' ' * 18) for c in names) + '\n')
PY
rm -rf gpurun_out/pmcm
