# per-iteration kernel table of the captured step: STEPS=<n> EXTRA='<bench.py flags>' bash tools/count_step_kernels.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/cnt && mkdir -p gpurun_out/cnt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cnt/stats -o run -- python3 bench.py --steps ${STEPS:-197} --warmup 0 --no-cpu-baseline --no-kernels $EXTRA > gpurun_out/cnt/log 2>&1
python3 - <<'PY'
import csv,glob,os,re
N=int(os.environ.get('STEPS','197'))+3
f=glob.glob('gpurun_out/cnt/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=[]
for r in rows:
    name=re.sub(r'\(anonymous namespace\)::|alignq_site::|void ','',r['Name'])[:80]
    calls=int(r['Calls']); avg=float(r['AverageNs'])/1e3
    out.append((calls/N*avg, f"{name:82s} calls {calls:6d} per-step {calls/N:6.2f} avg {avg:8.2f} us per-step-us {calls/N*avg:7.1f}"))
out.sort(reverse=True)
open('gpurun_out/cnt_summary.txt','w').write("\n".join(o[1] for o in out))
PY
rm -rf gpurun_out/cnt
