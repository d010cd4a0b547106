"""Condense the rocprofv3 outputs of tools/make_profiles.sh into the small files kept under profiles/ (runs on the GPU box,
because the raw per-dispatch counter tables exceed what gpurun copies back).

  <out>/kernel_stats_train_step.csv   rocprofv3 --kernel-trace --stats summary of `bench.py` (verbatim + header)
  <out>/pmc_hbm_bytes.csv             per kernel: dispatches, mean FETCH_SIZE / WRITE_SIZE per dispatch, corrected MB
  <out>/pmc_latest.json               what bench.py reads for roofline.traffic

Units / corrections (MI355X_MICROARCH.md, HBM): rocprofv3 reports FETCH_SIZE and WRITE_SIZE in KB per dispatch; on gfx950
FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads, so fetch bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact."""
import csv, glob, json, os, sys, collections

ROUND = os.environ.get("ROUND", "r06")

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    assert f, pattern
    return f[0]


# ---- kernel stats -------------------------------------------------------------------------------------------------
OURS = ("alignq_site", "site_fwd", "site_bwd", "slab_reduce", "site_prep", "bn_stats", "bn_bwd_apply", "bn_finalize", "bnq_", "corrl_", "conv3x3",
        "wgrad", "qgemm", "cdf_", "stem_", "convgen", "transition", "dgrad_s2", "mt_", "admm_update", "act_quant", "weight_quant", "weight_stats", "uniform_quantize", "sgd_", "admm_loss")


def is_ours(name):
    return "ck::" not in name and "_ZN2ck" not in name and any(k in name for k in OURS)


stats = one("stats/**/*kernel_stats.csv")
rows = list(csv.reader(open(stats)))
head, data = rows[0], rows[1:]
ours = [r for r in data if is_ours(r[0])]
others = [r for r in data if not is_ours(r[0])]
with open(os.path.join(out, "kernel_stats_train_step.csv"), "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-shapes --no-dp-probe --no-other-configs   (MI355X)\n")
    fo.write("# 3 eager warm-ups + HIP-graph capture + 33 replays of the ResNet-20 8W/8A CDF+ADMM step (batch 128), then bench.py's per-kernel\n")
    fo.write("# measurement loops (site kernels x ~55 launches per shape, act_quant / copy / add x 23 launches on 2^26 elements)\n")
    fo.write("# --- this repository's kernels (all of them), by total time ---\n")
    w = csv.writer(fo, quoting=csv.QUOTE_ALL)
    w.writerow(head)
    w.writerows(ours)
    fo.write(f"# --- other kernels (top 40 of {len(others)} by total time): MIOpen incl. its find-mode trials for the five convolutions\n")
    fo.write("# --- that stay on MIOpen, rocBLAS, torch elementwise ---\n")
    w.writerows(others[:40])


def counters(which, name):
    f = one(f"{which}/**/*counter_collection.csv")
    acc = collections.defaultdict(lambda: [0, 0.0, 0])
    with open(f) as fi:
        rd = csv.DictReader(fi)
        for row in rd:
            if row.get("Counter_Name") != name:
                continue
            k = row["Kernel_Name"]
            a = acc[k]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] = int(row.get("Grid_Size", 0) or 0)
    return acc


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    nf, sf, grid = fetch.get(k, [0, 0.0, 0])
    nw, sw, grid2 = write.get(k, [0, 0.0, 0])
    f_kb = sf / nf if nf else 0.0
    w_kb = sw / nw if nw else 0.0
    rows.append((k, grid or grid2, max(nf, nw), f_kb, 2 * f_kb * 1024 / 1e6, w_kb, w_kb * 1024 / 1e6,
                 (2 * f_kb + w_kb) * 1024 / 1e6))
rows.sort(key=lambda r: -r[7] * r[2])
with open(os.path.join(out, "pmc_hbm_bytes.csv"), "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shapes --no-dp-probe --no-other-configs\n")
    fo.write("# filler roles off in these two passes (ALIGNQ_FILL=0): every launch moves its own role's bytes only\n")
    fo.write("# Units: KB per dispatch (mean over dispatches). gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> fetch_corrected = 2*FETCH\n")
    fo.write("kernel,grid_threads,dispatches,FETCH_SIZE_KB,fetch_corrected_MB,WRITE_SIZE_KB,write_MB,total_corrected_MB\n")
    for r in rows[:60]:
        fo.write('"%s",%d,%d,%.1f,%.2f,%.1f,%.2f,%.2f\n' % (r[0][:110], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))


def avg_bytes(substr):
    sel = [r for r in rows if substr in r[0]]
    n = sum(r[2] for r in sel)
    return sum(r[7] * 1e6 * r[2] for r in sel) / n if n else None


latest = {"site_bwd": {"hbm_bytes_per_launch_avg": avg_bytes("site_bwd4_kernel")},
          "site_partials": {"hbm_bytes_per_launch_avg": avg_bytes("site_fwd4_kernel")},
          "act_quant_fwd": {"hbm_bytes_per_launch_avg": avg_bytes("act_quant_fwd_kernel")},
          "act_quant_bwd": {"hbm_bytes_per_launch_avg": avg_bytes("act_quant_bwd_kernel")},
          "source": f"profiles/{ROUND}_pmc_hbm_bytes.csv (2 x FETCH_SIZE + WRITE_SIZE per dispatch, mean over all dispatches of the "
                    "kernel in `bench.py --steps 2`: the three site shapes of ResNet-20 for the site kernels, 2^26 elements for act_quant)"}
json.dump(latest, open(os.path.join(out, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(latest, indent=1))
