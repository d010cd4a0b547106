#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: images/sec of one full training step
(zero_grad -> fwd -> CE + trans_loss -> bwd -> SGD.step -> ADMM_OPT.step) of ResNet-20 CIFAR-shape,
8W/8A, CDF alignment + ADMM, batch 128 per GPU (BASELINE.json configs[1]), synthetic 3x32x32 inputs
resident in HBM, random-init weights.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts the N ranks itself as a child
                                                          `python -m torch.distributed.run --nproc-per-node N bench.py ...`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Every rank exits with code 2 when WORLD_SIZE differs from --gpus.

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      : the dominant hand-written kernel of the step (by summed time over its launches in one step),
                  timed live with HIP events on the launch stream, vs the gfx950 peak that bounds it;
  kernels       : the same measurement for every hot-path kernel family (incl. the plain CDF-quantise
                  kernel on a 2^26-element tensor, the "quant-kernel HBM GB/s" of BASELINE.json);
  cpu_baseline  : the eager-torch CPU restatement of the reference (oracle/torch_ref.py, kind "port") timed on
                  this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 matrix peak (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (the reference's train_batch_size)")
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--model", default="resnet20", choices=["resnet20", "resnet56", "resnet50_dann"],
                    help="resnet50_dann = BASELINE config 5 (Office-31 shapes 3x224x224, use --batch 28)")
    ap.add_argument("--tree", default="admm", choices=["admm", "cdf"], help="resnet20 / resnet56: the CDF+ADMM tree (BASELINE "
                    "configs[1..3]) or the CDF-only tree of configs[0] (cdf_alignment/resnet-20-cifar-10); the headline line is the "
                    "default; the metric string names the tree")
    ap.add_argument("--lr", type=float, default=None,
                    help="learning rate (default: the reference's 0.04 for the CIFAR nets; 0.004 for resnet50_dann, whose "
                         "reference default 0.04 assumes ImageNet-pretrained weights (dann_office/model/resnet.py:274-288, no "
                         "network here): from RANDOM init 0.04 diverges within ~20 steps, in the eager-torch restatement too)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel roofline measurements")
    ap.add_argument("--no-shapes", action="store_true", help="skip the roofline-sized microbench shapes (kernels.roofline_shapes); "
                    "the rocprofv3 passes of tools/make_profiles.sh use it so that their per-kernel averages hold the CIFAR shapes only")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed steps per thread count of the CPU baseline (BASELINE.md: >= 10)")
    ap.add_argument("--no-dual", action="store_true", help="resnet50_dann: source and target pass as two traversals (the "
                    "default merges them: one convolution launch per layer for both batches)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short captured runs of BASELINE.json's other "
                    "configurations (extra key other_configs: ResNet-20 2W/2A, ResNet-56 4W/4A, ResNet-50-DANN batch 28)")
    ap.add_argument("--no-fuse-bn", action="store_true", help="keep torch/MIOpen batch-norm instead of folding it into the ADMM-site kernels")
    ap.add_argument("--dp-selftest", action="store_true", help="run the DP path (RCCL all-reduce, two graphs) even at N=1")
    ap.add_argument("--no-dp-probe", action="store_true", help="skip the short world-size-1 DP-path measurement (extra key dp_selftest)")
    ap.add_argument("--no-qconv", action="store_true", help="keep MIOpen for every convolution (default: Conv2d_Q's 3x3 "
                    "stride-1 body convolutions, forward and data gradient, run on alignq_conv3x3_nhwc)")
    ap.add_argument("--no-pack-bins", action="store_true", help="keep relu(act_q0(.)) of every block as an fp32 tensor instead of "
                    "its int8 / int16 level index (SURVEY 8f-N2; default: packed)")
    ap.add_argument("--nchw", action="store_true",
                    help="contiguous NCHW activations/weights instead of torch.channels_last (the default: MIOpen's NHWC "
                         "kernels need no transposes; 2.22 vs 2.50 ms per step)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch rehearsal without a GPU: the ranks form a gloo group on the CPU, run the timed region's "
                         "bookkeeping over a stub step and rank 0 prints a line whose metric says 'dry run' (tests/"
                         "test_bench_assembly.py drives `bench.py --gpus 2 --dry-run` through the self-launch)")
    ap.add_argument("--no-miopen-find", action="store_true",
                    help="torch.backends.cudnn.benchmark=False: MIOpen's immediate-mode solver choice instead of its find "
                         "step for the (not ours) convolutions; find is on by default, it is worth ~2%% of a step")
    return ap.parse_args()


def time_call(fn, reps, warm=3):
    """Average duration (s) of fn() over `reps` back-to-back enqueues, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def time_call_serial(fn, reps, warm=2):
    """Median duration (s) of fn() with every launch bracketed by its own event pair and a synchronisation: the conditions of a
    rocprofv3 --kernel-trace run (no launch follows another back to back)."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def time_call_rot(fn, reps, sets, warm=3):
    """time_call for fn(i) with i cycling over `sets` operand sets (cold operands: see measure_roofline_shapes)."""
    it = [0]

    def step():
        fn(it[0] % sets)
        it[0] += 1
    return time_call(step, reps, warm)


def measure_kernels(dev, B, k, site_F_counts, hw_of_F, folded=True, nhwc=False, shapes=True):
    """Per-kernel live timings through the C ABI, HIP events on the launch stream, over R rotating buffer sets.
    site_F_counts: {F: number of ADMM sites with F features}; hw_of_F: {F: H*W} (the channel size for the BN fold).
    folded=True times the entry points the training step actually uses (batch-norm + ReLU folded into the site kernels);
    nhwc=True their channels-last forms (the [B,F] buffers are then read as [B,HW,C])."""
    from alignq_amd import _lib as L
    lib = L.load()
    st = L.stream_ptr()
    out = {}
    names = ("bn_partial_stats", "site_partials", "site_reduce_loss", "site_bwd_prep", "site_bwd", "bn_bwd_apply")
    per_step = {n: [0.0, 0.0] for n in names}
    multi_sites = []
    A = torch.rand(B, B, device=dev)
    Gm = torch.rand(B, B, device=dev)
    R = 4   # buffer sets in rotation: a launch finds its operands where the training step finds them (written a few launches
    #         ago: out of the 8 x 4 MB L2s, still in the 256 MB Infinity Cache), not L2-hot from the previous identical launch
    for F, count in sorted(site_F_counts.items()):
        HW = hw_of_F[F]
        C = F // HW
        nh = int(bool(nhwc))
        xs = [torch.randn(B, F, device=dev) for _ in range(R)]
        gs = [torch.randn(B, F, device=dev) * 0.01 for _ in range(R)]
        xqs, dxs, dzs = ([torch.empty(B, F, device=dev) for _ in range(R)] for _ in range(3))
        statss = [torch.empty(4, F, device=dev) for _ in range(R)]
        wss = [torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev) for _ in range(R)]
        ws_bns = [torch.empty(lib.alignq_bn_nhwc_ws_bytes(C) if nh else lib.alignq_bn_ws_bytes(C), dtype=torch.uint8, device=dev)
                  for _ in range(R)]
        parts = [torch.empty(lib.alignq_site_bn_part_bytes(F, nh), dtype=torch.uint8, device=dev) for _ in range(R)]
        ress = [torch.randn(B, F, device=dev) for _ in range(R)]       # the block's shortcut (added before the ReLU)
        dress = [torch.empty(B, F, device=dev) for _ in range(R)]
        D = torch.empty(B, B, device=dev)
        S = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)       # fp32 S + its bf16 fragment image (prep writes both)
        scal = torch.empty(4, device=dev)
        dA, dG = torch.empty_like(A), torch.empty_like(Gm)
        one = torch.ones((), device=dev)
        gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        ab, save = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
        dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
        p = L.ptr
        turn = {"stats": 0, "part": 0, "pair": 0, "bwd": 0, "bnb": 0}

        def nxt(name):
            i = turn[name] % R
            turn[name] += 1
            return i

        if folded:
            stats_fn = lib.alignq_bn_partial_stats_nhwc if nh else lib.alignq_bn_partial_stats
            # channels-last: the captured step feeds the fold with the producing convolution's per-workgroup float partials
            # (conv_parts), not with a statistics pass of its own: time THAT entry-point form
            Wd = int(round(HW ** 0.5))
            conv_parts = lib.alignq_conv3x3_bn_parts(B, Wd, Wd, C) if nh else 0
            cparts = []
            if conv_parts > 0:
                for i in range(R):
                    zc = xs[i].view(conv_parts, (B * HW) // conv_parts, C)          # rows of [B*HW, C] split like the conv's tiles
                    cparts.append(torch.stack([zc.sum(1).t(), (zc * zc).sum(1).t()], 2).contiguous())   # [C][parts][2]

            def stats_i(i):
                return stats_fn(p(xs[i]), B, C, HW, p(ws_bns[i]), st)

            def part_i(i, res=False):
                return lib.alignq_site_partials_bn(p(xs[i]), p(cparts[i]) if conv_parts > 0 else p(ws_bns[i]), p(gam), p(bet), p(rm),
                                                   p(rv), p(nbt), 0.1, 1e-5, p(ab), p(save), C, HW, B, F, k, 2.0, 0.0, 1,
                                                   p(ress[i]) if res else None, nh, conv_parts, p(xqs[i]), None, p(statss[i]),
                                                   p(wss[i]), st)

            def bwd_i(i, res=False):
                return lib.alignq_site_bwd_apply_bn(p(gs[i]), p(S), p(xs[i]), p(ab), p(save), C, HW, nh, p(xqs[i]), None, 0,
                                                    p(dress[i]) if res else None, p(statss[i]), B, F, 2.0, 0.0, p(dxs[i]),
                                                    p(parts[i]), st)

            def bnb_i(i):
                return lib.alignq_bn_bwd_apply(p(dxs[i]), p(xs[i]), p(ab), p(save), p(parts[i]), B, C, HW, nh, p(dzs[i]), p(dgam),
                                               p(dbet), st)

            f_stats = lambda: stats_i(nxt("stats"))
            f_bnb = lambda: bnb_i(nxt("bnb"))
        else:
            stats_i = None

            def part_i(i, res=False):
                return lib.alignq_site_partials(p(xs[i]), B, F, k, 2.0, 0.0, p(xqs[i]), p(statss[i]), p(wss[i]), st)

            def bwd_i(i, res=False):
                return lib.alignq_site_bwd_apply(p(gs[i]), p(S), p(xs[i]), p(statss[i]), B, F, 2.0, 0.0, p(dxs[i]), st)

            f_stats = f_bnb = None
        f_part = lambda: part_i(nxt("part"))
        f_bwd = lambda: bwd_i(nxt("bwd"))
        f_part_res = lambda: part_i(nxt("part"), True)
        f_bwd_res = lambda: bwd_i(nxt("bwd"), True)

        def red_i(i):
            return lib.alignq_site_reduce_loss(p(wss[i]), B, F, p(D), p(A), p(Gm), B, 0.2, 0.3, p(scal), st)

        f_prep = lambda: lib.alignq_site_prep_fused(p(D), p(A), p(Gm), B, p(scal), 0.2, p(one), B, F, p(S), p(dA), p(dG), st)

        def fwd_pair():   # the reduction's arrival counter is re-armed by the partials kernel: time them as a pair
            i = nxt("pair")
            part_i(i)
            red_i(i)

        multi_sites.append((F, count, wss, D, S))
        for i in range(R):          # every set holds a finished forward (statistics, x_q, slabs) before anything is timed
            if stats_i:
                stats_i(i)
            part_i(i)
            red_i(i)
        f_prep()
        for i in range(R):
            bwd_i(i)

        # the step's mix: the second site of every residual block adds the shortcut before the ReLU (3 of the 7 sites of a
        # ResNet-20 stage, 9 of 19 for ResNet-56); the other sites have no residual operand
        n_res = (count - 1) // 2 if folded else 0
        t_stats = time_call(f_stats, 50) if f_stats else 0.0
        t_part_plain = time_call(f_part, 50)
        t_red = max(time_call(fwd_pair, 50) - t_part_plain, 0.0)
        t_prep = time_call(f_prep, 50)
        t_bwd_plain = time_call(f_bwd, 50)
        t_part_res = time_call(f_part_res, 50) if n_res else t_part_plain
        t_bwd_res = time_call(f_bwd_res, 50) if n_res else t_bwd_plain
        t_part = (t_part_plain * (count - n_res) + t_part_res * n_res) / count
        t_bwd = (t_bwd_plain * (count - n_res) + t_bwd_res * n_res) / count
        t_bnb = time_call(f_bnb, 50) if f_bnb else 0.0
        gram_flops = 2 * 2.0 * B * B * F            # two Grams, full-matrix count
        out[f"site_F{F}"] = {
            "bn_partial_stats_us": t_stats * 1e6, "partials_us": t_part * 1e6, "reduce_loss_us": t_red * 1e6,
            "bwd_prep_us": t_prep * 1e6, "bwd_us": t_bwd * 1e6, "bn_bwd_apply_us": t_bnb * 1e6,
            "partials_us_plain_residual": [t_part_plain * 1e6, t_part_res * 1e6],
            "bwd_us_plain_residual": [t_bwd_plain * 1e6, t_bwd_res * 1e6], "sites_with_residual": n_res,
            "partials_hbm_gbs": 8.0 * B * F / t_part / 1e9, "bwd_hbm_gbs": 12.0 * B * F / t_bwd / 1e9,
            "partials_tflops_fp32_equiv": gram_flops / t_part / 1e12, "sites": count, "C": C, "HW": HW}
        for name, t, fl in (("bn_partial_stats", t_stats, 0.0), ("site_partials", t_part, gram_flops),
                            ("site_reduce_loss", t_red, 0.0), ("site_bwd_prep", t_prep, 0.0), ("site_bwd", t_bwd, gram_flops),
                            ("bn_bwd_apply", t_bnb, 0.0)):
            per_step[name][0] += t * count
            per_step[name][1] += fl * count
    # ---- the two deferred multi-site launches of the captured step (fused.DeferredLosses): ALL sites' slab reduction + ADMM
    #      loss in one launch after the forward, ALL sites' S / dalterD / dgamma in one launch before the backward
    t_red_multi = t_prep_multi = 0.0
    if 64 < B <= 128:
        wsl, Dl, Al, Gl, scl, Fl, Sl, dAl, dGl = [], [], [], [], [], [], [], [], []
        for F, count, wss, _, _ in multi_sites:
            for j in range(count):
                wsl.append(wss[j % R]); Fl.append(F)
                Dl.append(torch.empty(B, B, device=dev)); scl.append(torch.empty(4, device=dev))
                Al.append(A); Gl.append(Gm)
                Sl.append(torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev))
                dAl.append(torch.empty_like(A)); dGl.append(torch.empty_like(Gm))
        nS = len(wsl)
        one = torch.ones((), device=dev)
        pa, i64 = L.ptr_array, L.i64_array
        args_r = (nS, pa(wsl), pa(Dl), pa(Al), pa(Gl), pa(scl), i64(Fl), B, B, 0.2, 0.3, st)
        args_p = (nS, pa(Dl), pa(Al), pa(Gl), pa(scl), L.ptr(one), i64(Fl), B, B, 0.2, pa(Sl), pa(dAl), pa(dGl), st)
        t_red_multi = time_call(lambda: lib.alignq_site_reduce_loss_multi(*args_r), 30)
        t_prep_multi = time_call(lambda: lib.alignq_site_prep_fused_multi(*args_p), 30)
        del wsl, Dl, scl, Sl, dAl, dGl
    del multi_sites
    # plain CDF-quantise kernels on a roofline-sized tensor (2^26 elements = 268 MB, beyond the 256 MiB L3)
    n = 1 << 26
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    g = torch.randn(n, device=dev)
    p = L.ptr
    t_f = time_call(lambda: lib.alignq_act_quant_fwd(p(x), p(y), None, n, k, 2.0, 0, st), 20)
    t_b = time_call(lambda: lib.alignq_act_quant_bwd(p(g), p(x), p(y), n, 2.0, st), 20)
    for kk in (2, 4):        # k in {2, 4, 8}
        t_k = time_call(lambda: lib.alignq_act_quant_fwd(p(x), p(y), None, n, kk, 2.0, 0, st), 10)
        out[f"act_quant_fwd_2p26_k{kk}"] = {"us": t_k * 1e6, "hbm_gbs": 8.0 * n / t_k / 1e9, "frac_of_8TBs": 8.0 * n / t_k / 1e9 / HBM_PEAK_GBS}
    # N2: the packed-bin forms (int16 bins for the 8-bit ADMM formula): forward 4 B read + 2 B written per element instead of
    # 4 + 4; backward 4 + 4 + 2 read and 4 written instead of 4 + 4 + 4 and 4 (ReLU mask from the bins instead of fp32 y)
    nb = lib.alignq_bin_bytes(k, 2.0, 0)
    if nb:
        bins = torch.empty(n * nb, dtype=torch.uint8, device=dev)
        t_pf = time_call(lambda: lib.alignq_act_quant_fwd_packed(p(x), None, p(bins), n, k, 2.0, 0, 0, st), 10)
        t_pb = time_call(lambda: lib.alignq_act_quant_bwd_packed(p(g), p(x), p(bins), p(y), n, k, 2.0, 0, 1, st), 10)
        t_mb = time_call(lambda: lib.alignq_act_quant_relu_bwd(p(g), p(x), p(x), p(y), n, 2.0, st), 10)
        out["act_quant_packed_2p26"] = {"bin_bytes": nb, "fwd_us": t_pf * 1e6, "fwd_bytes_per_elem": 4 + nb,
                                        "fwd_hbm_gbs": (4.0 + nb) * n / t_pf / 1e9, "bwd_masked_us": t_pb * 1e6,
                                        "bwd_masked_bytes_per_elem": 12 + nb, "bwd_masked_hbm_gbs": (12.0 + nb) * n / t_pb / 1e9,
                                        "bwd_masked_fp32_y_us": t_mb * 1e6, "bwd_masked_fp32_y_bytes_per_elem": 16}
        del bins
    # on-box streaming ceilings with the same traffic shapes: a device copy (1 read + 1 write per element)
    # and an elementwise add (2 reads + 1 write), both plain PyTorch-ROCm kernels
    t_copy = time_call(lambda: y.copy_(x), 20)
    t_add = time_call(lambda: torch.add(g, x, out=y), 20)
    copy_gbs, add_gbs = 8.0 * n / t_copy / 1e9, 12.0 * n / t_add / 1e9
    out["stream_ceiling_2p26"] = {"copy_us": t_copy * 1e6, "copy_gbs": copy_gbs, "add_us": t_add * 1e6, "add_gbs": add_gbs}
    out["act_quant_fwd_2p26"] = {"us": t_f * 1e6, "hbm_gbs": 8.0 * n / t_f / 1e9, "frac_of_8TBs": 8.0 * n / t_f / 1e9 / HBM_PEAK_GBS,
                                 "frac_of_copy_ceiling": t_copy / t_f,
                                 "note": "one x / y pair re-used by every launch (SURVEY 8d protocol): the memory-side cache keeps part "
                                         "of it between launches for these non-temporal readers (not for torch's copy); the figure to "
                                         "hold against HBM is cold_operands_2p26"}
    out["act_quant_bwd_2p26"] = {"us": t_b * 1e6, "hbm_gbs": 12.0 * n / t_b / 1e9, "frac_of_8TBs": 12.0 * n / t_b / 1e9 / HBM_PEAK_GBS,
                                 "frac_of_add_ceiling": t_add / t_b}
    # cold operands: the launches above re-use one x / g / y triple (805 MB against a 256 MB memory-side cache that keeps part of
    # it from launch to launch); here inputs AND outputs rotate over 4 sets, kernels and ceilings alike
    R = 4
    xs = [x] + [torch.randn(n, device=dev) for _ in range(R - 1)]
    gs = [g] + [torch.randn(n, device=dev) for _ in range(R - 1)]
    ys = [y] + [torch.empty(n, device=dev) for _ in range(R - 1)]
    tc_f = time_call_rot(lambda i: lib.alignq_act_quant_fwd(p(xs[i]), p(ys[i]), None, n, k, 2.0, 0, st), 20, R)
    tc_b = time_call_rot(lambda i: lib.alignq_act_quant_bwd(p(gs[i]), p(xs[i]), p(ys[i]), n, 2.0, st), 20, R)
    tc_copy = time_call_rot(lambda i: ys[i].copy_(xs[i]), 20, R)
    tc_add = time_call_rot(lambda i: torch.add(gs[i], xs[i], out=ys[i]), 20, R)
    out["cold_operands_2p26"] = {"sets": R, "act_quant_fwd_us": tc_f * 1e6, "act_quant_fwd_frac_of_8TBs": 8.0 * n / tc_f / 1e9 / HBM_PEAK_GBS,
                                 "act_quant_bwd_us": tc_b * 1e6, "act_quant_bwd_frac_of_8TBs": 12.0 * n / tc_b / 1e9 / HBM_PEAK_GBS,
                                 "copy_us": tc_copy * 1e6, "copy_gbs": 8.0 * n / tc_copy / 1e9, "add_us": tc_add * 1e6,
                                 "add_gbs": 12.0 * n / tc_add / 1e9,
                                 "note": "same kernels and torch ceilings with 4 rotating sets of every operand (3.2 GB touched "
                                         "between two uses of a buffer)"}
    del x, y, g, xs, gs, ys
    if shapes:
        out["roofline_shapes"] = measure_roofline_shapes(dev, k)
        out["office_shapes"] = measure_office_shapes(dev, k)
        out["corr_large"] = measure_corr_large(dev, k)
    n_sites = sum(site_F_counts.values())
    dom = max(("site_partials", "site_bwd"), key=lambda kname: per_step[kname][0])
    t_sum, fl_sum = per_step[dom]
    kernel_sym = {"site_partials": "site_fwd4_kernel<TF,true> (BN+ReLU folded)" if folded else "site_fwd4_kernel<TF,true>",
                  "site_bwd": "site_bwd4_kernel<TF,true,true> (BN+ReLU folded)" if folded else "site_bwd4_kernel<true,false>"}[dom]
    bytes_per_elem = 8.0 if dom == "site_partials" else 12.0          # fwd read x + write x_q; bwd read g,x + write dx
    by_sum = sum(bytes_per_elem * B * F * cnt for F, cnt in site_F_counts.items())
    traffic = None
    try:   # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 corrected + WRITE_SIZE)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch_avg")
    except Exception:
        pass
    # Since the Gram pair runs on split-bf16 MFMA (3 x v_mfma_f32_32x32x16_bf16 per 16 features: ~1.3 us of a
    # 10-15 us launch) the matrix pipe no longer bounds these kernels: the irreducible work is the HBM traffic.
    roofline = {"kernel": kernel_sym, "bound": "hbm", "achieved": by_sum / t_sum / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": by_sum / t_sum / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                  "2 x FETCH_SIZE + WRITE_SIZE per launch; a stored figure, not measured in this run)",
                "launches_per_step": n_sites, "avg_launch_us": t_sum / n_sites * 1e6,
                "bytes_per_launch_avg": by_sum / n_sites,
                "gram_tflops_fp32_equiv": fl_sum / t_sum / 1e12,
                "note": f"achieved = algorithmic bytes (8 B/elem fwd, 12 B/elem bwd, SURVEY 8d; the fused shortcut operand is "
                        f"not counted) summed over the step's {n_sites} sites / summed launch time (HIP events, back-to-back "
                        "launches over 4 rotating buffer sets, the step's mix of sites with and without a residual operand); "
                        "CIFAR sites are 2-8 MB per launch, i.e. launch-latency shapes (SURVEY H2); see kernels.act_quant_* "
                        "for the HBM-roofline-sized CDF-quantise kernel"}
    # what the CAPTURED step launches (tools/count_step_kernels.sh lists the same kernels from a rocprofv3 trace of the graph):
    # one folded site forward and one folded site backward per site, ONE slab_reduce_multi per step, the site preparation as a
    # role of the head-backward launch.
    # The per-site reduce / prep / statistics / bn_bwd_apply entry points are launched by the eager per-module API only.
    out["per_step_us"] = {"site_partials": per_step["site_partials"][0] * 1e6, "site_bwd": per_step["site_bwd"][0] * 1e6,
                          "slab_reduce_multi": t_red_multi * 1e6, "site_prep_multi": t_prep_multi * 1e6,
                          "note": "kernels of the captured step (graph); forward timed in its conv_parts form; in the step the "
                                  "site preparation is one role of the launch that also runs the classifier head's backward "
                                  "(alignq_head_ce_bwd_site_prep), timed here stand-alone"}
    out["eager_only_per_step_us"] = {kname: per_step[kname][0] * 1e6 for kname in
                                     ("bn_partial_stats", "site_reduce_loss", "site_bwd_prep", "bn_bwd_apply")}
    # Gram step on the matrix cores (north_star: "MFMA utilisation for the Gram step against gfx950 peak"): the forward issues
    # 6 v_mfma_f32_32x32x16_bf16 (3-term split, T and X Gram) per 16 features for each of the 10 upper-triangular 32x32 tiles;
    # one such MFMA is 32768 flop and occupies a SIMD's matrix pipe for 32 cycles (2.5 PFLOP/s dense bf16 / 1024 SIMDs / 2.4 GHz)
    n_mfma = sum(6 * 10 * (F // 16) * cnt for F, cnt in site_F_counts.items())
    t_fwd_sum = per_step["site_partials"][0]
    roofline["gram_mfma"] = {"bf16_mfma_per_step": n_mfma, "bf16_tflops_issued": n_mfma * 32768 / t_fwd_sum / 1e12,
                             "frac_of_bf16_peak": n_mfma * 32768 / t_fwd_sum / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                             "mfma_pipe_busy_frac": n_mfma * 32 / (1024 * 2.4e9) / t_fwd_sum,
                             "note": "issued bf16 MFMA flop of the site forward launches / their summed time; the Gram is 5-6 % of "
                                     "these latency- and VALU-bound launches"}
    return roofline, out


def measure_roofline_shapes(dev, k):
    """SURVEY.md §8d microbench shapes beyond the CIFAR sites: the roofline-sized [128, 524288] site (2^26 elements), the
    config-5 (Office, batch 28) sites [28, 802816] and [28, 100352] with the eps corr, and the weight quantiser on a
    ResNet-50-sized filter [512, 512, 3, 3].  HIP events on the launch stream; GB/s from the ALGORITHMIC bytes
    (8 B/element forward, 12 B/element backward; weights 28 B/element forward incl. cdf/pdf, 12 B backward)."""
    from alignq_amd import _lib as L
    lib = L.load()
    st = L.stream_ptr()
    p = L.ptr
    out = {}
    for B, F, eps in ((128, 524288, 0.0), (28, 802816, 1e-5), (28, 100352, 1e-5)):
        x = torch.randn(B, F, device=dev)
        g = torch.randn(B, F, device=dev) * 0.01
        xq, dx = torch.empty_like(x), torch.empty_like(x)
        stats = torch.empty(4, F, device=dev)
        ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
        D, A, Gm = torch.empty(B, B, device=dev), torch.rand(B, B, device=dev), torch.rand(B, B, device=dev)
        scal, one = torch.empty(4, device=dev), torch.ones((), device=dev)
        S = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev)
        dA, dG = torch.empty_like(A), torch.empty_like(Gm)
        f_fwd = lambda: lib.alignq_site_partials(p(x), B, F, k, 2.0, eps, p(xq), p(stats), p(ws), st)
        f_fwd()
        lib.alignq_site_reduce_loss(p(ws), B, F, p(D), p(A), p(Gm), B, 0.2, 0.3, p(scal), st)
        lib.alignq_site_prep_fused(p(D), p(A), p(Gm), B, p(scal), 0.2, p(one), B, F, p(S), p(dA), p(dG), st)
        f_bwd = lambda: lib.alignq_site_bwd_apply(p(g), p(S), p(x), p(stats), B, F, 2.0, eps, p(dx), st)
        t_f, t_b = time_call(f_fwd, 10), time_call(f_bwd, 10)
        n = B * F
        out[f"site_{B}x{F}"] = {"fwd_us": t_f * 1e6, "fwd_hbm_gbs": 8.0 * n / t_f / 1e9, "fwd_frac_of_8TBs": 8.0 * n / t_f / 1e9 / HBM_PEAK_GBS,
                                "bwd_us": t_b * 1e6, "bwd_hbm_gbs": 12.0 * n / t_b / 1e9, "bwd_frac_of_8TBs": 12.0 * n / t_b / 1e9 / HBM_PEAK_GBS,
                                "mbytes": 4.0 * n / 1e6, "eps": eps}
        # the same launches over ROTATING operand sets (>= 1 GB touched between two uses of a buffer): the figures above re-read
        # one x / g pair, part of which the 256 MB memory-side cache still holds from the previous launch; these do not
        sets = max(3, int(1.2e9 / (8.0 * n)) + 1)
        xs = [x] + [torch.randn(B, F, device=dev) for _ in range(sets - 1)]
        gs = [g] + [torch.randn(B, F, device=dev) * 0.01 for _ in range(sets - 1)]
        t_fc = time_call_rot(lambda i: lib.alignq_site_partials(p(xs[i]), B, F, k, 2.0, eps, p(xq), p(stats), p(ws), st), 12, sets)
        t_bc = time_call_rot(lambda i: lib.alignq_site_bwd_apply(p(gs[i]), p(S), p(xs[i]), p(stats), B, F, 2.0, eps, p(dx), st), 12, sets)
        t_fs, t_bs = time_call_serial(f_fwd, 8), time_call_serial(f_bwd, 8)
        out[f"site_{B}x{F}"].update({"fwd_us_serial": t_fs * 1e6, "bwd_us_serial": t_bs * 1e6,
                                     "serial_note": "each launch alone between two synchronisations (what a rocprofv3 kernel trace "
                                                    "sees); *_us: 10 launches back to back"})
        out[f"site_{B}x{F}"].update({"operand_sets_cold": sets, "fwd_us_cold": t_fc * 1e6, "fwd_frac_of_8TBs_cold": 8.0 * n / t_fc / 1e9 / HBM_PEAK_GBS,
                                     "bwd_us_cold": t_bc * 1e6, "bwd_frac_of_8TBs_cold": 12.0 * n / t_bc / 1e9 / HBM_PEAK_GBS})
        del x, g, xq, dx, stats, ws, xs, gs
    # weights: one ResNet-50 layer4 3x3 filter
    nw = 512 * 512 * 9
    w = torch.randn(nw, device=dev) * 0.05
    gq = torch.randn(nw, device=dev)
    q, c, pdf, dw = (torch.empty_like(w) for _ in range(4))
    ms = torch.empty(2, device=dev)
    wsw = torch.empty(lib.alignq_weight_ws_bytes(nw), dtype=torch.uint8, device=dev)

    def w_fwd():
        lib.alignq_weight_stats(p(w), nw, p(ms), p(wsw), st)
        lib.alignq_weight_quant_fwd(p(w), p(ms), p(q), p(c), p(pdf), None, nw, k, 0, st)
    t_wf = time_call(w_fwd, 20)
    t_wb = time_call(lambda: lib.alignq_weight_quant_bwd(p(gq), p(w), p(ms), p(dw), nw, p(wsw), st), 20)
    out["weight_512x512x3x3"] = {"fwd_us": t_wf * 1e6, "fwd_hbm_gbs": 20.0 * nw / t_wf / 1e9, "fwd_note": "stats (2 launches) + "
                                 "quantise: reads w twice (stats, apply), writes q, cdf, pdf = 20 B/element",
                                 "bwd_us": t_wb * 1e6, "bwd_hbm_gbs": 20.0 * nw / t_wb / 1e9, "bwd_note": "two passes: reads g, w "
                                 "twice, writes dw = 20 B/element", "mbytes": 4.0 * nw / 1e6}
    return out


def measure_office_shapes(dev, k):
    """Configuration 5's in-scope chains at the network's own shapes, as OfficeTrainStep(dual=True) launches
    them: channels-last tensors holding the source and the target batch back to back (groups = 2 x 28 samples).
      bnq_*    : `relu(act_q(bn(z)))` of the plain sites (dann_office/model/resnet.py:134-143, stem :230-233) folded:
                 alignq_bnq_fwd (statistics pass + apply pass: 4 + 8 = 12 B/element, + 1 bit of ReLU mask) and alignq_bnq_bwd (sums pass
                 g, z, mask + apply pass g, z, mask -> dz: 8 + 12 + 2 bits = 20.25 B/element; with the mask read from the fp32 y: 28); `stats_us` is the statistics pass alone (alignq_bnq_stats), apply = chain - stats.
      site_*   : the bottleneck tail `relu(act_q3(bn3(z))[0] + identity)` (resnet.py:146-154) at layer4's [28, 100352] and layer1's
                 [28, 802816]: forward alignq_bnq_stats + alignq_site1_groups_fwd + alignq_site1_groups_reduce_loss (reads z twice and the
                 residual, writes y: 16 B/element), backward alignq_site_prep_fused_multi + alignq_site1_groups_bwd + alignq_bnq_bwd_dx
                 (alignq_site1_groups_bwd_bn: g, y, z -> dx, dres and per-column batch-norm sums; then dx, z -> dz: 32 B/element;
                 `bwd_us_with_sums_pass`: alignq_site1_groups_bwd + alignq_bnq_bwd_dx with its own pass over dx and z, 40 B/element).
    HIP events on the launch stream, 4 rotating operand sets (a launch finds its operands where the step finds them)."""
    from alignq_amd import _lib as L
    lib = L.load()
    st = L.stream_ptr()
    p = L.ptr
    out = {}
    R, G = 4, 2
    for C, H in ((256, 56), (64, 112)):
        Bt = 28 * G
        P = 28 * H * H
        n = Bt * C * H * H
        zs = [torch.randn(Bt, H, H, C, device=dev) * 1.3 + 0.2 for _ in range(R)]
        gs = [torch.randn(Bt, H, H, C, device=dev) * 0.01 for _ in range(R)]
        ys, dzs = ([torch.empty(Bt, H, H, C, device=dev) for _ in range(R)] for _ in range(2))
        gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        ab, save = torch.empty(G, 2, C, device=dev), torch.empty(G, 2, C, device=dev)
        dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(lib.alignq_bnq_ws_bytes(C, G), dtype=torch.uint8, device=dev)
        masks = [torch.empty(lib.alignq_bnq_mask_bytes(P, C, G), dtype=torch.uint8, device=dev) for _ in range(R)]

        def f_fwd(i):
            L.check(lib.alignq_bnq_fwd(p(zs[i]), P, C, G, p(gam), p(bet), p(rm), p(rv), p(nbt), 0.1, 1e-5, k, 2.0, 0, 1, None, p(ab), p(save),
                                       p(ys[i]), p(masks[i]), p(ws), st), "alignq_bnq_fwd")

        def f_stats(i):
            L.check(lib.alignq_bnq_stats(p(zs[i]), P, C, G, p(gam), p(bet), p(rm), p(rv), p(nbt), 0.1, 1e-5, p(ab), p(save), p(ws), st),
                    "alignq_bnq_stats")

        def f_bwd(i):
            L.check(lib.alignq_bnq_bwd(p(gs[i]), p(zs[i]), None, p(masks[i]), p(ab), p(save), P, C, G, 2.0, 1, p(dzs[i]), None, p(dgam), p(dbet),
                                       p(ws), st), "alignq_bnq_bwd")

        def f_bwd_y(i):           # round 3's form: the ReLU mask from the fp32 y (28 B/element)
            L.check(lib.alignq_bnq_bwd(p(gs[i]), p(zs[i]), p(ys[i]), None, p(ab), p(save), P, C, G, 2.0, 1, p(dzs[i]), None, p(dgam), p(dbet),
                                       p(ws), st), "alignq_bnq_bwd")
        for i in range(R):
            f_fwd(i)
        t_f, t_s, t_b = time_call_rot(f_fwd, 20, R), time_call_rot(f_stats, 20, R), time_call_rot(f_bwd, 20, R)
        t_by = time_call_rot(f_bwd_y, 20, R)
        t_a = max(t_f - t_s, 1e-9)
        out[f"bnq_{Bt}x{C}x{H}x{H}"] = {
            "elements": n, "fwd_us": t_f * 1e6, "fwd_hbm_gbs": 12.0 * n / t_f / 1e9, "fwd_frac_of_8TBs": 12.0 * n / t_f / 1e9 / HBM_PEAK_GBS,
            "stats_us": t_s * 1e6, "stats_frac_of_8TBs": 4.0 * n / t_s / 1e9 / HBM_PEAK_GBS,
            "apply_fwd_us": t_a * 1e6, "apply_fwd_frac_of_8TBs": 8.0 * n / t_a / 1e9 / HBM_PEAK_GBS,
            "bwd_us": t_b * 1e6, "bwd_hbm_gbs": 20.25 * n / t_b / 1e9, "bwd_frac_of_8TBs": 20.25 * n / t_b / 1e9 / HBM_PEAK_GBS,
            "bwd_us_mask_from_fp32_y": t_by * 1e6, "fwd_bytes_per_elem": 12.125, "bwd_bytes_per_elem": 20.25}
        del zs, gs, ys, dzs
    for C, H in ((2048, 7), (256, 56)):
        B, Bt = 28, 28 * G
        F, P = C * H * H, 28 * H * H
        n = Bt * F
        zs = [torch.randn(Bt, H, H, C, device=dev) * 1.1 + 0.15 for _ in range(R)]
        rs = [torch.relu(torch.randn(Bt, H, H, C, device=dev)) for _ in range(R)]
        gs = [torch.randn(Bt, H, H, C, device=dev) * 0.01 for _ in range(R)]
        ys, dxs, dress = ([torch.empty(Bt, H, H, C, device=dev) for _ in range(R)] for _ in range(3))
        gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        ab, save = torch.empty(G, 2, C, device=dev), torch.empty(G, 2, C, device=dev)
        dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws_bn = torch.empty(lib.alignq_bnq_ws_bytes(C, G), dtype=torch.uint8, device=dev)
        stats = torch.empty(G, 4, F, device=dev)
        D, scal = torch.empty(G, B, B, device=dev), torch.empty(G, 4, device=dev)
        A, Gm = torch.rand(B, B, device=dev), torch.rand(B, B, device=dev)
        dA, dG = torch.empty(G, B, B, device=dev), torch.empty(G, B, B, device=dev)
        one = torch.ones((), device=dev)
        wsb = lib.alignq_site_ws_bytes(B, F)
        ws = torch.empty(wsb * G, dtype=torch.uint8, device=dev)
        cols = torch.empty(lib.alignq_site1_cols_bytes(F, G), dtype=torch.uint8, device=dev)
        rmask = torch.empty(lib.alignq_site1_mask_bytes(B, F, G), dtype=torch.uint8, device=dev)       # the forward's one-bit ReLU mask
        sb = lib.alignq_site_bwd_ws_bytes(B)
        S = torch.empty(sb * G, dtype=torch.uint8, device=dev)
        Sg = [S[i * sb:(i + 1) * sb] for i in range(G)]
        prep_args = (G, L.ptr_array([D[i] for i in range(G)]), L.ptr_array([A] * G), L.ptr_array([Gm] * G),
                     L.ptr_array([scal[i] for i in range(G)]), p(one), L.i64_array([F] * G), B, B, 0.2, L.ptr_array(Sg),
                     L.ptr_array([dA[i] for i in range(G)]), L.ptr_array([dG[i] for i in range(G)]), st)

        def s_fwd(i):
            L.check(lib.alignq_bnq_stats(p(zs[i]), P, C, G, p(gam), p(bet), p(rm), p(rv), p(nbt), 0.1, 1e-5, p(ab), p(save), p(ws_bn), st),
                    "alignq_bnq_stats")
            L.check(lib.alignq_site1_groups_fwd_m(p(zs[i]), p(ab), C, B, F, G, k, 2.0, 1e-5, p(rs[i]), 1, p(ys[i]), p(stats), p(ws), p(rmask), st),
                    "alignq_site1_groups_fwd_m")
            L.check(lib.alignq_site1_groups_reduce_loss(p(ws), B, F, G, p(D), p(A), p(Gm), B, 0.2, 0.3, p(scal), st),
                    "alignq_site1_groups_reduce_loss")

        def s_fwd_site(i):
            L.check(lib.alignq_site1_groups_fwd_m(p(zs[i]), p(ab), C, B, F, G, k, 2.0, 1e-5, p(rs[i]), 1, p(ys[i]), p(stats), p(ws), p(rmask), st),
                    "alignq_site1_groups_fwd_m")

        def s_bwd(i):
            L.check(lib.alignq_site_prep_fused_multi(*prep_args), "alignq_site_prep_fused_multi")
            L.check(lib.alignq_site1_groups_bwd(p(gs[i]), None, p(ys[i]), p(S), p(zs[i]), p(ab), C, p(stats), B, F, G, 2.0, 1e-5, p(dxs[i]),
                                                p(dress[i]), st), "alignq_site1_groups_bwd")
            L.check(lib.alignq_bnq_bwd_dx(p(dxs[i]), p(zs[i]), p(ab), p(save), P, C, G, p(dxs[i]), p(dgam), p(dbet), p(ws_bn), st),
                    "alignq_bnq_bwd_dx")

        def s_bwd_cols(i):        # round 4 (the step's path): the site backward leaves per-column sums for the batch-norm backward
            L.check(lib.alignq_site_prep_fused_multi(*prep_args), "alignq_site_prep_fused_multi")
            L.check(lib.alignq_site1_groups_bwd_bn_m(p(gs[i]), None, p(rmask), p(S), p(zs[i]), p(ab), p(save), C, p(stats), B, F, G, 2.0, 1e-5,
                                                     p(dxs[i]), p(dress[i]), p(dgam), p(dbet), p(cols), p(ws_bn), st),
                    "alignq_site1_groups_bwd_bn_m")

        def s_bwd_site(i):
            L.check(lib.alignq_site1_groups_bwd(p(gs[i]), None, p(ys[i]), p(S), p(zs[i]), p(ab), C, p(stats), B, F, G, 2.0, 1e-5, p(dxs[i]),
                                                p(dress[i]), st), "alignq_site1_groups_bwd")
        for i in range(R):
            s_fwd(i)       # (the statistics of the LAST set stay in `stats` for the backward launches: timing only)
        t_f, t_fs = time_call_rot(s_fwd, 20, R), time_call_rot(s_fwd_site, 20, R)
        t_b5, t_bs = time_call_rot(s_bwd, 20, R), time_call_rot(s_bwd_site, 20, R)
        t_b = time_call_rot(s_bwd_cols, 20, R)
        out[f"bn_site_2x{B}x{F}"] = {
            "elements": n, "fwd_us": t_f * 1e6, "fwd_hbm_gbs": 16.0 * n / t_f / 1e9, "fwd_frac_of_8TBs": 16.0 * n / t_f / 1e9 / HBM_PEAK_GBS,
            "site_fwd_kernel_us": t_fs * 1e6, "site_fwd_kernel_frac_of_8TBs": 12.0 * n / t_fs / 1e9 / HBM_PEAK_GBS,
            "bwd_us": t_b * 1e6, "bwd_hbm_gbs": 28.125 * n / t_b / 1e9, "bwd_frac_of_8TBs": 28.125 * n / t_b / 1e9 / HBM_PEAK_GBS,
            "bwd_us_with_sums_pass": t_b5 * 1e6,
            "site_bwd_kernel_us": t_bs * 1e6, "site_bwd_kernel_frac_of_8TBs": 20.0 * n / t_bs / 1e9 / HBM_PEAK_GBS,
            "fwd_bytes_per_elem": 16.125, "bwd_bytes_per_elem": 28.125,
            "note": "bwd (round 5): the ReLU mask from the forward's one-bit mask instead of y; site_*_kernel: alignq_site1_groups_fwd (z, residual -> y: 12 B/element) / alignq_site1_groups_bwd (g, y, z -> dx, "
                    "dres: 20 B/element) alone"}
        del zs, rs, gs, ys, dxs, dress
    return out


def measure_corr_large(dev, k):
    """what the rows above 128 cost.  corr(x, x) on the blocked exact-fp32 Gram (corr_large_kernels.hip,
    v_mfma_f32_32x32x2_f32) at B in {256, 1024}, F = 16384: forward / backward time, fp32-MFMA TFLOP/s against the 157 TFLOP/s
    dense fp32-matrix peak.  The flop are the ones the matrix pipe EXECUTES: forward 2 * 128^2 * F per upper-triangular block pair
    (nb (nb + 1) / 2 pairs, nb = B / 128: G is symmetric), backward 2 B^2 F (ONE contraction with the symmetrised S = dG + dG^T;
    rounds before 4 counted 4 B^2 F here, i.e. two products, and reported a fraction above 1).  HBM fraction on the algorithmic bytes
    (forward 4 B/element, backward 8); and the ADMM site at B = 256 (ops.site_unfused: round 4's pair kernels on the blocked Gram -
    ops.SiteLargeFn - plus the ADMM loss, autograd backward) against the FUSED site at B = 128 run twice on the same 256 rows."""
    from alignq_amd import _lib as L, ops
    from alignq_amd.admm import ADMM
    lib = L.load()
    st = L.stream_ptr()
    p = L.ptr
    out = {}
    F = 16384
    for B in (256, 1024):
        x = torch.randn(B, F, device=dev)
        dG = torch.randn(B, B, device=dev) * 1e-3
        G, stats, dx = torch.empty(B, B, device=dev), torch.empty(2, F, device=dev), torch.empty(B, F, device=dev)
        ws = torch.empty(lib.alignq_site_ws_bytes(B, F), dtype=torch.uint8, device=dev)
        wsb = torch.empty(lib.alignq_site_bwd_ws_bytes(B), dtype=torch.uint8, device=dev)
        f_fwd = lambda: L.check(lib.alignq_corr_fwd(p(x), B, F, 0.0, p(G), p(stats), p(ws), st), "alignq_corr_fwd")      # noqa: E731
        f_bwd = lambda: L.check(lib.alignq_corr_bwd(p(dG), p(x), p(stats), B, F, 0.0, p(dx), p(wsb), st), "alignq_corr_bwd")  # noqa: E731
        f_fwd()
        t_f, t_b = time_call(f_fwd, 20), time_call(f_bwd, 20)
        n = B * F
        nb = (B + 127) // 128
        fl_f, fl_b = nb * (nb + 1) // 2 * 2.0 * 128 * 128 * F, 2.0 * B * B * F
        out[f"corr_{B}x{F}"] = {
            "fwd_us": t_f * 1e6, "fwd_tflops_fp32": fl_f / t_f / 1e12, "fwd_frac_of_157_tflops": fl_f / t_f / 1e12 / 157.0,
            "fwd_frac_of_8TBs": 4.0 * n / t_f / 1e9 / HBM_PEAK_GBS,
            "bwd_us": t_b * 1e6, "bwd_tflops_fp32": fl_b / t_b / 1e12, "bwd_frac_of_157_tflops": fl_b / t_b / 1e12 / 157.0,
            "bwd_frac_of_8TBs": 8.0 * n / t_b / 1e9 / HBM_PEAK_GBS}
        del x, dx, ws
    # the ADMM site at 256 rows: composed (the only form above 128 rows) vs two fused 128-row sites
    B = 256
    x = torch.randn(B, F, device=dev)
    gq = torch.randn(B, F, device=dev) * 0.01
    admm = ADMM(B).to(dev)
    one = torch.ones((), device=dev)

    def composed():
        xi = x.detach().requires_grad_(True)
        xq, loss, _ = ops.site_unfused(xi, admm, k, 2.0, 0.0, 0)
        torch.autograd.backward([xq, loss], [gq, one])
        admm.alterD.grad = admm.gamma.grad = None
    admm1 = ADMM(128).to(dev)

    def fused_twice():
        for h in range(2):
            xi = x[h * 128:(h + 1) * 128].detach().requires_grad_(True)
            xq, loss, _ = ops.SiteFn.apply(xi, admm1.alterD, admm1.gamma, k, 2.0, 0.0, admm1.mu, admm1.rho)
            torch.autograd.backward([xq, loss], [gq[h * 128:(h + 1) * 128], one])
        admm1.alterD.grad = admm1.gamma.grad = None
    def graphed(fn):                      # device time only: the launches of one call captured into a HIP graph, replays timed
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        return time_call(gr.replay, 20)
    t_ce, t_fe = time_call(composed, 10), time_call(fused_twice, 10)
    t_c, t_f2 = graphed(composed), graphed(fused_twice)
    out["site_256x16384"] = {"composed_fwd_bwd_us": t_c * 1e6, "fused_128_rows_twice_fwd_bwd_us": t_f2 * 1e6, "ratio": t_c / t_f2,
                             "composed_eager_us": t_ce * 1e6, "fused_twice_eager_us": t_fe * 1e6,
                             "note": "HIP-graph replays of the autograd functions' launches (device time); *_eager_us: the same "
                                     "calls launched eagerly (host-bound).  The two halves are NOT the same computation as the "
                                     "256-row site (their correlation matrices are 128 x 128): a cost yardstick per element only"}
    return out


def cpu_baseline(batch, bits, model, steps, tree="admm", thread_counts=(8, 16, 32, 64, 128)):
    """The eager-torch restatement of the reference on the host cores: same workload, bounded sample.  Oversubscribing
    torch's intra-op pool hurts this elementwise-heavy workload (64 threads ran slower than 16 on the GPU box), so a sweep
    of thread counts {8, 16, 32, 64, 128} (those the box offers) is timed, >= 10 steps each, and the FASTEST is reported
    with the whole list."""
    from oracle import torch_ref as R
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cfg = R.Config(tree=tree, bitW=bits, abitW=bits, train_batch_size=batch)
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(batch, 3, 32, 32, generator=gen)
    y = torch.randint(0, 10, (batch,), generator=gen)
    best = None
    tried = {}
    for cores in sorted({max(1, min(avail, c)) for c in thread_counts}):
        torch.set_num_threads(cores)
        torch.manual_seed(0)
        net = (R.resnet20 if model == "resnet20" else R.resnet56)(cfg).train()
        step = R.TrainStep(net, cfg)
        for _ in range(2):
            step(x, y)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step(x, y)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        tried[str(cores)] = batch / med
        if best is None or med < best[1]:
            best = (cores, med)
    cores, med = best
    return {"value": batch / med, "unit": "images/sec", "cores": cores, "host_cores": os.cpu_count(), "usable_cores": avail,
            "kind": "port",
            "sample": f"{steps} full training steps (median) of the same workload after 2 warm-ups per thread count, batch "
                      f"{batch}, oracle/torch_ref.py on torch-CPU {torch.__version__}; best of the thread sweep",
            "thread_sweep_images_per_sec": tried, "s_per_step": med}


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator is created; the driver parses stdout for the one
    JSON line, so file descriptor 1 points at stderr while process groups are set up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def office_dp_probe(dev, steps=12):
    """Configuration 5's data-parallel form (dp.attach_office: >= 4 gradient buckets all-reduced from backward hooks, D in the
    last) at world size 1, inside the process group dp_probe has open: what one of the 8 ranks executes per step
    minus the wire time."""
    import gc
    from alignq_amd import config, dp
    from alignq_amd.resnet_office import resnet50_dann
    from alignq_amd.train_step import OfficeTrainStep
    saved = (config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size)
    try:
        config.args.bitW = config.args.abitW = 8
        config.args.train_batch_size = config.args.eval_batch_size = 28
        torch.manual_seed(0)
        model = resnet50_dann(8, 8).to(dev).train()
        step = OfficeTrainStep(model, lr=0.004, channels_last=True)
        hook = dp.attach_office(step, force=True)
        xs, xt = torch.randn(28, 3, 224, 224, device=dev), torch.randn(28, 3, 224, 224, device=dev)
        ys = torch.randint(0, 31, (28,), device=dev)
        step.capture(xs, ys, xt, warmup=2)
        for _ in range(3):
            step(xs, ys, xt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(xs, ys, xt)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out = {"ms_per_step": ms, "buckets_mib": [int(b.flat.numel()) * 4 // 2 ** 20 for b, _ in hook._phase], "world_size": 1}
        del step, model, hook
    except Exception as e:
        out = {"error": repr(e)[:200]}
    finally:
        config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size = saved
        gc.collect()
        torch.cuda.empty_cache()
    return out


def dp_probe(dev, a, steps=50):
    """ms per step of the DATA-PARALLEL form of the same step at world size 1 (its own model, after the main measurement)."""
    import torch.distributed as dist
    from alignq_amd import dp
    from alignq_amd.resnet import resnet20_quant, resnet56_quant
    from alignq_amd.train_step import TrainStep
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            dist.barrier()                   # creates the communicator (and prints RCCL's banner) here
        torch.manual_seed(0)
        model = (resnet20_quant if a.model == "resnet20" else resnet56_quant)(a.bits, a.bits).to(dev).train()
        step = TrainStep(model, fuse_bn=not a.no_fuse_bn, channels_last=not a.nchw, qconv=not a.no_qconv)
        hook = dp.attach(step, force=True)
        x = torch.randn(a.batch, 3, 32, 32, device=dev)
        y = torch.randint(0, 10, (a.batch,), device=dev)
        step.capture(x, y, warmup=3)
        for _ in range(5):
            step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(x, y)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out = {"ms_per_step": ms, "bucket_bytes": int(hook.bucket.flat.numel()) * 4, "world_size": 1, "backend": "nccl (RCCL)",
               "note": "two HIP graphs (fwd+bwd+pack | unpack+optimizers) with one eager RCCL AVG all-reduce of the flat bucket"}
        del step, model, hook
        if a.model == "resnet20" and a.bits == 8 and not a.no_other_configs:
            out["office"] = office_dp_probe(dev)
    except Exception as e:           # never fail the headline line on the probe
        out = {"error": repr(e)[:200]}
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    return out


# Algorithmic HBM bytes per GPU and step of the in-scope kernels (20 B per activation element of a
# quantiser site; weights 20-28 B each, quantised once per pass)
_CFG_BYTES = {"resnet20": 128 * 200704 * 20 + 0.27e6 * 28, "resnet20_cdf": 128 * 200704 * 20 + 0.27e6 * 20, "resnet56": 128 * 544768 * 20 + 0.85e6 * 28,
              "resnet50_dann": 2 * 28 * 9608704 * 20 + 2 * 23.5e6 * 20 + 23.5e6 * 8}


def other_configs(dev, a, steps=30, only=None):
    """BASELINE.json's other configurations through the same code, one GPU's share each, short captured runs (the headline
    line above stays configs[1]): configs[2] ResNet-20 2W/2A (batch 1024 / 8 GPUs = 128 per GPU), configs[3] ResNet-56 4W/4A
    (512 / 4 = 128), configs[4] ResNet-50-DANN Office-31 8W/8A (224 / 8 = 28, source + target pass).  Iteration order of the
    reference: cdf_alignment_admm/resnet-20-cifar-10/main.py:288-378, dann_office/main.py:343-456."""
    import gc
    from alignq_amd import config
    from alignq_amd.resnet import resnet20_quant, resnet56_quant
    from alignq_amd.resnet_office import resnet50_dann
    from alignq_amd.train_step import OfficeTrainStep, TrainStep
    saved = (config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size)
    out = {}
    gen = torch.Generator().manual_seed(1)
    for name, kind, bits, batch in (("resnet20_cdf_only_8w8a_b128", "resnet20_cdf", 8, 128), ("resnet20_2w2a_b128", "resnet20", 2, 128),
                                    ("resnet56_4w4a_b128", "resnet56", 4, 128), ("resnet50_dann_8w8a_b28", "resnet50_dann", 8, 28)):
        if only is not None and name not in only:
            continue
        try:
            config.args.bitW = config.args.abitW = bits
            config.args.train_batch_size = config.args.eval_batch_size = batch
            torch.manual_seed(0)
            if kind == "resnet50_dann":
                model = resnet50_dann(bits, bits).to(dev).train()
                st = OfficeTrainStep(model, lr=0.004, channels_last=True)
                xs = torch.randn(batch, 3, 224, 224, generator=gen).to(dev)
                xt = torch.randn(batch, 3, 224, 224, generator=gen).to(dev)
                ys = torch.randint(0, 31, (batch,), generator=gen).to(dev)
                st.capture(xs, ys, xt, warmup=2)
                run = lambda: st(xs, ys, xt)                         # noqa: E731
                images = 2 * batch
            else:
                if kind == "resnet20_cdf":      # configs[0]: the CDF-only tree (no correlation / ADMM term), batch-norm folded too
                    model = resnet20_quant(bits, bits, tree="cdf").to(dev).train()
                else:
                    model = (resnet20_quant if kind == "resnet20" else resnet56_quant)(bits, bits).to(dev).train()
                st = TrainStep(model, lr=0.04, channels_last=True)
                x = torch.randn(batch, 3, 32, 32, generator=gen).to(dev)
                y = torch.randint(0, 10, (batch,), generator=gen).to(dev)
                st.capture(x, y, warmup=3)
                x, y = st.static_inputs()
                run = lambda: st(x, y)                               # noqa: E731
                images = batch
            n = steps if kind != "resnet50_dann" else max(20, steps // 2)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                res = run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            ce = float(res[1].detach())
            out[name] = {"ms_per_step": ms, "images_per_sec": images / ms * 1e3, "steps": n, "per_gpu_batch": batch,
                         "algorithmic_mbytes_per_step": _CFG_BYTES[kind] / 1e6,
                         "in_scope_bytes_over_step_time_gbs": _CFG_BYTES[kind] / (ms * 1e-3) / 1e9,
                         "final_loss": ce, "finite": bool(ce == ce and abs(ce) != float("inf"))}
            del st, model, run
        except Exception as e:            # never fail the headline line on the extras
            out[name] = {"error": repr(e)[:300]}
        gc.collect()
        torch.cuda.empty_cache()
    config.args.bitW, config.args.abitW, config.args.train_batch_size, config.args.eval_batch_size = saved
    out["note"] = ("one GPU's share of BASELINE.json configs[2..4], HIP-graph replay, channels-last; config 5: every Conv2d_Q "
                   "convolution (stem included) on alignq_qconv_* (round 5), batch-norms folded into the quantiser / site kernels with their "
                   "statistics from the convolutions' epilogues, source and target batch in one traversal; lr 0.004 from random "
                   "init; CPU numbers of configs 4 and 5 on the same kind of box: profiles/r05_cpu_baseline_configs.json")
    return out


def timed_steps(step, x, y, steps, world, dist, dev, sync=None):
    """The timed region of the driver contract: EXACTLY `steps` steps bracketed by a barrier + device synchronisation on both
    sides; the elapsed time is the MAX over the ranks (an all-reduce, so every rank holds it).  `sync` / `dist` are parameters
    so that tests/test_bench_assembly.py can drive this bookkeeping without a GPU or a real process group."""
    sync = sync or torch.cuda.synchronize

    def fence():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    fence()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step(x, y)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def headline(a, elapsed, images_per_step, world, office, final_ce, final_tl):
    """The driver's JSON line (before the roofline / cpu_baseline / extras are attached): `value` is the WHOLE-JOB aggregate -
    every rank processed steps x images_per_step images in the max-over-ranks time - `scaling` weak (fixed per-GPU batch)."""
    images = a.steps * images_per_step * world
    return {
        "metric": (f"images/sec (train step, CDF-only tree) {a.model} {a.bits}-bit" if (a.tree == "cdf" and not office) else
                   "images/sec (train step, CDF+ADMM) ResNet-20 8-bit" if (a.model == "resnet20" and a.bits == 8) else
                   f"images/sec (train step, CDF+ADMM) {a.model} {a.bits}-bit"),
        "value": images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"{a.model} Office-31 shape 3x224x224 DANN, {a.bits}W/{a.bits}A CDF+ADMM full train step "
                                f"(cdf_alignment_admm/dann_office: source+target pass), batch {a.batch}+{a.batch}/GPU, "
                                f"random init at lr {a.lr if a.lr is not None else 0.004}, "
                                if office else
                                f"{a.model} CIFAR-shape 3x32x32, {a.bits}W/{a.bits}A CDF-only full train step "
                                f"(cdf_alignment/resnet-20-cifar-10), batch {a.batch}/GPU, "
                                if a.tree == "cdf" else
                                f"{a.model} CIFAR-shape 3x32x32, {a.bits}W/{a.bits}A CDF+ADMM full train step "
                                f"(cdf_alignment_admm/resnet-20-cifar-10), batch {a.batch}/GPU, ")
                               + f"{'HIP-graph replay' if not a.no_graph else 'eager launches'}"
                               + ("" if (office or a.no_fuse_bn) else ", batch-norm folded into the site kernels")
                               + ("" if (a.no_miopen_find or not (office or a.nchw or a.no_qconv)) else
                                  ", MIOpen find mode for the convolutions")
                               + ("" if a.nchw else ", channels-last tensors")
                               + ("" if (office or a.nchw or a.no_qconv) else
                                  ", all Conv2d_Q convolutions on alignq_conv*_nhwc (exact-product bf16 MFMA)")
                               + ("" if (not office or a.nchw or a.no_qconv) else
                                  ", every Conv2d_Q convolution (7x7 stem, 1x1, 3x3, stride 1 and 2) on alignq_qconv_* "
                                  "(exact-product bf16 / f16 MFMA GEMMs)"),
                   "global_batch": a.batch * world, "parallelism": f"dp{world}",
                   "final_ce": final_ce, "final_trans_loss": final_tl},
    }


def free_port():
    """A port nobody listens on right now (bound to port 0 and released): two `bench.py --gpus N` on one box, or a stale rank of a
    killed run, would collide on a fixed rendezvous port and the caller would get no JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(a, argv, port=None):
    """The command line `python bench.py --gpus N ...` (N > 1, not already a rank of a launch) starts as a CHILD process:
    one rank per GPU of this node under torch.distributed.run, same arguments.  127.0.0.1 rendezvous (the host name of a
    GPU box may not resolve)."""
    port = port or (int(os.environ["MASTER_PORT"]) if os.environ.get("MASTER_PORT") else free_port())
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv):
    """`--gpus N` without a launcher around it: this process has touched no GPU yet (importing torch does not), so it starts
    the N ranks as a child process, relays rank 0's stdout (the one JSON line) and stderr untouched, and returns the child's
    exit code.  Never os.exec* here: a process image replaced after HIP initialisation takes the machine down."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(launcher_command(a, argv), env=env)


def check_world(a, world):
    """Every rank: the number of ranks the launcher gave must be the --gpus the line will report."""
    if world != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
                         f"(or run `python bench.py --gpus {a.gpus}` alone, which starts the ranks itself)\n")
        sys.exit(2)


def dry_run(a, rank, world):
    """No GPU, no kernels: the launch chain (self_launch -> torch.distributed.run -> ranks), the fences and the max-over-ranks
    clock of timed_steps, and the line assembly, on a gloo group.  The line cannot be mistaken for a measurement."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo")

    def stub(_x, _y):
        time.sleep(0.001 * (1 + rank))           # the last rank is the slowest: the line must carry ITS time
        return None, torch.tensor(0.0), None
    elapsed, _ = timed_steps(stub, None, None, a.steps, world, dist, torch.device("cpu"), sync=lambda: None)
    if rank == 0:
        res = headline(a, elapsed, a.batch, world, a.model == "resnet50_dann", 0.0, None)
        res["metric"] = "dry run (launch rehearsal on CPU/gloo, no GPU work): " + res["metric"]
        res["data"] = "none (dry run)"
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    check_world(a, world)
    if a.dry_run:
        return dry_run(a, rank, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback in the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.backends.cudnn.benchmark = not a.no_miopen_find
    import torch.distributed as dist
    if world > 1 or a.dp_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with _StdoutToStderr():
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()                   # creates the communicator (and prints RCCL's banner) here, not on stdout
    from alignq_amd import _lib, config, dp
    from alignq_amd.resnet import resnet20_quant, resnet56_quant
    from alignq_amd.train_step import TrainStep
    if not os.path.exists(_lib.SO_PATH) and rank == 0:
        import __graft_entry__
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    _lib.load()

    config.args.bitW = config.args.abitW = a.bits
    config.args.train_batch_size = a.batch
    config.args.eval_batch_size = a.batch
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(rank)
    office = a.model == "resnet50_dann"
    if office:
        from alignq_amd.resnet_office import resnet50_dann
        from alignq_amd.train_step import OfficeTrainStep
        model = resnet50_dann(a.bits, a.bits).to(dev).train()
        ostep = OfficeTrainStep(model, lr=a.lr if a.lr is not None else 0.004, channels_last=not a.nchw,
                                fuse_bn=not a.no_fuse_bn, dual=(False if a.no_dual else None), qconv=not a.no_qconv)
        if world > 1 or a.dp_selftest:       # >= 4 gradient buckets all-reduced from autograd hooks during the backward
            office_hook = dp.attach_office(ostep, force=a.dp_selftest)
        xs = torch.randn(a.batch, 3, 224, 224, generator=gen).to(dev)
        xt = torch.randn(a.batch, 3, 224, 224, generator=gen).to(dev)
        ys = torch.randint(0, 31, (a.batch,), generator=gen).to(dev)
        if a.no_graph:
            for _ in range(2):
                ostep(xs, ys, xt)
        else:
            ostep.capture(xs, ys, xt, warmup=2)

        def step(_x, _y):
            return ostep(xs, ys, xt)
        x = y = None
        images_per_step = 2 * a.batch            # source + target images both pass through the network
    else:
        model = (resnet20_quant if a.model == "resnet20" else resnet56_quant)(a.bits, a.bits, tree=a.tree).to(dev).train()
        step = TrainStep(model, lr=a.lr if a.lr is not None else 0.04, fuse_bn=not a.no_fuse_bn, channels_last=not a.nchw,
                         qconv=not a.no_qconv, pack_bins=not a.no_pack_bins)
        if world > 1 or a.dp_selftest:
            dp.attach(step, force=a.dp_selftest)
        x = torch.randn(a.batch, 3, 32, 32, generator=gen).to(dev)
        y = torch.randint(0, 10, (a.batch,), generator=gen).to(dev)
        if a.no_graph:
            for _ in range(3):
                step(x, y)
        else:
            step.capture(x, y, warmup=3)
            # inputs resident in HBM (BASELINE metric): the synthetic batch sits in the buffers the captured step reads, as a
            # loader's host-to-device copy would leave it (no per-step device-to-device staging copy in the timed region)
            x, y = step.static_inputs()
        images_per_step = a.batch
    for _ in range(a.warmup):
        step(x, y)

    elapsed, out = timed_steps(step, x, y, a.steps, world, dist, dev)
    logits, ce, tl = out
    assert torch.isfinite(ce).item(), "training step produced a non-finite loss"

    if rank == 0:
        res = headline(a, elapsed, images_per_step, world, office, float(ce.detach()),
                       float(tl.detach()) if tl is not None else None)
        if not a.no_kernels and not office:
            counts = {}
            units = [3, 3, 3] if a.model == "resnet20" else [9, 9, 9]
            # site F per stage: 16x32x32, 32x16x16, 64x8x8 ; stem + 2/block + 1 skip in the first block of stages 2,3
            counts[16384] = 1 + 2 * units[0]
            counts[8192] = 2 * units[1] + 1
            counts[4096] = 2 * units[2] + 1
            roof, kernels = measure_kernels(dev, a.batch, a.bits, counts, {16384: 1024, 8192: 256, 4096: 64},
                                            folded=not a.no_fuse_bn, nhwc=not a.nchw, shapes=not a.no_shapes)
            res["roofline"] = roof
            res["kernels"] = kernels
        if world == 1 and not a.no_cpu_baseline and not office:
            res["cpu_baseline"] = cpu_baseline(a.batch, a.bits, a.model, a.cpu_steps)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
            if a.model == "resnet20" and not a.no_other_configs:
                # BASELINE.json configs[0] (the reference's own CPU-runnable case: CDF-only tree, cdf_alignment/resnet-20-cifar-10/
                # main.py:269-315) on the same host cores: the three thread counts around the ADMM workload's best
                best = res["cpu_baseline"]["cores"]
                res["cpu_baseline_cdf_only"] = cpu_baseline(a.batch, a.bits, a.model, a.cpu_steps, tree="cdf",
                                                            thread_counts=(max(8, best // 2), best, best * 2))

        if world == 1 and not office and not a.dp_selftest and not a.no_dp_probe:
            # the data-parallel path at world size 1 (RCCL all-reduce of the flat gradient + D bucket between two HIP graphs):
            # what one rank of the N > 1 runs executes per step, minus the wire time (no 8-GPU node in this session)
            res["dp_selftest"] = dp_probe(dev, a)
        if world == 1 and not office and not a.no_other_configs and a.model == "resnet20" and a.bits == 8:
            res["other_configs"] = other_configs(dev, a)
            off_dp = res.get("dp_selftest", {}).get("office", {})
            if "ms_per_step" in off_dp and "resnet50_dann_8w8a_b28" in res["other_configs"]:
                res["other_configs"]["resnet50_dann_8w8a_b28"]["dp_selftest_ms"] = off_dp["ms_per_step"]
            gpu1 = res["other_configs"].get("resnet20_cdf_only_8w8a_b128", {}).get("images_per_sec")
            if gpu1 and "cpu_baseline_cdf_only" in res:
                res["cpu_baseline_cdf_only"]["gpu_images_per_sec"] = gpu1
                res["cpu_baseline_cdf_only"]["gpu_speedup"] = gpu1 / res["cpu_baseline_cdf_only"]["value"]
        print(json.dumps(res))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
