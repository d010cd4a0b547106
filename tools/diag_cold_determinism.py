"""Determinism audit (round 6): every site launch form of the CIFAR / Office shapes, launched right behind a large GEMM (cold L2, the
chip's state left by foreign code), REPS times, against the same launch on an idle chip - bit for bit.  Found with it: the backward
twin at F = 8192 (two site workgroups per CU) differed by one ulp in scattered elements until site4_kernels.hip was built without
SLP vectorisation (NOTES.md, round 6)."""
import sys, os, ctypes, importlib.util, numpy as np, torch
sys.path.insert(0, '.')
spec = importlib.util.spec_from_file_location("r6", "tests/test_gpu_round6.py"); r6 = importlib.util.module_from_spec(spec); spec.loader.exec_module(r6)
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')
REPS = int(os.environ.get("REPS", 20))
m1 = torch.randn(4096, 4096, device=dev)
k = 8

def audit(B, C, H):
    HW, F = H * H, C * H * H
    g = torch.Generator().manual_seed(B + C + H)
    mk = lambda sc=1.5: (torch.randn(B, H, H, C, generator=g) * sc + 0.2).to(dev).permute(0, 3, 1, 2)
    z = [mk(), mk()]; gy = [mk(1e-2), mk(1e-2)]
    gam = [(torch.rand(C, generator=g) + 0.5).to(dev) for _ in range(2)]; bet = [(torch.randn(C, generator=g) * 0.2).to(dev) for _ in range(2)]
    A, Gm = (torch.randn(128, 128, generator=g) * 0.05).to(dev), (torch.randn(128, 128, generator=g) * 0.05).to(dev)

    def fwd(twin, cold):
        a, ta = r6._site_bn_args(L, lib, dev, z[0], gam[0], bet[0], k, True, True, C, HW, B, F)
        b, tb = r6._site_bn_args(L, lib, dev, z[1], gam[1], bet[1], k, False, False, C, HW, B, F)
        torch.cuda.synchronize()
        if cold: torch.mm(m1, m1)
        rc = lib.alignq_site_partials_bn_twin(ctypes.byref(a), ctypes.byref(b), None) if twin else L.EUNSUPPORTED
        if rc == L.EUNSUPPORTED:
            for s_ in (a, b):
                L.check(lib.alignq_site_partials_bn(s_.z, s_.bn_part, s_.bn_gamma, s_.bn_beta, s_.running_mean, s_.running_var, s_.num_batches_tracked,
                                                    s_.momentum, s_.bn_eps, s_.ab, s_.save, C, HW, B, F, k, 2.0, 0.0, s_.relu, None, 1, 0, s_.xq,
                                                    s_.bins_out, s_.stats, s_.ws, None), "single")
        else:
            L.check(rc, "fwd twin")
        out = []
        for t in (ta, tb):
            D, scal = torch.empty(B, B, device=dev), torch.empty(4, device=dev)
            L.check(lib.alignq_site_reduce_loss(L.ptr(t["ws"]), B, F, L.ptr(D), L.ptr(A), L.ptr(Gm), 128, 0.2, 0.3, L.ptr(scal), None), "reduce")
            t["D"], t["scal"] = D, scal
            out += [D, scal[:1], t["stats"], t["ab"], t["save"]] + ([t["y"]] if t["y"] is not None else [t["bins"]])
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in out], (ta, tb), rc != L.EUNSUPPORTED

    ref, (ta, tb), twin_ok = fwd(False, False)
    S = []
    for t in (ta, tb):
        s_ = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev); one = torch.ones((), device=dev)
        dA, dG = torch.empty_like(A), torch.empty_like(Gm)
        L.check(lib.alignq_site_prep_fused(L.ptr(t["D"]), L.ptr(A), L.ptr(Gm), 128, L.ptr(t["scal"]), 0.2, L.ptr(one), B, F, L.ptr(s_), L.ptr(dA),
                                           L.ptr(dG), None), "prep")
        S.append(s_)

    def bwd(twin, cold):
        outs, structs = [], []
        for i, t in enumerate((ta, tb)):
            dx = torch.full_like(z[i], float("nan")); part = torch.zeros(lib.alignq_site_bn_part_bytes(F, 1), dtype=torch.uint8, device=dev)
            bins = t["bins"]
            structs.append(L.SiteBwdBnArgs(L.ptr(gy[i]), L.ptr(S[i]), L.ptr(z[i]), L.ptr(t["ab"]), L.ptr(t["save"]), C, HW, 1, None, L.ptr(bins),
                                           2 if bins is not None else 0, None, L.ptr(t["stats"]), B, F, 2.0, 0.0, L.ptr(dx), L.ptr(part)))
            outs += [dx, part]
        torch.cuda.synchronize()
        if cold: torch.mm(m1, m1)
        rc = lib.alignq_site_bwd_apply_bn_twin(ctypes.byref(structs[0]), ctypes.byref(structs[1]), None) if twin else L.EUNSUPPORTED
        if rc == L.EUNSUPPORTED:
            for q in structs:
                L.check(lib.alignq_site_bwd_apply_bn(q.g, q.S, q.z, q.ab, q.save, q.C, q.HW, q.nhwc, q.y_relu, q.y_bins, q.y_bin_bytes, q.dresidual,
                                                     q.stats, q.B, q.F, q.act_range, q.eps, q.dx, q.dx_part, None), "single")
        else:
            L.check(rc, "bwd twin")
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in outs], rc != L.EUNSUPPORTED

    # the single launches with their filler role (F <= 8192: two pending filter-gradient slab reductions ride along, as in the step)
    slabs = [torch.randn(64, 36864, generator=g).to(dev) * 1e-3 for _ in range(2)]

    def bwd_fill(cold):
        outs = []
        calls = []
        for i, t in enumerate((ta, tb)):
            dx = torch.full_like(z[i], float("nan")); part = torch.zeros(lib.alignq_site_bn_part_bytes(F, 1), dtype=torch.uint8, device=dev)
            dws = [torch.full((36864,), float("nan"), device=dev) for _ in range(2)]
            bins = t["bins"]
            calls.append((L.ptr(gy[i]), L.ptr(S[i]), L.ptr(z[i]), L.ptr(t["ab"]), L.ptr(t["save"]), C, HW, 1, None, L.ptr(bins),
                          2 if bins is not None else 0, None, L.ptr(t["stats"]), B, F, 2.0, 0.0, L.ptr(dx), L.ptr(part), 2,
                          L.ptr_array(slabs), L.ptr_array(dws), (ctypes.c_int * 2)(64, 64), (ctypes.c_int * 2)(36864, 36864), None))
            outs += [dx, part] + dws
        torch.cuda.synchronize()
        for c in calls:
            if cold: torch.mm(m1, m1)
            L.check(lib.alignq_site_bwd_apply_bn_fill(*c), "bwd fill")
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in outs]

    bref, _ = bwd(False, False)
    if lib.alignq_site_bwd_fill_slots(B, F) >= 2:
        fref = bwd_fill(False)
        bad = sum(any(x.tobytes() != y.tobytes() for x, y in zip(bwd_fill(True), fref)) for _ in range(REPS))
        print(f"B={B} C={C} H={H} F={F}: bwd single + 2 fillers: {bad} of {REPS} repetitions differ", flush=True)
    for name, fn, rf in (("fwd single", lambda: fwd(False, True)[0], ref), ("fwd twin", lambda: fwd(True, True)[0], ref),
                         ("bwd single", lambda: bwd(False, True)[0], bref), ("bwd twin", lambda: bwd(True, True)[0], bref)):
        bad = 0
        for _ in range(REPS):
            v = fn()
            bad += any(not (x.tobytes() == y.tobytes()) for x, y in zip(v, rf))
        print(f"B={B} C={C} H={H} F={F}: {name}: {bad} of {REPS} repetitions differ", flush=True)

for shape in ((128, 16, 32), (128, 32, 16), (128, 64, 8), (100, 32, 16))[slice(1, 2) if os.environ.get("ONLY_F8192") else slice(None)]:
    audit(*shape)
