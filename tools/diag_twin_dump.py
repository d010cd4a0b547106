"""Diagnostic (round 6): where the co-resident backward twin launch first differs.  Needs the diagnostic build
(-DALIGNQ_DIAG_CORESIDENT -DALIGNQ_DIAG_DUMP of site4_kernels.hip, ALIGNQ_SO): the kernel dumps t / jac (stage 0), the MFMA accumulators
(1), the per-column projection sums (2) and the assembled tile (3); a one-workgroup-per-CU launch is the reference."""
import sys, os, ctypes, importlib.util, numpy as np, torch
sys.path.insert(0, '.')
spec = importlib.util.spec_from_file_location("r6", "tests/test_gpu_round6.py"); r6 = importlib.util.module_from_spec(spec); spec.loader.exec_module(r6)
from alignq_amd import _lib as L
lib = L.load(); dev = torch.device('cuda:0')
lib.alignq_debug_set_dump.argtypes = [ctypes.c_void_p]; lib.alignq_debug_set_dump.restype = ctypes.c_int
lib.alignq_debug_one_per_cu.argtypes = [ctypes.c_int]; lib.alignq_debug_one_per_cu.restype = None
B, C, H = 128, 32, 16
k, HW, F = 8, H * H, C * H * H
g = torch.Generator().manual_seed(B + C + H)
mk = lambda sc=1.5: (torch.randn(B, H, H, C, generator=g) * sc + 0.2).to(dev).permute(0, 3, 1, 2)
z = [mk(), mk()]; gy = [mk(1e-2), mk(1e-2)]
gam = [(torch.rand(C, generator=g) + 0.5).to(dev) for _ in range(2)]; bet = [(torch.randn(C, generator=g) * 0.2).to(dev) for _ in range(2)]
A, Gm = (torch.randn(128, 128, generator=g) * 0.05).to(dev), (torch.randn(128, 128, generator=g) * 0.05).to(dev)
a, ta = r6._site_bn_args(L, lib, dev, z[0], gam[0], bet[0], k, True, True, C, HW, B, F)
b, tb = r6._site_bn_args(L, lib, dev, z[1], gam[1], bet[1], k, False, False, C, HW, B, F)
L.check(lib.alignq_site_partials_bn_twin(ctypes.byref(a), ctypes.byref(b), None), "fwd twin")
S = []
for t in (ta, tb):
    D, scal = torch.empty(B, B, device=dev), torch.empty(4, device=dev)
    L.check(lib.alignq_site_reduce_loss(L.ptr(t["ws"]), B, F, L.ptr(D), L.ptr(A), L.ptr(Gm), 128, 0.2, 0.3, L.ptr(scal), None), "reduce")
    s_ = torch.empty(lib.alignq_site_bwd_ws_bytes(B) // 4, device=dev); one = torch.ones((), device=dev)
    dA, dG = torch.empty_like(A), torch.empty_like(Gm)
    L.check(lib.alignq_site_prep_fused(L.ptr(D), L.ptr(A), L.ptr(Gm), 128, L.ptr(scal), 0.2, L.ptr(one), B, F, L.ptr(s_), L.ptr(dA), L.ptr(dG), None), "prep")
    S.append(s_)
m1 = torch.randn(4096, 4096, device=dev)
dump = torch.zeros(4, 2, 256, 128, 32, 2, device=dev)
assert lib.alignq_debug_set_dump(dump.data_ptr()) == 0


def launch(one_per_cu):
    lib.alignq_debug_one_per_cu(int(one_per_cu))
    outs, structs = [], []
    for i, t in enumerate((ta, tb)):
        dx = torch.full_like(z[i], float("nan")); part = torch.zeros(lib.alignq_site_bn_part_bytes(F, 1), dtype=torch.uint8, device=dev)
        bins = t["bins"]
        structs.append(L.SiteBwdBnArgs(L.ptr(gy[i]), L.ptr(S[i]), L.ptr(z[i]), L.ptr(t["ab"]), L.ptr(t["save"]), C, HW, 1, None, L.ptr(bins),
                                       2 if bins is not None else 0, None, L.ptr(t["stats"]), B, F, 2.0, 0.0, L.ptr(dx), L.ptr(part)))
        outs += [dx, part]
    dump.zero_()
    torch.cuda.synchronize()
    torch.mm(m1, m1)
    L.check(lib.alignq_site_bwd_apply_bn_twin(ctypes.byref(structs[0]), ctypes.byref(structs[1]), None), "bwd twin")
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs], dump.cpu().numpy().copy()


ref, dref = launch(True)
ref2, dref2 = launch(True)
print("one-per-CU launches agree:", all(x.tobytes() == y.tobytes() for x, y in zip(ref, ref2)), "dumps agree:", dref.tobytes() == dref2.tobytes(), flush=True)
names = ["t / jac (phase A) | with mask bit 16: the standardised x / t operands as the projection reads them from LDS", "accX / accT (after the MFMAs)", "projection sums", "assembled tile (Os) | with mask bit 64: the staged t operands (hi, lo) read back by the staging thread right behind its stores"]
found = 0
for rep in range(60):
    got, dgot = launch(False)
    if all(x.tobytes() == y.tobytes() for x, y in zip(got, ref)):
        continue
    found += 1
    print(f"--- launch {rep}: outputs differ", flush=True)
    for st in range(4):
        diff = np.argwhere(dgot[st].view(np.uint32) != dref[st].view(np.uint32))
        print(f"  stage {st} {names[st]}: {len(diff)} words differ", flush=True)
        if len(diff):
            tiles = sorted(set(map(tuple, diff[:, :2].tolist())))
            print("     (site, tile):", tiles[:10], "| rows", sorted(set(diff[:, 2].tolist()))[:24], "| cols", sorted(set(diff[:, 3].tolist())),
                  "| which", sorted(set(diff[:, 4].tolist())))
            for d in diff[:int(os.environ.get('SHOW', 6))]:
                i = tuple(d.tolist())
                print("       ", i, "got", float(dgot[st][i]), "ref", float(dref[st][i]), flush=True)
    if found == 3:
        break
print("failing launches examined:", found)
