#!/usr/bin/env python3
"""Per-layer timing of Conv2d_Q's convolutions at the ResNet-50 / Office-31 shapes (B = 56 = the merged source + target batch of
BASELINE config 5): alignq_qconv_fwd / _dgrad / _wgrad against torch's (MIOpen's) fp32 convolution on the same tensors.
HIP events on the launch stream, 4 rotating operand sets.  Runs on the GPU box:  python tools/qconv_bench.py [--B 56]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignq_amd import _lib as L  # noqa: E402

CL = torch.channels_last
# (name, C_in, C_out, H_in, KS, stride, count in the network, input is a level tensor)
LAYERS = [
    ("l1.0.conv1", 64, 64, 56, 1, 1, 1, True), ("l1.0.ds", 64, 256, 56, 1, 1, 1, True), ("l1.conv1", 256, 64, 56, 1, 1, 2, False),
    ("l1.conv2", 64, 64, 56, 3, 1, 3, True), ("l1.conv3", 64, 256, 56, 1, 1, 3, True),
    ("l2.0.conv1", 256, 128, 56, 1, 1, 1, False), ("l2.0.conv2", 128, 128, 56, 3, 2, 1, True), ("l2.0.ds", 256, 512, 56, 1, 2, 1, False),
    ("l2.conv1", 512, 128, 28, 1, 1, 3, False), ("l2.conv2", 128, 128, 28, 3, 1, 3, True), ("l2.conv3", 128, 512, 28, 1, 1, 4, True),
    ("l3.0.conv1", 512, 256, 28, 1, 1, 1, False), ("l3.0.conv2", 256, 256, 28, 3, 2, 1, True), ("l3.0.ds", 512, 1024, 28, 1, 2, 1, False),
    ("l3.conv1", 1024, 256, 14, 1, 1, 5, False), ("l3.conv2", 256, 256, 14, 3, 1, 5, True), ("l3.conv3", 256, 1024, 14, 1, 1, 6, True),
    ("l4.0.conv1", 1024, 512, 14, 1, 1, 1, False), ("l4.0.conv2", 512, 512, 14, 3, 2, 1, True), ("l4.0.ds", 1024, 2048, 14, 1, 2, 1, False),
    ("l4.conv1", 2048, 512, 7, 1, 1, 2, False), ("l4.conv2", 512, 512, 7, 3, 1, 2, True), ("l4.conv3", 512, 2048, 7, 1, 1, 3, True),
]


def time_rot(fn, sets, reps=12, warm=2):
    for i in range(warm):
        fn(i % sets)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        fn(i % sets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=56)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--i16", action="store_true", help="level inputs as int16 indices (N2) instead of fp32 values")
    ap.add_argument("--no-ksplit", action="store_true", help="data gradient without the split-K scratch (one workgroup per tile)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    lib = L.load()
    st = L.stream_ptr()
    p = L.ptr
    R = 3
    tot = {k: 0.0 for k in ("fwd", "dgrad", "wgrad", "ref_fwd", "ref_dgrad", "ref_wgrad")}
    rows = []
    for name, cin, cout, H, ks, s, cnt, lev in LAYERS:
        if a.only and a.only not in name:
            continue
        B = a.B
        Ho = (H - 1) // s + 1
        pad = (ks - 1) // 2
        xs = []
        for i in range(R):
            if lev:
                x = (torch.clamp(torch.round(torch.randn(B, cin, H, H, device=dev) * 150), 0, 510) / 255.0)
            else:
                x = torch.relu(torch.randn(B, cin, H, H, device=dev) * 1.3)
            xs.append(x.contiguous(memory_format=CL))
        w = (torch.round(torch.tanh(torch.randn(cout, cin, ks, ks, device=dev)) * 255) / 255).contiguous(memory_format=CL)
        gys = [(torch.randn(B, cout, Ho, Ho, device=dev) * 1e-3).contiguous(memory_format=CL) for _ in range(R)]
        ys = [torch.empty(B, cout, Ho, Ho, device=dev).contiguous(memory_format=CL) for _ in range(R)]
        dxs = [torch.empty(B, cin, H, H, device=dev).contiguous(memory_format=CL) for _ in range(R)]
        dw = torch.empty_like(w)
        ws = torch.empty(max(16, lib.alignq_qconv_wgrad_ws_bytes(B, H, H, cin, cout, ks, s)), dtype=torch.uint8, device=dev)
        xl = 255.0 if lev else 0.0
        wbf, whf = torch.empty_like(w, dtype=torch.int16), torch.empty_like(w, dtype=torch.int16)
        L.check(lib.alignq_qconv_pack_weights(1, L.ptr_array([w]), L.i64_array([w.numel()]), 8, L.ptr_array([wbf]), L.ptr_array([whf]), st), "pack")
        wfw = whf if lev else wbf
        xb = 2 if (lev and a.i16) else 0
        xin = [torch.round(x * 255.0).to(torch.int16).contiguous(memory_format=CL) for x in xs] if xb else xs
        f = lambda i: L.check(lib.alignq_qconv_fwd(p(xin[i]), p(wfw), p(ys[i]), B, H, H, cin, cout, ks, s, 8, xl, xb, 1, None, st), "fwd")
        t_f = time_rot(f, R)
        nwd = 0 if a.no_ksplit else lib.alignq_qconv_dgrad_ws_bytes(B, H, H, cin, cout, ks, s)
        wsd = torch.empty(nwd, dtype=torch.uint8, device=dev) if nwd else None
        d = lambda i: L.check(lib.alignq_qconv_dgrad(p(gys[i]), p(wbf), p(dxs[i]), B, H, H, cin, cout, ks, s, 8, p(wsd), st), "dgrad")
        t_d = time_rot(d, R)
        import ctypes
        ns = ctypes.c_int(0)
        g = lambda i: L.check(lib.alignq_qconv_wgrad(p(xin[i]), p(gys[i]), None, p(ws), B, H, H, cin, cout, ks, s, xl, xb, ctypes.byref(ns), st), "wgrad")
        t_w = time_rot(g, R)
        g2 = lambda i: L.check(lib.alignq_qconv_wgrad(p(xin[i]), p(gys[i]), p(dw), p(ws), B, H, H, cin, cout, ks, s, xl, xb, None, st), "wgrad")
        t_w2 = time_rot(g2, R)
        r_f = r_d = r_w = float("nan")
        if not a.no_ref:
            r_f = time_rot(lambda i: torch.nn.functional.conv2d(xs[i], w, stride=s, padding=pad), R)
            r_d = time_rot(lambda i: torch.ops.aten.convolution_backward(gys[i], xs[i], w, None, (s, s), (pad, pad), (1, 1), False, (0, 0), 1,
                                                                         (True, False, False)), R)
            r_w = time_rot(lambda i: torch.ops.aten.convolution_backward(gys[i], xs[i], w, None, (s, s), (pad, pad), (1, 1), False, (0, 0), 1,
                                                                         (False, True, False)), R)
        gf = 2.0 * B * Ho * Ho * cin * cout * ks * ks / 1e9
        mb = 4.0 * B * (H * H * cin + Ho * Ho * cout) / 1e6
        rows.append({"layer": name, "count": cnt, "gflop": gf, "mbytes": mb, "fwd_us": t_f, "dgrad_us": t_d, "wgrad_us": t_w,
                     "ref_fwd_us": r_f, "ref_dgrad_us": r_d, "ref_wgrad_us": r_w})
        print(f"{name:12s} x{cnt} {gf:6.1f} GF {mb:6.1f} MB | fwd {t_f:7.1f} ({r_f:7.1f})  dgrad {t_d:7.1f} ({r_d:7.1f})  "
              f"wgrad {t_w:7.1f} +red {t_w2 - t_w:5.1f} [{ns.value:3d}] ({r_w:7.1f}) us | fwd {gf / t_f * 1e3:6.1f} GF/s/1e3  {mb / t_f:5.2f} TB/s", flush=True)
        for k, v in (("fwd", t_f), ("dgrad", t_d if t_d == t_d else r_d), ("wgrad", t_w), ("ref_fwd", r_f), ("ref_dgrad", r_d), ("ref_wgrad", r_w)):
            tot[k] += cnt * v
        del xs, gys, ys, dxs
    print("per step (us, weighted by layer count):", json.dumps({k: round(v, 1) for k, v in tot.items()}))
    print("ours", round(tot["fwd"] + tot["dgrad"] + tot["wgrad"], 1), "ref", round(tot["ref_fwd"] + tot["ref_dgrad"] + tot["ref_wgrad"], 1))


if __name__ == "__main__":
    main()
