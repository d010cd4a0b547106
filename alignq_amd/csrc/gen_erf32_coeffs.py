#!/usr/bin/env python3
"""Derive the coefficient tables of the ALIGNQ-ERF32 / ALIGNQ-EXP32 specification.

The reference evaluates erf/exp through torch-CPU, i.e. Intel MKL VML (closed source, CPU-dispatch
dependent): its last-ulp behaviour cannot be restated.  This repo therefore fixes its OWN fp32
erf/exp as a sequence of IEEE-754 single operations (fma, mul, add, rint, exact power-of-two
scaling) that gives identical bits on x86 (oracle/alignq_oracle.c) and on gfx950
(alignq_amd/csrc/alignq_math.h).  This script fits the polynomials (weighted least squares on
Chebyshev nodes, float64) and writes the tables as C hex-floats to include/alignq_erf32_coeffs.h.
Both implementations include that one data header, so they cannot drift apart.

    nerf32(y) ~ erf(y/sqrt(2)) = 2*Phi(y)-1   (round 3: ONE evaluated branch per element, no exp, ~9 vector ops)
       a = min(|y|, 5.625)  (NaN stays NaN);  u = a + 2^20  (RN: u - 2^20 = a rounded to the nearest 1/8, ties to even)
       k = bits(u) - bits(2^20)  in 0..45;    d = a - P[k]           (P[k] ~ k/8 is an fp32 at which 2*Phi-1 is within
       res = fma(fma(fma(fma(C4[k],d,C3[k]),d,C2[k]),d,C1[k]),d,C0[k])   1e-3 ulp of the fp32 C0[k]: no constant-term error)
       sign restored by copysign.  Entry 45 is the constant 1 (1-erf(5.5625/sqrt 2) < 2^-25).
       Measured over EVERY fp32 in [0, 6] (tests/native/verify_nerf.c): |error| <= 0.57 * 2^-24 absolute.
    erf32(x)  (round 1-2, two regions + exp; kept as data for exp32's users and for the record):  a=|x|
       a <  0.875 : a + a*PA(a*a)                 PA degree 6  (erf(a)/a - 1)
       a <  4.0   : 1 - exp32(-PB(a))             PB degree 7  (-log erfc(a))
       else       : 1                              (1-erf(4) < 2^-25)
    exp32(x):  n=rint(x*log2e); r = fma(n,-LN2_HI,x); r = fma(n,-LN2_LO,r)
               e^r = 1 + (r + r*r*PE(r))           PE degree 5  ((e^r-1-r)/r^2)
               result scaled by 2^(n>>1) then 2^(n-(n>>1)); x<-104 -> 0; x>88.7 -> +inf
"""
import os

import numpy as np
import scipy.special as sp

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "..", "include", "alignq_erf32_coeffs.h")

ERF_T = 0.875
ERF_HI = 4.0


def cheb_nodes(a, b, n):
    k = np.arange(n)
    return 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * (k + 0.5) / n)


def fit_pa(deg=6):
    s = cheb_nodes(0.0, ERF_T * ERF_T, 800)
    a = np.sqrt(s)
    f = sp.erf(a) / a - 1.0
    return np.polyfit(s, f, deg)[::-1]          # lowest order first


def fit_pb(deg=7):
    a = cheb_nodes(ERF_T, ERF_HI, 1200)
    f = -np.log(sp.erfc(a))
    w = sp.erfc(a)                               # absolute error of the result is erfc * dp
    V = np.vander(a, deg + 1, increasing=True)
    c, *_ = np.linalg.lstsq(V * w[:, None], f * w, rcond=None)
    return c


def fit_pe(deg=5):
    r = cheb_nodes(-0.5 * np.log(2) * 1.01, 0.5 * np.log(2) * 1.01, 800)
    f = (np.expm1(r) - r) / (r * r)
    return np.polyfit(r, f, deg)[::-1]


NERF_STEP = 0.125
NERF_N = 46
NERF_YMAX = (NERF_N - 1) * NERF_STEP


def nerf_true(y):
    return sp.erf(np.asarray(y, dtype=np.float64) / np.sqrt(2.0))


def fit_nerf():
    """Per node k: an fp32 centre P[k] near k/8 whose function value is (almost) an fp32 number C0[k], and the
    near-minimax degree-4 remainder d*(C1 + C2 d + C3 d^2 + C4 d^3) on the node's interval (Chebyshev-node least squares)."""
    rows = []
    h = NERF_STEP / 2
    for k in range(NERF_N):
        ck = k * NERF_STEP
        lo_y, hi_y = max(0.0, ck - h), min(NERF_YMAX, ck + h)
        if np.float32(nerf_true(lo_y)) == np.float32(1.0):          # the whole interval rounds to 1
            rows.append((1.0, ck, 0.0, 0.0, 0.0, 0.0))
            continue
        if k == 0:
            pk, c0 = np.float32(0.0), np.float32(0.0)
        else:
            slope = np.sqrt(2 / np.pi) * np.exp(-ck * ck / 2)
            ulp = float(np.spacing(np.float32(ck)))
            win = min(1 / 32, max(2.0 ** -11, 4 * 2.0 ** -24 / slope))
            n_c = int(min(win / ulp, 200000))
            stride = max(1, int(win / ulp / n_c))
            cands = (np.float32(ck) + np.arange(-n_c, n_c + 1, dtype=np.float64) * ulp * stride).astype(np.float32)
            v = nerf_true(cands.astype(np.float64))
            err = np.abs(v - np.float32(v).astype(np.float64)) / np.spacing(np.float32(v)).astype(np.float64)
            j = int(np.argmin(err + 1e-4 * np.abs(cands.astype(np.float64) - ck) / win))
            pk, c0 = cands[j], np.float32(v[j])
        d = cheb_nodes(lo_y - float(pk), hi_y - float(pk), 400)
        tgt = nerf_true(float(pk) + d) - float(c0)
        V = np.stack([d, d ** 2, d ** 3, d ** 4], 1)
        c, *_ = np.linalg.lstsq(V, tgt, rcond=None)
        rows.append((float(c0), float(pk), *[float(x) for x in c]))
    return rows


def hexf(v):
    return float(np.float32(v)).hex() + "f"


def main():
    pa, pb, pe = fit_pa(), fit_pb(), fit_pe()
    ln2 = np.log(2.0)
    ln2_hi = float(np.float32(np.floor(ln2 * 2 ** 15) / 2 ** 15))   # 16 significant bits: n*hi exact for |n|<256
    ln2_lo = ln2 - ln2_hi
    lines = [
        "/* GENERATED by alignq_amd/csrc/gen_erf32_coeffs.py — do not edit.",
        " * Pure data: coefficient tables of the ALIGNQ-ERF32 / ALIGNQ-EXP32 specification",
        " * (see that script's docstring and DESIGN.md §3).  Lowest order first. */",
        "#ifndef ALIGNQ_ERF32_COEFFS_H",
        "#define ALIGNQ_ERF32_COEFFS_H",
        f"#define ALIGNQ_ERF_T   {hexf(ERF_T)}",
        f"#define ALIGNQ_ERF_HI  {hexf(ERF_HI)}",
        f"#define ALIGNQ_LOG2E   {hexf(1.0 / ln2)}",
        f"#define ALIGNQ_LN2_HI  {hexf(ln2_hi)}",
        f"#define ALIGNQ_LN2_LO  {hexf(ln2_lo)}",
    ]
    for name, c in (("PA", pa), ("PB", pb), ("PE", pe)):
        for i, v in enumerate(c):
            lines.append(f"#define ALIGNQ_{name}{i}  {hexf(v)}   /* {v:+.12e} */")
    rows = fit_nerf()
    lines += [
        "/* ALIGNQ-NERF32: nerf32(y) ~ erf(y/sqrt(2)), table-driven, one degree-4 polynomial per 1/8-wide node. */",
        f"#define ALIGNQ_NERF_N     {NERF_N}",
        f"#define ALIGNQ_NERF_YMAX  {hexf(NERF_YMAX)}",
        "#define ALIGNQ_NERF_MAGIC 0x1.0000000000000p+20f   /* ulp(2^20) = 1/8 */",
        "/* [N][4] = C1..C4 (the 16-byte record a lane fetches first) */",
        "#define ALIGNQ_NERF_POLY { \\",
    ]
    for r in rows:
        lines.append("  { " + ", ".join(hexf(v) for v in r[2:6]) + " }, \\")
    lines += ["}", "/* [N][4] = C0 (function value at the centre), P (the centre), 0, 0: the same 16-byte stride as the first record */",
              "#define ALIGNQ_NERF_CENTRE { \\"]
    for r in rows:
        lines.append("  { " + hexf(r[0]) + ", " + hexf(r[1]) + ", 0x0.0p+0f, 0x0.0p+0f }, \\")
    lines += ["}"]
    lines.append("#endif")
    with open(OUT, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
