"""ctypes binding of the plain-C oracle (oracle/alignq_oracle.c) for tests / smoke / cpu_baseline only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")

FORMULA_ADMM, FORMULA_CDF = 0, 1


def _build():
    srcs = [os.path.join(ROOT, "oracle", "alignq_oracle.c"), os.path.join(ROOT, "include", "alignq_erf32_coeffs.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _build()
        _lib = ctypes.CDLL(SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


c_long, c_int, c_float = ctypes.c_long, ctypes.c_int, ctypes.c_float


def nerf32(x):
    """ALIGNQ-NERF32: erf(x / sqrt(2)) = 2*Phi(x) - 1 as the repo specifies it (one table node + degree-4 polynomial)."""
    x = _f32(x); y = np.empty_like(x)
    lib().oq_nerf32(_p(x), _p(y), c_long(x.size)); return y


def exp32(x):
    x = _f32(x); y = np.empty_like(x)
    lib().oq_exp32(_p(x), _p(y), c_long(x.size)); return y


def act_quant_fwd(x, k, r, formula):
    x = _f32(x)
    xq, t, bins = np.empty_like(x), np.empty_like(x), np.empty(x.shape, np.int32)
    lib().oq_act_quant_fwd(_p(x), _p(xq), _p(t), _p(bins), c_long(x.size), c_int(k), c_float(r), c_int(formula))
    return xq, t, bins


def act_quant_bwd(g, x, r):
    g, x = _f32(g), _f32(x); dx = np.empty_like(x)
    lib().oq_act_quant_bwd(_p(g), _p(x), _p(dx), c_long(x.size), c_float(r)); return dx


def weight_stats(w):
    w = _f32(w); ms = np.empty(2, np.float32)
    lib().oq_weight_stats(_p(w), c_long(w.size), _p(ms)); return ms


def weight_quant_fwd(w, ms, k, formula):
    w, ms = _f32(w), _f32(ms)
    q, c, pdf, bins = np.empty_like(w), np.empty_like(w), np.empty_like(w), np.empty(w.shape, np.int32)
    lib().oq_weight_quant_fwd(_p(w), _p(ms), _p(q), _p(c), _p(pdf), _p(bins), c_long(w.size), c_int(k), c_int(formula))
    return q, c, pdf, bins


def weight_quant_bwd(g, w, ms):
    g, w, ms = _f32(g), _f32(w), _f32(ms); dw = np.empty_like(w)
    lib().oq_weight_quant_bwd(_p(g), _p(w), _p(ms), _p(dw), c_long(w.size)); return dw


def corr_fwd(x, eps=0.0):
    x = _f32(x); B, F = x.shape; G = np.empty((B, B), np.float32)
    lib().oq_corr_fwd(_p(x), c_int(B), c_long(F), c_float(eps), _p(G)); return G


def corr_bwd(dG, x, eps=0.0):
    x, dG = _f32(x), _f32(dG); B, F = x.shape; dx = np.empty_like(x)
    lib().oq_corr_bwd(_p(dG), _p(x), c_int(B), c_long(F), c_float(eps), _p(dx)); return dx


def site_fwd(x, k, r, eps=0.0):
    x = _f32(x); B = x.shape[0]; F = x.size // B
    xq, D = np.empty_like(x), np.empty((B, B), np.float32)
    lib().oq_site_fwd(_p(x), c_int(B), c_long(F), c_int(k), c_float(r), c_float(eps), _p(xq), _p(D)); return xq, D


def site_bwd(g, dD, x, r, eps=0.0):
    x, dD = _f32(x), _f32(dD); B = x.shape[0]; F = x.size // B
    g = None if g is None else _f32(g)
    dx = np.empty_like(x)
    lib().oq_site_bwd(_p(g), _p(dD), _p(x), c_int(B), c_long(F), c_float(r), c_float(eps), _p(dx)); return dx


def admm_loss(D, A, gamma, mu, rho):
    D, A, gamma = _f32(D), _f32(A), _f32(gamma); b, dim = D.shape[0], A.shape[0]
    loss = np.empty(1, np.float32); dD = np.empty_like(D); dA = np.empty_like(A); dg = np.empty_like(A)
    lib().oq_admm_loss(_p(D), c_int(b), _p(A), _p(gamma), c_int(dim), c_float(mu), c_float(rho), _p(loss), _p(dD), _p(dA), _p(dg))
    return float(loss[0]), dD, dA, dg


def admm_update(D, A, gamma, mu, rho):
    D = _f32(D); A, gamma = _f32(A).copy(), _f32(gamma).copy(); b, dim = D.shape[0], A.shape[0]
    lib().oq_admm_update(_p(D), c_int(b), _p(A), _p(gamma), c_int(dim), c_float(mu), c_float(rho)); return A, gamma


def sgd_step(p, g, buf, lr, mom, damp, wd, nesterov, first):
    p, g = _f32(p).copy(), _f32(g).copy()
    buf = np.zeros_like(p) if buf is None else _f32(buf).copy()
    lib().oq_sgd_step(_p(p), _p(g), _p(buf), c_long(p.size), c_float(lr), c_float(mom), c_float(damp), c_float(wd), c_int(nesterov), c_int(first))
    return p, g, buf


def sgd_grad_approx(d, w_cdf, w_pdf, bitW, lam, lam2):
    d, w_cdf, w_pdf = _f32(d), _f32(w_cdf), _f32(w_pdf); out = np.empty_like(d)
    lib().oq_sgd_grad_approx(_p(d), _p(w_cdf), _p(w_pdf), _p(out), c_long(d.size), c_int(bitW), c_float(lam), c_float(lam2)); return out


def bn_fold_ab(z, C, nhwc, gamma, beta, bn_eps=1e-5):
    """z: [B,F] in memory order (F = C*HW); returns ab [2,C], save [2,C], unbiased variance [C]."""
    z = _f32(z); B = z.shape[0]; HW = z.size // B // C
    ab, save, vu = np.empty((2, C), np.float32), np.empty((2, C), np.float32), np.empty(C, np.float32)
    lib().oq_bn_fold_ab(_p(z), c_int(B), c_int(C), c_long(HW), c_int(nhwc), _p(None if gamma is None else _f32(gamma)),
                        _p(None if beta is None else _f32(beta)), c_float(bn_eps), _p(ab), _p(save), _p(vu))
    return ab, save, vu


def bn_site_fwd(z, C, nhwc, ab, k, r, eps=0.0, residual=None, relu=False):
    z, ab = _f32(z), _f32(ab); B = z.shape[0]; HW = z.size // B // C
    res = None if residual is None else _f32(residual)
    y, D, x = np.empty_like(z), np.empty((B, B), np.float32), np.empty_like(z)
    lib().oq_bn_site_fwd(_p(z), c_int(B), c_int(C), c_long(HW), c_int(nhwc), _p(ab), c_int(k), c_float(r), c_float(eps),
                         _p(res), c_int(int(relu)), _p(y), _p(D), _p(x))
    return y, D, x


def bn_site_bwd(g_y, dD, z, C, nhwc, ab, save, y_relu, r, eps=0.0):
    z, ab, save, dD = _f32(z), _f32(ab), _f32(save), _f32(dD); B = z.shape[0]; HW = z.size // B // C
    g_y = None if g_y is None else _f32(g_y)
    y_relu = None if y_relu is None else _f32(y_relu)
    dz, dres, dx = np.empty_like(z), np.empty_like(z), np.empty_like(z)
    dg, db = np.empty(C, np.float32), np.empty(C, np.float32)
    lib().oq_bn_site_bwd(_p(g_y), _p(dD), _p(z), c_int(B), c_int(C), c_long(HW), c_int(nhwc), _p(ab), _p(save), _p(y_relu),
                         c_float(r), c_float(eps), _p(dz), _p(dg), _p(db), _p(dres), _p(dx))
    return dz, dg, db, dres, dx


def bn_apply(z, C, nhwc, ab):
    z, ab = _f32(z), _f32(ab); B = z.shape[0]; HW = z.size // B // C
    x = np.empty_like(z)
    lib().oq_bn_apply(_p(z), c_int(B), c_int(C), c_long(HW), c_int(nhwc), _p(ab), _p(x)); return x


def corr_xy_fwd(x, y, eps=0.0):
    x, y = _f32(x), _f32(y); B, F = x.shape; G = np.empty((B, B), np.float32)
    lib().oq_corr_xy_fwd(_p(x), _p(y), c_int(B), c_long(F), c_float(eps), _p(G)); return G


def corr_xy_bwd(dG, x, y, eps=0.0):
    x, y, dG = _f32(x), _f32(y), _f32(dG); B, F = x.shape
    dx, dy = np.empty_like(x), np.empty_like(y)
    lib().oq_corr_xy_bwd(_p(dG), _p(x), _p(y), c_int(B), c_long(F), c_float(eps), _p(dx), _p(dy)); return dx, dy
