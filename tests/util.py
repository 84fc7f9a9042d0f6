"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_DIR = os.path.join(ROOT, "gpurun_out")


def diag(msg):
    """Append a line to gpurun_out/diag.log (merged back from the GPU box) and print it."""
    print(msg)
    try:
        os.makedirs(DIAG_DIR, exist_ok=True)
        with open(os.path.join(DIAG_DIR, "diag.log"), "a") as f:
            f.write(msg + "\n")
    except OSError:
        pass


def check(name, got, ref, rtol_scale, atol=0.0):
    """|got-ref| <= rtol_scale * max|ref| + atol everywhere; logs error statistics either way."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    tol = rtol_scale * scale + atol
    worst = np.unravel_index(int(err.argmax()), err.shape) if err.size else ()
    nbad = int((err > tol).sum())
    diag("[check] %-34s max_err %.3e (at %s: got %.5g ref %.5g) rms_err %.3e scale %.3e tol %.3e bad %d/%d %s"
         % (name, err.max() if err.size else 0, worst, got[worst] if err.size else 0, ref[worst] if err.size else 0,
            float(np.sqrt((err ** 2).mean())) if err.size else 0, scale, tol, nbad, err.size,
            "OK" if nbad == 0 and np.isfinite(got).all() else "FAIL"))
    assert np.isfinite(got).all(), "%s: non-finite values" % name
    assert nbad == 0, "%s: %d/%d elements exceed tol %.3e (max err %.3e at %s)" % (name, nbad, err.size, tol, err.max(), worst)


# Second objective (D logit / hinge).  Element-wise |got - ref| <= D_ATOL + D_RTOL * |ref|.  Round 2 used 5e-3 * max|ref| + 2e-3
# (~4.5e-3 absolute at |D| ~ 0.5): an order-of-magnitude regression would have passed.  The fp16-storage engine delivers
# 3e-4 .. 1.4e-3 absolute over the mini / mid / ffhq cases (gpurun_out/diag.log), so the bar sits at 1.5e-3 + 1.5e-3 |ref|.
D_ATOL, D_RTOL = 1.5e-3, 1.5e-3
# WHAT THESE BARS ARE: a REGRESSION GUARD, not a parity bar.  north_star states a tolerance for the CLIP similarity only
# (1e-3 relative; asserted as such wherever `sim` is compared); for the second objective it states none, so the bar below is derived from
# the engine's own measured error against the oracle / the reference-generated fixtures and exists to catch the D path getting WORSE.
# Per-architecture bars: ~1.35 x the largest error the engine delivers on that architecture over every case of the GPU
# suite (gpurun_out/diag.log of the r04 run; the runs are bitwise reproducible), so that a 2 x regression of the D path fails on
# the architecture where it happens instead of hiding under the loosest case's bar.  north_star states no D tolerance.
# Largest |error| of the r04 GPU run per architecture: mini 7.3e-4, mid 1.56e-3 (one case: P8 bs4 vs the oracle; its other cases <= 6.7e-4),
# ffhq 6.1e-4, church 1.8e-4, car 8.0e-4.
D_TOL = {"mini": (1.0e-3, 5e-4), "mid": (2.0e-3, 5e-4), "ffhq": (9e-4, 5e-4), "church": (4e-4, 5e-4), "car": (1.1e-3, 5e-4)}


def check_logits(name, got, ref, atol=None, rtol=None, case=None):
    a0, r0 = D_TOL.get(case, (D_ATOL, D_RTOL))
    atol = a0 if atol is None else atol
    rtol = r0 if rtol is None else rtol
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, got.shape, ref.shape)
    err = np.abs(got - ref)
    tol = atol + rtol * np.abs(ref)
    worst = np.unravel_index(int((err - tol).argmax()), err.shape) if err.size else ()
    nbad = int((err > tol).sum())
    diag("[check] %-34s max_err %.3e (worst vs tol at %s: got %.5g ref %.5g tol %.3e) bad %d/%d %s"
         % (name, err.max() if err.size else 0, worst, got[worst] if err.size else 0, ref[worst] if err.size else 0,
            tol[worst] if err.size else 0, nbad, err.size, "OK" if nbad == 0 and np.isfinite(got).all() else "FAIL"))
    assert np.isfinite(got).all(), "%s: non-finite values" % name
    assert nbad == 0, "%s: %d/%d elements exceed %.1e + %.1e |ref| (max err %.3e)" % (name, nbad, err.size, atol, rtol, err.max())


def nhwc(t):
    return np.ascontiguousarray(np.asarray(t).transpose(0, 2, 3, 1))


def nchw(a):
    return np.ascontiguousarray(np.asarray(a).transpose(0, 3, 1, 2))


def style_tables(latent, dense_w, dense_b, W, demod=True, eps=1e-8):
    """Host math of the engine's style path (csrc/engine.cpp run_styles) in numpy:
    s = latent @ (A/sqrt(L))^T + b ; sn = s/smax ; dscale = rsqrt(sum sn^2 Wsq + eps/smax^2) (= d*smax)."""
    L = dense_w.shape[1]
    s = latent @ (dense_w / np.sqrt(L)).T + dense_b
    smax = np.maximum(np.abs(s).max(axis=1, keepdims=True), 1e-20)
    sn = s / smax
    cout, cin, ks, _ = W.shape
    coef2 = 1.0 / (cin * ks * ks)
    wsq = (W.astype(np.float64) ** 2).sum(axis=(2, 3)) * coef2        # [cout, cin]
    if demod:
        dscale = 1.0 / np.sqrt((sn.astype(np.float64) ** 2) @ wsq.T + eps / smax ** 2)
    else:
        dscale = np.repeat(smax, cout, axis=1)
    return sn.astype(np.float32), smax.astype(np.float32), dscale.astype(np.float32)
