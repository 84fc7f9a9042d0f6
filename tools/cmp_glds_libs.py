"""Bitwise comparison of the LDS-DMA conv family between two builds of libglass.so:  python tools/cmp_glds_libs.py dump <file>  under each GLASS_LIB, then  cmp <a> <b>."""
import math, sys
import numpy as np

def cases():
    from clip_glass_amd import ops
    rng = np.random.default_rng(29)
    out = {}
    for tag, (B, H, W, Cin, Cout) in {"p128": (8, 128, 128, 128, 256), "p512": (4, 64, 64, 512, 512), "p256": (4, 64, 64, 256, 128)}.items():
        x = rng.standard_normal((B, H, W, Cin)).astype(np.float16).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, 3, 3)) / math.sqrt(9 * Cin)).astype(np.float32)
        ds = rng.uniform(0.5, 2.0, (B, Cout)).astype(np.float32)
        noise = rng.standard_normal((B, H, W)).astype(np.float32)
        bias = rng.standard_normal(Cout).astype(np.float32) * 0.2
        kw = dict(dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True, out_scale=0.7)
        out[tag] = ops.conv(x, w, impl=5, **kw)
        out[tag + "_plain"] = ops.conv(x, w, impl=5, bias=bias, act=True)
        xs = np.full((B, H // 2, W // 2, Cin), np.nan, dtype=np.float32)
        out[tag + "_xsy"] = ops.conv(x, w, impl=5, xs_out=xs, bias=bias, act=True, out_scale=0.7)
        out[tag + "_xs"] = xs
        sn = rng.uniform(-1.0, 1.0, (B, Cin)).astype(np.float32)
        out[tag + "_sty"] = ops.conv(x, w, impl=5, sn=sn, **kw)
    return out

if sys.argv[1] == "dump":
    np.savez(sys.argv[2], **cases())
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print("%-12s equal %s  max|d| %.3e  n_diff %d / %d" % (k, np.array_equal(a[k], b[k]), d.max(), int((d > 0).sum()), d.size))
