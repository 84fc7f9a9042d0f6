import sys, math, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from clip_glass_amd import ops
rng = np.random.default_rng(41)
B, H, W, C = 3, 128, 256, 64
x = rng.standard_normal((B, H, W, C)).astype(np.float16).astype(np.float32)
w = (rng.standard_normal((C, C, 3, 3)) / math.sqrt(9 * C)).astype(np.float32)
ds = rng.uniform(0.5, 2.0, (B, C)).astype(np.float32)
noise = rng.standard_normal((B, H, W)).astype(np.float32)
bias = rng.standard_normal(C).astype(np.float32) * 0.2
wrgb = (rng.standard_normal((3, C)) / math.sqrt(C)).astype(np.float32)
brgb = (rng.standard_normal(3) * 0.1).astype(np.float32)
srgb = rng.uniform(0.2, 1.0, (B, C)).astype(np.float32)
smax = rng.uniform(0.5, 3.0, B).astype(np.float32)
for name, kw3 in (("full", dict(dscale=ds, noise=noise, noise_strength=0.3, batch_size=1, bias=bias, act=True)),
                  ("no noise", dict(dscale=ds, bias=bias, act=True)),
                  ("no noise, no dscale", dict(bias=bias, act=True)),
                  ("nothing", dict())):
    feat = ops.conv(x, w, impl=5, **kw3)
    got, fy = ops.conv(x, w, impl=5, torgb=dict(w=wrgb, b=brgb, sn=srgb, smax=smax, yprev=None), both=True, **kw3)
    fb = np.abs(fy - feat) > 1e-3
    print(name, ': feature map of the toRGB instance: bad frac %.4f' % fb.mean(), 'per tilecol', [round(float(fb[:, :, c*32:(c+1)*32].mean()), 3) for c in range(8)], 'per row%8', [round(float(fb[:, r::8].mean()), 3) for r in range(8)])
    d = (fy - feat)
    idx = np.argwhere(fb)[:3]
    for i in idx: print('   at', tuple(i), 'got', fy[tuple(i)], 'want', feat[tuple(i)])
kw3 = dict()
feat = ops.conv(x, w, impl=5, **kw3)
got, fy = ops.conv(x, w, impl=5, torgb=dict(w=wrgb, b=brgb, sn=srgb, smax=smax, yprev=None), both=True, **kw3)
fb = np.abs(fy - feat) > 1e-3
print("bitmap (fraction of 64 channels bad, x10) of b=0 rows 0..7, cols 32..63")
for r in range(8):
    print(' '.join('%d' % min(9, int(10 * fb[0, r, c].mean())) for c in range(32, 64)))
print("per channel bad frac in that tile", [round(float(fb[0, 0:8, 32:64, ch].mean()), 2) for ch in range(64)])
# which single (chunk, tap) contribution explains the error?  try: error == contribution of input shifted ...
err = (fy - feat)[0, 0:8, 32:64]
print("mean abs err per row", [round(float(np.abs(err[r]).mean()), 4) for r in range(8)])
