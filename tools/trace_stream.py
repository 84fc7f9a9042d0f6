"""dev tool: phase timestamps of one workgroup of conv_stream (G 32->32 conv at 1024^2) (GLASS_STREAM_TRACE)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = "gpurun_out/stream_trace.txt"
os.environ["GLASS_STREAM_TRACE"] = path
if os.path.exists(path):
    os.remove(path)
from clip_glass_amd import ops
rs = np.random.RandomState(0)
B, H = 8, 1024
x = rs.randn(B, H, H, 32).astype(np.float32); w = rs.randn(32, 32, 3, 3).astype(np.float32)
noise = rs.randn(B, H, H).astype(np.float32); bias = rs.randn(32).astype(np.float32)
y = ops.conv(x, w, noise=noise, noise_strength=0.1, batch_size=1, bias=bias, act=True, impl=4)
rows = [l.split() for l in open(path) if not l.startswith('#')]
a = np.array([[int(v) for v in r] for r in rows], dtype=np.float64)
T = a[:, 2:].reshape(-1, 7, 4)
n = T.shape[0]
names = ["sync0", "stage", "sync1", "issue", "mfma", "xs", "epilogue"]
for w_ in range(4):
    d = [(T[3:n - 1, ph + 1, w_] - T[3:n - 1, ph, w_]).mean() for ph in range(6)]
    d.append((T[4:n, 0, w_] - T[3:n - 1, 6, w_]).mean())
    print("wave", w_, " ".join("%s=%.0f" % (nm, v) for nm, v in zip(names, d)), "total=%.0f" % sum(d))
