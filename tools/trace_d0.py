"""dev tool (build with `make -C clip_glass_amd/csrc TRACE=1`): phase timestamps of one workgroup of dblock0_kernel (GLASS_D0_TRACE)."""
import os, sys, math, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = "gpurun_out/d0_trace.txt"
os.environ["GLASS_D0_TRACE"] = path
if os.path.exists(path):
    os.remove(path)
from clip_glass_amd import engine
engine.load_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lib", "libglass_trace.so"))   # conv_d0.o built with -DGLASS_DEV_TRACE
from clip_glass_amd import ops, synth
B, R = 16, 1024
rs = np.random.RandomState(0)
y = rs.randn(B, 3, R, R).astype(np.float32) * 0.8
fw = rs.randn(32, 3).astype(np.float32) / math.sqrt(3); fb = rs.randn(32).astype(np.float32) * 0.3
w0 = rs.randn(32, 32, 3, 3).astype(np.float32); b0 = rs.randn(32).astype(np.float32) * 0.3
w1 = rs.randn(64, 32, 3, 3).astype(np.float32); b1 = rs.randn(64).astype(np.float32) * 0.3; ws = rs.randn(64, 32, 1, 1).astype(np.float32)
ops.dblock0(y, fw, fb, w0, b0, w1, ws, b1)
rows = [l.split() for l in open(path) if not l.startswith('#')]
a = np.array([[int(v) for v in r] for r in rows], dtype=np.float64)
T = a[:, 2:].reshape(-1, 16, 4)
names = ["B0", "P1", "B1", "P2", "mfma0", "epi+hfir", "B2", "P4", "B3", "mfma1", "epi1+store", "loop"]
real = [i for i in range(4, T.shape[0] - 1) if T[i, 11, 0] > 0 and T[i + 1, 0, 0] > 0]
for w in range(4):
    print("wave", w, "conv0 epilogue -> row image %.0f, row image -> FIR operands landed %.0f, FIR + ring writes %.0f" % tuple(
        np.mean([T[i, b_, w] - T[i, a_, w] for i in real if T[i, 12, w] > 0]) for a_, b_ in ((5, 12), (12, 13), (13, 6))))
    d = [np.mean([T[i, ph + 1, w] - T[i, ph, w] for i in real]) for ph in range(11)]
    d.append(np.mean([T[i + 1, 0, w] - T[i, 11, w] for i in real]))
    print("wave", w, " ".join("%s=%.0f" % (n, v) for n, v in zip(names, d)), "total=%.0f" % sum(d))
