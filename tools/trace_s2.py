"""dev tool: stage phase stamps of one mid-grid workgroup of conv_s2 (conv_s2_kernel<trace>; GLASS_S2_TRACE)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = "gpurun_out/s2_trace.txt"
os.environ["GLASS_S2_TRACE"] = path
if os.path.exists(path):
    os.remove(path)
from clip_glass_amd import ops
rs = np.random.RandomState(0)
for B, R, Cin, Cout in ((32, 256, 64, 128), (32, 64, 256, 512)):
    hb = rs.randn(B, R + 1, R + 1, Cin).astype(np.float32); xs = rs.randn(B, R // 2, R // 2, Cin).astype(np.float32)
    w1 = rs.randn(Cout, Cin, 3, 3).astype(np.float32); ws = rs.randn(Cout, Cin, 1, 1).astype(np.float32)
    ops.conv(hb, w1, skip=(xs, ws), stride=2, pad=0, bias=rs.randn(Cout).astype(np.float32), act=True, out_scale=2.0 ** -0.5, impl=5)
blocks = open(path).read().split("# ")[1:]
names = ["wait", "barrier", "issue", "mfma"]
NS = 4      # stages per chunk
for blk in blocks:
    lines = blk.strip().split("\n")
    print(lines[0][:70])
    a = np.array([[int(v) for v in l.split()] for l in lines[1:]], dtype=np.float64)
    T = a[:, 2:].reshape(-1, 7, 8)
    n = int((T[:, 4, 0] > 0).sum())
    for f in range(NS):
        idx = [g for g in range(NS, n - 1) if g % NS == f]
        for w_ in (0, 3, 7):
            d = [np.mean([T[g, ph + 1, w_] - T[g, ph, w_] for g in idx]) for ph in range(4)]
            tot = np.mean([T[g + 1, 0, w_] - T[g, 0, w_] for g in idx])
            print("  f=%d wave %d " % (f, w_) + " ".join("%s=%.0f" % (nm, v) for nm, v in zip(names, d)) + "  stage=%.0f" % tot)
    ep = [g for g in range(n) if T[g, 6, 0] > 0]
    if ep:
        print("  epilogue (wave 0): barrier=%.0f body=%.0f" % (np.mean([T[g, 5, 0] - T[g, 4, 0] for g in ep]), np.mean([T[g, 6, 0] - T[g, 5, 0] for g in ep])))
    print("  total per chunk (wave 0): %.0f" % np.mean([T[g + NS, 0, 0] - T[g, 0, 0] for g in range(NS, n - NS - 1)]))
    for g in range(min(n, 10)):
        print("   stage", g, "wave0:", " ".join("%.0f" % (T[g, ph, 0] - T[g, 0, 0]) for ph in range(5)), " wave7:", " ".join("%.0f" % (T[g, ph, 7] - T[g, 0, 0]) for ph in range(5)))
