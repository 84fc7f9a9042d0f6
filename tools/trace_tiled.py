"""dev tool: K-stage phase stamps of one mid-grid workgroup of the stride-2 conv (conv_tiled<3,2,4,128,skip,spl,deep,trace>; GLASS_TILED_TRACE)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = "gpurun_out/tiled_trace.txt"
os.environ["GLASS_TILED_TRACE"] = path
if os.path.exists(path):
    os.remove(path)
from clip_glass_amd import ops
rs = np.random.RandomState(0)
for B, R, Cin, Cout in ((32, 256, 64, 128), (32, 64, 256, 512)):
    hb = rs.randn(B, R + 1, R + 1, Cin).astype(np.float32); xs = rs.randn(B, R // 2, R // 2, Cin).astype(np.float32)
    w1 = rs.randn(Cout, Cin, 3, 3).astype(np.float32); ws = rs.randn(Cout, Cin, 1, 1).astype(np.float32)
    ops.conv(hb, w1, skip=(xs, ws), stride=2, pad=0, bias=rs.randn(Cout).astype(np.float32), act=True, out_scale=2.0 ** -0.5, impl=2)
blocks = open(path).read().split("# ")[1:]
names = ["barrier0", "store", "barrier1", "issue", "mfma"]
for blk in blocks:
    lines = blk.strip().split("\n")
    print(lines[0][:90])
    a = np.array([[int(v) for v in l.split()] for l in lines[1:]], dtype=np.float64)
    T = a[:, 2:].reshape(-1, 7, 4)
    n = int((T[:, 5, 0] > 0).sum())
    for w_ in range(4):
        print("  wave", w_, "operand wait=%.0f  LDS writes=%.0f" % ((T[1:n, 6, w_] - T[1:n, 1, w_]).mean(), (T[1:n, 2, w_] - T[1:n, 6, w_]).mean()))
        d = [(T[1:n, ph + 1, w_] - T[1:n, ph, w_]).mean() for ph in range(5)]
        gap = (T[2:n, 0, w_] - T[1:n - 1, 5, w_]).mean()
        print("  wave", w_, " ".join("%s=%.0f" % (nm, v) for nm, v in zip(names, d)), "next=%.0f" % gap, "stage=%.0f cycles over %d stages" % (sum(d) + gap, n - 1))
    for st in range(min(n, 7)):
        print("   stage", st, "wave0 phases:", " ".join("%.0f" % (T[st, ph, 0] - T[st, 0, 0]) for ph in (0, 1, 6, 2, 3, 4, 5)))
