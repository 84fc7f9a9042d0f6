import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ["GLASS_DOWN_TRACE"] = "gpurun_out/down_trace.txt"
from clip_glass_amd import ops
B, R = 16, 1024
rs = np.random.RandomState(0)
h = rs.randn(B, R, R, 32).astype(np.float32); x = rs.randn(B, R, R, 32).astype(np.float32)
w1 = rs.randn(64, 32, 3, 3).astype(np.float32); ws = rs.randn(64, 32, 1, 1).astype(np.float32); b1 = rs.randn(64).astype(np.float32)
y = ops.dblock_down(h, x, w1, ws, b1)
print(y.shape, float(np.abs(y).mean()))
