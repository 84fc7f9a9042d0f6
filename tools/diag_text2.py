"""dev tool: which op of the text tower depends on a text's POSITION in the batch?  Each op alone, texts permuted."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clip_glass_amd import ops
rs = np.random.RandomState(0)
n, L, W = 8, 77, 512
perm = np.array([3, 1, 7, 0, 5, 2, 6, 4])
out = []
def cmp(tag, x, y):
    bad = x != y
    out.append("%-34s mismatched %7d / %d  max|d| %.3e" % (tag, bad.sum(), bad.size, np.abs(x - y).max()))
def pt(a):      # permute whole texts of a [n*L, ...] row matrix
    return a.reshape(n, L, -1)[perm].reshape(n * L, -1)
a = rs.randn(n * L, W).astype(np.float32)
for N, mode, name in ((3 * W, 0, "qkv (f16 out)"), (W, 3, "out-proj (f32 out)"), (4 * W, 1, "fc quickgelu"), (W, 3, "N=512 f32")):
    w = (rs.randn(N, W) / np.sqrt(W)).astype(np.float32); b = rs.randn(N).astype(np.float32) * 0.1
    cmp("gemm " + name, ops.gemm(pt(a), w, b, mode=mode, impl=2), pt(ops.gemm(a, w, b, mode=mode, impl=2)))
acc = rs.randn(n * L, W).astype(np.float32)
w = (rs.randn(W, 4 * W) / np.sqrt(4 * W)).astype(np.float32); b = rs.randn(W).astype(np.float32) * 0.1
a4 = rs.randn(n * L, 4 * W).astype(np.float32)
cmp("gemm residual K=2048", ops.gemm(pt(a4), w, b, mode=2, impl=2, acc=pt(acc)), pt(ops.gemm(a4, w, b, mode=2, impl=2, acc=acc)))
qkv = rs.randn(n * L, 3 * W).astype(np.float32)
cmp("attention causal L=77", ops.attention(pt(qkv), n, L, 8, causal=True), pt(ops.attention(qkv, n, L, 8, causal=True)))
g = rs.randn(W).astype(np.float32); bb = rs.randn(W).astype(np.float32)
cmp("layernorm", ops.layernorm(pt(a), g, bb), pt(ops.layernorm(a, g, bb)))
wt = (rs.randn(W, 512) / np.sqrt(W)).astype(np.float32)
x8 = rs.randn(n, W).astype(np.float32)
cmp("dense 8 rows", ops.dense(x8[perm], wt), ops.dense(x8, wt)[perm])
open(os.path.join(ROOT, "gpurun_out", "diag_text2.log"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
