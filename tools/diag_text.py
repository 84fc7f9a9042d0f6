"""dev tool: where does the text tower's output depend on the launch (rows / position)?  gpurun_out/diag_text.log"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from clip_glass_amd import synth
from clip_glass_amd.engine import Engine
g = np.load(os.path.join(ROOT, "tests", "golden", "clip_text_full.npz"))
V = (128, 2, 2, 8, 32, 512)
sd = synth.make_state(synth.clip_visual_spec(V[0], V[1], V[3], V[4], V[5]), 0)
sd.update(synth.make_state(synth.clip_text_spec(), 0))
e = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1, max_pop=64, clip=V, noise_mode=0)
e.load_state(sd); e.finalize()
tok = g["tokens"].astype(np.int64)
a = e.encode_text(tok)
out = []
def cmp(tag, x, y):
    bad = (x != y)
    out.append("%-28s mismatched %6d / %d  rows with mismatch %s  max|d| %.3e" % (tag, bad.sum(), bad.size, np.nonzero(bad.any(axis=1))[0].tolist(), np.abs(x - y).max()))
cmp("repeat 8", e.encode_text(tok), a)
perm = np.array([3, 1, 7, 0, 5, 2, 6, 4])
cmp("permuted 8", e.encode_text(tok[perm]), a[perm])
for reps in (2, 3, 4, 8):
    t = np.tile(tok, (reps, 1))
    cmp("%d rows" % (8 * reps), e.encode_text(t), np.tile(a, (reps, 1)))
cmp("1 row", e.encode_text(tok[:1]), a[:1])
cmp("2 rows", e.encode_text(tok[:2]), a[:2])
cmp("7 rows", e.encode_text(tok[:7]), a[:7])
open(os.path.join(ROOT, "gpurun_out", "diag_text.log"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
