"""Winograd F(2x2, 3x3) in the engine's arithmetic vs the direct form (numerics half of the go / no-go).

Operands are what a conv_gldsp layer sees: fp16 activations (style already applied), fp16 weights, fp32 accumulation on the matrix
cores (products of fp16 values are exact in fp32).  Winograd adds two roundings the direct form does not have: the transformed
weights U = G g G^T (fp32 at finalize -> fp16 operand) and the transformed input V = B^T d B (sums of four fp16 activations -> fp16
operand; formed either in packed fp16 on the VALU while staging, or in fp32 and rounded once).  Errors are reported against an fp64
convolution of the SAME fp16 operands, relative to the output's rms — the direct kernel's own error on that scale is the accumulation
order only (~1e-7) plus the fp16 store (4.9e-4 max, 1.4e-4 rms relative per element).

  python tools/winograd_numerics.py            # CPU, ~10 s
"""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
C, O, H = 128, 128, 32            # one 128 -> 128 layer slice at 32 x 32 is enough for the error statistics (K = 9 * 128)
x = torch.randn(2, C, H, H, dtype=torch.float64)
x = (x * torch.rand(2, C, 1, 1, dtype=torch.float64) * 1.5).to(torch.float16)          # "x * s / smax": per-channel scales <= 1.5
w = (torch.randn(O, C, 3, 3, dtype=torch.float64) / np.sqrt(9 * C)).to(torch.float16)

ref = F.conv2d(x.double(), w.double(), padding=1)
rms = ref.pow(2).mean().sqrt()

direct = F.conv2d(x.float(), w.float(), padding=1)                                     # fp32 accumulate of exact products
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd(v_dtype, v_in_fp16_arith):
    xp = F.pad(x, (1, 1, 1, 1))
    # tiles: [N, C, th, tw, 4, 4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    if v_in_fp16_arith:            # packed-fp16 VALU: every add rounds to fp16 (row pass, then column pass)
        dh = d.to(torch.float16)
        r = torch.stack([dh[..., 0, :] - dh[..., 2, :], dh[..., 1, :] + dh[..., 2, :], dh[..., 2, :] - dh[..., 1, :], dh[..., 1, :] - dh[..., 3, :]], dim=-2)
        V = torch.stack([r[..., 0] - r[..., 2], r[..., 1] + r[..., 2], r[..., 2] - r[..., 1], r[..., 1] - r[..., 3]], dim=-1)
    else:
        V = torch.einsum("ia,nctsab,jb->nctsij", Bt, d.double(), Bt).to(v_dtype)
    U = torch.einsum("ia,ocab,jb->ocij", G, w.double(), G).to(torch.float16)            # finalize(): fp32/fp64 transform, fp16 operand
    M = torch.einsum("ocij,nctsij->notsij", U.float(), V.float())                      # matrix cores: exact products, fp32 sums
    Y = torch.einsum("ia,notsab,jb->notsij", At.float(), M, At.float())               # output transform in fp32 (epilogue)
    N_, O_, th, tw = Y.shape[:4]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N_, O_, th * 2, tw * 2)


def report(name, y):
    e = (y.double() - ref)
    print("%-62s rms err / rms(out) = %.3e   max err / rms(out) = %.3e" % (name, e.pow(2).mean().sqrt() / rms, e.abs().max() / rms))


report("direct, fp32 accumulate (conv_gldsp today)", direct)
report("direct, result stored as fp16 (what every layer does anyway)", direct.to(torch.float16))
report("winograd, V formed in fp32 and rounded once to fp16", winograd(torch.float16, False))
report("winograd, V formed in packed fp16 on the VALU", winograd(torch.float16, True))
report("winograd, V kept in fp32 (transform error of U only)", winograd(torch.float32, False))
vmax = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2).double().abs().amax()
print("max |d| = %.2f -> |V| <= 4 max|d| = %.2f (fp16 range 65504: the style normalisation that keeps x in range has to leave 4x headroom)" % (vmax, 4 * vmax))
