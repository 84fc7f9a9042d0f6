"""Index-level CPU emulation of conv_s2.hip's LDS staging (test infrastructure, no GPU): the row-parity halves, de-interleaved columns and
XOR-swizzled 16-byte pieces that the DMA writes are exactly what the MFMA fragment reads expect — every lane of every tap reads input
pixel (2 r + ky, 2 lr + tx), channel piece lc, and every byte of a half is written exactly once.  The formulas below restate
clip_glass_amd/csrc/conv_s2.hip (o_src / e_src, the xf address, the skip operand); a change there must be mirrored here."""
import numpy as np

PXR = 65
ODD_V, EVEN_V = 8 * PXR * 4, 9 * PXR * 4


def dma_image(n_vec, parity):
    """vector v of a half -> (input row, input column, logical 8-channel piece) it receives (tile-relative)."""
    out = {}
    for v in range(n_vec):
        P = v >> 2
        row, q = divmod(P, PXR)
        col = 2 * q if q < 33 else 2 * (q - 33) + 1
        lc = (v & 3) ^ ((P >> 2) & 3)
        out[v] = (2 * row + parity, col, lc)
    return out


def test_halves_hold_every_input_pixel_once():
    for n_vec, parity, rows in ((ODD_V, 1, range(1, 16, 2)), (EVEN_V, 0, range(0, 17, 2))):
        img = dma_image(n_vec, parity)
        got = sorted(img.values())
        want = sorted((r, c, lc) for r in rows for c in range(65) for lc in range(4))
        assert got == want


def test_fragment_reads_match_the_convolution():
    odd, even = dma_image(ODD_V, 1), dma_image(EVEN_V, 0)
    for wave in range(8):
        wr = wave & 3
        for i in range(2):
            r = wr * 2 + i
            for ky in range(3):
                half = odd if ky == 1 else even
                prow = r if ky == 1 else r + (ky >> 1)
                for tx in range(3):
                    for lr in range(32):
                        q = 33 + lr if (tx & 1) else lr + (tx >> 1)
                        P = prow * PXR + q
                        for lc in range(4):                       # lc = kk * 2 + kh
                            v = P * 4 + (lc ^ ((P >> 2) & 3))     # byte address / 16 of the lane's read
                            assert half[v] == (2 * r + ky, 2 * lr + tx, lc)


def test_weight_and_skip_images():
    # weights: vector v = k * 512 + t of a slot -> row tx * 128 + n, piece (v & 3) ^ ((row >> 2) & 3); read at row * 64 + ((lc ^ ((row >> 2) & 3)) << 4)
    img = {}
    for v in range(3 * 512):
        row = v >> 2
        img[v] = (row >> 7, row & 127, (v & 3) ^ ((row >> 2) & 3))
    for tx in range(3):
        for n in range(128):
            for lc in range(4):
                row = tx * 128 + n
                assert img[row * 4 + (lc ^ ((row >> 2) & 3))] == (tx, n, lc)
    # skip operand: vector v = k * 512 + t -> pixel px = v >> 2 = tile row * 32 + column
    simg = {v: ((v >> 2) >> 5, (v >> 2) & 31, (v & 3) ^ (((v >> 2) >> 2) & 3)) for v in range(2 * 512)}
    for wr in range(4):
        for i in range(2):
            for lr in range(32):
                P = (wr * 2 + i) * 32 + lr
                for lc in range(4):
                    assert simg[P * 4 + (lc ^ ((P >> 2) & 3))] == (wr * 2 + i, lr, lc)


def test_wave_round_ownership():
    # wave-round r = k * 8 + wave covers vectors r * 64 .. + 63: the per-wave DMA counts the vmcnt schedule relies on
    for n_vec, want in ((ODD_V, [5, 4, 4, 4, 4, 4, 4, 4]), (EVEN_V, [5, 5, 5, 5, 5, 4, 4, 4])):
        rounds = (n_vec + 63) // 64
        assert [sum(1 for k in range(5) if k * 8 + w < rounds) for w in range(8)] == want
