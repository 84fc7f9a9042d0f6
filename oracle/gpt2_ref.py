"""Oracle: GPT-2 greedy decode + img2txt scoring (config GPT2), torch-CPU fp32.  TEST INFRASTRUCTURE —
see oracle/__init__.py.  Restates gpt2/model.py, gpt2/sample.py and models.py:32-62 of the reference;
weights under the keys `gpt2.transformer.*` (clip_glass_amd/synth.py gpt2_spec)."""
import math

import torch
import torch.nn.functional as F


def _ln(x, w, b, eps=1e-5):
    # gpt2/model.py:15-28 (TF-style LayerNorm, eps inside the sqrt)
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + eps)) + b


def _gelu(x):
    # gpt2/model.py:12-13
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))


def _conv1d(x, w, b):
    # gpt2/model.py:39-43: addmm(bias, x, weight[nx, nf])
    return x @ w + b


def forward(sd, input_ids, past=None, n_head=None):
    """gpt2/model.py:144-177 + 194-211: returns (logits [B, T, V], presents)."""
    p = "gpt2.transformer."
    wte, wpe = sd[p + "wte.weight"], sd[p + "wpe.weight"]
    D = wte.shape[1]
    n_head = n_head or D // 64
    past_len = 0 if past is None else past[0][0].shape[-2]
    T = input_ids.shape[-1]
    pos = torch.arange(past_len, past_len + T)
    h = wte[input_ids] + wpe[pos][None]
    presents = []
    i = 0
    while p + "h.%d.ln_1.weight" % i in sd:
        q = p + "h.%d." % i
        a = _conv1d(_ln(h, sd[q + "ln_1.weight"], sd[q + "ln_1.bias"]), sd[q + "attn.c_attn.weight"], sd[q + "attn.c_attn.bias"])
        qq, kk, vv = a.split(D, dim=2)
        B = a.shape[0]
        qq = qq.view(B, T, n_head, D // n_head).permute(0, 2, 1, 3)
        kk = kk.view(B, T, n_head, D // n_head).permute(0, 2, 1, 3)
        vv = vv.view(B, T, n_head, D // n_head).permute(0, 2, 1, 3)
        if past is not None:                                       # model.py:87-90
            kk = torch.cat((past[i][0], kk), dim=-2)
            vv = torch.cat((past[i][1], vv), dim=-2)
        presents.append((kk, vv))
        w = (qq @ kk.transpose(-1, -2)) / math.sqrt(vv.shape[-1])  # model.py:59-62 (scale=True)
        nd, ns = w.shape[-2], w.shape[-1]
        b = torch.tril(torch.ones(ns, ns))[ns - nd:ns, :ns]
        w = w * b - 1e10 * (1 - b)                                 # model.py:63-64
        w = torch.softmax(w, dim=-1)
        a = (w @ vv).permute(0, 2, 1, 3).reshape(B, T, D)
        h = h + _conv1d(a, sd[q + "attn.c_proj.weight"], sd[q + "attn.c_proj.bias"])
        m = _gelu(_conv1d(_ln(h, sd[q + "ln_2.weight"], sd[q + "ln_2.bias"]), sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"]))
        h = h + _conv1d(m, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"])
        i += 1
    h = _ln(h, sd[p + "ln_f.weight"], sd[p + "ln_f.bias"])
    return h @ wte.t(), presents                                   # tied lm_head (model.py:181-191)


def sample_sequence(sd, context, length, temperature=0.7, top_k=40, detail=None):
    """gpt2/sample.py:21-36 with sample=False (models.py:50-60)."""
    prev = context
    output = context
    past = None
    margins = []
    with torch.no_grad():
        for _ in range(length):
            logits, past = forward(sd, prev, past)
            logits = logits[:, -1, :] / temperature
            values, _ = torch.topk(logits, top_k)                  # sample.py:10-19 top_k_logits
            logits = torch.where(logits < values[:, -1:], torch.full_like(logits, -1e10), logits)
            probs = F.softmax(logits, dim=-1)
            _, prev = torch.topk(probs, k=1, dim=-1)
            top2 = torch.topk(logits, 2).values
            margins.append((top2[:, 0] - top2[:, 1]) * temperature)
            output = torch.cat((output, prev), dim=1)
    if detail is not None:
        detail["margins"] = torch.stack(margins, dim=1)            # raw-logit gap between the top two tokens
    return output


def parse_out(out, dim_z, eot, decode, max_text_len):
    """models.py:32-42"""
    texts = []
    for seq in out:
        seq = [int(t) for t in seq]
        text = seq[dim_z:seq.index(eot)] if eot in seq else seq[dim_z:]
        texts.append(decode(text)[:max_text_len])
    return texts
