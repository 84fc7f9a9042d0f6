"""Writes the small weight-container fixtures under tests/golden/containers/ with the REFERENCE's own writers
(build container only; TEST INFRASTRUCTURE).  The files are data — pickled dicts of tensors, no code:

  mini/G.pth, mini/D.pth   stylegan2/models.py:111-132,258-262 `.save()` of the reference Generator / Discriminator
                           ({'name','kwargs','state_dict', 'G_mapping': {...}, 'G_synthesis': {...}})
  clip_mini.pt             `clip.model.build_model(state).state_dict()` (clip/model.py:363-399 key set, fp16 weights as
                           convert_weights leaves them) + the three geometry scalars of the jit archive
  gpt2_mini.bin            HF-style GPT-2 state: no `transformer.` prefix, TF LayerNorm names (.g/.b) on some keys, the
                           `h.N.attn.bias` causal-mask buffers present (what gpt2/utils.py:10-51 load_weight is written for)

so that the product's loaders (clip_glass_amd/models.py, generator.py) are exercised on the GPU box as well, where
/root/reference does not exist.  Weights are the deterministic synthetic ones of the mini geometry."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh          # noqa: E402
import glass_models as M          # noqa: E402
from clip_glass_amd import synth  # noqa: E402

OUT = os.path.join(HERE, "containers")
CLIP_TEXT = dict(width=64, layers=2, vocab=512, ctx=16)
GPT2 = dict(n_embd=64, n_layer=2, vocab=256, n_positions=64)


def clip_mini_state(seed=0):
    c = M.CONFIGS["mini"]
    w, layers, heads, patch, res, emb = c["clip"]
    sd = synth.make_state(synth.clip_visual_spec(w, layers, patch, res, emb), seed)
    sd.update(synth.make_state(synth.clip_text_spec(width=CLIP_TEXT["width"], layers=CLIP_TEXT["layers"], ctx=CLIP_TEXT["ctx"],
                                                    vocab=CLIP_TEXT["vocab"], out_dim=emb), seed))
    return sd


def main():
    assert rh.available(), "needs /root/reference"
    name = "mini"
    c = M.CONFIGS[name]
    sd = M.make_state(name, 0)
    os.makedirs(os.path.join(OUT, name), exist_ok=True)
    rh.build_ref_G(sd, c["channels"], c["latent"], c["mapping"]).save(os.path.join(OUT, name, "G.pth"))
    rh.build_ref_D(sd, c["channels"]).save(os.path.join(OUT, name, "D.pth"))
    model = rh.build_ref_clip(clip_mini_state(), fp32=False)          # fp16 weights, as clip.load leaves them
    st = dict(model.state_dict())
    st["input_resolution"] = torch.tensor(c["clip"][4])
    st["context_length"] = torch.tensor(CLIP_TEXT["ctx"])
    st["vocab_size"] = torch.tensor(CLIP_TEXT["vocab"])
    torch.save(st, os.path.join(OUT, "clip_mini.pt"))
    g = synth.make_state(synth.gpt2_spec(**GPT2), 2)
    gmodel, _ = rh.build_ref_gpt2(g, GPT2["n_embd"], GPT2["n_layer"], GPT2["vocab"])
    hf = {}
    for k, v in gmodel.state_dict().items():
        k = k[len("transformer."):] if k.startswith("transformer.") else k
        if ".ln_1." in k or k.startswith("ln_f."):                   # TF-era LayerNorm names on part of the keys
            k = k.replace(".weight", ".g").replace(".bias", ".b")
        hf[k] = v.clone()
    torch.save(hf, os.path.join(OUT, "gpt2_mini.bin"))
    for root, _, files in os.walk(OUT):
        for f in files:
            print(os.path.relpath(os.path.join(root, f), OUT), os.path.getsize(os.path.join(root, f)))


if __name__ == "__main__":
    main()
