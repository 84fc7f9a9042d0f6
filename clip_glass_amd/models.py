"""Model wrappers — mirror of /root/reference/models.py for the StyleGAN2 configs.

Weight ingestion: reads the reference's own container
({'name','kwargs','state_dict', 'G_mapping': {...}, 'G_synthesis': {...}} written by
stylegan2/models.py:111-132,258-262) with torch.load on the host and hands every tensor
to the engine under its reference key.  `weights = "synthetic:<seed>"` regenerates the
deterministic synthetic weights instead (no checkpoints exist in this environment).
"""
import os
import sys

import numpy as np

from . import synth


def _flatten_container(state, prefix):
    """reference _serialize() output -> {prefix + key: ndarray}; FIR / upsample buffers are constants, dropped."""
    out = {}
    for k, v in state.get("state_dict", {}).items():
        if k.endswith("filter_kernel") or k.endswith("upsample.weight") or k == "dlatent_avg":
            continue
        out[prefix + k] = np.asarray(v.float().cpu().numpy() if hasattr(v, "float") else v, dtype=np.float32)
    for sub in ("G_mapping", "G_synthesis"):
        if sub in state:
            out.update(_flatten_container(state[sub], sub + "."))
    return out


def _channels_from_state(sd):
    """Recover the channel list (reference G order: last -> first) from G_synthesis weight shapes."""
    ch, b = [], 0
    while "G_synthesis.conv_blocks.%d.conv_block.0.layer.layer.weight" % b in sd:
        ch.append(sd["G_synthesis.conv_blocks.%d.conv_block.0.layer.layer.weight" % b].shape[0])
        b += 1
    return ch[::-1]


class StyleGAN2:
    """models.py:90-129.  Holds host tensors; the device copy lives in the engine."""

    def __init__(self, config):
        self.config = config
        w = str(config.weights)
        if w.startswith("synthetic"):
            seed = int(w.split(":")[1]) if ":" in w else 0
            channels = list(getattr(config, "channels", synth.FFHQ_CHANNELS))
            self.state = synth.make_state(synth.stylegan2_g_spec(channels, config.dim_z, getattr(config, "mapping_layers", 8)), seed)
            self.state.update(synth.make_state(synth.stylegan2_d_spec(channels), seed))
        else:
            if not os.path.exists(os.path.join(w, "G.pth")):
                print("Weights not found!\nRun : ./download-weights.sh StyleGAN2-<model>")   # models.py:93-101
                sys.exit(1)
            import torch
            g = torch.load(os.path.join(w, "G.pth"), map_location="cpu", weights_only=False)
            d = torch.load(os.path.join(w, "D.pth"), map_location="cpu", weights_only=False)
            self.state = _flatten_container(g, "")
            self.state.update(_flatten_container(d, "D."))
            channels = _channels_from_state(self.state)
        self.channels = channels            # reference G order (last -> first)

    def has_discriminator(self):
        return True


class DeepMindBigGAN:
    """models.py:64-86 — BigGAN-deep as released in pytorch-pretrained-biggan (`BigGAN.from_pretrained(config.weights)`).

    `config.weights`: "synthetic[:seed]" (deterministic synthetic weights of the geometry in `config.biggan_geometry`,
    default = the released biggan-deep-<res> layer table), or a directory / file holding the package's
    `pytorch_model.bin` state dict (+ optional `config.json`), i.e. what `from_pretrained` caches.  Tensors go to the
    engine under "biggan." + the package's own key (spectral-norm weight_orig / weight_u / weight_v included)."""

    def __init__(self, config):
        self.config = config
        w = str(config.weights)
        geo = dict(getattr(config, "biggan_geometry", {}) or {})
        res = int(geo.get("output_dim", 512 if "512" in getattr(config, "config", w) + w else 256 if "256" in getattr(config, "config", w) + w else 128))
        if w.startswith("synthetic"):
            seed = int(w.split(":")[1]) if ":" in w else 0
            self.geometry = dict(layers=geo.get("layers", synth.BIGGAN_LAYERS[res]), attention_pos=geo.get("attention_pos", 8),
                                 ch=geo.get("ch", 128), z_dim=config.dim_z, num_classes=config.num_classes,
                                 n_stats=geo.get("n_stats", 51), eps=geo.get("eps", 1e-4))
            g = self.geometry
            self.state = synth.make_biggan_state(synth.biggan_spec(g["layers"], g["attention_pos"], g["ch"], g["z_dim"],
                                                                    g["num_classes"], g["n_stats"]), seed)
        else:
            path = os.path.join(w, "pytorch_model.bin") if os.path.isdir(w) else w
            if not os.path.exists(path):
                print("Weights not found!\nExpected the pytorch-pretrained-biggan checkpoint at %s" % path)
                sys.exit(1)
            import json
            import torch
            sd = torch.load(path, map_location="cpu")
            self.state = {"biggan." + k: v.float().numpy() for k, v in sd.items()}
            cj = os.path.join(os.path.dirname(path), "config.json")
            pc = json.load(open(cj)) if os.path.exists(cj) else {}
            layers = [(int(bool(u)), int(a), int(b)) for u, a, b in pc.get("layers", synth.BIGGAN_LAYERS[res])]
            self.geometry = dict(layers=layers, attention_pos=pc.get("attention_layer_position", 8),
                                 ch=pc.get("channel_width", 128), z_dim=pc.get("z_dim", config.dim_z),
                                 num_classes=pc.get("num_classes", config.num_classes), n_stats=pc.get("n_stats", 51),
                                 eps=pc.get("eps", 1e-4))
        self.geometry["truncation"] = float(getattr(config, "truncation", 1.0))     # models.py:78,84
        self.D = None

    def has_discriminator(self):
        return False


def remap_gpt2_state(sd):
    """HF `gpt2-pytorch_model.bin` keys -> engine tensors, as gpt2/utils.py:10-51 does: TF-style LayerNorm names
    (.g/.b/.w -> .weight/.bias), the missing `transformer.` prefix; the causal-mask buffers `h.N.attn.bias`
    (NOT `h.N.attn.c_attn.bias`, the QKV bias) and the tied `lm_head` are not weights."""
    import re
    state = {}
    for k, v in sd.items():
        for old, new in ((".g", ".weight"), (".b", ".bias"), (".w", ".weight")):                 # gpt2/utils.py:13-26
            if k.endswith(old):
                k = k[:-len(old)] + new
                break
        if k.startswith("lm_head."):                                                             # tied to wte (model.py:175-178)
            continue
        if not k.startswith("transformer."):                                                     # utils.py:46-48
            k = "transformer." + k
        if re.fullmatch(r"transformer\.h\.\d+\.attn\.(bias|masked_bias)", k):
            continue
        state["gpt2." + k] = np.asarray(v.float().numpy() if hasattr(v, "float") else v, dtype=np.float32)
    return state


class GPT2:
    """models.py:13-62 — GPT-2 small as a token-latent text generator (img2txt).  Host part: weights
    (HF `gpt2-pytorch_model.bin` remapped as gpt2/utils.py:10-51 does, or synthetic), BPE, parse_out; the
    decode itself runs on the device (glass_engine_gpt2_decode, fp32)."""

    def __init__(self, config):
        self.config = config
        w = str(config.weights)
        if w.startswith("synthetic"):
            seed = int(w.split(":")[1]) if ":" in w else 0
            geo = getattr(config, "gpt2_geometry", dict(n_embd=768, n_layer=12))
            self.state = synth.make_state(synth.gpt2_spec(geo["n_embd"], geo["n_layer"], config.encoder_size), seed)
        else:
            if not os.path.exists(w):
                print("Weights not found!\nRun: ./download-weights.sh GPT2")                 # models.py:18-20
                sys.exit(1)
            import torch
            self.state = remap_gpt2_state(torch.load(w, map_location="cpu"))
        self.enc = None
        enc_path, vocab_path = getattr(config, "encoder", None), getattr(config, "vocab", None)
        if enc_path and vocab_path and os.path.exists(enc_path) and os.path.exists(vocab_path):
            from .gpt2_bpe import Gpt2Bpe
            self.enc = Gpt2Bpe(enc_path, vocab_path)
            self.init_tokens = np.asarray(self.enc.encode(config.init_text), dtype=np.int64)   # models.py:30
        else:
            self.init_tokens = np.asarray(getattr(config, "init_tokens", [1169, 4286, 286]), dtype=np.int64)  # "the picture of"
        self.engine = None    # set by Generator

    def has_discriminator(self):
        return False

    def decode_tokens(self, z):
        """models.py:45-60: context = z ++ init_tokens, 30 greedy steps."""
        z = np.asarray(z, dtype=np.int64)
        ctx = np.concatenate([z, np.tile(self.init_tokens, (z.shape[0], 1))], axis=1)
        return self.engine.gpt2_decode(ctx, self.config.max_tokens_len)

    def parse_out(self, out):
        """models.py:32-42"""
        eot = self.enc.eot if self.enc is not None else self.config.encoder_size - 1
        texts = []
        for seq in out:
            seq = [int(t) for t in seq]
            if eot in seq:
                text = seq[self.config.dim_z:seq.index(eot)]      # empty when <|endoftext|> sits in the latent part
            else:
                text = seq[self.config.dim_z:]
            texts.append(self.enc.decode(text)[:self.config.max_text_len])
        return texts

    def generate(self, z, minibatch=None):
        if self.enc is None:
            raise RuntimeError("GPT-2 BPE assets not found (config.encoder / config.vocab)")
        return self.parse_out(self.decode_tokens(z))
