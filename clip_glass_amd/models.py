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
    def __init__(self, config):
        raise NotImplementedError("BigGAN-deep (BASELINE config C3) is a later row of SURVEY §8; "
                                  "pytorch-pretrained-biggan's source is absent (parity unpinned)")


class GPT2:
    def __init__(self, config):
        raise NotImplementedError("GPT-2 img2txt (BASELINE config C5) is a later row of SURVEY §8")
