"""Evaluation façade — mirror of /root/reference/generator.py on top of the HIP engine.

The reference loads CLIP + the GAN, pre-computes the text feature once (generator.py:16-27)
and exposes generate / clip_similarity / discriminate / save.  Here the GAN, the resize and
CLIP live on the device behind ONE call (`evaluate`), so `_evaluate` makes a single trip;
`generate` / `save` serve run.py's callbacks (run.py:45-51,118-125).
"""
import numpy as np

from . import synth
from .engine import Engine
from .utils import save_grid, save_image

CLIP_VIT_B32 = (768, 12, 12, 32, 224, 512)


def _load_clip_state(config, with_text):
    w = str(getattr(config, "clip_weights", "synthetic:0"))
    if w.startswith("synthetic"):
        seed = int(w.split(":")[1]) if ":" in w else 0
        geom = tuple(getattr(config, "clip_geometry", CLIP_VIT_B32))
        state = synth.make_state(synth.clip_visual_spec(geom[0], geom[1], geom[3], geom[4], geom[5]), seed)
        if with_text:
            tg = getattr(config, "clip_text_geometry", dict(width=512, layers=12))
            state.update(synth.make_state(synth.clip_text_spec(width=tg["width"], layers=tg["layers"], out_dim=geom[5]), seed))
        return state, geom
    import torch   # reference: clip.load -> torch.jit.load(archive).state_dict() (clip/clip.py:64-78)
    try:
        sd = torch.jit.load(w, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(w, map_location="cpu")
    state = {"clip." + k: v.float().numpy() for k, v in sd.items()
             if k.startswith("visual.") or (with_text and not k.startswith("visual.") and v.dim() > 0
                                            and k not in ("input_resolution", "context_length", "vocab_size"))}
    width = state["clip.visual.conv1.weight"].shape[0]
    patch = state["clip.visual.conv1.weight"].shape[-1]
    grid = round((state["clip.visual.positional_embedding"].shape[0] - 1) ** 0.5)
    layers = len([k for k in state if k.endswith(".attn.in_proj_weight")])
    return state, (width, layers, width // 64, patch, patch * grid, state["clip.visual.proj"].shape[1])


class Generator:
    def __init__(self, config, dist=None):
        self.config = config
        self.augmentation = None
        if config.task != "txt2img":
            raise NotImplementedError("img2txt (GPT2) is a later row of SURVEY §8")
        self.model = config.model(config)                                   # generator.py:19
        need_text = getattr(config, "target_features", None) is None
        clip_state, geom = _load_clip_state(config, need_text)
        pop = int(getattr(config, "max_pop", max(config.pop_size, config.batch_size)))
        pop = (pop + config.batch_size - 1) // config.batch_size * config.batch_size
        device = getattr(config, "device", 0)
        device = int(str(device).split(":")[1]) if ":" in str(device) else (device if isinstance(device, int) else 0)
        self.engine = Engine(self.model.channels[::-1], latent_size=config.dim_z,
                             mapping_layers=getattr(config, "mapping_layers", 8), batch_size=config.batch_size,
                             use_discriminator=bool(config.use_discriminator and config.problem_args["n_obj"] == 2),
                             n_obj=config.problem_args["n_obj"], max_pop=pop, chunk=getattr(config, "chunk", 0),
                             clip=geom, noise_mode=getattr(config, "noise_mode", 1),
                             noise_seed=getattr(config, "noise_seed", 0), device=device)
        self.engine.load_state(self.model.state)
        self.engine.load_state(clip_state)
        self.engine.finalize()
        self.generation = 0
        if getattr(config, "target_features", None) is not None:            # pre-computed text feature
            self.text_features = np.asarray(config.target_features, np.float32).reshape(1, -1)
        else:                                                               # generator.py:23-24
            from .tokenizer import DEFAULT_BPE, ClipTokenizer
            tok = ClipTokenizer(getattr(config, "bpe_path", DEFAULT_BPE))
            self.tokens = tok.tokenize([self.config.target])
            self.text_features = self.engine.encode_text(self.tokens)
        self.engine.set_target(self.text_features[0])

    # --- the hot path: generate + clip_similarity + discriminate in ONE device pass -------------
    def evaluate(self, ls, noise=None, first_minibatch=0):
        (z,) = ls()
        F = self.engine.evaluate(z, generation=self.generation, first_minibatch=first_minibatch, noise=noise)
        self.generation += 1
        return F

    def generate(self, ls, minibatch=None, noise=None):
        """generator.py:29-34 — images [P,3,R,R] float32 after config.norm (biggan_norm)."""
        (z,) = ls()
        bs = self.config.batch_size
        P = z.shape[0]
        if minibatch is None:                       # run.py:118: whole input as ONE G call
            pad = (-P) % bs
        else:
            assert z.shape[0] % minibatch == 0      # models.py:112
            pad = 0
        if pad:
            z = np.concatenate([z, np.repeat(z[-1:], pad, axis=0)])
        img = self.engine.generate(z, generation=self.generation, noise=noise)[:P]
        return img

    def has_discriminator(self):
        return self.model.has_discriminator()

    def save(self, input, path):
        """generator.py:63-72"""
        if input.shape[0] > 1:
            save_grid(input, path)
        else:
            save_image(input[0], path)
