"""Evaluation façade — mirror of /root/reference/generator.py on top of the HIP engine.

The reference loads CLIP + the GAN, pre-computes the text feature once (generator.py:16-27)
and exposes generate / clip_similarity / discriminate / save.  Here the GAN, the resize and
CLIP live on the device behind ONE call (`evaluate`), so `_evaluate` makes a single trip;
`generate` / `save` serve run.py's callbacks (run.py:45-51,118-125).
"""
import os

import numpy as np

from . import synth
from .engine import Engine
from .utils import save_grid, save_image

CLIP_VIT_B32 = (768, 12, 12, 32, 224, 512)


def clip_geometry_from_state(state):
    """(width, layers, heads, patch, input_res, embed) of the VISUAL tower from a CLIP state dict keyed as
    clip/model.py:363-399 reads it (only `clip.visual.*` blocks count: the text tower has resblocks too)."""
    width = state["clip.visual.conv1.weight"].shape[0]
    patch = state["clip.visual.conv1.weight"].shape[-1]
    grid = round((state["clip.visual.positional_embedding"].shape[0] - 1) ** 0.5)
    layers = len([k for k in state if k.startswith("clip.visual.") and k.endswith(".attn.in_proj_weight")])
    return (width, layers, width // 64, patch, patch * grid, state["clip.visual.proj"].shape[1])


def clip_state_from_checkpoint(sd, with_text):
    """reference CLIP state dict (clip.load(...).state_dict(), clip/clip.py:64-78; keys as build_model reads them,
    clip/model.py:363-399) -> engine tensors under "clip." + key.  The three geometry scalars of the jit archive and
    logit_scale are not weights of either tower."""
    state = {}
    for k, v in sd.items():
        if k in ("input_resolution", "context_length", "vocab_size", "logit_scale"):
            continue
        if k.startswith("visual.") or with_text:
            state["clip." + k] = np.asarray(v.float().numpy() if hasattr(v, "float") else v, dtype=np.float32)
    return state


def _load_clip_state(config, with_text):
    w = getattr(config, "clip_weights", None)
    if w is None:
        # the reference always loads the pretrained ViT-B/32 (clip/clip.py:29-33); a silent random-weight CLIP would run a
        # whole search against meaningless fitness values
        raise RuntimeError("config.clip_weights is not set: pass the CLIP ViT-B/32 checkpoint (--clip-weights PATH), or "
                           "'synthetic:<seed>' explicitly for tests / benchmarks")
    w = str(w)
    if w.startswith("synthetic"):
        seed = int(w.split(":")[1]) if ":" in w else 0
        geom = tuple(getattr(config, "clip_geometry", CLIP_VIT_B32))
        state = synth.make_state(synth.clip_visual_spec(geom[0], geom[1], geom[3], geom[4], geom[5]), seed)
        if with_text:
            tg = getattr(config, "clip_text_geometry", dict(width=512, layers=12))
            state.update(synth.make_state(synth.clip_text_spec(width=tg["width"], layers=tg["layers"],
                                                               vocab=tg.get("vocab", 49408), out_dim=geom[5]), seed))
        return state, geom
    import torch   # reference: clip.load -> torch.jit.load(archive).state_dict() (clip/clip.py:64-78)
    try:
        sd = torch.jit.load(w, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(w, map_location="cpu")
    state = clip_state_from_checkpoint(sd, with_text)
    return state, clip_geometry_from_state(state)


def clip_preprocess(path_or_image, n_px=224):
    """clip/clip.py:68-74: Resize(n_px, BICUBIC) -> CenterCrop -> RGB -> ToTensor -> Normalize.
    torchvision is absent here; restated with PIL (third-party, unpinned)."""
    from PIL import Image
    img = Image.open(path_or_image) if isinstance(path_or_image, str) else path_or_image
    w, h = img.size
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nw, nh = int(n_px * w / h), n_px
    img = img.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
    img = img.crop((left, top, left + n_px, top + n_px)).convert("RGB")
    a = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
    mean = np.array([0.48145466, 0.4578275, 0.40821073], np.float32)[:, None, None]
    std = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)[:, None, None]
    return (a - mean) / std


class Generator:
    def __init__(self, config, dist=None):
        self.config = config
        self.augmentation = None
        self.model = config.model(config)                                   # generator.py:19
        # dist = None: pick up the process group when one is initialised (a `torchrun` launch: one process per GPU, backend
        # "nccl" = RCCL); every rank then scores its contiguous shard of the SAME population and one all-gather returns all rows
        if dist is None:
            try:
                import torch.distributed as td
                if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
                    dist = td
            except ImportError:
                pass
        sharded = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
        device = getattr(config, "device", 0)
        if sharded and getattr(config, "device_per_rank", None) is None:
            # one process per GPU: every rank sees the same command line ("cuda", "cuda:0", 0 ...), so the device is this rank's
            # LOCAL_RANK unless the caller names a device per rank explicitly (config.device_per_rank = [ids], indexed by rank)
            device = int(os.environ.get("LOCAL_RANK", dist.get_rank()))
        elif sharded:
            device = int(config.device_per_rank[dist.get_rank()])
        elif ":" in str(device):
            device = int(str(device).split(":")[1])
        elif not isinstance(device, int):       # bare "cuda"
            device = 0
        pop = int(getattr(config, "max_pop", max(config.pop_size, config.batch_size)))
        self.generation = 0
        self.sharder = None
        if config.task == "img2txt":                                        # generator.py:25-27, 52-59
            clip_state, geom = _load_clip_state(config, True)
            self.engine = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1,
                                 max_pop=pop, clip=geom, noise_mode=0, device=device)
            self.engine.load_state(self.model.state)
            self.engine.load_state(clip_state)
            self.engine.finalize()
            self.model.engine = self.engine
            if getattr(config, "target_features", None) is not None:
                self.image_features = np.asarray(config.target_features, np.float32).reshape(1, -1)
            else:
                self.image_features = self.engine.encode_image(clip_preprocess(config.target, geom[4])[None])
            from .tokenizer import DEFAULT_BPE, ClipTokenizer
            self.tokenizer = ClipTokenizer(getattr(config, "bpe_path", DEFAULT_BPE))
            return
        need_text = getattr(config, "target_features", None) is None
        clip_state, geom = _load_clip_state(config, need_text)
        pop = (pop + config.batch_size - 1) // config.batch_size * config.batch_size
        if hasattr(self.model, "geometry"):     # BigGAN-deep (models.py:64-86)
            self.engine = Engine([], batch_size=config.batch_size, max_pop=pop, chunk=getattr(config, "chunk", 0),
                                 clip=geom, device=device, biggan=self.model.geometry)
        else:
            self.engine = Engine(self.model.channels[::-1], latent_size=config.dim_z,
                                 mapping_layers=getattr(config, "mapping_layers", 8), batch_size=config.batch_size,
                                 use_discriminator=bool(config.use_discriminator and config.problem_args["n_obj"] == 2),
                                 n_obj=config.problem_args["n_obj"], max_pop=pop, chunk=getattr(config, "chunk", 0),
                                 clip=geom, noise_mode=getattr(config, "noise_mode", 1),
                                 noise_seed=getattr(config, "noise_seed", 0), device=device)
        self.engine.load_state(self.model.state)
        self.engine.load_state(clip_state)
        self.engine.finalize()
        if getattr(config, "target_features", None) is not None:            # pre-computed text feature
            self.text_features = np.asarray(config.target_features, np.float32).reshape(1, -1)
        else:                                                               # generator.py:23-24
            from .tokenizer import DEFAULT_BPE, ClipTokenizer
            tok = ClipTokenizer(getattr(config, "bpe_path", DEFAULT_BPE))
            self.tokens = tok.tokenize([self.config.target])
            self.text_features = self.engine.encode_text(self.tokens)
        self.engine.set_target(self.text_features[0])
        if sharded:
            from .parallel import ShardedEvaluator
            self.sharder = ShardedEvaluator(self.engine, dist, dist.get_rank(), dist.get_world_size(), config.batch_size,
                                            device=device)

    def clip_similarity_texts(self, texts):
        """generator.py:52-59 (img2txt branch): tokenize -> encode_text -> cosine vs the target image feature;
        a tokenisation failure zeroes the WHOLE population, as the reference's bare except does."""
        try:
            tokens = self.tokenizer.tokenize(texts)
        except Exception:
            return np.zeros(len(texts), np.float32)
        tf = self.engine.encode_text(tokens).astype(np.float64)
        im = self.image_features.astype(np.float64)
        den = np.maximum(np.linalg.norm(tf, axis=1) * np.linalg.norm(im, axis=1), 1e-8)
        return ((tf @ im.T)[:, 0] / den).astype(np.float32)

    # --- the hot path: generate + clip_similarity + discriminate in ONE device pass -------------
    def evaluate(self, ls, noise=None, first_minibatch=0):
        if self.config.task == "img2txt":           # problem.py:19-20,27 with the GPT2 config
            texts = self.model.generate(*ls())
            self.last_texts = texts
            return -self.clip_similarity_texts(texts)[:, None]
        z = ls.population()
        if self.sharder is not None:        # one process per GPU: this rank scores its shard, ONE all-gather of the rows
            if noise is not None or first_minibatch:
                # caller-provided planes / offsets address ONE engine's minibatches; the sharded path derives both from the
                # global minibatch index (device noise), so accepting them here would silently ignore them
                raise ValueError("noise / first_minibatch cannot be combined with a sharded (multi-GPU) Generator")
            F = self.sharder.evaluate_global(z, generation=self.generation)
        else:
            F = self.engine.evaluate(z, generation=self.generation, first_minibatch=first_minibatch, noise=noise)
        self.generation += 1
        return F

    def generate(self, ls, minibatch=None, noise=None):
        """generator.py:29-34 — images [P,3,R,R] float32 after config.norm (biggan_norm); texts for img2txt."""
        if self.config.task == "img2txt":
            return self.model.generate(*ls())
        z = ls.population()
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
        if self.config.task == "img2txt":
            with open(path, "w") as f:
                f.write("\n".join(input))
            return
        if input.shape[0] > 1:
            save_grid(input, path)
        else:
            save_image(input[0], path)
