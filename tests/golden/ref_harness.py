"""Import harness for the *reference* implementation (build container only).

TEST INFRASTRUCTURE.  This file never runs on the GPU box: /root/reference does
not exist there.  It is used (a) to pin the oracle (oracle/ vs the imported
reference) and (b) by make_golden.py to generate the committed fixtures.

It registers stub modules for the third-party packages the reference imports but
this image lacks (torchvision, kornia, ftfy, pytorch_pretrained_biggan, pymoo) and
builds a synthetic `stylegan2` package object exposing only models/modules (the
real stylegan2/__init__.py drags in the trainer -> tensorboard).  Nothing from the
reference is copied; its modules are imported from where they lie.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("GLASS_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference


def available():
    return os.path.isdir(os.path.join(REF, "stylegan2"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(modname, path, package=None):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    if package is not None:
        mod.__package__ = package
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_loaded = {}


def load_reference():
    """Returns a dict of the imported reference modules."""
    if _loaded:
        return _loaded
    import torch
    import torch.nn.functional as F

    # --- third-party stubs ------------------------------------------------
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms",
                          Compose=lambda x: x, Resize=lambda *a, **k: None,
                          CenterCrop=lambda *a, **k: None, ToTensor=lambda: None,
                          Normalize=lambda *a, **k: None)
    tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None,
                     make_grid=lambda x: x)
    # kornia==0.4.1 resize(input, size) == bilinear, align_corners=False, no antialias
    _stub("kornia", resize=lambda x, size: F.interpolate(
        x, size=size, mode="bilinear", align_corners=False))
    _stub("ftfy", fix_text=lambda s: s)
    _stub("pytorch_pretrained_biggan", BigGAN=object,
          truncated_noise_sample=lambda *a, **k: None)
    _stub("matplotlib")
    _stub("matplotlib.pyplot")
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    pm = _stub("pymoo")
    pm.model = _stub("pymoo.model")

    class Problem:  # attribute bag standing in for pymoo.model.problem.Problem
        def __init__(self, n_var=-1, n_obj=-1, n_constr=0, xl=None, xu=None, **kw):
            self.n_var, self.n_obj, self.n_constr, self.xl, self.xu = n_var, n_obj, n_constr, xl, xu
    pm.model.problem = _stub("pymoo.model.problem", Problem=Problem)

    # --- stylegan2 package object with only what the run path needs --------
    sg = types.ModuleType("stylegan2")
    sg.__path__ = [os.path.join(REF, "stylegan2")]
    sys.modules["stylegan2"] = sg
    ut = _stub("stylegan2.utils")

    def lerp(a, b, beta):
        if isinstance(beta, (int, float)):
            if beta == 0:
                return b
            if beta == 1:
                return a
        return a + beta * (b - a) if False else b + beta * (a - b)  # unused (truncation off)
    ut.lerp = lerp
    ut.unwrap_module = lambda m: m
    sg.utils = ut
    sg.modules = _load("stylegan2.modules", os.path.join(REF, "stylegan2", "modules.py"), "stylegan2")
    sg.models = _load("stylegan2.models", os.path.join(REF, "stylegan2", "models.py"), "stylegan2")

    # --- clip (tokenizer path is cwd-relative and read at import) ----------
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        if REF not in sys.path:
            sys.path.insert(0, REF)
        import clip.model as clip_model  # noqa
        import clip.clip as clip_clip  # noqa
    finally:
        os.chdir(cwd)

    _loaded.update(dict(sg_models=sg.models, sg_modules=sg.modules,
                        clip_model=clip_model, clip_clip=clip_clip, torch=torch))
    return _loaded


def load_author_modules():
    """problem.py / generator.py / latent.py / models.py / config.py / utils.py."""
    load_reference()
    out = {}
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        # gpt2 pieces imported by models.py
        import importlib
        for name in ["utils", "latent", "models", "config", "generator", "problem"]:
            out[name] = importlib.import_module(name)
    finally:
        os.chdir(cwd)
    return out


# ---------------------------------------------------------------------------
# Builders: instantiate the reference's own nn.Modules and load a synthetic state.
# ---------------------------------------------------------------------------
def _sub(sd, prefix):
    import torch
    return {k[len(prefix):]: torch.as_tensor(v) for k, v in sd.items() if k.startswith(prefix)}


def build_ref_G(sd, channels, latent_size=512, mapping_layers=8):
    R = load_reference()
    m = R["sg_models"]
    Gm = m.GeneratorMapping(latent_size=latent_size, num_layers=mapping_layers)
    Gs = m.GeneratorSynthesis(latent_size=latent_size, channels=list(channels))
    missing = Gm.load_state_dict(_sub(sd, "G_mapping."), strict=True)
    Gs.load_state_dict(_sub(sd, "G_synthesis."), strict=False)  # FIR buffers are constants, not in sd
    G = m.Generator(G_mapping=Gm, G_synthesis=Gs)
    return G.eval()


def build_ref_D(sd, channels):
    R = load_reference()
    D = R["sg_models"].Discriminator(channels=list(channels))
    D.load_state_dict(_sub(sd, "D."), strict=False)
    return D.eval()


def build_ref_clip(sd, fp32=True):
    """clip/model.py:363-399 build_model on a synthetic full state dict; .float() per SURVEY 8(c)."""
    import torch
    R = load_reference()
    st = _sub(sd, "clip.")
    patch = st["visual.conv1.weight"].shape[-1]
    grid = round((st["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    st["input_resolution"] = torch.tensor(patch * grid)
    st["context_length"] = torch.tensor(st["positional_embedding"].shape[0])
    st["vocab_size"] = torch.tensor(st["token_embedding.weight"].shape[0])
    model = R["clip_model"].build_model(st)
    return model.float() if fp32 else model


def build_ref_gpt2(sd, n_embd, n_layer, vocab):
    """The reference's own GPT2LMHeadModel (gpt2/model.py) loaded with a synthetic state."""
    import torch
    load_reference()
    import importlib
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        gm = importlib.import_module("gpt2.model")
        gc = importlib.import_module("gpt2.config")
        gs = importlib.import_module("gpt2.sample")
    finally:
        os.chdir(cwd)
    n_pos = int(sd["gpt2.transformer.wpe.weight"].shape[0])
    cfg = gc.GPT2Config(vocab_size_or_config_json_file=vocab, n_positions=n_pos, n_ctx=n_pos, n_embd=n_embd, n_layer=n_layer,
                        n_head=n_embd // 64)
    model = gm.GPT2LMHeadModel(cfg)
    st = {k[len("gpt2."):]: torch.as_tensor(v) for k, v in sd.items() if k.startswith("gpt2.")}
    missing = model.load_state_dict(st, strict=False)
    assert not [k for k in missing.missing_keys if not k.endswith("attn.bias") and "lm_head" not in k], missing
    model.set_tied()
    return model.eval(), gs.sample_sequence
