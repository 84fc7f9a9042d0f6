"""ctypes binding of libglass.so (include/glass.h) — the drop-in boundary.

The library is built in-tree by `__graft_entry__.build()` / `make -C clip_glass_amd/csrc`.
There is NO CPU fallback: if the shared library is missing or the HIP device is
absent, construction raises (the reference would `sys.exit(1)` on missing weights,
models.py:18-20,93-101; here errors surface as RuntimeError with glass_last_error()).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglass.so")
MAX_BLOCKS = 12
MAX_BG_LAYERS = 16
GEN_STYLEGAN2, GEN_BIGGAN_DEEP = 0, 1


class GlassConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_blocks", C.c_int32), ("channels", C.c_int32 * MAX_BLOCKS),
                ("latent_size", C.c_int32), ("mapping_layers", C.c_int32), ("batch_size", C.c_int32),
                ("mbstd_group", C.c_int32), ("use_discriminator", C.c_int32), ("n_obj", C.c_int32),
                ("max_pop", C.c_int32), ("chunk", C.c_int32),
                ("clip_width", C.c_int32), ("clip_layers", C.c_int32), ("clip_heads", C.c_int32),
                ("clip_patch", C.c_int32), ("clip_res", C.c_int32), ("clip_embed", C.c_int32),
                ("noise_mode", C.c_int32), ("noise_seed", C.c_uint64),
                ("generator", C.c_int32), ("bg_ch", C.c_int32), ("bg_z_dim", C.c_int32), ("bg_num_classes", C.c_int32),
                ("bg_n_layers", C.c_int32), ("bg_layers", (C.c_int32 * 3) * MAX_BG_LAYERS),
                ("bg_attention_pos", C.c_int32), ("bg_n_stats", C.c_int32), ("bg_eps", C.c_float),
                ("bg_truncation", C.c_float)]


class GlassNoise(C.Structure):
    _fields_ = [("n_minibatches", C.c_int32), ("n_layers", C.c_int32),
                ("planes", C.POINTER(C.POINTER(C.c_float)))]


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


_lib = None


def load_library(path=None):
    """Load libglass.so; raises OSError/RuntimeError loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("GLASS_LIB") or LIB_PATH      # GLASS_LIB: A/B knob (a second build of libglass.so on the same box)
    if not os.path.exists(path):
        raise RuntimeError("libglass.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C clip_glass_amd/csrc` — there is no CPU fallback" % path)
    lib = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    lib.glass_last_error.restype = C.c_char_p
    lib.glass_version.restype = C.c_char_p
    lib.glass_engine_create.argtypes = [C.POINTER(GlassConfig), C.POINTER(C.c_void_p)]
    lib.glass_engine_destroy.argtypes = [C.c_void_p]
    lib.glass_engine_destroy.restype = None
    lib.glass_engine_load_tensor.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_int32, C.POINTER(C.c_int64)]
    lib.glass_engine_finalize.argtypes = [C.c_void_p]
    lib.glass_engine_set_target.argtypes = [C.c_void_p, fp, C.c_int32]
    lib.glass_engine_encode_text.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, fp]
    lib.glass_engine_encode_image.argtypes = [C.c_void_p, fp, C.c_int32, fp]
    lib.glass_engine_gpt2_decode.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.glass_engine_evaluate.argtypes = [C.c_void_p, fp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GlassNoise), fp]
    lib.glass_engine_generate.argtypes = [C.c_void_p, fp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GlassNoise), fp]
    lib.glass_engine_last_details.argtypes = [C.c_void_p, C.c_int32, fp, fp, fp]
    lib.glass_engine_last_gpu_ms.argtypes = [C.c_void_p, fp]
    if hasattr(lib, "glass_engine_last_F_device"):      # (absent from older A/B builds loaded through GLASS_LIB)
        lib.glass_engine_last_F_device.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.glass_engine_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    lib.glass_engine_set_overlap.argtypes = [C.c_void_p, C.c_int32]
    lib.glass_engine_set_biggan_tap.argtypes = [C.c_void_p, C.c_int32]
    lib.glass_engine_set_profile_filter.argtypes = [C.c_void_p, C.c_char_p]
    lib.glass_engine_get_profile.argtypes = [C.c_void_p, C.POINTER(ProfRow), C.c_int32, C.POINTER(C.c_int32)]
    lib.glass_device_info.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def _check(lib, rc):
    if rc != 0:
        raise RuntimeError("libglass error %d: %s" % (rc, lib.glass_last_error().decode()))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def device_info(device=0):
    lib = load_library()
    name = C.create_string_buffer(256)
    cus = C.c_int32()
    mem = C.c_int64()
    _check(lib, lib.glass_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return dict(name=name.value.decode(), cus=cus.value, hbm_bytes=mem.value)


class Engine:
    """One engine per (process, GPU).  Not thread-safe; evaluate() is blocking."""

    def __init__(self, channels, latent_size=512, mapping_layers=8, batch_size=4, use_discriminator=True,
                 n_obj=2, max_pop=64, chunk=0, clip=(768, 12, 12, 32, 224, 512), noise_mode=1, noise_seed=0,
                 mbstd_group=4, device=0, biggan=None):
        """`biggan` = dict(layers=[(up, in_mult, out_mult), ...], attention_pos, ch, z_dim, num_classes, n_stats, eps,
        truncation) selects the BigGAN-deep generator (channels must then be empty, no discriminator)."""
        self.lib = load_library()
        cfg = GlassConfig()
        cfg.device = device
        if biggan is not None:
            layers = list(biggan["layers"])
            cfg.generator = GEN_BIGGAN_DEEP
            cfg.bg_ch, cfg.bg_z_dim = int(biggan.get("ch", 128)), int(biggan.get("z_dim", 128))
            cfg.bg_num_classes = int(biggan.get("num_classes", 1000))
            cfg.bg_n_layers = len(layers)
            for i, (up, a, b) in enumerate(layers):
                cfg.bg_layers[i][0], cfg.bg_layers[i][1], cfg.bg_layers[i][2] = int(up), int(a), int(b)
            cfg.bg_attention_pos = int(biggan.get("attention_pos", 8))
            cfg.bg_n_stats, cfg.bg_eps = int(biggan.get("n_stats", 51)), float(biggan.get("eps", 1e-4))
            cfg.bg_truncation = float(biggan.get("truncation", 1.0))
            latent_size = cfg.bg_z_dim + cfg.bg_num_classes
            channels, use_discriminator, n_obj, noise_mode = [], False, 1, 0
        cfg.n_blocks = len(channels)
        for i, c in enumerate(channels):  # LOW -> HIGH resolution
            cfg.channels[i] = int(c)
        cfg.latent_size, cfg.mapping_layers, cfg.batch_size = latent_size, mapping_layers, batch_size
        cfg.mbstd_group, cfg.use_discriminator, cfg.n_obj = mbstd_group, int(bool(use_discriminator)), n_obj
        cfg.max_pop, cfg.chunk = max_pop, chunk
        (cfg.clip_width, cfg.clip_layers, cfg.clip_heads, cfg.clip_patch, cfg.clip_res, cfg.clip_embed) = clip
        cfg.noise_mode, cfg.noise_seed = noise_mode, noise_seed
        self.cfg = cfg
        self.channels = list(channels)
        self.res = 4 << (len(channels) - 1) if channels else 0
        if biggan is not None:
            self.res = 4 << sum(1 for l in biggan["layers"] if l[0])
        self.n_noise = 1 + 2 * (len(channels) - 1) if channels else 0
        self._h = C.c_void_p()
        _check(self.lib, self.lib.glass_engine_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.glass_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- weights -------------------------------------------------------------
    def load_tensor(self, name, array):
        a = _f32(array)
        dims = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        _check(self.lib, self.lib.glass_engine_load_tensor(self._h, name.encode(), _fp(a), a.ndim, dims))

    def load_state(self, state):
        for k, v in state.items():
            self.load_tensor(k, np.asarray(v))

    def finalize(self):
        _check(self.lib, self.lib.glass_engine_finalize(self._h))

    def set_target(self, feat):
        f = _f32(feat).reshape(-1)
        _check(self.lib, self.lib.glass_engine_set_target(self._h, _fp(f), f.size))

    def encode_text(self, tokens):
        """CLIP.encode_text (clip/model.py:307-320): tokens int [n, ctx] -> float32 [n, clip_embed]."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty((t.shape[0], self.cfg.clip_embed), dtype=np.float32)
        _check(self.lib, self.lib.glass_engine_encode_text(self._h, t.ctypes.data_as(C.POINTER(C.c_int32)), t.shape[0],
                                                            t.shape[1], _fp(out)))
        return out

    def encode_image(self, images):
        """CLIP.encode_image on preprocessed images [n,3,R,R] float32 -> [n, clip_embed] (generator.py:26-27)."""
        a = _f32(images)
        out = np.empty((a.shape[0], self.cfg.clip_embed), dtype=np.float32)
        _check(self.lib, self.lib.glass_engine_encode_image(self._h, _fp(a), a.shape[0], _fp(out)))
        return out

    def gpt2_decode(self, context, length):
        """gpt2/sample.py:21-36 with sample=False: int tokens [P, n] -> [P, n + length] (greedy, fp32)."""
        c = np.ascontiguousarray(context, dtype=np.int32)
        out = np.empty((c.shape[0], c.shape[1] + length), dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        _check(self.lib, self.lib.glass_engine_gpt2_decode(self._h, c.ctypes.data_as(ip), c.shape[0], c.shape[1], length,
                                                            out.ctypes.data_as(ip)))
        return out

    # --- the pass ------------------------------------------------------------
    def _noise_arg(self, noise):
        """noise: list (per minibatch) of lists (per layer) of [res,res] float32 planes."""
        if noise is None:
            return None, None
        flat = [_f32(p) for mb in noise for p in mb]
        arr = (C.POINTER(C.c_float) * len(flat))(*[_fp(p) for p in flat])
        gn = GlassNoise(len(noise), len(noise[0]), arr)
        return gn, (flat, arr)

    def evaluate(self, x, generation=0, first_minibatch=0, noise=None):
        """problem.py:14-29 — returns F float32 [P, n_obj]."""
        z = _f32(x)
        P = z.shape[0]
        out = np.empty((P, self.cfg.n_obj), dtype=np.float32)
        gn, keep = self._noise_arg(noise)
        _check(self.lib, self.lib.glass_engine_evaluate(self._h, _fp(z), P, generation, first_minibatch,
                                                         C.byref(gn) if gn is not None else None, _fp(out)))
        del keep
        return out

    def generate(self, x, generation=0, first_minibatch=0, noise=None):
        """generator.py:29-34 — images float32 [P,3,R,R] in [0,1]."""
        z = _f32(x)
        P = z.shape[0]
        out = np.empty((P, 3, self.res, self.res), dtype=np.float32)
        gn, keep = self._noise_arg(noise)
        _check(self.lib, self.lib.glass_engine_generate(self._h, _fp(z), P, generation, first_minibatch,
                                                         C.byref(gn) if gn is not None else None, _fp(out)))
        del keep
        return out

    def details(self, P):
        feat = np.empty((P, self.cfg.clip_embed), dtype=np.float32)
        dis = np.empty((P,), dtype=np.float32)
        sim = np.empty((P,), dtype=np.float32)
        _check(self.lib, self.lib.glass_engine_last_details(self._h, P, _fp(feat), _fp(dis), _fp(sim)))
        return dict(features=feat, dis=dis, sim=sim)

    def last_F_device(self, P):
        """The last evaluate()'s fitness rows as a torch CUDA tensor VIEW [P, n_obj] of the engine's own buffer (no copy; valid until the
        engine's next call) — what the RCCL all-gather of a generation sends (parallel.py)."""
        import torch
        ptr = C.c_void_p()
        _check(self.lib, self.lib.glass_engine_last_F_device(self._h, P, C.byref(ptr)))

        class _View:      # CUDA array interface (v2): torch.as_tensor wraps device memory it does not own
            __cuda_array_interface__ = dict(shape=(P, int(self.cfg.n_obj)), typestr="<f4", data=(int(ptr.value), False), version=2)
        return torch.as_tensor(_View(), device=torch.device("cuda", int(self.cfg.device)))

    def last_gpu_ms(self):
        ms = C.c_float()
        _check(self.lib, self.lib.glass_engine_last_gpu_ms(self._h, C.byref(ms)))
        return ms.value

    def set_profiling(self, on):
        _check(self.lib, self.lib.glass_engine_set_profiling(self._h, int(on)))

    def biggan_tap(self, block):
        """Record the activation after GenBlock `block` (-1: after self-attention) on the next pass; see biggan_tap_result."""
        _check(self.lib, self.lib.glass_engine_set_biggan_tap(self._h, int(block)))

    def biggan_tap_result(self):
        dims = (C.c_int32 * 4)()
        self.lib.glass_engine_get_biggan_tap.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int32)]
        _check(self.lib, self.lib.glass_engine_get_biggan_tap(self._h, None, 0, dims))
        out = np.empty(tuple(int(d) for d in dims), dtype=np.float32)
        _check(self.lib, self.lib.glass_engine_get_biggan_tap(self._h, _fp(out), out.size, dims))
        return out

    def set_overlap(self, on):
        _check(self.lib, self.lib.glass_engine_set_overlap(self._h, int(on)))

    def set_profile_filter(self, kernel_substr):
        _check(self.lib, self.lib.glass_engine_set_profile_filter(self._h, (kernel_substr or "").encode()))

    def profile(self):
        n = C.c_int32()
        _check(self.lib, self.lib.glass_engine_get_profile(self._h, None, 0, C.byref(n)))
        rows = (ProfRow * max(n.value, 1))()
        _check(self.lib, self.lib.glass_engine_get_profile(self._h, rows, n.value, C.byref(n)))
        return [dict(name=r.name.decode(), launches=r.launches, total_ms=r.total_ms, flops=r.flops, bytes=r.bytes)
                for r in rows[:n.value]]
