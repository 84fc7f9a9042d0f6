"""Latent spaces — mirror of /root/reference/latent.py (numpy instead of torch Parameters:
the device copy happens inside glass_engine_evaluate)."""
import numpy as np


class StyleGAN2LatentSpace:
    """latent.py:27-41"""

    def __init__(self, config):
        self.config = config
        self.z = np.random.randn(self.config.batch_size, self.config.dim_z).astype(np.float32)

    def set_values(self, z):
        self.z = np.asarray(z, dtype=np.float32)

    def set_from_population(self, x):
        self.z = np.asarray(x).astype(float).astype(np.float32)     # latent.py:38

    def forward(self):
        return (self.z,)

    __call__ = forward

    def population(self):
        return self.z

    def state_dict(self):       # run.py:101 torch.save(ls.state_dict()) (the reference's nn.Module holds z as a plain tensor: its
        return {"z": self.z}    # state_dict is empty; the latents are kept here so ls_result is usable)


class DeepMindBigGANLatentSpace:
    """latent.py:4-24 — config C3.  `forward()` mirrors the reference (clip z, softmax the class bits); the engine
    takes the raw population rows (`population()`) and applies the same two ops on the device."""

    def __init__(self, config):
        self.config = config
        self.z = np.zeros((config.batch_size, config.dim_z), np.float32)
        self.class_labels = np.zeros((config.batch_size, config.num_classes), np.float32)

    def set_values(self, z, class_labels):
        self.z, self.class_labels = np.asarray(z, np.float32), np.asarray(class_labels, np.float32)

    def set_from_population(self, x):
        x = np.asarray(x)
        self.z = x[:, :self.config.dim_z].astype(float).astype(np.float32)
        self.class_labels = x[:, self.config.dim_z:].astype(float).astype(np.float32)

    def forward(self):
        z = np.clip(self.z, -2, 2)
        e = np.exp(self.class_labels - self.class_labels.max(axis=1, keepdims=True))
        return z, e / e.sum(axis=1, keepdims=True)

    __call__ = forward

    def population(self):
        """Raw rows [z | class bits] float32 — what glass_engine_evaluate takes for the BigGAN generator."""
        return np.concatenate([self.z, self.class_labels], axis=1).astype(np.float32)

    def state_dict(self):                                           # run.py:101
        return {"z": self.z, "class_labels": self.class_labels}


class GPT2LatentSpace:
    """latent.py:44-58 — config C5: rows of token ids (the first dim_z tokens of every GPT-2 context, models.py:45-62)."""

    def __init__(self, config):
        self.config = config
        self.z = np.random.randint(0, config.encoder_size, size=(config.batch_size, config.dim_z)).astype(np.int64)

    def set_values(self, z):
        self.z = np.asarray(z, np.int64)

    def set_from_population(self, x):
        self.z = np.asarray(x).astype(int).astype(np.int64)

    def forward(self):
        return (self.z,)

    __call__ = forward

    def state_dict(self):       # run.py:101 (the reference's nn.Module holds z as a plain tensor: its state_dict is empty;
        return {"z": self.z}    # the tokens are kept here so ls_result is usable)
