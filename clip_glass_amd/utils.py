"""Host-side helpers mirroring /root/reference/utils.py (numpy; the device applies the
same maps inside its kernels: csrc/kernels_misc.hip finalize_image / fromrgb / resize)."""
import numpy as np


def biggan_norm(images):
    """utils.py:14-17"""
    return np.clip((images + 1) / 2.0, 0, 1)


def biggan_denorm(images):
    """utils.py:19-21"""
    return images * 2 - 1


def save_grid(images, path, nrow=8, padding=2):
    """utils.py:5-7 (torchvision.utils.make_grid + save_image) with PIL only."""
    from PIL import Image
    images = np.asarray(images)
    n, c, h, w = images.shape
    ncol = min(nrow, n)
    nr = (n + ncol - 1) // ncol
    grid = np.zeros((c, nr * (h + padding) + padding, ncol * (w + padding) + padding), np.float32)
    for i in range(n):
        r, q = divmod(i, ncol)
        y0, x0 = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = images[i]
    save_image(grid, path)


def save_image(image, path):
    """torchvision.utils.save_image semantics: [0,1] float -> uint8 (x*255+0.5, clamp)."""
    from PIL import Image
    a = np.clip(np.asarray(image) * 255 + 0.5, 0, 255).astype(np.uint8).transpose(1, 2, 0)
    Image.fromarray(a).save(path)
