"""Synthetic SIFT-like / SURF-like descriptors for tests and benchmarks.

There is no image data (and no OpenCV) where this runs, so the descriptors the reference would get
from cv2's SURF extractor (model/features.py:905-1012) are replaced by random rows with the same
statistics that matter to a nearest-neighbour search: clustered, non-negative, clipped and
re-normalised the way SIFT descriptors are, norm 512, with a little real-valued jitter so that
exact distance ties have probability zero (a rounded variant exists to produce ties on purpose).
"""
import numpy as np


def sift_like(n, d=128, n_modes=256, seed=0, rounded=False, dtype=np.float32):
    rng = np.random.default_rng(seed)
    centres = np.abs(rng.standard_normal((n_modes, d)))
    out = np.empty((n, d), dtype=dtype)
    step = 1 << 16
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        m = rng.integers(0, n_modes, size=hi - lo)
        x = np.abs(centres[m] + 0.35 * rng.standard_normal((hi - lo, d)))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        np.minimum(x, 0.2, out=x)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        x *= 512.0
        if rounded:
            x = np.rint(x)
        else:
            x += rng.uniform(-0.5, 0.5, size=x.shape)
        out[lo:hi] = x
    return out


def surf_like(n, d=128, n_modes=256, seed=0, dtype=np.float32):
    """Signed, unit-norm rows (cv2 SURF-128 convention)."""
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((n_modes, d))
    m = rng.integers(0, n_modes, size=n)
    x = centres[m] + 0.5 * rng.standard_normal((n, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(dtype)


def sift_like_device(n, d=128, n_modes=256, seed=0, device="cuda"):
    """Same recipe generated on the device with torch's generator (a different random stream
    than `sift_like`; use it where only the distribution matters, e.g. large benchmark inputs)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    centres = torch.randn((n_modes, d), generator=g, device=device).abs_()
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 1 << 20
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        m = torch.randint(0, n_modes, (hi - lo,), generator=g, device=device)
        x = (centres[m] + 0.35 * torch.randn((hi - lo, d), generator=g, device=device)).abs_()
        x /= x.norm(dim=1, keepdim=True)
        x.clamp_(max=0.2)
        x /= x.norm(dim=1, keepdim=True)
        x *= 512.0
        x += torch.rand((hi - lo, d), generator=g, device=device) - 0.5
        out[lo:hi] = x
    return out


def image_offsets(n_images, per_image, ragged=False, seed=0):
    """int64 offsets [n_images + 1]; ragged: Poisson sizes with two empty images."""
    if not ragged:
        return np.arange(n_images + 1, dtype=np.int64) * int(per_image)
    rng = np.random.default_rng(seed)
    sizes = rng.poisson(per_image, size=n_images).astype(np.int64)
    if n_images >= 4:
        sizes[1] = 0
        sizes[n_images - 2] = 0
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
