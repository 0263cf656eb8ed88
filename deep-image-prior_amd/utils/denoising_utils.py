"""Denoising input synthesis, API-compatible with the reference's utils/denoising_utils.py:6-16."""
import numpy as np

from .common_utils import *  # noqa: F401,F403  (the reference re-exports common_utils the same way)


def get_noisy_image(img_np, sigma):
    """Adds N(0, sigma^2) noise (numpy global RNG) and clips to [0,1].  Returns (PIL, ndarray)."""
    img_noisy_np = np.clip(img_np + np.random.normal(scale=sigma, size=img_np.shape), 0, 1).astype(np.float32)
    return np_to_pil(img_noisy_np), img_noisy_np
