"""Inpainting mask helpers with the reference's names and behaviour (utils/inpainting_utils.py:7-22)
so that `from utils.inpainting_utils import *` in inpainting.ipynb / restoration.ipynb resolves to
this backend.  Host side only."""
import numpy as np

from .common_utils import *  # noqa: F401,F403


def get_text_mask(for_image, sz=20):
    """White mask of the image's size with the words "hello world" drawn in black at (128, 128)
    (:7-16).  The reference hard-codes FreeSansBold; fall back to PIL's default font if absent."""
    from PIL import Image, ImageDraw, ImageFont
    try:
        font = ImageFont.truetype('/usr/share/fonts/truetype/freefont/FreeSansBold.ttf', sz)
    except OSError:
        font = ImageFont.load_default()
    mask = Image.fromarray(np.array(for_image) * 0 + 255)
    ImageDraw.Draw(mask).text((128, 128), "hello world", font=font, fill='rgb(0, 0, 0)')
    return mask


def get_bernoulli_mask(for_image, zero_fraction=0.95):
    """Per-element Bernoulli mask (numpy global RNG): 1 with probability 1 - zero_fraction (:18-22)."""
    keep = (np.random.random_sample(size=pil_to_np(for_image).shape) > zero_fraction).astype(int)
    return np_to_pil(keep)
