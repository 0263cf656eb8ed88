"""Super-resolution helpers with the reference's names and behaviour (utils/sr_utils.py:3-94), so
that `from utils.sr_utils import *` in super-resolution.ipynb / sr_prior_effect.ipynb resolves to
this backend.  Host-side image preparation plus the TV regulariser; the network, the Lanczos
`Downsampler` and the optimisation loop they feed are the MI355X path."""
import numpy as np
import torch

from .common_utils import *  # noqa: F401,F403


def put_in_center(img_np, target_size):
    """Zero canvas 3 x target_size with `img_np` (C x H x W) pasted centred (:3-16)."""
    th, tw = target_size
    h, w = img_np.shape[1], img_np.shape[2]
    top, left = int((th - h) / 2), int((tw - w) / 2)
    bottom, right = int((th + h) / 2), int((tw + w) / 2)
    canvas = np.zeros([3, th, tw])
    canvas[:, top:bottom, left:right] = img_np
    return canvas


def load_LR_HR_imgs_sr(fname, imsize, factor, enforse_div32=None):
    """Loads `fname`, optionally resizes to `imsize`, optionally centre-crops to multiples of 32
    (enforse_div32 == 'CROP'), and produces the `factor`-times smaller LR image with PIL's
    antialiasing (Lanczos) filter (:18-66).  Returns the reference's dict of PIL / numpy pairs."""
    from PIL import Image
    orig_pil, orig_np = get_image(fname, -1)
    if imsize != -1:
        orig_pil, orig_np = get_image(fname, imsize)
    if enforse_div32 == 'CROP':
        w, h = orig_pil.size
        nw, nh = w - w % 32, h - h % 32
        hr_pil = orig_pil.crop([(w - nw) / 2, (h - nh) / 2, (w + nw) / 2, (h + nh) / 2])
        hr_np = pil_to_np(hr_pil)
    else:
        hr_pil, hr_np = orig_pil, orig_np
    lr_size = [hr_pil.size[0] // factor, hr_pil.size[1] // factor]
    lr_pil = hr_pil.resize(lr_size, getattr(Image, 'ANTIALIAS', Image.LANCZOS))   # Pillow >= 10 dropped ANTIALIAS
    lr_np = pil_to_np(lr_pil)
    print('HR and LR resolutions: %s, %s' % (str(hr_pil.size), str(lr_pil.size)))
    return {'orig_pil': orig_pil, 'orig_np': orig_np, 'LR_pil': lr_pil, 'LR_np': lr_np, 'HR_pil': hr_pil,
            'HR_np': hr_np}


def get_baselines(img_LR_pil, img_HR_pil):
    """Bicubic, unsharp-masked bicubic and nearest-neighbour up-sampling of the LR image (:69-80)."""
    from PIL import Image, ImageFilter
    bicubic_pil = img_LR_pil.resize(img_HR_pil.size, Image.BICUBIC)
    nearest_pil = img_LR_pil.resize(img_HR_pil.size, Image.NEAREST)
    sharp_pil = bicubic_pil.filter(ImageFilter.UnsharpMask())
    return pil_to_np(bicubic_pil), pil_to_np(sharp_pil), pil_to_np(nearest_pil)


def tv_loss(x, beta=0.5):
    """Total-variation regulariser sum((dh^2 + dw^2)^beta) over the interior (:84-94); `x` is the
    net output (an autograd tensor on the GPU), so this composes with the HIP backward unchanged."""
    dh = (x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2
    dw = (x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2
    return torch.sum((dh[:, :, :-1] + dw[:, :, :, :-1]) ** beta)
