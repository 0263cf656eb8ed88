"""Optimisation loop and tensor helpers, API-compatible with the reference's
utils/common_utils.py (optimize :198-232, get_params :29-53, get_noise :118-153,
np_to_torch/torch_to_np :183-195, image helpers :13-27,89-114,155-181).

`optimize('adam', ...)` keeps the reference's zero_grad() -> closure() -> step() contract but
steps ALL parameters with one fused gfx950 launch over the flat parameter arena
(dip_adam_step) instead of torch.optim.Adam's per-tensor kernels.
"""
import numpy as np
import torch

from dip_optim import FusedAdam, GraphedIteration, ArenaLBFGS


# ------------------------------------------------------------------ image helpers (host side)
def crop_image(img, d=32):
    """Centre-crop a PIL image so both sides are divisible by `d`."""
    w, h = img.size[0] - img.size[0] % d, img.size[1] - img.size[1] % d
    box = [int((img.size[0] - w) / 2), int((img.size[1] - h) / 2),
           int((img.size[0] + w) / 2), int((img.size[1] + h) / 2)]
    return img.crop(box)


def load(path):
    from PIL import Image
    return Image.open(path)


def get_image(path, imsize=-1):
    """Load an image, optionally resize to `imsize` (int or (w,h); -1 keeps the size)."""
    from PIL import Image
    img = load(path)
    if isinstance(imsize, int):
        imsize = (imsize, imsize)
    if imsize[0] != -1 and img.size != imsize:
        down = getattr(Image, 'ANTIALIAS', Image.LANCZOS)   # Pillow >= 10 dropped ANTIALIAS
        img = img.resize(imsize, Image.BICUBIC if imsize[0] > img.size[0] else down)
    return img, pil_to_np(img)


def pil_to_np(img_PIL):
    """PIL (W x H x C, 0..255) -> float32 array C x H x W in [0,1]."""
    ar = np.array(img_PIL)
    ar = ar.transpose(2, 0, 1) if ar.ndim == 3 else ar[None, ...]
    return ar.astype(np.float32) / 255.


def np_to_pil(img_np):
    from PIL import Image
    ar = np.clip(img_np * 255, 0, 255).astype(np.uint8)
    ar = ar[0] if img_np.shape[0] == 1 else ar.transpose(1, 2, 0)
    return Image.fromarray(ar)


def np_to_torch(img_np):
    """C x H x W numpy -> 1 x C x H x W tensor (shares memory)."""
    return torch.from_numpy(img_np)[None, :]


def torch_to_np(img_var):
    """1 x C x H x W tensor -> C x H x W numpy (device sync + D2H copy)."""
    return img_var.detach().cpu().numpy()[0]


def get_image_grid(images_np, nrow=8):
    """Tile C x H x W images into one grid image (padding 2, like torchvision.utils.make_grid)."""
    imgs = [np.asarray(x, dtype=np.float32) for x in images_np]
    c, h, w = imgs[0].shape
    ncol = min(nrow, len(imgs))
    nr = (len(imgs) + ncol - 1) // ncol
    pad = 2
    grid = np.zeros((c, nr * (h + pad) + pad, ncol * (w + pad) + pad), dtype=np.float32)
    for k, im in enumerate(imgs):
        r, cc = divmod(k, ncol)
        grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + cc * (w + pad): pad + cc * (w + pad) + w] = im
    return grid


def plot_image_grid(images_np, nrow=8, factor=1, interpolation='lanczos'):
    import matplotlib.pyplot as plt
    n_channels = max(x.shape[0] for x in images_np)
    assert n_channels in (1, 3), "images should have 1 or 3 channels"
    images_np = [x if x.shape[0] == n_channels else np.concatenate([x, x, x], axis=0) for x in images_np]
    grid = get_image_grid(images_np, nrow)
    plt.figure(figsize=(len(images_np) + factor, 12 + factor))
    if images_np[0].shape[0] == 1:
        plt.imshow(grid[0], cmap='gray', interpolation=interpolation)
    else:
        plt.imshow(grid.transpose(1, 2, 0), interpolation=interpolation)
    plt.show()
    return grid


# ------------------------------------------------------------------ hot-path API
def fill_noise(x, noise_type):
    """In-place U(0,1) ('u') or N(0,1) ('n')."""
    if noise_type == 'u':
        x.uniform_()
    elif noise_type == 'n':
        x.normal_()
    else:
        assert False


def get_noise(input_depth, method, spatial_size, noise_type='u', var=1. / 10):
    """1 x input_depth x H x W network input: 'noise' (CPU generator, scaled by `var`) or
    'meshgrid' (2 channels of normalised coordinates, float64 like the reference)."""
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    if method == 'noise':
        net_input = torch.zeros([1, input_depth, spatial_size[0], spatial_size[1]])
        fill_noise(net_input, noise_type)
        net_input *= var
    elif method == 'meshgrid':
        assert input_depth == 2
        X, Y = np.meshgrid(np.arange(0, spatial_size[1]) / float(spatial_size[1] - 1),
                           np.arange(0, spatial_size[0]) / float(spatial_size[0] - 1))
        net_input = np_to_torch(np.concatenate([X[None, :], Y[None, :]]))
    else:
        assert False
    return net_input


def get_params(opt_over, net, net_input, downsampler=None):
    """Parameters to optimise: comma-separated subset of 'net', 'down', 'input'.
    ('down' REPLACES the list, as in the reference.)"""
    params = []
    for opt in opt_over.split(','):
        if opt == 'net':
            params += [x for x in net.parameters()]
        elif opt == 'down':
            assert downsampler is not None
            params = [x for x in downsampler.parameters()]
            for x in params:        # dip-amd Downsampler: fixed taps unless its parameters require grad (models/downsampler.py)
                x.requires_grad_(True)
        elif opt == 'input':
            net_input.requires_grad = True
            params += [net_input]
        else:
            assert False, 'what is it?'
    return params


def optimize(optimizer_type, parameters, closure, LR, num_iter, graph=False):
    """Runs the optimisation loop: `num_iter` x { zero_grad(); closure(); step() }.

    'adam': fused multi-tensor Adam on the GPU arena (torch.optim.Adam defaults).
    'LBFGS': 100 Adam warm-up steps (lr 1e-3), then L-BFGS with max_iter=num_iter, lr=LR and both
             tolerances at -1, exactly as the reference sets torch.optim.LBFGS up -- executed on the
             flat parameter/gradient arenas (dip_optim.ArenaLBFGS).
    graph=True (extension, 'adam' only): the iteration is captured once into a hipGraph after three
    eager iterations and replayed; the closure must then be replay-safe (see
    dip_optim.GraphedIteration)."""
    if optimizer_type == 'LBFGS':
        optimizer = FusedAdam(parameters, lr=0.001)
        for j in range(100):
            optimizer.zero_grad()
            closure()
            optimizer.step()
        print('Starting optimization with LBFGS')

        def closure2():
            optimizer.zero_grad()
            return closure()
        optimizer = ArenaLBFGS(parameters, max_iter=num_iter, lr=LR, tolerance_grad=-1, tolerance_change=-1)
        optimizer.step(closure2)
    elif optimizer_type == 'adam':
        print('Starting optimization with ADAM')
        optimizer = FusedAdam(parameters, lr=LR)
        if graph and num_iter > 3:
            GraphedIteration(optimizer, closure, warmup=3).run(num_iter - 3)
            return
        for j in range(num_iter):
            optimizer.zero_grad()
            closure()
            optimizer.step()
    else:
        assert False
