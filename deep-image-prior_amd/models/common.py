"""Building blocks of the skip-net, API-compatible with the reference's models/common.py.

The blocks are ordinary torch.nn modules used as PARAMETER HOLDERS: they give the net the
reference's module tree, state_dict() keys and parameter-initialisation RNG order.  On the
MI355X path they are never called -- `models.skip.SkipNet.forward` hands the whole tree to the
HIP engine (dip_engine.py).  Reference semantics being mirrored:
  Module.add child naming "1","2",...   models/common.py:6-9
  Concat (children "0","1", dim=1)      models/common.py:11-42
  act() / bn() / conv()                 models/common.py:76-124
"""
import torch
import torch.nn as nn

from .downsampler import Downsampler


def _append_child(self, module):
    """`Module.add`: register `module` under the 1-based running index (so the first child of a
    Sequential is "1", not "0") -- this is what shapes the reference's state_dict keys."""
    self.add_module(str(len(self) + 1), module)


torch.nn.Module.add = _append_child


class Concat(nn.Module):
    """Runs every child on the same input and concatenates along `dim`.  Children are named
    "0", "1", ...  Inputs of unequal spatial size are centre-cropped to the smallest one."""

    def __init__(self, dim, *branches):
        super().__init__()
        self.dim = dim
        for k, b in enumerate(branches):
            self.add_module(str(k), b)

    def __len__(self):
        return len(self._modules)

    def forward(self, x):
        ys = [m(x) for m in self._modules.values()]
        h = min(y.shape[2] for y in ys)
        w = min(y.shape[3] for y in ys)
        cropped = []
        for y in ys:
            if y.shape[2] != h or y.shape[3] != w:
                t, l = (y.size(2) - h) // 2, (y.size(3) - w) // 2
                y = y[:, :, t:t + h, l:l + w]
            cropped.append(y)
        return torch.cat(cropped, dim=self.dim)


class GenNoise(nn.Module):
    """Emits N(0,1) noise shaped like its input but with `dim2` channels (unused by skip())."""

    def __init__(self, dim2):
        super().__init__()
        self.dim2 = dim2

    def forward(self, x):
        shape = list(x.size())
        shape[1] = self.dim2
        return torch.zeros(shape, dtype=x.dtype, device=x.device).normal_()


class Swish(nn.Module):
    def __init__(self):
        super().__init__()
        self.s = nn.Sigmoid()

    def forward(self, x):
        return x * self.s(x)


def act(act_fun='LeakyReLU'):
    """Activation factory: 'LeakyReLU' (slope 0.2, in place) | 'Swish' | 'ELU' | 'none', or a
    module class to instantiate (reference: models/common.py:76-92).  All four strings run on the
    gfx950 engine (DipTransform.slope encodes them, csrc/dip_common.h: dip_act / dip_act_grad);
    a module CLASS is instantiated as in the reference and runs natively when it builds one of the stateless
    activations the kernels know (models/skip.py: _act_code_of_module_class); others raise at the first forward."""
    if not isinstance(act_fun, str):
        return act_fun()
    table = {
        'LeakyReLU': lambda: nn.LeakyReLU(0.2, inplace=True),
        'Swish': Swish,
        'ELU': nn.ELU,
        'none': nn.Sequential,
    }
    assert act_fun in table, act_fun
    return table[act_fun]()


def bn(num_features):
    return nn.BatchNorm2d(num_features)


def conv(in_f, out_f, kernel_size, stride=1, bias=True, pad='zero', downsample_mode='stride'):
    """[ReflectionPad2d] + Conv2d [+ pooling/Lanczos down-sampler].  With pad='reflection' a
    padder module is ALWAYS present (ReflectionPad2d(0) for 1x1), so the Conv2d is child "1";
    with pad='zero' it is child "0"."""
    pool = None
    if stride != 1 and downsample_mode != 'stride':
        if downsample_mode == 'avg':
            pool = nn.AvgPool2d(stride, stride)
        elif downsample_mode == 'max':
            pool = nn.MaxPool2d(stride, stride)
        elif downsample_mode in ('lanczos2', 'lanczos3'):
            pool = Downsampler(n_planes=out_f, factor=stride, kernel_type=downsample_mode, phase=0.5,
                               preserve_size=True, _dense=True)
        else:
            raise AssertionError(downsample_mode)
        stride = 1

    half = int((kernel_size - 1) / 2)
    mods = []
    if pad == 'reflection':
        mods.append(nn.ReflectionPad2d(half))
        half = 0
    mods.append(nn.Conv2d(in_f, out_f, kernel_size, stride, padding=half, bias=bias))
    if pool is not None:
        mods.append(pool)
    return nn.Sequential(*mods)
