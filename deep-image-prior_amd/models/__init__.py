"""Model factory, API-compatible with the reference's models/__init__.py:8-31.

Only the 'skip' (MI355X-native) and 'identity' nets are provided: the reference's ResNet / UNet /
texture_nets backbones are outside the accelerated path and raise.
"""
import torch.nn as nn

from .skip import skip
from .downsampler import Downsampler
from .common import Concat, act, bn, conv


def get_net(input_depth, NET_TYPE, pad, upsample_mode, n_channels=3, act_fun='LeakyReLU', skip_n33d=128,
            skip_n33u=128, skip_n11=4, num_scales=5, downsample_mode='stride'):
    if NET_TYPE == 'skip':
        def per_scale(v):
            return [v] * num_scales if isinstance(v, int) else v
        return skip(input_depth, n_channels,
                    num_channels_down=per_scale(skip_n33d),
                    num_channels_up=per_scale(skip_n33u),
                    num_channels_skip=per_scale(skip_n11),
                    upsample_mode=upsample_mode, downsample_mode=downsample_mode,
                    need_sigmoid=True, need_bias=True, pad=pad, act_fun=act_fun)
    if NET_TYPE == 'identity':
        assert input_depth == 3
        return nn.Sequential()
    if NET_TYPE in ('ResNet', 'UNet', 'texture_nets'):
        raise NotImplementedError(
            f"dip-amd: NET_TYPE={NET_TYPE!r} is outside the MI355X-native hot path (skip-net only)")
    assert False, NET_TYPE
