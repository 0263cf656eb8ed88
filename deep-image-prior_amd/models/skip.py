"""`skip()` -- the hour-glass encoder-decoder with skip connections, API- and state_dict-
compatible with the reference's models/skip.py:5-100, executed by hand-written gfx950 kernels.

The function builds the SAME torch.nn module tree as the reference (same child names, same
construction order -> same parameter RNG stream under torch.manual_seed, same state_dict keys
such as `1.0.1.1.weight`), but the returned object is a `SkipNet`: an nn.Sequential whose
forward() hands the whole tree to the HIP engine (dip_engine.SkipEngine) instead of calling the
children.  There is no CPU / eager fallback.
"""
import torch
import torch.nn as nn

from .common import Concat, act, bn, conv


def _listify(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


class SkipNet(nn.Sequential):
    """nn.Sequential-shaped container (so `.parameters()`, `.state_dict()`, `.type(dtype)`,
    `.cuda()` behave as for the reference net) whose forward runs on the MI355X engine."""

    def forward(self, input):
        eng = self.__dict__.get('_dip_engine')
        if eng is None:
            raise RuntimeError("dip-amd: this SkipNet was not built by models.skip.skip()")
        if isinstance(eng, Exception):
            raise eng
        import dip_engine
        return dip_engine.run_net(eng, input)

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ('_dip_engine',):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        spec = self.__dict__.get('_dip_spec')
        if spec is not None:
            _attach_engine(new, *spec)
        return new


def _attach_engine(model, scale_paths, out_path, need_sigmoid, pad, act_fun):
    """(Re)creates the HIP engine for `model` from child-name paths (so deepcopy works)."""
    import dip_engine

    def at(path):
        m = model
        for name in path:
            m = m._modules[name]
        return m

    scales = []
    for sp in scale_paths:
        s = dip_engine.ScalePlan()
        s.ns, s.upsample_mode = sp['ns'], sp['upsample_mode']
        s.pool = sp.get('pool')
        for k in ('skip_conv', 'skip_bn', 'down_a', 'down_a_bn', 'down_b', 'down_b_bn', 'cat_bn', 'up', 'up_bn',
                  'up1', 'up1_bn', 'down_ds'):
            setattr(s, k, at(sp[k]) if sp.get(k) is not None else None)
        scales.append(s)
    model.__dict__['_dip_spec'] = (scale_paths, out_path, need_sigmoid, pad, act_fun)
    # DipTransform.slope code of the activation (include/dip_hip.h): LeakyReLU(0.2) | none | Swish | ELU
    act_codes = {'LeakyReLU': 0.2, 'none': 1.0, 'Swish': -1.0, 'ELU': -2.0}
    try:
        if isinstance(act_fun, str) and act_fun not in act_codes:
            raise NotImplementedError(f"dip-amd: act_fun={act_fun!r} has no gfx950 kernel (the strings LeakyReLU, Swish, ELU and "
                                      "'none' do, as in the reference's act(); module classes: _act_code_of_module_class)")
        act_code = act_codes[act_fun] if isinstance(act_fun, str) else _act_code_of_module_class(act_fun)
        for sp in scale_paths:
            if sp['unsupported']:
                raise NotImplementedError("dip-amd: " + sp['unsupported'])
        model.__dict__['_dip_engine'] = dip_engine.SkipEngine(model, scales, at(out_path), need_sigmoid, pad,
                                                              act_slope=act_code)
    except NotImplementedError as e:   # surfaces at the first forward(), construction stays cheap
        model.__dict__['_dip_engine'] = e


def _act_code_of_module_class(act_fun):
    """act_fun given as a module class / factory (models/common.py:90-91 of the reference: `return act_fun()`): the stateless
    element-wise activations that the kernels' loaders know (DipTransform.slope codes, include/dip_hip.h) are recognised by
    instantiating one -- nn.LeakyReLU (any negative_slope in [0, 1]), nn.ReLU, nn.ELU(alpha=1), nn.SiLU / Swish, nn.Identity /
    an empty nn.Sequential; anything else has no gfx950 kernel."""
    import torch.nn as nn
    from .common import Swish
    m = act_fun()
    if isinstance(m, nn.LeakyReLU) and 0.0 < m.negative_slope <= 1.0:
        return float(m.negative_slope)
    if isinstance(m, nn.ReLU) or (isinstance(m, nn.LeakyReLU) and m.negative_slope == 0.0):
        return -3.0                     # DIP_ACT_RELU
    if isinstance(m, nn.ELU) and m.alpha == 1.0:
        return -2.0
    if isinstance(m, (nn.SiLU, Swish)):
        return -1.0
    if isinstance(m, nn.Identity) or (isinstance(m, nn.Sequential) and len(m) == 0):
        return 1.0
    raise NotImplementedError(f"dip-amd: act_fun={act_fun!r} builds a {type(m).__name__}: no gfx950 kernel (LeakyReLU, ReLU, "
                              "ELU(alpha=1), SiLU / Swish and Identity have one)")


def skip(
        num_input_channels=2, num_output_channels=3,
        num_channels_down=[16, 32, 64, 128, 128], num_channels_up=[16, 32, 64, 128, 128],
        num_channels_skip=[4, 4, 4, 4, 4],
        filter_size_down=3, filter_size_up=3, filter_skip_size=1,
        need_sigmoid=True, need_bias=True,
        pad='zero', upsample_mode='nearest', downsample_mode='stride', act_fun='LeakyReLU',
        need1x1_up=True):
    """Assembles the encoder-decoder with skip connections.

    Arguments (identical to the reference):
        act_fun: 'LeakyReLU|Swish|ELU|none' or a module class
        pad: 'zero|reflection'
        upsample_mode: 'nearest|bilinear' (or a per-scale list)
        downsample_mode: 'stride|avg|max|lanczos2|lanczos3' (or a per-scale list)
    """
    assert len(num_channels_down) == len(num_channels_up) == len(num_channels_skip)
    n = len(num_channels_down)
    upsample_mode = _listify(upsample_mode, n)
    downsample_mode = _listify(downsample_mode, n)
    filter_size_down = _listify(filter_size_down, n)
    filter_size_up = _listify(filter_size_up, n)

    model = SkipNet()
    level = model                  # the Sequential that receives this scale's modules
    path = []                      # child-name path of `level` inside `model`
    depth_in = num_input_channels
    scale_paths = []

    for i in range(n):
        ns, nd, nu = num_channels_skip[i], num_channels_down[i], num_channels_up[i]
        deeper, side = nn.Sequential(), nn.Sequential()
        sp = {'ns': ns, 'upsample_mode': upsample_mode[i], 'unsupported': None}

        # -- child "1": Concat(side, deeper) or deeper alone; child "2": BatchNorm over the concat
        if ns != 0:
            level.add(Concat(1, side, deeper))
            side_path, deep_path = path + ['1', '0'], path + ['1', '1']
        else:
            level.add(deeper)
            side_path, deep_path = None, path + ['1']
        k_deep = num_channels_up[i + 1] if i < n - 1 else nd
        level.add(bn(ns + k_deep))
        sp['cat_bn'] = path + ['2']

        # -- skip branch: 1x1 conv -> BN -> act
        if ns != 0:
            side.add(conv(depth_in, ns, filter_skip_size, bias=need_bias, pad=pad))
            side.add(bn(ns))
            side.add(act(act_fun))
            sp['skip_conv'] = side_path + ['1']
            sp['skip_bn'] = side_path + ['2']

        # -- deeper branch: strided conv -> BN -> act -> conv -> BN -> act -> [next scale] -> upsample
        deeper.add(conv(depth_in, nd, filter_size_down[i], 2, bias=need_bias, pad=pad,
                        downsample_mode=downsample_mode[i]))
        deeper.add(bn(nd))
        deeper.add(act(act_fun))
        deeper.add(conv(nd, nd, filter_size_down[i], bias=need_bias, pad=pad))
        deeper.add(bn(nd))
        deeper.add(act(act_fun))
        sp['down_a'], sp['down_a_bn'] = deep_path + ['1'], deep_path + ['2']
        sp['down_b'], sp['down_b_bn'] = deep_path + ['4'], deep_path + ['5']
        sp['pool'] = None
        if downsample_mode[i] in ('avg', 'max'):
            sp['pool'] = downsample_mode[i]     # stride-1 conv + AvgPool2d / MaxPool2d(2, 2): dip_{avg,max}pool2_fwd/bwd
        elif downsample_mode[i] in ('lanczos2', 'lanczos3'):
            # the reference puts a Downsampler with a TRAINABLE dense nd x nd x 8 x 8 (12 x 12) stride-2 conv weight
            # behind the stride-1 conv (models/common.py:107-110): the engine runs it as a convolution of its own
            # (replication padding, ScalePlan.down_ds); the BatchNorm that follows normalises ITS output
            sp['pool'] = 'lanczos'
        elif downsample_mode[i] != 'stride':
            sp['unsupported'] = f"downsample_mode={downsample_mode[i]!r} has no gfx950 kernel"

        inner = nn.Sequential()
        if i != n - 1:
            deeper.add(inner)
        deeper.add(nn.Upsample(scale_factor=2, mode=upsample_mode[i]))
        if upsample_mode[i] not in ('nearest', 'bilinear'):
            sp['unsupported'] = f"upsample_mode={upsample_mode[i]!r} has no gfx950 kernel"

        # -- decoder: conv -> BN -> act [-> 1x1 conv -> BN -> act]
        level.add(conv(ns + k_deep, nu, filter_size_up[i], 1, bias=need_bias, pad=pad))
        level.add(bn(nu))
        level.add(act(act_fun))
        sp['up'], sp['up_bn'] = path + ['3'], path + ['4']
        if need1x1_up:
            level.add(conv(nu, nu, 1, bias=need_bias, pad=pad))
            level.add(bn(nu))
            level.add(act(act_fun))
            sp['up1'], sp['up1_bn'] = path + ['6'], path + ['7']
        scale_paths.append(sp)

        depth_in = nd
        level = inner
        path = deep_path + ['7']

    model.add(conv(num_channels_up[0], num_output_channels, 1, bias=need_bias, pad=pad))
    out_path = [str(len(model))]
    if need_sigmoid:
        model.add(nn.Sigmoid())

    # conv() blocks are Sequentials: resolve paths down to the nn.Conv2d inside them
    def conv_path(p):
        blk = model
        for name in p:
            blk = blk._modules[name]
        for name, m in blk._modules.items():
            if isinstance(m, nn.Conv2d):
                return p + [name]
        raise AssertionError(p)

    for sp in scale_paths:
        if sp.get('pool') == 'lanczos':       # conv() = Sequential([pad], Conv2d, Downsampler): its dense holder conv
            blk = model
            for name in sp['down_a']:
                blk = blk._modules[name]
            ds = [name for name, m in blk._modules.items() if type(m).__name__ == 'Downsampler']
            sp['down_ds'] = sp['down_a'] + [ds[0], 'downsampler_']
        for k in ('skip_conv', 'down_a', 'down_b', 'up', 'up1'):
            if sp.get(k) is not None:
                sp[k] = conv_path(sp[k])
    _attach_engine(model, scale_paths, conv_path(out_path), need_sigmoid, pad,
                   act_fun)
    return model
