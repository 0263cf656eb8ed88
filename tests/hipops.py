"""Thin test-side wrappers that drive single libdip_hip.so entry points with NCHW torch tensors
(the layout conversions here are plain torch permutes: test plumbing, not product code)."""
import ctypes as C

import torch

import dip_native as N
from dip_native import round_up


def stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def to_nhwc(x, Cs=None):
    """[1,C,H,W] -> flat NHWC with channel stride Cs (zero padded)."""
    _, Cc, H, W = x.shape
    Cs = Cs or round_up(Cc, 4)
    out = torch.zeros(H, W, Cs, dtype=torch.float32, device=x.device)
    out[:, :, :Cc] = x[0].permute(1, 2, 0)
    return out.contiguous()


def from_nhwc(buf, Cc, H, W, Cs=None):
    Cs = Cs or round_up(Cc, 4)
    return buf.view(H, W, Cs)[:, :, :Cc].permute(2, 0, 1)[None].contiguous()


def pack(w, need_dgrad=True):
    """OIHW weight -> (packed buffer, fwd_off, dgrad_off) through dip_pack_weights."""
    lib = N.lib()
    Cout, Cin, ks, _ = w.shape
    fwd = ks * ks * round_up(Cin, 4) * round_up(Cout, 32)
    dg = ks * ks * round_up(Cout, 4) * round_up(Cin, 32)
    packed = torch.full((fwd + dg,), float("nan"), dtype=torch.float32, device=w.device)
    rec = (N.DipPackRec * 1)()
    rec[0] = N.DipPackRec(0, 0, fwd if need_dgrad else -1, Cout, Cin, ks, round_up(Cin, 4), round_up(Cout, 32),
                          round_up(Cout, 4), round_up(Cin, 32))
    recs = torch.frombuffer(bytearray(bytes(rec)), dtype=torch.uint8).to(w.device)
    wc = w.contiguous().float()
    N.check(lib.dip_pack_weights(wc.data_ptr(), packed.data_ptr(), recs.data_ptr(), 1, fwd + dg, stream(w.device)))
    torch.cuda.synchronize()
    return packed, 0, fwd


def transform(a, b, slope):
    if a is None:
        return N.DipTransform(None, None, 1.0), None
    Cs = round_up(a.numel(), 4)
    t = torch.zeros(2, Cs, dtype=torch.float32, device=a.device)
    t[0, :a.numel()] = a
    t[1, :b.numel()] = b
    return N.DipTransform(t.data_ptr(), t.data_ptr() + 4 * Cs, float(slope)), t


def conv_fwd(x, w, bias, stride, pad_mode, tr=(None, None, 1.0), want_stats=False, split=False):
    """x [1,Cin,H,W], w OIHW -> y [1,Cout,Ho,Wo] (+ stats partial tensor)."""
    lib = N.lib()
    dev = x.device
    _, Cin, H, W = x.shape
    Cout, _, ks, _ = w.shape
    P = (ks - 1) // 2
    Ho, Wo = (H + 2 * P - ks) // stride + 1, (W + 2 * P - ks) // stride + 1
    xb = to_nhwc(x)
    packed, fo, _ = pack(w)
    Cy = round_up(Cout, 4)
    y = torch.full((Ho * Wo * Cy,), float("nan"), dtype=torch.float32, device=dev)
    trd, keep = transform(*tr)
    ksplit, ntiles, wsf = N.conv_plan(Ho, Wo, round_up(Cin, 4), Cout, ks, stride) if split else \
        (1, lib.dip_conv_ntiles(Ho, Wo), 0)
    ws = torch.full((max(wsf, 4),), float("nan"), dtype=torch.float32, device=dev)
    CoutP = round_up(Cout, 32)
    stats = torch.full((ntiles * 3 * CoutP,), float("nan"), dtype=torch.float32, device=dev) if want_stats else None
    bb = bias.contiguous().float() if bias is not None else None
    d = N.DipConvDesc(xb.data_ptr(), H, W, round_up(Cin, 4), round_up(Cin, 4), trd, packed.data_ptr() + 4 * fo,
                      bb.data_ptr() if bb is not None else None, y.data_ptr(), Ho, Wo, Cy, Cout, 0, ks, stride,
                      pad_mode if P > 0 else N.PAD_ZERO, P, 1, 0, stats.data_ptr() if want_stats else None,
                      ksplit, ws.data_ptr() if ksplit > 1 else None)
    N.check(lib.dip_conv_igemm(C.byref(d), stream(dev)), "conv_igemm")
    torch.cuda.synchronize()
    out = from_nhwc(y, Cout, Ho, Wo)
    padvals = y.view(Ho, Wo, Cy)[:, :, Cout:]
    assert torch.all(padvals == 0), "pad channels must be written as zeros"
    if want_stats:
        return out, stats.view(ntiles, 3, CoutP)
    return out


def conv_dgrad(dy, w, stride, pad_mode, Hin, Win, split=False):
    """dy [1,Cout,Ho,Wo] -> gradient wrt the conv input [1,Cin,Hin,Win] (fold included)."""
    lib = N.lib()
    dev = dy.device
    Cout, Cin, ks, _ = w.shape
    P = (ks - 1) // 2
    _, _, Ho, Wo = dy.shape
    reflect = pad_mode in (N.PAD_REFLECT, N.PAD_REPLICATE) and P > 0
    pad = P if reflect else 0
    Hg, Wg = Hin + 2 * pad, Win + 2 * pad
    off = (ks - 1) if reflect else (ks - 1 - P)
    packed, _, do = pack(w)
    dyb = to_nhwc(dy)
    Cg = round_up(Cin, 4)
    g = torch.full((Hg * Wg * Cg,), float("nan"), dtype=torch.float32, device=dev)
    if not split:
        ksplit, wsf = 1, 0
    elif isinstance(split, int) and not isinstance(split, bool):      # forced split-K factor
        ksplit, wsf = split, split * Hg * Wg * Cg
    elif stride == 2:
        ksplit, _, wsf = N.conv_plan_dil2(Hg, Wg, round_up(Cout, 4), Cin, ks)
    else:
        ksplit, _, wsf = N.conv_plan(Hg, Wg, round_up(Cout, 4), Cin, ks, 1)
    ws = torch.full((max(wsf, 4),), float("nan"), dtype=torch.float32, device=dev)
    d = N.DipConvDesc(dyb.data_ptr(), Ho, Wo, round_up(Cout, 4), round_up(Cout, 4), N.DipTransform(None, None, 1.0),
                      packed.data_ptr() + 4 * do, None, g.data_ptr(), Hg, Wg, Cg, Cin, 0, ks, 1, N.PAD_ZERO, off,
                      stride, 0, None, ksplit, ws.data_ptr() if ksplit > 1 else None)
    N.check(lib.dip_conv_igemm(C.byref(d), stream(dev)), "conv_igemm(dgrad)")
    src = N.DipGradSrc(g.data_ptr(), pad, (2 if pad_mode == N.PAD_REPLICATE else 1) if pad else 0, Cg, 0)
    gx = torch.empty(1, Cin, Hin, Win, dtype=torch.float32, device=dev)
    N.check(lib.dip_fold_to_nchw(C.byref(src), Hin, Win, Cin, gx.data_ptr(), stream(dev)), "fold")
    torch.cuda.synchronize()
    return gx


def conv_wgrad(x, dy, ks, stride, pad_mode, tr=(None, None, 1.0), nsplit=None, bias=True, tap_groups=0,
               chan_block=0):
    lib = N.lib()
    dev = x.device
    _, Cin, H, W = x.shape
    _, Cout, Ho, Wo = dy.shape
    P = (ks - 1) // 2
    xb, dyb = to_nhwc(x), to_nhwc(dy)
    CinP, CoutP = round_up(Cin, 32), round_up(Cout, 32)
    nt = lib.dip_conv_wgrad_ntiles(Ho, Wo)
    planned, pg, pcb = N.wgrad_plan2(Ho, Wo, Cin, Cout, ks, stride)
    thin = (ks == 1 and Cout <= 8) or (Cin <= 4 and ks in (3, 5, 7))      # streaming kernels: one slab per block
    if thin or nsplit == "plan":
        nsplit, tap_groups, chan_block = planned, pg, pcb
    else:
        nsplit = nsplit or max(1, min(nt, 7))
    partial = torch.full((nsplit * ks * ks * CinP * CoutP,), float("nan"), dtype=torch.float32, device=dev)
    bpart = torch.full((nsplit * CoutP,), float("nan"), dtype=torch.float32, device=dev)
    trd, keep = transform(*tr)
    d = N.DipWgradDesc(xb.data_ptr(), H, W, round_up(Cin, 4), Cin, trd, dyb.data_ptr(), Ho, Wo, round_up(Cout, 4),
                       Cout, ks, stride, pad_mode if P > 0 else N.PAD_ZERO, P, partial.data_ptr(),
                       bpart.data_ptr() if bias else None, nsplit, tap_groups, chan_block)
    N.check(lib.dip_conv_wgrad(C.byref(d), stream(dev)), "conv_wgrad")
    dw = torch.full((Cout, Cin, ks, ks), float("nan"), dtype=torch.float32, device=dev)
    db = torch.full((Cout,), float("nan"), dtype=torch.float32, device=dev)
    N.check(lib.dip_wgrad_reduce(partial.data_ptr(), bpart.data_ptr() if bias else None, nsplit, ks, Cin, Cout,
                                 dw.data_ptr(), db.data_ptr() if bias else None, stream(dev)), "wgrad_reduce")
    torch.cuda.synchronize()
    return dw, (db if bias else None)


def lrelu_masks(net, spec):
    """LeakyReLU branch pattern (z = a*y + b > 0) realised by the HIP forward, keyed by the
    oracle's BatchNorm key, as CPU bool tensors [1,C,H,W] (see dip_oracle._bn_act)."""
    import dip_oracle as O
    eng = net.__dict__["_dip_engine"]
    keys, _ = O.scale_keys(spec)
    masks = {}
    for i, k in enumerate(keys):
        st = eng.sc[i].st
        for name, key in (("s_act", k.skip_bn), ("d1", k.down_a_bn), ("d2", k.down_b_bn), ("u", k.up_bn),
                          ("u1", k.up1_bn)):
            a = st.get(name)
            if a is None or key is None:
                continue
            state = a.bn.state.view(4, a.Cs)
            y = a.buf.view(a.H, a.W, a.Cs)
            z = torch.addcmul(state[3], state[2], y)            # fma(a, y, b), as the kernels do
            masks[key] = (z[:, :, :a.C] > 0).permute(2, 0, 1)[None].contiguous().cpu()
    return masks


def _zeros_i32(n, dev):
    return torch.zeros(n, dtype=torch.int32, device=dev)


def conv_small_fwd(x, w, bias, stride, pad_mode, tr=(None, None, 1.0), bn=None):
    """dip_conv_small forward: y [1,Cout,Ho,Wo], the partial rows [rows][3][CoutP], and -- bn = dict(gamma, beta, eps,
    momentum, running_mean, running_var) -- the state block / running statistics dip_bn_finalize makes of the rows."""
    lib = N.lib()
    dev = x.device
    _, Cin, H, W = x.shape
    Cout, _, ks, _ = w.shape
    P = (ks - 1) // 2
    Ho, Wo = (H + 2 * P - ks) // stride + 1, (W + 2 * P - ks) // stride + 1
    xb = to_nhwc(x)
    packed, fo, _ = pack(w)
    Cy = round_up(Cout, 4)
    CoutP = round_up(Cout, 32)
    y = torch.full((Ho * Wo * Cy,), float("nan"), dtype=torch.float32, device=dev)
    trd, keep = transform(*tr)
    bb = bias.contiguous().float() if bias is not None else None
    d = N.DipConvDesc(xb.data_ptr(), H, W, round_up(Cin, 4), round_up(Cin, 4), trd, packed.data_ptr() + 4 * fo,
                      bb.data_ptr() if bb is not None else None, y.data_ptr(), Ho, Wo, Cy, Cout, 0, ks, stride,
                      pad_mode if P > 0 else N.PAD_ZERO, P, 1, 0, None, 1, None)
    rows = lib.dip_conv_small_rows(C.byref(d))
    assert rows > 0, "shape not served by dip_conv_small"
    stats = torch.full((rows * 3 * CoutP,), float("nan"), dtype=torch.float32, device=dev)
    d.stats = stats.data_ptr()
    out = {}
    for _ in range(2):                           # (launching twice must not change anything)
        N.check(lib.dip_conv_small(C.byref(d), stream(dev)), "conv_small")
    if bn is not None:
        Cs = round_up(Cout, 4)
        state = torch.full((4 * Cs,), float("nan"), dtype=torch.float32, device=dev)
        gamma, beta = bn["gamma"].to(dev).float().contiguous(), bn["beta"].to(dev).float().contiguous()
        rm, rv = bn["running_mean"].to(dev).float().clone(), bn["running_var"].to(dev).float().clone()
        N.check(lib.dip_bn_finalize(stats.data_ptr(), rows, CoutP, Cout, gamma.data_ptr(), beta.data_ptr(), float(bn["eps"]),
                                    float(bn["momentum"]), state.data_ptr(), Cs, rm.data_ptr(), rv.data_ptr(), stream(dev)),
                "bn_finalize")
        out = dict(state=state.view(4, Cs), running_mean=rm, running_var=rv)
    torch.cuda.synchronize()
    padvals = y.view(Ho, Wo, Cy)[:, :, Cout:]
    assert torch.all(padvals == 0), "pad channels must be written as zeros"
    return from_nhwc(y, Cout, Ho, Wo), stats.view(rows, 3, CoutP), out


def conv_small_dgrad(dy, w, stride, pad_mode, Hin, Win, bnb=None, accumulate_into=None):
    """dip_conv_small data gradient of a conv (OIHW w, stride, pad_mode) wrt its [1,Cin,Hin,Win] input, folded to NCHW.
    bnb = dict(y=[1,Cin,Hin,Win] raw output of the producer conv, state=[4,Cs] (mean, rstd, a, b), slope): phase 1 + 2 of the
    producer BatchNorm's backward ride in the launch; returns (gx, dict(coef, dgamma, dbeta))."""
    lib = N.lib()
    dev = dy.device
    Cout, Cin, ks, _ = w.shape
    P = (ks - 1) // 2
    _, _, Ho, Wo = dy.shape
    reflect = pad_mode in (N.PAD_REFLECT, N.PAD_REPLICATE) and P > 0
    pad = P if reflect else 0
    Hg, Wg = Hin + 2 * pad, Win + 2 * pad
    off = (ks - 1) if reflect else (ks - 1 - P)
    packed, _, do = pack(w)
    dyb = to_nhwc(dy)
    Cg = round_up(Cin, 4)
    g = torch.full((Hg * Wg * Cg,), float("nan"), dtype=torch.float32, device=dev)
    d = N.DipConvDesc(dyb.data_ptr(), Ho, Wo, round_up(Cout, 4), round_up(Cout, 4), N.DipTransform(None, None, 1.0),
                      packed.data_ptr() + 4 * do, None, g.data_ptr(), Hg, Wg, Cg, Cin, 0, ks, 1, N.PAD_ZERO, off,
                      stride, 0, None, 1, None)
    rows = lib.dip_conv_small_rows(C.byref(d))
    assert rows > 0, "shape not served by dip_conv_small"
    out = {}
    if bnb is not None:
        Cs = Cg
        yb = to_nhwc(bnb["y"].to(dev))
        state = bnb["state"].to(dev).float().contiguous()
        assert state.shape == (4, Cs)
        part = torch.full((rows * 2 * Cs,), float("nan"), dtype=torch.float32, device=dev)
        coef = torch.full((2 * Cs,), float("nan"), dtype=torch.float32, device=dev)
        dgamma = torch.full((Cin,), float("nan"), dtype=torch.float32, device=dev)
        dbeta = torch.full((Cin,), float("nan"), dtype=torch.float32, device=dev)
        d.bnb_y, d.bnb_state, d.bnb_partials = yb.data_ptr(), state.data_ptr(), part.data_ptr()
        d.bnb_Cy, d.bnb_Cs, d.bnb_pad, d.bnb_slope = Cg, Cs, pad, float(bnb["slope"])
        out = dict(coef=coef.view(2, Cs), dgamma=dgamma, dbeta=dbeta, keep=(yb, state, part))
    N.check(lib.dip_conv_small(C.byref(d), stream(dev)), "conv_small(dgrad)")
    if bnb is not None:
        N.check(lib.dip_bn_bwd_finalize2(part.data_ptr(), rows, None, 0, 0, Cs, Cin, Hin * Win, dgamma.data_ptr(),
                                         dbeta.data_ptr(), coef.data_ptr(), stream(dev)), "bn_bwd_finalize2")
    src = N.DipGradSrc(g.data_ptr(), pad, (2 if pad_mode == N.PAD_REPLICATE else 1) if pad else 0, Cg, 0)
    gx = torch.empty(1, Cin, Hin, Win, dtype=torch.float32, device=dev)
    N.check(lib.dip_fold_to_nchw(C.byref(src), Hin, Win, Cin, gx.data_ptr(), stream(dev)), "fold")
    torch.cuda.synchronize()
    return gx, out


def conv_dgrad_ring(dy, w, Hin, Win):
    """Data gradient of a reflection-padded 3x3 stride-1 conv as the engine runs it at >= 128 x 128: the interior H x W
    positions through dip_conv_igemm (zero-padded correlation, off = 1) + dip_conv_dgrad_ring for what the reflected ring
    folds onto the frame rows / columns.  Returns the gradient [1,Cin,H,W] (no fold pass)."""
    lib = N.lib()
    dev = dy.device
    Cout, Cin, ks, _ = w.shape
    assert ks == 3
    packed, _, do = pack(w)
    dyb = to_nhwc(dy)
    Cg = round_up(Cin, 4)
    g = torch.full((Hin * Win * Cg,), float("nan"), dtype=torch.float32, device=dev)
    d = N.DipConvDesc(dyb.data_ptr(), Hin, Win, round_up(Cout, 4), round_up(Cout, 4), N.DipTransform(None, None, 1.0),
                      packed.data_ptr() + 4 * do, None, g.data_ptr(), Hin, Win, Cg, Cin, 0, 3, 1, N.PAD_ZERO, 1, 1, 0, None,
                      1, None)
    N.check(lib.dip_conv_igemm(C.byref(d), stream(dev)), "conv_igemm(interior dgrad)")
    N.check(lib.dip_conv_dgrad_ring(C.byref(d), stream(dev)), "conv_dgrad_ring")
    torch.cuda.synchronize()
    return from_nhwc(g, Cin, Hin, Win)


def pack_bf3(w):
    """OIHW 3x3 weight -> (int16 buffer with the three bf16 planes, fwd_off, dgrad_off in elements) through
    dip_pack_weights_bf3."""
    lib = N.lib()
    Cout, Cin, ks, _ = w.shape
    nchF, nchD = (round_up(Cin, 4) + 15) // 16, (round_up(Cout, 4) + 15) // 16
    CoutP, CinP = round_up(Cout, 32), round_up(Cin, 32)
    nf, nd = ks * ks * nchF * 3 * CoutP * 16, ks * ks * nchD * 3 * CinP * 16
    buf = torch.full((nf + nd,), 0x7fc0, dtype=torch.int16, device=w.device)          # bf16 NaN
    rec = (N.DipPackRec3 * 1)()
    rec[0] = N.DipPackRec3(0, 0, nf, Cout, Cin, ks, nchF, CoutP, nchD, CinP)
    recs = torch.frombuffer(bytearray(bytes(rec)), dtype=torch.uint8).to(w.device)
    wc = w.contiguous().float()
    N.check(lib.dip_pack_weights_bf3(wc.data_ptr(), buf.data_ptr(), recs.data_ptr(), 1, (nf + nd) // 3, stream(w.device)))
    torch.cuda.synchronize()
    return buf, 0, nf


def conv_bf3(x, w, bias, pad_mode, tr=(None, None, 1.0), terms=9, dgrad_of=None):
    """3x3 (or 1x1) stride-1 convolution (forward with BatchNorm partials, or -- dgrad_of=(Hin, Win) with x = dy -- the data
    gradient, folded) on the bf16 matrix pipe: dip_conv_igemm with DipConvDesc.wp3 set and dip_conv_bf3_set_terms(terms)."""
    lib = N.lib()
    dev = x.device
    N.check(lib.dip_conv_bf3_set_terms(terms))
    try:
        Cout, Cin, ks, _ = w.shape
        packed, fo, do = pack(w)
        p3, fo3, do3 = pack_bf3(w)
        trd, keep = transform(*tr)
        if dgrad_of is None:
            _, _, H, W = x.shape
            xb = to_nhwc(x)
            Cy, CoutP = round_up(Cout, 4), round_up(Cout, 32)
            y = torch.full((H * W * Cy,), float("nan"), dtype=torch.float32, device=dev)
            ntiles = lib.dip_conv_ntiles(H, W)
            stats = torch.full((ntiles * 3 * CoutP,), float("nan"), dtype=torch.float32, device=dev)
            bb = bias.contiguous().float() if bias is not None else None
            d = N.DipConvDesc(xb.data_ptr(), H, W, round_up(Cin, 4), round_up(Cin, 4), trd, packed.data_ptr() + 4 * fo,
                              bb.data_ptr() if bb is not None else None, y.data_ptr(), H, W, Cy, Cout, 0, ks, 1,
                              pad_mode if ks == 3 else N.PAD_ZERO, (ks - 1) // 2, 1, 0, stats.data_ptr(), 1, None)
            d.wp3 = p3.data_ptr() + 2 * fo3
            assert lib.dip_conv_variant(C.byref(d)) == 7, "descriptor not taken by the bf16-pipe kernel"
            N.check(lib.dip_conv_igemm(C.byref(d), stream(dev)), "conv_igemm(bf3)")
            torch.cuda.synchronize()
            return from_nhwc(y, Cout, H, W), stats.view(ntiles, 3, CoutP)
        Hin, Win = dgrad_of
        reflect = pad_mode in (N.PAD_REFLECT, N.PAD_REPLICATE) and ks == 3
        pad = 1 if reflect else 0
        Hg, Wg = Hin + 2 * pad, Win + 2 * pad
        off = (2 if reflect else 1) if ks == 3 else 0
        dyb = to_nhwc(x)
        Cg = round_up(Cin, 4)
        g = torch.full((Hg * Wg * Cg,), float("nan"), dtype=torch.float32, device=dev)
        d = N.DipConvDesc(dyb.data_ptr(), Hin, Win, round_up(Cout, 4), round_up(Cout, 4), N.DipTransform(None, None, 1.0),
                          packed.data_ptr() + 4 * do, None, g.data_ptr(), Hg, Wg, Cg, Cin, 0, ks, 1, N.PAD_ZERO, off, 1, 0, None,
                          1, None)
        d.wp3 = p3.data_ptr() + 2 * do3
        assert lib.dip_conv_variant(C.byref(d)) in (3, 7)
        N.check(lib.dip_conv_igemm(C.byref(d), stream(dev)), "conv_igemm(bf3 dgrad)")
        src = N.DipGradSrc(g.data_ptr(), pad, (2 if pad_mode == N.PAD_REPLICATE else 1) if pad else 0, Cg, 0)
        gx = torch.empty(1, Cin, Hin, Win, dtype=torch.float32, device=dev)
        N.check(lib.dip_fold_to_nchw(C.byref(src), Hin, Win, Cin, gx.data_ptr(), stream(dev)), "fold")
        torch.cuda.synchronize()
        return gx
    finally:
        lib.dip_conv_bf3_set_terms(-1)
