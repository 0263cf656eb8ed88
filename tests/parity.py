"""Shared parity criterion of the whole-net tests (tests/test_net_gpu.py, tests/test_fullsize_gpu.py)
and of __graft_entry__.smoke(): the HIP skip-net against the CPU oracle (oracle/dip_oracle.py).

Iteration-1 criterion:
  * output  >= 100 dB PSNR vs the oracle's fp32 output,
  * loss    rel. err <= 1e-5,
  * every gradient tensor k that is NOT analytically zero -- purely relative:
        ||g_hip_k - g64_k|| <= RATIO * ||g_ref32_k - g64n_k|| + FLOOR * ||g64_k||
    (g64 = fp64 oracle on the LeakyReLU branch pattern the HIP forward realised, g64n = fp64 oracle
    on its own pattern, g_ref32 = the reference's own fp32 CPU path: the HIP gradient has to be as
    close to the fp64 truth as the reference itself is, up to RATIO for the different summation
    order of the matrix core, plus an fp32 roundoff floor relative to THAT tensor's norm);
  * the analytically-zero tensors (bias of every conv that feeds a train-mode BatchNorm: the
    BatchNorm subtracts the batch mean, so d loss / d bias == 0 exactly; SURVEY 8c "redundant
    biases"; likewise beta of the concat BatchNorm in front of a reflection-padded conv + BatchNorm)
    hold roundoff in both implementations; they are sums of O(max_k ||g_k||) terms that
    cancel, so their bound is absolute:  ||g_hip_k|| <= RATIO * ||g_ref32_k|| + ZFLOOR * max_k ||g64_k||.
The unmasked comparison (against g64n, the oracle's own branch pattern) is reported next to the
masked one so the effect of imposing the HIP branch pattern stays visible.

Binding on top of the ratio (round-3, VERDICT r02 "make parity binding"):
  * SURVEY 8c's plain criterion: rel-L2 of every non-zero gradient tensor against the fp64 truth <= REL_L2
    (1e-4) -- `check()` asserts it next to the ratio -- wherever the REFERENCE's own fp32 path meets it; on the
    tensors where the reference itself is worse than 1e-4 (sums with heavy cancellation, e.g. beta of the first
    BatchNorm at 256^2: reference 1.2e-3, HIP 2.9e-4) the bound is RATIO x the reference's own rel-L2 (round 4; round 3
    had 1 x, which two correct fp32 evaluations of a roundoff-dominated sum meet only by chance; measured: HIP worst
    9e-6 .. 2.3e-3, reference worst 1.3e-3 .. 2.5e-2);
  * the masked truth cannot hide a real error: the elements whose LeakyReLU branch differs between the HIP
    forward and the fp64 oracle are counted (`mask_report`); over the whole net they must be fewer than MASK_FRAC
    of the elements (measured: 0 .. 10 of 1e5 .. 4e6 elements at test sizes, 151 of 1.1e8 at 512^2) and every one
    of them must sit at |z| <= MASK_Z * rms(z) of the fp64 pre-activation (measured <= 2.8e-5: the roundoff band
    of an fp32 forward through ~25 layers), where both branches are a correct fp32 answer.
"""
import numpy as np
import torch

import dip_oracle as O

RATIO = 4.0        # summation-order factor (sequential fp32 MFMA accumulation over K <= 2304 / 4096-pixel slabs)
FLOOR = 2e-5       # fp32 roundoff floor, relative to the tensor's own norm
ZFLOOR = 1e-7      # roundoff floor of the analytically-zero tensors, relative to the largest gradient norm
REL_L2 = 1e-4      # every non-zero gradient tensor, rel-L2 against the fp64 truth
MASK_FRAC = 1e-5   # LeakyReLU branch mismatches HIP vs fp64 oracle: fraction of the net's activated elements ...
MASK_Z = 1e-4      # ... and how far from the kink (|z| / rms(z)) a mismatching element may sit


def zero_grad_keys(spec, sd=None):
    """Names of the parameters whose gradient is analytically zero:
      * the bias of every conv that is followed by a train-mode BatchNorm (all convs of skip() except
        the output conv, models/skip.py:57-98 of the reference; with downsample_mode='avg' the pooling
        in between is linear and shift-preserving, so the statement still holds);
      * beta of the BatchNorm over the concat (see below);
      * with the state_dict `sd` given: gamma of a BatchNorm whose beta is all zero (the default
        initialisation) and whose output reaches, through LeakyReLU (positively homogeneous) and
        up-sampling / concatenation only, a PER-CHANNEL BatchNorm -- the skip-branch BatchNorm, the last
        BatchNorm of every scale below the top, and the deepest down_b BatchNorm: scaling one of their
        channels is undone by the concat BatchNorm, so d loss / d gamma_c == 0 for every channel."""
    keys, _ = O.scale_keys(spec)
    out = set()
    n = spec.n_scales
    for i, k in enumerate(keys):
        if spec.need_bias:
            for c in (k.skip_conv, k.down_a, k.down_b, k.up, k.up1):
                if c is not None:
                    out.add(c + ".bias")
        if getattr(k, "down_a_ds", None):       # the Downsampler's conv (always biased) feeds the BatchNorm directly
            out.add(k.down_a_ds + ".bias")
        # beta of the BatchNorm over the concat: it feeds (without an activation) the reflection-padded
        # decoder conv, whose output BatchNorm removes the per-channel constant a constant input shift
        # produces (with zero padding the border breaks this, so only for pad == 'reflection')
        if spec.pad == "reflection":
            out.add(k.cat_bn + ".bias")
        if sd is not None:
            cands = [k.skip_bn] if k.skip_bn else []
            if i >= 1:
                cands.append(k.up1_bn if k.up1_bn else k.up_bn)
            if i == n - 1:
                cands.append(k.down_b_bn)
            for b in cands:
                beta = sd.get(b + ".bias")
                if beta is not None and float(torch.as_tensor(beta).abs().max()) == 0.0:
                    out.add(b + ".weight")
    return out


def oracle_grads(spec, sd, z, loss_fn, dtype, masks=None, z_requires_grad=False, zrec=None):
    """Oracle forward/backward in `dtype` (fp64 = the truth, fp32 = the reference's own roundoff).
    `masks`: LeakyReLU branch pattern of the HIP forward (hipops.lrelu_masks); `zrec`: dict that receives the
    oracle's pre-activations (for mask_report)."""
    onet = O.OracleNet(spec, {k: v.to(dtype) for k, v in sd.items()})
    zz = z.to(dtype)
    if z_requires_grad:
        zz = zz.clone().requires_grad_(True)
    out = onet(zz, None, masks, zrec)
    loss = loss_fn(out, dtype)
    loss.backward()
    grads = {k: p.grad.detach() for k, p in zip(onet.names, onet.params)}
    if z_requires_grad:
        grads["__input__"] = zz.grad.detach()
    return out.detach(), loss.item(), grads


def grad_report(named_grads, g64, g32, g64n, zero_keys, ratio=RATIO, floor=FLOOR, zfloor=ZFLOOR):
    """Returns {"worst": err/tol over all tensors (masked truth), "worst_key", "worst_unmasked",
    "worst_unmasked_key", "worst_zero", "n_zero"}; the test asserts worst <= 1."""
    dbl = lambda t: torch.as_tensor(t).detach().cpu().double()
    gscale = max(dbl(v).norm().item() for k, v in g64.items() if k not in zero_keys)
    rep = {"worst": 0.0, "worst_key": None, "worst_unmasked": 0.0, "worst_unmasked_key": None, "worst_zero": 0.0,
           "n_zero": 0, "worst_rel": 0.0, "worst_rel_key": None, "worst_rel_ref": 0.0, "worst_rel_excess": 0.0,
           "relaxed": []}          # tensors whose rel-L2 bound is RATIO x the reference's own (> REL_L2): the set is printed, so it cannot grow silently
    for k, g in named_grads.items():
        g = dbl(g)
        t, tn, r = dbl(g64[k]), dbl(g64n[k]), dbl(g32[k])
        if k in zero_keys:
            rep["n_zero"] += 1
            e_hip, e_ref = g.norm().item(), r.norm().item()
            tol = ratio * e_ref + zfloor * gscale + 1e-30
            q = e_hip / tol
            rep["worst_zero"] = max(rep["worst_zero"], q)
            desc = f"{k} [analytically zero] (|g_hip| {e_hip:.2e}, |g_ref32| {e_ref:.2e}, gscale {gscale:.2e})"
            qn = q
        else:
            e_hip, e_ref = (g - t).norm().item(), (r - tn).norm().item()
            tol = ratio * e_ref + floor * t.norm().item() + 1e-30
            q = e_hip / tol
            qn = (g - tn).norm().item() / (ratio * e_ref + floor * tn.norm().item() + 1e-30)
            desc = f"{k} (err {e_hip:.2e}, ref-fp32 err {e_ref:.2e}, |g| {t.norm().item():.2e})"
            rel, rel_ref = e_hip / (t.norm().item() + 1e-30), e_ref / (tn.norm().item() + 1e-30)
            rep["worst_rel"] = max(rep["worst_rel"], rel)
            rep["worst_rel_ref"] = max(rep["worst_rel_ref"], rel_ref)
            # plain SURVEY bound wherever the reference's own fp32 path meets it; a tensor on which the reference itself
            # misses 1e-4 is roundoff-dominated (a sum with heavy cancellation: beta of a 4-channel skip BatchNorm,
            # |g| ~ 1e-5 from 16384 terms) -- two correct fp32 evaluations differ there by a random factor, so the bound is
            # RATIO x the reference's own rel-L2, the same factor as the primary criterion (round 4: with "1 x the
            # reference" the test was a coin flip per such tensor -- 2.3e-3 vs the reference's 8.4e-4 on s2.skip_bn.beta at
            # 512^2 after the low-resolution layers changed their summation order, 0.69 of the primary bound)
            excess = rel / max(REL_L2, RATIO * rel_ref)       # continuous in rel_ref (the 1e-4 -> 4e-4 jump at rel_ref = 1e-4 is gone)
            if RATIO * rel_ref > REL_L2:
                rep["relaxed"].append(f"{k} (reference rel-L2 {rel_ref:.1e}, HIP {rel:.1e})")
            if excess > rep["worst_rel_excess"]:
                rep["worst_rel_excess"], rep["worst_rel_key"] = excess, desc
        if q > rep["worst"]:
            rep["worst"], rep["worst_key"] = q, desc
        if qn > rep["worst_unmasked"]:
            rep["worst_unmasked"], rep["worst_unmasked_key"] = qn, k
    return rep


def mask_report(masks_hip, zrec):
    """LeakyReLU branch pattern of the HIP forward (hipops.lrelu_masks) against the fp64 oracle's own
    pre-activations `zrec` (oracle_grads(..., zrec=...) of the unmasked fp64 run): per BatchNorm the number of
    elements on different branches and how far from the kink the farthest of them sits."""
    rep = {"n": 0, "frac": 0.0, "frac_key": None, "zrel": 0.0, "zrel_key": None, "numel": 0}
    for key, m in masks_hip.items():
        z = zrec[key]
        diff = m.to(torch.bool) != (z > 0)
        n = int(diff.sum())
        rep["n"] += n
        rep["numel"] += z.numel()
        if n:
            frac = n / z.numel()
            zrel = float(z[diff].abs().max()) / (float(z.double().pow(2).mean().sqrt()) + 1e-30)
            if frac > rep["frac"]:
                rep["frac"], rep["frac_key"] = frac, key
            if zrel > rep["zrel"]:
                rep["zrel"], rep["zrel_key"] = zrel, key
    return rep


def check(rep, mrep=None):
    """The binding assertions of the iteration-1 gradient criterion (see the module docstring)."""
    assert rep["worst"] <= 1.0, fmt(rep)
    assert rep["worst_zero"] <= 1.0, fmt(rep)
    assert rep["worst_rel_excess"] <= 1.0, (f"rel-L2 beyond {REL_L2} (or {RATIO} x the reference's own where that misses it) by x{rep['worst_rel_excess']:.2f} "
                                            f"[{rep['worst_rel_key']}]; " + fmt(rep))
    if mrep is not None:
        assert mrep["n"] < MASK_FRAC * mrep["numel"] + 1 and mrep["zrel"] <= MASK_Z, fmt_masks(mrep)


def fmt_masks(mrep):
    return (f"LeakyReLU branch mismatches vs fp64 oracle: {mrep['n']} of {mrep['numel']} elements, worst tensor "
            f"{mrep['frac']:.1e} [{mrep['frac_key']}], farthest from the kink |z|/rms {mrep['zrel']:.1e} [{mrep['zrel_key']}]")


def fmt(rep):
    return (f"grad err/tol masked {rep['worst']:.2f} [{rep['worst_key']}], unmasked {rep['worst_unmasked']:.2f} "
            f"[{rep['worst_unmasked_key']}], zero-tensors {rep['worst_zero']:.2f} (n={rep['n_zero']}), "
            f"worst rel-L2 {rep['worst_rel']:.2e} (reference fp32 vs its fp64: {rep['worst_rel_ref']:.2e}; "
            f"vs max(1e-4, {RATIO:g} x reference) x{rep['worst_rel_excess']:.2f}; {len(rep.get('relaxed', []))} tensor(s) on the relaxed bound"
            + (": " + "; ".join(rep["relaxed"][:6]) if rep.get("relaxed") else "") + ")")


def psnr(a, b):
    return O.psnr(np.asarray(a), np.asarray(b))
