"""One forward + backward of a 256x256 two-scale 128-channel skip-net on the MI355X backend in a process of its own,
so that the A/B switches of the library (DIP_* environment variables, read once per process) can be set per run.
Writes the output, the loss and every gradient to <out.npz>.  Used by
tests/test_net_gpu.py::test_ab_switch_branches_compute_the_same_gradients.  Test infrastructure only.

    [DIP_...=1 ...] python tests/switch_probe.py <out.npz>
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()


def main():
    from models.skip import skip
    from utils.loss_head import MSEHead
    dev = torch.device("cuda:0")
    if os.environ.get("PROBE_POISON"):           # debugging aid: fresh allocations of this process hold 1e30 instead of zeros
        junk = [torch.full((1 << 26,), 1e30, device=dev) for _ in range(8)]      # 2 GiB
        torch.cuda.synchronize()
        del junk
    hw = (256, 256)     # >= 65536 pixels: one-pass launches, the weights-resident 1x1 kernel, 132-column data gradients
    kw = dict(num_channels_down=[128, 128], num_channels_up=[128, 128], num_channels_skip=[4, 4],
              upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")
    torch.manual_seed(5)
    z = (torch.rand(1, 4, *hw) * 0.1).to(dev)        # 4 input planes: the thin-input weight-gradient kernel
    target = torch.rand(1, 3, *hw).to(dev)
    mask = (torch.rand(1, 1, *hw) > 0.3).float().to(dev)
    torch.manual_seed(6)
    net = skip(4, 3, **kw).to(dev)
    rec = {}
    # the notebooks' spelling ...
    out = net(z)
    loss = torch.nn.functional.mse_loss(out * mask, target * mask)
    loss.backward()
    torch.cuda.synchronize()
    rec["out"] = out.detach().cpu().numpy()
    rec["loss"] = np.float64(loss.item())
    for k, p in net.named_parameters():
        rec["g/" + k] = p.grad.detach().cpu().numpy()
        p.grad = None
    # LeakyReLU branch pattern of this forward (bit-packed sign of z = a*y + b per BatchNorm + activation, as
    # tests/hipops.lrelu_masks): two runs whose forwards differ in the last bit may disagree on a few elements
    eng = net.__dict__["_dip_engine"]
    for i, sc in enumerate(eng.sc):
        for name in ("s_act", "d1", "d2", "u", "u1"):
            a = sc.st.get(name)
            if a is None or getattr(a, "bn", None) is None:
                continue
            state = a.bn.state.view(4, a.Cs)
            zz = torch.addcmul(state[3], state[2], a.buf.view(a.H, a.W, a.Cs))
            rec[f"m/s{i}.{name}"] = np.packbits((zz[:, :, :a.C] > 0).cpu().numpy().reshape(-1))
    # ... and the fused loss head (dip_loss_head_*)
    head = MSEHead(net, target, mask=mask)
    loss2, _ = head(z)
    loss2.backward()
    torch.cuda.synchronize()
    rec["loss_head"] = np.float64(loss2.item())
    for k, p in net.named_parameters():
        rec["gh/" + k] = p.grad.detach().cpu().numpy()
    rec["n_bwd_ops"] = np.int64(len(eng.bwd_ops))
    np.savez(sys.argv[1], **rec)
    print("probe ok", {k: v for k, v in os.environ.items() if k.startswith("DIP_")})


if __name__ == "__main__":
    main()
