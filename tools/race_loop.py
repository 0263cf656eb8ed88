"""Determinism stress for a launch schedule: the switch probe's net (256x256, two scales, 128 channels), fused loss head, N
backward passes in ONE process, every gradient compared bit for bit with the first pass.  A cross-stream race shows as a
mismatch.    [DIP_DEFER_WGRAD=-1 ...] python tools/race_loop.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()


def main():
    from models.skip import skip
    from utils.loss_head import MSEHead
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    hw = (int(os.environ.get("RACE_HW", "256")),) * 2          # RACE_HW=512 RACE_NET=deep: the headline net
    nsc = 5 if os.environ.get("RACE_NET") == "deep" else 2       # deep: the default net's five scales (conv_small, bn_bwd_one, ...)
    kw = dict(num_channels_down=[128] * nsc, num_channels_up=[128] * nsc, num_channels_skip=[4] * nsc,
              upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")
    torch.manual_seed(5)
    z = (torch.rand(1, 4, *hw) * 0.1).to(dev)
    target = torch.rand(1, 3, *hw).to(dev)
    mask = (torch.rand(1, 1, *hw) > 0.3).float().to(dev)
    torch.manual_seed(6)
    net = skip(4, 3, **kw).to(dev)
    head = MSEHead(net, target, mask=mask)
    plain = os.environ.get("RACE_PLAIN") == "1"
    ref, bad = None, 0
    for it in range(n):
        for p in net.parameters():
            p.grad = None
        if plain and it % 2 == 0:
            out = net(z)
            loss = torch.nn.functional.mse_loss(out * mask, target * mask)
        else:
            loss, _ = head(z)
        loss.backward()
        torch.cuda.synchronize()
        cur = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        key = "plain" if (plain and it % 2 == 0) else "head"
        if ref is None:
            ref = {}
        if key not in ref:
            ref[key] = cur
            continue
        diff = [k for k in cur if not torch.equal(cur[k], ref[key][k])]
        if diff:
            bad += 1
            worst = max(diff, key=lambda k: float((cur[k].double() - ref[key][k].double()).abs().max()))
            if bad == 1:
                print("   differing:", diff, flush=True)
                eng = net.__dict__["_dip_engine"]
                print("   bwd ops:", [n for _, _, n in eng.bwd_ops], flush=True)
            print(f"pass {it} ({key}): {len(diff)} tensors differ, e.g. {worst}: max|d| "
                  f"{float((cur[worst].double() - ref[key][worst].double()).abs().max()):.3e}", flush=True)
    print(f"{n} passes, {bad} with a mismatch", {k: v for k, v in os.environ.items() if k.startswith('DIP_')})


if __name__ == "__main__":
    main()
