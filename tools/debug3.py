import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
ge.build()
import dip_oracle as O, hipops
from test_net_gpu import _oracle_grads, _grad_report
from models.skip import skip
dev = torch.device("cuda:0")
hw, mode, nskip = (64, 64), "nearest", 128
torch.manual_seed(123)
kw = dict(num_channels_down=[128] * 5, num_channels_up=[128] * 5, num_channels_skip=[nskip] * 5,
          upsample_mode=mode, need_sigmoid=True, need_bias=True, pad="reflection")
net = skip(32, 3, **kw)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()
      if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
z = torch.rand(1, 32, *hw) * 0.1
target = torch.rand(1, 3, *hw)
spec = O.SkipSpec(32, 3, [128] * 5, [128] * 5, [nskip] * 5, pad="reflection", upsample_mode=mode)
lf = lambda o_, dt: torch.nn.functional.mse_loss(o_, target.to(dt))
_, _, g64n = _oracle_grads(spec, sd, z, lf, torch.float64)
oo, lo, g32 = _oracle_grads(spec, sd, z, lf, torch.float32)
net = net.to(dev)
out = net(z.to(dev))
loss = torch.nn.functional.mse_loss(out, target.to(dev))
loss.backward()
torch.cuda.synchronize()
masks = hipops.lrelu_masks(net, spec)
_, _, g64 = _oracle_grads(spec, sd, z, lf, torch.float64, masks)
grads = {k: p.grad for k, p in net.named_parameters()}
print("vs natural:", _grad_report(grads, g64n, g32, g64n))
print("vs masked :", _grad_report(grads, g64, g32, g64n))
for k in ("7.bias", "7.weight", "6.1.weight", "4.weight"):
    print(k, "masked-vs-natural truth diff:", (g64[k] - g64n[k]).norm().item(), " hip-vs-natural:", (grads[k].cpu().double() - g64n[k]).norm().item(),
          " |g|:", g64n[k].norm().item())
print("sd keys sample", list(sd)[:3], "param dtype", next(iter(sd.values())).dtype, "sd tensor device", next(iter(sd.values())).device)
print("is sd aliasing net params?", any(v.data_ptr() == p.data_ptr() for v in sd.values() for p in net.parameters()))

import torch.nn.functional as F
def run(masks_):
    onet = O.OracleNet(spec, {k: v.double() for k, v in sd.items()})
    o = onet(z.double(), None, masks_)
    l = F.mse_loss(o, target.double()); l.backward()
    return o.detach(), {k: p.grad.detach() for k, p in zip(onet.names, onet.params)}
o_nat, g_nat = run(None)
for key in sorted(masks):
    o_m, g_m = run({key: masks[key]})
    d = (g_m["7.bias"] - g_nat["7.bias"]).norm().item()
    if d > 1e-8 or (o_m - o_nat).abs().max().item() > 1e-9:
        print("imposing only", key, "-> out maxdiff %.3e" % (o_m - o_nat).abs().max().item(), " d(7.bias) %.3e" % d,
              " mask shape", tuple(masks[key].shape), masks[key].dtype, masks[key].stride())
print("single-key sweep done")
