"""GPU debug: per-tensor gradient error table vs the fp64 oracle, and isolation of individual
backward kernels by recomputing them in fp64 from the engine's OWN buffers."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dip_oracle as O
from models.skip import skip

dev = torch.device("cuda:0")
torch.manual_seed(123)
hw, mode, nskip = (64, 64), "nearest", 128
kw = dict(num_channels_down=[128] * 5, num_channels_up=[128] * 5, num_channels_skip=[nskip] * 5,
          upsample_mode=mode, need_sigmoid=True, need_bias=True, pad="reflection")
net = skip(32, 3, **kw)
sd = {k: v.detach().clone() for k, v in net.state_dict().items()
      if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
z = torch.rand(1, 32, *hw) * 0.1
target = torch.rand(1, 3, *hw)
spec = O.SkipSpec(32, 3, [128] * 5, [128] * 5, [nskip] * 5, pad="reflection", upsample_mode=mode)


def oracle(dt):
    onet = O.OracleNet(spec, {k: v.to(dt) for k, v in sd.items()})
    taps = {}
    out = onet(z.to(dt), taps)
    for t in taps.values():
        t.retain_grad()
    loss = torch.nn.functional.mse_loss(out, target.to(dt))
    loss.backward()
    return out.detach(), {k: p.grad.detach().double() for k, p in zip(onet.names, onet.params)}, taps


o64, g64, taps64 = oracle(torch.float64)
o32, g32, _ = oracle(torch.float32)
net = net.to(dev)
out = net(z.to(dev))
loss = torch.nn.functional.mse_loss(out, target.to(dev))
loss.backward()
torch.cuda.synchronize()
rows = []
for k, p in net.named_parameters():
    g = p.grad.detach().cpu().double()
    eh, er, n = (g - g64[k]).norm().item(), (g32[k] - g64[k]).norm().item(), g64[k].norm().item()
    rows.append((eh / (4 * er + 2e-5 * n + 1e-12), k, eh, er, n))
rows.sort(reverse=True)
print("worst 25 tensors: err/tol, name, err_hip, err_ref32, |g64|")
for r in rows[:25]:
    print("  %8.2f %-32s %.3e %.3e %.3e" % r)

eng = net.__dict__["_dip_engine"]
s0 = eng.sc[0]
H, W = hw


def nhwc(buf, C, Hh=H, Ww=W):
    Cs = (C + 3) // 4 * 4
    return buf.view(Hh, Ww, Cs)[:, :, :C].permute(2, 0, 1).cpu().double()


# --- up1 conv (6.1) weight gradient recomputed in fp64 from the engine's own u and dy buffers
u = s0.st["u"]
st = u.bn.state.view(4, u.Cs).cpu().double()
uy = nhwc(u.buf, u.C)
uact = st[2, :u.C].view(-1, 1, 1) * uy + st[3, :u.C].view(-1, 1, 1)
uact = torch.maximum(uact, 0.2 * uact)
dy = nhwc(s0.dbg["dy_last"], s0.up1.Cout)
dW = torch.einsum("ohw,chw->oc", dy, uact)
gW = dict(net.named_parameters())["6.1.weight"].grad.detach().cpu().double()[:, :, 0, 0]
print("6.1.weight: |hip - fp64(from own buffers)| / |g| = %.3e ;  |hip - oracle64| / |g| = %.3e" %
      ((gW - dW).norm() / dW.norm(), (gW - g64["6.1.weight"][:, :, 0, 0]).norm() / dW.norm()))
# is dy (grad wrt the raw up1 conv output) itself right?  oracle tap: none for up1; compare dy_u vs oracle "up0_raw".grad
dyu = nhwc(s0.dbg["dy_u"], s0.up.Cout)
ref = taps64["up0_raw"].grad[0]
print("dy wrt up0_raw: rel err vs oracle64 = %.3e" % ((dyu - ref).norm() / ref.norm()))
dcat = nhwc(s0.dbg["dcat"], s0.up.Cin)
refc = taps64["cat0"].grad[0]
print("dcat0: rel err vs oracle64 = %.3e" % ((dcat - refc).norm() / refc.norm()))
