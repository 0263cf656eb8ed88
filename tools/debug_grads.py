"""GPU debug: per-tensor gradient error table vs the fp64 oracle, and isolation of individual
backward kernels by recomputing them in fp64 from the engine's OWN buffers."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tools/: a developer diagnostic that checks the HIP path against the oracle, like the tests do)
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

# ---- stage-by-stage check of the top of the backward pass from the engine's own buffers
top = eng.dbg_top
last = top["last"]
C_ = last.C
gout = (2.0 * (out.detach() - target.to(dev)) / out.numel()).cpu().double()[0]
o_ = out.detach().cpu().double()[0]
dyo_ref = gout * o_ * (1 - o_)
dyo = nhwc(eng.dy_out, 3)
print("dy_out rel err: %.3e" % ((dyo - dyo_ref).norm() / dyo_ref.norm()))
Wout = dict(net.named_parameters())["9.1.weight"].detach().cpu().double()[:, :, 0, 0]     # [3,128]
du_ref = torch.einsum("ohw,oc->chw", dyo, Wout)
du = nhwc(top["du_last"][0], C_)
print("du_last (dgrad of out conv) rel err: %.3e" % ((du - du_ref).norm() / du_ref.norm()))
st1 = last.bn.state.view(4, last.Cs).cpu().double()
y1 = nhwc(last.buf, C_)
mean_t = y1.reshape(C_, -1).mean(1); var_t = y1.reshape(C_, -1).var(1, unbiased=False)
print("state mean err %.3e  rstd rel err %.3e" % ((st1[0, :C_] - mean_t).abs().max(), ((st1[1, :C_] - 1 / torch.sqrt(var_t + 1e-5)) * torch.sqrt(var_t + 1e-5)).abs().max()))
xh = (y1 - st1[0, :C_].view(-1, 1, 1)) * st1[1, :C_].view(-1, 1, 1)
zz = st1[2, :C_].view(-1, 1, 1) * y1 + st1[3, :C_].view(-1, 1, 1)
dz = torch.where(zz > 0, du, 0.2 * du)
N_ = H * W
k1 = dz.reshape(C_, -1).sum(1) / N_
k2 = (dz * xh).reshape(C_, -1).sum(1) / N_
coef = last.bn.coef.view(2, last.Cs).cpu().double()
print("coef k1 rel err %.3e  k2 rel err %.3e" % (((coef[0, :C_] - k1).norm() / k1.norm()), ((coef[1, :C_] - k2).norm() / k2.norm())))
dy_ref = st1[2, :C_].view(-1, 1, 1) * (dz - k1.view(-1, 1, 1) - xh * k2.view(-1, 1, 1))
dyl = nhwc(top["dy_last"], C_)
print("dy_last rel err vs fp64 from own buffers: %.3e" % ((dyl - dy_ref).norm() / dy_ref.norm()))
gam = dict(net.named_parameters())["7.weight"]
print("dgamma rel err %.3e dbeta rel err %.3e" % (((gam.grad.cpu().double() - k2 * N_).norm() / (k2 * N_).norm()),
      ((dict(net.named_parameters())["7.bias"].grad.cpu().double() - k1 * N_).norm() / (k1 * N_).norm())))
print("dgamma vs oracle64 rel %.3e ; own-buffer dgamma vs oracle64 rel %.3e" % (
      ((gam.grad.cpu().double() - g64["7.weight"]).norm() / g64["7.weight"].norm()), ((k2 * N_ - g64["7.weight"]).norm() / g64["7.weight"].norm())))

# ---- forward intermediates vs the fp64 oracle
uy_ref = taps64["up0_raw"].detach()[0]
print("forward up0_raw (conv 3.1 output incl. bias): rel err %.3e, max abs %.3e" % ((uy - uy_ref).norm() / uy_ref.norm(), (uy - uy_ref).abs().max()))
pc = ((uy - uy_ref).reshape(uy.shape[0], -1).mean(1))
print("   per-channel mean of the error: max |.| %.3e  (per-channel std of y: %.3e)" % (pc.abs().max(), uy_ref.reshape(uy.shape[0], -1).std(1).mean()))
cat_ref = taps64["cat0"].detach()[0]
catb = nhwc(s0.st["cat"], s0.up.Cin)
print("forward cat0: rel err %.3e" % ((catb - cat_ref).norm() / cat_ref.norm()))
uact_ref = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(
    taps64["up0_raw"].detach(), None, None, sd["4.weight"].double(), sd["4.bias"].double(), True, 0.1, 1e-5), 0.2)[0]
print("forward u = lrelu(bn(up0_raw)): rel err %.3e" % ((uact - uact_ref).norm() / uact_ref.norm()))
dW_mixed = torch.einsum("ohw,chw->oc", dy, uact_ref)
print("dW(6.1) with OUR dy and ORACLE u: rel err vs oracle64 %.3e" % ((dW_mixed - g64["6.1.weight"][:, :, 0, 0]).norm() / dW.norm()))

# ---- oracle top-of-net replicated in fp64 to get the gradient wrt the raw output of conv 6.1
import torch.nn.functional as F
sdd = {k: v.double() for k, v in sd.items()}
u_o = F.leaky_relu(F.batch_norm(taps64["up0_raw"].detach(), None, None, sdd["4.weight"], sdd["4.bias"], True, 0.1, 1e-5), 0.2)
y61 = F.conv2d(u_o, sdd["6.1.weight"], sdd["6.1.bias"]).detach().requires_grad_(True)
act = F.leaky_relu(F.batch_norm(y61, None, None, sdd["7.weight"], sdd["7.bias"], True, 0.1, 1e-5), 0.2)
act.retain_grad()
o_ = torch.sigmoid(F.conv2d(act, sdd["9.1.weight"], sdd["9.1.bias"]))
F.mse_loss(o_, target.double()).backward()
dy_true = y61.grad[0]
print("check: einsum(dy_true, u_oracle) vs oracle grad 6.1.weight: %.3e" %
      ((torch.einsum("ohw,chw->oc", dy_true, u_o[0]) - g64["6.1.weight"][:, :, 0, 0]).norm() / g64["6.1.weight"].norm()))
print("our u1_y vs oracle y61: rel %.3e" % ((y1 - y61.detach()[0]).norm() / y61.detach().norm()))
print("our dy_last vs oracle dy: rel %.3e" % ((dyl - dy_true).norm() / dy_true.norm()))
print("our du_last vs oracle act.grad: rel %.3e" % ((du - act.grad[0]).norm() / act.grad.norm()))
print("our out vs oracle out: rel %.3e" % ((out.detach().cpu().double()[0] - o_.detach()[0]).norm() / o_.norm()))
e = (dyl - dy_true)
print("error structure: per-channel mean of err / rms(dy): %.3e ; corr(err, xhat) per channel max: %.3e" % (
      (e.reshape(C_, -1).mean(1).abs().max() / dy_true.pow(2).mean().sqrt()),
      ((e * xh).reshape(C_, -1).mean(1).abs().max() / dy_true.pow(2).mean().sqrt())))
