import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tools/: a developer diagnostic that checks the HIP path against the oracle, like the tests do)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
ge.build()
import dip_oracle as O, hipops
import torch.nn.functional as F
from models.skip import skip
dev = torch.device("cuda:0")
torch.manual_seed(123)
hw, mode, nskip = (64, 64), "nearest", 128
kw = dict(num_channels_down=[128] * 5, num_channels_up=[128] * 5, num_channels_skip=[nskip] * 5,
          upsample_mode=mode, need_sigmoid=True, need_bias=True, pad="reflection")
net = skip(32, 3, **kw)
sd = {k: v.detach().clone().double() for k, v in net.state_dict().items()
      if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
z = torch.rand(1, 32, *hw) * 0.1
spec = O.SkipSpec(32, 3, [128] * 5, [128] * 5, [nskip] * 5, pad="reflection", upsample_mode=mode)
# natural fp64 masks: hook _bn_act
nat = {}
orig = O._bn_act
def hooked(x, sd_, key, act=True, eps=1e-5, masks=None):
    y = F.batch_norm(x, None, None, sd_[key + ".weight"], sd_[key + ".bias"], True, 0.1, eps)
    if act:
        nat[key] = (y > 0)
        return F.leaky_relu(y, 0.2)
    return y
O._bn_act = hooked
O.skip_forward(spec, sd, z.double())
O._bn_act = orig
net = net.to(dev)
out = net(z.to(dev))
torch.cuda.synchronize()
m = hipops.lrelu_masks(net, spec)
print("keys engine:", sorted(m)[:50])
print("keys oracle:", sorted(nat)[:50])
for k in sorted(nat):
    if k not in m:
        print("MISSING", k); continue
    a, b = m[k], nat[k]
    print(k, tuple(a.shape), tuple(b.shape), "mismatches:", int((a != b).sum()) if a.shape == b.shape else "shape!")

# does the backward pass disturb anything the mask extraction reads?
eng = net.__dict__["_dip_engine"]
snap = {}
for i, s in enumerate(eng.sc):
    for name in ("s_act", "d1", "d2", "u", "u1"):
        a = s.st.get(name)
        if a is not None:
            snap[(i, name)] = (a.buf.clone(), a.bn.state.clone())
target = torch.rand(1, 3, *hw).to(dev)
loss = F.mse_loss(out, target)
loss.backward()
torch.cuda.synchronize()
for (i, name), (b0, s0_) in snap.items():
    a = eng.sc[i].st[name]
    db, ds = (a.buf - b0).abs().max().item(), (a.bn.state - s0_).abs().max().item()
    if db or ds:
        print("CHANGED by backward:", i, name, "buf maxdiff", db, "state maxdiff", ds)
m2 = hipops.lrelu_masks(net, spec)
for k in sorted(m):
    d = int((m[k] != m2[k]).sum())
    if d:
        print("mask changed after backward:", k, d)
print("post-backward check done")
