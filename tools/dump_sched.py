import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
ge.build()
from models.skip import skip
from utils.loss_head import MSEHead
dev = torch.device("cuda:0")
kw = dict(num_channels_down=[128, 128], num_channels_up=[128, 128], num_channels_skip=[4, 4], upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")
torch.manual_seed(6)
net = skip(4, 3, **kw).to(dev)
z = (torch.rand(1, 4, 256, 256) * 0.1).to(dev); target = torch.rand(1, 3, 256, 256).to(dev)
head = MSEHead(net, target)
loss, _ = head(z); loss.backward(); torch.cuda.synchronize()
eng = net.__dict__["_dip_engine"]
ops = eng.bwd_ops
deps = eng._backward_deps(ops)
bulk2 = eng._bulk2
cls = eng._BWD_SIDE if not bulk2 else (lambda n: 3 if n in bulk2 else eng._BWD_SIDE(n))
sched = eng._schedule(ops, cls, lambda n: False, deps)
print("bulk2:", sorted(bulk2)[:6], "bulk2_max_pixels", eng.bulk2_max_pixels)
for c in sched:
    if c[0] == "launch": print(f"  L s{c[2]} {ops[c[1]][2]}")
    elif c[0] == "record": print(f"  R s{c[2]} {c[1]}")
    else: print(f"  W s{c[1]} <- {c[2]}")
