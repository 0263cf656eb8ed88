"""Host-CPU thread sweep of the CPU oracle (default net, 256x256) to pick a sane thread count for
bench.py's cpu_baseline on many-core GPU hosts (all 256 hardware threads is far from the best)."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.add_to_path()
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dip_oracle as O
from models import get_net

torch.manual_seed(0)
net = get_net(32, 'skip', 'reflection', skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5, upsample_mode='bilinear')
sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in O.param_shapes(O.default_spec())}
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
z = torch.rand(1, 32, size, size) * 0.1
t = torch.rand(1, 3, size, size)
res = {}
for nt in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "8,16,32,64,128,256".split(","))]:
    if nt > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(nt)
    onet = O.OracleNet(O.default_spec(), sd)
    opt = torch.optim.Adam(onet.params, lr=0.01)
    ts = []
    for it in range(4):
        t0 = time.time()
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(onet(z), t)
        loss.backward()
        opt.step()
        ts.append(time.time() - t0)
    res[nt] = min(ts[1:])
    print(f"threads {nt:4d}: {res[nt]*1e3:9.1f} ms/iter  ({1/res[nt]:.3f} it/s) at {size}x{size}", flush=True)
print(json.dumps(res))
