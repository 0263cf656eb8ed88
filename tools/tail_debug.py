"""Debug aid: which rows / taps of the weight gradient of a 132-channel layer differ from fp64 (the streaming tail kernel)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import torch.nn.functional as F
import conftest  # noqa: F401  (paths)
import hipops as H
import dip_native as N

dev = torch.device("cuda:0")
Cin, Cout, Hh, Ww = 132, 128, 128, 128
g = torch.Generator().manual_seed(1)
x = torch.randn(1, Cin, Hh, Ww, generator=g)
dy = torch.randn(1, Cout, Hh, Ww, generator=g)
xp = F.pad(x.double(), (1, 1, 1, 1), mode="reflect")
w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
(F.conv2d(xp, w) * dy.double()).sum().backward()
ref = w.grad
dw, db = H.conv_wgrad(x.to(dev), dy.to(dev), 3, 1, N.PAD_REFLECT, (None, None, 1.0), nsplit="plan")
err = (dw.cpu().double() - ref).abs()
print("main rows max err", err[:, :128].max().item(), "tail rows max err", err[:, 128:].max().item(), "scale", ref.abs().max().item())
for c in range(128, 132):
    print("c", c, "per-tap max err", [f"{err[:, c, t // 3, t % 3].max().item():.2e}" for t in range(9)])
print("per-column-block max err (tail rows)", [f"{err[o:o + 32, 128:].max().item():.2e}" for o in range(0, 128, 32)])
