"""Which Python line issues the non-repo device launches of one iteration (hipMemcpy D2D = rocclr copyBuffer,
ATen fill / elementwise kernels)?  torch.profiler with stacks over 3 iterations of bench.py's Fit.

    python tools/find_copies.py [fused|notebook]            (GPU box)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
import bench  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fused"
    dev = torch.device("cuda:0")
    fit = bench.Fit("default", 0, dev, kind)
    for _ in range(3):
        fit.step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(3):
            fit.step()
        torch.cuda.synchronize()
    rows = {}
    for ev in prof.events():
        name = ev.name
        if not any(k in name for k in ("Memcpy", "memcpy", "copy_", "fill_", "zero_", "aten::add", "aten::mul", "aten::empty",
                                       "aten::zeros", "aten::ones")):
            continue
        stack = [s for s in (ev.stack or []) if "site-packages/torch" not in s and "<built-in" not in s][:4]
        key = (name, " <- ".join(stack))
        rows[key] = rows.get(key, 0) + 1
    for (name, stack), n in sorted(rows.items(), key=lambda kv: -kv[1])[:60]:
        print(f"{n / 3:6.1f}/it  {name:40s} {stack}")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25))


if __name__ == "__main__":
    main()
