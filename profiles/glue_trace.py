"""Which small launches (torch elementwise kernels, fills, memcpys) does one training step enqueue besides the lp_* kernels, and from where?
torch.profiler with Python stacks around one step of the bench's model; groups device events by the innermost frame inside the package.
    python profiles/glue_trace.py > gpurun_out/r03_glue_trace.txt"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lightning_pose_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev, 17, 384)
batch = bench.synth_batch(dev, 0, 384, 64, 128, 17)
tr = Trainer(max_epochs=1, data_parallel=False)
tr.setup(model)
model.train()
model.total_unsupervised_importance = torch.tensor(1.0)
for i in range(3):
    tr.training_batch(model, batch, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.training_batch(model, batch, 3)
    torch.cuda.synchronize()
ev = prof.events()
by = collections.Counter()
dur = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CUDA:
        continue
    name = e.name
    if name.startswith("void lp::") or name.startswith("lp::"):
        continue
    by[(name[:70],)] += 1
    dur[(name[:70],)] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
print("device events outside lp:: kernels in one step")
for k, n in by.most_common():
    print(f"{n:4d} {dur[k]:9.1f} us  {k[0]}")
print()
print("CPU-side ops that launched them (op name, innermost package frame), by count:")
sites = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        continue
    if not e.name.startswith("aten::") and "Memcpy" not in e.name and "memcpy" not in e.name:
        continue
    kt = sum(1 for k in e.kernels) if getattr(e, "kernels", None) else 0
    if kt == 0:
        continue
    if e.cpu_children and any(getattr(c, "kernels", None) for c in e.cpu_children):
        continue   # count the innermost launching op only
    site = "?"
    for fr in (e.stack or []):
        if "lightning-pose_amd" in fr or "lightning_pose_amd" in fr:
            site = fr.split("lightning-pose_amd/")[-1][:90]
            break
    sites[(e.name, site)] += kt
for (name, site), n in sites.most_common(60):
    print(f"{n:4d} {name:34s} {site}")
