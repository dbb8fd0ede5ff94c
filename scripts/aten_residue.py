"""Which ATen ops (and from where) still launch kernels inside one steady-state GAN step: torch.profiler over two steps,
grouped by op name with the Python call site (diagnostic for DESIGN.md section 1 / VERDICT r02 item 8)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
class A: pass
args = A(); args.patch_size = 48; args.num_channels = 256; args.num_blocks = 32; args.workload = "gan"; args.batch = 16; args.lr = 5e-7
dev = torch.device("cuda", 0)
trainer, G, D, vgg = bench.build(args, dev, 1)
lr, hr = bench.synth_batch(16, 48, 1234, dev)
for _ in range(3): trainer.gan_step(lr, hr)
torch.cuda.synchronize()
R = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(R): trainer.gan_step(lr, hr)
    torch.cuda.synchronize()
agg = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    kt = sum(k.duration for k in ev.kernels) if ev.kernels else 0
    if not ev.kernels:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "/repo/" in fr and "site-packages" not in fr and "scripts/aten_residue" not in fr:
            site = fr.split("/repo/")[-1]; break
    agg[(ev.name, site)] += len(ev.kernels); dur[(ev.name, site)] += kt
print(f"ATen ops that launched kernels, per step (over {R} steps): launches, total us, op @ first repo frame")
for k, n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n / R:7.1f} {dur[k] / R:9.1f} us  {k[0]:28s} @ {k[1]}")
print("total launches/step", sum(agg.values()) / R, " us/step", sum(dur.values()) / R)
