"""Which weight packs / torch ops run per steady-state GAN step (diagnostic): counts ops.pack_* calls by shape and caller."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pesr_amd import ops
class A: pass
args = A(); args.patch_size = 48; args.num_channels = 256; args.num_blocks = 32; args.workload = "gan"; args.batch = 16; args.lr = 5e-7
dev = torch.device("cuda", 0)
trainer, G, D, vgg = bench.build(args, dev, 1)
lr, hr = bench.synth_batch(16, 48, 1234, dev)
for _ in range(2): trainer.gan_step(lr, hr)
cnt = collections.Counter()
def wrap(name):
    f = getattr(ops, name)
    def g(w, *a, **k):
        st = traceback.extract_stack(limit=8)
        who = " < ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in st[-5:-1])
        cnt[(name, tuple(w.shape), a, who)] += 1
        return f(w, *a, **k)
    setattr(ops, name, g)
for n in ("pack_conv3x3", "pack_conv3x3_wino", "pack_conv3x3_wino4", "pack_bias_ps"): wrap(n)
trainer.gan_step(lr, hr)
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]): print(v, k)
# torch-side kernels of one step
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    trainer.gan_step(lr, hr); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=60))
