"""Where the host's time per eager GAN step goes: cProfile over a few steps enqueued without waiting for the GPU
(top functions by own time).  PESR_FORCE_DP=1 + a 1-rank group adds the data-parallel hooks."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
class A: pass
args = A(); args.patch_size = 48; args.num_channels = 256; args.num_blocks = 32; args.workload = "gan"; args.batch = 16; args.lr = 5e-7
dev = torch.device("cuda", 0)
if os.environ.get("PESR_FORCE_DP") == "1":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
trainer, G, D, vgg = bench.build(args, dev, 1)
lr, hr = bench.synth_batch(16, 48, 1234, dev)
for _ in range(3): trainer.gan_step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): trainer.gan_step(lr, hr)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue {1e3 * (t1 - t0) / 5:.1f} ms/step, until GPU idle {1e3 * (time.perf_counter() - t0) / 5:.1f} ms/step", flush=True)
torch.autograd.set_multithreading_enabled(False)      # profiling only: the backward pass runs on this thread, where cProfile sees it
for _ in range(2): trainer.gan_step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): trainer.gan_step(lr, hr)
print(f"single-threaded autograd: host enqueue {1e3 * (time.perf_counter() - t0) / 5:.1f} ms/step", flush=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3): trainer.gan_step(lr, hr)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
