"""Host (Python + launch) time of one GAN step vs its GPU time: enqueue 10 steps without synchronising."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
class A: pass
args = A(); args.patch_size = 48; args.num_channels = 256; args.num_blocks = 32; args.workload = sys.argv[1] if len(sys.argv) > 1 else "gan"; args.batch = 16; args.lr = 5e-7
dev = torch.device("cuda", 0)
trainer, G, D, vgg = bench.build(args, dev, 1)
lr, hr = bench.synth_batch(16, 48, 1234, dev)
step = trainer.gan_step if args.workload == "gan" else trainer.pretrain_step
for _ in range(3): step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step(lr, hr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{args.workload}: host enqueue {1e3 * (t1 - t0) / 10:.1f} ms/step, until GPU idle {1e3 * (t2 - t0) / 10:.1f} ms/step")
