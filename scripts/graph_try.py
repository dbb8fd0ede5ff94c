"""hipGraph capture of the GAN step: replayed steps vs eager steps from the same state (small config), then timing at the
benchmark configuration."""
import os, sys, time, copy, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model import Generator, Discriminator, VGG
from pesr_amd.optim import FlatAdam
from pesr_amd.step import Trainer

def build(opt, seed=0):
    torch.manual_seed(seed)
    dev = torch.device("cuda")
    G, D = Generator(opt).to(dev), Discriminator(opt).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore"); V = VGG().to(dev)
    oG = FlatAdam([p for p in G.parameters() if p.requires_grad], lr=5e-5)
    oD = FlatAdam(D.parameters(), lr=5e-5)
    return Trainer(G, D, V, oG, oD)

def batches(n, B, ps, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(B, 3, ps, ps, generator=g).mul(255).cuda(), torch.rand(B, 3, 4 * ps, 4 * ps, generator=g).mul(255).cuda()) for _ in range(n)]

def main():
    big = len(sys.argv) > 1 and sys.argv[1] == "big"
    opt = {"patch_size": 48 if big else 16, "num_channels": 256 if big else 64, "depth": 32 if big else 3, "res_scale": 0.1, "spectral_norm": False}
    B = 16 if big else 4
    data = batches(6, B, opt["patch_size"])
    ta, tb = build(opt), build(opt)
    for lr, hr in data[:2]:
        ta.gan_step(lr, hr); tb.gan_step(lr, hr)
    step = tb.capture_gan_step(*data[0])
    worst = 0.0
    for lr, hr in data[2:]:
        la = ta.gan_step(lr, hr)
        lb = step(lr, hr)
        for k in la:
            a, b = la[k].item(), lb[k].item()
            worst = max(worst, abs(a - b) / (abs(a) + 1e-12))
    pa = torch.cat([p.detach().flatten() for p in ta.G.parameters()]); pb = torch.cat([p.detach().flatten() for p in tb.G.parameters()])
    da = torch.cat([p.detach().flatten() for p in ta.D.parameters()]); db = torch.cat([p.detach().flatten() for p in tb.D.parameters()])
    print(f"4 replayed vs 4 eager steps: worst relative loss difference {worst:.3e}; max |dG param| {(pa - pb).abs().max().item():.3e}, max |dD param| {(da - db).abs().max().item():.3e}; steps {tb.optim_G.steps} {ta.optim_G.steps}")
    for name, f in (("eager", lambda: ta.gan_step(*data[0])), ("graph", lambda: step(*data[0]))):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{name}: {dt * 1e3:.2f} ms/step")

if __name__ == "__main__":
    main()
