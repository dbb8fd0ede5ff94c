"""Diagnostic: one GAN step (C=256, depth 2, 48->192, B=2) in fp32 and bf16 modes against both oracles; prints every loss."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import dis_sd, gen_sd, vgg_sd
from model import Discriminator, Generator, VGG
from oracle import bf16 as OB, detrand, step as OS
from pesr_amd import ops
from pesr_amd.optim import FlatAdam
from pesr_amd.step import Trainer
warnings.filterwarnings("ignore")
C, depth, ps, B = 256, 2, 48, int(os.environ.get("B", "2"))
g_sd, d_sd, v_sd = gen_sd(C, depth), dis_sd(ps), vgg_sd()
lr = detrand.image_batch((B, 3, ps, ps), 700); hr = detrand.image_batch((B, 3, 4 * ps, 4 * ps), 701)
cfg = {"depth": depth, "res_scale": 0.1, "learning_rate": 5e-5}
def oracle(b16, dtype=torch.float32):
    cv = lambda sd: {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    st = OS.TrainState(cv(g_sd), cv(d_sd), cv(v_sd), cfg)
    with OB.enabled(b16, 1):
        return OS.gan_step(st, lr.to(dtype), hr.to(dtype))
def gpu(prec):
    ops.set_precision(prec); ops.BF16_MIN_WGS = 1
    G = Generator({"num_channels": C, "depth": depth, "res_scale": 0.1}); G.load_state_dict(g_sd); G.cuda()
    D = Discriminator({"patch_size": ps, "spectral_norm": False}); D.load_state_dict(d_sd); D.cuda()
    V = VGG(); V.load_state_dict(v_sd); V.cuda()
    tr = Trainer(G, D, V, FlatAdam(G.parameters(), lr=5e-5), FlatAdam(D.parameters(), lr=5e-5))
    log = tr.gan_step(lr.cuda(), hr.cuda())
    return {k: float(v) for k, v in log.items()}
r32, rb = oracle(False), oracle(True)
r64, rb64 = oracle(False, torch.float64), oracle(True, torch.float64)
g32, gb = gpu("fp32"), gpu("bf16")
for k in ("vgg", "g", "tv", "d"):
    print(f"{k:4s} oracle32 {r32[k]:.7f} oracle64 {r64[k]:.7f} gpu32 {g32[k]:.7f} | oracle_bf16 {rb[k]:.7f} oracle_bf16(64) {rb64[k]:.7f} gpu_bf16 {gb[k]:.7f}"
          f" | rel gpu32-o32 {abs(g32[k]-r32[k])/abs(r32[k]):.2e} o32-o64 {abs(r32[k]-r64[k])/abs(r64[k]):.2e}"
          f" gpub-ob {abs(gb[k]-rb[k])/abs(rb[k]):.2e} ob-ob64 {abs(rb[k]-rb64[k])/abs(rb64[k]):.2e}")
