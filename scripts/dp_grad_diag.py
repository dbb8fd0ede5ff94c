"""How far the first-step GAN gradients of the small DP-test configuration move between kernel dispatches (diagnostic for the
tolerance model of tests/test_dp_gpu.py): error vs the fp32 CPU oracle per dispatch, for the worst tensors."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
warnings.filterwarnings("ignore")
from helpers import dis_sd, gen_sd, vgg_sd
from model import Discriminator, Generator, VGG
from oracle import detrand, step as OS
from pesr_amd import ops
from pesr_amd.optim import FlatAdam
from pesr_amd.step import Trainer

C, depth, ps, B = 64, 2, 24, 4

def oracle_grads(seed):
    lr = detrand.image_batch((B, 3, ps, ps), seed); hr = detrand.image_batch((B, 3, 4 * ps, 4 * ps), seed + 100)
    st = OS.TrainState(gen_sd(C, depth), dis_sd(ps), vgg_sd(), {"depth": depth, "res_scale": 0.1, "learning_rate": 5e-5, "dp_replicas": 1})
    OS.gan_step(st, lr, hr)
    ref = {("G", k): v.grad.clone() for k, v in st.g.items() if v.grad is not None}
    ref.update({("D", k): v.grad.clone() for k, v in st.d.items() if v.grad is not None})
    return lr, hr, ref

def run(lr, hr, ref, **flags):
    for k, v in flags.items(): setattr(ops, k, v)
    dev = torch.device("cuda")
    G = Generator({"num_channels": C, "depth": depth, "res_scale": 0.1}); G.load_state_dict(gen_sd(C, depth)); G.to(dev)
    D = Discriminator({"patch_size": ps, "spectral_norm": False}); D.load_state_dict(dis_sd(ps)); D.to(dev)
    V = VGG(); V.load_state_dict(vgg_sd()); V.to(dev)
    oG, oD = FlatAdam(G.parameters(), lr=5e-5), FlatAdam(D.parameters(), lr=5e-5)
    tr = Trainer(G, D, V, oG, oD)
    tr.gan_step(lr.to(dev), hr.to(dev))
    out = {}
    for name, net in (("G", G), ("D", D)):
        for k, p in net.named_parameters():
            if (name, k) in ref and p.grad is not None:
                mx = float(ref[(name, k)].abs().max())
                if mx > 0: out[(name, k)] = float((p.grad.cpu() - ref[(name, k)]).abs().max()) / mx
    for k in flags: setattr(ops, k, True)
    return out

print("worst per-tensor gradient error vs the fp32 oracle (fraction of the tensor's maximum), per batch and dispatch")
for seed in (700, 710, 720, 730, 740, 750):
    lr, hr, ref = oracle_grads(seed)
    row = {}
    for name, fl in (("default", {}), ("no rgb_out", {"USE_RGB_OUT": False}), ("no wino4", {"USE_WINO4": False}), ("direct only", {"USE_WINO": False})):
        r = run(lr, hr, ref, **fl)
        k = max(r, key=r.get)
        row[name] = (r[k], k[0] + "." + k[1])
    print(seed, "   ".join(f"{n}: {v[0]:.2e} ({v[1]})" for n, v in row.items()))
