"""Per-phase GPU time of one GAN step (events on the compute stream)."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from model import Generator, Discriminator, VGG
from pesr_amd import functional as PF
from pesr_amd.model.basic import nhwc
from pesr_amd.optim import FlatAdam
from pesr_amd.model.focal_loss import FocalLoss
torch.manual_seed(0)
dev = torch.device("cuda")
opt = {"patch_size": 48, "num_channels": 256, "depth": 32, "res_scale": 0.1, "spectral_norm": False}
G, D = Generator(opt).to(dev), Discriminator(opt).to(dev)
with warnings.catch_warnings():
    warnings.simplefilter("ignore"); V = VGG().to(dev)
oG, oD = FlatAdam(G.parameters(), lr=5e-5), FlatAdam(D.parameters(), lr=5e-5)
lr = torch.randint(0, 256, (16, 3, 48, 48)).float().to(dev)
hr = torch.randint(0, 256, (16, 3, 192, 192)).float().to(dev).contiguous(memory_format=torch.channels_last)
ones = torch.ones(16, 1, device=dev)
fl = FocalLoss(1.0)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
GF = {"D fwd (hr)": 113.2, "G fwd": 3705, "D fwd (sr.detach)": 113.2, "D bwd (dgrad+wgrad x2)": 449, "Adam D": 0, "D fwd (sr)": 113.2,
      "D fwd (hr) nograd": 113.2, "VGG fwd x2": 917, "losses": 0, "bwd: D dgrad + VGG dgrad": 572, "bwd: G": 7410, "Adam G": 0}
def step(timed):
    global marks
    marks = []
    mark("start")
    for p in D.parameters(): p.requires_grad = True
    oD.zero_grad()
    pr = D(hr); mark("D fwd (hr)")
    sr = G(lr); mark("G fwd")
    pf = D(sr.detach()); mark("D fwd (sr.detach)")
    dl = F.binary_cross_entropy_with_logits(pr - pf, ones)
    dl.backward(); mark("D bwd (dgrad+wgrad x2)")
    oD.step(); mark("Adam D")
    for p in D.parameters(): p.requires_grad = False
    oG.zero_grad()
    pf = D(sr); mark("D fwd (sr)")
    with torch.no_grad(): pr = D(hr)
    mark("D fwd (hr) nograd")
    fs, fh = V(sr, hr); mark("VGG fwd x2")
    vl = PF.mse_loss(nhwc(fs), nhwc(fh)) * 50; tv = PF.tv_loss(nhwc(sr)) * 1e-6; l1 = PF.l1_loss(nhwc(sr), nhwc(hr)) * 0
    gl = fl(pf - pr, ones); tot = l1 + vl + gl + tv; mark("losses")
    # split backward: first to sr, then through G
    (gsr,) = torch.autograd.grad(tot, sr, retain_graph=False); mark("bwd: D dgrad + VGG dgrad")
    sr.backward(gsr); mark("bwd: G")
    oG.step(); mark("Adam G")
for _ in range(3): step(False)
torch.cuda.synchronize()
acc = {}
R = 5
for _ in range(R):
    step(True); torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0) + e0.elapsed_time(e1)
tot = 0
for n, t in acc.items():
    t /= R; tot += t
    gf = GF.get(n, 0)
    print(f"{n:28s} {t:8.2f} ms  {gf/t if t else 0:7.1f} TF/s" if gf else f"{n:28s} {t:8.2f} ms")
print(f"{'total':28s} {tot:8.2f} ms")
