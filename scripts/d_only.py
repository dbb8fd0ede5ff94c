import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model import Discriminator, VGG
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "D"
dev = torch.device("cuda")
hr = torch.randint(0, 256, (16, 3, 192, 192)).float().to(dev).contiguous(memory_format=torch.channels_last)
if which == "D":
    D = Discriminator({"patch_size": 48, "spectral_norm": False}).to(dev)
    for i in range(4):
        x = hr.clone().requires_grad_(True)
        out = D(x)
        if i >= 2: out.sum().backward()
else:
    with warnings.catch_warnings():
        warnings.simplefilter("ignore"); V = VGG().to(dev)
    for i in range(4):
        x = hr.clone().requires_grad_(True)
        fa, fb = V(x, hr)
        if i >= 2: fa.sum().backward()
torch.cuda.synchronize()
