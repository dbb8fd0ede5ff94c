import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model import Discriminator
torch.manual_seed(0)
dev = torch.device("cuda")
hr = torch.randint(0, 256, (16, 3, 192, 192)).float().to(dev).contiguous(memory_format=torch.channels_last)
D = Discriminator({"patch_size": 48, "spectral_norm": False}).to(dev)
with torch.no_grad():
    for i in range(3): D(hr)
torch.cuda.synchronize()
