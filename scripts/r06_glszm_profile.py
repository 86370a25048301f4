import sys; sys.path.insert(0, "/root/repo")
import torch, bench
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
kind = sys.argv[1]
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
l2, m2 = bench.make_volume(size, 32, kind, 3, dev)
n = int(m2.sum().item())
for _ in range(6):
    engine.glszm_compact(l2, m2, 32, n)
torch.cuda.synchronize()
