import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from probaforms_amd import _engine
for _ in range(5):
    p = _engine.DeviceShuffle.draw(1000000, 5, torch.device("cuda:0"))
torch.cuda.synchronize()
print(int(p[0]))
