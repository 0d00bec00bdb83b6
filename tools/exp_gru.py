import sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn as nn
from ratrack_amd.train_ops import gru_step
gru = nn.GRU(128,128,5).cuda()
x = torch.randn(64,128,device="cuda",requires_grad=True); h = torch.randn(5,64,128,device="cuda",requires_grad=True)
from torch.profiler import profile, ProfilerActivity
for _ in range(3):
    y,h1 = gru_step(x,h,gru); (y.sum()+h1.sum()).backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        y,h1 = gru_step(x,h,gru); (y.sum()+h1.sum()).backward()
    torch.cuda.synchronize()
for e in prof.key_averages():
    if "gru" in e.key: print(e.key[:60], e.device_time_total/e.count)
