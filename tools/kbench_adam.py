import os, sys, torch
sys.path.insert(0, os.getcwd())
import mil_amd
from mil_amd import ops
n=9_800_000
p=torch.randn(n,device="cuda"); g=torch.randn(n,device="cuda"); m=torch.zeros(n,device="cuda"); v=torch.zeros(n,device="cuda")
def run(): ops.adam_step(p,g,m,v,3)
for _ in range(3): run()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); e1.synchronize()
t=e0.elapsed_time(e1)/50*1e3
print(f"adam {n} params: {t:.1f} us  {n*28/t/1e6:.2f} TB/s")
