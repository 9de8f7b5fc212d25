"""Practical HBM read ceiling on this box: torch reductions / copies over 512 MiB (twice the Infinity Cache) next to the
attention-pool pass over the same bytes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops
from mil_amd.bags import BagLayout
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
B,N,L=64,4096,512
x=torch.randn(B*N,L,device="cuda"); scores=torch.randn(B*N,device="cuda"); lay=BagLayout.uniform(B,N,x.device)
nbytes=x.numel()*4
t=timed(lambda: x.sum()); print(f"torch sum      {t:7.1f} us {nbytes/t/1e6:.2f} TB/s")
t=timed(lambda: x.max()); print(f"torch max      {t:7.1f} us {nbytes/t/1e6:.2f} TB/s")
y=torch.empty_like(x)
t=timed(lambda: y.copy_(x)); print(f"torch copy     {t:7.1f} us {2*nbytes/t/1e6:.2f} TB/s (read+write)")
t=timed(lambda: ops.attn_pool_fwd(x,scores,lay)); print(f"attn_pool_fwd  {t:7.1f} us {nbytes/t/1e6:.2f} TB/s")
t=timed(lambda: ops.attn_pool_partial(x,scores,lay)); print(f"pool_partial   {t:7.1f} us {nbytes/t/1e6:.2f} TB/s")
