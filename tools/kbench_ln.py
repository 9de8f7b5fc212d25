"""LayerNorm backward (k_layernorm_bwd) at the row counts of the fusion paths."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops
def timed(fn, iters=30):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
E=512
for rows in (320, 770, 10240, 32800):
    for frozen in (False, True):
        x=torch.randn((rows,E),device="cuda"); g=torch.randn(E,device="cuda").requires_grad_(not frozen); b=torch.randn(E,device="cuda").requires_grad_(not frozen)
        dy=torch.randn((rows,E),device="cuda")
        stats=torch.empty((rows,2),device="cuda"); y=torch.empty_like(x)
        ops._lib.lib().mil_layernorm_fwd(ops._p(x),ops._p(g),ops._p(b),rows,E,1e-5,ops._p(y),ops._p(stats),ops._stream())
        t=timed(lambda: ops._layer_norm_bwd(x,g.detach(),stats,dy,None,not frozen))
        print(f"rows={rows:6d} params={'no ' if frozen else 'yes'}: {t:6.1f} us  {3*rows*E*4/t/1e6:.2f} TB/s")
