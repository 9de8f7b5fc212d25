"""Mid-size linear layers (65..1000 rows): one-launch kernels of csrc/mid_linear.hip against the tiled GEMM path."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops, _lib
def timed(fn, iters=50):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
L=_lib.lib(); P=ops._p; S=ops._stream
for (M,N,K) in [(320,512,512),(320,2048,512),(320,512,2048),(320,256,512),(770,512,512),(770,1536,512),(770,2048,512),(770,512,2048)]:
    x=torch.randn((M,K),device="cuda"); W=torch.randn((N,K),device="cuda")/K**0.5; b=torch.randn(N,device="cuda"); y=torch.empty((M,N),device="cuda")
    dy=torch.randn((M,N),device="cuda"); dx=torch.empty((M,K),device="cuda"); dW=torch.empty((N,K),device="cuda"); db=torch.empty(N,device="cuda")
    def fwd_mid(): assert L.mil_linear_mid_fwd(P(x),K,P(W),K,P(b),1,None,0,P(y),N,M,N,K,S())==0
    def fwd_old(): ops.gemm(x,0,W,0,M,N,K,bias=b,act=1)
    def bwd_mid(): assert L.mil_linear_mid_bwd(P(dy),N,P(y),N,1,P(x),K,P(W),K,P(dx),K,P(dW),K,P(db),M,N,K,S())==0
    def bwd_old():
        dpre=ops.act_bwd(dy,y,1); ops.gemm(dpre,0,W,1,M,K,N); ops.linear_bwd_params(dpre,None,0,x)
    fwd_mid(); ref=torch.tanh(x@W.t()+b); e=float((y-ref).abs().max())
    bwd_mid(); dp=dy*(1-y*y); e2=max(float((dx-dp@W).abs().max()), float((dW-dp.t()@x).abs().max()), float((db-dp.sum(0)).abs().max()))
    print(f"M={M} N={N} K={K}: fwd mid {timed(fwd_mid):6.1f} us  tiled {timed(fwd_old):6.1f} us | bwd mid {timed(bwd_mid):6.1f} us  tiled {timed(bwd_old):6.1f} us | err {e:.1e} {e2:.1e}")
