"""Text-tower shapes of the learnable-prompt step (about 10 k live rows): mil_gemm as dispatched (64-row tiles where they
fill the chip better) against a build without them (tools/variants/lib_DLG_NO_TILE64.so via MIL_HIP_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
M=10300
for (N,K) in [(512,512),(512,2048),(1536,512),(2048,512)]:
    A=torch.randn((M,K),device="cuda"); B=torch.randn((K,N),device="cuda"); Bt=B.t().contiguous()
    t=timed(lambda: ops.gemm(A,0,B,1,M,N,K)); t2=timed(lambda: ops.gemm(A,0,Bt,0,M,N,K))
    print(f"M={M} N={N} K={K}: NN {t:.1f} us {2*M*N*K/t/1e6:.1f} TF   NT {t2:.1f} us {2*M*N*K/t2/1e6:.1f} TF")
