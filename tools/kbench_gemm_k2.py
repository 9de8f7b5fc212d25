"""Fixed cost of a k_gemm launch: time against K (down to one slice) and against the number of rounds (M)."""
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
N=512
for M in (16384, 32768, 65536):
    for K in (32, 64, 128, 384, 768):
        A=torch.randn((M,K),device="cuda"); Bt=torch.randn((N,K),device="cuda"); C=torch.zeros((M,N),device="cuda")
        t=timed(lambda: ops.gemm(A,0,Bt,0,M,N,K,out=C,split_k=False))
        print(f"M={M} ({M//128*4//512} rounds) K={K}: NT {t:.1f} us {2*M*N*K/t/1e6:.1f} TF")
