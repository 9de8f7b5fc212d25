import torch, sys
sys.path.insert(0,__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import mil_amd
from mil_amd import ops
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
import glob, ctypes, os
from mil_amd import _lib
_lib.lib()
libs = [None] + sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "*.so")))
def load(path):
    h = ctypes.CDLL(path)
    for name, (res, a) in _lib.SIGNATURES.items():
        fn = getattr(h, name); fn.restype = res; fn.argtypes = a
    return h
main = _lib._lib
for q in libs:
  _lib._lib = main if q is None else load(q)
  print("==", "main" if q is None else os.path.basename(q))
  for (M,N,K) in [(32768,512,384),(32768,512,768),(32768,512,1536),(32768,768,384)]:
    A=torch.randn((M,K),device="cuda"); B=torch.randn((K,N),device="cuda"); Bt=B.t().contiguous(); C=torch.zeros((M,N),device="cuda")
    t=timed(lambda: ops.gemm(A,0,B,1,M,N,K,out=C,accumulate=True))
    t2=timed(lambda: ops.gemm(A,0,Bt,0,M,N,K,out=C))
    print("  " + f"M={M} N={N} K={K}: NN+acc {t:.1f} us {2*M*N*K/t/1e6:.1f} TF   NT {t2:.1f} us {2*M*N*K/t2/1e6:.1f} TF")
