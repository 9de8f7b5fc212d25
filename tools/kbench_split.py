"""Split-bf16 NT product against the fp32 MFMA GEMM: speed and error (vs float64 on the host for a sample)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops
def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dev = torch.device("cuda")
for (M, N, K) in [(24640, 512, 512), (24640, 2048, 512), (24640, 512, 2048), (10300, 512, 512), (10300, 1536, 512)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    ref64 = (A[:512].double() @ W.double().t() + b.double())
    f = ops.gemm(A, 0, W, 0, M, N, K, bias=b)
    line = f"M={M} N={N} K={K}: fp32 {timed(lambda: ops.gemm(A, 0, W, 0, M, N, K, bias=b)):7.1f} us (err {float((f[:512].double() - ref64).abs().max() / ref64.abs().max()):.1e})"
    for np_ in (2, 3):
        Wp = ops.split_bf16(W, np_)
        o = ops.gemm_split(A, Wp, bias=b)
        t = timed(lambda: ops.gemm_split(A, Wp, bias=b))
        line += f" | x{np_}: {t:7.1f} us ({2.0 * M * N * K / t / 1e6:6.1f} TF-eq, err {float((o[:512].double() - ref64).abs().max() / ref64.abs().max()):.1e})"
    print(line)
