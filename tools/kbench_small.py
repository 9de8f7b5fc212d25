"""Time the token-side linear kernels alone (HIP events, back-to-back launches)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd
from mil_amd import ops
dev = torch.device("cuda")
def t(fn, n=20, reps=20):
    """n launches captured in one hipGraph (no host launch cost in the timing), replayed reps times."""
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * reps) * 1e3
for (M, N, K) in [(32, 512, 512), (32, 2048, 512), (32, 512, 2048), (32, 256, 512), (64, 512, 512)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    y = ops.linear_small_fwd(x, W, b, 2)
    f = t(lambda: ops.linear_small_fwd(x, W, b, 2))
    bw = t(lambda: ops.linear_small_bwd(dy, y, 2, x, W, True, True, True))
    bw_nodx = t(lambda: ops.linear_small_bwd(dy, y, 2, x, W, False, True, True))
    bw_dxonly = t(lambda: ops.linear_small_bwd(dy, y, 2, x, W, True, False, False))
    print(f"M={M} N={N} K={K}: fwd {f:.1f} us  bwd {bw:.1f} us  (dW+db only {bw_nodx:.1f}, dx only {bw_dxonly:.1f})")
# power-of-two row stride or not?
from mil_amd import _lib
for (M, N, K, pad) in [(32, 512, 2048, 0), (32, 512, 2048, 32), (32, 512, 2048, 8), (32, 2048, 512, 0), (32, 2048, 512, 32)]:
    x = torch.randn(M, K, device=dev); Wb = torch.randn(N, K + pad, device=dev); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev)
    L = _lib.lib(); st = torch.cuda.current_stream
    def f(): L.mil_linear_small_fwd(x.data_ptr(), K, Wb.data_ptr(), K + pad, b.data_ptr(), 0, None, 0, y.data_ptr(), N, M, N, K, torch.cuda.current_stream().cuda_stream)
    def g(): L.mil_linear_small_bwd(dy.data_ptr(), N, None, 0, 0, x.data_ptr(), K, Wb.data_ptr(), K + pad, dx.data_ptr(), K, None, 0, None, M, N, K, torch.cuda.current_stream().cuda_stream)
    print(f"M={M} N={N} K={K} ldw={K + pad}: fwd {t(f):.1f} us, dx only {t(g):.1f} us")
