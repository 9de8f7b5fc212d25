"""fc_pathology at the capacity buckets of the one-bag regime (rows x 768 -> 512, tanh): mil_gemm as dispatched WITH a device
row count (the bucket form: 64 x 128 tiles) and without one (the exact-shape form: 256 x 256 LDS-DMA tiles where they fill the
chip), and the weight gradient (mil_linear_bwd_params).  HIP events, 20 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa: F401
from mil_amd import ops


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


N, K = 512, 768
for M in (2048, 3072, 4096, 6144, 8192, 12288, 16384):
    A = torch.randn((M, K), device="cuda")
    W = torch.randn((N, K), device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    dy = torch.randn((M, N), device="cuda")
    rows = torch.tensor([M - 100], device="cuda", dtype=torch.int32)
    t_rows = timed(lambda: ops.gemm(A, 0, W, 0, M, N, K, bias=b, act=1, rows_dev=rows))
    t_exact = timed(lambda: ops.gemm(A, 0, W, 0, M, N, K, bias=b, act=1))
    y = ops.gemm(A, 0, W, 0, M, N, K, bias=b, act=1)
    t_dw = timed(lambda: ops.linear_bwd_params(dy, y, 1, A, None, None, True))
    fl = 2.0 * M * N * K
    print(f"rows {M:6d}: fwd bucket form {t_rows:6.1f} us ({fl / t_rows / 1e6:5.1f} TF)  exact form {t_exact:6.1f} us ({fl / t_exact / 1e6:5.1f} TF)"
          f"   dW {t_dw:6.1f} us ({fl / t_dw / 1e6:5.1f} TF)")
