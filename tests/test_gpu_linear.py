"""GPU parity of the generic fp32-MFMA linear (K3a: fc_pathology, projections, MLPs) against torch CPU fp32."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from mil_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N,act", [(1, 512, 512, "tanh"), (10, 512, 256, "none"), (77, 512, 2048, "quickgelu"),
                                       (300, 768, 512, "tanh"), (1024, 512, 256, "none"), (130, 2048, 512, "none"),
                                       (64, 512, 2048, "relu"), (231, 64, 192, "none"), (9, 256, 64, "none")])
def test_linear_act_fwd_bwd(M, K, N, act):
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    go = torch.randn(M, N, generator=g)
    dev = torch.device("cuda")
    xd, Wd, bd = (t.to(dev).requires_grad_(True) for t in (x, W, b))
    y = ops.linear_act(xd, Wd, bd, act)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    pre = F.linear(xr, Wr, br)
    ref = {"none": pre, "tanh": torch.tanh(pre), "relu": torch.relu(pre),
           "quickgelu": pre * torch.sigmoid(1.702 * pre)}[act]
    assert rel_err(y.detach().cpu(), ref.detach()) <= 2e-6
    if act == "quickgelu":
        return                       # forward-only activation (frozen CLIP text tower)
    (y * go.to(dev)).sum().backward()
    (ref * go).sum().backward()
    assert rel_err(xd.grad.cpu(), xr.grad) <= 1e-5
    assert rel_err(Wd.grad.cpu(), Wr.grad) <= 1e-5
    assert rel_err(bd.grad.cpu(), br.grad) <= 1e-5


def test_linear_residual_and_accumulate():
    g = torch.Generator().manual_seed(5)
    x, W, r = torch.randn(50, 256, generator=g), torch.randn(512, 256, generator=g) / 16, torch.randn(50, 512, generator=g)
    dev = torch.device("cuda")
    xd, Wd, rd = (t.to(dev).requires_grad_(True) for t in (x, W, r))
    y = ops.linear_act(xd, Wd, None, "none", residual=rd)
    y.sum().backward()
    assert rel_err(y.detach().cpu(), x @ W.t() + r) <= 2e-6
    assert float((rd.grad.cpu() - 1).abs().max()) == 0.0
    out = torch.ones(50, 512, device=dev)
    ops.gemm(xd.detach(), 0, Wd.detach(), 0, 50, 512, 256, out=out, accumulate=True)
    assert rel_err(out.cpu(), x @ W.t() + 1) <= 2e-6


@pytest.mark.parametrize("b_mode", [0, 1])
def test_tall_gemm_with_a_split_last_round(b_mode):
    """M x N tiles = 592 = one round of 512 + 80: the last 20 row tiles run as a second launch with split-K and the
    epilogue in the reduce kernel; the result must not depend on that plan."""
    from mil_amd import ops, _lib
    M, N, K = 128 * 148 - 37, 512, 256
    assert _lib.lib().mil_gemm_workspace_floats(M, N, K, 0) > 0
    g = torch.Generator().manual_seed(b_mode)
    A = torch.randn((M, K), generator=g)
    B = torch.randn((N, K) if b_mode == 0 else (K, N), generator=g) / K ** 0.5
    bias = torch.randn((N,), generator=g)
    res = torch.randn((M, N), generator=g)
    ref = A @ (B.t() if b_mode == 0 else B) + bias + res
    out = ops.gemm(A.cuda(), 0, B.cuda(), b_mode, M, N, K, bias=bias.cuda(), act=0, residual=res.cuda())
    assert rel_err(out.cpu(), ref) <= 2e-6
    plain = ops.gemm(A.cuda(), 0, B.cuda(), b_mode, M, N, K, bias=bias.cuda(), act=0, residual=res.cuda(), split_k=False)
    assert rel_err(out.cpu(), plain.cpu()) <= 2e-6


def test_fused_quickgelu_mlp_matches_torch():
    """ops.mlp_quickgelu (tall path): pre-activation stored from the c_fc epilogue, QuickGELU' folded into the epilogue of
    dout . W2 - forward and every gradient against torch on the host."""
    from mil_amd import ops
    M, W = 300, 128
    g = torch.Generator().manual_seed(7)
    x = torch.randn((M, W), generator=g)
    W1 = torch.randn((4 * W, W), generator=g) / W ** 0.5
    b1 = torch.randn((4 * W,), generator=g) * 0.1
    W2 = torch.randn((W, 4 * W), generator=g) / (4 * W) ** 0.5
    b2 = torch.randn((W,), generator=g) * 0.1
    res = torch.randn((M, W), generator=g)
    go = torch.randn((M, W), generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (x, W1, b1, W2, b2, res)]
    pre = ref_in[0] @ ref_in[1].t() + ref_in[2]
    ref = (pre * torch.sigmoid(1.702 * pre)) @ ref_in[3].t() + ref_in[4] + ref_in[5]
    ref.backward(go)
    dev_in = [t.cuda().requires_grad_(True) for t in (x, W1, b1, W2, b2, res)]
    out = ops.mlp_quickgelu(*dev_in[:5], residual=dev_in[5])
    out.backward(go.cuda())
    assert rel_err(out.detach().cpu(), ref.detach()) <= 3e-6
    for a, b, name in zip(dev_in, ref_in, ("x", "W1", "b1", "W2", "b2", "residual")):
        assert rel_err(a.grad.cpu(), b.grad) <= 1e-5, name


@pytest.mark.parametrize("M,K,N,act,x_grad", [(300, 768, 512, "tanh", False), (4099, 768, 512, "tanh", False),
                                              (130, 2048, 512, "none", False), (1000, 512, 2048, "relu", False),
                                              (65, 64, 132, "tanh", False), (300, 768, 512, "tanh", True),
                                              (33000, 512, 256, "relu", False)])
def test_parameter_backward_with_fused_activation_and_bias_gradient(M, K, N, act, x_grad):
    """mil_linear_bwd_params: dW and db from one product launch (+ fold); with no input gradient wanted (fc_pathology: the
    bag features carry none) the activation derivative is applied while dy is staged.  Row counts cover one slice, odd
    tails, a clamped column tile (N = 132) and many splits."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    go = torch.randn(M, N, generator=g)
    dev = torch.device("cuda")
    xd = x.to(dev).requires_grad_(x_grad)
    Wd, bd = (t.to(dev).requires_grad_(True) for t in (W, b))
    y = ops.linear_act(xd, Wd, bd, act)
    (y * go.to(dev)).sum().backward()
    xr, Wr, br = (t.clone().double().requires_grad_(True) for t in (x, W, b))
    pre = F.linear(xr, Wr, br)
    ref = {"none": pre, "tanh": torch.tanh(pre), "relu": torch.relu(pre)}[act]
    (ref * go.double()).sum().backward()
    assert rel_err(Wd.grad.cpu(), Wr.grad.float()) <= 1e-5
    assert rel_err(bd.grad.cpu(), br.grad.float()) <= 1e-5
    if x_grad:
        assert rel_err(xd.grad.cpu(), xr.grad.float()) <= 1e-5


@pytest.mark.parametrize("M,N,K,b_mode,extras", [(66000, 512, 64, 0, "bias_tanh"), (40001, 640, 96, 0, "residual"),
                                                 (33000, 512, 128, 1, "accumulate"), (70000, 520, 32, 1, "plain"),
                                                 (66000, 2048, 64, 0, "aux")])
def test_tall_gemm_over_several_rounds(M, N, K, b_mode, extras):
    """Several rounds of the chip (more than 512 tiles, no split-K): ragged last row tile, a clamped last column tile,
    one-slice and multi-slice contractions, every epilogue flavour."""
    g = torch.Generator().manual_seed(M + N + K)
    dev = torch.device("cuda")
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5          # y = A W^T
    ref = A.double() @ W.double().t()
    Ad = A.to(dev)
    Bd = (W if b_mode == 0 else W.t().contiguous()).to(dev)
    if extras == "bias_tanh":
        b = torch.randn(N, generator=g)
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K, bias=b.to(dev), act=ops.ACT["tanh"])
        ref = torch.tanh(ref + b.double())
    elif extras == "residual":
        res = torch.randn(M, N, generator=g)
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K, residual=res.to(dev))
        ref = ref + res.double()
    elif extras == "accumulate":
        out = torch.ones(M, N, device=dev)
        ops.gemm(Ad, 0, Bd, b_mode, M, N, K, out=out, accumulate=True)
        ref = ref + 1
    elif extras == "aux":
        b = torch.randn(N, generator=g)
        pre = torch.empty(M, N, device=dev)
        out = ops.gemm_aux(Ad, Bd, b_mode, M, N, K, pre, 1, bias=b.to(dev), act=ops.ACT["quickgelu"])
        p = ref + b.double()
        assert rel_err(pre.cpu(), p.float()) <= 2e-6
        ref = p * torch.sigmoid(1.702 * p)
    else:
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K)
    assert rel_err(out.cpu(), ref.float()) <= 2e-6


@pytest.mark.parametrize("M,K,N,act,mode", [(333, 512, 512, "tanh", "all"), (65, 256, 64, "relu", "all"),
                                            (1000, 512, 512, "none", "residual"), (320, 512, 2048, "relu", "all"),
                                            (320, 2048, 512, "none", "frozen"), (999, 64, 136, "tanh", "all")])
def test_mid_size_layers_one_launch_per_product(M, K, N, act, mode):
    """65..1024 rows take csrc/mid_linear.hip (32 x 32 / 64 x 64 tiles, K contracted inside the workgroup): ragged row
    counts (the weight gradient contracts over M in groups of 8), clamped tiles, residual, frozen weights (dx only)."""
    assert ops._mid_ok(M, N, K)
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    res = torch.randn(M, N, generator=g)
    go = torch.randn(M, N, generator=g)
    dev = torch.device("cuda")
    frozen = mode == "frozen"
    xd = x.to(dev).requires_grad_(True)
    Wd, bd = (t.to(dev).requires_grad_(not frozen) for t in (W, b))
    rd = res.to(dev).requires_grad_(True) if mode == "residual" else None
    y = ops.linear_act(xd, Wd, bd, act, residual=rd)
    (y * go.to(dev)).sum().backward()
    xr, Wr, br, rr = (t.clone().double().requires_grad_(True) for t in (x, W, b, res))
    pre = F.linear(xr, Wr, br)
    ref = {"none": pre, "tanh": torch.tanh(pre), "relu": torch.relu(pre)}[act]
    if mode == "residual":
        ref = ref + rr
    (ref * go.double()).sum().backward()
    assert rel_err(y.detach().cpu(), ref.detach().float()) <= 2e-6
    assert rel_err(xd.grad.cpu(), xr.grad.float()) <= 1e-5
    if frozen:
        assert Wd.grad is None and bd.grad is None
    else:
        assert rel_err(Wd.grad.cpu(), Wr.grad.float()) <= 1e-5
        assert rel_err(bd.grad.cpu(), br.grad.float()) <= 1e-5
    if mode == "residual":
        assert rel_err(rd.grad.cpu(), rr.grad.float()) <= 1e-6


@pytest.mark.parametrize("M,N,K,b_mode,extras", [(10300, 512, 512, 0, "bias_tanh"), (10300, 512, 2048, 1, "accumulate"),
                                                 (10241, 520, 256, 1, "residual"), (9000, 512, 512, 0, "aux"),
                                                 (19000, 512, 256, 0, "plain")])
def test_64_row_tiles_for_partial_rounds(M, N, K, b_mode, extras):
    """Tall products (from half a round of the chip upwards) take k_gemm64: 64 x 128 tiles in an unpadded, XOR-swizzled
    LDS image, three workgroups per CU, no split-K."""
    g = torch.Generator().manual_seed(M + N + K)
    dev = torch.device("cuda")
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    ref = A.double() @ W.double().t()
    Ad = A.to(dev)
    Bd = (W if b_mode == 0 else W.t().contiguous()).to(dev)
    if extras == "bias_tanh":
        b = torch.randn(N, generator=g)
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K, bias=b.to(dev), act=ops.ACT["tanh"])
        ref = torch.tanh(ref + b.double())
    elif extras == "residual":
        res = torch.randn(M, N, generator=g)
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K, residual=res.to(dev))
        ref = ref + res.double()
    elif extras == "accumulate":
        out = torch.ones(M, N, device=dev)
        ops.gemm(Ad, 0, Bd, b_mode, M, N, K, out=out, accumulate=True)
        ref = ref + 1
    elif extras == "aux":
        b = torch.randn(N, generator=g)
        pre = torch.empty(M, N, device=dev)
        out = ops.gemm_aux(Ad, Bd, b_mode, M, N, K, pre, 1, bias=b.to(dev), act=ops.ACT["quickgelu"])
        p = ref + b.double()
        assert rel_err(pre.cpu(), p.float()) <= 2e-6
        ref = p * torch.sigmoid(1.702 * p)
    else:
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K)
    assert rel_err(out.cpu(), ref.float()) <= 2e-6


@pytest.mark.parametrize("M,N,K,act", [(32768, 512, 768, "tanh"), (49152 + 77, 256, 512, "none"), (65536, 512, 64, "relu")])
def test_tall_nt_product_on_the_low_valu_kernel(M, N, K, act):
    """mil_gemm_nt2 (256 x 256 tiles, LDS-DMA, csrc/linear_nt2.hip) directly and through mil_gemm's dispatch (fc_pathology's
    shape is the first case) against torch in float64; a ragged last row tile in the second case."""
    from mil_amd import _lib
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).to("cuda")
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to("cuda")
    b = torch.randn(N, generator=g).to("cuda")
    C = torch.empty(M, N, device="cuda")
    rc = _lib.lib().mil_gemm_nt2(ops._p(A), K, ops._p(W), K, ops._p(C), N, M, N, K, ops._p(b), ops.ACT[act], ops._stream())
    assert rc == 0
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 300, M)]).to("cuda")
    ref = A[rows].double() @ W.double().T + b.double()
    ref = torch.tanh(ref) if act == "tanh" else (torch.relu(ref) if act == "relu" else ref)
    assert float((C[rows].double() - ref).abs().max()) <= 2e-5
    via = ops.gemm(A, 0, W, 0, M, N, K, bias=b, act=ops.ACT[act])
    assert float((via[rows].double() - ref).abs().max()) <= 2e-5
    assert bool(torch.isfinite(C).all())


@pytest.mark.parametrize("rows,N,K,act", [(32768, 512, 768, "tanh"), (40000 + 13, 256, 256, "none"), (32768, 512, 512, "relu")])
def test_weight_gradient_of_a_tall_activation_on_the_low_valu_kernel(rows, N, K, act):
    """ops.linear_bwd_params at fc_pathology's size (and two more): dispatched to k_gemm_tn2 (csrc/linear_nt2.hip), checked
    against float64 - dW = (dY (.) act'(Y))^T X and db = the column sums."""
    g = torch.Generator().manual_seed(5)
    dy = (torch.randn(rows, N, generator=g) * 1e-2).to("cuda")
    x = torch.randn(rows, K, generator=g).to("cuda")
    pre = torch.randn(rows, N, generator=g).to("cuda")
    y = torch.tanh(pre) if act == "tanh" else (torch.relu(pre) if act == "relu" else None)
    dW, db = ops.linear_bwd_params(dy, y, ops.ACT[act], x)
    G = dy.double()
    if act == "tanh":
        G = G * (1 - y.double() ** 2)
    elif act == "relu":
        G = G * (y > 0).double()
    refW, refb = G.T @ x.double(), G.sum(0)
    assert rel_err(dW.double(), refW) <= 2e-5
    assert rel_err(db.double(), refb) <= 2e-5


@pytest.mark.parametrize("M,N,K,b_mode,extras", [(770, 1536, 512, 0, "bias_tanh"), (770, 512, 2048, 0, "aux"),
                                                 (770, 2048, 512, 1, "plain"), (770, 512, 1536, 1, "accumulate"),
                                                 (770, 516, 2048, 1, "residual"), (2048, 512, 768, 0, "bias_tanh"),
                                                 (1217, 2048, 512, 0, "aux"), (513, 520, 1024, 0, "residual"),
                                                 (770, 1536, 352, 0, "bias_tanh"), (300, 1024, 288, 1, "plain"),
                                                 (77, 1536, 512, 0, "aux"), (77, 512, 2048, 1, "residual")])
def test_64_x_64_tiles_for_a_few_hundred_rows(M, N, K, b_mode, extras):
    """Products of at most 2048 rows whose 64 x 128 tiles would leave CUs idle (the text tower at one bag x 10 prompts x 77 tokens)
    take k_gemm64n: 64 x 64 tiles, K split over blockIdx.z (raw partial tiles + k_splitk_reduce's epilogue) when the tiles alone
    do not fill the chip."""
    g = torch.Generator().manual_seed(M + N + K)
    dev = torch.device("cuda")
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    ref = A.double() @ W.double().t()
    Ad = A.to(dev)
    Bd = (W if b_mode == 0 else W.t().contiguous()).to(dev)
    if extras == "bias_tanh":
        b = torch.randn(N, generator=g)
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K, bias=b.to(dev), act=ops.ACT["tanh"])
        ref = torch.tanh(ref + b.double())
    elif extras == "residual":
        res = torch.randn(M, N, generator=g)
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K, residual=res.to(dev))
        ref = ref + res.double()
    elif extras == "accumulate":
        out = torch.ones(M, N, device=dev)
        ops.gemm(Ad, 0, Bd, b_mode, M, N, K, out=out, accumulate=True)
        ref = ref + 1
    elif extras == "aux":
        b = torch.randn(N, generator=g)
        pre = torch.empty(M, N, device=dev)
        out = ops.gemm_aux(Ad, Bd, b_mode, M, N, K, pre, 1, bias=b.to(dev), act=ops.ACT["quickgelu"])
        p = ref + b.double()
        assert rel_err(pre.cpu(), p.float()) <= 2e-6
        ref = p * torch.sigmoid(1.702 * p)
    else:
        out = ops.gemm(Ad, 0, Bd, b_mode, M, N, K)
    assert rel_err(out.cpu(), ref.float()) <= 2e-6
