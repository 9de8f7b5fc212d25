"""GPU: grouped skinny products and the two softmax stages of the multi-token absorbed attention against torch,
ragged groups, forward and autograd backward (ops._GroupedNT/_GroupedNN/_GroupedTN are closed under differentiation)."""
import pytest
import torch

from conftest import rel_err
from mil_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
LENS = [300, 64, 1, 517]


def _off():
    off = torch.zeros(len(LENS) + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(LENS), 0)
    return off


def _grads(fn, ins, go):
    ins = [t.clone().requires_grad_(True) for t in ins]
    out = fn(*ins)
    out.backward(go)
    return out.detach(), [t.grad for t in ins]


@pytest.mark.parametrize("N", [96, 64, 32, 128])          # 32 / 64 / 96: the skinny kernels (skinny_gemm.h); 128: k_gemm groups
def test_grouped_products_and_their_backwards(N):
    g = torch.Generator().manual_seed(0)
    R, G, K = sum(LENS), len(LENS), 512
    off = _off()
    A = torch.randn((R, K), generator=g)
    B = torch.randn((G, N, K), generator=g) / K ** 0.5
    bias = torch.randn((G, N), generator=g)
    go = torch.randn((R, N), generator=g)
    seg = [(int(off[i]), int(off[i + 1])) for i in range(G)]

    def ref_nt(A, B, bias):
        return torch.cat([A[a:b] @ B[i].t() + bias[i] for i, (a, b) in enumerate(seg)], 0)

    o_ref, g_ref = _grads(ref_nt, [A, B, bias], go)
    o, gr = _grads(lambda a, b, c: ops._GroupedNT.apply(a, b, c, off.to(DEV), max(LENS)), [A.to(DEV), B.to(DEV), bias.to(DEV)],
                   go.to(DEV))
    assert rel_err(o.cpu(), o_ref) <= 2e-6
    for x, y in zip(gr, g_ref):
        assert rel_err(x.cpu(), y) <= 1e-5

    P = torch.randn((R, N), generator=g)
    V = torch.randn((G, N, K), generator=g) / N ** 0.5
    res = torch.randn((R, K), generator=g)
    bo = torch.randn((K,), generator=g)
    go2 = torch.randn((R, K), generator=g)

    def ref_nn(P, V, bo, res):
        return torch.cat([P[a:b] @ V[i] for i, (a, b) in enumerate(seg)], 0) + bo + res

    o_ref, g_ref = _grads(ref_nn, [P, V, bo, res], go2)
    o, gr = _grads(lambda p, v, b, r: ops._GroupedNN.apply(p, v, b, r, off.to(DEV), max(LENS)),
                   [P.to(DEV), V.to(DEV), bo.to(DEV), res.to(DEV)], go2.to(DEV))
    assert rel_err(o.cpu(), o_ref) <= 2e-6
    for x, y in zip(gr, g_ref):
        assert rel_err(x.cpu(), y) <= 1e-5

    go3 = torch.randn((G, N, K), generator=g)

    def ref_tn(P, X):
        return torch.stack([P[a:b].t() @ X[a:b] for a, b in seg], 0)

    o_ref, g_ref = _grads(ref_tn, [P, A], go3)
    o, gr = _grads(lambda p, x: ops._GroupedTN.apply(p, x, off.to(DEV), G, max(LENS)), [P.to(DEV), A.to(DEV)], go3.to(DEV))
    assert rel_err(o.cpu(), o_ref) <= 2e-6
    for x, y in zip(gr, g_ref):
        assert rel_err(x.cpu(), y) <= 1e-5


def test_column_and_row_softmax_stages():
    g = torch.Generator().manual_seed(1)
    R, G, T, H, ld = sum(LENS), len(LENS), 10, 8, 96
    TH = T * H
    off = _off()
    S = torch.randn((R, ld), generator=g) * 3
    go = torch.randn((R, ld), generator=g)
    seg = [(int(off[i]), int(off[i + 1])) for i in range(G)]

    def ref_col(S):
        out = torch.zeros_like(S)
        out[:, :TH] = torch.cat([torch.softmax(S[a:b, :TH], 0) for a, b in seg], 0)
        return out

    o_ref, (g_ref,) = _grads(ref_col, [S], go)
    o, (gr,) = _grads(lambda s: ops._GrpColSoftmax.apply(s * 1.0, off.to(DEV), G, TH), [S.to(DEV)], go.to(DEV))
    assert float((o.cpu() - o_ref).abs().max()) <= 1e-6 and float(o[:, TH:].abs().max()) == 0.0
    assert rel_err(gr.cpu(), g_ref) <= 1e-5

    def ref_row(S):
        out = torch.zeros_like(S)
        out[:, :TH] = torch.softmax(S[:, :TH].reshape(R, T, H), 1).reshape(R, TH)
        return out

    o_ref, (g_ref,) = _grads(ref_row, [S], go)
    o, (gr,) = _grads(lambda s: ops._RowSoftmaxT.apply(s * 1.0, T, H), [S.to(DEV)], go.to(DEV))
    assert float((o.cpu() - o_ref).abs().max()) <= 1e-6 and float(o[:, TH:].abs().max()) == 0.0
    assert rel_err(gr.cpu(), g_ref) <= 1e-5


@pytest.mark.parametrize("lens,hint", [([3000, 2500], 3000), ([4096], 4096), ([9000, 40], 9000), ([17000], 17000),
                                       ([3000, 2500], 0), ([40000], 40000)])
def test_column_softmax_long_groups(lens, hint):
    """Bags longer than 2048 rows (the authors' reach ~15 000 patches): narrower column blocks keep the group in registers;
    hint 0 (length unknown) and groups beyond every register shape take the re-reading loop."""
    g = torch.Generator().manual_seed(2)
    R, G, T, H, ld = sum(lens), len(lens), 10, 8, 96
    TH = T * H
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    S = torch.randn((R, ld), generator=g) * 3
    go = torch.randn((R, ld), generator=g)
    seg = [(int(off[i]), int(off[i + 1])) for i in range(G)]

    def ref_col(S):
        out = torch.zeros_like(S)
        out[:, :TH] = torch.cat([torch.softmax(S[a:b, :TH], 0) for a, b in seg], 0)
        return out

    o_ref, (g_ref,) = _grads(ref_col, [S], go)
    o, (gr,) = _grads(lambda s: ops._GrpColSoftmax.apply(s * 1.0, off.to(DEV), G, TH, hint), [S.to(DEV)], go.to(DEV))
    # fp32 sums over up to 40 000 rows in a different order than the reference: a few 1e-5 relative
    assert float((o.cpu() - o_ref).abs().max()) <= 5e-5 and rel_err(o.cpu(), o_ref) <= 5e-5 and float(o[:, TH:].abs().max()) == 0.0
    assert rel_err(gr.cpu(), g_ref) <= 5e-5


@pytest.mark.parametrize("G,rows,N", [(40, 500, 96), (3, 700, 64), (64, 260, 32)])
def test_skinny_nt_wide_and_narrow_workgroups(G, rows, N):
    """k_skinny_nt picks 32-row workgroups for a few bags and 64-row ones otherwise: both against torch, ragged last tiles."""
    g = torch.Generator().manual_seed(G + rows)
    lens = [rows - (i % 7) for i in range(G)]
    off = torch.zeros(G + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(lens), 0)
    R, K = sum(lens), 512
    A = torch.randn((R, K), generator=g)
    B = torch.randn((G, N, K), generator=g) / K ** 0.5
    bias = torch.randn((G, N), generator=g)
    out = ops._gg_nt(A.to(DEV), B.to(DEV), bias.to(DEV), off.to(DEV), max(lens))
    ref = torch.cat([A[int(off[i]):int(off[i + 1])] @ B[i].t() + bias[i] for i in range(G)], 0)
    assert rel_err(out.cpu(), ref) <= 2e-6


@pytest.mark.parametrize("lens,T,ld", [([5000], 3, 32), ([2600, 2100, 4000], 7, 64), ([12288], 12, 96), ([3000] * 9, 10, 96)])
def test_column_softmax_few_long_groups_row_parallel_form(lens, T, ld):
    """At most 8 groups of more than 2048 rows: partial column statistics per 256-row chunk, then every chunk folds them and
    rewrites its rows (k_gcs_stats / k_gcs_apply); nine groups fall back to the one-workgroup-per-column-block form."""
    g = torch.Generator().manual_seed(5)
    R, G, H = sum(lens), len(lens), 8
    TH = T * H
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    S = torch.randn((R, ld), generator=g) * 3
    go = torch.randn((R, ld), generator=g)
    seg = [(int(off[i]), int(off[i + 1])) for i in range(G)]

    def ref_col(S):
        out = torch.zeros_like(S)
        out[:, :TH] = torch.cat([torch.softmax(S[a:b, :TH], 0) for a, b in seg], 0)
        return out

    o_ref, (g_ref,) = _grads(ref_col, [S], go)
    o, (gr,) = _grads(lambda s: ops._GrpColSoftmax.apply(s * 1.0, off.to(DEV), G, TH, max(lens)), [S.to(DEV)], go.to(DEV))
    assert float((o.cpu() - o_ref).abs().max()) <= 5e-5 and rel_err(o.cpu(), o_ref) <= 5e-5
    assert TH == ld or float(o[:, TH:].abs().max()) == 0.0
    assert rel_err(gr.cpu(), g_ref) <= 5e-5


@pytest.mark.parametrize("lens,M,N", [([9001], 96, 512), ([2048, 3000, 2500], 32, 512), ([12288], 64, 256), ([15592], 96, 512),
                                      ([4000, 0, 2100], 96, 512)])
def test_grouped_contraction_of_a_few_long_groups(lens, M, N):
    """C_g = A[rows_g]^T . X[rows_g] for at most 8 groups of >= 2048 rows: k_gemm64tn (64 x 64 tiles, ~3 workgroups per CU,
    rows of a split beyond the group's end contribute zero) + k_grouped_fold."""
    g = torch.Generator().manual_seed(sum(lens) + M)
    R, G = sum(lens), len(lens)
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    A = torch.randn((R, M), generator=g)
    X = torch.randn((R, N), generator=g)
    ref = torch.stack([A[int(off[i]):int(off[i + 1])].double().t() @ X[int(off[i]):int(off[i + 1])].double() for i in range(G)])
    out = ops._gg_tn(A.to(DEV), X.to(DEV), off.to(DEV), G, max(lens))
    assert out.shape == (G, M, N)
    assert rel_err(out.cpu(), ref.float()) <= 3e-6
    if 0 in lens:
        assert float(out[lens.index(0)].abs().max()) == 0.0
