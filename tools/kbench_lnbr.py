#!/usr/bin/env python3
"""Times the fused LayerNorm(keys + row) + absorbed-pool pair (mil_lnbr_absorbed_pool_value_fwd / _bwd) and the separate
launches it replaces at the fusion bench's shape (32 bags x 1024 keys x 512), C ABI calls on preallocated buffers, HIP events.

    python tools/kbench_lnbr.py [--bags 32] [--keys 1024] [--reps 50]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mil_amd  # noqa: E402,F401
from mil_amd import _lib  # noqa: E402
from mil_amd.segments import AttnSegs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bags", type=int, default=32)
    ap.add_argument("--keys", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, N, E, H, C = a.bags, a.keys, 512, 8, 32
    rows = B * N
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (sc * torch.randn(*s, generator=g)).to(dev)        # noqa: E731
    # several copies of the big tensors so that a repetition does not find them in the 256 MB Infinity Cache
    NC = 6
    xs, dys = [r(rows, E) for _ in range(NC)], [r(rows, E) for _ in range(NC)]
    row, gamma, beta = r(B, E), 1 + r(E, sc=0.1), r(E, sc=0.1)
    pe, Qp = r(N, E, sc=0.3), r(B, H, E, sc=0.05)
    Wv, bv = r(H * C, E, sc=0.05), r(H * C, sc=0.05)
    segs = AttnSegs.make([1] * B, [N] * B, dev)
    s_it = AttnSegs.make([N] * B, [1] * B, dev)
    ys = [torch.empty(rows, E, device=dev) for _ in range(NC)]
    dxs = [torch.empty(rows, E, device=dev) for _ in range(NC)]
    stats = torch.empty(rows, 2, device=dev)
    pooled, lse, o = torch.empty(B, H, E, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H * C, device=dev)
    dpooled = r(B, H, E, sc=0.1)
    d_o, dg, db, dQp = torch.empty(B, E, device=dev), torch.empty(E, device=dev), torch.empty(E, device=dev), torch.empty(B, H, E, device=dev)
    nt = segs.ntiles
    wsf = torch.empty(nt * H * (E + 2), device=dev)
    wsb = torch.empty(nt * H * E + 16 * rows + 3 * nt * E + 4 * 2048 * E, device=dev)
    L = _lib.lib()
    p = lambda t: t.data_ptr()      # noqa: E731
    st = torch.cuda.current_stream().cuda_stream

    def fwd_fused(i):
        _lib.check(L.mil_lnbr_absorbed_pool_value_fwd(p(xs[i]), p(row), p(gamma), p(beta), 1e-5, p(pe), p(Qp), p(segs.k_off),
                                                      p(segs.tile_map), p(segs.bag_tile_off), nt, B, H, C, E, p(Wv), p(bv),
                                                      p(ys[i]), p(stats), p(pooled), p(lse), p(o), p(wsf), st), "fwd")

    def fwd_split(i):
        _lib.check(L.mil_layernorm_bagrow_fwd(p(xs[i]), p(row), p(s_it.q_bag), p(gamma), p(beta), rows, E, 1e-5, p(ys[i]),
                                              p(stats), st), "ln")
        _lib.check(L.mil_absorbed_pool_value_fwd(p(ys[i]), p(pe), p(Qp), p(segs.k_off), p(segs.tile_map), p(segs.bag_tile_off),
                                                 nt, B, H, C, E, p(Wv), p(bv), p(pooled), p(lse), p(o), p(wsf), st), "pool")

    def bwd_fused(i):
        _lib.check(L.mil_lnbr_absorbed_pool_bwd(p(xs[i]), p(row), p(gamma), p(beta), p(stats), p(ys[i]), p(pe), p(Qp), p(lse),
                                                p(dpooled), p(pooled), p(segs.k_off), p(segs.tile_map), p(segs.bag_tile_off),
                                                nt, rows, B, H, C, E, p(dys[i]), p(dxs[i]), p(d_o), p(dg), p(db), p(dQp),
                                                p(wsb), st), "bwd")

    def bwd_split(i):
        _lib.check(L.mil_absorbed_pool_bwd(p(ys[i]), p(pe), p(Qp), p(lse), p(dpooled), p(pooled), p(segs.k_off), p(segs.tile_map),
                                           p(segs.bag_tile_off), nt, rows, B, H, C, E, p(dys[i]), p(dxs[i]), p(dQp), p(wsb), st),
                   "pool bwd")
        _lib.check(L.mil_layernorm_bagrow_bwd(p(xs[i]), p(row), p(s_it.q_bag), p(s_it.q_off), B, p(gamma), p(dxs[i]), p(stats),
                                              rows, E, p(dxs[(i + 1) % NC]), p(d_o), p(dg), p(db), p(wsb), st), "ln bwd")

    def timed(fn):
        for i in range(NC):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(a.reps):
            fn(k % NC)
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / a.reps * 1e3, 1)

    for i in range(NC):
        fwd_fused(i)            # stats / pooled / lse of the last copy stay valid for every copy's backward timing
    out = {"shape": f"{B} x {N} x {E}", "fwd_fused_us": timed(fwd_fused), "fwd_split_us": timed(fwd_split)}
    fwd_fused(0)
    out["bwd_fused_us"] = timed(bwd_fused)
    os.environ["MIL_LNBR_BWD"] = "r16"
    out["bwd_fused_r16_us"] = timed(bwd_fused)
    os.environ.pop("MIL_LNBR_BWD")
    out["bwd_split_us"] = timed(bwd_split)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
