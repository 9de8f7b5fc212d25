#!/usr/bin/env python3
"""Does a hipGraph replay run two branches of a captured step CONCURRENTLY on this runtime?  One branch is a chain of
dependent tiny launches (the token-side chain of the fusion step: ~5 us each, the chip idle), the other one chip-wide
product (the gate weight gradient: ~100 us).  Captured once on one stream (serial) and once with the product forked
onto a second stream (event fork / join inside the capture); prints both replay times.

    python tools/graph_fork_probe.py [--chain 40] [--n 2048]"""
import argparse
import json
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain", type=int, default=40)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only-two", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t = torch.zeros((32, 512), device=dev)
    A = torch.randn((a.n, a.n), device=dev)
    Bm = torch.randn((a.n, a.n), device=dev)
    C = torch.empty((a.n, a.n), device=dev)
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()

    def chain():
        for _ in range(a.chain):
            t.add_(1.0)

    def body(fork: bool, with_chain: bool = True, with_mm: bool = True):
        if fork:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if with_mm:
                    torch.mm(A, Bm, out=C)
            if with_chain:
                chain()
            torch.cuda.current_stream().wait_stream(side)
        else:
            if with_mm:
                torch.mm(A, Bm, out=C)
            if with_chain:
                chain()

    def timed(fork, **kw):
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            body(fork, **kw)
        torch.cuda.current_stream().wait_stream(cap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            body(fork, **kw)
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e6

    def two_graphs():
        """the two branches as two graphs replayed on two streams (fork / join by events, outside any capture)"""
        gs = []
        for kw in (dict(with_mm=False), dict(with_chain=False)):
            cap.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                body(False, **kw)
            gs.append(g)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

        def once():
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                gs[1].replay()
            with torch.cuda.stream(s1):
                gs[0].replay()
                s1.wait_stream(s2)
        for _ in range(10):
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            once()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e6

    if a.only_two:
        torch.mm(A, Bm, out=C)
        torch.cuda.synchronize()
        print(json.dumps({"two_graphs_two_streams_us": round(two_graphs(), 1)}))
        return
    out = {"chain_only_us": round(timed(False, with_mm=False), 1), "mm_only_us": round(timed(False, with_chain=False), 1),
           "serial_us": round(timed(False), 1), "forked_us": round(timed(True), 1), "chain": a.chain, "n": a.n}
    out["two_graphs_two_streams_us"] = round(two_graphs(), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
