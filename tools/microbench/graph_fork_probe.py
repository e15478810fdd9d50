#!/usr/bin/env python
"""When does a forked branch of a REPLAYED HIP graph start?  (profiles/r06_notes.md sections 7-8: in the step graphs a branch forked from the
middle of a chain starts where that chain ends, although the captured edges allow it to start at the fork.)

A chain of three ~100 us spin kernels k1 k2 k3 on stream A; a branch kb (~100 us) on stream B that depends on k1 only; A joins B at the end.
If kb starts at the fork a replay takes ~300 us, if it starts where A's chain ends ~400 us.  Variants change HOW the fork is expressed:
  plain      B.wait_event(event recorded on A behind k1)
  backwait   + A waits for an event recorded on B right behind that wait (an edge back into the parent in front of k2)
  third      + a one-element launch on a stream C between k1 and k2 that A waits for
  both_off   k2 k3 move to a fresh stream too: both continuations leave the stream k1 ran on
  root       kb does not depend on k1 at all (a root of the graph): the reference for "starts at once"
usage (GPU box): python tools/microbench/graph_fork_probe.py"""
import time

import torch

dev = torch.device('cuda:0')
CYC = int(2.0e5)                 # ~100 us of torch.cuda._sleep


def build(variant):
    A, B, C, D = (torch.cuda.Stream() for _ in range(4))
    x = torch.zeros(1, device=dev)

    def body():
        with torch.cuda.stream(A):
            torch.cuda._sleep(CYC)                                   # k1
            ev = torch.cuda.Event(); ev.record(A)
        if variant == 'root':
            with torch.cuda.stream(B):
                torch.cuda._sleep(CYC)
        else:
            B.wait_event(ev)
            if variant == 'backwait':
                evb = torch.cuda.Event(); evb.record(B); A.wait_event(evb)
            if variant == 'third':
                C.wait_event(ev)
                with torch.cuda.stream(C):
                    x.add_(1.0)
                A.wait_stream(C)
            with torch.cuda.stream(B):
                torch.cuda._sleep(CYC)                               # kb
        cont = A
        if variant == 'both_off':
            D.wait_event(ev)
            cont = D
        with torch.cuda.stream(cont):
            torch.cuda._sleep(CYC)                                   # k2
            torch.cuda._sleep(CYC)                                   # k3
        if cont is not A:
            A.wait_stream(cont)
        A.wait_stream(B)
    # warm-up, then capture on A
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=A, capture_error_mode='thread_local'):
        body()
    return g


def build_step(variant):
    """the shape of critic step 1 of the headline graph: main A: P -> N -> G1..G6 -> L (joins E); side B: an earlier tail T (root), then E1..E5
    behind (T, P); ahead C: behind G6, four launches, joined at the very end.  Serial chains: A = 9 units, B = 1 + 5 units behind P.
    If E runs beside G a replay is ~10 units (P N G1-6 L ... + the ahead tail), if E waits for G ~15."""
    A, B, C = (torch.cuda.Stream() for _ in range(3))
    U = CYC // 4                                                    # ~25 us units

    def body():
        with torch.cuda.stream(B):
            torch.cuda._sleep(U)                                     # T: the side stream's earlier work
            evT = torch.cuda.Event(); evT.record(B)
        with torch.cuda.stream(A):
            if variant != 'no_tail_dep':
                A.wait_event(evT)                                    # (the previous step's update waits for both streams)
            torch.cuda._sleep(U)                                     # P
        B.wait_stream(A)                                             # fork: E behind P
        with torch.cuda.stream(A):
            torch.cuda._sleep(U)                                     # N
            for _ in range(6):
                torch.cuda._sleep(U)                                 # G1..G6
        with torch.cuda.stream(B):
            for _ in range(5):
                torch.cuda._sleep(U)                                 # E1..E5
        if variant != 'no_ahead':
            C.wait_stream(A)
            with torch.cuda.stream(C):
                for _ in range(4):
                    torch.cuda._sleep(U)
        A.wait_stream(B)
        with torch.cuda.stream(A):
            torch.cuda._sleep(U)                                     # L
        if variant != 'no_ahead':
            A.wait_stream(C)
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=A, capture_error_mode='thread_local'):
        body()
    return g


def main():
    for variant in ('step', 'no_ahead', 'no_tail_dep'):
        g = build_step(variant)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        print('step-shape %-12s %.1f us per replay (a unit is ~%.0f us: 10 units if E runs beside G, 15 if it waits)' % (
            variant, 1e6 * (time.perf_counter() - t0) / n, 86.0 / 4), flush=True)
    for variant in ('root', 'plain', 'backwait', 'third', 'both_off'):
        g = build(variant)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        print('%-9s %.1f us per replay' % (variant, 1e6 * (time.perf_counter() - t0) / n), flush=True)


if __name__ == '__main__':
    main()
