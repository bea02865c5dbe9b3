#!/usr/bin/env python3
"""Experiment: the CFG pair (uncond, cond) as two INDEPENDENT B = 1 UNet calls on two HIP streams instead of one B = 2 call.

A UNet call at CFG batch 2 is ~420 dependent launches, most of them latency-bound kernels on a fraction of the 256 CUs
(DESIGN.md 4).  The two samples never interact (no cross-sample op: test_batch_rows_are_independent), so two chains of B = 1
launches can overlap each other's latencies.  This script measures it with two model instances (same weights, one per stream,
each with its own pinned context row); a product version would share the weights inside the library.

    python tools/two_stream.py [iters]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
_, u0, _ = bench.build_gpu_model(dev)
_, u1, _ = bench.build_gpu_model(dev)
g = torch.Generator(device='cpu').manual_seed(4)
x = torch.randn(2, 4, 64, 64, generator=g).to(dev)
ctx = (0.1 * torch.randn(2, 77, 768, generator=g)).to(dev)
t = torch.tensor([481, 481], device=dev)


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# (a) one B = 2 call
u0.pin_context(ctx)
ref = u0(x, t, context=ctx).clone()
ms2 = timed(lambda: u0(x, t, context=ctx), iters)
u0.unpin_context()
print(f'B=2, one stream            : {ms2:.3f} ms per CFG pair', flush=True)

# (b) one B = 1 call (how much of the B = 2 time is latency?)
c0, c1 = ctx[:1].contiguous(), ctx[1:].contiguous()
x0, x1 = x[:1].contiguous(), x[1:].contiguous()
t0_, t1_ = t[:1].contiguous(), t[1:].contiguous()
u0.pin_context(c0)
ms1 = timed(lambda: u0(x0, t0_, context=c0), iters)
print(f'B=1, one stream            : {ms1:.3f} ms per call (x2 sequential = {2 * ms1:.3f})', flush=True)

# (c) two B = 1 calls on two streams
u1.pin_context(c1)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
out = [None, None]


def pair():
    cur = torch.cuda.current_stream()
    s0.wait_stream(cur); s1.wait_stream(cur)
    with torch.cuda.stream(s0):
        out[0] = u0(x0, t0_, context=c0)
    with torch.cuda.stream(s1):
        out[1] = u1(x1, t1_, context=c1)
    cur.wait_stream(s0); cur.wait_stream(s1)


msp = timed(pair, iters)
print(f'B=1 + B=1, two streams     : {msp:.3f} ms per CFG pair', flush=True)
torch.cuda.synchronize()
e = torch.cat(out, 0)
print(f'max |two-stream - B=2| = {float((e - ref).abs().max()):.3e} (different tiles / splits per shape: rounding-level)', flush=True)
# (d) host enqueue time of one call (is the host the limit with twice the launches?)
torch.cuda.synchronize()
h0 = time.perf_counter()
for _ in range(5):
    pair()
h1 = time.perf_counter()
torch.cuda.synchronize()
print(f'host enqueue time of a pair: {(h1 - h0) / 5 * 1e3:.3f} ms', flush=True)
