#!/usr/bin/env python3
"""GPU probe: does replaying one config-3 step (moments of both sets + the whole Frechet chain) as a HIP graph shorten it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fadtk_amd import hip, dist as fdist

N, D = 100_000, 512
g = torch.Generator(device="cuda"); g.manual_seed(10)
a = torch.randn((N, D), generator=g, device="cuda").to(torch.float16)
b = (1.02 * torch.randn((N, D), generator=g, device="cuda") + 0.01).to(torch.float16)
sh = fdist.SharedStats(D, 2, 0)
ma, mb = sh.moments

def step_enqueue():
    ma.reset(); mb.reset()
    hip.Moments.update_multi([ma, mb], [a, b])
    return hip.FrechetJob(ma, mb, mean_dtype=0)

for _ in range(5): print("plain", step_enqueue().result()[0])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step_enqueue().result()
t_plain = (time.perf_counter() - t0) / 200
jobs = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200):
    jobs.append(step_enqueue())
    if len(jobs) == 3: jobs.pop(0).result()
for j in jobs: j.result()
t_pipe = (time.perf_counter() - t0) / 200
print(f"blocking {t_plain*1e6:.1f} us/step, three in flight {t_pipe*1e6:.1f} us/step")
for mode in (2, 1):                                   # what the library's own HIP-event timing costs the stream
    ma.set_timing(mode)
    jobs = []
    for i in range(20): step_enqueue().result()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200):
        jobs.append(step_enqueue())
        if len(jobs) == 3: jobs.pop(0).result()
    for j in jobs: j.result()
    t_ev = (time.perf_counter() - t0) / 200
    k, r, _ = ma.last_timing()
    print(f"three in flight with timing mode {mode}: {t_ev*1e6:.1f} us/step (tile kernel {k*1e3:.1f} us, reduce {r*1e3:.1f} us)")
ma.set_timing(False)
# the same with three pairs of accumulators used in turn (three sets of partial-tile buffers: 96 MB instead of 32)
lanes = [fdist.SharedStats(D, 2, 0) for _ in range(3)]
def step_on(k):
    x, y = lanes[k].moments
    x.reset(); y.reset()
    hip.Moments.update_multi([x, y], [a, b])
    return hip.FrechetJob(x, y, mean_dtype=0)
for i in range(20): step_on(i % 3).result()
jobs = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200):
    jobs.append(step_on(i % 3))
    if len(jobs) == 3: jobs.pop(0).result()
for j in jobs: j.result()
print(f"three in flight over three pairs of accumulators: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us/step")
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): step_enqueue().result()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            job = step_enqueue()
        for _ in range(5): gr.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): gr.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 200
    print(f"graph replay {t_graph*1e6:.1f} us/step (back to back, no result collection)")
    job.cancel()
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
