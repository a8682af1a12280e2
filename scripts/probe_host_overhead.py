#!/usr/bin/env python3
"""GPU probe: host-side cost of the Python calls around one bench step (where does the inter-step gap go?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fadtk_amd import hip, _capi as K

def t(fn, n=2000):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6

a = torch.randn((100000, 512), device="cuda", dtype=torch.float16)
b = torch.randn((100000, 512), device="cuda", dtype=torch.float16)
ma, mb = hip.Moments(512), hip.Moments(512)
print(f"current_stream_ptr      {t(lambda: K.current_stream_ptr(0)):7.2f} us")
print(f"rows_view(a)            {t(lambda: K.rows_view(a)):7.2f} us")
print(f"ma.reset()              {t(lambda: ma.reset()):7.2f} us")
lib = K.load_library()
print(f"bare ctypes call        {t(lambda: lib.fad_version()):7.2f} us")
def step():
    ma.reset(); mb.reset()
    hip.Moments.update_multi([ma, mb], [a, b])
    return hip.frechet_from_moments(ma, mb, mean_dtype=0)
for _ in range(5): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"step wall               {(time.perf_counter() - t0) / n * 1e6:7.1f} us")
# enqueue-only cost of the moments part (no sync inside): how long the host needs to issue it
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    ma.reset(); mb.reset(); hip.Moments.update_multi([ma, mb], [a, b])
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"moments enqueue (host)  {(t1 - t0) / n * 1e6:7.1f} us per step; GPU done after {(time.perf_counter() - t0) / n * 1e6:7.1f} us per step")
