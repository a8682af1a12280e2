#!/usr/bin/env python3
"""GPU probe: throughput of the log-mel front ends (device-resident clips)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fadtk_amd import hip
def bench(name, fn, clips, secs_per_clip):
    fn(clips); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn(clips)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:8s} {len(clips):4d} clips x {secs_per_clip:4.0f} s: {dt*1e3:8.2f} ms  -> {len(clips)/dt:10.0f} clips/s  {len(clips)*secs_per_clip/dt:12.0f} x real time")
g = torch.Generator(device="cuda").manual_seed(0)
bench("vggish", lambda c: hip.logmel_vggish(c), [torch.randn(160000, generator=g, device="cuda") * 0.1 for _ in range(256)], 10)
bench("whisper", lambda c: hip.logmel_whisper(c), [torch.randn(480000, generator=g, device="cuda") * 0.1 for _ in range(128)], 30)
bench("htsat", lambda c: hip.logmel_htsat(c), [torch.randn(480000, generator=g, device="cuda") * 0.1 for _ in range(128)], 10)
for a, b in ((44100, 16000), (48000, 16000), (44100, 48000)):
    clip = torch.randn(10 * a, generator=g, device="cuda") * 0.1
    bench(f"rs{a // 1000}>{b // 1000}", lambda c: [hip.resample_kaiser(x, a, b, quantize_pcm16=True) for x in c], [clip] * 64, 10)
