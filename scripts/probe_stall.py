#!/usr/bin/env python3
"""GPU probe for the occasional 30-70 ms stalls of a single blocking call: N blocking fad_frechet_from_moments calls on a
decaying pair (float64 route) and on a flat pair (eight-launch chain), per-call wall time with the host clock at its start -- outliers
(> 5 x the median) are printed with their index and start offset, so that a `rocprofv3 --hip-trace --kernel-trace` of this script can be
searched for what the host / the device did at that moment (scripts/rocpd_long_calls.py)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip, _capi as K

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
# every pass of Python's cyclic collector, timed (gc.callbacks): a long blocking call that coincides with one is the interpreter's, not the library's
gc_log, _gc_t0 = [], [0.0]
def _gc_cb(phase, info):
    if phase == "start": _gc_t0[0] = time.perf_counter()
    else: gc_log.append((info["generation"], (time.perf_counter() - _gc_t0[0]) * 1e3, time.perf_counter()))
gc.callbacks.append(_gc_cb)
def schedstat():
    try:
        with open("/proc/thread-self/schedstat") as f:
            a = f.read().split()
        return int(a[0]), int(a[1])
    except Exception:       # noqa: BLE001
        return 0, 0
def cpu_throttle():
    for pth in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(l.split() for l in open(pth).read().splitlines())
            return {k: int(v) for k, v in d.items() if "throttled" in k}
        except Exception:   # noqa: BLE001
            continue
    return {}
print("cgroup throttling counters at start:", cpu_throttle(), "; BLAS / OpenMP threads:", os.environ.get("OMP_NUM_THREADS"), os.environ.get("OPENBLAS_NUM_THREADS"))
rng = np.random.default_rng(0)
d, n = 512, 20000
t_origin = time.perf_counter()
for decay, tag in ((2.0, "k^-2 (float64 route)"), (0.0, "flat (eight-launch chain)")):
    lam = np.arange(1, d + 1) ** (-decay / 2.0)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    a = torch.from_numpy(((rng.standard_normal((n, d)) * lam) @ q.T).astype(np.float16)).cuda()
    b = torch.from_numpy(((1.05 * rng.standard_normal((n, d)) * lam) @ q.T + 0.01).astype(np.float16)).cuda()
    for gc_on in (True, False):
        (gc.enable if gc_on else gc.disable)()
        with hip.Moments(d) as ma, hip.Moments(d) as mb:
            hip.Moments.update_multi([ma, mb], [a, b])
            for _ in range(5): hip.frechet_from_moments(ma, mb, mean_dtype=K.FAD_F16)
            ts, starts, sched = [], [], []
            for i in range(n_calls):
                torch.cuda.synchronize(); s0 = schedstat(); t0 = time.perf_counter()
                hip.frechet_from_moments(ma, mb, mean_dtype=K.FAD_F16)
                ts.append((time.perf_counter() - t0) * 1e3); starts.append((t0 - t_origin) * 1e3)
                s1 = schedstat(); sched.append(((s1[0] - s0[0]) / 1e6, (s1[1] - s0[1]) / 1e6))
        ts = np.array(ts); med = float(np.median(ts))
        # (index, ms, start offset, ms this thread spent ON a CPU during the call, ms it spent runnable but WAITING for a CPU -- /proc/thread-self/schedstat)
        out = [(int(i), round(float(ts[i]), 3), round(starts[i], 1), round(sched[i][0], 2), round(sched[i][1], 2)) for i in np.nonzero(ts > 5 * med)[0]]
        print(f"{tag}, python gc {'on ' if gc_on else 'off'}: median {med:.3f} ms, p99 {np.percentile(ts, 99):.3f}, max {ts.max():.3f}; "
              f"outliers (> 5 x median) (index, ms, start offset ms, on-CPU ms, runnable-wait ms): {out}", flush=True)
gc.enable()
long_gc = [(g, round(ms, 1), round((t - t_origin) * 1e3, 1)) for g, ms, t in gc_log if ms > 2.0]
print("cgroup throttling counters at end:", cpu_throttle())
print(f"collector passes: {len(gc_log)} in all; longer than 2 ms (generation, ms, end offset ms): {long_gc}")
