// Host (pageable) rows -> HBM: the host side of fad_moments_update / _update_segmented / the per-song call when the caller hands
// numpy arrays, as fadtk's own callers do (fad.py:42-48 calc_embd_statistics(embd_lst), utils.py:13-16 np.load).
//
// Three routes, FAD_H2D_MODE (read once per thread):
//   pageable (default)  one hipMemcpy2DAsync from the caller's pageable buffer: the runtime's own pinned staging
//   threads             chunks of a few MiB; T host threads each own two PINNED buffers and one copy stream -- the memcpy of
//                       chunk c + 2 overlaps the DMA of chunk c, several SDMA transfers in flight
//   register            hipHostRegister the caller's pages in place, one DMA, unregister
// Measured on the MI355X box, [100000 x 512] float16 = 102.4 MB (scripts/probe_host_path.py, round 3): pageable 2.04 ms =
// 50.1 GB/s, register 2.01 ms = 51.1 GB/s, threads 2.10-2.42 ms = 42-49 GB/s (4 / 8 / 16 threads, 1-16 MiB chunks) with 9 ms
// outliers when the workers are scheduled late.  PCIe Gen5 x16 carries ~50 GB/s here whichever way the bytes are staged, so
// the runtime's route stays the default (round 2's "34 GB/s" was the whole calc_embd_statistics call -- handle creation, the
// finalize kernel and the 2 MB covariance coming back included -- not the copy); the other two stay selectable for hosts whose
// runtime path is slower.  In every mode the call returns with all reads of the caller's buffer done and the copy ordered in
// front of whatever follows on `st`; only `register` waits for the device (the unregister must).
//
// FAD_H2D_THREADS  1..16 (default 8)      FAD_H2D_CHUNK_KB  (default 4096)
#include "fad_common.h"

#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>

namespace fad {

namespace {

struct Stager {
    static constexpr int kMaxThreads = 16;
    int nthreads = 0;
    size_t chunk = 0;
    void* pin[kMaxThreads][2] = {};
    hipStream_t cs[kMaxThreads] = {};
    hipEvent_t ev[kMaxThreads][2] = {};
    hipEvent_t done[kMaxThreads] = {};
    hipEvent_t enter = nullptr;
    bool used[kMaxThreads][2] = {};                  // ev[t][s] has been recorded: a DMA out of pin[t][s] may still be in flight
    int mode = -1;                                   // 0 threads, 1 register, 2 pageable

    void release_all() {
        for (int t = 0; t < kMaxThreads; ++t) {
            for (int s = 0; s < 2; ++s) {
                if (pin[t][s]) (void)hipHostFree(pin[t][s]);
                if (ev[t][s]) (void)hipEventDestroy(ev[t][s]);
                pin[t][s] = nullptr; ev[t][s] = nullptr; used[t][s] = false;
            }
            if (done[t]) (void)hipEventDestroy(done[t]);
            if (cs[t]) (void)hipStreamDestroy(cs[t]);
            done[t] = nullptr; cs[t] = nullptr;
        }
        if (enter) (void)hipEventDestroy(enter);
        enter = nullptr; nthreads = 0;
    }

    int configure() {
        if (mode >= 0) return FAD_OK;
        const char* m = getenv("FAD_H2D_MODE");
        mode = (m && m[0] == 'r') ? 1 : (m && m[0] == 't') ? 0 : 2;
        const char* t = getenv("FAD_H2D_THREADS");
        int want = t ? atoi(t) : 8;
        if (want < 1) want = 1;
        if (want > kMaxThreads) want = kMaxThreads;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw > 0 && (unsigned)want > hw) want = (int)hw;
        nthreads = want;
        const char* c = getenv("FAD_H2D_CHUNK_KB");
        const long kb = c ? atol(c) : 4096;
        chunk = (size_t)(kb < 64 ? 64 : (kb > (1 << 16) ? (1 << 16) : kb)) << 10;
        FAD_HIP_TRY(hipEventCreateWithFlags(&enter, hipEventDisableTiming));
        return FAD_OK;
    }

    int lane(int t) {                                // resources of worker t, created on first use
        if (cs[t]) return FAD_OK;
        FAD_HIP_TRY(hipStreamCreateWithFlags(&cs[t], hipStreamNonBlocking));
        FAD_HIP_TRY(hipEventCreateWithFlags(&done[t], hipEventDisableTiming));
        for (int s = 0; s < 2; ++s) {
            FAD_HIP_TRY(hipHostMalloc(&pin[t][s], chunk, hipHostMallocDefault));
            FAD_HIP_TRY(hipEventCreateWithFlags(&ev[t][s], hipEventDisableTiming));
        }
        return FAD_OK;
    }
};

Stager& thread_stager(int device) {
    static thread_local PerThreadDevice<Stager> set;
    return set.get(device);
}

}  // namespace

// dst (device, row pitch dpitch bytes) <- src (host, row pitch spitch bytes), `rows` rows of `width` bytes.
int host_to_device_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, int device,
                      hipStream_t st) {
    if (rows == 0 || width == 0) return FAD_OK;
    Stager& S = thread_stager(device);
    FAD_TRY(S.configure());
    const size_t total = rows * width;
    if (S.mode == 2 || total < ((size_t)1 << 20)) {  // small blocks: the runtime's own staging is as fast and costs no threads
        FAD_HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, st));
        return FAD_OK;
    }
    if (S.mode == 1) {
        // pin the caller's pages in place, one DMA, unpin (the unregister has to wait for the transfer)
        const size_t span = (rows - 1) * spitch + width;
        if (hipHostRegister(const_cast<void*>(src), span, hipHostRegisterDefault) == hipSuccess) {
            hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            (void)hipHostUnregister(const_cast<void*>(src));
            if (e != hipSuccess) return set_error(FAD_ERR_HIP, "registered host copy failed: %s", hipGetErrorString(e));
            return FAD_OK;
        }
        (void)hipGetLastError();                     // memory that cannot be registered (read-only maps ...): the staged route
    }
    size_t rows_per_chunk = S.chunk / width;
    if (rows_per_chunk < 1) {                        // a single row above the chunk size: let the runtime stage it
        FAD_HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, st));
        return FAD_OK;
    }
    const size_t nchunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
    const int T = (int)(nchunks < (size_t)S.nthreads ? nchunks : (size_t)S.nthreads);
    for (int t = 0; t < T; ++t) FAD_TRY(S.lane(t));
    // whatever `st` still does with dst (the kernels of a previous block read the same staging area) comes first
    FAD_HIP_TRY(hipEventRecord(S.enter, st));
    for (int t = 0; t < T; ++t) FAD_HIP_TRY(hipStreamWaitEvent(S.cs[t], S.enter, 0));
    std::atomic<int> failed{0};
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { failed.store(1); return; }
        for (size_t c = (size_t)t; c < nchunks && !failed.load(std::memory_order_relaxed); c += (size_t)T) {
            const int slot = (int)((c / (size_t)T) & 1);
            if (S.used[t][slot] && hipEventSynchronize(S.ev[t][slot]) != hipSuccess) { failed.store(1); return; }
            const size_t r0 = c * rows_per_chunk;
            const size_t nr = (rows - r0 < rows_per_chunk) ? rows - r0 : rows_per_chunk;
            char* p = static_cast<char*>(S.pin[t][slot]);
            const char* s0 = static_cast<const char*>(src) + r0 * spitch;
            if (spitch == width) memcpy(p, s0, nr * width);
            else for (size_t r = 0; r < nr; ++r) memcpy(p + r * width, s0 + r * spitch, width);
            char* d0 = static_cast<char*>(dst) + r0 * dpitch;
            hipError_t e = (dpitch == width) ? hipMemcpyAsync(d0, p, nr * width, hipMemcpyHostToDevice, S.cs[t])
                                             : hipMemcpy2DAsync(d0, dpitch, p, width, width, nr, hipMemcpyHostToDevice, S.cs[t]);
            if (e == hipSuccess) e = hipEventRecord(S.ev[t][slot], S.cs[t]);
            if (e != hipSuccess) { failed.store(1); return; }
            S.used[t][slot] = true;
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(T > 1 ? T - 1 : 0);
    for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);                                         // the calling thread is worker 0
    for (std::thread& th : pool) th.join();
    if (failed.load()) {
        (void)hipGetLastError();
        for (int t = 0; t < T; ++t) (void)hipStreamSynchronize(S.cs[t]);      // nothing of this call stays in flight
        return set_error(FAD_ERR_HIP, "host-to-device staging failed");
    }
    for (int t = 0; t < T; ++t) {
        FAD_HIP_TRY(hipEventRecord(S.done[t], S.cs[t]));
        FAD_HIP_TRY(hipStreamWaitEvent(st, S.done[t], 0));
    }
    return FAD_OK;
}

}  // namespace fad
