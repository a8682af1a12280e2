// Internal helpers shared by the HIP translation units of libfad_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/fad_hip.h"

namespace fad {

// ---- thread-local error text -------------------------------------------------------------
char* err_buf();
int set_error(int code, const char* fmt, ...);

#define FAD_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return ::fad::set_error(FAD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                                    hipGetErrorString(e_), __FILE__, __LINE__);            \
    } while (0)

#define FAD_TRY(expr)                 \
    do {                              \
        int s_ = (expr);              \
        if (s_ != FAD_OK) return s_;  \
    } while (0)

int check_device(int device);         // FAD_OK if `device` is a gfx950 GPU
int num_cus(int device);

// Host (pageable) -> device copy of `rows` rows of `width` bytes through pinned, multi-threaded staging (host_stage.cpp).
// On return every read of `src` is done and the copy is ordered before later work on `st`; the device is never waited for.
int host_to_device_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, int device,
                      hipStream_t st);

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// The quotient np.mean forms from its float32 running sum (fadtk/fad.py:48 -> numpy _methods._mean: true_divide(float32 sum, intp
// count) resolves to the float64 loop and is cast back to float32).  Identical to a float32 division while n is a float32 value
// (n < 2^24, or even above it); for larger odd row counts only this form is numpy's.
__host__ __device__ __forceinline__ double numpy_mean_of_f32_sum(float run, double n) { return (double)(float)((double)run / n); }
static inline size_t dtype_size(int dt) {
    switch (dt) { case FAD_F16: case FAD_BF16: return 2; case FAD_F32: return 4; case FAD_F64: return 8; }
    return 0;
}

// RAII device switch (the caller's current device is restored on scope exit).
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Growable device buffer owned by a handle.
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int reserve(size_t bytes);
    void release();
};

// moments.hip: numpy's float32 running column sums of every segment [d_offsets[s], d_offsets[s + 1]) of DEVICE rows (float16 / bfloat16 /
// float32) -> d_out [n_segments x d] float32 (segment_running_sums: per-file means of the online path, per-song means of --indiv)
// (`jobs`: scratch for the job table of the LDS-staged walk -- float16 rows, 16-byte aligned, segments of a few hundred rows or more:
//  one workgroup per (segment, 32 columns) -- else one thread walks eight columns of a segment with its loads in flight)
int segment_running_sums_launch(const void* rows, int64_t ld, int d, int dtype, const int64_t* d_offsets, int64_t n_segments, float* d_out,
                                hipStream_t st, DevBuf* jobs = nullptr, int64_t mean_rows = 0, int device = 0);

// moments.hip: the library's side stream of a device (non-blocking, high priority; nullptr with FAD_MOMENTS_RUNSUM_SIDE=0): the running-sum walks
hipStream_t moments_side_stream(int device);
// moments.hip: covariances of B songs of float16 frames on the moments tile kernels (for frechet.hip's batched per-song chain)
bool song_cov_f16_ok(const void* rows, int64_t ld, int d);
int song_cov_f16_launch(const void* rows, int64_t ld, int d, const int64_t* d_offsets, const int64_t* d_song_ids, int64_t B,
                        int64_t max_frames, const double* d_mean_exact, const double* d_var_exact, double* d_cov_out, DevBuf& scratch,
                        int device, hipStream_t st);

// Scratch buffers of the handle-less entry points: one set per (host thread, device), so callers in a thread pool
// (fad.py:229, 387 use tmap) never share scratch memory, and the memory goes back to the device when the thread ends.
// T needs a default constructor and release_all().
template <typename T> struct PerThreadDevice {
    static constexpr int kSlots = 16;
    T slot[kSlots]; int dev[kSlots];
    PerThreadDevice() { for (int& d : dev) d = -1; }
    ~PerThreadDevice() {
        for (int i = 0; i < kSlots; ++i)
            if (dev[i] >= 0) { DeviceGuard g(dev[i]); if (g.ok) slot[i].release_all(); }
    }
    T& get(int device) {
        const int i = device & (kSlots - 1);
        if (dev[i] != device) {
            if (dev[i] >= 0) { DeviceGuard g(dev[i]); if (g.ok) slot[i].release_all(); }
            dev[i] = device;
        }
        return slot[i];
    }
};

// ---- fp64 GEMM on v_mfma_f64_16x16x4_f64 (gemm_f64.hip) ------------------------------------
// One launch = `ntypes` (1 or 2) GEMM shapes x `batch` independent problems:
//   C = alpha * A * B + beta_eye * I,  operands of problem b at base + b * stride (stride 0 = shared),
// optional per-workgroup partial sums of (C - gamma I)^2 at partials[b][slot], and an optional
// per-problem skip flag (skip[b * skip_stride] != 0 -> that problem's workgroups exit at once).
struct GemmType {
    const double* A; int64_t sa;
    const double* B; int64_t sb;
    double* C; int64_t sc;
    double alpha, beta_eye, gamma;
    double* partials;
    int b_upper = 0;        // B is upper triangular (zeros stored below the diagonal): column tile j stops at k < (j + 1) * tile
    int sym = 0;            // the product is known to be symmetric (commuting symmetric factors): only the tiles on and above the
                            // diagonal are computed, each off-diagonal tile is stored twice (a hint: honoured by the 64 x 64 kernel)
    // Newton-Schulz T product with a per-problem step scale on the DEVICE: when set, problem b reads m = mu[b * mu_stride] and uses
    // alpha = -0.5 m^3, beta_eye = 1.5 m, gamma = 1.5 m - 0.5 m^3 instead of the three constants above
    const double* mu = nullptr; int64_t mu_stride = 0;
};
// returns the number of partial slots per problem (>0) or a negative fad_status
// `check` (optional): one extra workgroup per problem runs ns_check_block (ns_check.h) beside the GEMM tiles.
struct NsCheckArgs;
int gemm_f64_launch(int d, const GemmType* types, int ntypes, int64_t batch, const int* skip, int skip_stride,
                    hipStream_t stream, int device, int partial_stride = 0, const NsCheckArgs* check = nullptr);
int gemm_f64_slots(int d, int ntypes, int64_t batch, int device);
int gemm_f64_slots_max(int d);

// ---- Newton-Schulz trace-sqrt driver (frechet_f64.hip: run_ns; workspaces in frechet_internal.h) ---------------------------------------
struct NsWorkspace {
    DevBuf mats;      // 6 * d*d doubles: A, Y0, Y1, Z0, Z1, T
    DevBuf small;     // partials, per-iteration stats, flags
    DevBuf stage;     // host->device staging of mu/cov
    void* pinned = nullptr; size_t pinned_cap = 0;
    void release();
};

}  // namespace fad
