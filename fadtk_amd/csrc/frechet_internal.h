// Shared between the three translation units of the Frechet distance (gfx950):
//   frechet_f64.hip    the all-float64 Newton-Schulz iteration (kernels of the scale, run_ns)
//   frechet.hip        single pair / batch of pairs: mixed-precision chains, workspaces, the C ABI of fad_frechet*
//   frechet_songs.hip  per-song scores against one baseline (fad_frechet_batched_vs_baseline): kernels and the route dispatcher
#pragma once
#include "fad_common.h"
#include "ns_check.h"
#include "ns32.h"

#include <cstdlib>
#include <vector>

struct fad_moments;
namespace fad {
const double* moments_packed(const fad_moments* h);
int moments_settle(const fad_moments* h, hipStream_t st);      // pending reset -> zeros
int moments_mark_read(const fad_moments* h, hipStream_t st);   // behind the launch of a kernel that reads the handle's statistics asynchronously
int moments_device(const fad_moments* h);
int moments_dim(const fad_moments* h);
const float* moments_runsum(const fad_moments* h);          // numpy's running column sums when they cover the handle's rows, else nullptr

constexpr int kStatScal = 8;                         // doubles per tile: sumsq, cross, trA, tr1, tr2, (3 spare)
static inline int64_t stat_blocks(int d) { return cdiv(d, 32); }
static inline size_t stat_doubles(int d) { const int64_t nb = stat_blocks(d); return (size_t)(2 * nb * d + kStatScal * nb * nb); }

// pairs of one batched chain (fad_frechet_from_moments_multi_begin): 8 until round 5 -- 128 workgroups per product, half the chip;
// 16 pairs put a workgroup of the 128 x 128-tile kernels on every CU (a chain of 16 costs little more than a chain of 8); beyond that the
// chain's time grows with the pairs (16 / 24 / 32 measure alike per score, r05v)
constexpr int kMaxMultiPairs = 32;

struct Workspace : NsWorkspace {
    DevBuf rows, offs, songbuf, songmat, rows2, songrun, songjobs;     // per-song path (songrun: numpy's float32 running column sums per song)
    DevBuf base_root;                               // ... sqrt(Sigma_b) | I | zeros of the symmetric D x D route
    void* song_pin = nullptr; size_t song_pin_cap = 0;      // ... and its pinned staging: offsets going up, scores coming down
    DevBuf mats32;                                  // low-precision leg: Y32[2], Z32[2], T32 (floats)
    DevBuf fast;                                    // the eight-launch chain (ns_fast.h): header, digit planes, split planes
    int fast_gen = 0;                               // per-call token of that chain (MatHdr::flag_gen)
    DevBuf songcov;                                 // ... scratch of the float16 per-song covariances (partial tiles, column sums, shifts)
    DevBuf fast_songs;                              // ... its batched form for songs: baseline digits + one block per song
    void* fast_songs_pin = nullptr; size_t fast_songs_pin_cap = 0;      // ... and what its correction kernel leaves for the host
    DevBuf fast_pairs;                              // ... and for B independent pairs (fad_frechet_from_moments_multi_begin): one block per pair
    void* fast_pairs_pin = nullptr; size_t fast_pairs_pin_cap = 0;
    struct Multi {                                  // an in-flight batch of pairs
        int count = 0, gen = 0;
        bool enqueued = false;                      // the batched chain is on the stream (else: end() scores the pairs one by one)
        const fad_moments_t* h1[kMaxMultiPairs] = {nullptr}; const fad_moments_t* h2[kMaxMultiPairs] = {nullptr};
    } multi;
    // an in-flight score (fad_frechet_from_moments_begin .. fad_frechet_end): everything the collecting side needs
    bool busy = false;
    struct Job {
        int d = 0, device = 0, k = 0, mean_dtype = -1, ddof = 1;
        bool mixed = false;                         // the low-precision chain was enqueued (else: end() runs the synchronous path)
        bool fast = false;                          // ... in its eight-launch form (ns_fast.h); nsf_prepare has staged (mu, Sigma)
        int gen = 0;                                // ... and this is its token
        double eps = 0.0;
        hipStream_t stream = nullptr;
        const double *cov1 = nullptr, *cov2 = nullptr, *mu1 = nullptr, *mu2 = nullptr;
    } job;
    hipEvent_t done_ev = nullptr;
    struct Pool* pool = nullptr;
    void release_all() {
        release(); rows.release(); offs.release(); songbuf.release(); songmat.release(); rows2.release(); songrun.release(); songjobs.release(); mats32.release(); base_root.release(); fast.release();
        fast_songs.release(); songcov.release(); fast_pairs.release();
        if (fast_pairs_pin) { (void)hipHostFree(fast_pairs_pin); fast_pairs_pin = nullptr; fast_pairs_pin_cap = 0; }
        if (fast_songs_pin) { (void)hipHostFree(fast_songs_pin); fast_songs_pin = nullptr; fast_songs_pin_cap = 0; }
        if (done_ev) { (void)hipEventDestroy(done_ev); done_ev = nullptr; }
        if (song_pin) { (void)hipHostFree(song_pin); song_pin = nullptr; song_pin_cap = 0; }
    }
    int reserve_song_pin(size_t bytes) {
        if (song_pin && song_pin_cap >= bytes) return FAD_OK;
        if (song_pin) (void)hipHostFree(song_pin);
        song_pin = nullptr; song_pin_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&song_pin, bytes + bytes / 2 + 4096, hipHostMallocDefault));
        song_pin_cap = bytes + bytes / 2 + 4096;
        return FAD_OK;
    }
};


// One pool of workspaces per (host thread, device), returned to the device when the thread ends (PerThreadDevice):
// calls from a thread pool (fad.py:229, 387 use tmap) never share scratch memory, and one thread can keep up to
// kSlots scores in flight (fad_frechet_from_moments_begin) -- each owns a slot until fad_frechet_end collects it.
struct Pool {
    static constexpr int kSlots = 8;
    Workspace slot[kSlots];
    int lp_iters = 5;                               // iterations the low-precision leg needed last time on this thread
    bool lp_hopeless = false;                       // ... or gave up on at once (a decaying spectrum): the next score enqueues iteration 0
                                                    // and the closing kernel only -- the LAUNCH count follows the history, never the value
    bool lp_verify = false;                         // ... or whose correction needed the verification products (ns_fast.h: SP_V2 / SP_V3): the next
                                                    // score enqueues them behind the correction at once instead of after a trip to the host
    int lp_wide = -1;                               // FAD_FRECHET_WIDE (read once): 0 = the chain only serves flat spectra, as in round 4
    int f64_iters = 0;                              // ... and the float64 iteration (single pair), 0 = not known yet
    int f64_iters_multi = 0;                        // ... and the float64 iteration of a BATCH of pairs (fad_frechet_multi_end)
    int mixed = -1;                                 // FAD_FRECHET_MIXED (read once): 0 = always the fp64 iteration
    int fast = -1;                                  // FAD_FRECHET_FAST (read once): 0 = round 2's twelve-launch float32 chain
    double pred_thr = 0.0;                          // FAD_FRECHET_PRED_THR (read once; -1 = the built-in rule), see pred_threshold
    void release_all() { for (Workspace& w : slot) w.release_all(); }
};
Pool& thread_pool(int device);                      // frechet.hip
Workspace* free_slot(int device);

struct NsProblem {                  // B problems of dimension d; strides in elements (0 = shared)
    int d; int64_t B;
    const double* cov1; int64_t s_cov1;
    const double* cov2; int64_t s_cov2;
    const double* mu1; int64_t s_mu1;
    const double* mu2; int64_t s_mu2;
    int mean_dtype;                 // ns_prepare: dtype whose rounding the mean term reproduces, or -1 (float64)
    int sym = 0;                    // cov1 cov2 is symmetric (then so is every iterate): the products may skip the mirrored tiles
};

static inline int ns_pstride(int d) {                // partial slots per problem: GEMM tiles or ns_first blocks
    const int64_t a = gemm_f64_slots_max(d), b = cdiv((int64_t)d * d, 256);
    return (int)(a > b ? a : b);
}
static inline size_t ns_small_bytes(int d, int64_t B) {
    return (size_t)B * (sizeof(NsState) + sizeof(Ns32State) + ((size_t)ns_pstride(d) + stat_doubles(d)) * sizeof(double)) + 256;
}

// rs = sum_k rowabs[k][i], cs = sum_k colabs[k][i] in a fixed order, with the loads of eight partials in flight at once
// (a plain loop issued them one dependent round trip after the other: 17 us for D = 512)
__device__ __forceinline__ void sum_partials(const double* __restrict__ rowabs, const double* __restrict__ colabs, int nb,
                                             int d, int i, double& rs, double& cs) {
    rs = 0.0; cs = 0.0;
    int k = 0;
    for (; k + 8 <= nb; k += 8) {
        double r[8], c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { r[q] = rowabs[(int64_t)(k + q) * d + i]; c[q] = colabs[(int64_t)(k + q) * d + i]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { rs += r[q]; cs += c[q]; }
    }
    for (; k < nb; ++k) { rs += rowabs[(int64_t)k * d + i]; cs += colabs[(int64_t)k * d + i]; }
}

constexpr int kSymMaxIter = 16;   // symmetric per-song route: iterates beyond this mean a spread the route's sqrt(Sigma_b) cannot carry

// ---- frechet_f64.hip
// Enqueue + run the batched iteration.  On return host_states (pinned, B entries) holds the final
// per-problem state; the caller turns it into scores.  States must have been cleared by the caller
// (so that pre-kernels like finalize_for_frechet can raise too_few).
// reuse_prepared: A = C1 C2 (first matrix of ws.mats) and the armed state are those of a float32 attempt on the same problem
// that just gave up (mixed_begin: same buffer, same ns_prepare) -- product, statistics and scale are not formed again.
int run_ns(const NsProblem& pb, int max_iter, double tol, int device, hipStream_t stream, Workspace& ws,
           NsState** host_states, bool reuse_prepared = false, double** y_bufs = nullptr, int first_chunk = 0);
void enqueue_mark_states_done(NsState* st, uint32_t mask, int B, hipStream_t stream);   // B <= 32 problems: those of `mask` need no iteration
// launches of that file's small kernels for the other two
void enqueue_clear_states(NsState* st, int64_t B, hipStream_t stream);                 // per-call reset of B iteration states
void enqueue_add_diag(double* M, int d, double eps, hipStream_t stream);               // M += eps I (fad.py:94-99)
void enqueue_finalize_for_frechet(const double* acc1, const double* acc2, int d, int ddof, double* mus, double* covs, NsState* st,
                                  hipStream_t stream, const float* run1 = nullptr, const float* run2 = nullptr);   // packed moments -> (mu, Sigma) x 2
void enqueue_ns_prepare(const double* stats_all, int d, int nb, const double* mu1, int64_t m1, const double* mu2, int64_t m2,
                        int mean_dtype, NsState* st_all, int mean_given, Ns32State* s32, int64_t B, hipStream_t stream);

// ---- frechet.hip: the low-precision chain batched over songs (fast_songs), for the per-song dispatcher
bool fast_song_dim(int d);
int64_t fast_songs_capacity(int d, size_t budget_bytes);
// Environment switches of the per-song routes, read ONCE per call of fad_frechet_batched_vs_baseline (tests flip them between calls
// of one process to run every kernel family on the same songs):
struct SongKnobs {
    long big_min = 8;           // FAD_SONG_BIG: smallest batch that iterates on the 128 x 128 tiles of ns_fast_big.h (0 = never)
    int res = 2;                // FAD_SONG_RES: D = 128 -- 2 products and iteration in one workgroup per song, 1 the iteration only, 0 the batched kernels
    int fast = 1;               // FAD_SONG_FAST: 0 = float64 routes only, 1 = low-precision chain with float64 fallback, 2 = strict (error if it accepts no song)
    bool gram = true;           // FAD_SONG_GRAM: songs of fewer frames than dimensions through the n x n Gram matrix
    bool stats16 = true;        // FAD_SONG_STATS16: float16 frames -> per-song statistics on the packed-f16 kernel
    bool cov16 = true;          // FAD_SONG_COV16: float16 frames -> per-song covariances on the moments tile kernels
    bool sym = true;            // FAD_SONG_SYM: the symmetric route sqrt(Sigma_b) Sigma_s sqrt(Sigma_b) for long songs
    int64_t sym_max_mult = 8;   // FAD_SONG_SYM_MAX_FRAMES_PER_DIM
    bool trace = false;         // FAD_FAST_TRACE: one stderr line per song of the low-precision chain
    double l0_scale = 0.5;      // FAD_SONG_L0_SCALE: multiplier on the x_min estimate the scaled steps start from (measured 3 / 2 / 1 / 0.5 / 0.25:
                                // 10 / 10 / 9 / 8 / 8 iterations at 32 x [1500 x 768], 10 / 9 / 8 / 7 / 8 at [1200 x 512], 7 / 6 / 6 / 7 / 8 at [2250 x 128])
    bool scaled = true;         // FAD_SONG_SCALED: scaled Newton-Schulz steps on the 128 x 128-tile family of the low-precision chain
    static SongKnobs from_env();
};
// covs: B covariances [d x d] float64 on the device; -> tr_sqrt[b] and ok[b] (1: accepted, 0: hand the song to the float64 routes)
int fast_songs(int d, int64_t B, const double* dcov_b, const double* covs, hipStream_t st, Workspace& ws,
               std::vector<double>& tr_sqrt, std::vector<char>& ok, int device, const SongKnobs& knobs);

}  // namespace fad
