// Frechet distance between Gaussians on the GPU (gfx950): ONE pair of (mu, Sigma) -- or a batch of independent pairs -- from host
// arrays, device arrays or packed moments.  Replaces calc_frechet_distance, fadtk/fad.py:51-120:
//     FAD = ||mu1 - mu2||^2 + tr C1 + tr C2 - 2 tr sqrt(C1 C2)
// This file: the workspaces (one pool per host thread and device), the mixed-precision chains (round 2's float32 iteration with a
// float64 correction; round 3's eight launches on split-float16 operands with exact int8 products, ns_fast.h; its batched forms for
// songs and for independent pairs, ns_fast_big.h / ns_fast_res.h), the hand-over to the all-float64 iteration (frechet_f64.hip)
// and the C ABI of fad_frechet / fad_frechet_from_moments*.  The per-song entry point is frechet_songs.hip.
#include "frechet_internal.h"
#include "ns_mean.h"
#include "ns_fast.h"
#include "ns_fast_big.h"
#include "ns_fast_res.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <type_traits>

namespace fad {

Pool& thread_pool(int device) {
    static thread_local PerThreadDevice<Pool> set;
    return set.get(device);
}
Workspace* free_slot(int device) {
    Pool& p = thread_pool(device);
    for (Workspace& w : p.slot)
        if (!w.busy) { w.pool = &p; return &w; }
    return nullptr;
}

SongKnobs SongKnobs::from_env() {
    SongKnobs k;
    auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
    if (const char* e = getenv("FAD_SONG_BIG")) k.big_min = atol(e);
    if (const char* e = getenv("FAD_SONG_RES")) k.res = (e[0] == '0') ? 0 : (e[0] == '1' ? 1 : 2);
    if (const char* e = getenv("FAD_SONG_FAST")) k.fast = (e[0] == '0') ? 0 : (e[0] == '2' ? 2 : 1);
    k.gram = !off("FAD_SONG_GRAM"); k.stats16 = !off("FAD_SONG_STATS16"); k.cov16 = !off("FAD_SONG_COV16"); k.sym = !off("FAD_SONG_SYM");
    if (const char* e = getenv("FAD_SONG_SYM_MAX_FRAMES_PER_DIM")) k.sym_max_mult = (int64_t)atoll(e);
    if (const char* e = getenv("FAD_FAST_TRACE")) k.trace = e[0] == '1';
    k.scaled = !off("FAD_SONG_SCALED");
    if (const char* e = getenv("FAD_SONG_L0_SCALE")) { const double v = atof(e); if (v > 0.0) k.l0_scale = v; }
    return k;
}


// ==========================================================================================
// Mixed-precision leg (single pair, d % 64 == 0): Newton-Schulz in fp32 on the f32-input MFMA down to the fp32
// floor, then ONE fp64 correction
//     tr sqrt(A) = tr Y + 1/2 tr(Z (A - Y Y)) + O(err^2),      A, Y Y and the traces in fp64,
// (first-order Newton step of X -> X^2 = A around Y with Z ~ Y^-1; SURVEY.md section 7 H1 measured 2e-9).  With
// S = sqrt(A), D = S - Y and Z = S^-1 + G the neglected terms are 1/2 tr(S^-1 D^2) and 1/2 tr(G R), bounded by
//     est = ||Z||^3 ||R||_F^2 / 8 + ||Z|| r ||R||_F / 2,    ||Z|| <= sqrt(||Z||_1 ||Z||_inf),  r = last residual;
// the result is accepted when est <= 1e-9 |tr| (measured: est overestimates the true error 10-1000x; config 3:
// est 2e-12, error 2e-13), otherwise -- ill-conditioned or rank-deficient products, fp32 not converging -- the
// all-fp64 iteration above runs from scratch.  Per call: A (fp64 GEMM), statistics, 4-5 fp32 iterations at
// ~11 us instead of ~25, Y Y (fp64 GEMM on fp32 operands), two small reduction kernels, ONE host sync; the result
// is written straight into pinned host memory.  The number of blind iterations is the count the previous call on
// this thread needed (scores of one run need the same count; a short batch is topped up two at a time).
// ==========================================================================================
struct MixedResult {       // status / pieces of the result, in pinned host memory (written by ns32_finish, or by fast_decide on the host)
    int status;            // 0: low-precision iteration not finished yet, 1: accepted, 2: rejected -> fp64 iteration,
                           // 4: a PREDICTED final iterate was rejected -> iterate on from `iters` with the strict threshold
    int iters, decided_at, nonfinite, too_few0, too_few1;
    double tr_scaled, c, tr1, tr2, mean_term, res, est;
    int prepared;          // A = C1 C2 (float64) and the armed state are valid: the float64 route may start from them
    int pad;
};

// One block: reduce the partials, decide, write the result where the host reads it (pinned host memory).
__global__ __launch_bounds__(256) void ns32_finish(const double* __restrict__ stats, int d, int nb,
                                                   const NsState* __restrict__ st, Ns32State* __restrict__ s32,
                                                   MixedResult* __restrict__ out, int max_low) {
    __shared__ double red[4];
    __shared__ double red3[12];
    const int tid = threadIdx.x;
    const bool live = s32->ok != 0;
    double mr = 0.0, mc = 0.0, corr = 0.0, r2 = 0.0, tr = 0.0;
    if (live) {
        const double* rowabs = stats;
        const double* colabs = stats + (int64_t)nb * d;
        const double* scal = stats + 2 * (int64_t)nb * d;
        for (int i = tid; i < d; i += 256) {
            double rs, cs;
            sum_partials(rowabs, colabs, nb, d, i, rs, cs);
            mr = fmax(mr, rs); mc = fmax(mc, cs);
        }
        for (int k = tid; k < nb * nb; k += 256) {
            const double* sc = scal + (int64_t)kStatScal * k;
            corr += sc[0]; r2 += sc[1]; tr += sc[2];
        }
    }
    const double zinf = block_max(mr, red), zone = block_max(mc, red);
    double v3[3] = {corr, r2, tr};
    block_sum_n<3>(v3, red3);
    corr = v3[0]; r2 = v3[1]; tr = v3[2];
    if (tid != 0) return;
    MixedResult o;
    o.status = 0; o.iters = s32->final_iter; o.decided_at = s32->decided_at; o.nonfinite = st->nonfinite;
    o.too_few0 = st->too_few[0]; o.too_few1 = st->too_few[1];
    o.c = st->c; o.tr1 = st->tr1; o.tr2 = st->tr2; o.mean_term = st->mean_term;
    o.tr_scaled = 0.0; o.res = 0.0; o.est = 0.0; o.prepared = 1; o.pad = 0;
    if (st->done || s32->failed) {
        o.status = 2;                                 // bad / zero product or fp32 gave up: the fp64 path decides
    } else if (live) {
        const int f = s32->final_iter;
        const int fm = f < 16 ? f : 15;
        // residual of the final iterate: measured when the check stopped AT it, else the bound from the one before
        double res = s32->res[fm];
        if (s32->decided_at == f - 1) { const double rp = s32->res[f - 1 < 16 ? f - 1 : 15]; res = 0.75 * rp * rp + 0.25 * rp * rp * rp; if (res < 2e-6) res = 2e-6; }
        const double zn = sqrt(zinf * zone), rn = sqrt(r2);
        const double trs = tr + 0.5 * corr;
        const double est = zn * zn * zn * rn * rn / 8.0 + zn * res * rn / 2.0;
        const bool finite = (trs == trs) && !isinf(trs) && (est == est) && !isinf(est);
        o.tr_scaled = trs; o.res = res; o.est = est;
        o.status = (finite && est <= 1e-9 * fabs(trs)) ? 1 : 2;
        if (o.status == 2 && finite && !s32->strict && s32->decided_at == f - 1 && f + 1 < max_low) {
            // the iterate was taken as final on a PREDICTED residual and the correction cannot absorb it: nothing is
            // lost -- (Y_f, Z_f) are intact, the iteration goes on from there and only the fp32 floor ends it now
            o.status = 4;
            s32->done = 0; s32->finished = 0; s32->ok = 0; s32->skip_corr = 1; s32->upd_skip[0] = 0; s32->upd_skip[1] = 0;
            s32->strict = 1;
        }
    }
    *out = o;            // pinned host memory: visible to the host once the stream has been synchronised
}

constexpr int kMaxLow = 14;
// Pairs on the eight-launch chain (round 5): scaled steps take a k^-1 spectrum (condition 3e5 of the product) to the float32-class floor in
// 12-13 iterations; below an x_min estimate of kWideL0Min the float64 route is the right one (scripts/ns_emulate_verify.py)
constexpr int kMaxLowWide = 22;
// (start = kWideL0Scale x the x_min estimate of ns_l0_from_participation, which is a third of the power-law model's x_min: emulated at
//  0.5 / 1 / 2 / 3: 12 / 12 / 11 / 10 iterations for k^-1, 9 / 8 / 7 / 7 for k^-0.5; a start above the true x_min only slows the smallest
//  eigenvalues down, and the k^-1 pair's true x_min is 7x the estimate)
constexpr double kWideL0Scale = 3.0, kWideL0Min = 9e-4;
static bool wide_enabled(Pool* p) {
    if (p && p->lp_wide < 0) { const char* e = getenv("FAD_FRECHET_WIDE"); p->lp_wide = (e && e[0] == '0') ? 0 : 1; }
    return !p || p->lp_wide != 0;
}
static int max_low_of(const Workspace& ws) { return (ws.job.fast && wide_enabled(ws.pool)) ? kMaxLowWide : kMaxLow; }

// When may the check of iteration k declare Y_{k+1} final from the bound b = 3/4 r_k^2 + 1/4 r_k^3 on its residual?
// The fp64 correction leaves an error of about (||Z||^3/8 + ||Z||/2) b^2 (ns32_finish: est, with ||R|| <~ b), which has
// to stay below 1e-9 |tr sqrt| ~ 1e-9 d for a flat spectrum: b <~ 2.5e-3 sqrt-ish of d/512 for ||Z|| ~ 2-3.  The
// Frobenius bound b itself overestimates the residual it predicts ~10x (measured, config 3: b = 1.4e-3, next residual
// 1.4e-4, est 3e-10 |tr|), so 2.5e-3 d/512 is taken as is.  Waiting for the fp32 floor instead (b <= 2e-6, round 1)
// costs one more iteration -- two launches of ~10 us -- on every well-conditioned score.  A rejected prediction costs
// one correction and one more trip to the host; the rest of THAT score then runs strict (Ns32State::strict), so the
// result is a function of the inputs alone, never of what the thread scored before.  FAD_FRECHET_PRED_THR (read once
// per thread) overrides the rule (tests use it to force a rejection).
static double pred_threshold(Pool* p, int d) {
    if (p && p->pred_thr == 0.0) { const char* e = getenv("FAD_FRECHET_PRED_THR"); p->pred_thr = (e && atof(e) > 0.0) ? atof(e) : -1.0; }
    if (p && p->pred_thr > 0.0) return p->pred_thr;
    return 2.5e-3 * (double)d / 512.0;
}

struct MixedBufs {
    double* A; float *Y[2], *Z[2], *T;
    NsState* dstate; double* partials; double* tilestats; Ns32State* s32; MixedResult* hres;
    unsigned nb;
};
static MixedBufs mixed_bufs(Workspace& ws, int d) {
    const int64_t dd = (int64_t)d * d;
    MixedBufs m;
    m.A = static_cast<double*>(ws.mats.p);
    m.Y[0] = static_cast<float*>(ws.mats32.p); m.Y[1] = m.Y[0] + dd;
    m.Z[0] = m.Y[1] + dd; m.Z[1] = m.Y[1] + 2 * dd;
    m.T = m.Y[1] + 3 * dd;
    m.dstate = static_cast<NsState*>(ws.small.p);
    m.partials = reinterpret_cast<double*>(m.dstate + 1);
    m.tilestats = m.partials + ns_pstride(d);
    m.s32 = reinterpret_cast<Ns32State*>(m.tilestats + stat_doubles(d));
    m.hres = reinterpret_cast<MixedResult*>(static_cast<char*>(ws.pinned) + sizeof(NsState));
    m.nb = (unsigned)stat_blocks(d);
    return m;
}

// iterations [ws.job.k, upto) of the low-precision leg, then the closing kernels (fp64 correction, result -> pinned host)
static int mixed_enqueue(Workspace& ws, int upto) {
    const int d = ws.job.d;
    hipStream_t stream = ws.job.stream;
    MixedBufs m = mixed_bufs(ws, d);
    int rc;
    for (int& k = ws.job.k; k < upto; ++k) {
        const int cur = k & 1;
        Gemm32Args g;
        memset(&g, 0, sizeof(g));
        if (k == 0) {
            // iteration 0 in one launch: Y0 = A/c and T0 = (3I - Y0)/2 are formed while A is staged, Y1 = Y0 T0, Z1 = T0
            // (Z0 = I needs no product).  No check rides on it -- its residual ||I - Y0|| decides nothing a well-posed
            // problem cares about (a non-finite product was caught by ns_prepare); the first check is iteration 1's.
            g.C[0] = m.Y[1]; g.C[1] = m.Z[1]; g.alpha[0] = 1.0f; g.ntypes = 1; g.A64 = m.A; g.st64 = m.dstate;
            g.skip = &m.s32->done;               // (set by ns_prepare: bad / zero product, or a spectrum the float32 leg cannot serve)
            FAD_TRY(gemm_f32_first_launch(d, g, stream));
            continue;
        }
        // T = (3I - Z Y)/2 and the residual partials of iteration k
        g.A[0] = m.Z[cur]; g.B[0] = m.Y[cur]; g.C[0] = m.T; g.alpha[0] = -0.5f; g.beta_eye[0] = 1.5f; g.gamma[0] = 1.0f;
        g.partials[0] = m.partials; g.skip = &m.s32->done; g.ntypes = 1;
        const int nslots = gemm_f32_launch(d, g, stream);
        if (nslots < 0) return nslots;
        memset(&g, 0, sizeof(g));
        // Y <- Y T, Z <- T Z + the check of iteration k as an extra workgroup
        g.A[0] = m.Y[cur]; g.B[0] = m.T; g.C[0] = m.Y[cur ^ 1]; g.alpha[0] = 1.0f;
        g.A[1] = m.T; g.B[1] = m.Z[cur]; g.C[1] = m.Z[cur ^ 1]; g.alpha[1] = 1.0f;
        g.ntypes = 2;
        g.skip = &m.s32->upd_skip[k & 1];
        g.check = 1; g.k = k; g.max_low = kMaxLow; g.nslots = nslots; g.chk_partials = m.partials; g.st = m.s32; g.st64 = m.dstate;
        g.thr_pred = pred_threshold(ws.pool, d);
        rc = gemm_f32_launch(d, g, stream);
        if (rc < 0) return rc;
    }
    // fp64 correction on the final iterate (which of the ping-pong buffers: known on the device only): Y Y in fp64 with the
    // statistics of R = A/c - Y Y and Z formed in the epilogue (one launch instead of product + ns32_corr_partials)
    NsProductExt ext;
    memset(&ext, 0, sizeof(ext));
    ext.stats = m.tilestats; ext.st = m.dstate; ext.A64 = m.A; ext.Z32 = m.Z[0]; ext.Z32_alt = m.Z[1];
    FAD_TRY(gemm_f64_correction_launch(d, m.Y[0], m.Y[1], &m.s32->final_iter, &m.s32->skip_corr, ext, stream));
    hipLaunchKernelGGL(ns32_finish, dim3(1), dim3(256), 0, stream, m.tilestats, d, (int)m.nb, m.dstate, m.s32, m.hres, kMaxLow);
    FAD_HIP_TRY(hipGetLastError());
    if (!ws.done_ev) FAD_HIP_TRY(hipEventCreateWithFlags(&ws.done_ev, hipEventDisableTiming));
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, stream));
    return FAD_OK;
}

// ==========================================================================================
// The eight-launch form of the chain (ns_fast.h): exact products on the int8 MFMA, iteration on split-float16 operands.
// D in {256, 512, 768, 1024}; FAD_FRECHET_FAST=0 keeps round 2's float32 chain (the two are compared in the tests).
// ==========================================================================================
static bool mixed_eligible(Workspace& ws, int d, int max_iter, double tol);
static bool fast_dim(int d) { return d == 256 || d == 384 || d == 512 || d == 768 || d == 1024; }
static bool fast_eligible(Workspace& ws, int d, int max_iter, double tol) {
    Pool* p = ws.pool;
    if (p && p->fast < 0) { const char* e = getenv("FAD_FRECHET_FAST"); p->fast = (e && e[0] == '0') ? 0 : 1; }
    return (!p || p->fast) && fast_dim(d) && mixed_eligible(ws, d, max_iter, tol);
}

struct FastBufs {
    nsf::MatHdr* hdr;                               // [2]
    uint4* digC[2];
    nsf::SplitMat P, Y[2], Z[2], T;
    uint4 *digY[2], *digYt[2];
    nsf::SplitMat Rv, Pv, Ev;                       // verification: kVerScale R, P' = Z R', E' = kVerScale (I - Z Y)
    // pinned host memory behind NsState + MixedResult: what the correction kernel leaves for fast_decide (and the verification's record)
    int* host_words; double* host_vals; double* host_stats; double* host_vstats;
};
static size_t fast_bytes(int d) {
    const size_t dd = (size_t)d * d;
    return 256 + 2 * 6 * dd + 6 * 8 * dd + 4 * 6 * dd + 3 * 8 * dd + 256;
}
static size_t fast_pinned_bytes(int d) {
    const size_t nb = (size_t)d / 32;
    return sizeof(NsState) + sizeof(MixedResult) + 64 + nsf::kHostWords * sizeof(int) + nsf::kHostVals * sizeof(double) +
           (nsf::kTileStats + 2 + nsf::kVerStats) * nb * nb * sizeof(double) + 128;
}
static FastBufs fast_bufs(Workspace& ws, int d) {
    const size_t dd = (size_t)d * d;
    char* p = static_cast<char*>(ws.fast.p);
    FastBufs f;
    f.hdr = reinterpret_cast<nsf::MatHdr*>(p); p += 256;
    for (int i = 0; i < 2; ++i) { f.digC[i] = reinterpret_cast<uint4*>(p); p += 6 * dd; }
    nsf::SplitMat* mats[6] = {&f.P, &f.Y[0], &f.Y[1], &f.Z[0], &f.Z[1], &f.T};
    for (nsf::SplitMat* m : mats) {
        m->a = reinterpret_cast<uint4*>(p); p += 4 * dd;
        m->at = reinterpret_cast<uint4*>(p); p += 4 * dd;
    }
    for (int i = 0; i < 2; ++i) { f.digY[i] = reinterpret_cast<uint4*>(p); p += 6 * dd; f.digYt[i] = reinterpret_cast<uint4*>(p); p += 6 * dd; }
    nsf::SplitMat* vmats[3] = {&f.Rv, &f.Pv, &f.Ev};
    for (nsf::SplitMat* m : vmats) {
        m->a = reinterpret_cast<uint4*>(p); p += 4 * dd;
        m->at = reinterpret_cast<uint4*>(p); p += 4 * dd;
    }
    char* h = static_cast<char*>(ws.pinned);
    f.host_words = nullptr; f.host_vals = nullptr; f.host_stats = nullptr; f.host_vstats = nullptr;
    if (h && ws.pinned_cap >= fast_pinned_bytes(d)) {
        const size_t nb2 = ((size_t)d / 32) * ((size_t)d / 32);
        h += ((sizeof(NsState) + sizeof(MixedResult) + 63) / 64) * 64;
        f.host_vals = reinterpret_cast<double*>(h); h += nsf::kHostVals * sizeof(double);
        f.host_stats = reinterpret_cast<double*>(h); h += (nsf::kTileStats + 2) * nb2 * sizeof(double);
        f.host_vstats = reinterpret_cast<double*>(h); h += nsf::kVerStats * nb2 * sizeof(double);
        f.host_words = reinterpret_cast<int*>(h);
    }
    return f;
}

// K1 on `stream`: mu of both sets and (from packed moments) their covariances into the slot's staging area, state reset, scales,
// digit planes.  acc1 == nullptr: the caller's device matrices cov1 / cov2 are used as they are.
static int fast_prepare(Workspace& ws, int d, int ddof, const double* acc1, const double* acc2, const double* cov1, const double* cov2,
                        const double* mu1, const double* mu2, int mean_dtype, double* mus, double* covs, hipStream_t st,
                        const float* run1 = nullptr, const float* run2 = nullptr) {
    void* const before = ws.fast.p;
    FAD_TRY(ws.fast.reserve(fast_bytes(d)));
    FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
    FastBufs f = fast_bufs(ws, d);
    if (ws.fast.p != before) FAD_HIP_TRY(hipMemsetAsync(f.hdr, 0, 256, st));       // a fresh header: no stale token in its flag words
    ws.job.gen = ++ws.fast_gen;
    if (ws.fast_gen > (1 << 30)) ws.fast_gen = 1;
    nsf::PrepArgs a;
    memset(&a, 0, sizeof(a));
    a.acc[0] = acc1; a.acc[1] = acc2; a.cov_in[0] = cov1; a.cov_in[1] = cov2; a.mu_in[0] = mu1; a.mu_in[1] = mu2;
    a.run[0] = run1; a.run[1] = run2;
    a.d = d; a.ddof = ddof; a.gen = ws.job.gen; a.mean_dtype = mean_dtype;
    a.mus = mus; a.covs = covs;
    a.dig[0] = f.digC[0]; a.dig[1] = f.digC[1];
    a.st = static_cast<NsState*>(ws.small.p);
    a.hdr[0] = f.hdr; a.hdr[1] = f.hdr + 1;
    hipLaunchKernelGGL(nsf::nsf_prepare, dim3((unsigned)((int64_t)d * d / 2048 + 1), 2), dim3(512), 0, st, a);
    FAD_HIP_TRY(hipGetLastError());
    return FAD_OK;
}

template <int NS> static void fast_launch_split(int mode, unsigned t, unsigned B, const nsf::SplitArgs& g, hipStream_t st) {
    if (mode == nsf::SP_FIRST) hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_FIRST>), dim3(t, t, B), dim3(512), 0, st, g);
    else if (mode == nsf::SP_T) hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_T>), dim3(t, t, B), dim3(512), 0, st, g);
    else if (mode == nsf::SP_V2) hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_V2>), dim3(t, t, 2 * B), dim3(512), 0, st, g);
    else if (mode == nsf::SP_V3) hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_V3>), dim3(t, t, B), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((nsf::nsf_split<NS, nsf::SP_U>), dim3(t, t, 3 * B), dim3(512), 0, st, g);
}
// The two verification launches (ns_fast.h: SP_V2 / SP_V3) behind a correction whose R' planes are in place: for every problem of the
// batch with a final iterate.  Records land in pinned host memory (vstats / vwords, hstride apart).
static void fast_split(int d, int mode, const nsf::SplitArgs& g, hipStream_t st, unsigned B);
static void fast_verify_launch(int d, unsigned B, nsf::SplitArgs g, const nsf::SplitMat (&Y)[2], const nsf::SplitMat (&Z)[2], const nsf::SplitMat& Rv,
                               const nsf::SplitMat& Pv, const nsf::SplitMat& Ev, const Ns32State* s32, double* vstats, int* vwords, int64_t hstride,
                               hipStream_t st) {
    g.sel = &s32->final_iter; g.skip = &s32->skip_corr;
    g.Zf[0] = Z[0]; g.Zf[1] = Z[1]; g.Yf[0] = Y[0]; g.Yf[1] = Y[1];
    g.vstats = vstats; g.vwords = vwords; g.hstride = hstride;
    g.B[0] = Rv; g.C[0] = Pv; g.C[1] = Ev;
    fast_split(d, nsf::SP_V2, g, st, B);
    g.B[0] = Pv; g.A[1] = Ev; g.C[0] = nsf::SplitMat{nullptr, nullptr}; g.C[1] = nsf::SplitMat{nullptr, nullptr};
    fast_split(d, nsf::SP_V3, g, st, B);
}
static void fast_split(int d, int mode, const nsf::SplitArgs& g, hipStream_t st, unsigned B) {
    const unsigned t = (unsigned)(d / 32);
    switch (d) {
        case 128: fast_launch_split<1>(mode, t, B, g, st); break;
        case 256: fast_launch_split<2>(mode, t, B, g, st); break;
        case 384: fast_launch_split<3>(mode, t, B, g, st); break;
        case 512: fast_launch_split<4>(mode, t, B, g, st); break;
        case 768: fast_launch_split<6>(mode, t, B, g, st); break;
        default: fast_launch_split<8>(mode, t, B, g, st); break;
    }
}
// the batched form of SP_T / SP_U on 128 x 128 tiles (ns_fast_big.h); 64 KiB + of dynamic LDS: the attribute is set once per device
template <int NJ> static void fast_big_attrs() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_big<nsf::SP_T, NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kBigLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_big<nsf::SP_U, NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kBigLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_big<nsf::SP_FIRST, NJ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kBigLds);
}
template <int NJ> static void fast_big_launch(int d, int mode, const nsf::SplitArgs& g, hipStream_t st, unsigned B) {
    const int tt = nsf::big_tiles(d, NJ);
    const unsigned one = (unsigned)nsf::big_grid((int)B, tt), two = (unsigned)nsf::big_grid((int)B, 2 * tt);
    if (mode == nsf::SP_FIRST) hipLaunchKernelGGL((nsf::nsf_big<nsf::SP_FIRST, NJ>), dim3(one), dim3(256), nsf::kBigLds, st, g);
    else if (mode == nsf::SP_T) hipLaunchKernelGGL((nsf::nsf_big<nsf::SP_T, NJ>), dim3(one), dim3(256), nsf::kBigLds, st, g);
    else hipLaunchKernelGGL((nsf::nsf_big<nsf::SP_U, NJ>), dim3(two + B), dim3(256), nsf::kBigLds, st, g);
}
// the residual partials an SP_T launch of the batch leaves per problem (the check riding on the SP_U launch sums them: g.nslots)
// FAD_BIG_NARROW_BELOW (probe switch): narrow tiles for launches of fewer than that many wide-tile workgroups (default: big_nj's 512)
static int fast_big_nj(int d, int products, unsigned B) {
    static const int below = [] { const char* e = getenv("FAD_BIG_NARROW_BELOW"); return e ? atoi(e) : 0; }();
    if (below > 0) return (products * (d / 128) * (d / 128) * (int)B < below) ? 1 : 2;
    return nsf::big_nj(d, products, (int)B);
}
static int fast_big_t_slots(int d, unsigned B) { return nsf::big_tiles(d, fast_big_nj(d, 1, B)); }
static int fast_split_big(int d, int mode, nsf::SplitArgs g, hipStream_t st, unsigned B, int device) {
    static std::atomic<unsigned> ready{0};
    if (device >= 0 && device < 32 && !(ready.load(std::memory_order_acquire) & (1u << device))) {
        fast_big_attrs<1>(); fast_big_attrs<2>();
        FAD_HIP_TRY(hipGetLastError());
        ready.fetch_or(1u << device, std::memory_order_release);
    }
    g.nprob = (int)B;
    if (fast_big_nj(d, mode == nsf::SP_U ? 2 : 1, B) == 1) fast_big_launch<1>(d, mode, g, st, B);
    else fast_big_launch<2>(d, mode, g, st, B);
    return FAD_OK;
}
static int fast_i8_big(int d, int mode, const nsf::I8Args& g, hipStream_t st, unsigned B, int device) {
    static std::atomic<unsigned> ready{0};
    if (device >= 0 && device < 32 && !(ready.load(std::memory_order_acquire) & (1u << device))) {
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_i8_big<nsf::I8_A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kI8BigLds));
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_i8_big<nsf::I8_G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kI8BigLds));
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_i8_big<nsf::I8_G, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kI8BigLds));
        ready.fetch_or(1u << device, std::memory_order_release);
    }
    const unsigned grid = (unsigned)nsf::big_grid((int)B, (d / 128) * (d / 64));
    if (mode == nsf::I8_A) hipLaunchKernelGGL((nsf::nsf_i8_big<nsf::I8_A>), dim3(grid), dim3(512), nsf::kI8BigLds, st, g, (int)B);
    else if (g.Rv.a) hipLaunchKernelGGL((nsf::nsf_i8_big<nsf::I8_G, true>), dim3(grid), dim3(512), nsf::kI8BigLds, st, g, (int)B);
    else hipLaunchKernelGGL((nsf::nsf_i8_big<nsf::I8_G>), dim3(grid), dim3(512), nsf::kI8BigLds, st, g, (int)B);
    return FAD_OK;
}
template <int NS8> static void fast_launch_i8(int mode, unsigned t, unsigned B, const nsf::I8Args& g, hipStream_t st) {
    if (mode == nsf::I8_A) hipLaunchKernelGGL((nsf::nsf_i8<NS8, nsf::I8_A>), dim3(t, t, B), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((nsf::nsf_i8<NS8, nsf::I8_G>), dim3(t, t, B), dim3(512), 0, st, g);
}
static void fast_i8(int d, int mode, const nsf::I8Args& g, hipStream_t st, unsigned B = 1) {
    const unsigned t = (unsigned)(d / 32);
    switch (d) {
        case 128: case 256: fast_launch_i8<1>(mode, t, B, g, st); break;       // (d = 128: four k-steps, half the waves idle)
        case 384: case 512: fast_launch_i8<2>(mode, t, B, g, st); break;                  // (d = 384: twelve k-steps, waves 6, 7 idle)
        case 768: fast_launch_i8<3>(mode, t, B, g, st); break;
        default: fast_launch_i8<4>(mode, t, B, g, st); break;
    }
}

// iterations [ws.job.k, upto) of the chain, then the exact correction, whose partials land in pinned host memory
static nsf::SplitArgs fast_split_args(Workspace& ws, const MixedBufs& m, const FastBufs& f, int d) {
    nsf::SplitArgs g;
    memset(&g, 0, sizeof(g));
    g.d = d; g.gen = ws.job.gen; g.hA = f.hdr; g.hB = f.hdr + 1; g.st = m.dstate; g.s32 = m.s32;
    if (wide_enabled(ws.pool)) { g.scaled = 1; g.lp_wide = 1; g.l0_scale = kWideL0Scale; g.l0_min = kWideL0Min; }
    return g;
}
static int fast_verify_enqueue(Workspace& ws) {
    const int d = ws.job.d;
    MixedBufs m = mixed_bufs(ws, d);
    FastBufs f = fast_bufs(ws, d);
    f.host_words[13] = 0;
    fast_verify_launch(d, 1, fast_split_args(ws, m, f, d), f.Y, f.Z, f.Rv, f.Pv, f.Ev, m.s32, f.host_vstats, f.host_words, 0, ws.job.stream);
    FAD_HIP_TRY(hipGetLastError());
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, ws.job.stream));
    return FAD_OK;
}
static int fast_enqueue(Workspace& ws, int upto) {
    const int d = ws.job.d;
    hipStream_t stream = ws.job.stream;
    MixedBufs m = mixed_bufs(ws, d);
    FastBufs f = fast_bufs(ws, d);
    if (!f.host_words) return set_error(FAD_ERR_ALLOC, "pinned result area of the fast Frechet chain is missing");
    const int nslots = (d / 32) * (d / 32);
    const int max_low = max_low_of(ws);
    for (int& k = ws.job.k; k < upto; ++k) {
        nsf::SplitArgs g = fast_split_args(ws, m, f, d);
        if (k == 0) {
            // A = C1 C2 (exact) + its statistics + the mean term, then iteration 0: Y1 = Y0 T0, Z1 = T0
            nsf::I8Args a;
            memset(&a, 0, sizeof(a));
            a.Adig = f.digC[0]; a.Bdig = f.digC[1]; a.d = d; a.gen = ws.job.gen; a.hA = f.hdr; a.hB = f.hdr + 1; a.stats = m.tilestats;
            a.A64 = m.A; a.P = f.P; a.st = m.dstate;
            fast_i8(d, nsf::I8_A, a, stream);
            g.A[0] = f.P; g.B[0] = f.P; g.C[0] = f.Y[1]; g.C[1] = f.Z[1]; g.Cdig[0] = f.digY[1]; g.Cdig_t[0] = f.digYt[1];
            g.A64 = m.A; g.statsA = m.tilestats;
            fast_split(d, nsf::SP_FIRST, g, stream, 1);
            continue;
        }
        const int cur = k & 1;
        // T = (3I - Z Y)/2 (scaled steps: 1.5 mu I - 0.5 mu^3 Z Y, mu from the device) and the residual partials of iteration k
        g.A[0] = f.Z[cur]; g.B[0] = f.Y[cur]; g.C[0] = f.T; g.alpha = -0.5f; g.beta_eye = 1.5f; g.gamma = 1.0f;
        g.partials = m.partials; g.skip = &m.s32->done; g.k = k;
        fast_split(d, nsf::SP_T, g, stream, 1);
        // Y <- Y T, Z <- T Z + the check of iteration k as an extra workgroup
        g = fast_split_args(ws, m, f, d);
        g.A[0] = f.Y[cur]; g.B[0] = f.T; g.C[0] = f.Y[cur ^ 1];
        g.A[1] = f.T; g.B[1] = f.Z[cur]; g.C[1] = f.Z[cur ^ 1];
        g.Cdig[0] = f.digY[cur ^ 1]; g.Cdig_t[0] = f.digYt[cur ^ 1];
        g.skip = &m.s32->upd_skip[k & 1];
        g.k = k; g.max_low = max_low; g.nslots = nslots; g.chk_partials = m.partials;
        g.thr_pred = pred_threshold(ws.pool, d);
        fast_split(d, nsf::SP_U, g, stream, 1);
    }
    // exact correction on the final iterate (which of the ping-pong buffers: known on the device only)
    nsf::I8Args a;
    memset(&a, 0, sizeof(a));
    a.Adig = f.digY[0]; a.Bdig = f.digYt[0]; a.Adig_alt = f.digY[1]; a.Bdig_alt = f.digYt[1]; a.sel = &m.s32->final_iter;
    a.d = d; a.gen = ws.job.gen; a.hA = f.hdr; a.hB = f.hdr + 1; a.skip = &m.s32->skip_corr; a.stats = f.host_stats; a.st = m.dstate; a.A64in = m.A;
    a.Y[0] = f.Y[0]; a.Y[1] = f.Y[1]; a.Z[0] = f.Z[0]; a.Z[1] = f.Z[1];
    a.s32 = m.s32; a.host_words = f.host_words; a.host_vals = f.host_vals;
    const bool wide = wide_enabled(ws.pool);
    if (wide) { a.Rv = f.Rv; a.scaled = 1; }
    f.host_words[12] = 0;                          // (the kernel stamps the snapshot with this score's token)
    f.host_words[13] = 0;                          // (... and the verification its own)
    fast_i8(d, nsf::I8_G, a, stream);
    FAD_HIP_TRY(hipGetLastError());
    if (!ws.done_ev) FAD_HIP_TRY(hipEventCreateWithFlags(&ws.done_ev, hipEventDisableTiming));
    // a thread whose last score needed the verification products gets them behind the correction at once (the LAUNCH count follows the
    // history, never the value: a score the norm bound accepts ignores the record)
    if (wide && ws.pool && ws.pool->lp_verify) return fast_verify_enqueue(ws);
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, stream));
    return FAD_OK;
}

// The closing decision, on the host (the enqueued chain has been waited for): reduce the correction's per-tile partials, bound the
// neglected terms, accept / reject -- what ns32_finish does on the device for the float32 chain, minus a launch.
// hw / hv / hsx: what nsf_i8<G> left for ONE problem.  -> status 1 accepted, 2 rejected, 4 a predicted final iterate was rejected
// (the iteration may go on from it), 0 not finished yet.
// hvx (pairs on the wide chain, else nullptr): the verification record of SP_V3, [nb * nb][kVerStats]; valid when hw[13] carries the token.
// -> additionally status 5: the norm bound rejects the correction and no verification record is there yet -- enqueue the two launches.
static void fast_decide_one(const int* hw, const double* hv, const double* hsx, int nb, MixedResult* out, const double* hvx = nullptr,
                            int gen = 0, int max_low = kMaxLow) {
    MixedResult o;
    memset(&o, 0, sizeof(o));
    const bool bad = hw[0] != 0;
    o.status = 0; o.iters = hw[7]; o.decided_at = hw[8]; o.nonfinite = hw[2]; o.too_few0 = hw[3]; o.too_few1 = hw[4];
    o.c = hv[0]; o.tr1 = hv[1]; o.tr2 = hv[2]; o.mean_term = hv[3];
    // (the float64 route never starts from THIS chain's product: A was formed from covariances on a fixed-point grid of 2^-41,
    //  fine for the flat spectra the chain accepts, but the small eigenvalues of the spectra it gives up on move with a perturbation
    //  divided by their own square root -- measured 4e-3 of a FAD on a k^-3 spectrum)
    o.prepared = 0;
    const bool ok = hw[5] != 0, failed = hw[6] != 0, strict = hw[9] != 0;
    if (bad || hw[1] || failed) {
        o.status = 2;                                // bad / zero product or the low-precision leg gave up: the float64 route decides
    } else if (ok && !hw[11]) {
        double corr = 0.0, r2 = 0.0, tr = 0.0, zinf = 0.0, zone = 0.0;
        for (int t = 0; t < nb * nb; ++t) { corr += hsx[nsf::kTileStats * t]; r2 += hsx[nsf::kTileStats * t + 1]; tr += hsx[nsf::kTileStats * t + 2]; }
        const double* zmax = hsx + (size_t)nsf::kTileStats * nb * nb;     // [ty * nb + tx]: (row part, column part) of |Z|, rows block tx, columns block ty
        for (int x = 0; x < nb; ++x) {
            double rs = 0.0, cs = 0.0;
            for (int y = 0; y < nb; ++y) { rs += zmax[2 * (y * nb + x)]; cs += zmax[2 * (x * nb + y) + 1]; }
            if (rs > zinf) zinf = rs;                // >= the largest row sum of |Z| over row block x
            if (cs > zone) zone = cs;                // >= the largest column sum over column block x
        }
        const int fi = hw[7];
        double res = hv[4 + ns32_slot(fi)];
        if (hw[8] == fi - 1) { const double rp = hv[4 + ns32_slot(fi - 1)]; res = 0.75 * rp * rp + 0.25 * rp * rp * rp; if (res < 2e-6) res = 2e-6; }
        const double zn = std::sqrt(zinf * zone), rn = std::sqrt(r2);
        double trs = tr + 0.5 * corr;
        double est = zn * zn * zn * rn * rn / 8.0 + zn * res * rn / 2.0;
        bool finite = std::isfinite(trs) && std::isfinite(est);
        // accepted when the bound on the neglected terms is below 1e-9 of the trace, or moves the DISTANCE by less than 1e-5 of
        // itself (10x inside the 1e-4 bar AS A BOUND: it overestimates the true error 10..10^5 times, most for spread spectra
        // where the norm bound of Z is 3-4x its 2-norm and enters cubed -- a song of 2 D frames, condition 400: bound 2e-8 of the
        // trace, true error 2e-13, scripts/ns_emulate_split.py)
        double fad = o.mean_term + o.tr1 + o.tr2 - 2.0 * std::sqrt(o.c) * trs;
        bool accept = finite && (est <= 1e-9 * std::fabs(trs) || 2.0 * std::sqrt(o.c) * est <= 1e-5 * std::fabs(fad));
        const bool scaled = hvx && hw[14] != 0;
        // a PREDICTED final iterate the correction cannot absorb while the residual behind the prediction was still far from the
        // float32-class floor: the iteration goes on from it (cheaper than verifying an iterate that is not there yet).  A scaled chain whose
        // residual has reached the floor changes nothing by iterating: that one is verified.
        const bool predicted = hw[8] == fi - 1 && fi >= 1;
        const double r_pred = predicted ? hv[4 + ns32_slot(fi - 1)] : 0.0;
        // (the rule's own threshold lets r_pred be at most ~0.06-0.08: above 0.1 the prediction was forced -- FAD_FRECHET_PRED_THR -- and the
        //  iterate is nowhere near; r05c: a threshold of 2e-2 sent every k^-1 pair round the go-on loop, 0.26 -> 0.32 ms)
        const bool can_go_on = predicted && !strict && fi + 1 < max_low && (!scaled || r_pred > 0.1);
        bool need_verify = false;
        if (!accept && finite && hvx && !can_go_on) {
            // The norm bound says nothing for ill-conditioned products (||Z|| ~ 500 for a k^-1 spectrum, cubed).  The verification products
            // measure what it bounds (ns_fast.h): with P = Z R and E = I - Z Y,  1/2 tr(E P) completes the first-order term (Z is only an
            // approximate inverse of Y), and 1/8 |tr(Z P P)| ESTIMATES the second-order one -- it overestimates the commuting part and was
            // seen 0.7 .. 10x the true error (scripts/ns_emulate_verify.py: k^-0.25 .. k^-1.25, D = 512), hence the factor 4; what remains
            // is O(||E||^2 ||P||).  Accepted when that (factor included) moves the distance by less than 4e-6 of itself: ~1e-6 expected.
            if (hw[13] == gen) {
                const double inv = 1.0 / ((double)nsf::kVerScale * (double)nsf::kVerScale);
                double qp = 0.0, ep = 0.0, pp = 0.0, ee = 0.0;
                for (int t = 0; t < nb * nb; ++t) { qp += hvx[nsf::kVerStats * t]; ep += hvx[nsf::kVerStats * t + 1]; pp += hvx[nsf::kVerStats * t + 2]; ee += hvx[nsf::kVerStats * t + 3]; }
                qp *= inv; ep *= inv; pp *= inv; ee *= inv;
                const double trs_v = trs + 0.5 * ep;
                const double est_v = 4.0 * std::fabs(qp) / 8.0 + ee * std::sqrt(pp);
                const double fad_v = o.mean_term + o.tr1 + o.tr2 - 2.0 * std::sqrt(o.c) * trs_v;
                const bool fin_v = std::isfinite(trs_v) && std::isfinite(est_v);
                if (fin_v && (est_v <= 1e-9 * std::fabs(trs_v) || 2.0 * std::sqrt(o.c) * est_v <= 4e-6 * std::fabs(fad_v))) {
                    accept = true; trs = trs_v; est = est_v; fad = fad_v; o.pad = 1;      // (pad = 1: accepted on the verification record)
                }
            } else {
                need_verify = true;
            }
        }
        o.tr_scaled = trs; o.res = res; o.est = est;
        o.status = accept ? 1 : 2;
        if (!accept && finite && can_go_on) o.status = 4;
        else if (!accept && need_verify) o.status = 5;
    }
    *out = o;
}

static int fast_decide(Workspace& ws) {
    const int d = ws.job.d, nb = d / 32;
    MixedBufs m = mixed_bufs(ws, d);
    FastBufs f = fast_bufs(ws, d);
    if (f.host_words[12] != ws.job.gen) return set_error(FAD_ERR_HIP, "the correction kernel of the fast Frechet chain left no result");
    const bool wide = wide_enabled(ws.pool);
    fast_decide_one(f.host_words, f.host_vals, f.host_stats, nb, m.hres, wide ? f.host_vstats : nullptr, ws.job.gen, max_low_of(ws));
    if (m.hres->status == 4) {
        // the iterate was taken as final on a PREDICTED residual and the correction cannot absorb it: nothing is lost --
        // (Y_f, Z_f) are intact, the iteration goes on from there and only the float32 floor ends it now
        hipLaunchKernelGGL(nsf::nsf_rearm, dim3(1), dim3(64), 0, ws.job.stream, m.s32);
        FAD_HIP_TRY(hipGetLastError());
    }
    return FAD_OK;
}

// ==========================================================================================
// The same chain for a BATCH of songs against one baseline (fad_frechet_batched_vs_baseline, songs with at least D + 1 frames:
// every song is a full D x D problem): tr sqrt(Sigma_b Sigma_s) for B songs in eight launches of B times the workgroups.
// Replaces the per-song scipy.linalg.sqrtm / eig of fadtk/fad.py:373-378 for those songs; songs whose product the chain does not
// accept (spread spectra, non-finite input) are handed back to the float64 routes.  D in {128, 256, 384, 512, 768, 1024}.
// ==========================================================================================
bool fast_song_dim(int d) { return d == 128 || d == 256 || d == 384 || d == 512 || d == 768 || d == 1024; }

struct SongBlock {                               // byte offsets inside one song's device block, and its size
    size_t hdr, st, s32, partials, stats, A64, P, Y[2], Z[2], T, digS, digY[2], digYt[2], stride;
};
static SongBlock song_block(int d) {
    const size_t dd = (size_t)d * d, nb = (size_t)d / 32;
    SongBlock b; size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    b.hdr = take(sizeof(nsf::MatHdr)); b.st = take(sizeof(NsState)); b.s32 = take(sizeof(Ns32State));
    b.partials = take(nb * nb * sizeof(double)); b.stats = take(nsf::kTileStats * nb * nb * sizeof(double));
    b.A64 = take(8 * dd); b.P = take(8 * dd);
    b.Y[0] = take(8 * dd); b.Y[1] = take(8 * dd); b.Z[0] = take(8 * dd); b.Z[1] = take(8 * dd); b.T = take(8 * dd);
    b.digS = take(6 * dd);
    b.digY[0] = take(6 * dd); b.digYt[0] = take(6 * dd); b.digY[1] = take(6 * dd); b.digYt[1] = take(6 * dd);
    b.stride = o;
    return b;
}
// per problem in pinned host memory: [kHostVals doubles | (kTileStats + 2) nb^2 doubles | kHostWords ints | kVerStats nb^2 doubles]
static size_t host_vstats_off(int d) {
    const size_t nb = (size_t)d / 32;
    return (nsf::kHostVals + (nsf::kTileStats + 2) * nb * nb) * sizeof(double) + ((nsf::kHostWords * sizeof(int) + 7) & ~(size_t)7);
}
static size_t song_host_stride(int d) {
    const size_t nb = (size_t)d / 32;
    return (host_vstats_off(d) + nsf::kVerStats * nb * nb * sizeof(double) + 63) & ~(size_t)63;
}
int64_t fast_songs_capacity(int d, size_t budget_bytes) {
    const int64_t n = (int64_t)(budget_bytes / song_block(d).stride);
    return n < 1 ? 1 : (n > 16384 ? 16384 : n);
}

// covs: B covariances [d x d] float64 on the device; -> tr_sqrt[b] and ok[b] (1: accepted, 0: hand the song to the float64 routes)
int fast_songs(int d, int64_t B, const double* dcov_b, const double* covs, hipStream_t st, Workspace& ws,
               std::vector<double>& tr_sqrt, std::vector<char>& ok, int device, const SongKnobs& knobs) {
    // knobs.big_min (FAD_SONG_BIG) = smallest batch that iterates on the 128 x 128 tiles of ns_fast_big.h (default 8: a handful of songs
    // fills the chip only on 32 x 32 tiles; 0 = never; tests force either kernel family on the same songs)
    const bool big = knobs.big_min > 0 && B >= knobs.big_min;
    // knobs.res (FAD_SONG_RES) = 0: D = 128 iterates through the batched kernels like the other dimensions (tests compare)
    const bool resident = d == 128 && knobs.res != 0;
    const bool res_full = resident && knobs.res != 1;            // 1: only the iteration resident; default: the exact products too
    // scaled Newton-Schulz steps per song (ns_check.h): the 128 x 128-tile family only (iteration 0, T and U all run there for d >= 256)
    const bool scaled_steps = knobs.scaled && big && d >= 256;
    const bool scaled_res = knobs.scaled && resident;                 // ... and the resident kernel of D = 128
    const size_t dd = (size_t)d * d;
    const int nb = d / 32;
    const SongBlock L = song_block(d);
    const size_t hs = song_host_stride(d);
    FAD_TRY(ws.fast_songs.reserve(512 + 6 * dd + (size_t)B * L.stride));
    if (!ws.fast_songs_pin || ws.fast_songs_pin_cap < (size_t)B * hs) {
        if (ws.fast_songs_pin) (void)hipHostFree(ws.fast_songs_pin);
        ws.fast_songs_pin = nullptr; ws.fast_songs_pin_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&ws.fast_songs_pin, (size_t)B * hs + 4096, hipHostMallocDefault));
        ws.fast_songs_pin_cap = (size_t)B * hs + 4096;
    }
    char* base = static_cast<char*>(ws.fast_songs.p);
    nsf::MatHdr* hdr_b = reinterpret_cast<nsf::MatHdr*>(base);
    uint4* dig_b = reinterpret_cast<uint4*>(base + 512);
    char* blk = base + 512 + ((6 * dd + 255) & ~(size_t)255);
    char* hpin = static_cast<char*>(ws.fast_songs_pin);
    double* h_vals = reinterpret_cast<double*>(hpin);
    double* h_stats = h_vals + nsf::kHostVals;
    int* h_words = reinterpret_cast<int*>(h_stats + (size_t)(nsf::kTileStats + 2) * nb * nb);
    auto at = [&](size_t off) { return blk + off; };
    auto mat = [&](size_t off) { nsf::SplitMat m; m.a = reinterpret_cast<uint4*>(at(off)); m.at = reinterpret_cast<uint4*>(at(off + 4 * dd)); return m; };
    const int gen = ++ws.fast_gen;
    if (ws.fast_gen > (1 << 30)) ws.fast_gen = 1;
    // fresh headers carry no stale token
    FAD_HIP_TRY(hipMemsetAsync(hdr_b, 0, sizeof(nsf::MatHdr), st));
    FAD_HIP_TRY(hipMemset2DAsync(at(L.hdr), L.stride, 0, sizeof(nsf::MatHdr), (size_t)B, st));

    nsf::PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.cov_in[0] = dcov_b; pa.cov_in[1] = covs; pa.d = d; pa.ddof = 1; pa.gen = gen; pa.mean_dtype = -1;
    pa.dig[0] = dig_b; pa.dig[1] = reinterpret_cast<uint4*>(at(L.digS));
    pa.st = reinterpret_cast<NsState*>(at(L.st)); pa.hdr[0] = hdr_b; pa.hdr[1] = reinterpret_cast<nsf::MatHdr*>(at(L.hdr));
    pa.batch = 1; pa.pstride = (int64_t)L.stride;
    hipLaunchKernelGGL(nsf::nsf_prepare, dim3((unsigned)(dd / 2048), (unsigned)(1 + B)), dim3(512), 0, st, pa);

    NsState* st0 = reinterpret_cast<NsState*>(at(L.st));
    Ns32State* s32_0 = reinterpret_cast<Ns32State*>(at(L.s32));
    double* partials = reinterpret_cast<double*>(at(L.partials));
    double* stats = reinterpret_cast<double*>(at(L.stats));
    double* A64 = reinterpret_cast<double*>(at(L.A64));
    const nsf::SplitMat P = mat(L.P), Y[2] = {mat(L.Y[0]), mat(L.Y[1])}, Z[2] = {mat(L.Z[0]), mat(L.Z[1])}, T = mat(L.T);
    uint4* digY[2] = {reinterpret_cast<uint4*>(at(L.digY[0])), reinterpret_cast<uint4*>(at(L.digY[1]))};
    uint4* digYt[2] = {reinterpret_cast<uint4*>(at(L.digYt[0])), reinterpret_cast<uint4*>(at(L.digYt[1]))};
    auto split_args = [&]() {
        nsf::SplitArgs g;
        memset(&g, 0, sizeof(g));
        g.d = d; g.gen = gen; g.hA = hdr_b; g.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); g.pstride = (int64_t)L.stride;
        g.st = st0; g.s32 = s32_0;
        g.scaled = scaled_steps ? 1 : 0; g.l0_scale = knobs.l0_scale;
        return g;
    };
    {
        nsf::I8Args a;
        memset(&a, 0, sizeof(a));
        a.Adig = dig_b; a.Bdig = reinterpret_cast<uint4*>(at(L.digS)); a.d = d; a.gen = gen; a.hA = hdr_b;
        a.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); a.pstride = (int64_t)L.stride; a.stats = stats; a.A64 = A64; a.P = P; a.st = st0;
        if (res_full) { /* nsf_res128<FULL> forms the product itself */ }
        else if (big && d >= 256) FAD_TRY(fast_i8_big(d, nsf::I8_A, a, st, (unsigned)B, device));
        else fast_i8(d, nsf::I8_A, a, st, (unsigned)B);
        if (resident) {
            // D = 128: the whole iteration of a song in one workgroup (ns_fast_res.h)
            static std::atomic<unsigned> ready{0};
            if (device >= 0 && device < 32 && !(ready.load(std::memory_order_acquire) & (1u << device))) {
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_res128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kResLds));
                FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nsf::nsf_res128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nsf::kResLdsFull));
                ready.fetch_or(1u << device, std::memory_order_release);
            }
            nsf::ResArgs r;
            memset(&r, 0, sizeof(r));
            r.gen = gen; r.max_low = kMaxLow; r.thr_pred = pred_threshold(ws.pool, d); r.hA = hdr_b; r.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr));
            r.pstride = (int64_t)L.stride; r.A64 = A64; r.statsA = stats; r.st = st0; r.s32 = s32_0;
            r.Y[0] = Y[0]; r.Y[1] = Y[1]; r.Z[0] = Z[0]; r.Z[1] = Z[1];
            r.scaled = scaled_res ? 1 : 0; r.l0_scale = 2.0 * knobs.l0_scale;          // (D = 128: 6 iterations at the full estimate, 7 at half of it)
            if (res_full) {
                // ... and the two exact products with it: A = Sigma_b Sigma_s in front, the correction behind; the host record is this kernel's
                r.Adig = dig_b; r.Bdig = reinterpret_cast<uint4*>(at(L.digS)); r.hstride = (int64_t)hs;
                r.stats = h_stats; r.host_words = h_words; r.host_vals = h_vals;
                for (int64_t b = 0; b < B; ++b) reinterpret_cast<int*>(reinterpret_cast<char*>(h_words) + b * hs)[12] = 0;
                hipLaunchKernelGGL(nsf::nsf_res128<true>, dim3((unsigned)B), dim3(256), nsf::kResLdsFull, st, r);
            } else {
                hipLaunchKernelGGL(nsf::nsf_res128<false>, dim3((unsigned)B), dim3(256), nsf::kResLds, st, r);
            }
        } else {
            nsf::SplitArgs g = split_args();
            g.A[0] = P; g.B[0] = P; g.C[0] = Y[1]; g.C[1] = Z[1]; g.A64 = A64; g.statsA = stats;      // (no digit planes: nsf_digitize, below)
            if (big && d >= 256) FAD_TRY(fast_split_big(d, nsf::SP_FIRST, g, st, (unsigned)B, device));
            else fast_split(d, nsf::SP_FIRST, g, st, (unsigned)B);
        }
    }
    tr_sqrt.assign((size_t)B, 0.0); ok.assign((size_t)B, 0);
    std::vector<char> settled((size_t)B, 0);
    // iterations 1..8 blind (a song of 2 D .. 20 D frames needs 7-11: its product has a condition number of a few hundred), then the
    // correction for the songs whose check finished them; if any is still iterating, the rest of the budget in one go (every song
    // stops itself: the launches of a finished song exit at once)
    const bool trace = knobs.trace;
    int k = 1, upto = resident ? 1 : 9;                 // (resident: nothing left to launch but the correction)
    for (;;) {
        for (; k < upto; ++k) {
            const int cur = k & 1;
            nsf::SplitArgs g = split_args();
            g.A[0] = Z[cur]; g.B[0] = Y[cur]; g.C[0] = T; g.alpha = -0.5f; g.beta_eye = 1.5f; g.gamma = 1.0f;
            g.partials = partials; g.skip = &s32_0->done; g.k = k;
            if (big) FAD_TRY(fast_split_big(d, nsf::SP_T, g, st, (unsigned)B, device));
            else fast_split(d, nsf::SP_T, g, st, (unsigned)B);
            g = split_args();
            g.A[0] = Y[cur]; g.B[0] = T; g.C[0] = Y[cur ^ 1]; g.A[1] = T; g.B[1] = Z[cur]; g.C[1] = Z[cur ^ 1];
            g.skip = &s32_0->upd_skip[k & 1];
            g.k = k; g.max_low = kMaxLow; g.nslots = big ? fast_big_t_slots(d, (unsigned)B) : nb * nb; g.chk_partials = partials; g.thr_pred = pred_threshold(ws.pool, d);
            if (big) FAD_TRY(fast_split_big(d, nsf::SP_U, g, st, (unsigned)B, device));
            else fast_split(d, nsf::SP_U, g, st, (unsigned)B);
        }
        if (!res_full) {
            nsf::DigArgs dg;
            memset(&dg, 0, sizeof(dg));
            dg.d = d; dg.gen = gen; dg.hA = hdr_b; dg.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); dg.pstride = (int64_t)L.stride; dg.s32 = s32_0;
            dg.Y[0] = Y[0]; dg.Y[1] = Y[1]; dg.dig[0] = digY[0]; dg.dig[1] = digY[1]; dg.dig_t[0] = digYt[0]; dg.dig_t[1] = digYt[1];
            hipLaunchKernelGGL(nsf::nsf_digitize, dim3((unsigned)((dd / 16 + 255) / 256), 2, (unsigned)B), dim3(256), 0, st, dg);
        }
        if (!res_full) {
            nsf::I8Args a;
            memset(&a, 0, sizeof(a));
            a.Adig = digY[0]; a.Bdig = digYt[0]; a.Adig_alt = digY[1]; a.Bdig_alt = digYt[1]; a.sel = &s32_0->final_iter;
            a.d = d; a.gen = gen; a.hA = hdr_b; a.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdr)); a.pstride = (int64_t)L.stride; a.hstride = (int64_t)hs;
            a.skip = &s32_0->skip_corr; a.stats = h_stats; a.st = st0; a.A64in = A64;
            a.Y[0] = Y[0]; a.Y[1] = Y[1]; a.Z[0] = Z[0]; a.Z[1] = Z[1]; a.s32 = s32_0; a.host_words = h_words; a.host_vals = h_vals;
            for (int64_t b = 0; b < B; ++b) reinterpret_cast<int*>(reinterpret_cast<char*>(h_words) + b * hs)[12] = 0;
            if (big && d >= 256) FAD_TRY(fast_i8_big(d, nsf::I8_G, a, st, (unsigned)B, device));
            else fast_i8(d, nsf::I8_G, a, st, (unsigned)B);
        }
        FAD_HIP_TRY(hipGetLastError());
        FAD_HIP_TRY(hipStreamSynchronize(st));
        bool pending = false;
        for (int64_t b = 0; b < B; ++b) {
            if (settled[b]) continue;
            const int* hw = reinterpret_cast<const int*>(reinterpret_cast<const char*>(h_words) + b * hs);
            const double* hv = reinterpret_cast<const double*>(reinterpret_cast<const char*>(h_vals) + b * hs);
            const double* hx = reinterpret_cast<const double*>(reinterpret_cast<const char*>(h_stats) + b * hs);
            if (hw[12] != gen) return set_error(FAD_ERR_HIP, "the correction kernel of the batched fast chain left no result for song %lld", (long long)b);
            MixedResult r;
            fast_decide_one(hw, hv, hx, nb, &r);
            if (r.status == 0 && k < kMaxLow && !resident) { pending = true; continue; }      // not finished yet: more iterations for this song
            settled[b] = 1;
            if (r.status == 1) { ok[b] = 1; tr_sqrt[b] = std::sqrt(r.c) * r.tr_scaled; }
            if (trace)
                fprintf(stderr, "[fad fast songs] song %lld: status %d iters %d decided_at %d res %.3e est %.3e tr %.6e c %.3e (words bad %d done %d ok %d failed %d skipped %d)\n",
                        (long long)b, r.status, r.iters, r.decided_at, r.res, r.est, r.tr_scaled, r.c, hw[0], hw[1], hw[5], hw[6], hw[11]);
            if (trace && b == 0) {
                fprintf(stderr, "[fad fast songs] song 0 residuals:");
                for (int q = 1; q < 16 && q <= r.iters; ++q) fprintf(stderr, " %.3e", hv[4 + q]);
                fprintf(stderr, "\n");
            }
        }
        if (!pending) break;
        upto = kMaxLow;
    }
    return FAD_OK;
}


// ==========================================================================================
// The chain for B INDEPENDENT PAIRS of packed moments in one sequence of launches (fad_frechet_from_moments_multi_begin): the
// same eight kernels as a single pair with B times the workgroups -- a launch of this chain costs ~4 us before it does
// anything and its workgroups are latency-bound, so B scores cost little more than one (bench.py keeps several scores in
// flight: their chains are enqueued as ONE batch).  Every buffer of pair b lives b * stride bytes behind pair 0's.
// ==========================================================================================
struct PairBlock {
    size_t hdrA, hdrB, st, s32, partials, stats, A64, P, Y[2], Z[2], T, digA, digB, digY[2], digYt[2], mus, covs, Rv, Pv, Ev, stride;
};
static PairBlock pair_block(int d) {
    const size_t dd = (size_t)d * d, nb = (size_t)d / 32;
    PairBlock b; size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    b.hdrA = take(sizeof(nsf::MatHdr)); b.hdrB = take(sizeof(nsf::MatHdr)); b.st = take(sizeof(NsState)); b.s32 = take(sizeof(Ns32State));
    b.partials = take(nb * nb * sizeof(double)); b.stats = take(nsf::kTileStats * nb * nb * sizeof(double));
    b.A64 = take(8 * dd); b.P = take(8 * dd);
    b.Y[0] = take(8 * dd); b.Y[1] = take(8 * dd); b.Z[0] = take(8 * dd); b.Z[1] = take(8 * dd); b.T = take(8 * dd);
    b.digA = take(6 * dd); b.digB = take(6 * dd);
    b.digY[0] = take(6 * dd); b.digYt[0] = take(6 * dd); b.digY[1] = take(6 * dd); b.digYt[1] = take(6 * dd);
    b.mus = take(2 * (size_t)d * sizeof(double)); b.covs = take(2 * dd * sizeof(double));
    b.Rv = take(8 * dd); b.Pv = take(8 * dd); b.Ev = take(8 * dd);      // verification planes (ns_fast.h: SP_V2 / SP_V3)
    b.stride = o;
    return b;
}

// the two verification launches for the B pairs of the slot's batch (32 x 32-tile kernels: a rare, short stage)
static int pairs_verify_enqueue(Workspace& ws, int d, int B, hipStream_t st) {
    const size_t dd = (size_t)d * d;
    const PairBlock L = pair_block(d);
    const size_t hs = song_host_stride(d);
    char* blk = static_cast<char*>(ws.fast_pairs.p);
    char* hpin = static_cast<char*>(ws.fast_pairs_pin);
    auto at = [&](size_t off) { return blk + off; };
    auto mat = [&](size_t off) { nsf::SplitMat m; m.a = reinterpret_cast<uint4*>(at(off)); m.at = reinterpret_cast<uint4*>(at(off + 4 * dd)); return m; };
    nsf::SplitArgs g;
    memset(&g, 0, sizeof(g));
    g.d = d; g.gen = ws.multi.gen; g.hA = reinterpret_cast<nsf::MatHdr*>(at(L.hdrA)); g.hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdrB));
    g.pstride = (int64_t)L.stride; g.astride = (int64_t)L.stride;
    g.st = reinterpret_cast<NsState*>(at(L.st)); g.s32 = reinterpret_cast<Ns32State*>(at(L.s32));
    const nsf::SplitMat Y[2] = {mat(L.Y[0]), mat(L.Y[1])}, Z[2] = {mat(L.Z[0]), mat(L.Z[1])};
    const int nb = d / 32;
    int* h_words = reinterpret_cast<int*>(hpin + (nsf::kHostVals + (size_t)(nsf::kTileStats + 2) * nb * nb) * sizeof(double));
    double* h_vstats = reinterpret_cast<double*>(hpin + host_vstats_off(d));
    for (int b = 0; b < B; ++b) reinterpret_cast<int*>(reinterpret_cast<char*>(h_words) + (size_t)b * hs)[13] = 0;
    fast_verify_launch(d, (unsigned)B, g, Y, Z, mat(L.Rv), mat(L.Pv), mat(L.Ev), g.s32, h_vstats, h_words, (int64_t)hs, st);
    FAD_HIP_TRY(hipGetLastError());
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, st));
    return FAD_OK;
}

// enqueue: K1 (pairs mode) .. K8 for B pairs; nothing is waited for
static int pairs_enqueue(Workspace& ws, int d, int B, const fad_moments_t* const* h1, const fad_moments_t* const* h2, int ddof,
                         int mean_dtype, hipStream_t st) {
    const size_t dd = (size_t)d * d;
    const int nb = d / 32;
    // FAD_PAIRS_BIG = smallest batch whose products run on the 128 x 128 / 128 x 64 tiles of ns_fast_big.h (their operand traffic per
    // block is a tenth of the 32 x 32 kernels'; below, too few workgroups to fill the chip); read per call, 0 = never
    const char* big_env = getenv("FAD_PAIRS_BIG");
    const long big_min = big_env ? atol(big_env) : 3;
    const bool big = big_min > 0 && B >= big_min && d >= 256;
    // (scaled steps: iteration 0 of the 32 x 32-tile family takes them as well; the batch only needs them on ONE family at a time)
    const bool wide = wide_enabled(ws.pool);
    const int max_low = wide ? kMaxLowWide : kMaxLow;
    const int device = ws.job.device;
    const PairBlock L = pair_block(d);
    const size_t hs = song_host_stride(d);
    void* const before = ws.fast_pairs.p;
    FAD_TRY(ws.fast_pairs.reserve((size_t)(B > 16 ? kMaxMultiPairs : B > 8 ? 16 : B > 4 ? 8 : 4) * L.stride + 256));      // (room for a full batch at once: no regrowth between calls)
    if (!ws.fast_pairs_pin || ws.fast_pairs_pin_cap < (size_t)B * hs) {
        if (ws.fast_pairs_pin) (void)hipHostFree(ws.fast_pairs_pin);
        ws.fast_pairs_pin = nullptr; ws.fast_pairs_pin_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&ws.fast_pairs_pin, (size_t)kMaxMultiPairs * hs + 4096, hipHostMallocDefault));
        ws.fast_pairs_pin_cap = (size_t)kMaxMultiPairs * hs + 4096;
    }
    char* blk = static_cast<char*>(ws.fast_pairs.p);
    char* hpin = static_cast<char*>(ws.fast_pairs_pin);
    double* h_vals = reinterpret_cast<double*>(hpin);
    double* h_stats = h_vals + nsf::kHostVals;
    int* h_words = reinterpret_cast<int*>(h_stats + (size_t)(nsf::kTileStats + 2) * nb * nb);
    auto at = [&](size_t off) { return blk + off; };
    auto mat = [&](size_t off) { nsf::SplitMat m; m.a = reinterpret_cast<uint4*>(at(off)); m.at = reinterpret_cast<uint4*>(at(off + 4 * dd)); return m; };
    const int gen = ++ws.fast_gen;
    if (ws.fast_gen > (1 << 30)) ws.fast_gen = 1;
    ws.multi.gen = gen;
    if (ws.fast_pairs.p != before) {               // fresh headers carry no stale token
        FAD_HIP_TRY(hipMemset2DAsync(at(L.hdrA), L.stride, 0, sizeof(nsf::MatHdr), (size_t)B, st));
        FAD_HIP_TRY(hipMemset2DAsync(at(L.hdrB), L.stride, 0, sizeof(nsf::MatHdr), (size_t)B, st));
    }
    nsf::MatHdr* hA = reinterpret_cast<nsf::MatHdr*>(at(L.hdrA));
    nsf::MatHdr* hB = reinterpret_cast<nsf::MatHdr*>(at(L.hdrB));
    NsState* st0 = reinterpret_cast<NsState*>(at(L.st));
    Ns32State* s32_0 = reinterpret_cast<Ns32State*>(at(L.s32));
    double* partials = reinterpret_cast<double*>(at(L.partials));
    double* stats = reinterpret_cast<double*>(at(L.stats));
    double* A64 = reinterpret_cast<double*>(at(L.A64));
    const nsf::SplitMat P = mat(L.P), Y[2] = {mat(L.Y[0]), mat(L.Y[1])}, Z[2] = {mat(L.Z[0]), mat(L.Z[1])}, T = mat(L.T);
    uint4* digY[2] = {reinterpret_cast<uint4*>(at(L.digY[0])), reinterpret_cast<uint4*>(at(L.digY[1]))};
    uint4* digYt[2] = {reinterpret_cast<uint4*>(at(L.digYt[0])), reinterpret_cast<uint4*>(at(L.digYt[1]))};

    static_assert(2 * kMaxMultiPairs <= nsf::kPrepMaxSets, "PrepArgs holds two sets per pair of a batch");
    nsf::PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    for (int b = 0; b < B; ++b) {
        pa.accs[2 * b] = moments_packed(h1[b]); pa.accs[2 * b + 1] = moments_packed(h2[b]);
        pa.runs[2 * b] = moments_runsum(h1[b]); pa.runs[2 * b + 1] = moments_runsum(h2[b]);
    }
    pa.acc[0] = pa.accs[0]; pa.acc[1] = pa.accs[1];
    pa.d = d; pa.ddof = ddof; pa.gen = gen; pa.mean_dtype = mean_dtype;
    pa.mus = reinterpret_cast<double*>(at(L.mus)); pa.covs = reinterpret_cast<double*>(at(L.covs));
    pa.dig[0] = reinterpret_cast<uint4*>(at(L.digA)); pa.dig[1] = reinterpret_cast<uint4*>(at(L.digB));
    pa.st = st0; pa.hdr[0] = hA; pa.hdr[1] = hB;
    pa.batch = 2; pa.pstride = (int64_t)L.stride; pa.no_covs = 1;
    pa.per = ((dd / 2048) % 4 == 0) ? 4 : 2;              // (d a multiple of 64: d * d / 2048 is even)
    hipLaunchKernelGGL(nsf::nsf_prepare, dim3((unsigned)(dd / 2048 / pa.per + 1), (unsigned)(2 * B)), dim3(512), 0, st, pa);

    auto split_args = [&]() {
        nsf::SplitArgs g;
        memset(&g, 0, sizeof(g));
        g.d = d; g.gen = gen; g.hA = hA; g.hB = hB; g.pstride = (int64_t)L.stride; g.astride = (int64_t)L.stride;
        g.st = st0; g.s32 = s32_0;
        if (wide) { g.scaled = 1; g.lp_wide = 1; g.l0_scale = kWideL0Scale; g.l0_min = kWideL0Min; }
        return g;
    };
    {
        nsf::I8Args a;
        memset(&a, 0, sizeof(a));
        a.Adig = reinterpret_cast<uint4*>(at(L.digA)); a.Bdig = reinterpret_cast<uint4*>(at(L.digB)); a.d = d; a.gen = gen; a.hA = hA; a.hB = hB;
        a.pstride = (int64_t)L.stride; a.astride = (int64_t)L.stride; a.stats = stats; a.A64 = A64; a.P = P; a.st = st0;
        if (big) FAD_TRY(fast_i8_big(d, nsf::I8_A, a, st, (unsigned)B, device));
        else fast_i8(d, nsf::I8_A, a, st, (unsigned)B);
        nsf::SplitArgs g = split_args();
        g.A[0] = P; g.B[0] = P; g.C[0] = Y[1]; g.C[1] = Z[1]; g.A64 = A64; g.statsA = stats;
        if (big) FAD_TRY(fast_split_big(d, nsf::SP_FIRST, g, st, (unsigned)B, device));      // (digit planes of the final Y only: nsf_digitize below)
        else { g.Cdig[0] = digY[1]; g.Cdig_t[0] = digYt[1]; fast_split(d, nsf::SP_FIRST, g, st, (unsigned)B); }
    }
    int want = ws.pool ? ws.pool->lp_iters : 5;
    if (want < 2) want = 2;                          // (a pair that is not through after the blind batch goes the single way)
    if (want > max_low) want = max_low;
    for (int k = 1; k < want; ++k) {
        const int cur = k & 1;
        nsf::SplitArgs g = split_args();
        g.A[0] = Z[cur]; g.B[0] = Y[cur]; g.C[0] = T; g.alpha = -0.5f; g.beta_eye = 1.5f; g.gamma = 1.0f;
        g.partials = partials; g.skip = &s32_0->done; g.k = k;
        if (big) FAD_TRY(fast_split_big(d, nsf::SP_T, g, st, (unsigned)B, device));
        else fast_split(d, nsf::SP_T, g, st, (unsigned)B);
        g = split_args();
        g.A[0] = Y[cur]; g.B[0] = T; g.C[0] = Y[cur ^ 1]; g.A[1] = T; g.B[1] = Z[cur]; g.C[1] = Z[cur ^ 1];
        if (!big) { g.Cdig[0] = digY[cur ^ 1]; g.Cdig_t[0] = digYt[cur ^ 1]; }
        g.skip = &s32_0->upd_skip[k & 1];
        g.k = k; g.max_low = max_low; g.nslots = big ? fast_big_t_slots(d, (unsigned)B) : nb * nb; g.chk_partials = partials; g.thr_pred = pred_threshold(ws.pool, d);
        if (big) FAD_TRY(fast_split_big(d, nsf::SP_U, g, st, (unsigned)B, device));
        else fast_split(d, nsf::SP_U, g, st, (unsigned)B);
    }
    if (big) {
        nsf::DigArgs dg;
        memset(&dg, 0, sizeof(dg));
        dg.d = d; dg.gen = gen; dg.hA = hA; dg.hB = hB; dg.pstride = (int64_t)L.stride; dg.astride = (int64_t)L.stride; dg.s32 = s32_0;
        dg.Y[0] = Y[0]; dg.Y[1] = Y[1]; dg.dig[0] = digY[0]; dg.dig[1] = digY[1]; dg.dig_t[0] = digYt[0]; dg.dig_t[1] = digYt[1];
        hipLaunchKernelGGL(nsf::nsf_digitize, dim3((unsigned)((dd / 16 + 255) / 256), 2, (unsigned)B), dim3(256), 0, st, dg);
    }
    {
        nsf::I8Args a;
        memset(&a, 0, sizeof(a));
        a.Adig = digY[0]; a.Bdig = digYt[0]; a.Adig_alt = digY[1]; a.Bdig_alt = digYt[1]; a.sel = &s32_0->final_iter;
        a.d = d; a.gen = gen; a.hA = hA; a.hB = hB; a.pstride = (int64_t)L.stride; a.astride = (int64_t)L.stride; a.hstride = (int64_t)hs;
        a.skip = &s32_0->skip_corr; a.stats = h_stats; a.st = st0; a.A64in = A64;
        a.Y[0] = Y[0]; a.Y[1] = Y[1]; a.Z[0] = Z[0]; a.Z[1] = Z[1]; a.s32 = s32_0; a.host_words = h_words; a.host_vals = h_vals;
        if (wide) { a.Rv = mat(L.Rv); a.scaled = 1; }
        for (int b = 0; b < B; ++b) { int* w = reinterpret_cast<int*>(reinterpret_cast<char*>(h_words) + (size_t)b * hs); w[12] = 0; w[13] = 0; }
        if (big) FAD_TRY(fast_i8_big(d, nsf::I8_G, a, st, (unsigned)B, device));
        else fast_i8(d, nsf::I8_G, a, st, (unsigned)B);
    }
    FAD_HIP_TRY(hipGetLastError());
    if (!ws.done_ev) FAD_HIP_TRY(hipEventCreateWithFlags(&ws.done_ev, hipEventDisableTiming));
    // (a thread whose last scores needed the verification products gets them behind the correction at once: fast_enqueue)
    if (wide && ws.pool && ws.pool->lp_verify) return pairs_verify_enqueue(ws, d, B, st);
    FAD_HIP_TRY(hipEventRecord(ws.done_ev, st));
    return FAD_OK;
}

// Enqueue the whole low-precision chain of ONE problem on `stream` (nothing is waited for): C1 C2, statistics, scale,
// iteration 0, the blind batch of iterations, the closing kernels.  The state words must have been cleared.
static int mixed_begin(const NsProblem& pb, int device, hipStream_t stream, Workspace& ws) {
    const int d = pb.d;
    const int64_t dd = (int64_t)d * d;
    FAD_TRY(ws.mats.reserve((size_t)(6 * dd) * sizeof(double)));
    FAD_TRY(ws.mats32.reserve((size_t)(5 * dd) * sizeof(float)));
    const size_t hbytes = ws.job.fast ? fast_pinned_bytes(d) : sizeof(NsState) + sizeof(MixedResult);
    if (!ws.pinned || ws.pinned_cap < hbytes) {
        if (ws.pinned) (void)hipHostFree(ws.pinned);
        ws.pinned = nullptr; ws.pinned_cap = 0;
        FAD_HIP_TRY(hipHostMalloc(&ws.pinned, hbytes + 4096, hipHostMallocDefault));
        ws.pinned_cap = hbytes + 4096;
    }
    MixedBufs m = mixed_bufs(ws, d);
    m.hres->status = -1;
    ws.job.d = d; ws.job.device = device; ws.job.stream = stream; ws.job.k = 0;
    if (ws.job.fast) {                           // (nsf_prepare is on the stream already: fast_prepare)
        ws.job.mu1 = pb.mu1; ws.job.mu2 = pb.mu2; ws.job.mean_dtype = pb.mean_dtype;
        int want = ws.pool ? ws.pool->lp_iters : 5;
        if (want < 2) want = 2;
        if (want > max_low_of(ws)) want = max_low_of(ws);
        if (ws.pool && ws.pool->lp_hopeless) want = 1;       // (a pair that does iterate is topped up by mixed_finish)
        return fast_enqueue(ws, want);
    }
    // A = C1 C2 with its tile statistics from the epilogue and the mean term from a spare workgroup (one launch instead of
    // product + ns_tilestats), then the scale
    NsProductExt ext;
    memset(&ext, 0, sizeof(ext));
    ext.stats = m.tilestats; ext.mu1 = pb.mu1; ext.mu2 = pb.mu2; ext.mean_dtype = pb.mean_dtype; ext.st = m.dstate;
    FAD_TRY(gemm_f64_product_stats_launch(d, pb.cov1, pb.cov2, m.A, &m.dstate->done, ext, stream));
    enqueue_ns_prepare(m.tilestats, d, (int)m.nb, pb.mu1, 0, pb.mu2, 0, pb.mean_dtype, m.dstate, 1, m.s32, 1, stream);
    int want = ws.pool ? ws.pool->lp_iters : 5;
    if (want < 2) want = 2;
    if (want > kMaxLow) want = kMaxLow;
    return mixed_enqueue(ws, want);
}

// Wait for the chain, top it up two iterations at a time while the device says "not finished yet".
// -> FAD_OK with res->status 1 (accepted: res holds the pieces) or 2 (run the fp64 iteration).
static int mixed_finish(Workspace& ws, MixedResult* res) {
    MixedBufs m = mixed_bufs(ws, ws.job.d);
    const int max_low = max_low_of(ws);
    bool asked = false;
    for (;;) {
        FAD_HIP_TRY(hipEventSynchronize(ws.done_ev));
        if (ws.job.fast) FAD_TRY(fast_decide(ws));
        if (m.hres->status == 5) {                 // the correction needs the verification products: two launches, one more wait
            if (asked) { m.hres->status = 2; break; }                    // (no record came back: the float64 route)
            asked = true;
            FAD_TRY(fast_verify_enqueue(ws));
            continue;
        }
        if (m.hres->status == 4) {                 // predicted final iterate rejected: go on from it (state re-armed on the device)
            ws.job.k = m.hres->iters;
            m.hres->status = 0;
        } else if (m.hres->status != 0 || ws.job.k >= max_low) {
            break;
        }
        if (ws.job.k >= max_low) break;
        // (the wide chain tops up four iterations at a time: a decaying spectrum needs 10-13, and every trip to the host costs 20-30 us)
        const int step = (ws.job.fast && max_low > kMaxLow) ? 4 : 2;
        const int upto = (ws.job.k + step < max_low) ? ws.job.k + step : max_low;
        FAD_TRY(ws.job.fast ? fast_enqueue(ws, upto) : mixed_enqueue(ws, upto));
    }
    *res = *m.hres;
    if (res->status == 0) res->status = 2;
    if (res->status == 1 && res->decided_at >= 0 && ws.pool) ws.pool->lp_iters = res->decided_at + 1;
    if (ws.pool && ws.job.fast) {
        ws.pool->lp_hopeless = res->status == 2 && res->iters < 0;      // given up before the first check
        ws.pool->lp_verify = res->status == 1 && res->pad == 1;         // accepted on the verification record: the next score gets it unasked
    }
    return FAD_OK;
}

static bool mixed_eligible(Workspace& ws, int d, int max_iter, double tol) {
    Pool* p = ws.pool;
    if (p && p->mixed < 0) { const char* e = getenv("FAD_FRECHET_MIXED"); p->mixed = (e && e[0] == '0') ? 0 : 1; }
    return (!p || p->mixed) && d % 64 == 0 && max_iter <= 0 && tol <= 0.0;
}

// single pair, with the reference's eps fallback; cov/mu are DEVICE pointers
static int frechet_single(int d, const double* cov1, const double* cov2, const double* mu1, const double* mu2,
                          double eps, int max_iter, double tol, int mean_dtype, int device, hipStream_t stream,
                          Workspace& ws, double* out_fad, fad_diag_t* diag, bool check_few) {
    const int64_t dd = (int64_t)d * d;
    NsState* hs = nullptr;
    bool reuse = false;
    NsProblem pb{d, 1, cov1, 0, cov2, 0, mu1, 0, mu2, 0, mean_dtype};
    if (mixed_eligible(ws, d, max_iter, tol)) {
        MixedResult r;
        if (!ws.job.mixed) FAD_TRY(mixed_begin(pb, device, stream, ws));       // (an async job enqueued it already)
        ws.job.mixed = false;
        FAD_TRY(mixed_finish(ws, &r));
        if (check_few && (r.too_few0 || r.too_few1))
            return set_error(FAD_ERR_TOO_FEW_ROWS, "FAD requires at least two embedding window frames in each set");
        if (r.status == 1) {
            const double tr_sqrt = sqrt(r.c) * r.tr_scaled;
            if (out_fad) *out_fad = r.mean_term + r.tr1 + r.tr2 - 2.0 * tr_sqrt;
            if (diag) {
                diag->iters = r.iters + 1; diag->converged = 3; diag->used_eps = 0; diag->route = ws.job.fast ? 2 : 1;
                diag->residual = r.res; diag->scale = r.c; diag->mean_term = r.mean_term; diag->tr1 = r.tr1; diag->tr2 = r.tr2;
                diag->tr_sqrt = tr_sqrt; diag->verified = (ws.job.fast && r.pad == 1) ? 1 : 0; diag->reserved = 0;
            }
            return FAD_OK;
        }
        // rejected: the float64 iteration takes over from the product C1 C2 and the state that was armed for this very
        // problem (nothing in the low-precision leg writes to either) -- unless the chain never got that far (prepared = 0)
        reuse = r.prepared != 0;
    }
    FAD_TRY(run_ns(pb, max_iter, tol, device, stream, ws, &hs, reuse));
    if (check_few && (hs->too_few[0] || hs->too_few[1]))
        return set_error(FAD_ERR_TOO_FEW_ROWS, "FAD requires at least two embedding window frames in each set");
    bool used_eps = false;
    if (hs->nonfinite && eps > 0.0) {
        // fad.py:94-99: add eps to both diagonals and take the root again
        FAD_TRY(ws.stage.reserve((size_t)(4 * dd + 2 * d) * sizeof(double)));     // keeps the first 2dd+2d intact
        double* E1 = static_cast<double*>(ws.stage.p) + 2 * dd + 2 * d;
        double* E2 = E1 + dd;
        FAD_HIP_TRY(hipMemcpyAsync(E1, cov1, dd * sizeof(double), hipMemcpyDeviceToDevice, stream));
        FAD_HIP_TRY(hipMemcpyAsync(E2, cov2, dd * sizeof(double), hipMemcpyDeviceToDevice, stream));
        enqueue_add_diag(E1, d, eps, stream);
        enqueue_add_diag(E2, d, eps, stream);
        enqueue_clear_states(static_cast<NsState*>(ws.small.p), 1, stream);
        NsProblem pe{d, 1, E1, 0, E2, 0, mu1, 0, mu2, 0, mean_dtype};
        FAD_TRY(run_ns(pe, max_iter, tol, device, stream, ws, &hs));
        used_eps = true;
    }
    if (hs->nonfinite) {
        if (diag) { memset(diag, 0, sizeof(*diag)); diag->used_eps = used_eps; diag->residual = hs->res_last; }
        return set_error(FAD_ERR_NOT_FINITE,
                         "sqrt(C1 C2) did not stay finite (NaN/Inf input or a product with negative eigenvalues)");
    }
    const double tr_sqrt = sqrt(hs->c) * hs->tr_last;
    double tr1 = hs->tr1, tr2 = hs->tr2;          // traces refer to the caller's inputs (fad.py:119-120)
    if (used_eps) { tr1 -= eps * d; tr2 -= eps * d; }
    if (out_fad) *out_fad = hs->mean_term + tr1 + tr2 - 2.0 * tr_sqrt;
    if (diag) {
        diag->iters = hs->final_iter + 1; diag->converged = hs->conv; diag->used_eps = used_eps ? 1 : 0;
        diag->route = 0; diag->residual = hs->res_last; diag->scale = hs->c;
        diag->mean_term = hs->mean_term; diag->tr1 = tr1; diag->tr2 = tr2; diag->tr_sqrt = tr_sqrt; diag->verified = 0; diag->reserved = 0;
    }
    if (hs->conv == 0)
        return set_error(FAD_ERR_NOT_CONVERGED, "Newton-Schulz stopped at max_iter with residual %.3e", hs->res_last);
    return FAD_OK;
}

// (first-use warm-up, common.cpp: warm_code_objects -- loading this translation unit's code object costs ~75 ms at the first launch)
const void* code_object_anchor_frechet() { return reinterpret_cast<const void*>(&ns32_finish); }
}  // namespace fad

using namespace fad;

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int fad_frechet(int d, const double* mu1, const double* cov1, const double* mu2, const double* cov2,
                double eps, int max_iter, double tol, int on_device, int device, void* stream,
                double* out_fad, fad_diag_t* diag) {
    if (d < 1 || d > 16384) return set_error(FAD_ERR_INVALID, "d=%d out of range", d);
    if (!mu1 || !mu2 || !cov1 || !cov2 || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    if (!g.ok) return set_error(FAD_ERR_HIP, "cannot select device %d", device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.job = Workspace::Job();
    FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
    const bool fast = fast_eligible(ws, d, max_iter, tol);
    if (!fast) enqueue_clear_states(static_cast<NsState*>(ws.small.p), 1, st);
    const int64_t dd = (int64_t)d * d;
    const double *dc1 = cov1, *dc2 = cov2, *dm1 = mu1, *dm2 = mu2;
    if (!on_device) {
        FAD_TRY(ws.stage.reserve((size_t)(4 * dd + 2 * d) * sizeof(double)));
        double* s = static_cast<double*>(ws.stage.p);
        FAD_HIP_TRY(hipMemcpyAsync(s, cov1, dd * sizeof(double), hipMemcpyHostToDevice, st));
        FAD_HIP_TRY(hipMemcpyAsync(s + dd, cov2, dd * sizeof(double), hipMemcpyHostToDevice, st));
        FAD_HIP_TRY(hipMemcpyAsync(s + 2 * dd, mu1, d * sizeof(double), hipMemcpyHostToDevice, st));
        FAD_HIP_TRY(hipMemcpyAsync(s + 2 * dd + d, mu2, d * sizeof(double), hipMemcpyHostToDevice, st));
        dc1 = s; dc2 = s + dd; dm1 = s + 2 * dd; dm2 = s + 2 * dd + d;
    }
    if (fast) {                                    // state reset, scales and digit planes from the caller's matrices
        ws.job.fast = true;
        FAD_TRY(fast_prepare(ws, d, 1, nullptr, nullptr, dc1, dc2, dm1, dm2, -1, nullptr, nullptr, st));
    }
    return frechet_single(d, dc1, dc2, dm1, dm2, eps, max_iter, tol, -1, device, st, ws, out_fad, diag, false);
}

// A score from two moments handles: checks, scratch, the job record; (mu, Sigma) of both handles go to the slot's staging
// area and the iteration state is cleared.  (Forming Sigma inside the first product instead -- from the packed statistics,
// between registers and LDS -- was measured: 18.7 us against 12.4 + 4.8 for product + this launch; dropped.)
static int stage_from_moments(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps, int mean_dtype,
                              hipStream_t st, Workspace& ws, int max_iter = 0, double tol = 0.0) {
    if (!h1 || !h2) return set_error(FAD_ERR_INVALID, "NULL argument");
    const int d = moments_dim(h1), device = moments_device(h1);
    if (moments_dim(h2) != d)
        return set_error(FAD_ERR_SHAPE, "Training and test covariances have different dimensions (%d vs %d)", d, moments_dim(h2));
    if (moments_device(h2) != device) return set_error(FAD_ERR_INVALID, "handles live on different devices");
    FAD_TRY(moments_settle(h1, st));
    FAD_TRY(moments_settle(h2, st));
    FAD_TRY(ws.small.reserve(ns_small_bytes(d, 1)));
    const int64_t dd = (int64_t)d * d;
    FAD_TRY(ws.stage.reserve((size_t)(4 * dd + 2 * d) * sizeof(double)));
    double* s = static_cast<double*>(ws.stage.p);
    ws.job = Workspace::Job();
    ws.job.d = d; ws.job.device = device; ws.job.stream = st; ws.job.eps = eps; ws.job.mean_dtype = mean_dtype; ws.job.ddof = ddof;
    ws.job.cov1 = s; ws.job.cov2 = s + dd; ws.job.mu1 = s + 2 * dd; ws.job.mu2 = s + 2 * dd + d;
    if (fast_eligible(ws, d, max_iter, tol)) {     // the eight-launch chain: its first kernel does this staging as well
        ws.job.fast = true;
        FAD_TRY(fast_prepare(ws, d, ddof, moments_packed(h1), moments_packed(h2), nullptr, nullptr, nullptr, nullptr, mean_dtype, s + 2 * dd, s, st,
                             moments_runsum(h1), moments_runsum(h2)));
        FAD_TRY(moments_mark_read(h1, st));
        return moments_mark_read(h2, st);
    }
    enqueue_finalize_for_frechet(moments_packed(h1), moments_packed(h2), d, ddof, s + 2 * dd, s, static_cast<NsState*>(ws.small.p), st,
                                 moments_runsum(h1), moments_runsum(h2));
    FAD_HIP_TRY(hipGetLastError());
    FAD_TRY(moments_mark_read(h1, st));
    return moments_mark_read(h2, st);
}

int fad_frechet_from_moments(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps,
                             int max_iter, double tol, int mean_dtype, void* stream, double* out_fad, fad_diag_t* diag) {
    if (!h1 || !h2 || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    const int device = moments_device(h1);
    DeviceGuard g(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.pool = &thread_pool(device);
    FAD_TRY(stage_from_moments(h1, h2, ddof, eps, mean_dtype, st, ws, max_iter, tol));
    const Workspace::Job j = ws.job;
    return frechet_single(j.d, j.cov1, j.cov2, j.mu1, j.mu2, eps, max_iter, tol, mean_dtype, device, st, ws, out_fad, diag, true);
}

int fad_frechet_from_moments_begin(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps, int mean_dtype,
                                   void* stream, fad_frechet_job_t** job) {
    if (!h1 || !h2 || !job) return set_error(FAD_ERR_INVALID, "NULL argument");
    *job = nullptr;
    const int device = moments_device(h1);
    DeviceGuard g(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.pool = &thread_pool(device);
    const bool mixed = mixed_eligible(ws, moments_dim(h1), 0, 0.0);
    FAD_TRY(stage_from_moments(h1, h2, ddof, eps, mean_dtype, st, ws));
    if (mixed) {
        const Workspace::Job& j = ws.job;
        NsProblem pb{j.d, 1, j.cov1, 0, j.cov2, 0, j.mu1, 0, j.mu2, 0, mean_dtype};
        FAD_TRY(mixed_begin(pb, device, st, ws));
        ws.job.mixed = true;
    }
    ws.busy = true;
    *job = reinterpret_cast<fad_frechet_job_t*>(wsp);
    return FAD_OK;
}

int fad_frechet_end(fad_frechet_job_t* job, double* out_fad, fad_diag_t* diag) {
    if (!job || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    Workspace& ws = *reinterpret_cast<Workspace*>(job);
    if (!ws.busy) return set_error(FAD_ERR_INVALID, "this job was collected already");
    DeviceGuard g(ws.job.device);
    ws.busy = false;                               // the slot is free again whatever happens below
    const Workspace::Job j = ws.job;
    // the low-precision chain is in flight (or nothing is: the synchronous path runs now); frechet_single collects /
    // tops up / falls back to the fp64 iteration exactly as the blocking entry point does
    return frechet_single(j.d, j.cov1, j.cov2, j.mu1, j.mu2, j.eps, 0, 0.0, j.mean_dtype, j.device, j.stream, ws, out_fad, diag, true);
}

int fad_frechet_from_moments_multi_begin(int count, const fad_moments_t* const* h1, const fad_moments_t* const* h2, int ddof, double eps,
                                         int mean_dtype, void* stream, fad_frechet_job_t** job) {
    if (!h1 || !h2 || !job) return set_error(FAD_ERR_INVALID, "NULL argument");
    *job = nullptr;
    if (count < 1 || count > kMaxMultiPairs) return set_error(FAD_ERR_INVALID, "count=%d out of range [1, %d]", count, kMaxMultiPairs);
    for (int b = 0; b < count; ++b) if (!h1[b] || !h2[b]) return set_error(FAD_ERR_INVALID, "pair %d: NULL handle", b);
    const int device = moments_device(h1[0]), d = moments_dim(h1[0]);
    for (int b = 0; b < count; ++b) {
        if (moments_dim(h1[b]) != d || moments_dim(h2[b]) != d)
            return set_error(FAD_ERR_SHAPE, "Training and test covariances have different dimensions (pair %d: %d vs %d, pair 0: %d)", b,
                             moments_dim(h1[b]), moments_dim(h2[b]), d);
        if (moments_device(h1[b]) != device || moments_device(h2[b]) != device) return set_error(FAD_ERR_INVALID, "handles live on different devices");
    }
    DeviceGuard g(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    Workspace* wsp = free_slot(device);
    if (!wsp) return set_error(FAD_ERR_INVALID, "all %d Frechet slots of this thread are in flight: collect one with fad_frechet_end", Pool::kSlots);
    Workspace& ws = *wsp;
    ws.pool = &thread_pool(device);
    ws.job = Workspace::Job();
    ws.job.d = d; ws.job.device = device; ws.job.stream = st; ws.job.eps = eps; ws.job.mean_dtype = mean_dtype; ws.job.ddof = ddof;
    ws.multi = Workspace::Multi();
    ws.multi.count = count;
    for (int b = 0; b < count; ++b) { ws.multi.h1[b] = h1[b]; ws.multi.h2[b] = h2[b]; }
    if (fast_eligible(ws, d, 0, 0.0)) {
        for (int b = 0; b < count; ++b) { FAD_TRY(moments_settle(h1[b], st)); FAD_TRY(moments_settle(h2[b], st)); }
        FAD_TRY(pairs_enqueue(ws, d, count, h1, h2, ddof, mean_dtype, st));
        for (int b = 0; b < count; ++b) { FAD_TRY(moments_mark_read(h1[b], st)); FAD_TRY(moments_mark_read(h2[b], st)); }
        ws.multi.enqueued = true;
    }
    ws.busy = true;
    *job = reinterpret_cast<fad_frechet_job_t*>(wsp);
    return FAD_OK;
}

int fad_frechet_multi_end(fad_frechet_job_t* job, int count, double* out_fad, fad_diag_t* diag) {
    if (!job || !out_fad) return set_error(FAD_ERR_INVALID, "NULL argument");
    Workspace& ws = *reinterpret_cast<Workspace*>(job);
    if (!ws.busy) return set_error(FAD_ERR_INVALID, "this job was collected already");
    if (count != ws.multi.count) return set_error(FAD_ERR_INVALID, "this job holds %d pairs, not %d", ws.multi.count, count);
    const Workspace::Job j = ws.job;
    const Workspace::Multi m = ws.multi;
    DeviceGuard g(j.device);
    const int d = j.d, nb = d / 32;
    bool done[kMaxMultiPairs] = {false};
    int chain_status[kMaxMultiPairs];                      // fast_decide_one's verdict on the pair (-1: no record)
    for (int b = 0; b < kMaxMultiPairs; ++b) chain_status[b] = -1;
    int rc = FAD_OK;
    if (m.enqueued) {
        const hipError_t e = hipEventSynchronize(ws.done_ev);
        if (e != hipSuccess) { ws.busy = false; return set_error(FAD_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e)); }
        const size_t hs = song_host_stride(d);
        const char* hpin = static_cast<const char*>(ws.fast_pairs_pin);
        const size_t off_stats = nsf::kHostVals * sizeof(double), off_words = off_stats + (size_t)(nsf::kTileStats + 2) * nb * nb * sizeof(double);
        const size_t off_vstats = host_vstats_off(d);
        const bool wide = wide_enabled(ws.pool);
        int learnt = -1;
        bool any_verified = false;
        for (int round = 0; round < 2; ++round) {
            bool ask = false;
            for (int b = 0; b < count; ++b) {
                if (done[b]) continue;
                const double* hv = reinterpret_cast<const double*>(hpin + (size_t)b * hs);
                const double* hx = reinterpret_cast<const double*>(hpin + (size_t)b * hs + off_stats);
                const int* hw = reinterpret_cast<const int*>(hpin + (size_t)b * hs + off_words);
                const double* hvx = reinterpret_cast<const double*>(hpin + (size_t)b * hs + off_vstats);
                if (hw[12] != m.gen) continue;              // (no record: the single route decides)
                MixedResult r;
                fast_decide_one(hw, hv, hx, nb, &r, wide ? hvx : nullptr, m.gen, wide ? kMaxLowWide : kMaxLow);
                chain_status[b] = r.status;
                if (r.too_few0 || r.too_few1) {
                    if (rc == FAD_OK) rc = set_error(FAD_ERR_TOO_FEW_ROWS, "FAD requires at least two embedding window frames in each set");
                    out_fad[b] = NAN; done[b] = true;
                    continue;
                }
                if (r.status == 5 && round == 0) { ask = true; continue; }       // its correction needs the verification products
                if (r.status != 1) continue;
                const double tr_sqrt = sqrt(r.c) * r.tr_scaled;
                out_fad[b] = r.mean_term + r.tr1 + r.tr2 - 2.0 * tr_sqrt;
                if (diag) {
                    fad_diag_t& q = diag[b];
                    memset(&q, 0, sizeof(q));
                    q.iters = r.iters + 1; q.converged = 3; q.used_eps = 0; q.route = 2;
                    q.residual = r.res; q.scale = r.c; q.mean_term = r.mean_term; q.tr1 = r.tr1; q.tr2 = r.tr2; q.tr_sqrt = tr_sqrt;
                    q.verified = r.pad == 1 ? 1 : 0;
                }
                if (r.decided_at + 1 > learnt) learnt = r.decided_at + 1;
                if (r.pad == 1) any_verified = true;
                done[b] = true;
            }
            if (!ask) break;
            // two more launches for the whole batch (pairs without a final iterate skip), one more wait
            const int rv = pairs_verify_enqueue(ws, d, count, j.stream);
            if (rv != FAD_OK) { if (rc == FAD_OK) rc = rv; break; }
            const hipError_t e2 = hipEventSynchronize(ws.done_ev);
            if (e2 != hipSuccess) { ws.busy = false; return set_error(FAD_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e2)); }
        }
        if (learnt > 0 && ws.pool) ws.pool->lp_iters = learnt;
        if (ws.pool && wide) ws.pool->lp_verify = any_verified;

        // ---- what the chain declined or did not finish: the float64 iteration over ALL of them as ONE batch (round 5; before, pair by
        // pair through fad_frechet_from_moments: 0.7 ms each for k^-2 spectra -- three 512^3 float64 products per step on 256 and 512
        // workgroups; sixteen problems per launch fill the chip: ~0.3 ms per pair).  K1 left every pair's (mu, Sigma) as the single route
        // would form them (numpy's means where the handles carry them) in its block; the product Sigma_1 Sigma_2 is formed anew in
        // float64 (fast_decide_one: why not the chain's).  Pairs the batch cannot close (no finite root: the eps fallback of fad.py:94-99;
        // too few rows) stay for the single entry point below, which reports them as it always did.  FAD_PAIRS_F64_BATCH=0: pair by pair.
        static const bool f64_batch = [] { const char* e = getenv("FAD_PAIRS_F64_BATCH"); return !(e && e[0] == '0'); }();
        // (only what the chain REJECTED or declined -- status 2: "the float64 route decides".  A pair the blind batch merely did not finish
        //  -- status 0 / 4 -- goes to the single entry point as before: it tops the chain up by a few launches and teaches the thread the
        //  launch count its kind needs; sent here instead, every batch of k^-1 pairs ran 21 float64 steps: r06b, 2 570 scores/s for 5 200)
        int declined = 0;
        for (int b = 0; b < count; ++b) if (!done[b] && chain_status[b] == 2) ++declined;
        if (f64_batch && rc == FAD_OK && declined >= 2) {
            const PairBlock L = pair_block(d);
            const size_t dd = (size_t)d * d;
            char* blk = static_cast<char*>(ws.fast_pairs.p);
            const int64_t stride = (int64_t)(L.stride / sizeof(double));
            const double* mus = reinterpret_cast<const double*>(blk + L.mus);
            const double* covs = reinterpret_cast<const double*>(blk + L.covs);
            int r2 = ws.small.reserve(ns_small_bytes(d, count));
            NsState* hst = nullptr;
            if (r2 == FAD_OK) {
                NsState* dstates = static_cast<NsState*>(ws.small.p);
                enqueue_clear_states(dstates, count, j.stream);
                uint32_t mask = 0;
                for (int b = 0; b < count; ++b) if (done[b] || chain_status[b] != 2) mask |= (1u << b);
                if (mask) enqueue_mark_states_done(dstates, mask, count, j.stream);
                // (K1 of a batch leaves the covariances out: formed here, element for element as it would have)
                nsf::PrepArgs pc;
                memset(&pc, 0, sizeof(pc));
                for (int b = 0; b < count; ++b) { pc.accs[2 * b] = moments_packed(m.h1[b]); pc.accs[2 * b + 1] = moments_packed(m.h2[b]); }
                pc.d = d; pc.ddof = j.ddof; pc.batch = 2; pc.pstride = (int64_t)L.stride;
                pc.covs = reinterpret_cast<double*>(blk + L.covs);
                hipLaunchKernelGGL(nsf::nsf_pairs_covs, dim3((unsigned)(dd / 2048), (unsigned)(2 * count)), dim3(512), 0, j.stream, pc);
                NsProblem pb{d, count, covs, stride, covs + dd, stride, mus, stride, mus + d, stride, j.mean_dtype};
                r2 = run_ns(pb, 0, 0.0, j.device, j.stream, ws, &hst, false, nullptr, (ws.pool && ws.pool->f64_iters_multi > 0) ? ws.pool->f64_iters_multi + 1 : 0);
            }
            if (r2 == FAD_OK && hst) {
                int most = 0;
                for (int b = 0; b < count; ++b) {
                    if (done[b] || chain_status[b] != 2) continue;
                    const NsState& q = hst[b];
                    if (q.nonfinite || q.conv == 0 || !q.finished || q.final_iter < 0) continue;          // -> the single entry point
                    const double tr_sqrt = sqrt(q.c) * q.tr_last;
                    out_fad[b] = q.mean_term + q.tr1 + q.tr2 - 2.0 * tr_sqrt;
                    if (diag) {
                        fad_diag_t& g2 = diag[b];
                        memset(&g2, 0, sizeof(g2));
                        g2.iters = q.final_iter + 1; g2.converged = q.conv; g2.used_eps = 0; g2.route = 0; g2.residual = q.res_last;
                        g2.scale = q.c; g2.mean_term = q.mean_term; g2.tr1 = q.tr1; g2.tr2 = q.tr2; g2.tr_sqrt = tr_sqrt;
                    }
                    if (q.final_iter + 1 > most) most = q.final_iter + 1;
                    done[b] = true;
                }
                if (most > 0 && ws.pool) ws.pool->f64_iters_multi = most;
            }
        }
    }
    ws.busy = false;                               // the slot is free again: the single route below takes any free one
    ws.multi = Workspace::Multi();
    for (int b = 0; b < count; ++b) {
        if (done[b]) continue;
        // not eligible, not finished within the blind batch, or rejected: exactly what the single entry point does for this pair
        const int r1 = fad_frechet_from_moments(m.h1[b], m.h2[b], j.ddof, j.eps, 0, 0.0, j.mean_dtype, j.stream, &out_fad[b], diag ? &diag[b] : nullptr);
        if (r1 != FAD_OK && rc == FAD_OK) rc = r1;
    }
    return rc;
}

int fad_frechet_cancel(fad_frechet_job_t* job) {
    if (!job) return set_error(FAD_ERR_INVALID, "NULL argument");
    Workspace& ws = *reinterpret_cast<Workspace*>(job);
    if (!ws.busy) return FAD_OK;
    DeviceGuard g(ws.job.device);
    // the enqueued kernels still write into the slot's buffers: it may only be handed out again once they are through
    if ((ws.job.mixed || ws.multi.enqueued) && ws.done_ev) FAD_HIP_TRY(hipEventSynchronize(ws.done_ev));
    else FAD_HIP_TRY(hipStreamSynchronize(ws.job.stream));
    ws.busy = false;
    ws.job = Workspace::Job();
    ws.multi = Workspace::Multi();
    return FAD_OK;
}

}  // extern "C"
