/*
 * fad_hip.h -- C ABI of libfad_hip.so, the MI355X (gfx950) implementation of the FAD hot path.
 *
 * The reference (microsoft/fadtk v1.1.0) is pure Python and has no FFI for this path; each
 * entry point below replaces the numpy/scipy call(s) cited next to it (paths relative to the
 * reference root).  A maintainer binds these with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns FAD_OK (0) or a negative fad_status; nothing throws across the ABI;
 *     fad_last_error() gives a thread-local message for the last failure on the calling thread.
 *   - `on_device` != 0 means the data pointers are device (HBM) pointers on the handle's GPU;
 *     0 means host pointers (the library stages them over PCIe itself).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Work is enqueued on
 *     that stream; functions that return results to HOST memory synchronise it before returning.
 *   - row-major everywhere; `ld` is the row pitch in ELEMENTS.
 *   - distinct handles are independent; handle-less functions are re-entrant; one handle must not
 *     be used from two threads at once.
 *   - there is NO CPU fallback: without a usable GPU every compute entry returns
 *     FAD_ERR_NO_DEVICE.
 */
#ifndef FAD_HIP_H
#define FAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAD_ABI_VERSION 2

typedef enum fad_status {
    FAD_OK = 0,
    FAD_ERR_INVALID = -1,       /* bad argument (NULL, negative size, unknown dtype ...)        */
    FAD_ERR_NO_DEVICE = -2,     /* no HIP device / wrong architecture                            */
    FAD_ERR_HIP = -3,           /* a HIP runtime call failed (message has the hipError string)   */
    FAD_ERR_ALLOC = -4,         /* device or host allocation failed                              */
    FAD_ERR_SHAPE = -5,         /* dimension mismatch (fad.py:78-81 AssertionError)              */
    FAD_ERR_TOO_FEW_ROWS = -6,  /* N < 2 frames (fad.py:46-47 AssertionError)                    */
    FAD_ERR_NOT_FINITE = -7,    /* NaN/Inf in the inputs or a diverged root (fad.py:102-106 ValueError) */
    FAD_ERR_NOT_CONVERGED = -8  /* iteration hit max_iter (result still written; see fad_diag_t) */
} fad_status;

typedef enum fad_dtype { FAD_F16 = 0, FAD_BF16 = 1, FAD_F32 = 2, FAD_F64 = 3 } fad_dtype;

/* ------------------------------------------------------------------ library / device */
int fad_version(void);                       /* FAD_ABI_VERSION                                   */
int fad_device_count(void);                  /* number of gfx950 devices visible (0 if none)      */
const char* fad_last_error(void);            /* thread-local, never NULL                          */
const char* fad_device_arch(int device);     /* e.g. "gfx950:sramecc+:xnack-"; "" on failure      */

/* ------------------------------------------------------------------ running moments
 * Replaces calc_embd_statistics (fadtk/fad.py:42-48), _process_file (fadtk/utils.py:13-16) and
 * the merge loop of calculate_embd_statistics_online (fadtk/utils.py:36-45).
 *
 * A handle accumulates the sufficient statistics (n, sum x, sum x x^T) of all rows fed so far,
 * in float64 in HBM, packed as [n | sum_x (D) | sum_xxT (D*D, row-major, symmetric)] =
 * 1 + D + D*D doubles.  Raw moments are sum-reducible, so merging datasets or GPUs is an
 * element-wise add of packed buffers (one RCCL all-reduce across ranks).
 */
typedef struct fad_moments fad_moments_t;

int fad_moments_create(int d, int device, fad_moments_t** out);
int fad_moments_destroy(fad_moments_t* h);
int fad_moments_reset(fad_moments_t* h, void* stream);
/* fad_moments_reset for `count` handles in one call (a scoring loop that re-feeds the accumulators of sixteen scores: one call into
 * the library instead of thirty-two). */
int fad_moments_reset_multi(int count, fad_moments_t* const* hs, void* stream);
/* The zeroing of a reset (or bind) is deferred: an update right after it overwrites the accumulator instead.  Whoever
 * reads the packed buffer BEHIND the library's back (a collective over a bound buffer) calls this first so that a
 * handle that received no rows holds zeros. */
int fad_moments_settle(fad_moments_t* h, void* stream);
/* Keep the packed statistics in the CALLER's device buffer (packed_len doubles, 16-byte aligned, same device) from
 * now on, and reset them.  The buffer then is what a collective runs over in place -- e.g. two handles bound to the
 * halves of one allocation are summed across ranks by a single all-reduce with no export/import copies (bench.py).
 * The caller keeps ownership and must outlive the handle or re-bind. */
int fad_moments_bind(fad_moments_t* h, double* device_packed);
int fad_moments_dim(const fad_moments_t* h);                 /* D, or <0 on error                */
int64_t fad_moments_packed_len(const fad_moments_t* h);      /* 1 + D + D*D                      */

/* Feed a block of `n` frames (rows) of `d` features.  n == 0 is a no-op. */
int fad_moments_update(fad_moments_t* h, const void* rows, int64_t n, int64_t ld, int dtype,
                       int on_device, void* stream);

/* Reference-order means (off by default).  The reference's np.mean(embd_lst, axis=0) (fadtk/fad.py:48) adds the rows one after the
 * other in float32 (float16 frames widened first) and divides by n in float32: for frames with a sizeable offset the float16-rounded
 * result differs from the rounded EXACT mean -- what fad_moments_finalize returns otherwise -- by one ulp in a few dimensions, worth
 * 2e-5 .. 5e-4 of a small Frechet distance.  With the switch on, every update also carries numpy's float32 running column sums (one lane
 * per column walks the rows in order, on a stream of its own beside the update's other kernels: csrc/moments_kernels.h) and
 * fad_moments_finalize returns mu = float32(float64(run) / n) (numpy's quotient) widened to double; the covariance is unchanged.  Covers plain updates
 * (fad_moments_update / _multi, host or device rows, float16 / bfloat16 / float32) in the order they are fed; statistics that were imported,
 * all-reduced or fed while the switch was off have no row order: finalize then falls back to the exact mean.  The
 * fad_frechet_from_moments* entry points take the same mean for their mean term (then rounded as `mean_dtype` asks) from a handle whose
 * running sums cover its rows, the exact one otherwise.
 * enabled = 2 ("detached"): as 1, and the caller vouches that every frame matrix it feeds is COMPLETE when the update is called and stays
 * unchanged until the handle's statistics are next read (finalize / export / merge / allreduce / fad_frechet_from_moments*): the walk
 * then neither waits for the work queued on the caller's stream nor holds that stream up -- it runs beside everything and the readers
 * wait for it.  With 1 the walk starts behind the caller's stream and the update does not return the stream before it is through. */
int fad_moments_set_reference_mean(fad_moments_t* h, int enabled);

/* Feed `count` (1..32) frame matrices to `count` DIFFERENT handles of one dimension, dtype and device with ONE
 * launch of each kernel (more than 16: one launch on the 256-column-slab route -- float16, D >= 512, no running sums asked for --
 * else sixteen at a time): rows[i] (a DEVICE pointer, n[i] frames, pitch ld[i]) goes to hs[i].  The two datasets of a
 * FAD score (fad.py:292-302 calls calc_embd_statistics / load_stats twice), the 25 resamples of score_inf
 * (fad.py:333-341) ...: the workgroup slots of the GPU are shared out over all sets in proportion to their rows,
 * so the fixed costs of a pass (one 64 KiB partial tile per workgroup, pipeline fill, launches) are paid once.
 * Results are identical to `count` separate fad_moments_update calls up to the fp64 summation order of the
 * partial tiles.  Sets with n[i] == 0 are skipped. */
int fad_moments_update_multi(int count, fad_moments_t* const* hs, const void* const* rows, const int64_t* n,
                             const int64_t* ld, int dtype, void* stream);

/* fad_moments_update_multi over GATHERED rows: set i is fed the n_idx[i] frames rows[idx[i][0]], rows[idx[i][1]], ... of ONE resident
 * matrix `rows` [n_src x ld] (device; idx[i]: device, int32, 0 <= idx < n_src), in that order -- the resamples with replacement of
 * score_inf (fad.py:333-337: np.random.choice + fancy indexing, 1.3 GB of copies at config 3) without materialising them: for float16
 * frames of D >= 512 the gather rides on the row offsets of the slab kernel's LDS-DMA loads and of the running-sum walk; other inputs
 * are gathered into the handles' staging areas first.  Results as fad_moments_update_multi on the materialised matrices (count <= 16). */
int fad_moments_update_multi_indexed(int count, fad_moments_t* const* hs, const void* rows, int64_t n_src, int64_t ld, int dtype,
                                     const int32_t* const* idx, const int64_t* n_idx, void* stream);

/* Same, for `n_segments` files/songs stored back to back: segment s owns rows
 * [offsets[s], offsets[s+1]) (offsets is a HOST array of n_segments+1 entries).  If
 * seg_sums != NULL it receives the per-segment column sums, [n_segments x D] float64 (host or
 * device per on_device): the per-file means of utils.py:16 and the per-song means of
 * fad.py:377 are seg_sums / segment length. */
int fad_moments_update_segmented(fad_moments_t* h, const void* rows, int64_t n, int64_t ld, int dtype,
                                 const int64_t* offsets, int64_t n_segments, double* seg_sums,
                                 int on_device, void* stream);

/* The per-file mean terms of the online statistics.  fadtk merges per-file (mean, scatter, n) triplets
 * (utils.py:13-16, 36-40) and np.mean of a float16 file is float16, so the dataset covariance it returns is
 *   (sum xx^T - sum_f n_f m_f m_f^T  +  sum_f n_f m~_f m~_f^T - N mu~ mu~^T) / (N - 1),  mu~ = sum_f n_f m~_f / N
 * with m_f the exact and m~_f the dtype-rounded mean of file f.  Given the per-file column sums (seg_sums
 * [n_files x D] float64, from fad_moments_update_segmented) and sizes (int64 [n_files]) -- on_device bit 0: seg_sums is
 * a device pointer, bit 1: sizes is (the usual call has the sums in HBM and the sizes on the host: on_device = 1) --
 * this accumulates the rows sqrt(n_f) m_f into `exact`, sqrt(n_f) m~_f into `rounded` and n_f m~_f into
 * `weighted` (three distinct handles of dimension D): afterwards sum_xxT(exact) and sum_xxT(rounded) are the two
 * D x D terms and sum_x(weighted) = N mu~ -- all sum-reducible across batches and ranks.  Empty files add nothing. */
int fad_moments_update_file_means(fad_moments_t* exact, fad_moments_t* rounded, fad_moments_t* weighted,
                                  const double* seg_sums, const int64_t* sizes, int64_t n_files, int dtype,
                                  int on_device, void* stream);

/* The two calls above with the reference's OWN per-file means.  np.mean of a float16 (float32) file (utils.py:16; per song fad.py:377)
 * adds the file's rows one after the other in float32 and rounds the quotient to the file's dtype: for a file of a few thousand frames
 * with an offset that differs from the rounded exact mean -- what fad_moments_update_file_means forms from seg_sums -- by one ulp in
 * ~0.3 % of the columns.  fad_moments_update_segmented_ref additionally returns those float32 running column sums per segment
 * (seg_runsums [n_segments x D] float32, host or device like seg_sums; NULL = none; float16 / bfloat16 / float32 rows: for float64
 * rows numpy's sum is the exact one) -- a second walk over rows the tile kernel has just read.  fad_moments_update_file_means_ref takes
 * them (on_device bit 2: seg_runsums is a device pointer; NULL = the rounded exact means) and forms m~_f = round(float32(run_f / n_f)). */
int fad_moments_update_segmented_ref(fad_moments_t* h, const void* rows, int64_t n, int64_t ld, int dtype,
                                     const int64_t* offsets, int64_t n_segments, double* seg_sums, float* seg_runsums,
                                     int on_device, void* stream);
int fad_moments_update_file_means_ref(fad_moments_t* exact, fad_moments_t* rounded, fad_moments_t* weighted,
                                      const double* seg_sums, const float* seg_runsums, const int64_t* sizes, int64_t n_files, int dtype,
                                      int on_device, void* stream);

/* dst += src.  (Rows from another handle have no place in dst's row order: a handle with fad_moments_set_reference_mean on falls back
 * to the exact mean afterwards, as after import / allreduce.) */
int fad_moments_merge(fad_moments_t* dst, const fad_moments_t* src, void* stream);
/* Copy the packed float64 statistics out / in (the buffer an RCCL all-reduce runs over). */
int fad_moments_export(const fad_moments_t* h, double* packed, int on_device, void* stream);
int fad_moments_import(fad_moments_t* h, const double* packed, int on_device, void* stream);
int fad_moments_count(const fad_moments_t* h, int64_t* n, void* stream);

/* Sum the packed statistics of all ranks IN PLACE with one collective (the only
 * exchange of the data-parallel path; replaces the pickled per-file (mean, scatter, n) tuples of
 * utils.py:19-46 / the process pool of fad_batch.py:43-48).  `rccl_comm` is the caller's ncclComm_t;
 * ncclAllReduce(count = 1 + D + D*D, ncclFloat64, ncclSum) is looked up in the RCCL library the host
 * process already has loaded (else librccl.so.1), so the library adds no second RCCL to the process.
 * FAD_ERR_INVALID when no RCCL can be found, FAD_ERR_HIP when the collective reports an error.
 * (torch.distributed does not hand out its communicator: the Python host reduces the same buffer through
 * torch.distributed.all_reduce instead, fadtk_amd/dist.py.) */
int fad_moments_allreduce(fad_moments_t* h, void* rccl_comm, void* stream);

/* mu = sum_x / n ; cov = (sum_xxT - n mu mu^T) / (n - ddof)   (np.mean / np.cov, fad.py:48).
 * mu [D], cov [D*D] float64.  n < 2 -> FAD_ERR_TOO_FEW_ROWS (fad.py:46-47). */
int fad_moments_finalize(const fad_moments_t* h, int ddof, double* mu, double* cov, int64_t* n,
                         int on_device, void* stream);

/* ------------------------------------------------------------------ Frechet distance
 * Replaces calc_frechet_distance (fadtk/fad.py:51-120):
 *   ||mu1-mu2||^2 + tr C1 + tr C2 - 2 tr sqrt(C1 C2)
 * tr sqrt(C1 C2) = sum_i sqrt(lambda_i(C1 C2)) -- the value the reference returns through
 * scipy.linalg.eig (fad.py:91-92) -- is computed with a coupled Newton-Schulz iteration on MFMA tiles: for
 * well-conditioned products (and d a multiple of 64) the iterations run in low precision (float32 on the f32 MFMA; for
 * d = 256 / 512 / 768 / 1024 split float16 on the f16 MFMA with the two products that need it formed EXACTLY through
 * base-128 digit planes on the int8 MFMA) and one float64-accurate correction restores float64 accuracy; otherwise (or
 * when max_iter / tol are given) everything runs in float64.
 * eps: added to both diagonals for a retry when the first attempt diverges (fad.py:94-99).
 * max_iter <= 0 -> default (64); tol <= 0 -> default.
 */
typedef struct fad_diag {
    int32_t iters;          /* Newton-Schulz iterations executed                                 */
    int32_t converged;      /* 1: residual < tol, 2: trace stagnated / divergence guard (rank-deficient or near-singular
                               product), 3: float32 iterations + float64 correction accepted, 0: max_iter */
    int32_t used_eps;       /* 1 if the eps-regularised retry produced the result                */
    int32_t route;          /* with converged == 3: 1 = float32 iterations on the f32 MFMA, 2 = split-float16 iterations + exact
                               int8-MFMA products (d = 256 / 384 / 512 / 768 / 1024); 0 otherwise (all-float64 iteration)  */
    double residual;        /* ||I - Z Y||_F at the last iteration                               */
    double scale;           /* c with Y0 = C1 C2 / c                                             */
    double mean_term;       /* ||mu1 - mu2||^2 in float64                                        */
    double tr1, tr2;        /* tr C1, tr C2                                                      */
    double tr_sqrt;         /* tr sqrt(C1 C2)                                                    */
    int32_t verified;       /* route 2 only: 1 = accepted on the MEASURED verification record (1/2 tr(EP) added, 4 x 1/8 |tr(ZPP)| +
                               ||E||^2 ||P|| as the error estimate: csrc/frechet.hip) where the norm bound says nothing (decaying
                               spectra); 0 = accepted on the norm bound, or another route                                  */
    int32_t reserved;
} fad_diag_t;

int fad_frechet(int d, const double* mu1, const double* cov1, const double* mu2, const double* cov2,
                double eps, int max_iter, double tol, int on_device, int device, void* stream,
                double* out_fad, fad_diag_t* diag);

/* Same, straight from two moment handles (no host round trip of mu/cov).
 * mean_dtype: how ||mu1 - mu2||^2 is formed.  FAD_F16 / FAD_BF16 / FAD_F32 = as the reference does for embeddings
 * of that dtype: np.mean keeps the dtype (fad.py:48 on the float16 arrays of model_loader.py:47-48), so the means are
 * rounded to it, subtracted in it, and diff.dot(diff) (fad.py:83, 119) is accumulated in float32 and rounded to it
 * (float16: bit for bit what numpy returns).  Anything else (FAD_F64, -1): float64 means.
 * FAD_MEAN_SECOND_ONLY | dtype: only the mean of the SECOND handle is rounded to the dtype, difference and dot product stay in
 * float64 -- score_inf's case (fad.py:333-341): a float64 baseline mean from the statistics cache against np.mean of resampled
 * float16 frames. */
#define FAD_MEAN_SECOND_ONLY 16
int fad_frechet_from_moments(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps,
                             int max_iter, double tol, int mean_dtype, void* stream, double* out_fad, fad_diag_t* diag);

/* Several scores in flight.  The square-root chain of ONE score is a dozen dependent launches of small kernels: it is
 * bound by launch latency.  begin() puts (mu, Sigma) of both handles and the whole chain on `stream` and returns at once;
 * end() waits for it and delivers the same result (and the same errors) as fad_frechet_from_moments -- topping the
 * iteration up or falling back to the float64 iteration when the device says so.  A caller that begins score k+1 before it
 * ends score k never leaves the device waiting between scores (bench.py: +10 % scores/s on ONE stream); with one stream
 * per score the chains and moments passes of consecutive scores overlap as well (another +20 %: bench.py's timed layout).
 * The handles may be reset / fed again ON `stream` as soon as begin() has returned (their statistics were copied out in
 * stream order); on another stream only after end().  A job must be ended (or cancelled) by the host thread that began it:
 * the slots are thread-local, 8 per thread and device.  cancel() gives the slot back without a result. */
typedef struct fad_frechet_job fad_frechet_job_t;
int fad_frechet_from_moments_begin(const fad_moments_t* h1, const fad_moments_t* h2, int ddof, double eps, int mean_dtype,
                                   void* stream, fad_frechet_job_t** job);
int fad_frechet_end(fad_frechet_job_t* job, double* out_fad, fad_diag_t* diag);
int fad_frechet_cancel(fad_frechet_job_t* job);

/* `count` (1..FAD_MULTI_MAX_PAIRS) independent scores in ONE job: pair b = (h1[b], h2[b]).  The eight launches of the square-root chain carry all
 * pairs (each launch costs ~4 us before it does anything and its workgroups are latency-bound: B chains as one batch take little
 * longer than one); multi_end() delivers out_fad[b] (and diag[b], or diag == NULL) exactly as fad_frechet_from_moments would for
 * that pair -- a pair the batch does not finish or accept goes through that very entry point.  Same stream / thread rules as
 * begin() / end(); one slot per job.  Used by bench.py for the scores it keeps in flight and by score_inf (fad.py:304-351: 25
 * independent scores against one baseline). */
#define FAD_MULTI_MAX_PAIRS 32   /* (8 until round 5; 16 pairs put a workgroup of the batched kernels on every CU, 32 two) */
int fad_frechet_from_moments_multi_begin(int count, const fad_moments_t* const* h1, const fad_moments_t* const* h2, int ddof,
                                         double eps, int mean_dtype, void* stream, fad_frechet_job_t** job);
int fad_frechet_multi_end(fad_frechet_job_t* job, int count, double* out_fad, fad_diag_t* diag);

/* ------------------------------------------------------------------ per-song FAD (--indiv)
 * Replaces the loop of score_individual (fadtk/fad.py:373-387): for every song s (rows
 * [offsets[s], offsets[s+1]) of `rows`), FAD between the baseline (mu_b, cov_b) and that song's
 * own (mu_s, cov_s).  Songs with fewer than 2 frames get status FAD_ERR_TOO_FEW_ROWS and a NaN
 * score (the reference drops them, fad.py:380-391).
 * mean_mode: 0 = song mean in float64; 1 = round the song mean to the input dtype first, as
 * np.mean does for float16 (model_loader.py:47-48 + fad.py:48).
 * on_device = 1: rows, mu_b and cov_b are DEVICE pointers (a caller scoring many batches against one baseline uploads it
 * once); offsets, out_scores and out_status are host pointers either way.  cov_b is read as (cov_b + cov_b^T)/2.
 * Routes, chosen per song by its frame count n (the result does not depend on the route beyond ~1e-8 of a score):
 *   n = 2: closed form; n - 1 < D: the Gram form on the n - 1 non-zero eigenvalues; n >= D + 1 and D in {128, 256, 384, 512, 768,
 *   1024}: the low-precision chain of the single pair, batched over the songs (exact int8-MFMA products, split-float16
 *   Newton-Schulz, one float64-accurate correction, accepted per song on a bound of what the correction neglects), with
 *   float16 frames also the covariances on the float16 matrix pipe; whatever that chain does not accept -- and every other D --
 *   the float64 Newton-Schulz routes.  Environment switches (tests, diagnosis): FAD_SONG_FAST=0 (float64 routes only),
 *   FAD_SONG_BIG=<smallest batch on 128 x 128 tiles>, FAD_SONG_RES=0, FAD_SONG_COV16=0, FAD_SONG_STATS16=0, FAD_FAST_TRACE=1.
 */
int fad_frechet_batched_vs_baseline(int d, const double* mu_b, const double* cov_b,
                                    const void* rows, int64_t n_rows, int64_t ld, int dtype,
                                    const int64_t* offsets, int64_t n_songs, int mean_mode,
                                    int on_device, int device, void* stream,
                                    double* out_scores, int32_t* out_status);

/* ------------------------------------------------------------------ log-mel front ends
 * Replace the third-party feature extraction that runs inside ModelLoader._get_embedding
 * (fadtk/model_loader.py:99,108 VGGish / torchvggish mel_features; :661,666 Whisper /
 * transformers WhisperFeatureExtractor; :385,406 CLAP-HTSAT / torchlibrosa).
 * wav: float32 mono samples of n_clips clips stored back to back, clip c = wav[offsets[c] ..
 * offsets[c+1]) (offsets: HOST array of n_clips+1).  wav/out are host or device per on_device.
 *
 * VGGish  (16 kHz): periodic-Hann 400 / hop 160 / FFT 512 magnitude, 64 HTK mels 125-7500 Hz,
 *          log(mel + 0.01), non-overlapping examples of 96 frames, incomplete tail dropped.
 *          out [total_examples][96][64]; example_offsets (host, n_clips+1, may be NULL) receives the
 *          first example of each clip.
 * Whisper (16 kHz): clip zero-padded / cut to 480000 samples, centred reflect STFT 400 / hop 160,
 *          power, n_mels (80 or 128) Slaney mels 0-8 kHz, log10(max(.,1e-10)), clamp to
 *          (clip max - 8), (x + 4) / 4.   out [n_clips][n_mels][3000].
 * HTSAT   (48 kHz): centred reflect STFT 1024 / hop 480, power, 64 Slaney mels 50-14000 Hz,
 *          10 log10(max(.,1e-10)).  Every clip must give n_frames_out = 1 + n_samples/480 frames.
 *          out [n_clips][n_frames_out][64].
 */
int64_t fad_logmel_vggish_num_examples(int64_t n_samples);
int fad_logmel_vggish(const float* wav, const int64_t* offsets, int64_t n_clips, float* out,
                      int64_t out_capacity_examples, int64_t* example_offsets, int on_device, int device,
                      void* stream);
int fad_logmel_whisper(const float* wav, const int64_t* offsets, int64_t n_clips, int n_mels, float* out,
                       int on_device, int device, void* stream);
int fad_logmel_htsat(const float* wav, const int64_t* offsets, int64_t n_clips, int64_t n_frames_out,
                     float* out, int on_device, int device, void* stream);

/* ------------------------------------------------------------------ audio normalisation
 * Replaces the resampler of FrechetAudioDistance.load_audio (fadtk/fad.py:151-159):
 *   torchaudio.transforms.Resample(fs, model_sr, lowpass_filter_width=64, rolloff=0.9475937167399596,
 *                                  resampling_method="sinc_interp_kaiser", beta=14.769656459379492)
 * on a mono float32 signal (the mono mix of fad.py:150 is the caller's), output length
 * ceil(new_sr * n / orig_sr) = fad_resample_num_samples().  quantize_pcm16 != 0 also applies the 16-bit round
 * trip the reference makes through its cache file (fad.py:160 saves PCM_S 16, model_loader.py:64 reads int16 /
 * 32768): out = clamp(rint(y * 32768), -32768, 32767) / 32768.
 * Rates whose ratio needs a filter table above 256 MiB (nearly coprime rates) are refused with FAD_ERR_INVALID. */
int64_t fad_resample_num_samples(int64_t n, int orig_sr, int new_sr);
int fad_resample_kaiser(const float* wav, int64_t n, int orig_sr, int new_sr, int quantize_pcm16, float* out,
                        int64_t out_capacity, int on_device, int device, void* stream);

/* ------------------------------------------------------------------ diagnostics (NOT part of the drop-in surface)
 * Nothing in fadtk corresponds to these two calls and no binding of the reference needs them: they exist for bench.py's
 * roofline object (HIP events around the tile kernel on the stream it is launched on) and for the GPU tests that check
 * which kernel variant ran.
 *
 * Opt-in HIP-event timing (bench.py's roofline): while enabled every update records events around
 * its tile kernel on the caller's stream (no synchronisation); last_timing() returns the AVERAGE
 * duration in ms of the tile kernel and of the reduce kernels over the updates recorded since the
 * last query (at most 256), and which tile kernel ran (0 = fp16/bf16 MFMA on 128 x 128 tiles, 1 = generic fp64,
 * 2 = fp16 MFMA on 256-column slabs, D >= 512).
 * A fad_moments_update_multi call is ONE update recorded on hs[0]: its tile-kernel time covers all sets.
 * enabled = 2 records the two events around the tile kernel only (ms_reduce comes back 0): an event record between two
 * kernels costs the stream a few microseconds, and the one behind the reduce sits in front of whatever the caller
 * enqueues next. */
/* Give back the handle's host-input staging area (and scratch) when it holds more than `keep_bytes`: a handle cached between calls
 * (fadtk_amd/fad.py keeps one per thread) otherwise pins up to 1 GiB of HBM for the life of its thread.  Synchronises the device. */
int fad_moments_trim(fad_moments_t* h, int64_t keep_bytes);
int fad_moments_set_timing(fad_moments_t* h, int enabled);
int fad_moments_last_timing(fad_moments_t* h, float* ms_main_kernel, float* ms_reduce_kernel,
                            int* kernel_variant);

/* A HIP stream confined to a subset of the device's CUs (hipExtStreamCreateWithCUMask; `mask` = `words` x 32 bits, bit i = CU i in
 * the runtime's numbering, which on this 8-XCD part walks the XCDs first: bit i -> XCD i % 8).  bench.py's --chain-cus layout
 * experiment puts the square-root chains and the moments kernels on disjoint CU sets with it; nothing in the library uses it.
 * The stream is the caller's (destroy with fad_stream_destroy); `*stream` is a hipStream_t. */
int fad_stream_create_cu_mask(int device, const uint32_t* mask, int words, void** stream);
int fad_stream_destroy(int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FAD_HIP_H */
