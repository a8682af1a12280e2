// Batched STFT + mel + log front ends on gfx950 (VGGish, Whisper, CLAP-HTSAT).
//
// Replaces the third-party feature extraction that runs (mostly on the CPU, one file at a time)
// inside ModelLoader._get_embedding -- fadtk/model_loader.py:99,108 (torchvggish mel_features),
// :661,666 (transformers WhisperFeatureExtractor), :385,406 (torchlibrosa inside HTSAT).
//
// One kernel template for all three.  A workgroup owns 32 consecutive frames of one clip:
//   1. the sample span of those frames goes to LDS once (zero / reflect padding applied there),
//      skewed by one float per hop so that 16 frames read conflict-free;
//   2. STFT as a real DFT on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate):
//        [32 frames x K samples] x [K x 2*bins]  with the window folded into the cos/sin tables
//      (tables built on the host in fp64, L2 resident), power / magnitude formed in registers;
//   3. mel projection [32 x bins] x [bins x n_mels] on the same MFMA, from LDS;
//   4. log compression and the model's output layout.
// Whisper's "clamp to (clip max - 8), (x + 4) / 4" needs the clip maximum: pass 1 keeps an
// order-preserving integer atomicMax per clip, a second tiny kernel applies it.
#include "fad_common.h"

#include <cmath>
#include <mutex>
#include <vector>

namespace fad {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { FE_VGGISH = 0, FE_WHISPER = 1, FE_HTSAT = 2 };

struct FrontEnd {            // compile-time description of a front end
    int kind, nfft, win, hop, bins, nmel, center;
};

template <int KIND> struct Cfg;
template <> struct Cfg<FE_VGGISH> { static constexpr int NFFT = 512, WIN = 400, HOP = 160, BINS = 257, CENTER = 0; };
template <> struct Cfg<FE_WHISPER> { static constexpr int NFFT = 400, WIN = 400, HOP = 160, BINS = 201, CENTER = 1; };
template <> struct Cfg<FE_HTSAT> { static constexpr int NFFT = 1024, WIN = 1024, HOP = 480, BINS = 513, CENTER = 1; };

constexpr int FT = 32;                       // frames per workgroup
constexpr int WHISPER_SAMPLES = 480000;      // 30 s @ 16 kHz (pad / trim target)
constexpr int WHISPER_FRAMES = 3000;

__device__ __forceinline__ unsigned f32_order_key(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_key(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct LogmelArgs {
    const float* wav; const int64_t* offsets;      // clip c = wav[offsets[c] .. offsets[c+1])
    const int64_t* frame_base;                     // first output frame index of clip c (VGGish: example*96)
    const int64_t* clip_frames;                    // number of frames to produce for clip c
    const float* wcos; const float* wsin;          // [WIN][bins_pad], window folded in
    const float* melw;                             // [bins_pad][nmel_pad]
    float* out; unsigned* clip_max;                // Whisper: per-clip max key
    int nmel, nmel_pad, bins_pad;
    int64_t out_frames_per_clip;                   // HTSAT / Whisper: frames per clip in `out`
};

template <int KIND>
__global__ __launch_bounds__(256) void logmel_kernel(LogmelArgs a) {
    using C = Cfg<KIND>;
    constexpr int K = C::WIN, HOP = C::HOP;
    constexpr int SPAN = (FT - 1) * HOP + K;
    constexpr int SPAN_SK = SPAN + 2 * (SPAN / HOP) + 2;      // skewed: two pad floats per hop
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                          // [SPAN_SK]
    float* P = lds + ((SPAN_SK + 3) & ~3);                    // [FT][bins_pad + 1]
    const int ppitch = a.bins_pad + 1;

    const int clip = blockIdx.y;
    const int64_t nfr = a.clip_frames[clip];
    const int64_t f0 = (int64_t)blockIdx.x * FT;
    if (f0 >= nfr) return;
    const float* w = a.wav + a.offsets[clip];
    const int64_t nsamp = a.offsets[clip + 1] - a.offsets[clip];
    const int64_t L = (KIND == FE_WHISPER) ? WHISPER_SAMPLES : nsamp;       // logical signal length
    const int64_t p0 = f0 * HOP - (C::CENTER ? C::NFFT / 2 : 0);           // first sample of frame f0

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;

    for (int s = tid; s < SPAN; s += 256) {
        int64_t p = p0 + s;
        if (C::CENTER) {                                       // reflect padding (numpy / torch 'reflect')
            if (p < 0) p = -p;
            if (p >= L) p = 2 * (L - 1) - p;
        }
        float v = 0.f;
        if (p >= 0 && p < L && p < nsamp) v = w[p];            // Whisper: zeros beyond the clip up to 30 s
        xs[s + 2 * (s / HOP)] = v;                            // frame i starts at i*(HOP+2): lanes li hit even banks,
                                                               // the k+1 half of the wave the odd ones
    }
    __syncthreads();

    // ---- STFT: each wave takes bin tiles wave, wave+4, ...
    const int nbt = a.bins_pad / 16;
    const int r0 = li * (HOP + 2);                             // skewed start of frame li
    const int r1 = (16 + li) * (HOP + 2);
    for (int bt = wave; bt < nbt; bt += 4) {
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, s0 = c0, c1 = c0, s1 = c0;
        const float* wc = a.wcos + bt * 16 + li;
        const float* ws = a.wsin + bt * 16 + li;
#pragma unroll 4
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + lk;
            const int ksk = k + 2 * (k / HOP);                 // sample i*HOP + k lives at i*(HOP+2) + k + 2*(k/HOP)
            const float a0 = xs[r0 + ksk];
            const float a1 = xs[r1 + ksk];
            const float bc = wc[(int64_t)k * a.bins_pad], bs = ws[(int64_t)k * a.bins_pad];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bc, c0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bs, s0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bc, c1, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bs, s1, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {                    // D: col = li (bin), row = lk*4 + reg (frame)
            const int fr = lk * 4 + reg, bin = bt * 16 + li;
            float p0v = c0[reg] * c0[reg] + s0[reg] * s0[reg];
            float p1v = c1[reg] * c1[reg] + s1[reg] * s1[reg];
            if (KIND == FE_VGGISH) { p0v = sqrtf(p0v); p1v = sqrtf(p1v); }
            P[fr * ppitch + bin] = p0v;
            P[(16 + fr) * ppitch + bin] = p1v;
        }
    }
    __syncthreads();

    // ---- mel projection + log + store: tiles (frame tile ft, mel tile mt) over the 4 waves
    const int nmt = a.nmel_pad / 16;
    float wmax = -3.0e38f;
    for (int t = wave; t < 2 * nmt; t += 4) {
        const int ft = t / nmt, mt = t - ft * nmt;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* prow = P + (ft * 16 + li) * ppitch;
        const float* mw = a.melw + mt * 16 + li;
#pragma unroll 4
        for (int k0 = 0; k0 < a.bins_pad; k0 += 4) {
            const int k = k0 + lk;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(prow[k], mw[(int64_t)k * a.nmel_pad], acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t f = f0 + ft * 16 + lk * 4 + reg;
            const int m = mt * 16 + li;
            if (f >= nfr || m >= a.nmel) continue;
            const float e = acc[reg];
            if (KIND == FE_VGGISH) {
                a.out[(a.frame_base[clip] + f) * a.nmel + m] = logf(e + 0.01f);
            } else if (KIND == FE_WHISPER) {
                const float v = log10f(fmaxf(e, 1e-10f));
                a.out[((int64_t)clip * a.nmel + m) * a.out_frames_per_clip + f] = v;
                wmax = fmaxf(wmax, v);
            } else {
                a.out[((int64_t)clip * a.out_frames_per_clip + f) * a.nmel + m] = 10.f * log10f(fmaxf(e, 1e-10f));
            }
        }
    }
    if (KIND == FE_WHISPER) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
        if (lane == 0 && wmax > -1.0e38f) atomicMax(a.clip_max + clip, f32_order_key(wmax));
    }
}

// Whisper pass 2: x <- (max(x, clipmax - 8) + 4) / 4
__global__ __launch_bounds__(256) void whisper_normalize(float* __restrict__ out, const unsigned* __restrict__ clip_max,
                                                         int64_t per_clip) {
    const int clip = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_clip) return;
    const float floor_v = f32_from_key(clip_max[clip]) - 8.0f;
    float* p = out + (int64_t)clip * per_clip + i;
    *p = (fmaxf(*p, floor_v) + 4.0f) * 0.25f;
}

// ------------------------------------------------------------------------------------------
// host: tables (fp64 -> fp32), cached per (front end, n_mels, device)
// ------------------------------------------------------------------------------------------
static double hz_to_mel_htk(double f) { return 1127.0 * std::log(1.0 + f / 700.0); }
static double hz_to_mel_slaney(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = 15.0, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz_slaney(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = 15.0, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// VGGish (AudioSet mel_features): triangles in the HTK-mel domain, DC bin zeroed, no normalisation
static void mel_htk_vggish(int bins, int nmel, double sr, double fmin, double fmax, std::vector<double>& w) {
    w.assign((size_t)bins * nmel, 0.0);
    const double nyq = sr / 2.0, lo = hz_to_mel_htk(fmin), hi = hz_to_mel_htk(fmax);
    for (int b = 1; b < bins; ++b) {
        const double m = hz_to_mel_htk(nyq * b / (bins - 1));
        for (int i = 0; i < nmel; ++i) {
            const double lower = lo + (hi - lo) * i / (nmel + 1), center = lo + (hi - lo) * (i + 1) / (nmel + 1),
                         upper = lo + (hi - lo) * (i + 2) / (nmel + 1);
            const double v = std::min((m - lower) / (center - lower), (upper - m) / (upper - center));
            w[(size_t)b * nmel + i] = v > 0.0 ? v : 0.0;
        }
    }
}

// librosa / transformers mel_filter_bank(norm="slaney", mel_scale="slaney"): triangles in Hz, area-normalised
static void mel_slaney(int bins, int nmel, double sr, double fmin, double fmax, std::vector<double>& w) {
    w.assign((size_t)bins * nmel, 0.0);
    std::vector<double> ff(nmel + 2);
    const double lo = hz_to_mel_slaney(fmin), hi = hz_to_mel_slaney(fmax);
    for (int i = 0; i < nmel + 2; ++i) ff[i] = mel_to_hz_slaney(lo + (hi - lo) * i / (nmel + 1));
    for (int b = 0; b < bins; ++b) {
        const double f = (sr / 2.0) * b / (bins - 1);
        for (int i = 0; i < nmel; ++i) {
            const double down = (f - ff[i]) / (ff[i + 1] - ff[i]), up = (ff[i + 2] - f) / (ff[i + 2] - ff[i + 1]);
            const double v = std::min(down, up);
            w[(size_t)b * nmel + i] = (v > 0.0 ? v : 0.0) * 2.0 / (ff[i + 2] - ff[i]);
        }
    }
}

struct Tables {
    int kind = -1, nmel = 0, device = -1, bins_pad = 0, nmel_pad = 0;
    float *wcos = nullptr, *wsin = nullptr, *melw = nullptr;
};
static std::mutex g_tab_mu;
static std::vector<Tables> g_tabs;

static int get_tables(int kind, int nmel, int device, Tables* out) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    for (const Tables& t : g_tabs)
        if (t.kind == kind && t.nmel == nmel && t.device == device) { *out = t; return FAD_OK; }
    int nfft, win, bins; double sr;
    if (kind == FE_VGGISH) { nfft = 512; win = 400; bins = 257; sr = 16000.0; }
    else if (kind == FE_WHISPER) { nfft = 400; win = 400; bins = 201; sr = 16000.0; }
    else { nfft = 1024; win = 1024; bins = 513; sr = 48000.0; }
    Tables t; t.kind = kind; t.nmel = nmel; t.device = device;
    t.bins_pad = (int)cdiv(bins, 16) * 16; t.nmel_pad = (int)cdiv(nmel, 16) * 16;
    std::vector<float> hc((size_t)win * t.bins_pad, 0.f), hs((size_t)win * t.bins_pad, 0.f);
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < win; ++k) {
        const double wk = 0.5 - 0.5 * std::cos(two_pi * k / win);            // periodic Hann
        for (int j = 0; j < bins; ++j) {
            const int64_t ph = ((int64_t)j * k) % nfft;                      // exact phase reduction
            const double ang = two_pi * (double)ph / nfft;
            hc[(size_t)k * t.bins_pad + j] = (float)(wk * std::cos(ang));
            hs[(size_t)k * t.bins_pad + j] = (float)(-wk * std::sin(ang));
        }
    }
    std::vector<double> mw;
    if (kind == FE_VGGISH) mel_htk_vggish(bins, nmel, sr, 125.0, 7500.0, mw);
    else if (kind == FE_WHISPER) mel_slaney(bins, nmel, sr, 0.0, 8000.0, mw);
    else mel_slaney(bins, nmel, sr, 50.0, 14000.0, mw);
    std::vector<float> hm((size_t)t.bins_pad * t.nmel_pad, 0.f);
    for (int b = 0; b < bins; ++b)
        for (int i = 0; i < nmel; ++i) hm[(size_t)b * t.nmel_pad + i] = (float)mw[(size_t)b * nmel + i];
    FAD_HIP_TRY(hipMalloc(&t.wcos, hc.size() * sizeof(float)));
    FAD_HIP_TRY(hipMalloc(&t.wsin, hs.size() * sizeof(float)));
    FAD_HIP_TRY(hipMalloc(&t.melw, hm.size() * sizeof(float)));
    FAD_HIP_TRY(hipMemcpy(t.wcos, hc.data(), hc.size() * sizeof(float), hipMemcpyHostToDevice));
    FAD_HIP_TRY(hipMemcpy(t.wsin, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));
    FAD_HIP_TRY(hipMemcpy(t.melw, hm.data(), hm.size() * sizeof(float), hipMemcpyHostToDevice));
    g_tabs.push_back(t);
    *out = t;
    return FAD_OK;
}

struct FeWorkspace {
    DevBuf wav, meta, out, cmax;
    void release_all() { wav.release(); meta.release(); out.release(); cmax.release(); }
};
static FeWorkspace& fe_ws(int device) {
    static thread_local PerThreadDevice<FeWorkspace> ws;
    return ws.get(device);
}

template <int KIND>
static int run_logmel(const float* wav, const int64_t* offsets, int64_t n_clips, int nmel,
                      const std::vector<int64_t>& frame_base, const std::vector<int64_t>& clip_frames,
                      int64_t out_frames_per_clip, int64_t out_elems, float* out, int on_device, int device,
                      hipStream_t st) {
    using C = Cfg<KIND>;
    Tables tb;
    FAD_TRY(get_tables(KIND, nmel, device, &tb));
    FeWorkspace& ws = fe_ws(device);
    const int64_t total = offsets[n_clips];
    const float* dwav = wav;
    if (!on_device) {
        FAD_TRY(ws.wav.reserve((size_t)(total > 0 ? total : 1) * sizeof(float)));
        if (total > 0) FAD_HIP_TRY(hipMemcpyAsync(ws.wav.p, wav, (size_t)total * sizeof(float), hipMemcpyHostToDevice, st));
        dwav = static_cast<const float*>(ws.wav.p);
    }
    // meta: offsets | frame_base | clip_frames
    const size_t mlen = (size_t)(n_clips + 1) + 2 * (size_t)n_clips;
    std::vector<int64_t> meta(mlen);
    for (int64_t c = 0; c <= n_clips; ++c) meta[c] = offsets[c];
    for (int64_t c = 0; c < n_clips; ++c) { meta[n_clips + 1 + c] = frame_base[c]; meta[2 * n_clips + 1 + c] = clip_frames[c]; }
    FAD_TRY(ws.meta.reserve(mlen * sizeof(int64_t)));
    FAD_HIP_TRY(hipMemcpyAsync(ws.meta.p, meta.data(), mlen * sizeof(int64_t), hipMemcpyHostToDevice, st));
    FAD_HIP_TRY(hipStreamSynchronize(st));                       // `meta` is a stack-lifetime host buffer
    const int64_t* dmeta = static_cast<const int64_t*>(ws.meta.p);

    float* dout = out;
    if (!on_device) {
        FAD_TRY(ws.out.reserve((size_t)(out_elems > 0 ? out_elems : 1) * sizeof(float)));
        dout = static_cast<float*>(ws.out.p);
    }
    int64_t max_frames = 0;
    for (int64_t c = 0; c < n_clips; ++c) max_frames = std::max(max_frames, clip_frames[c]);
    if (max_frames > 0 && out_elems > 0) {
        LogmelArgs a;
        a.wav = dwav; a.offsets = dmeta; a.frame_base = dmeta + n_clips + 1; a.clip_frames = dmeta + 2 * n_clips + 1;
        a.wcos = tb.wcos; a.wsin = tb.wsin; a.melw = tb.melw; a.out = dout; a.clip_max = nullptr;
        a.nmel = nmel; a.nmel_pad = tb.nmel_pad; a.bins_pad = tb.bins_pad; a.out_frames_per_clip = out_frames_per_clip;
        if (KIND == FE_WHISPER) {
            FAD_TRY(ws.cmax.reserve((size_t)n_clips * sizeof(unsigned)));
            FAD_HIP_TRY(hipMemsetAsync(ws.cmax.p, 0, (size_t)n_clips * sizeof(unsigned), st));
            a.clip_max = static_cast<unsigned*>(ws.cmax.p);
        }
        constexpr int SPAN = (FT - 1) * C::HOP + C::WIN;
        constexpr int SPAN_SK = SPAN + 2 * (SPAN / C::HOP) + 2;
        const size_t lds = ((size_t)((SPAN_SK + 3) & ~3) + (size_t)FT * (tb.bins_pad + 1)) * sizeof(float);
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&logmel_kernel<KIND>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int64_t c0 = 0; c0 < n_clips; c0 += 32768) {                 // gridDim.y limit
            const int64_t nc = std::min<int64_t>(32768, n_clips - c0);
            LogmelArgs b = a;
            b.offsets += c0; b.frame_base += c0; b.clip_frames += c0;
            if (KIND != FE_VGGISH) b.out += c0 * out_frames_per_clip * nmel;
            if (b.clip_max) b.clip_max += c0;
            hipLaunchKernelGGL((logmel_kernel<KIND>), dim3((unsigned)cdiv(max_frames, FT), (unsigned)nc), dim3(256), lds, st, b);
            if (KIND == FE_WHISPER) {
                const int64_t per_clip = (int64_t)nmel * out_frames_per_clip;
                hipLaunchKernelGGL(whisper_normalize, dim3((unsigned)cdiv(per_clip, 256), (unsigned)nc), dim3(256), 0, st,
                                   b.out, b.clip_max, per_clip);
            }
        }
        FAD_HIP_TRY(hipGetLastError());
    }
    if (!on_device) {
        if (out_elems > 0) FAD_HIP_TRY(hipMemcpyAsync(out, dout, (size_t)out_elems * sizeof(float), hipMemcpyDeviceToHost, st));
        FAD_HIP_TRY(hipStreamSynchronize(st));
    }
    return FAD_OK;
}

static int check_clips(const float* wav, const int64_t* offsets, int64_t n_clips, const void* out) {
    if (!offsets || n_clips < 0 || !out) return set_error(FAD_ERR_INVALID, "NULL or negative argument");
    for (int64_t c = 0; c < n_clips; ++c)
        if (offsets[c] < 0 || offsets[c] > offsets[c + 1]) return set_error(FAD_ERR_INVALID, "offsets must be non-decreasing");
    if (n_clips > 0 && offsets[n_clips] > 0 && !wav) return set_error(FAD_ERR_INVALID, "wav is NULL");
    return FAD_OK;
}

// (first-use warm-up, common.cpp: warm_code_objects -- loading this translation unit's code object costs ~75 ms at the first launch)
const void* code_object_anchor_logmel() { return reinterpret_cast<const void*>(&whisper_normalize); }
}  // namespace fad

using namespace fad;

extern "C" {

// 16 kHz samples -> number of 0.96 s examples (96 frames of 10 ms, non-overlapping; tail dropped)
int64_t fad_logmel_vggish_num_examples(int64_t n_samples) {
    if (n_samples < 400) return 0;
    const int64_t frames = 1 + (n_samples - 400) / 160;
    return frames < 96 ? 0 : 1 + (frames - 96) / 96;
}

int fad_logmel_vggish(const float* wav, const int64_t* offsets, int64_t n_clips, float* out,
                      int64_t out_capacity_examples, int64_t* example_offsets, int on_device, int device, void* stream) {
    FAD_TRY(check_clips(wav, offsets, n_clips, out));
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    std::vector<int64_t> base(n_clips), frames(n_clips);
    int64_t ex = 0;
    for (int64_t c = 0; c < n_clips; ++c) {
        const int64_t e = fad_logmel_vggish_num_examples(offsets[c + 1] - offsets[c]);
        if (example_offsets) example_offsets[c] = ex;
        base[c] = ex * 96; frames[c] = e * 96; ex += e;
    }
    if (example_offsets) example_offsets[n_clips] = ex;
    if (ex > out_capacity_examples)
        return set_error(FAD_ERR_SHAPE, "output holds %lld examples, %lld needed", (long long)out_capacity_examples, (long long)ex);
    return run_logmel<FE_VGGISH>(wav, offsets, n_clips, 64, base, frames, 0, ex * 96 * 64, out, on_device, device,
                                 static_cast<hipStream_t>(stream));
}

int fad_logmel_whisper(const float* wav, const int64_t* offsets, int64_t n_clips, int n_mels, float* out,
                       int on_device, int device, void* stream) {
    FAD_TRY(check_clips(wav, offsets, n_clips, out));
    if (n_mels != 80 && n_mels != 128) return set_error(FAD_ERR_INVALID, "n_mels must be 80 or 128, got %d", n_mels);
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    std::vector<int64_t> base(n_clips, 0), frames(n_clips, WHISPER_FRAMES);
    return run_logmel<FE_WHISPER>(wav, offsets, n_clips, n_mels, base, frames, WHISPER_FRAMES,
                                  n_clips * (int64_t)n_mels * WHISPER_FRAMES, out, on_device, device,
                                  static_cast<hipStream_t>(stream));
}

int fad_logmel_htsat(const float* wav, const int64_t* offsets, int64_t n_clips, int64_t n_frames_out, float* out,
                     int on_device, int device, void* stream) {
    FAD_TRY(check_clips(wav, offsets, n_clips, out));
    if (n_frames_out < 1) return set_error(FAD_ERR_INVALID, "n_frames_out must be positive");
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    std::vector<int64_t> base(n_clips, 0), frames(n_clips);
    for (int64_t c = 0; c < n_clips; ++c) {
        const int64_t n = offsets[c + 1] - offsets[c];
        if (n < 2) return set_error(FAD_ERR_SHAPE, "clip %lld has %lld samples; reflect padding needs at least 2", (long long)c, (long long)n);
        if (1 + n / 480 != n_frames_out)
            return set_error(FAD_ERR_SHAPE, "clip %lld yields %lld frames, expected %lld (pad or cut clips to one length)",
                             (long long)c, (long long)(1 + n / 480), (long long)n_frames_out);
        frames[c] = n_frames_out;
    }
    return run_logmel<FE_HTSAT>(wav, offsets, n_clips, 64, base, frames, n_frames_out, n_clips * n_frames_out * 64, out,
                                on_device, device, static_cast<hipStream_t>(stream));
}

}  // extern "C"
