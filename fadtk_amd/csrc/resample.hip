// Kaiser-windowed sinc resampling of mono audio on the GPU (gfx950).
//
// Replaces the torchaudio call of FrechetAudioDistance.load_audio, fadtk/fad.py:151-159:
//     torchaudio.transforms.Resample(fs, model_sr, lowpass_filter_width=64, rolloff=0.9475937167399596,
//                                    resampling_method="sinc_interp_kaiser", beta=14.769656459379492)
// and, optionally, the 16-bit PCM round trip that follows it (fad.py:160 saves PCM_S/16, model_loader.py:64 reads
// int16 / 32768).  torchaudio is a third-party dependency that is not under /root/reference; the algorithm below is
// its published polyphase scheme (functional.resample -> _get_sinc_resample_kernel / _apply_sinc_resample_kernel):
//   orig, new = sr / gcd;  base = min(orig, new) * rolloff;  width = ceil(lowpass_filter_width * orig / base)
//   K[p][j]   = sinc(pi t) * kaiser(t) * base / orig,   t = clamp((-p/new + (j - width)/orig) * base, +-lpfw)
//   out[f*new + p] = sum_j K[p][j] * padded[f*orig + j],   padded = width zeros + wav + (width + orig) zeros
//   length ceil(new * n / orig).
// The table is built in fp64 on the host and rounded to fp32 (as torchaudio does); the kernel accumulates in fp32.
//
// Kernel: one thread per (frame group, phase), FB consecutive frames each: the samples the workgroup's frames touch
// are staged in LDS once, the table is stored tap-major so the phases of a workgroup read consecutive floats, and
// each table value is used for FB multiply-adds.
#include "fad_common.h"

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace fad {

constexpr double kLowpassWidth = 64.0;
constexpr double kRolloff = 0.9475937167399596;
constexpr double kBeta = 14.769656459379492;
constexpr int64_t kMaxTableBytes = (int64_t)256 << 20;
constexpr int kMaxSpanFloats = 16384;               // 64 KiB of LDS

template <int FB>
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ wav, int64_t n,
                                                       const float* __restrict__ KT, int orig, int nnew, int width,
                                                       int taps, int groups, int64_t frames, int64_t n_out,
                                                       float* __restrict__ out, int quantize) {
    // thread -> (frame group, phase): with few phases (48 kHz -> 16 kHz has ONE) a workgroup covers `groups` runs of
    // FB frames instead of idling 255 lanes; blockIdx.y tiles the phases when there are more than 256
    extern __shared__ float span[];                 // padded[f0*orig .. f0*orig + (groups*FB-1)*orig + taps)
    const int tid = threadIdx.x;
    const int64_t f0 = (int64_t)blockIdx.x * groups * FB;
    const int span_len = (groups * FB - 1) * orig + taps;
    for (int s = tid; s < span_len; s += 256) {
        const int64_t i = f0 * orig + s - width;
        span[s] = (i >= 0 && i < n) ? wav[i] : 0.f;
    }
    __syncthreads();
    const int per = nnew < 256 ? nnew : 256;        // phases handled by one workgroup row
    const int grp = tid / per;
    const int p = blockIdx.y * 256 + tid % per;
    if (grp >= groups || p >= nnew) return;
    float acc[FB];
#pragma unroll
    for (int fr = 0; fr < FB; ++fr) acc[fr] = 0.f;
    const float* kt = KT + p;
    const float* sp = span + grp * FB * orig;
    for (int j = 0; j < taps; ++j) {
        const float k = kt[(int64_t)j * nnew];
#pragma unroll
        for (int fr = 0; fr < FB; ++fr) acc[fr] = fmaf(k, sp[fr * orig + j], acc[fr]);
    }
#pragma unroll
    for (int fr = 0; fr < FB; ++fr) {
        const int64_t f = f0 + grp * FB + fr, o = f * nnew + p;
        if (f >= frames || o >= n_out) continue;
        float v = acc[fr];
        if (quantize) v = fminf(fmaxf(rintf(v * 32768.f), -32768.f), 32767.f) * (1.f / 32768.f);
        out[o] = v;
    }
}

__global__ __launch_bounds__(256) void passthrough_kernel(const float* __restrict__ wav, int64_t n, float* __restrict__ out,
                                                          int quantize) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = wav[i];
    if (quantize) v = fminf(fmaxf(rintf(v * 32768.f), -32768.f), 32767.f) * (1.f / 32768.f);
    out[i] = v;
}

struct ResampleTable { float* dev = nullptr; int orig = 0, nnew = 0, width = 0, taps = 0; };

static int64_t gcd64(int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; }

// Tap-major table KT[j][p] on `device`, cached per (orig, new, device) for the life of the process.
static int get_table(int orig_sr, int new_sr, int device, hipStream_t stream, ResampleTable* out) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, ResampleTable> cache;
    const int64_t g = gcd64(orig_sr, new_sr);
    const int orig = (int)(orig_sr / g), nnew = (int)(new_sr / g);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({orig, nnew, device});
    if (it != cache.end()) { *out = it->second; return FAD_OK; }
    const double base = (double)(orig < nnew ? orig : nnew) * kRolloff;
    const int width = (int)std::ceil(kLowpassWidth * orig / base);
    const int64_t taps = 2 * (int64_t)width + orig;
    if (taps * nnew * (int64_t)sizeof(float) > kMaxTableBytes || taps > kMaxSpanFloats)
        return set_error(FAD_ERR_INVALID, "resampling %d -> %d Hz needs a %lld x %d filter table; use rates with a larger "
                         "common divisor", orig_sr, new_sr, (long long)taps, nnew);
    std::vector<float> kt((size_t)taps * nnew);
    const double i0_beta = std::cyl_bessel_i(0.0, kBeta);
    for (int p = 0; p < nnew; ++p)
        for (int64_t j = 0; j < taps; ++j) {
            double t = (-(double)p / nnew + (double)(j - width) / orig) * base;
            t = t < -kLowpassWidth ? -kLowpassWidth : (t > kLowpassWidth ? kLowpassWidth : t);
            const double r = t / kLowpassWidth;
            const double window = std::cyl_bessel_i(0.0, kBeta * std::sqrt(1.0 - r * r)) / i0_beta;
            const double x = t * M_PI;
            const double sinc = (x == 0.0) ? 1.0 : std::sin(x) / x;
            kt[(size_t)j * nnew + p] = (float)(sinc * window * (base / orig));
        }
    ResampleTable tb;
    tb.orig = orig; tb.nnew = nnew; tb.width = width; tb.taps = (int)taps;
    FAD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&tb.dev), kt.size() * sizeof(float)));
    FAD_HIP_TRY(hipMemcpyAsync(tb.dev, kt.data(), kt.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    FAD_HIP_TRY(hipStreamSynchronize(stream));      // kt leaves scope
    cache[{orig, nnew, device}] = tb;
    *out = tb;
    return FAD_OK;
}

// (first-use warm-up, common.cpp: warm_code_objects -- loading this translation unit's code object costs ~75 ms at the first launch)
const void* code_object_anchor_resample() { return reinterpret_cast<const void*>(&passthrough_kernel); }
}  // namespace fad

using namespace fad;

extern "C" {

int64_t fad_resample_num_samples(int64_t n, int orig_sr, int new_sr) {
    if (n < 0 || orig_sr <= 0 || new_sr <= 0) return set_error(FAD_ERR_INVALID, "n=%lld orig_sr=%d new_sr=%d", (long long)n, orig_sr, new_sr);
    const int64_t g = gcd64(orig_sr, new_sr);
    const int64_t orig = orig_sr / g, nnew = new_sr / g;
    return (nnew * n + orig - 1) / orig;            // ceil(new * n / orig)
}

int fad_resample_kaiser(const float* wav, int64_t n, int orig_sr, int new_sr, int quantize_pcm16, float* out,
                        int64_t out_capacity, int on_device, int device, void* stream) {
    const int64_t n_out = fad_resample_num_samples(n, orig_sr, new_sr);
    if (n_out < 0) return (int)n_out;
    if ((n > 0 && !wav) || (n_out > 0 && !out)) return set_error(FAD_ERR_INVALID, "NULL argument");
    if (out_capacity < n_out) return set_error(FAD_ERR_SHAPE, "output holds %lld samples, %lld needed", (long long)out_capacity, (long long)n_out);
    if (n_out == 0) return FAD_OK;
    FAD_TRY(check_device(device));
    DeviceGuard g(device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ResampleTable tb;
    if (orig_sr != new_sr) FAD_TRY(get_table(orig_sr, new_sr, device, st, &tb));

    struct Stage { DevBuf in, out; void release_all() { in.release(); out.release(); } };
    static thread_local PerThreadDevice<Stage> stages;
    Stage& stg = stages.get(device);
    DevBuf& stage_in = stg.in; DevBuf& stage_out = stg.out;
    const float* dwav = wav; float* dout = out;
    if (!on_device) {
        FAD_TRY(stage_in.reserve((size_t)n * sizeof(float)));
        FAD_TRY(stage_out.reserve((size_t)n_out * sizeof(float)));
        FAD_HIP_TRY(hipMemcpyAsync(stage_in.p, wav, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
        dwav = static_cast<const float*>(stage_in.p); dout = static_cast<float*>(stage_out.p);
    }
    if (orig_sr == new_sr) {                        // torchaudio returns the waveform untouched
        hipLaunchKernelGGL(passthrough_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, dwav, n, dout, quantize_pcm16);
        FAD_HIP_TRY(hipGetLastError());
        if (!on_device) {
            FAD_HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost, st));
            FAD_HIP_TRY(hipStreamSynchronize(st));
        }
        return FAD_OK;
    }
    const int64_t frames = n / tb.orig + 1;         // conv1d output length over the padded signal
    const unsigned gy = (unsigned)cdiv(tb.nnew, 256);
    const bool fb8 = 7 * (int64_t)tb.orig + tb.taps <= kMaxSpanFloats;
    const int fb = fb8 ? 8 : 1;
    // frame groups per workgroup: as many as the 256 lanes and the 64 KiB of LDS allow
    int64_t groups = tb.nnew < 256 ? 256 / tb.nnew : 1;
    const int64_t by_lds = ((int64_t)kMaxSpanFloats - tb.taps + tb.orig) / ((int64_t)fb * tb.orig);
    if (groups > by_lds) groups = by_lds;
    if (groups < 1) groups = 1;
    const size_t lds = (size_t)((groups * fb - 1) * tb.orig + tb.taps) * sizeof(float);
    const dim3 grid((unsigned)cdiv(frames, groups * fb), gy);
    if (fb8) {
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&resample_kernel<8>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxSpanFloats * sizeof(float))));
        hipLaunchKernelGGL(resample_kernel<8>, grid, dim3(256), lds, st, dwav, n, tb.dev, tb.orig, tb.nnew, tb.width, tb.taps,
                           (int)groups, frames, n_out, dout, quantize_pcm16);
    } else {
        FAD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&resample_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxSpanFloats * sizeof(float))));
        hipLaunchKernelGGL(resample_kernel<1>, grid, dim3(256), lds, st, dwav, n, tb.dev, tb.orig, tb.nnew, tb.width, tb.taps,
                           (int)groups, frames, n_out, dout, quantize_pcm16);
    }
    FAD_HIP_TRY(hipGetLastError());
    if (!on_device) {
        FAD_HIP_TRY(hipMemcpyAsync(out, dout, (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost, st));
        FAD_HIP_TRY(hipStreamSynchronize(st));
    }
    return FAD_OK;
}

}  // extern "C"
