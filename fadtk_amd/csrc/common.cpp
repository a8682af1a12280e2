// Error text, device checks -- host-only helpers of libfad_hip.so.
#include "fad_common.h"

#include <cstdlib>
#include <mutex>

namespace fad {

char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static constexpr int kMaxDev = 64;
static int g_cus[kMaxDev];
static char g_arch[kMaxDev][256];
static int g_state[kMaxDev];          // 0 unknown, 1 ok, -1 bad
static std::mutex g_mu;

static int probe(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_state[device] != 0) return g_state[device];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_state[device] = -1; return -1; }
    snprintf(g_arch[device], sizeof(g_arch[device]), "%s", prop.gcnArchName);
    g_cus[device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g_state[device] = (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : -1;
    return g_state[device];
}

static void warm_code_objects(int device);
int check_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return set_error(FAD_ERR_NO_DEVICE, "no HIP device visible: libfad_hip has no CPU fallback");
    if (device < 0 || device >= count || device >= kMaxDev)
        return set_error(FAD_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, count);
    if (probe(device) != 1)
        return set_error(FAD_ERR_NO_DEVICE, "device %d is '%s', this library is built for gfx950 only", device,
                         g_arch[device]);
    warm_code_objects(device);
    return FAD_OK;
}

// The runtime loads a translation unit's code object at the FIRST launch of one of its kernels: ~75 ms each (rocprofv3 --hip-trace around
// scripts/probe_stall.py: one hipLaunchKernel of 75 ms where a route ran a kernel of a unit nothing had touched yet -- the "30-70 ms stalls of
// single blocking calls" of rounds 3-4).  Touching one kernel per unit when a device is first used moves all of that to one place.
// FAD_WARM_KERNELS=0 leaves the loading lazy.
const void* code_object_anchor_moments(); const void* code_object_anchor_gemm_f64(); const void* code_object_anchor_gemm_f32();
const void* code_object_anchor_frechet_f64(); const void* code_object_anchor_frechet(); const void* code_object_anchor_frechet_songs();
const void* code_object_anchor_logmel(); const void* code_object_anchor_resample();
static void warm_code_objects(int device) {
    static bool warm[kMaxDev] = {false};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (warm[device]) return;
        warm[device] = true;
    }
    const char* e = getenv("FAD_WARM_KERNELS");
    if (e && e[0] == '0') return;
    DeviceGuard g(device);
    if (!g.ok) return;
    const void* anchors[] = {code_object_anchor_moments(), code_object_anchor_gemm_f64(), code_object_anchor_gemm_f32(), code_object_anchor_frechet_f64(),
                             code_object_anchor_frechet(), code_object_anchor_frechet_songs(), code_object_anchor_logmel(), code_object_anchor_resample()};
    for (const void* a : anchors) {
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, a) != hipSuccess) (void)hipGetLastError();
    }
}

int num_cus(int device) { return (device >= 0 && device < kMaxDev && g_cus[device] > 0) ? g_cus[device] : 256; }

void NsWorkspace::release() {
    mats.release(); small.release(); stage.release();
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr; pinned_cap = 0;
}

}  // namespace fad

extern "C" {

int fad_version(void) { return FAD_ABI_VERSION; }

int fad_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return 0;
    int ok = 0;
    for (int i = 0; i < count && i < fad::kMaxDev; ++i) ok += (fad::probe(i) == 1);
    return ok;
}

const char* fad_last_error(void) { return fad::err_buf(); }

int fad_stream_create_cu_mask(int device, const uint32_t* mask, int words, void** stream) {
    if (!mask || words < 1 || !stream) return fad::set_error(FAD_ERR_INVALID, "NULL argument");
    FAD_TRY(fad::check_device(device));
    fad::DeviceGuard g(device);
    hipStream_t st = nullptr;
    FAD_HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
    *stream = st;
    return FAD_OK;
}

int fad_stream_destroy(int device, void* stream) {
    if (!stream) return FAD_OK;
    fad::DeviceGuard g(device);
    FAD_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return FAD_OK;
}

const char* fad_device_arch(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || device >= fad::kMaxDev) return "";
    fad::probe(device);
    return fad::g_arch[device];
}

}  // extern "C"
