// Kernel-by-kernel check of fadtk_amd/csrc/ns_fast.h on the GPU against plain host arithmetic (test infrastructure, gfx950).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tests/native/nsfast_check tests/native/nsfast_check.hip && tests/native/nsfast_check [d ...]
// Every kernel of the nine-launch Frechet chain is launched on random operands; the digit planes, the split planes in both
// orientations, the exact (int8 MFMA) products, the split-float16 products with their epilogues and the statistics the next
// kernel consumes are compared with values computed on the host in float64.  Exit code 0 = all checks passed.
#include "../../fadtk_amd/csrc/ns_fast.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace fad;
using namespace fad::nsf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static int g_fail = 0;
static void report(const char* what, double err, double tol) {
    const bool ok = (err <= tol) && (err == err);
    printf("  %-58s err %.3e  (tol %.1e)  %s\n", what, err, tol, ok ? "ok" : "FAIL");
    if (!ok) ++g_fail;
}
template <typename T> static T* dmalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T) + 64)); CK(hipMemset(p, 0xEE, n * sizeof(T))); return p; }
template <typename T> static std::vector<T> d2h(const T* p, size_t n) { std::vector<T> v(n); CK(hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost)); return v; }
template <typename T> static void h2d(T* p, const std::vector<T>& v) { CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); }

static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
static void host_split(float v, uint16_t& hi, uint16_t& lo) {
    _Float16 h = (_Float16)v; _Float16 l = (_Float16)((v - (float)h) * 2048.f);
    memcpy(&hi, &h, 2); memcpy(&lo, &l, 2);
}
static float host_used(uint16_t hi, uint16_t lo) { return h2f(hi) + h2f(lo) * (1.f / 2048.f); }
static double dig_value(const std::vector<int8_t>& dg, int row, int k, int d) {
    const size_t o = dig_off(row, k >> 4, d) + (k & 15);
    double v = 0.0;
    for (int p = 0; p < kDigits; ++p) v += (double)dg[o + 16 * p] * std::ldexp(1.0, 7 * p - 40);
    return v;
}

struct DevSplit { SplitMat m; };
static SplitMat alloc_split(int d) {
    SplitMat m; const size_t dd = (size_t)d * d;
    m.h = reinterpret_cast<_Float16*>(dmalloc<uint16_t>(dd)); m.l = reinterpret_cast<_Float16*>(dmalloc<uint16_t>(dd));
    m.th = reinterpret_cast<_Float16*>(dmalloc<uint16_t>(dd)); m.tl = reinterpret_cast<_Float16*>(dmalloc<uint16_t>(dd));
    return m;
}
// host image of a split matrix: used values (float) of X from the h/l planes and of X^T from the th/tl planes
struct HostSplit { std::vector<float> x, xt; };
static HostSplit fetch_split(const SplitMat& m, int d) {
    const size_t dd = (size_t)d * d;
    auto h = d2h(reinterpret_cast<const uint16_t*>(m.h), dd), l = d2h(reinterpret_cast<const uint16_t*>(m.l), dd);
    auto th = d2h(reinterpret_cast<const uint16_t*>(m.th), dd), tl = d2h(reinterpret_cast<const uint16_t*>(m.tl), dd);
    HostSplit s; s.x.resize(dd); s.xt.resize(dd);
    for (size_t i = 0; i < dd; ++i) { s.x[i] = host_used(h[i], l[i]); s.xt[i] = host_used(th[i], tl[i]); }
    return s;
}
static void upload_split(const SplitMat& m, const std::vector<float>& x, int d) {       // both orientations from a float matrix
    const size_t dd = (size_t)d * d;
    std::vector<uint16_t> h(dd), l(dd), th(dd), tl(dd);
    for (int r = 0; r < d; ++r)
        for (int c = 0; c < d; ++c) {
            uint16_t a, b; host_split(x[(size_t)r * d + c], a, b);
            h[(size_t)r * d + c] = a; l[(size_t)r * d + c] = b; th[(size_t)c * d + r] = a; tl[(size_t)c * d + r] = b;
        }
    h2d(reinterpret_cast<uint16_t*>(m.h), h); h2d(reinterpret_cast<uint16_t*>(m.l), l);
    h2d(reinterpret_cast<uint16_t*>(m.th), th); h2d(reinterpret_cast<uint16_t*>(m.tl), tl);
}
// C = A B in float64 on the values the MFMAs see
static std::vector<double> host_mm(const std::vector<float>& a, const std::vector<float>& b, int d) {
    std::vector<double> c((size_t)d * d, 0.0), bt((size_t)d * d);
    for (int k = 0; k < d; ++k) for (int j = 0; j < d; ++j) bt[(size_t)j * d + k] = b[(size_t)k * d + j];
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            const float* ai = &a[(size_t)i * d]; const double* bj = &bt[(size_t)j * d];
            for (int k = 0; k < d; ++k) s += (double)ai[k] * bj[k];
            c[(size_t)i * d + j] = s;
        }
    return c;
}
static std::vector<double> host_mm64(const std::vector<double>& a, const std::vector<double>& bT, int d) {   // A times (bT)^T
    std::vector<double> c((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0.0;
            const double* ai = &a[(size_t)i * d]; const double* bj = &bT[(size_t)j * d];
            for (int k = 0; k < d; ++k) s += ai[k] * bj[k];
            c[(size_t)i * d + j] = s;
        }
    return c;
}
static double max_abs_diff_T(const std::vector<float>& x, const std::vector<float>& xt, int d) {
    double m = 0.0;
    for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) m = std::fmax(m, std::fabs((double)x[(size_t)r * d + c] - (double)xt[(size_t)c * d + r]));
    return m;
}

template <int NS> static void launch_split(int mode, unsigned t, const SplitArgs& g) {
    if (mode == SP_FIRST) hipLaunchKernelGGL((nsf_split<NS, SP_FIRST>), dim3(t, t, 1), dim3(512), 0, 0, g);
    else if (mode == SP_T) hipLaunchKernelGGL((nsf_split<NS, SP_T>), dim3(t, t, 1), dim3(512), 0, 0, g);
    else hipLaunchKernelGGL((nsf_split<NS, SP_U>), dim3(t, t, 3), dim3(512), 0, 0, g);
}
static void run_split(int d, int mode, const SplitArgs& g) {
    const unsigned t = d / 32;
    switch (d) { case 256: launch_split<2>(mode, t, g); break; case 512: launch_split<4>(mode, t, g); break;
                 case 768: launch_split<6>(mode, t, g); break; default: launch_split<8>(mode, t, g); }
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
}
template <int NS8> static void launch_i8(int mode, unsigned t, const I8Args& g) {
    if (mode == I8_A) hipLaunchKernelGGL((nsf_i8<NS8, I8_A>), dim3(t, t, 2), dim3(512), 0, 0, g);
    else hipLaunchKernelGGL((nsf_i8<NS8, I8_G>), dim3(t, t, 1), dim3(512), 0, 0, g);
}
static void run_i8(int d, int mode, const I8Args& g) {
    const unsigned t = d / 32;
    switch (d) { case 256: launch_i8<1>(mode, t, g); break; case 512: launch_i8<2>(mode, t, g); break;
                 case 768: launch_i8<3>(mode, t, g); break; default: launch_i8<4>(mode, t, g); }
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
}

static void check_dim(int d) {
    printf("== d = %d\n", d);
    const size_t dd = (size_t)d * d;
    const int nb = d / 32;
    std::mt19937_64 rng(1234 + d);
    std::normal_distribution<double> nd(0.0, 1.0);

    // ---------------- K1: packed moments -> covariances, scales, digit planes
    const int n_rows = 3 * d;
    std::vector<double> acc[2];
    std::vector<double> cov_ref[2];
    double scale_ref[2], tr_ref[2];
    for (int s = 0; s < 2; ++s) {
        std::vector<double> x((size_t)n_rows * d);
        const double gain = (s == 0) ? 37.5 : 0.0021;                        // very different units: the scales must absorb them
        for (auto& v : x) v = gain * nd(rng) + (s ? 0.001 : 0.5);
        acc[s].assign(1 + d + dd, 0.0);
        acc[s][0] = n_rows;
        for (int r = 0; r < n_rows; ++r)
            for (int i = 0; i < d; ++i) {
                acc[s][1 + i] += x[(size_t)r * d + i];
            }
        for (int i = 0; i < d; ++i)
            for (int j = i; j < d; ++j) {
                double t = 0.0;
                for (int r = 0; r < n_rows; ++r) t += x[(size_t)r * d + i] * x[(size_t)r * d + j];
                acc[s][1 + d + (size_t)i * d + j] = t; acc[s][1 + d + (size_t)j * d + i] = t;
            }
        cov_ref[s].resize(dd);
        double mx = 0.0; tr_ref[s] = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                const double v = (acc[s][1 + d + (size_t)i * d + j] - (acc[s][1 + i] * acc[s][1 + j]) * (1.0 / n_rows)) * (1.0 / (n_rows - 1.0));
                cov_ref[s][(size_t)i * d + j] = v;
                if (i == j) { mx = std::fmax(mx, v); tr_ref[s] += v; }
            }
        int ex; (void)std::frexp(mx, &ex); scale_ref[s] = std::ldexp(1.0, -ex);
    }
    double* d_acc[2]; for (int s = 0; s < 2; ++s) { d_acc[s] = dmalloc<double>(1 + d + dd); h2d(d_acc[s], acc[s]); }
    double* d_mus = dmalloc<double>(2 * d); double* d_covs = dmalloc<double>(2 * dd);
    int8_t* d_dig[2] = {dmalloc<int8_t>(6 * dd), dmalloc<int8_t>(6 * dd)};
    NsState* d_st = dmalloc<NsState>(1); Ns32State* d_s32 = dmalloc<Ns32State>(1); FastHdr* d_hdr = dmalloc<FastHdr>(1);
    CK(hipMemset(d_st, 0, sizeof(NsState))); CK(hipMemset(d_s32, 0, sizeof(Ns32State)));
    PrepArgs pa; memset(&pa, 0, sizeof(pa));
    pa.acc[0] = d_acc[0]; pa.acc[1] = d_acc[1]; pa.d = d; pa.ddof = 1; pa.mus = d_mus; pa.covs = d_covs; pa.dig[0] = d_dig[0]; pa.dig[1] = d_dig[1];
    pa.st = d_st; pa.s32 = d_s32; pa.hdr = d_hdr;
    hipLaunchKernelGGL(nsf_prepare, dim3((unsigned)(dd / 4096), 2), dim3(256), 0, 0, pa);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    FastHdr hdr = d2h(d_hdr, 1)[0];
    auto covs = d2h(d_covs, 2 * dd);
    std::vector<int8_t> dig[2] = {d2h(d_dig[0], 6 * dd), d2h(d_dig[1], 6 * dd)};
    std::vector<double> Cn[2];                                              // the normalised matrices the digit planes represent
    for (int s = 0; s < 2; ++s) {
        double e_cov = 0.0, e_dig = 0.0, amax = 0.0;
        Cn[s].resize(dd);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) {
                const double ref = cov_ref[s][(size_t)i * d + j];
                e_cov = std::fmax(e_cov, std::fabs(covs[s * dd + (size_t)i * d + j] - ref) / std::fabs(cov_ref[s][(size_t)i * d + i]));
                const double v = dig_value(dig[s], i, j, d);
                Cn[s][(size_t)i * d + j] = v;
                e_dig = std::fmax(e_dig, std::fabs(v - ref * scale_ref[s]));
                amax = std::fmax(amax, std::fabs(v));
            }
        char buf[128];
        snprintf(buf, sizeof(buf), "K1 set %d: covariance vs host formula (relative to diag)", s); report(buf, e_cov, 1e-14);
        snprintf(buf, sizeof(buf), "K1 set %d: digit planes reconstruct s * Sigma (max |.| %.3f)", s, amax); report(buf, e_dig, std::ldexp(1.0, -41) * 1.01);
        snprintf(buf, sizeof(buf), "K1 set %d: scale (power of two)", s); report(buf, std::fabs(hdr.s[s] - scale_ref[s]), 0.0);
        snprintf(buf, sizeof(buf), "K1 set %d: trace", s); report(buf, std::fabs(hdr.tr[s] - tr_ref[s]) / tr_ref[s], 1e-13);
    }
    report("K1: bad flags clear", (double)(hdr.bad[0] | hdr.bad[1]), 0.0);

    // ---------------- K2: A = C1 C2 exact
    double* d_A64 = dmalloc<double>(dd); float* d_P = dmalloc<float>(dd); float* d_Pt = dmalloc<float>(dd);
    double* d_stats = dmalloc<double>(2 * (size_t)nb * d + 8 * (size_t)nb * nb);
    I8Args ia; memset(&ia, 0, sizeof(ia));
    ia.Adig = d_dig[0]; ia.Bdig = d_dig[1]; ia.d = d; ia.hdr = d_hdr; ia.stats = d_stats; ia.A64 = d_A64; ia.P = d_P; ia.Pt = d_Pt;
    ia.st = d_st; ia.mu1 = d_mus; ia.mu2 = d_mus + d; ia.mean_dtype = -1;
    run_i8(d, I8_A, ia);
    auto A64 = d2h(d_A64, dd); auto P = d2h(d_P, dd); auto Pt = d2h(d_Pt, dd);
    auto statsA = d2h(d_stats, (size_t)nb * d + 2 * (size_t)nb * nb);
    const std::vector<double> An = host_mm64(Cn[0], Cn[1], d);              // C2 planes hold C2 rows = columns of C2 (symmetric)
    const double inv_s12 = 1.0 / (hdr.s[0] * hdr.s[1]);
    {
        double e_a = 0.0, e_p = 0.0, e_pt = 0.0, fro2 = 0.0, tr = 0.0, amax = 0.0;
        for (size_t i = 0; i < dd; ++i) {
            e_a = std::fmax(e_a, std::fabs(A64[i] / inv_s12 - An[i]));
            e_p = std::fmax(e_p, std::fabs((double)P[i] - An[i]));
            amax = std::fmax(amax, std::fabs(An[i])); fro2 += An[i] * An[i];
        }
        for (int r = 0; r < d; ++r) { tr += An[(size_t)r * d + r]; for (int c = 0; c < d; ++c) e_pt = std::fmax(e_pt, std::fabs((double)Pt[(size_t)c * d + r] - An[(size_t)r * d + c])); }
        report("K2: A (float64, normalised units) vs host product of the digit values", e_a, 4e-15 * d / 512.0 + 1e-15);
        report("K2: P = float32 image", e_p / amax, 6.1e-8);
        report("K2: Pt = its transpose", e_pt / amax, 6.1e-8);
        double f2 = 0.0, t2 = 0.0, e_row = 0.0;
        for (int k = 0; k < nb * nb; ++k) { f2 += statsA[(size_t)nb * d + 2 * k]; t2 += statsA[(size_t)nb * d + 2 * k + 1]; }
        for (int r = 0; r < d; ++r) {
            double want = 0.0, got = 0.0;
            for (int c = 0; c < d; ++c) want += std::fabs(An[(size_t)r * d + c]);
            for (int k = 0; k < nb; ++k) got += statsA[(size_t)k * d + r];
            e_row = std::fmax(e_row, std::fabs(got - want) / want);
        }
        report("K2: sum a^2", std::fabs(f2 - fro2) / fro2, 1e-13);
        report("K2: tr A", std::fabs(t2 - tr) / std::fabs(tr), 1e-13);
        report("K2: row sums of |A|", e_row, 1e-13);
        // the mean term of the spare workgroup (float64 means)
        auto mus = d2h(d_mus, 2 * d); NsState st = d2h(d_st, 1)[0];
        double mt = 0.0; for (int i = 0; i < d; ++i) { const double g = mus[i] - mus[d + i]; mt += g * g; }
        report("K2: mean term (spare workgroup)", std::fabs(st.mean_term - mt) / mt, 1e-13);
    }

    // ---------------- K3: iteration 0 from P, Pt and K2's statistics.  Use a well-conditioned stand-in for A so that the kernel's
    // own rule accepts it: A := I + small noise (written into P / Pt / stats by re-running K2's epilogue arithmetic on the host)
    std::vector<float> Pw(dd), Ptw(dd);
    std::vector<double> statsW((size_t)nb * d + 2 * (size_t)nb * nb, 0.0);
    {
        for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
            const float v = (float)((r == c ? 0.8 + 0.4 * ((r * 37) % 11) / 11.0 : 0.0) + 0.004 * nd(rng));
            Pw[(size_t)r * d + c] = v; Ptw[(size_t)c * d + r] = v;
        }
        for (int ty = 0; ty < nb; ++ty) for (int tx = 0; tx < nb; ++tx) {
            double sq = 0.0, tr = 0.0;
            for (int rr = 0; rr < 32; ++rr) {
                double ra = 0.0;
                for (int cc = 0; cc < 32; ++cc) { const double v = Pw[(size_t)(ty * 32 + rr) * d + tx * 32 + cc]; sq += v * v; ra += std::fabs(v); if (ty * 32 + rr == tx * 32 + cc) tr += v; }
                statsW[(size_t)tx * d + ty * 32 + rr] = ra;
            }
            statsW[(size_t)nb * d + 2 * (ty * nb + tx)] = sq; statsW[(size_t)nb * d + 2 * (ty * nb + tx) + 1] = tr;
        }
        h2d(d_P, Pw); h2d(d_Pt, Ptw);
        std::vector<double> full(2 * (size_t)nb * d + 8 * (size_t)nb * nb, 0.0);
        std::copy(statsW.begin(), statsW.end(), full.begin());
        h2d(d_stats, full);
    }
    SplitMat Y[2] = {alloc_split(d), alloc_split(d)}, Z[2] = {alloc_split(d), alloc_split(d)}, T = alloc_split(d);
    int8_t* d_digY[2] = {dmalloc<int8_t>(6 * dd), dmalloc<int8_t>(6 * dd)}; int8_t* d_digYt[2] = {dmalloc<int8_t>(6 * dd), dmalloc<int8_t>(6 * dd)};
    double* d_part = dmalloc<double>((size_t)nb * nb);
    SplitArgs g; memset(&g, 0, sizeof(g));
    g.d = d; g.hdr = d_hdr; g.st = d_st; g.s32 = d_s32;
    g.C[0] = Y[1]; g.C[1] = Z[1]; g.Cdig[0] = d_digY[1]; g.Cdig_t[0] = d_digYt[1]; g.P = d_P; g.Pt = d_Pt; g.statsA = d_stats;
    run_split(d, SP_FIRST, g);
    NsState st = d2h(d_st, 1)[0]; Ns32State s32 = d2h(d_s32, 1)[0];
    // host: the scale by the same rule
    double c_ref;
    {
        double fro2 = 0.0, tr = 0.0, inf = 0.0;
        for (int r = 0; r < d; ++r) { double ra = 0.0; for (int c = 0; c < d; ++c) { const double v = Pw[(size_t)r * d + c]; fro2 += v * v; ra += std::fabs(v); } inf = std::fmax(inf, ra); tr += Pw[(size_t)r * d + r]; }
        double u = std::sqrt(fro2); if (inf < u) u = inf;
        c_ref = u / 2.5; const double wm = fro2 / tr; if (wm > c_ref && wm <= u) c_ref = wm;
    }
    report("K3: scale c (caller's units)", std::fabs(st.c - c_ref * inv_s12) / (c_ref * inv_s12), 1e-12);
    report("K3: state armed (s32 live, not failed)", (double)(s32.failed | s32.done | s32.finished | st.done | st.nonfinite), 0.0);
    std::vector<float> Y0(dd), T0(dd);
    const float invf = (float)(1.0 / (st.c / inv_s12));
    for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
        Y0[(size_t)r * d + c] = Pw[(size_t)r * d + c] * invf;
        T0[(size_t)r * d + c] = ((r == c) ? 1.5f : 0.f) - 0.5f * (Pw[(size_t)r * d + c] * invf);
    }
    auto usedm = [&](const std::vector<float>& x) { std::vector<float> u(x.size()); for (size_t i = 0; i < x.size(); ++i) { uint16_t a, b; host_split(x[i], a, b); u[i] = host_used(a, b); } return u; };
    {
        const std::vector<double> Y1 = host_mm(usedm(Y0), usedm(T0), d);
        HostSplit y1 = fetch_split(Y[1], d), z1 = fetch_split(Z[1], d);
        const std::vector<float> uT0 = usedm(T0);
        double e = 0.0, ez = 0.0;
        for (size_t i = 0; i < dd; ++i) { e = std::fmax(e, std::fabs((double)y1.x[i] - Y1[i])); ez = std::fmax(ez, std::fabs((double)z1.x[i] - (double)uT0[i])); }
        report("K3: Y1 = Y0 T0 (split planes) vs float64 product of the split operands", e, 3e-6);
        report("K3: Z1 = T0 (split planes)", ez, 0.0);
        report("K3: Y1^T planes are the transpose", max_abs_diff_T(y1.x, y1.xt, d), 0.0);
        report("K3: Z1^T planes are the transpose", max_abs_diff_T(z1.x, z1.xt, d), 0.0);
        auto dgy = d2h(d_digY[1], 6 * dd), dgt = d2h(d_digYt[1], 6 * dd);
        double ed = 0.0, edt = 0.0;
        for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
            ed = std::fmax(ed, std::fabs(dig_value(dgy, r, c, d) - (double)y1.x[(size_t)r * d + c]));
            edt = std::fmax(edt, std::fabs(dig_value(dgt, c, r, d) - (double)y1.x[(size_t)r * d + c]));
        }
        report("K3: digit planes of Y1", ed, std::ldexp(1.0, -41) * 1.01);
        report("K3: digit planes of Y1^T", edt, std::ldexp(1.0, -41) * 1.01);
    }

    // ---------------- K4: T = 1.5 I - 0.5 Z Y with residual partials (operands: what K3 left)
    HostSplit hy = fetch_split(Y[1], d), hz = fetch_split(Z[1], d);
    memset(&g, 0, sizeof(g));
    g.d = d; g.hdr = d_hdr; g.st = d_st; g.s32 = d_s32;
    g.A[0] = Z[1]; g.B[0] = Y[1]; g.C[0] = T; g.alpha = -0.5f; g.beta_eye = 1.5f; g.gamma = 1.0f; g.partials = d_part; g.skip = &d_s32->done;
    run_split(d, SP_T, g);
    HostSplit ht = fetch_split(T, d);
    double res_ref = 0.0;
    {
        const std::vector<double> ZY = host_mm(hz.x, hy.x, d);
        double e = 0.0, ss = 0.0;
        for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
            const double want = (r == c ? 1.5 : 0.0) - 0.5 * ZY[(size_t)r * d + c];
            e = std::fmax(e, std::fabs((double)ht.x[(size_t)r * d + c] - want));
            const double q = want - (r == c ? 1.0 : 0.0); ss += q * q;
        }
        auto part = d2h(d_part, (size_t)nb * nb);
        double got = 0.0; for (double v : part) got += v;
        res_ref = 2.0 * std::sqrt(ss);
        report("K4: T = (3I - Z Y)/2", e, 3e-6);
        report("K4: T^T planes are the transpose", max_abs_diff_T(ht.x, ht.xt, d), 0.0);
        report("K4: residual partials (sum (T - I)^2, relative)", std::fabs(got - ss) / ss, 1e-3);
    }

    // ---------------- K5: Y <- Y T, Z <- T Z, check of iteration 1, digits of the new Y
    memset(&g, 0, sizeof(g));
    g.d = d; g.hdr = d_hdr; g.st = d_st; g.s32 = d_s32;
    g.A[0] = Y[1]; g.B[0] = T; g.C[0] = Y[0]; g.A[1] = T; g.B[1] = Z[1]; g.C[1] = Z[0];
    g.Cdig[0] = d_digY[0]; g.Cdig_t[0] = d_digYt[0]; g.skip = &d_s32->upd_skip[1];
    g.k = 1; g.max_low = 14; g.nslots = nb * nb; g.chk_partials = d_part; g.thr_pred = 2.5e-3 * d / 512.0;
    run_split(d, SP_U, g);
    HostSplit hy2 = fetch_split(Y[0], d), hz2 = fetch_split(Z[0], d);
    {
        const std::vector<double> YT = host_mm(hy.x, ht.x, d), TZ = host_mm(ht.x, hz.x, d);
        double e = 0.0, e2 = 0.0;
        for (size_t i = 0; i < dd; ++i) { e = std::fmax(e, std::fabs((double)hy2.x[i] - YT[i])); e2 = std::fmax(e2, std::fabs((double)hz2.x[i] - TZ[i])); }
        report("K5: Y' = Y T", e, 3e-6);
        report("K5: Z' = T Z", e2, 3e-6);
        report("K5: Y'^T / Z'^T planes are the transposes", std::fmax(max_abs_diff_T(hy2.x, hy2.xt, d), max_abs_diff_T(hz2.x, hz2.xt, d)), 0.0);
        auto dgy = d2h(d_digY[0], 6 * dd), dgt = d2h(d_digYt[0], 6 * dd);
        double ed = 0.0;
        for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c)
            ed = std::fmax(ed, std::fmax(std::fabs(dig_value(dgy, r, c, d) - (double)hy2.x[(size_t)r * d + c]), std::fabs(dig_value(dgt, c, r, d) - (double)hy2.x[(size_t)r * d + c])));
        report("K5: digit planes of Y' and Y'^T", ed, std::ldexp(1.0, -41) * 1.01);
        Ns32State s2 = d2h(d_s32, 1)[0];
        report("K5: check block recorded the residual of iteration 1", std::fabs(s2.res[1] - res_ref) / res_ref, 1e-3);
        printf("      (residual %.3e, finished %d, final_iter %d, decided_at %d)\n", s2.res[1], s2.finished, s2.final_iter, s2.decided_at);
    }

    // ---------------- K8: exact G = Y Y on iterate 0 (sel even) with R = A/c - G folded into the statistics
    {
        // a float64 "A" such that R is small but not zero: A := c (Y Y + noise)
        Ns32State s2 = d2h(d_s32, 1)[0]; s2.final_iter = 2; s2.skip_corr = 0; s2.ok = 1; h2d(d_s32, std::vector<Ns32State>{s2});
        const std::vector<double> G = host_mm(hy2.x, hy2.x, d);
        std::vector<double> Aw(dd);
        for (size_t i = 0; i < dd; ++i) Aw[i] = st.c * (G[i] + 1e-7 * nd(rng));
        h2d(d_A64, Aw);
        I8Args ig; memset(&ig, 0, sizeof(ig));
        ig.Adig = d_digY[0]; ig.Bdig = d_digYt[0]; ig.Adig_alt = d_digY[1]; ig.Bdig_alt = d_digYt[1]; ig.sel = &d_s32->final_iter;
        ig.d = d; ig.hdr = d_hdr; ig.skip = &d_s32->skip_corr; ig.stats = d_stats; ig.st = d_st; ig.A64in = d_A64;
        ig.Y[0] = Y[0]; ig.Y[1] = Y[1]; ig.Z[0] = Z[0]; ig.Z[1] = Z[1];
        run_i8(d, I8_G, ig);
        auto sg = d2h(d_stats, 2 * (size_t)nb * d + 4 * (size_t)nb * nb);
        double corr = 0.0, r2 = 0.0, trY = 0.0, c_got = 0.0, r_got = 0.0, t_got = 0.0;
        for (int r = 0; r < d; ++r) {
            trY += hy2.x[(size_t)r * d + r];
            for (int c = 0; c < d; ++c) {
                const double R = Aw[(size_t)r * d + c] / st.c - G[(size_t)r * d + c];
                corr += (double)hz2.x[(size_t)c * d + r] * R; r2 += R * R;
            }
        }
        for (int k = 0; k < nb * nb; ++k) { c_got += sg[2 * (size_t)nb * d + 4 * k]; r_got += sg[2 * (size_t)nb * d + 4 * k + 1]; t_got += sg[2 * (size_t)nb * d + 4 * k + 2]; }
        report("K8: tr(Z R) (absolute, |R| ~ 1e-7)", std::fabs(c_got - corr), 2e-11 * d / 512.0);
        report("K8: ||R||_F^2", std::fabs(r_got - r2) / r2, 1e-5);
        report("K8: tr Y", std::fabs(t_got - trY), 1e-10);
        double e_row = 0.0, e_col = 0.0;
        for (int i = 0; i < d; ++i) {
            double wr = 0.0, wc = 0.0, gr = 0.0, gc = 0.0;
            for (int j = 0; j < d; ++j) { wr += std::fabs((double)hz2.x[(size_t)i * d + j]); wc += std::fabs((double)hz2.x[(size_t)j * d + i]); }
            for (int k = 0; k < nb; ++k) { gr += sg[(size_t)k * d + i]; gc += sg[(size_t)nb * d + (size_t)k * d + i]; }
            e_row = std::fmax(e_row, std::fabs(gr - wr) / wr); e_col = std::fmax(e_col, std::fabs(gc - wc) / wc);
        }
        report("K8: row sums of |Z|", e_row, 1e-12);
        report("K8: column sums of |Z|", e_col, 1e-12);
    }
}

int main(int argc, char** argv) {
    std::vector<int> dims;
    for (int i = 1; i < argc; ++i) dims.push_back(atoi(argv[i]));
    if (dims.empty()) dims = {512, 256, 768};
    for (int d : dims) check_dim(d);
    printf(g_fail ? "FAILED: %d check(s)\n" : "all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
