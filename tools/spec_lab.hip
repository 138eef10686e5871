// Kernel laboratory for the spectral preconditioner: the three 2-D kernels of one application (forward rows, columns + symbol,
// inverse rows with the x update) on synthetic data, each timed with HIP events, stand-alone (no handle, no solver).
// Variants under test live in tools/spec_lab_variants.hip.h; what wins is moved into ksfd_amd/csrc/spectral.hip.h.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_scratch/spec_lab tools/spec_lab.hip && tools/_scratch/spec_lab 4096 [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#define KSPEC_LAB_MINIMAL
#include "../ksfd_amd/csrc/pointwise.hip.h"
__device__ __forceinline__ double2 ksfd_ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
__device__ __forceinline__ double2 ksfd_ld2(const float *p) { const float2 t = *reinterpret_cast<const float2 *>(p); return make_double2((double)t.x, (double)t.y); }
#include "../ksfd_amd/csrc/spectral.hip.h"
#include "../ksfd_amd/csrc/spectral_plan.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <typename T> static T *up(const std::vector<T> &h)
{
    T *d = nullptr;
    CK(hipMalloc((void **)&d, sizeof(T) * h.size()));
    CK(hipMemcpy(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return d;
}

struct Lab {
    int n, F = 2, npair = 1, rb, ntiles, lg_rb, nblk_cols, thr_rows, thr_cols, lg_pl;
    KFFTPlan px, py;
    size_t lds_rows, lds_cols;
    long long plane;
    float *r32; double *x, *x0; kcf *W, *W2, *twx, *twy; int *posy, *kyofpos; int4 *pairtab; float *lx, *ly; int2 *ytab; int lgw = -1;
    KSpecSym Y;
    KSpecLin ex, add;
    hipStream_t st;
    hipEvent_t ea, eb;
    double last_min = 0.0;
};

static void lab_init(Lab &L, int n)
{
    L.n = n; L.plane = (long long)n * n;
    if (!spec_plan(n, L.px) || !spec_plan(n, L.py)) { fprintf(stderr, "no plan for %d\n", n); exit(2); }
    const size_t row_bytes = sizeof(kcf) * spec_sstride(L.px);
    int rb = (int)std::min<size_t>((160 * 1024 - 1024) / row_bytes, 16);
    while (rb & (rb - 1)) rb &= rb - 1;
    while (rb > 1 && (n % rb)) rb >>= 1;
    while (rb > 2 && n / rb < 512) rb >>= 1;
    L.rb = rb; L.ntiles = n / rb; L.lg_rb = 0; while ((1 << L.lg_rb) < rb) L.lg_rb++;
    L.lds_rows = row_bytes * rb;
    L.lds_cols = sizeof(kcf) * spec_sstride(L.py) * 2 * L.npair;
    L.thr_rows = (int)std::min<long long>(1024, std::max<long long>(256, (long long)rb * n / 16));
    L.thr_cols = (int)std::min<long long>(512, std::max<long long>(128, (long long)2 * L.npair * n / 16));
    std::vector<float> hr((size_t)L.F * L.plane);
    std::vector<double> hx((size_t)L.F * L.plane);
    srand(7);
    for (auto &v : hr) v = (float)(rand() / (double)RAND_MAX - 0.5);
    for (auto &v : hx) v = rand() / (double)RAND_MAX - 0.5;
    L.r32 = up(hr); L.x = up(hx); L.x0 = up(hx);
    L.lg_pl = 0; while ((1 << L.lg_pl) < n) L.lg_pl++;              // column stride of W = 2^lg_pl (as spec_build: padded for 3 * 2^k rows)
    CK(hipMalloc((void **)&L.W, sizeof(kcf) * L.npair * (size_t)n * ((size_t)1 << L.lg_pl)));
    CK(hipMalloc((void **)&L.W2, sizeof(kcf) * L.npair * L.plane));
    L.twx = up(spec_twiddles(L.px)); L.twy = up(spec_twiddles(L.py));
    L.posy = up(spec_positions(L.py)); L.kyofpos = up(spec_inverse(spec_positions(L.py)));
    const double h = 4.0 / 1536, ih2 = 1.0 / (h * h);
    L.lx = up(spec_symbol_table(n, ih2)); L.ly = up(spec_symbol_table(n, ih2));
    L.ytab = up(spec_partner_table(L.py, spec_symbol_table(n, ih2)));
    const std::vector<int> posx = spec_positions(L.px);
    std::vector<int4> pairs;
    for (int kx = 0; kx <= n / 2; kx++) {
        const int kxm = (n - kx) % n, j = posx[kx], jm = posx[kxm];
        if (kx == 0) pairs.push_back(make_int4(posx[0], posx[n / 2], 0, n / 2 + 1));
        else if (kx != n / 2) pairs.push_back(make_int4(j, jm, kx, 0));
    }
    std::sort(pairs.begin(), pairs.end(), [](const int4 &a, const int4 &b) { return a.x < b.x; });
    L.nblk_cols = (int)pairs.size();
    L.pairtab = up(pairs);
    memset(&L.Y, 0, sizeof L.Y);
    L.Y.nlig = 1; L.Y.shift = 11.6f; L.Y.a_rr = 2.8e-4f; L.Y.scale = (float)(1.0 / ((double)n * n)); L.Y.den_floor = 0.02f * 11.6f;
    L.Y.a_rU[0] = -3.1e-3f; L.Y.s[0] = 0.01f; L.Y.gam[0] = 0.01f; L.Y.D[0] = 1e-6f;
    memset(&L.ex, 0, sizeof L.ex); memset(&L.add, 0, sizeof L.add);
    L.add.n = 1; L.add.p[0] = L.x; L.add.a[0] = 1.0;
    CK(hipStreamCreate(&L.st));
    CK(hipEventCreate(&L.ea)); CK(hipEventCreate(&L.eb));
    CK(hipFuncSetAttribute((const void *)k_spec_rows_fwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_rows));
    CK(hipFuncSetAttribute((const void *)k_spec_rows_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_rows));
    CK(hipFuncSetAttribute((const void *)k_spec_cols<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_cols));
}

// median over 7 batches of `reps` launches (the clock the chip holds drifts by several per cent from batch to batch)
template <typename FN> static double timeit(Lab &L, int reps, FN fn)
{
    for (int i = 0; i < 3; i++) fn();
    CK(hipStreamSynchronize(L.st));
    double t[7];
    for (int b = 0; b < 7; b++) {
        CK(hipEventRecord(L.ea, L.st));
        for (int i = 0; i < reps; i++) fn();
        CK(hipEventRecord(L.eb, L.st));
        CK(hipEventSynchronize(L.eb));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, L.ea, L.eb));
        t[b] = 1e3 * ms / reps;
    }
    CK(hipGetLastError());
    std::sort(t, t + 7);
    L.last_min = t[0];
    return t[3];
}

// baseline launches, exactly as spec_apply issues them on one rank (tile-major forward store, fused edge stages)
static void base_fwd(Lab &L, int flags = 7, bool nofft = false)
{
    KFFTPlan p = L.px; p.flags = flags; p.lgw = L.lgw; if (nofft) p.nstage = 0;
    hipLaunchKernelGGL(k_spec_rows_fwd<float>, dim3(L.ntiles, L.npair), dim3(L.thr_rows), L.lds_rows, L.st, p, L.n, L.rb, -L.ntiles, L.F, (const float *)L.r32, L.plane, L.W2, (const kcf *)L.twx, L.ex);
}
static void base_cols(Lab &L, int flags = 7, bool nofft = false)
{
    KFFTPlan p = L.py; p.flags = flags; p.lgw = L.lgw; if (nofft) p.nstage = 0;
    const int lg_pl = L.lg_pl;
    hipLaunchKernelGGL(k_spec_cols<1>, dim3(L.nblk_cols), dim3(L.thr_cols), L.lds_cols, L.st, p, L.n, lg_pl, (long long)L.npair * L.n << lg_pl, L.W, (const kcf *)L.W2, L.lg_rb,
                       (const kcf *)L.twy, (const int4 *)L.pairtab, (const int *)L.posy, (const int *)L.kyofpos, (const float *)L.lx, (const float *)L.ly, (const int2 *)L.ytab, L.Y);
}
static void base_inv(Lab &L, int flags = 7, bool nofft = false, bool with_x = true)
{
    KFFTPlan p = L.px; p.flags = flags; if (nofft) p.nstage = 0;
    KSpecLin a = L.add; if (!with_x) a.n = 0;
    hipLaunchKernelGGL(k_spec_rows_inv, dim3(L.ntiles, L.npair), dim3(L.thr_rows), L.lds_rows, L.st, p, 1 << L.lg_pl, L.rb, L.ntiles, L.F, (const kcf *)L.W, L.x, L.plane, (const kcf *)L.twx, a);
}

static double maxdiff(Lab &L, const double *a, const double *b)
{
    std::vector<double> ha((size_t)L.F * L.plane), hb((size_t)L.F * L.plane);
    CK(hipMemcpy(ha.data(), a, sizeof(double) * ha.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, sizeof(double) * hb.size(), hipMemcpyDeviceToHost));
    double m = 0.0, s = 0.0;
    for (size_t i = 0; i < ha.size(); i++) { m = std::max(m, fabs(ha[i] - hb[i])); s = std::max(s, fabs(hb[i])); }
    return m / s;
}

#include "spec_lab_variants.hip.h"

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 20;
    Lab L;
    lab_init(L, n);
    const double N = (double)n * n, MB = 1e-6;
    printf("n=%d rb=%d lds_rows=%zu lds_cols=%zu thr_rows=%d thr_cols=%d\n", n, L.rb, L.lds_rows, L.lds_cols, L.thr_rows, L.thr_cols);
    auto report = [&](const char *name, double us, double bytes) { printf("%-44s %8.1f us (min %6.1f)  %7.1f MB  %6.2f TB/s\n", name, us, L.last_min, bytes * MB, bytes / us * 1e-6); fflush(stdout); };
    // ---- baseline
    report("fwd<float> baseline", timeit(L, reps, [&] { base_fwd(L); }), 16.0 * N);
    report("fwd<float> no FFT stages", timeit(L, reps, [&] { base_fwd(L, 7, true); }), 16.0 * N);
    base_fwd(L); CK(hipStreamSynchronize(L.st));
    report("cols<1> baseline", timeit(L, reps, [&] { base_cols(L); }), 16.0 * N);
    report("cols<1> no FFT stages", timeit(L, reps, [&] { base_cols(L, 7, true); }), 16.0 * N);
    report("cols<1> unfused edges (flags 0)", timeit(L, reps, [&] { base_cols(L, 0); }), 16.0 * N);
    base_cols(L); CK(hipStreamSynchronize(L.st));
    report("inv baseline (x += z)", timeit(L, reps, [&] { base_inv(L); }), 40.0 * N);
    report("inv baseline (z only)", timeit(L, reps, [&] { base_inv(L, 7, false, false); }), 24.0 * N);
    report("inv no FFT stages (x += z)", timeit(L, reps, [&] { base_inv(L, 7, true); }), 40.0 * N);
    report("whole application", timeit(L, reps, [&] { base_fwd(L); base_cols(L); base_inv(L); }), 72.0 * N);
    // reference result of one application for the variants: x0 + M^-1 r
    CK(hipMemcpy(L.x, L.x0, sizeof(double) * L.F * L.plane, hipMemcpyDeviceToDevice));
    base_fwd(L); base_cols(L); base_inv(L);
    CK(hipStreamSynchronize(L.st));
    double *xref = nullptr;
    CK(hipMalloc((void **)&xref, sizeof(double) * L.F * L.plane));
    CK(hipMemcpy(xref, L.x, sizeof(double) * L.F * L.plane, hipMemcpyDeviceToDevice));
    lab_variants(L, reps, xref, report);
    return 0;
}
