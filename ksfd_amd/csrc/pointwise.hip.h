// Pointwise physics + BLAS-1 style kernels (gfx950, wave64).
// All vectors use the slab layout of params.h; kernels touch the owned interior only:
//   element (c, p), p in [0, nloc)  ->  base[c*plane + ng*inner + p]
#pragma once
#include <hip/hip_runtime.h>
#include "params.h"

#define KSFD_WAVE 64
#define KSFD_BLOCK 256
#define KSFD_MAXDOT 34

// ---------------------------------------------------------------------------------------------
// Derivatives.groom (KSFD/ksfdsym.py:888-900): x = max(x, lo), NaN -> lo.  One compare covers both.
__device__ __forceinline__ double ksfd_clamp(double x, double lo) { return !(x >= lo) ? lo : x; }

// fp64 log / exp for the free energy.  The RHS kernel is bound by these (two logs and a tanh per point): the library versions
// cost ~60-70 instructions each with their special-case handling; here the arguments are known to be positive, finite and far
// from the subnormal range (inputs are clamped to >= 1e-7 and NaN-scrubbed, alpha + sum w U > 0), so the classical reductions
// suffice.  Both are faithful to < 1 ulp (fdlibm's polynomials and error bounds: e_log.c / e_exp.c, Sun Microsystems 1993).
__device__ __forceinline__ double ksfd_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);                 // v_rcp_f64: ~1e-8 relative
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;                                           // < 1 ulp for the normal-range arguments used here
}
__device__ __forceinline__ double ksfd_div(double a, double x)
{
    const double r = ksfd_rcp(x);
    const double q = a * r;
    return fma(fma(-x, q, a), r, q);                    // one residual correction: correctly rounded but for rare ties
}
// log(x); the fast path needs 0 < x < inf (what the clamped inputs give), anything else goes to the library
__device__ __forceinline__ double ksfd_log(double x)
{
    if (!(x > 0.0) || x > 1.0e300) return log(x);      // never taken with groomed inputs (a wave-uniform not-taken branch)
    // x = 2^k * m, m in [sqrt(1/2), sqrt(2)); f = m - 1; s = f/(2+f); log(m) = f - hfsq + s*(hfsq + R(s^2))
    int k = __builtin_amdgcn_frexp_exp(x);              // x = mant * 2^k, mant in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s = ksfd_div(f, 2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
// exp(x) for |x| < 700
__device__ __forceinline__ double ksfd_exp(double x)
{
    const double kf = rint(x * 1.44269504088896338700e+00);
    const double hi = fma(-kf, 6.93147180369123816490e-01, x), lo = kf * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * fma(t, fma(t, fma(t, fma(t, 4.13813679705723846039e-08, -1.65339022054652515390e-06), 6.61375632143793436117e-05),
                                        -2.77777777770155933842e-03), 1.66666666666666019037e-01);
    const double y = 1.0 - ((lo - ksfd_div(r * c, 2.0 - c)) - hi);
    return ldexp(y, (int)kf);
}

// Free energy G(rho,U) = sum_g -beta_g log(alpha_g + sum_l w_gl U_gl) + Vcap(rho) + s2 log(rho)
// (KSFD/ksfdsym.py:983-990, ksfdligand.py:527-547, ksfdsoln.py:147-161) and, for the Jacobian
// action, its partials G_rho and G_Ul (closed forms: SURVEY.md section 0).
// NL is a compile-time ligand count so U[]/GU[] stay in registers.
// The cap needs tanh(y) + 1 and 1 - tanh(y)^2 only: with t = exp(-2|y|), tanh(|y|) + 1 = 2/(1+t), tanh(-|y|) + 1 = 2t/(1+t) and
// 1 - tanh^2 = 4t/(1+t)^2 -- one exp and one reciprocal, and no cancellation anywhere (rho far below rhomax gives t ~ 1e-8:
// "tanh + 1" formed from a rounded tanh would keep 8 digits).
template <int NL, bool DERIV>
__device__ __forceinline__ void ksfd_G(const KPhys &P, double rho, const double (&U)[NL], double &G,
                                       double &Grho, double (&GU)[NL])
{
    double g = 0.0;
    for (int q = 0; q < P.ngroups; q++) {        // wave-uniform loop, constants come from SGPRs
        double s = P.grp_alpha[q];
#pragma unroll
        for (int l = 0; l < NL; l++) s += (P.lig_group[l] == q) ? P.lig_w[l] * U[l] : 0.0;
        g -= P.grp_beta[q] * ksfd_log(s);
        if (DERIV) {
            double nb = -P.grp_beta[q] * ksfd_rcp(s);
#pragma unroll
            for (int l = 0; l < NL; l++) GU[l] = (P.lig_group[l] == q) ? nb * P.lig_w[l] : GU[l];
        }
    }
    const double y = (rho - P.rhomax) * P.inv_cushion;
    const double ay = fmin(fabs(y), 300.0);      // exp(-600) is 0 to every use below; keeps the exponent arithmetic in range
    double thp1, sech2;                          // tanh(y) + 1, 1 - tanh(y)^2
    if (ay > 19.5) {                             // exp(-39) < 2^-56: tanh(|y|) rounds to 1 (lanes diverge only at such aggregates)
        thp1 = y > 0.0 ? 2.0 : 2.0 * ksfd_exp(-2.0 * ay);
        sech2 = 0.0;
        if (DERIV && !(y > 0.0)) sech2 = 2.0 * thp1;
    } else {
        const double t = ksfd_exp(-2.0 * ay);
        const double r = ksfd_rcp(1.0 + t);
        thp1 = y > 0.0 ? 2.0 * r : 2.0 * t * r;
        sech2 = 4.0 * t * r * r;
    }
    double cap, dcap;
    if (P.cap_kind == 1) {                        // witch
        double r = rho * P.inv_rhomax;
        cap = P.ms * thp1 * r;
        dcap = P.ms * (sech2 * r * P.inv_cushion + thp1 * P.inv_rhomax);
    } else {                                      // tophat
        cap = P.ms * thp1;
        dcap = P.ms * sech2 * P.inv_cushion;
    }
    G = g + cap + P.s2 * ksfd_log(rho);
    if (DERIV) Grho = P.s2 * ksfd_rcp(rho) + dcap;
}

// ---------------------------------------------------------------------------------------------
// reductions: wave64 shuffle tree, then one LDS hop across the 4 waves of a 256-thread block.
__device__ __forceinline__ double ksfd_wave_sum(double v)
{
#pragma unroll
    for (int o = KSFD_WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, KSFD_WAVE);
    return v;
}
__device__ __forceinline__ double ksfd_wave_max(double v)
{
#pragma unroll
    for (int o = KSFD_WAVE / 2; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, KSFD_WAVE));
    return v;
}

// Vector addressing helper: one blockIdx.y per field plane.
struct KVec {
    long long plane, off, nloc;   // off = ng*inner
    int nf;                       // number of field planes (F)
};

// All BLAS-1 kernels are templated on VW = doubles per lane per access (2 when the slab geometry is
// even, so every access is a 16-byte double2: 1 KiB per wave instruction; else 1).
template <int VW> struct KPack;
template <> struct KPack<1> { typedef double T; };
template <> struct KPack<2> { typedef double2 T; };
__device__ __forceinline__ double kget(const double &v, int) { return v; }
__device__ __forceinline__ double kget(const double2 &v, int e) { return e ? v.y : v.x; }
__device__ __forceinline__ void kset(double &v, int, double x) { v = x; }
__device__ __forceinline__ void kset(double2 &v, int e, double x) { if (e) v.y = x; else v.x = x; }
template <int VW> __device__ __forceinline__ typename KPack<VW>::T kload(const double *p)
{
    return *reinterpret_cast<const typename KPack<VW>::T *>(p);
}
template <int VW> __device__ __forceinline__ void kstore(double *p, const typename KPack<VW>::T &v)
{
    *reinterpret_cast<typename KPack<VW>::T *>(p) = v;
}

// out = sum_t a[t] * x[t]   (NT <= 6 inputs; out may alias any x[t]).  Used for the ROSW stage
// vectors (VecMAXPY/VecWAXPY in PETSc's TSStep_RosW) and the GMRES solution update.
// sum of v over the block -> *dst (wave shuffle tree, one LDS hop across the waves); every thread of the block must call it
__device__ __forceinline__ void ksfd_block_sum_to(double v, double *dst)
{
    __shared__ double red_[KSFD_BLOCK / KSFD_WAVE];
    v = ksfd_wave_sum(v);
    if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) red_[threadIdx.x / KSFD_WAVE] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) t += red_[q];
        *dst = t;
    }
}

struct KLin {
    const double *x[6];
    double a[6];
};
template <int NT, int VW>
__global__ void __launch_bounds__(KSFD_BLOCK) k_lincomb(KVec g, KLin L, double *out, double *part = nullptr)
{
    // part != NULL: also the block's share of ||out||^2 -> part[blockIdx.y*gridDim.x + blockIdx.x] (k_reduce_rows finishes it)
    const long long base = (long long)blockIdx.y * g.plane + g.off;
    const long long stride = (long long)gridDim.x * blockDim.x * VW;
    double nrm = 0.0;
    for (long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VW; p < g.nloc; p += stride) {
        typename KPack<VW>::T s, xv;
#pragma unroll
        for (int e = 0; e < VW; e++) kset(s, e, 0.0);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            xv = kload<VW>(L.x[t] + base + p);
#pragma unroll
            for (int e = 0; e < VW; e++) kset(s, e, kget(s, e) + L.a[t] * kget(xv, e));
        }
        kstore<VW>(out + base + p, s);
#pragma unroll
        for (int e = 0; e < VW; e++) nrm += kget(s, e) * kget(s, e);
    }
    if (part) ksfd_block_sum_to(nrm, part + (long long)blockIdx.y * gridDim.x + blockIdx.x);
}

// Krylov dot products: d[i] = <w, V_i> for i < k, and d[k] = <w, w>, in ONE pass over w.
// V_i = V + i*vstride.  Block partials go to part[i*nblk + b]; k_reduce_rows finishes them
// in a fixed order (bitwise reproducible; no float atomics).
template <int KMAX, int VW>
__global__ void __launch_bounds__(KSFD_BLOCK) k_multidot(KVec g, const double *__restrict__ w,
                                                         const double *__restrict__ V, long long vstride,
                                                         int k, double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE][KMAX + 1];
    double acc[KMAX + 1];
#pragma unroll
    for (int i = 0; i <= KMAX; i++) acc[i] = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x * VW;
    for (int c = 0; c < g.nf; c++) {
        const long long base = (long long)c * g.plane + g.off;
        for (long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VW; p < g.nloc; p += stride) {
            const typename KPack<VW>::T wv = kload<VW>(w + base + p);
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k) {
                    const typename KPack<VW>::T vv = kload<VW>(V + (long long)i * vstride + base + p);
#pragma unroll
                    for (int e = 0; e < VW; e++) acc[i] += kget(wv, e) * kget(vv, e);
                }
#pragma unroll
            for (int e = 0; e < VW; e++) acc[KMAX] += kget(wv, e) * kget(wv, e);
        }
    }
    const int lane = threadIdx.x & (KSFD_WAVE - 1), wv_ = threadIdx.x / KSFD_WAVE;
#pragma unroll
    for (int i = 0; i <= KMAX; i++) {
        if (i < k || i == KMAX) {
            double s = ksfd_wave_sum(acc[i]);
            if (lane == 0) red[wv_][i] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x <= KMAX) {
        int i = threadIdx.x;
        if (i < k || i == KMAX) {
            double s = 0.0;
            for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) s += red[q][i];
            int row = (i == KMAX) ? k : i;
            part[(long long)row * gridDim.x + blockIdx.x] = s;
        }
    }
}

// One-pass Gram-Schmidt data for GMRES with a lagged Gram row (see gmres() in ksfd_hip.hip):
//   rows 0..k-1   : d[i] = <w, V_i>
//   rows k..2k-1  : g[i] = <V_{k-1}, V_i>      (Gram row of the newest basis vector)
//   row  2k       : <w, w>
// Reads w and V_0..V_{k-1} exactly once.
template <int KMAX, int VW>
__global__ void __launch_bounds__(KSFD_BLOCK) k_multidot_gram(KVec g, const double *__restrict__ w,
                                                              const double *__restrict__ V, long long vstride,
                                                              int k, double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE][2 * KMAX + 1];
    double acc[2 * KMAX + 1];
#pragma unroll
    for (int i = 0; i <= 2 * KMAX; i++) acc[i] = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x * VW;
    for (int c = 0; c < g.nf; c++) {
        const long long base = (long long)c * g.plane + g.off;
        for (long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VW; p < g.nloc; p += stride) {
            const typename KPack<VW>::T wv = kload<VW>(w + base + p);
            const typename KPack<VW>::T lv = kload<VW>(V + (long long)(k - 1) * vstride + base + p);
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k) {
                    const typename KPack<VW>::T vv = (i == k - 1) ? lv : kload<VW>(V + (long long)i * vstride + base + p);
#pragma unroll
                    for (int e = 0; e < VW; e++) {
                        acc[i] += kget(wv, e) * kget(vv, e);
                        acc[KMAX + i] += kget(lv, e) * kget(vv, e);
                    }
                }
#pragma unroll
            for (int e = 0; e < VW; e++) acc[2 * KMAX] += kget(wv, e) * kget(wv, e);
        }
    }
    const int lane = threadIdx.x & (KSFD_WAVE - 1), wv_ = threadIdx.x / KSFD_WAVE;
#pragma unroll
    for (int i = 0; i <= 2 * KMAX; i++) {
        const bool live = (i < k) || (i >= KMAX && i < KMAX + k) || (i == 2 * KMAX);
        if (live) {
            double s = ksfd_wave_sum(acc[i]);
            if (lane == 0) red[wv_][i] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x <= 2 * KMAX) {
        const int i = threadIdx.x;
        const bool live = (i < k) || (i >= KMAX && i < KMAX + k) || (i == 2 * KMAX);
        if (live) {
            double s = 0.0;
            for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) s += red[q][i];
            const int row = (i == 2 * KMAX) ? 2 * k : (i >= KMAX ? k + (i - KMAX) : i);
            part[(long long)row * gridDim.x + blockIdx.x] = s;
        }
    }
}

// out[r] = reduce_b part[r*nblk + b]  (op 0 sum, 1 max); one block per row.
// pub != NULL: the results are also stored straight into host memory (mapped, coherent) and the block that finishes last
// raises *flag to seq, so the host can spin on the flag instead of paying a stream synchronisation + D2H copy
// (~20-30 us per reduction, 30 reductions per step: all of it GPU idle time).
__global__ void __launch_bounds__(KSFD_BLOCK) k_reduce_rows(const double *__restrict__ part, int nblk, int op,
                                                            double *__restrict__ out, double *pub = nullptr,
                                                            unsigned int *count = nullptr, unsigned long long *flag = nullptr,
                                                            unsigned long long seq = 0)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE];
    const double *row = part + (long long)blockIdx.x * nblk;
    double s = op ? -1.0e300 : 0.0;
    // eight loads in flight per thread (one memory latency per 8 x blockDim partials instead of one per blockDim: this kernel sits
    // between every residual evaluation and the host's decision); the order of the additions is fixed by the code, so results stay
    // bitwise reproducible from run to run
    for (int b0 = threadIdx.x; b0 < nblk; b0 += 8 * blockDim.x) {
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int b = b0 + q * blockDim.x; t[q] = b < nblk ? row[b] : (op ? -1.0e300 : 0.0); }
#pragma unroll
        for (int q = 0; q < 8; q++) s = op ? fmax(s, t[q]) : s + t[q];
    }
    s = op ? ksfd_wave_max(s) : ksfd_wave_sum(s);
    if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) red[threadIdx.x / KSFD_WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0];
        for (int q = 1; q < KSFD_BLOCK / KSFD_WAVE; q++) t = op ? fmax(t, red[q]) : t + red[q];
        out[blockIdx.x] = t;
        if (pub) {
            __hip_atomic_store(pub + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            if (atomicAdd(count, 1u) == gridDim.x - 1) {              // every row's store is ordered before its increment
                *count = 0;
                __threadfence_system();
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// results already reduced across ranks on the device (RCCL): hand them to the spinning host the same way
__global__ void k_publish(const double *__restrict__ res, int n, double *pub, unsigned long long *flag, unsigned long long seq)
{
    if (threadIdx.x < n) __hip_atomic_store(pub + threadIdx.x, res[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Gram-Schmidt update fused with the normalisation:  w = (w - sum_{i<k} h[i] V_i) * scale
struct KCoef {
    double h[KSFD_MAXDOT];
};
template <int KMAX, int VW>
__global__ void __launch_bounds__(KSFD_BLOCK) k_gs_update(KVec g, double *__restrict__ w,
                                                          const double *__restrict__ V, long long vstride, int k,
                                                          KCoef C, double scale)
{
    const long long base = (long long)blockIdx.y * g.plane + g.off;
    const long long stride = (long long)gridDim.x * blockDim.x * VW;
    for (long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VW; p < g.nloc; p += stride) {
        typename KPack<VW>::T s = kload<VW>(w + base + p);
#pragma unroll
        for (int i = 0; i < KMAX; i++)
            if (i < k) {
                const typename KPack<VW>::T vv = kload<VW>(V + (long long)i * vstride + base + p);
#pragma unroll
                for (int e = 0; e < VW; e++) kset(s, e, kget(s, e) - C.h[i] * kget(vv, e));
            }
#pragma unroll
        for (int e = 0; e < VW; e++) kset(s, e, kget(s, e) * scale);
        kstore<VW>(w + base + p, s);
    }
}

// x = beta*x + sum_{i<k} y[i] V_i   (GMRES solution update)
template <int KMAX, int VW>
__global__ void __launch_bounds__(KSFD_BLOCK) k_basis_axpy(KVec g, double *__restrict__ x,
                                                           const double *__restrict__ V, long long vstride, int k,
                                                           KCoef C, double beta, double *part = nullptr)
{
    const long long base = (long long)blockIdx.y * g.plane + g.off;
    const long long stride = (long long)gridDim.x * blockDim.x * VW;
    double nrm = 0.0;
    for (long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VW; p < g.nloc; p += stride) {
        typename KPack<VW>::T s;
        if (beta == 0.0) {
#pragma unroll
            for (int e = 0; e < VW; e++) kset(s, e, 0.0);
        } else {
            s = kload<VW>(x + base + p);
#pragma unroll
            for (int e = 0; e < VW; e++) kset(s, e, beta * kget(s, e));
        }
#pragma unroll
        for (int i = 0; i < KMAX; i++)
            if (i < k && C.h[i] != 0.0) {                 // wave-uniform; a zero coefficient costs no load (slots of unused recycled spaces)
                const typename KPack<VW>::T vv = kload<VW>(V + (long long)i * vstride + base + p);
#pragma unroll
                for (int e = 0; e < VW; e++) kset(s, e, kget(s, e) + C.h[i] * kget(vv, e));
            }
        kstore<VW>(x + base + p, s);
#pragma unroll
        for (int e = 0; e < VW; e++) nrm += kget(s, e) * kget(s, e);
    }
    if (part) ksfd_block_sum_to(nrm, part + (long long)blockIdx.y * gridDim.x + blockIdx.x);     // ||x||^2 share, see k_lincomb
}

// Step completion (PETSc TSEvaluateStep_RosW + TSErrorWeightedNorm, restated):
//   unew = u + sum_j bt[j] Y_j ;  err = sum_j (b2t[j]-bt[j]) Y_j ;
//   partial sum of (err / (atol + rtol*max(|unew|, |unew+err|)))^2.
// Writes unew over u (the caller keeps a rollback copy) and, if errout != NULL, err.
__global__ void __launch_bounds__(KSFD_BLOCK) k_rosw_finish(KVec g, double *__restrict__ u,
                                                            const double *__restrict__ Y, long long ystride,
                                                            double bt0, double bt1, double bt2, double bt3,
                                                            double e0, double e1, double e2, double e3,
                                                            double atol, double rtol, double *__restrict__ errout,
                                                            double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (int c = 0; c < g.nf; c++) {
        const long long base = (long long)c * g.plane + g.off;
        for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < g.nloc; p += stride) {
            const double y0 = Y[base + p], y1 = Y[ystride + base + p], y2 = Y[2 * ystride + base + p],
                         y3 = Y[3 * ystride + base + p];
            const double un = u[base + p] + (bt0 * y0 + bt1 * y1 + bt2 * y2 + bt3 * y3);
            const double er = e0 * y0 + e1 * y1 + e2 * y2 + e3 * y3;
            const double tol = atol + rtol * fmax(fabs(un), fabs(un + er));
            const double q = er / tol;
            acc += q * q;
            u[base + p] = un;
            if (errout) errout[base + p] = er;
        }
    }
    acc = ksfd_wave_sum(acc);
    if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) red[threadIdx.x / KSFD_WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) t += red[q];
        part[blockIdx.x] = t;
    }
}

// KSFDTS.groom on the stored state (KSFD/ksfdts.py:231-237)
// copy != NULL: the groomed owned points are stored there as well (the step's roll-back copy: no pass of its own)
__global__ void __launch_bounds__(KSFD_BLOCK) k_groom(KVec g, double *__restrict__ u, double rhomin, double Umin, double *__restrict__ copy = nullptr)
{
    const long long base = (long long)blockIdx.y * g.plane + g.off;
    const double lo = blockIdx.y == 0 ? rhomin : Umin;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < g.nloc; p += stride) {
        const double v = ksfd_clamp(u[base + p], lo);
        u[base + p] = v;
        if (copy) copy[base + p] = v;
    }
}

// sum of plane 0 (count_worms, KSFD/ksfdts.py:239-246)
__global__ void __launch_bounds__(KSFD_BLOCK) k_sum_rho(KVec g, const double *__restrict__ u, double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE];
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < g.nloc; p += stride) acc += u[g.off + p];
    acc = ksfd_wave_sum(acc);
    if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) red[threadIdx.x / KSFD_WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) t += red[q];
        part[blockIdx.x] = t;
    }
}

// rho *= factor (conserve_worms) or rho[p] *= f[p] (add_variance)
__global__ void __launch_bounds__(KSFD_BLOCK) k_mul_rho(KVec g, double *__restrict__ u, const double *__restrict__ f,
                                                        double scalar)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < g.nloc; p += stride)
        u[g.off + p] *= f ? f[p] : scalar;
}

// ---------------------------------------------------------------------------------------------
// Host-layout <-> device-layout permutations (run on the device, staged through a flat buffer).
// layout 0 PETSc: c + F*p ; layout 1 SoA: c*nloc + p ; layout 2 HDF5: C order over (c, x, y[, z]).
__device__ __forceinline__ long long ksfd_host_index(const KGeom &G, int layout, int c, long long p)
{
    if (layout == 0) return c + (long long)G.F * p;
    if (layout == 1) return (long long)c * G.nloc + p;
    long long i = p % G.nx, r = p / G.nx;
    if (G.dim == 1) return (long long)c * G.nx + i;
    long long j = r % G.ny, k = r / G.ny;
    if (G.dim == 2) return ((long long)c * G.nx + i) * G.ny + j;
    return (((long long)c * G.nx + i) * G.ny + j) * G.nz + k;
}
__global__ void __launch_bounds__(KSFD_BLOCK) k_from_host_layout(KGeom G, int layout, const double *__restrict__ flat,
                                                                 double *__restrict__ dev, long long devplane,
                                                                 long long devoff)
{
    const int c = blockIdx.y;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride)
        dev[(long long)c * devplane + devoff + p] = flat[ksfd_host_index(G, layout, c, p)];
}
__global__ void __launch_bounds__(KSFD_BLOCK) k_to_host_layout(KGeom G, int layout, const double *__restrict__ dev,
                                                               long long devplane, long long devoff,
                                                               double *__restrict__ flat)
{
    const int c = blockIdx.y;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride)
        flat[ksfd_host_index(G, layout, c, p)] = dev[(long long)c * devplane + devoff + p];
}


// ---------------------------------------------------------------------------------------------
// Device-resident small algebra of one GMRES iteration (pipelined solver, gmres_async in ksfd_hip.hip):
// from the reduced dots of iteration j (k = j+1 basis vectors):  d = V^T w, Gram row g = V^T v_j, ww = |w|^2
//   c = d + (I - G) d ,  hn^2 = ww - 2 c.d + c.G c     (CGS2 with the second projection done algebraically)
//   Hessenberg column [c; hn] -> previous Givens rotations -> new rotation -> residual estimate |g_{j+1}|
// Outputs: coef[0..k) and scale = 1/hn for the fused update kernel, mon[2j] = residual estimate, mon[2j+1] = hn.
// One thread: k <= 32, i.e. < 3k FMAs.
__global__ void k_gmres_coef(int j, int m, double beta, const double *__restrict__ dres, double *__restrict__ Gm,
                             double *__restrict__ H, double *__restrict__ cs, double *__restrict__ sn,
                             double *__restrict__ g, double *__restrict__ coef, double *__restrict__ scale,
                             double *__restrict__ mon)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int k = j + 1, ld = m + 1;
    if (j == 0) { for (int i = 0; i <= m; i++) g[i] = 0.0; g[0] = beta; }
    const double *d = dres, *gr = dres + k;
    const double ww = dres[2 * k];
    for (int i = 0; i < k; i++) { Gm[i * ld + j] = gr[i]; Gm[j * ld + i] = gr[i]; }
    double c[KSFD_MAXDOT];
    double cd = 0.0, cGc = 0.0;
    for (int i = 0; i < k; i++) {
        double s = 0.0;
        for (int l = 0; l < k; l++) s += ((i == l ? 1.0 : 0.0) - Gm[i * ld + l]) * d[l];
        c[i] = d[i] + s;
    }
    for (int i = 0; i < k; i++) {
        cd += c[i] * d[i];
        double s = 0.0;
        for (int l = 0; l < k; l++) s += Gm[i * ld + l] * c[l];
        cGc += c[i] * s;
    }
    double hn2 = ww - 2.0 * cd + cGc;
    if (!(hn2 > 0.0)) hn2 = (hn2 != hn2) ? hn2 : 0.0;       // keep NaN visible, clamp negatives
    const double hn = sqrt(hn2);
    for (int i = 0; i < k; i++) coef[i] = c[i];
    *scale = hn > 0.0 ? 1.0 / hn : 0.0;
    double *Hc = H + (size_t)ld * j;
    for (int i = 0; i < k; i++) Hc[i] = c[i];
    Hc[k] = hn;
    for (int i = 0; i < j; i++) {
        const double t = cs[i] * Hc[i] + sn[i] * Hc[i + 1];
        Hc[i + 1] = -sn[i] * Hc[i] + cs[i] * Hc[i + 1];
        Hc[i] = t;
    }
    const double den = hypot(Hc[j], Hc[j + 1]);
    cs[j] = den > 0.0 ? Hc[j] / den : 1.0;
    sn[j] = den > 0.0 ? Hc[j + 1] / den : 0.0;
    Hc[j] = den;
    Hc[j + 1] = 0.0;
    g[j + 1] = -sn[j] * g[j];
    g[j] = cs[j] * g[j];
    mon[2 * j] = fabs(g[j + 1]);
    mon[2 * j + 1] = hn;
}

// w = (w - sum_{i<k} coef[i] V_i) * (*scale), coefficients read from device memory
template <int KMAX, int VW>
__global__ void __launch_bounds__(KSFD_BLOCK) k_gs_update_dev(KVec g, double *__restrict__ w,
                                                              const double *__restrict__ V, long long vstride, int k,
                                                              const double *__restrict__ coef,
                                                              const double *__restrict__ scale)
{
    const long long base = (long long)blockIdx.y * g.plane + g.off;
    const long long stride = (long long)gridDim.x * blockDim.x * VW;
    double cf[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; i++) cf[i] = i < k ? coef[i] : 0.0;
    const double sc = *scale;
    for (long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VW; p < g.nloc; p += stride) {
        typename KPack<VW>::T s = kload<VW>(w + base + p);
#pragma unroll
        for (int i = 0; i < KMAX; i++)
            if (i < k) {
                const typename KPack<VW>::T vv = kload<VW>(V + (long long)i * vstride + base + p);
#pragma unroll
                for (int e = 0; e < VW; e++) kset(s, e, kget(s, e) - cf[i] * kget(vv, e));
            }
#pragma unroll
        for (int e = 0; e < VW; e++) kset(s, e, kget(s, e) * sc);
        kstore<VW>(w + base + p, s);
    }
}
