// Variants of the spectral kernels under test (included by tools/spec_lab.hip after the Lab helpers).
#pragma once

// ---------------------------------------------------------------------------------------------------------------------
// V1: persistent column kernel.  One block walks several {kx, -kx} pairs; the 16 first-stage inputs of the NEXT pair are
// requested before the last inverse stage of the current one, so their latency (a gather of 32-B pieces out of the tile-major
// array) overlaps that stage, its stores and the barrier.  One rank, tile-major input, whole columns, npair = 1.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lab_pair_of(int b, int k, int G, int total)
{
    // blocks b and b + 8 share an XCD: give each XCD a contiguous run of pairs per round (neighbouring positions share 128-B lines)
    const int idx = k * G + ((G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3));
    return idx < total ? idx : -1;
}

template <int LGS>
__device__ __forceinline__ void lab_cols_body(const KFFTPlan &PY, int nxl, kcf *__restrict__ W, const kcf *__restrict__ Wt, int lg_rb, const kcf *__restrict__ tw,
                                              const int4 *__restrict__ pairtab, const int *__restrict__ posy, const int *__restrict__ kyofpos,
                                              const float *__restrict__ lx, const float *__restrict__ ly, const KSpecSym &S, int npairs, kcf *lds)
{
    constexpr int S0 = 1 << LGS;                    // ny / 16
    const int ny = PY.n;
    const int sstride = ny + (ny >> 4) + 1;
    const int G = gridDim.x;
    const int s = threadIdx.x >> LGS, i = threadIdx.x & (S0 - 1);        // one first/last-stage butterfly per thread (blockDim = 2 * S0)
    const long long qs = (long long)S0 * nxl;
    auto src = [&](int4 pt) {
        return Wt + ((((long long)(i >> lg_rb)) * nxl + (s ? pt.y : pt.x)) << lg_rb) + (i & ((1 << lg_rb) - 1));
    };
    int cur = lab_pair_of(blockIdx.x, 0, G, npairs);
    if (cur < 0) return;
    int4 pt = pairtab[cur];
    kcf x[16];
    {
        const kcf *p0 = src(pt);
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = p0[q * qs];
    }
    for (int k = 0;; k++) {
        const bool self = pt.w != 0;
        const int kxA = pt.z, kxB = self ? pt.w - 1 : pt.z;
        // ---- forward stage 0 from the registers
        kc_dft<16, false>(x);
        {
            const kcf w1 = tw[i];
            kcf w = w1;
#pragma unroll
            for (int q = 1; q < 16; q++) { x[q] = kc_mul(x[q], w); w = kc_mul(w, w1); }
        }
        {
            kcf *b0 = lds + (long long)s * sstride + kspec_pad(i);
#pragma unroll
            for (int q = 0; q < 16; q++) b0[q * S0 + ((q * S0) >> 4)] = x[q];
        }
        __syncthreads();
        kspec_fft_fwd(PY, lds, sstride, 2, tw, 1);
        kspec_cols_symbol<1>(PY, lds, sstride, self, kxA, kxB, posy, kyofpos, lx, ly, S);
        __syncthreads();
        kspec_fft_inv(PY, lds, sstride, 2, tw, 1);
        // ---- next pair's inputs on their way while the last inverse stage runs
        const int nxt = lab_pair_of(blockIdx.x, k + 1, G, npairs);
        int4 ptn = pt;
        kcf nx[16];
        if (nxt >= 0) {
            ptn = pairtab[nxt];
            const kcf *p0 = src(ptn);
#pragma unroll
            for (int q = 0; q < 16; q++) nx[q] = p0[q * qs];
        }
        // ---- last inverse stage straight to memory
        {
            const kcf *b0 = lds + (long long)s * sstride + kspec_pad(i);
            kcf c[16];
#pragma unroll
            for (int q = 0; q < 16; q++) c[q] = b0[q * S0 + ((q * S0) >> 4)];
            const kcf w1 = kc_conj(tw[i]);
            kcf w = w1;
#pragma unroll
            for (int q = 1; q < 16; q++) { c[q] = kc_mul(c[q], w); w = kc_mul(w, w1); }
            kc_dft<16, true>(c);
            kcf *dst = W + ((long long)(s ? pt.y : pt.x)) * ny + i;
#pragma unroll
            for (int q = 0; q < 16; q++) dst[q * S0] = c[q];
        }
        if (nxt < 0) break;
        __syncthreads();                        // the LDS is free for the next pair
        pt = ptn;
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = nx[q];
    }
}

__global__ void __launch_bounds__(512) k_lab_cols_pp(KFFTPlan PY, int nxl, kcf *__restrict__ W, const kcf *__restrict__ Wt, int lg_rb, const kcf *__restrict__ tw,
                                                     const int4 *__restrict__ pairtab, const int *__restrict__ posy, const int *__restrict__ kyofpos,
                                                     const float *__restrict__ lx, const float *__restrict__ ly, KSpecSym S, int npairs)
{
    extern __shared__ kcf kspec_lds[];
    KSPEC_LGS_DISPATCH(PY.lg - 4, (lab_cols_body<LGS>(PY, nxl, W, Wt, lg_rb, tw, pairtab, posy, kyofpos, lx, ly, S, npairs, kspec_lds)));
}

template <typename REP>
static void lab_variants(Lab &L, int reps, const double *xref, REP report)
{
    const double N = (double)L.n * L.n;
    CK(hipFuncSetAttribute((const void *)k_lab_cols_pp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_cols));
    KFFTPlan py = L.py; py.flags = 7;
    for (int G : { 512, 256, 1024, 2048 }) {
        if (G > L.nblk_cols) continue;
        auto run = [&] {
            hipLaunchKernelGGL(k_lab_cols_pp, dim3(G), dim3(2 * (L.n >> 4)), L.lds_cols, L.st, py, L.n, L.W, (const kcf *)L.W2, L.lg_rb, (const kcf *)L.twy,
                               (const int4 *)L.pairtab, (const int *)L.posy, (const int *)L.kyofpos, (const float *)L.lx, (const float *)L.ly, L.Y, L.nblk_cols);
        };
        base_fwd(L); CK(hipStreamSynchronize(L.st));
        char nm[96]; snprintf(nm, sizeof nm, "V1 cols persistent+prefetch, grid %d", G);
        report(nm, timeit(L, reps, run), 16.0 * N);
        // correctness against the baseline result
        CK(hipMemcpy(L.x, L.x0, sizeof(double) * L.F * L.plane, hipMemcpyDeviceToDevice));
        base_fwd(L); run(); base_inv(L);
        CK(hipStreamSynchronize(L.st));
        printf("    max rel diff vs baseline %.3e\n", maxdiff(L, L.x, xref));
    }
}
