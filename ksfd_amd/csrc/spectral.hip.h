// libksfd_hip.so -- spectral preconditioner: the constant-coefficient part of shift*I - J inverted exactly in Fourier space.
//
// Why: on the periodic box the frozen Jacobian is  J = J0 + (variable-coefficient remainder), with
//   (J0 v)_rho = a_rr Lap(v_rho) + sum_l a_rU_l Lap(v_Ul) ,   (J0 v)_Ul = s_l v_rho - gamma_l v_Ul + D_l Lap(v_Ul)
// a_rr = mean(rho G_rho), a_rU_l = mean(rho G_Ul) over the grid (G_rho, G_U: the frozen planes of k_jcoef) and Lap the SAME
// 4th-order star the kernels apply (symbol L2(k) = sum_a (-30 + 32 cos th_a - 2 cos 2 th_a)/(12 h_a^2)).  While the state is a
// smooth perturbation of a uniform one -- the benchmark's random-perturbation start values, and the first phase of every
// production run -- the remainder is a few per cent of J0, so M = shift*I - J0 is an almost exact preconditioner at ANY
// stiffness: 3-4 GMRES iterations per stage system at h = 0.01 ... 100 (tests/experiments/fft_pc_experiment.py), where the
// Chebyshev polynomial needs ~13 and the V cycle ~5 iterations of ~20 fine-level passes each.  Once aggregates have formed
// (rho varying by factors) it stops converging and ksfd_step hands back to the multigrid V cycle (iteration-count trigger).
//
// M is block-diagonal in Fourier space with an "arrow" F x F block per wavenumber, inverted in closed form:
//   d_l = shift + gamma_l - D_l L2 ,  den = shift - a_rr L2 - sum_l a_rU_l L2 s_l / d_l
//   z_rho = (v_rho + sum_l (a_rU_l L2 / d_l) v_Ul) / den ,  z_Ul = (v_Ul + s_l z_rho) / d_l
// All multipliers are real and even in k, so M^-1 maps real fields to real fields and two real fields travel as ONE complex
// field c = v_a + i v_b through plain complex FFTs (no real-FFT packing); the symbol stage recovers v_a^(k), v_b^(k) from
// c^(k) and conj c^(-k).
//
// Hand-written radix-16/8/4/2 FFTs in LDS, fp32 storage and arithmetic (it is a preconditioner; flexible GMRES keeps
// z_j = M^-1 v_j, so the Arnoldi relation, the true residual and the solution stay fp64).  Three launches per application:
//   k_spec_rows_fwd : RB rows of v (fp64, F planes) -> complex fp32, DIF FFT along x in LDS, written TRANSPOSED
//                     W[pair][pos_x][y] (RB consecutive y per store segment)
//   k_spec_cols     : one block per {kx, -kx} column pair: DIF FFT along y (contiguous), symbol, DIT inverse FFT along y
//   k_spec_rows_inv : transposed read, DIT inverse FFT along x, z written as fp64 planes
// DIF forward leaves the spectrum in digit-reversed order and DIT inverse consumes exactly that order, so no permutation
// pass exists anywhere; the symbol stage looks positions up in small tables (posx/posy).
// HBM traffic per application: 8FN + 4FN | 4FN + 4FN | 4FN + 8FN = 32 F N bytes (algorithmic: read v, write z = 16 F N).
// 2-D (these kernels) and 3-D (k_spec3_*, below); one rank or 2, 4, 8 slab ranks (all-to-all transposes between the x- and the
// y/z-transforms: spectral_host.hip.h).
#pragma once

typedef float2 kcf;
#define KSPEC_MAXSTAGE 7
struct KFFTPlan {
    int n, lg, nstage;             // n = m << lg
    int m;                         // 1, or 3: extents 3 * 2^lg (the reference's own 2-D grids are 384^2 and 1536^2, options81:17, options84:15).
                                   // A radix-3 stage over the element stride 2^lg comes FIRST (DIF; last in the DIT inverse) and leaves three
                                   // independent power-of-two transforms, which the stages below run as 3 * nseq sequences: in the LDS a
                                   // sequence is m sub-sequences of 2^lg elements, kspec_ss() apart (kspec_lpos)
    int radix[KSPEC_MAXSTAGE];     // DIF stage order of the power-of-two part; prod = 2^lg
    int flags;                     // kernel variants (spec_apply): bit0/bit1 = first/last stage of k_spec_cols fused with its loads/stores, bit2 = first stage of k_spec_rows_fwd<float>,
                                   // bit3 = one rank: the column kernel stores its result TILE-MAJOR for the inverse row kernel (Wi[tile][pos][r], the layout
                                   // it reads its own input in), so that kernel reads one contiguous run per tile instead of gathering 32-B pieces
    int lgw;                       // layout of the forward work array on one rank (kspec_wt_index): < 0 tile-major, else log2 of the position group
};

// Element (tile, pos, r) of one field pair's forward work array (one rank: what the forward row kernel hands to the column kernel).
//   lgw < 0 : tile-major            Wt[tile][pos][r]                       -- the row kernel's store is one contiguous run per tile
//   lgw >= 0: position-group-major  Wg[pos >> lgw][tile][pos & (2^lgw - 1)][r]  with 2^lgw * rb * 8 B = one 128-B line per (group, tile):
//             the row kernel stores whole lines 128 B apart, and the 2^lgw column blocks of a group -- neighbours in the launch, on
//             one XCD -- read ONE contiguous region of ny * 2^lgw elements instead of lines strewn over the whole array at tile stride
//             (measured 4096^2: column kernel 116 -> 103 us, row kernel unchanged)
__device__ __forceinline__ long long kspec_wt_index(int lgw, int lg_rb, long long ntiles, int nxl, long long tile, int pos, int r)
{
    return lgw < 0 ? ((tile * nxl + pos) << lg_rb) + r
                   : ((((long long)(pos >> lgw) * ntiles + tile) << (lgw + lg_rb)) + ((pos & ((1 << lgw) - 1)) << lg_rb) + r);
}

// LDS geometry of one sequence of P.n elements: m sub-sequences of 2^lg elements, each padded by one element per 16 (+1)
__device__ __forceinline__ int kspec_ss(const KFFTPlan &P) { const int n2 = 1 << P.lg; return n2 + (n2 >> 4) + 1; }
__device__ __forceinline__ int kspec_sstride(const KFFTPlan &P) { return P.m * kspec_ss(P); }
__device__ __forceinline__ int kspec_lpos(const KFFTPlan &P, int e)      // padded position of element e
{
    if (P.m == 1) return e + (e >> 4);
    const int b = e >> P.lg, i = e & ((1 << P.lg) - 1);
    return b * kspec_ss(P) + i + (i >> 4);
}

// a few fp64 vectors with coefficients: added to the input of the forward row kernel / to the output of the inverse one
struct KSpecLin {
    int n;
    const double *p[3];
    double a[3];
};

struct KSpecSym {
    int nlig;
    float shift, a_rr, scale, den_floor;
    float a_rU[KSFD_MAXL], s[KSFD_MAXL], gam[KSFD_MAXL], D[KSFD_MAXL];
};

__device__ __forceinline__ int kspec_pad(int i) { return i + (i >> 4); }      // one element of padding per 16: conflict-free butterflies
__device__ __forceinline__ kcf kc_mul(kcf a, kcf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ kcf kc_add(kcf a, kcf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ kcf kc_sub(kcf a, kcf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ kcf kc_conj(kcf a) { return make_float2(a.x, -a.y); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ kcf kc_rot(kcf a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

// exp(-+ 2 pi i k / 16), k compile-time after unrolling
template <bool INV> __device__ __forceinline__ kcf kc_w16(int k)
{
    constexpr float c[16] = { 1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f, -0.38268343236508977f,
                              -0.70710678118654752f, -0.92387953251128674f, -1.0f, -0.92387953251128674f, -0.70710678118654752f,
                              -0.38268343236508977f, 0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f };
    constexpr float s[16] = { 0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f, 0.92387953251128674f,
                              0.70710678118654752f, 0.38268343236508977f, 0.0f, -0.38268343236508977f, -0.70710678118654752f,
                              -0.92387953251128674f, -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f };
    return make_float2(c[k & 15], INV ? s[k & 15] : -s[k & 15]);
}

template <bool INV> __device__ __forceinline__ void kc_dft2(kcf &a, kcf &b)
{
    const kcf t = kc_sub(a, b);
    a = kc_add(a, b);
    b = t;
}
template <bool INV> __device__ __forceinline__ void kc_dft4(kcf &x0, kcf &x1, kcf &x2, kcf &x3)
{
    const kcf a = kc_add(x0, x2), b = kc_sub(x0, x2), c = kc_add(x1, x3), d = kc_rot<INV>(kc_sub(x1, x3));
    x0 = kc_add(a, c); x2 = kc_sub(a, c); x1 = kc_add(b, d); x3 = kc_sub(b, d);
}
// in-place DFT of R points, natural order in and out.  R = Ra*Rb: X[q0 + Rb q1] = sum_m0 w_R^(m0 q0) w_Ra^(m0 q1) [sum_m1 x[Ra m1 + m0] w_Rb^(m1 q0)]
template <int R, bool INV> __device__ __forceinline__ void kc_dft(kcf (&x)[R])
{
    if constexpr (R == 2) kc_dft2<INV>(x[0], x[1]);
    else if constexpr (R == 4) kc_dft4<INV>(x[0], x[1], x[2], x[3]);
    else if constexpr (R == 8) {           // Ra = 2, Rb = 4
        kc_dft4<INV>(x[0], x[2], x[4], x[6]);          // m0 = 0: t[0][q0] in x[2 q0]
        kc_dft4<INV>(x[1], x[3], x[5], x[7]);          // m0 = 1: t[1][q0] in x[2 q0 + 1]
#pragma unroll
        for (int q0 = 1; q0 < 4; q0++) x[2 * q0 + 1] = kc_mul(x[2 * q0 + 1], kc_w16<INV>(2 * q0));   // w_8^(q0)
        kcf y[8];
#pragma unroll
        for (int q0 = 0; q0 < 4; q0++) { kcf a = x[2 * q0], b = x[2 * q0 + 1]; kc_dft2<INV>(a, b); y[q0] = a; y[q0 + 4] = b; }
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = y[q];
    } else {                                // R == 16: Ra = Rb = 4
        static_assert(R == 16, "radix");
#pragma unroll
        for (int m0 = 0; m0 < 4; m0++) kc_dft4<INV>(x[m0], x[4 + m0], x[8 + m0], x[12 + m0]);       // t[m0][q0] in x[4 q0 + m0]
#pragma unroll
        for (int q0 = 1; q0 < 4; q0++)
#pragma unroll
            for (int m0 = 1; m0 < 4; m0++) x[4 * q0 + m0] = kc_mul(x[4 * q0 + m0], kc_w16<INV>(m0 * q0));
        kcf y[16];
#pragma unroll
        for (int q0 = 0; q0 < 4; q0++) {
            kcf a = x[4 * q0], b = x[4 * q0 + 1], c = x[4 * q0 + 2], d = x[4 * q0 + 3];
            kc_dft4<INV>(a, b, c, d);                  // over m0 -> q1
            y[q0] = a; y[q0 + 4] = b; y[q0 + 8] = c; y[q0 + 12] = d;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = y[q];
    }
}

// x[q] *= w1^q, q = 1 .. R-1 (running product; a product tree of depth log2 R measured the same: tools/spec_lab.hip)
template <int R> __device__ __forceinline__ void kc_twiddle(kcf (&x)[R], kcf w1)
{
    kcf w = w1;
#pragma unroll
    for (int q = 1; q < R; q++) { x[q] = kc_mul(x[q], w); w = kc_mul(w, w1); }
}

// One in-place stage over `nseq` sequences of length n held in LDS (padded index, `sstride` elements apart).
// Forward (DIF): y_q = w_L^(i q) * DFT_R(x)_q ; inverse (DIT): x = IDFT_R( conj(w_L^(i q)) y_q ).  tw[k] = exp(-2 pi i k / n).
// LG_S = log2 of the element stride S inside a butterfly, compile-time: the padded positions of the R elements are then
// pad(e0) + CONSTANTS -- pad(e0 + q S) = pad(e0) + q S + (q S >> 4) in every case that occurs (S >= 16: whole runs of 16 are
// skipped; S < 16 <= R S: e0 = 16-aligned + i with i < S, no carry out of i; R S <= 16: the butterfly lies inside one run) --
// so the LDS instructions carry immediate offsets and one address is computed per butterfly instead of one per element
// (measured: the index arithmetic was ~40 % of the VALU instructions of a stage, and the column kernel is VALU-bound).
template <int R, bool INV, int LG_S>
__device__ __forceinline__ void kspec_stage(kcf *lds, int sstride, int nseq, int lg_n, const kcf *__restrict__ tw)
{
    constexpr int lgR = R == 16 ? 4 : (R == 8 ? 3 : (R == 4 ? 2 : 1));
    constexpr int lg_L = LG_S + lgR, S = 1 << LG_S;
    const int lg_per = lg_n - lgR;          // butterflies per sequence
    const int total = nseq << lg_per;
    for (int item = threadIdx.x; item < total; item += blockDim.x) {
        const int s = item >> lg_per, t = item & ((1 << lg_per) - 1);
        const int g = t >> LG_S, i = t & (S - 1);
        kcf *b0 = lds + (long long)s * sstride + kspec_pad((g << lg_L) + i);
        kcf x[R];
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = b0[q * S + ((q * S) >> 4)];
        if (!INV) {
            kc_dft<R, false>(x);
            if (LG_S > 0) {
                kc_twiddle<R>(x, tw[i << (lg_n - lg_L)]);
            }
        } else {
            if (LG_S > 0) {
                kc_twiddle<R>(x, kc_conj(tw[i << (lg_n - lg_L)]));
            }
            kc_dft<R, true>(x);
        }
#pragma unroll
        for (int q = 0; q < R; q++) b0[q * S + ((q * S) >> 4)] = x[q];
    }
}

// Stage 0 (radix 16, L = n: every plan starts with it) fused with the global-memory side of a kernel:
//   forward: the 16 butterfly inputs come straight from ld(sequence, element) -- no staging pass through the LDS, and the
//            16 loads of a thread are all in flight at once; outputs go to the LDS for the remaining stages
//   inverse: the last stage of the DIT inverse hands its 16 outputs to st(sequence, element, value)
// Consecutive lanes own consecutive elements, so every load/store instruction of a wave is one contiguous run.
template <int LG_S, typename LD>
__device__ __forceinline__ void kspec_stage0_fwd_from_t(kcf *lds, int sstride, int nseq, const kcf *__restrict__ tw, LD ld)
{
    constexpr int S = 1 << LG_S;
    const int total = nseq << LG_S;
    for (int item = threadIdx.x; item < total; item += blockDim.x) {
        const int s = item >> LG_S, i = item & (S - 1);
        kcf x[16];
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = ld(s, i, q);              // element i + q S
        kc_dft<16, false>(x);
        if (LG_S > 0) {
            kc_twiddle<16>(x, tw[i]);
        }
        kcf *b0 = lds + (long long)s * sstride + kspec_pad(i);
#pragma unroll
        for (int q = 0; q < 16; q++) b0[q * S + ((q * S) >> 4)] = x[q];
    }
}
template <int LG_S, typename ST>
__device__ __forceinline__ void kspec_stage0_inv_to_t(const kcf *lds, int sstride, int nseq, const kcf *__restrict__ tw, ST st)
{
    constexpr int S = 1 << LG_S;
    const int total = nseq << LG_S;
    for (int item = threadIdx.x; item < total; item += blockDim.x) {
        const int s = item >> LG_S, i = item & (S - 1);
        const kcf *b0 = lds + (long long)s * sstride + kspec_pad(i);
        kcf x[16];
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = b0[q * S + ((q * S) >> 4)];
        if (LG_S > 0) {
            kc_twiddle<16>(x, kc_conj(tw[i]));
        }
        kc_dft<16, true>(x);
        st(s, i, x);                                   // the 16 outputs: elements i + q S, q = 0..15
    }
}
// run-time stride -> the compile-time instance (n = 32 ... 16384)
#define KSPEC_LGS_DISPATCH(lgs, CALL) \
    switch (lgs) { \
    case 1: { constexpr int LGS = 1; CALL; } break; case 2: { constexpr int LGS = 2; CALL; } break; case 3: { constexpr int LGS = 3; CALL; } break; \
    case 4: { constexpr int LGS = 4; CALL; } break; case 5: { constexpr int LGS = 5; CALL; } break; case 6: { constexpr int LGS = 6; CALL; } break; \
    case 7: { constexpr int LGS = 7; CALL; } break; case 8: { constexpr int LGS = 8; CALL; } break; case 9: { constexpr int LGS = 9; CALL; } break; \
    default: { constexpr int LGS = 10; CALL; } break; }
template <typename LD>
__device__ __forceinline__ void kspec_stage0_fwd_from(kcf *lds, int sstride, int nseq, int lg_n, const kcf *__restrict__ tw, LD ld)
{
    KSPEC_LGS_DISPATCH(lg_n - 4, (kspec_stage0_fwd_from_t<LGS>(lds, sstride, nseq, tw, ld)));
}
template <typename ST>
__device__ __forceinline__ void kspec_stage0_inv_to(const kcf *lds, int sstride, int nseq, int lg_n, const kcf *__restrict__ tw, ST st)
{
    KSPEC_LGS_DISPATCH(lg_n - 4, (kspec_stage0_inv_to_t<LGS>(lds, sstride, nseq, tw, st)));
}

// plans are radix 16 from the top with one smaller last stage, or [.., 8, 4] (spec_plan): a radix below 16 runs at S = 1, radix 8 also at S = 4
template <bool INV>
__device__ __forceinline__ void kspec_stage_any(int radix, kcf *lds, int sstride, int nseq, int lg_n, int lg_L, const kcf *__restrict__ tw)
{
    if (radix == 16) {
        switch (lg_L - 4) {
        case 0: kspec_stage<16, INV, 0>(lds, sstride, nseq, lg_n, tw); break;
        case 1: kspec_stage<16, INV, 1>(lds, sstride, nseq, lg_n, tw); break;
        case 2: kspec_stage<16, INV, 2>(lds, sstride, nseq, lg_n, tw); break;
        case 3: kspec_stage<16, INV, 3>(lds, sstride, nseq, lg_n, tw); break;
        case 4: kspec_stage<16, INV, 4>(lds, sstride, nseq, lg_n, tw); break;
        case 5: kspec_stage<16, INV, 5>(lds, sstride, nseq, lg_n, tw); break;
        case 6: kspec_stage<16, INV, 6>(lds, sstride, nseq, lg_n, tw); break;
        case 7: kspec_stage<16, INV, 7>(lds, sstride, nseq, lg_n, tw); break;
        case 8: kspec_stage<16, INV, 8>(lds, sstride, nseq, lg_n, tw); break;
        case 9: kspec_stage<16, INV, 9>(lds, sstride, nseq, lg_n, tw); break;
        default: kspec_stage<16, INV, 10>(lds, sstride, nseq, lg_n, tw); break;      // n = 16384, the largest spec_plan accepts
        }
    }
    else if (radix == 8) { if (lg_L == 3) kspec_stage<8, INV, 0>(lds, sstride, nseq, lg_n, tw); else kspec_stage<8, INV, 2>(lds, sstride, nseq, lg_n, tw); }      // S = 1, or S = 4 in [.., 8, 4]
    else if (radix == 4) kspec_stage<4, INV, 0>(lds, sstride, nseq, lg_n, tw);
    else kspec_stage<2, INV, 0>(lds, sstride, nseq, lg_n, tw);
}
__device__ __forceinline__ int kspec_lg(int r) { return r == 16 ? 4 : (r == 8 ? 3 : (r == 4 ? 2 : 1)); }

// The radix-3 stage of a 3 * 2^lg transform: butterflies over the elements {i, i + 2^lg, i + 2 * 2^lg}, i.e. the same index of the three
// sub-sequences.  Forward (DIF): y_b = w_n^(i b) DFT3(x)_b; inverse (DIT): x = IDFT3(conj(w_n^(i b)) y_b).  tw3[i] = exp(-2 pi i i / n), i < 2^lg
// (stored behind the power-of-two table: spec_twiddles).
template <bool INV>
__device__ __forceinline__ void kspec_stage3(kcf *lds, int ss, int nseq, int lg, const kcf *__restrict__ tw3)
{
    const float sq = 0.86602540378443865f;
    const int total = nseq << lg;
    for (int item = threadIdx.x; item < total; item += blockDim.x) {
        const int s = item >> lg, i = item & ((1 << lg) - 1);
        kcf *b0 = lds + (long long)s * 3 * ss + i + (i >> 4);
        kcf x0 = b0[0], x1 = b0[ss], x2 = b0[2 * ss];
        kcf w1 = tw3[i];
        if (INV) w1 = kc_conj(w1);
        const kcf w2 = kc_mul(w1, w1);
        if (INV) { x1 = kc_mul(x1, w1); x2 = kc_mul(x2, w2); }
        const kcf t = kc_add(x1, x2), u = kc_sub(x1, x2);
        const kcf a = make_float2(x0.x - 0.5f * t.x, x0.y - 0.5f * t.y);
        const kcf r = INV ? make_float2(-sq * u.y, sq * u.x) : make_float2(sq * u.y, -sq * u.x);      // (+-i) (sqrt 3 / 2) u
        kcf y1 = kc_add(a, r), y2 = kc_sub(a, r);
        if (!INV) { y1 = kc_mul(y1, w1); y2 = kc_mul(y2, w2); }
        b0[0] = kc_add(x0, t); b0[ss] = y1; b0[2 * ss] = y2;
    }
}

// all stages from `first` on (power-of-two plans only: first > 0 is the fused first stage); the caller has synchronised after filling the
// LDS; returns synchronised.  sstride = kspec_sstride(P).
__device__ __forceinline__ void kspec_fft_fwd(const KFFTPlan &P, kcf *lds, int sstride, int nseq, const kcf *__restrict__ tw, int first = 0)
{
    if (P.m == 3) {
        kspec_stage3<false>(lds, kspec_ss(P), nseq, P.lg, tw + (1 << P.lg));
        __syncthreads();
        sstride = kspec_ss(P); nseq *= 3;
    }
    int lg_L = P.lg;
    for (int s = 0; s < first; s++) lg_L -= kspec_lg(P.radix[s]);
    for (int s = first; s < P.nstage; s++) {
        kspec_stage_any<false>(P.radix[s], lds, sstride, nseq, P.lg, lg_L, tw);
        __syncthreads();
        lg_L -= kspec_lg(P.radix[s]);
    }
}
// stages nstage-1 ... last (0 = all of them; last > 0: power-of-two plans, the fused last stage follows)
__device__ __forceinline__ void kspec_fft_inv(const KFFTPlan &P, kcf *lds, int sstride, int nseq, const kcf *__restrict__ tw, int last = 0)
{
    const int ss = P.m == 3 ? kspec_ss(P) : sstride, ns = P.m == 3 ? 3 * nseq : nseq;
    int lg_L = 0;
    for (int s = P.nstage - 1; s >= last; s--) {
        lg_L += kspec_lg(P.radix[s]);
        kspec_stage_any<true>(P.radix[s], lds, ss, ns, P.lg, lg_L, tw);
        __syncthreads();
    }
    if (P.m == 3) {
        kspec_stage3<true>(lds, ss, nseq, P.lg, tw + (1 << P.lg));
        __syncthreads();
    }
}

__device__ __forceinline__ int kspec_tile(int b, int ntiles)
{
    // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous band of row tiles, so that the RB-row store
    // segments of neighbouring tiles meet in ONE L2 and leave it as full lines
    return (ntiles & 7) ? b : (b & 7) * (ntiles >> 3) + (b >> 3);
}

// rows of v (F fp64 planes, x fastest, no ghosts) -> W[pair][pos][y] ; blockIdx.y = pair
// TIN: storage type of v (double; float for the defect-correction residual, which only this kernel ever reads)
template <typename TIN>
__global__ void __launch_bounds__(1024) k_spec_rows_fwd(KFFTPlan PX, int nyp /* column stride of W */, int rb, int ntiles, int F, const TIN *__restrict__ v, long long plane,
                                                        kcf *__restrict__ W, const kcf *__restrict__ tw, KSpecLin ex)
{
    // ex: the transform is taken of v + sum_j ex.a[j] * ex.p[j] (initial guess of the defect correction: b - sum c_j b_j)
    extern __shared__ kcf kspec_lds[];
    const int nx = PX.n, p = blockIdx.y;
    const int y0 = kspec_tile(blockIdx.x, ntiles < 0 ? -ntiles : ntiles) * rb;
    const int sstride = kspec_sstride(PX);
    const TIN *va = v + (long long)(2 * p) * plane + (long long)y0 * nx;
    const bool has_b = 2 * p + 1 < F;
    const TIN *vb = has_b ? va + plane : va;
    const int half = nx >> 1, lg_half = PX.lg - 1;
    const int lg_rb = 31 - __clz(rb);             // rb is a power of two (spec_build)
    auto row_of = [&](int idx) { return PX.m == 1 ? idx >> lg_half : (idx >> lg_half) / 3; };      // idx / half
    // loads in batches of 4 items per thread, all issued before the first LDS store (one memory latency per batch, not per item).
    // fp32 input (the residual of the defect correction): first stage straight from global memory, 16 x 2 scalar loads in flight
    // per thread (85 -> 76 us in the solver; with fp64 input the same fusion is SLOWER, 118 -> 136 us, and stays off)
    if (sizeof(TIN) == 4 && PX.m == 1 && PX.nstage > 0 && PX.radix[0] == 16 && ex.n == 0 && (PX.flags & 4)) {
        const int S0 = nx >> 4;
        kspec_stage0_fwd_from(kspec_lds, sstride, rb, PX.lg, tw, [&](int r, int i, int q) {
            const TIN *pa = va + (long long)r * nx + i, *pb = vb + (long long)r * nx + i;
            const TIN a = pa[q * S0];
            const TIN b = has_b ? pb[q * S0] : (TIN)0;
            return make_float2((float)a, (float)b);
        });
        __syncthreads();
        kspec_fft_fwd(PX, kspec_lds, sstride, rb, tw, 1);
    } else {
    for (int base = 0; base < rb * half; base += 4 * blockDim.x) {
        double2 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            a[u] = b[u] = make_double2(0.0, 0.0);
            if (idx < rb * half) {
                const int r = row_of(idx), x = 2 * (idx - r * half);
                a[u] = ksfd_ld2(va + (long long)r * nx + x);
                if (has_b) b[u] = ksfd_ld2(vb + (long long)r * nx + x);
                for (int j = 0; j < ex.n; j++) {            // wave-uniform, usually 0
                    const long long o = (long long)(2 * p) * plane + (long long)(y0 + r) * nx + x;
                    const double2 ea = ksfd_ld2(ex.p[j] + o);
                    a[u].x += ex.a[j] * ea.x; a[u].y += ex.a[j] * ea.y;
                    if (has_b) { const double2 eb = ksfd_ld2(ex.p[j] + o + plane); b[u].x += ex.a[j] * eb.x; b[u].y += ex.a[j] * eb.y; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            if (idx < rb * half) {
                const int r = row_of(idx), x = 2 * (idx - r * half);
                kcf *row = kspec_lds + r * sstride;
                row[kspec_lpos(PX, x)] = make_float2((float)a[u].x, (float)b[u].x);
                row[kspec_lpos(PX, x + 1)] = make_float2((float)a[u].y, (float)b[u].y);
            }
        }
    }
    __syncthreads();
    kspec_fft_fwd(PX, kspec_lds, sstride, rb, tw);
    }
    if (ntiles < 0) {
        // one rank: TILE-MAJOR store Wt[pair][tile][pos][r] -- one contiguous run per block (the transposed store below writes
        // 32-B segments, 151 -> 95 us); the column kernel gathers its columns from the tiles instead (strided READS are cheap: the
        // four positions of a 128-B line belong to blocks that run side by side on one XCD)
        const long long nt = -ntiles, t = y0 / rb;
        kcf *Wt = W + (long long)p * nt * nx * rb;
        if (PX.lgw < 0) {
            Wt += t * nx * rb;
            for (int idx = threadIdx.x; idx < rb * nx; idx += blockDim.x) {
                const int j = idx >> lg_rb, r = idx & (rb - 1);
                Wt[idx] = kspec_lds[r * sstride + kspec_lpos(PX, j)];
            }
        } else {
            for (int idx = threadIdx.x; idx < rb * nx; idx += blockDim.x) {
                const int j = idx >> lg_rb, r = idx & (rb - 1);
                Wt[kspec_wt_index(PX.lgw, lg_rb, nt, nx, t, j, r)] = kspec_lds[r * sstride + kspec_lpos(PX, j)];
            }
        }
        return;
    }
    kcf *Wp = W + (long long)p * nx * nyp + y0;
    for (int idx = threadIdx.x; idx < rb * nx; idx += blockDim.x) {
        const int j = idx >> lg_rb, r = idx & (rb - 1);
        Wp[(long long)j * nyp + r] = kspec_lds[r * sstride + kspec_lpos(PX, j)];
    }
}

// W[pair][pos][y] -> rows of z (F fp64 planes)
// z = sum_j add.a[j] * add.p[j] + M^-1 v (a p[j] may alias z: every element is read and written by the same thread)
__global__ void __launch_bounds__(1024) k_spec_rows_inv(KFFTPlan PX, int nyp, int rb, int ntiles, int F, const kcf *__restrict__ W,
                                                        double *z, long long plane, const kcf *__restrict__ tw, KSpecLin add)
{
    extern __shared__ kcf kspec_lds[];
    const int nx = PX.n, p = blockIdx.y;
    const int y0 = kspec_tile(blockIdx.x, ntiles) * rb;
    const int sstride = kspec_sstride(PX);
    const kcf *Wp = W + (long long)p * nx * nyp + y0;
    const int lg_rb = 31 - __clz(rb), lg_half = PX.lg - 1;
    const kcf *Wtile = W + ((long long)p * ntiles + y0 / rb) * nx * rb;      // (PX.flags & 8: the column kernel stored tile-major)
    for (int base = 0; base < rb * nx; base += 8 * blockDim.x) {
        kcf t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            if (idx < rb * nx) { const int j = idx >> lg_rb, r = idx & (rb - 1); t[u] = (PX.flags & 8) ? Wtile[idx] : Wp[(long long)j * nyp + r]; }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            if (idx < rb * nx) { const int j = idx >> lg_rb, r = idx & (rb - 1); kspec_lds[r * sstride + kspec_lpos(PX, j)] = t[u]; }
        }
    }
    __syncthreads();
    const bool has_b = 2 * p + 1 < F;
    const long long o0 = (long long)(2 * p) * plane + (long long)y0 * nx;
    if (PX.m == 1 && PX.nstage > 0 && PX.radix[0] == 16) {
        kspec_fft_inv(PX, kspec_lds, sstride, rb, tw, 1);
        const int S0 = nx >> 4;
        kspec_stage0_inv_to(kspec_lds, sstride, rb, PX.lg, tw, [&](int r, int i, kcf (&c)[16]) {
            const long long o = o0 + (long long)r * nx + i;
            // the vectors added to the result (the iterate x of the defect correction) in batches of 8 elements, ALL loads of a
            // batch issued before its first store: z may alias them, so left to itself the compiler keeps load-store order and
            // every thread has two loads in flight (197 us with the x update against 100 us without, 4096^2)
#pragma unroll
            for (int hb = 0; hb < 16; hb += 8) {
                double a[8], b[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { a[q] = (double)c[hb + q].x; b[q] = (double)c[hb + q].y; }
#pragma unroll
                for (int j = 0; j < 3; j++)         // (constant trip count: a run-time bound put the by-value struct into scratch memory)
                    if (j < add.n) {
                        double ta[8], tb[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) { ta[q] = add.p[j][o + (hb + q) * S0]; tb[q] = has_b ? add.p[j][o + (hb + q) * S0 + plane] : 0.0; }
#pragma unroll
                        for (int q = 0; q < 8; q++) { a[q] += add.a[j] * ta[q]; b[q] += add.a[j] * tb[q]; }
                    }
#pragma unroll
                for (int q = 0; q < 8; q++) { z[o + (hb + q) * S0] = a[q]; if (has_b) z[o + (hb + q) * S0 + plane] = b[q]; }
            }
        });
        return;
    }
    kspec_fft_inv(PX, kspec_lds, sstride, rb, tw);       // plans that do not start with radix 16 (experiment knob) / no transform (KSFD_SPEC_DIAG)
    const int half = nx >> 1;
    for (int idx = threadIdx.x; idx < rb * half; idx += blockDim.x) {
        const int r = PX.m == 1 ? idx >> lg_half : (idx >> lg_half) / 3, x = 2 * (idx - r * half);
        const kcf *row = kspec_lds + r * sstride;
        const kcf c0 = row[kspec_lpos(PX, x)], c1 = row[kspec_lpos(PX, x + 1)];
        const long long o = o0 + (long long)r * nx + x;
        double2 a = make_double2((double)c0.x, (double)c1.x), b = make_double2((double)c0.y, (double)c1.y);
        for (int j = 0; j < add.n; j++) {
            const double2 xa = *reinterpret_cast<const double2 *>(add.p[j] + o);
            a.x += add.a[j] * xa.x; a.y += add.a[j] * xa.y;
            if (has_b) { const double2 xb = *reinterpret_cast<const double2 *>(add.p[j] + o + plane); b.x += add.a[j] * xb.x; b.y += add.a[j] * xb.y; }
        }
        *reinterpret_cast<double2 *>(z + o) = a;
        if (has_b) *reinterpret_cast<double2 *>(z + o + plane) = b;
    }
}

// The symbol stage for one pair of points k, -k: a[p] = c^_p(k), b[p] = c^_p(-k) per field pair in; c^z_p(k), c^z_p(-k) out.
// L2 = the Laplacian symbol at k (sum over the axes).
template <int NL>
__device__ __forceinline__ void kspec_symbol(const KSpecSym &S, float L2, kcf (&a)[(NL + 2) / 2], kcf (&b)[(NL + 2) / 2])
{
    constexpr int F = NL + 1, npair = (F + 1) / 2;
    kcf vh[2 * npair];
#pragma unroll
    for (int p = 0; p < npair; p++) {
        const kcf bc = kc_conj(b[p]);
        const kcf su = kc_add(a[p], bc), di = kc_sub(a[p], bc);
        vh[2 * p] = make_float2(0.5f * su.x, 0.5f * su.y);
        vh[2 * p + 1] = make_float2(0.5f * di.y, -0.5f * di.x);          // (a - conj b) / (2i)
    }
    float den = S.shift - S.a_rr * L2;
    kcf num = vh[0];
    float invd[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const float d = S.shift + S.gam[l] - S.D[l] * L2;
        invd[l] = __builtin_amdgcn_rcpf(d);
        const float c = S.a_rU[l] * L2 * invd[l];
        den -= c * S.s[l];
        num.x += c * vh[l + 1].x; num.y += c * vh[l + 1].y;
    }
    if (!(fabsf(den) >= S.den_floor)) den = den < 0.0f ? -S.den_floor : S.den_floor;
    const float sc = S.scale * __builtin_amdgcn_rcpf(den);
    kcf zh[2 * npair];
    zh[0] = make_float2(num.x * sc, num.y * sc);
#pragma unroll
    for (int l = 0; l < NL; l++)
        zh[l + 1] = make_float2((vh[l + 1].x * S.scale + S.s[l] * zh[0].x) * invd[l], (vh[l + 1].y * S.scale + S.s[l] * zh[0].y) * invd[l]);
    if (F & 1) zh[F] = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int p = 0; p < npair; p++) {
        const kcf za = zh[2 * p], zb = zh[2 * p + 1];
        a[p] = make_float2(za.x - zb.y, za.y + zb.x);                     // c_z(k)  = z_a + i z_b
        b[p] = make_float2(za.x + zb.y, -za.y + zb.x);                    // c_z(-k) = conj(z_a) + i conj(z_b)
    }
}

// one block per {kx, -kx}: forward FFT along y, symbol, inverse FFT along y, in place in W
// (templated on the ligand count: the per-point arrays of the symbol stage must stay in registers -- with run-time loop
//  bounds they went to scratch memory and the kernel took 126 us of pure data movement)
// Column storage: column (pair p, local position jl) consists of `ny >> lg_pl` pieces of 2^lg_pl elements, `pstride` elements
// apart (one piece per slab rank after the all-to-all; a single piece of ny elements on one rank); pairtab[block] =
// (jlA, jlB, kxA, self): the two local positions the block owns, the wavenumber of the first, and whether both are self-paired.
// lg_rb >= 0 (one rank): the columns are READ from the tile-major array Wt[pair][tile][pos][r] the row kernel wrote
// (r = y mod 2^lg_rb) and written to W[pair][pos][y] for the inverse row kernel; lg_rb < 0: in place in W.
// the symbol stage of k_spec_cols for NL ligands.  item -> the pair of points k = (colA', ky), -k = (colB', -ky), walked in POSITION
// order of ky so that both LDS accesses of a wave are consecutive (the partner positions of consecutive positions run backwards)
template <int NL>
__device__ __forceinline__ void kspec_cols_symbol(const KFFTPlan &PY, kcf *lds, int sstride, bool self, int kxA, int kxB, const int *__restrict__ posy,
                                                  const int *__restrict__ kyofpos, const float *__restrict__ lx, const float *__restrict__ ly,
                                                  const int2 *__restrict__ ytab, const KSpecSym &S)
{
    constexpr int F = NL + 1, npair = (F + 1) / 2;
    const int ny = PY.n, half = ny >> 1;
    if (!self) {
        // every block but one.  ytab[pos] = (position of -ky, bits of ly[ky]) for ky = the wavenumber AT position pos: one coalesced
        // 8-B load per item, issued four items ahead -- the chain kyofpos[pos] -> posy[-ky], ly[ky] of three scattered, dependent table
        // look-ups per item was a quarter of the kernel (4096^2: 35 of 150 us with everything else in place)
        const float lxa = lx[kxA];
        for (int base = threadIdx.x; base < ny; base += 4 * blockDim.x) {
            int2 t[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int mpos = base + u * blockDim.x; if (mpos < ny) t[u] = ytab[mpos]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int mpos = base + u * blockDim.x;
                if (mpos >= ny) break;
                const int m = kspec_lpos(PY, mpos), mp = kspec_lpos(PY, t[u].x);
                const float L2 = lxa + __int_as_float(t[u].y);
                kcf a[npair], b[npair];
#pragma unroll
                for (int p = 0; p < npair; p++) { a[p] = lds[(2 * p) * sstride + m]; b[p] = lds[(2 * p + 1) * sstride + mp]; }
                kspec_symbol<NL>(S, L2, a, b);
#pragma unroll
                for (int p = 0; p < npair; p++) { lds[(2 * p) * sstride + m] = a[p]; lds[(2 * p + 1) * sstride + mp] = b[p]; }
            }
        }
        return;
    }
    // the block of the two self-paired columns kx = 0 and kx = nx/2: (c, ky) and (c, -ky) are one item
    for (int item = threadIdx.x; item < 2 * ny; item += blockDim.x) {
        const int c = item >= ny, mpos = item - c * ny;
        const int ky = kyofpos[mpos];
        if (ky > half) continue;
        const int kym = ky ? ny - ky : 0;
        const int m = kspec_lpos(PY, mpos), mp = kspec_lpos(PY, posy[kym]);
        const float L2 = lx[c ? kxB : kxA] + ly[ky];
        kcf a[npair], b[npair];
#pragma unroll
        for (int p = 0; p < npair; p++) { a[p] = lds[(2 * p + c) * sstride + m]; b[p] = lds[(2 * p + c) * sstride + mp]; }
        kspec_symbol<NL>(S, L2, a, b);
#pragma unroll
        for (int p = 0; p < npair; p++) { lds[(2 * p + c) * sstride + m] = a[p]; lds[(2 * p + c) * sstride + mp] = b[p]; }
    }
}
// compile-time ligand count for the per-point register arrays of the symbol stage, chosen INSIDE the kernel: the transforms around it
// do not depend on it, and one instance of them per kernel instead of twelve is what keeps the build at a minute
#define KSPEC_NL_SWITCH(nl, CALL) \
    switch (nl) { \
    case 1: { constexpr int NL = 1; CALL; } break; case 2: { constexpr int NL = 2; CALL; } break; case 3: { constexpr int NL = 3; CALL; } break; \
    case 4: { constexpr int NL = 4; CALL; } break; case 5: { constexpr int NL = 5; CALL; } break; case 6: { constexpr int NL = 6; CALL; } break; \
    case 7: { constexpr int NL = 7; CALL; } break; case 8: { constexpr int NL = 8; CALL; } break; case 9: { constexpr int NL = 9; CALL; } break; \
    case 10: { constexpr int NL = 10; CALL; } break; case 11: { constexpr int NL = 11; CALL; } break; default: { constexpr int NL = 12; CALL; } break; }

// NPAIR_T: number of field pairs at compile time for the common cases (1: one ligand, 2: two or three), 0 = run-time value.  With the
// sequence count known the loops around the transforms simplify: 134 us against 145 us for the run-time version at 4096^2, F = 2.
template <int NPAIR_T>
__global__ void __launch_bounds__(1024) k_spec_cols(KFFTPlan PY, int nxl, int lg_pl, long long pstride, kcf *__restrict__ W, const kcf *__restrict__ Wt, int lg_rb, const kcf *__restrict__ tw,
                                                   const int4 *__restrict__ pairtab, const int *__restrict__ posy, const int *__restrict__ kyofpos,
                                                   const float *__restrict__ lx, const float *__restrict__ ly, const int2 *__restrict__ ytab, KSpecSym S)
{
    extern __shared__ kcf kspec_lds[];
    const int npair = NPAIR_T ? NPAIR_T : (S.nlig + 2) / 2;
    const int ny = PY.n;
    const int4 pt = pairtab[kspec_tile(blockIdx.x, gridDim.x)];       // consecutive pairs (= neighbouring positions) on one XCD
    const bool self = pt.w != 0;                                // kx = 0 and kx = nx/2 are their own partners
    const int jA = pt.x, jB = pt.y, kxA = pt.z, kxB = self ? pt.w - 1 : pt.z;     // lx is even: lx[-kx] = lx[kx]
    const int sstride = kspec_sstride(PY);
    const int nseq = 2 * npair;
    const int half = ny >> 1, lg_half = PY.lg - 1;
    auto seq_of = [&](int idx) { return PY.m == 1 ? idx >> lg_half : (idx >> lg_half) / 3; };      // idx / half
    const int plmask = (1 << lg_pl) - 1;
    auto colat = [&](int s, int y) {                           // y even: a float4 never straddles two pieces
        return W + (long long)(y >> lg_pl) * pstride + (((long long)(s >> 1) * nxl + ((s & 1) ? jB : jA)) << lg_pl) + (y & plmask);
    };
    const bool tm_out = (PY.flags & 8) && lg_rb >= 1;           // result tile-major for the inverse row kernel (one rank; a float4 = two rows of one tile)
    const bool r16 = PY.m == 1 && PY.nstage > 0 && PY.radix[0] == 16 && (lg_rb < 0 || (ny >> 4) >= (1 << lg_rb));      // (a tile never holds two butterfly elements)
    if (r16 && (PY.flags & 1)) {
        // element y = i + q S0 (S0 = ny/16 >= 2^lg_rb and >= 2^lg_pl pieces are whole multiples): q moves by a constant stride
        const int S0 = ny >> 4;
        const long long qs = lg_rb >= 0 ? (PY.lgw < 0 ? (long long)S0 * nxl : (long long)S0 << PY.lgw) : (lg_pl >= PY.lg ? (long long)S0 : 0);
        kspec_stage0_fwd_from(kspec_lds, sstride, nseq, PY.lg, tw, [&](int s, int i, int q) {
            const kcf *p0 = lg_rb >= 0 ? Wt + (long long)(s >> 1) * nxl * ny + kspec_wt_index(PY.lgw, lg_rb, ny >> lg_rb, nxl, i >> lg_rb, (s & 1) ? jB : jA, i & ((1 << lg_rb) - 1))
                                       : colat(s, i);
            return qs ? p0[q * qs] : *colat(s, i + q * S0);
        });
        __syncthreads();
        kspec_fft_fwd(PY, kspec_lds, sstride, nseq, tw, 1);
    } else {
    for (int base = 0; base < nseq * half; base += 8 * blockDim.x) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            if (idx < nseq * half) {
                const int s = seq_of(idx), y = 2 * (idx - s * half);
                t[u] = lg_rb >= 0 ? *reinterpret_cast<const float4 *>(Wt + (long long)(s >> 1) * nxl * ny + kspec_wt_index(PY.lgw, lg_rb, ny >> lg_rb, nxl, y >> lg_rb, (s & 1) ? jB : jA, y & ((1 << lg_rb) - 1)))
                                   : *reinterpret_cast<const float4 *>(colat(s, y));
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            if (idx < nseq * half) {
                const int s = seq_of(idx), y = 2 * (idx - s * half);
                kcf *q = kspec_lds + s * sstride;
                q[kspec_lpos(PY, y)] = make_float2(t[u].x, t[u].y);
                q[kspec_lpos(PY, y + 1)] = make_float2(t[u].z, t[u].w);
            }
        }
    }
    __syncthreads();
    kspec_fft_fwd(PY, kspec_lds, sstride, nseq, tw);
    }
    if (NPAIR_T == 1) kspec_cols_symbol<1>(PY, kspec_lds, sstride, self, kxA, kxB, posy, kyofpos, lx, ly, ytab, S);
    else if (NPAIR_T == 2) {
        if (S.nlig == 2) kspec_cols_symbol<2>(PY, kspec_lds, sstride, self, kxA, kxB, posy, kyofpos, lx, ly, ytab, S);
        else kspec_cols_symbol<3>(PY, kspec_lds, sstride, self, kxA, kxB, posy, kyofpos, lx, ly, ytab, S);
    } else { KSPEC_NL_SWITCH(S.nlig, (kspec_cols_symbol<NL>(PY, kspec_lds, sstride, self, kxA, kxB, posy, kyofpos, lx, ly, ytab, S))); }
    __syncthreads();
    if (r16 && (PY.flags & 2)) {
        kspec_fft_inv(PY, kspec_lds, sstride, nseq, tw, 1);
        const int S0 = ny >> 4;
        const bool whole = lg_pl >= PY.lg;                       // one piece per column (one rank)
        kspec_stage0_inv_to(kspec_lds, sstride, nseq, PY.lg, tw, [&](int s, int i, kcf (&c)[16]) {
            if (tm_out) {               // y = i + q S0 -> tile (i >> lg_rb) + q (S0 >> lg_rb): a constant stride of S0 * nxl elements
                kcf *d0 = W + (long long)(s >> 1) * nxl * ny + kspec_wt_index(-1, lg_rb, ny >> lg_rb, nxl, i >> lg_rb, (s & 1) ? jB : jA, i & ((1 << lg_rb) - 1));
#pragma unroll
                for (int q = 0; q < 16; q++) d0[(long long)q * S0 * nxl] = c[q];
                return;
            }
#pragma unroll
            for (int q = 0; q < 16; q++) { if (whole) colat(s, i)[q * S0] = c[q]; else *colat(s, i + q * S0) = c[q]; }
        });
        return;
    }
    kspec_fft_inv(PY, kspec_lds, sstride, nseq, tw);
    for (int idx = threadIdx.x; idx < nseq * half; idx += blockDim.x) {
        const int s = seq_of(idx), y = 2 * (idx - s * half);
        const kcf *q = kspec_lds + s * sstride;
        const kcf c0 = q[kspec_lpos(PY, y)], c1 = q[kspec_lpos(PY, y + 1)];
        kcf *dst = tm_out ? W + (long long)(s >> 1) * nxl * ny + kspec_wt_index(-1, lg_rb, ny >> lg_rb, nxl, y >> lg_rb, (s & 1) ? jB : jA, y & ((1 << lg_rb) - 1)) : colat(s, y);
        *reinterpret_cast<float4 *>(dst) = make_float4(c0.x, c0.y, c1.x, c1.y);
    }
}

#ifndef KSPEC_LAB_MINIMAL      // (tools/spec_lab.hip builds the three plain 2-D kernels only)
// Column kernel for field counts whose 2*npair columns do not fit the LDS together (8192^2 x 3 fields needs 278 KB): two launches
// over blocks (column pair, FIELD pair = blockIdx.y), each holding only its own two columns.
//   phase 1: forward transform, the spectrum written to W in column layout (read from the tiles Wt when lg_rb >= 0, else in place)
//   phase 2: the block's two columns of the spectrum back into the LDS; the symbol stage takes the OTHER field pairs' values straight
//            from W (contiguous along y in both walking directions); inverse transform; result to Wout -- a different array, since
//            the blocks of the other field pairs still read W
// 2 + (npair + 2) passes over a pair's columns instead of 2: the price of keeping the solver at all for such sizes.
template <int NL>
__device__ __forceinline__ void kspec_cols_symbol_split(const KFFTPlan &PY, kcf *lds, int sstride, bool self, int kxA, int kxB, int p0, const kcf *__restrict__ W,
                                                        int nxl, int lg_pl, long long pstride, int jA, int jB, const int *__restrict__ posy,
                                                        const int *__restrict__ kyofpos, const float *__restrict__ lx, const float *__restrict__ ly, const KSpecSym &S)
{
    constexpr int F = NL + 1, npair = (F + 1) / 2;
    const int ny = PY.n, half = ny >> 1, plmask = (1 << lg_pl) - 1;
    const int nitem = self ? 2 * ny : ny;
    auto at = [&](int p, int c, int y) {
        return W + (long long)(y >> lg_pl) * pstride + (((long long)p * nxl + (c ? jB : jA)) << lg_pl) + (y & plmask);
    };
    for (int item = threadIdx.x; item < nitem; item += blockDim.x) {
        const int mpos = item & (ny - 1);
        const int ky = kyofpos[mpos];
        int ca = 0, cb = 1;
        if (self) { ca = cb = item >> PY.lg; if (ky > half) continue; }
        const int kym = (ny - ky) & (ny - 1);
        const int mposp = posy[kym];
        const int m = kspec_pad(mpos), mp = kspec_pad(mposp);
        const float L2 = lx[ca ? kxB : kxA] + ly[ky];
        kcf a[npair], b[npair];
#pragma unroll
        for (int p = 0; p < npair; p++) {
            if (p == p0) { a[p] = lds[ca * sstride + m]; b[p] = lds[cb * sstride + mp]; }
            else { a[p] = *at(p, ca, mpos); b[p] = *at(p, cb, mposp); }
        }
        kspec_symbol<NL>(S, L2, a, b);
        kcf ra = a[0], rb = b[0];
#pragma unroll
        for (int p = 1; p < npair; p++) if (p == p0) { ra = a[p]; rb = b[p]; }
        lds[ca * sstride + m] = ra; lds[cb * sstride + mp] = rb;
    }
}

// gridDim.z == 2: the two columns of a pair do not fit the LDS together either (columns of more than 8192 points): blockIdx.z picks ONE
// column, and the symbol stage of phase 2 takes the partner column's values from memory as well (contiguous, walked backwards)
template <int NL>
__device__ __forceinline__ void kspec_cols_symbol_split1(const KFFTPlan &PY, kcf *lds, bool self, int c0, int kxA, int kxB, int p0, const kcf *__restrict__ W,
                                                         int nxl, int lg_pl, long long pstride, int jA, int jB, const int *__restrict__ posy,
                                                         const int *__restrict__ kyofpos, const float *__restrict__ lx, const float *__restrict__ ly, const KSpecSym &S)
{
    constexpr int F = NL + 1, npair = (F + 1) / 2;
    const int ny = PY.n, half = ny >> 1, plmask = (1 << lg_pl) - 1;
    auto at = [&](int p, int c, int y) {
        return W + (long long)(y >> lg_pl) * pstride + (((long long)p * nxl + (c ? jB : jA)) << lg_pl) + (y & plmask);
    };
    const float lxc = lx[c0 ? kxB : kxA];
    for (int mpos = threadIdx.x; mpos < ny; mpos += blockDim.x) {
        const int ky = kyofpos[mpos];
        if (self && ky > half) continue;                           // a self-paired column holds both members of a pair: one item
        const int mposp = posy[ky ? ny - ky : 0];
        const int m = kspec_lpos(PY, mpos), mp = kspec_lpos(PY, mposp);
        const float L2 = lxc + ly[ky];
        kcf a[npair], b[npair];
#pragma unroll
        for (int p = 0; p < npair; p++) {
            if (self) { a[p] = p == p0 ? lds[m] : *at(p, c0, mpos); b[p] = p == p0 ? lds[mp] : *at(p, c0, mposp); }
            else if (c0 == 0) { a[p] = p == p0 ? lds[m] : *at(p, 0, mpos); b[p] = *at(p, 1, mposp); }
            else { a[p] = *at(p, 0, mposp); b[p] = p == p0 ? lds[m] : *at(p, 1, mpos); }
        }
        kspec_symbol<NL>(S, L2, a, b);
        kcf ra = a[0], rb = b[0];
#pragma unroll
        for (int p = 1; p < npair; p++) if (p == p0) { ra = a[p]; rb = b[p]; }
        if (self) { lds[m] = ra; lds[mp] = rb; }
        else lds[m] = c0 == 0 ? ra : rb;
    }
}

__global__ void __launch_bounds__(1024) k_spec_cols_split(int phase, KFFTPlan PY, int nxl, int lg_pl, long long pstride, kcf *__restrict__ W, const kcf *__restrict__ Wt, int lg_rb,
                                                         kcf *__restrict__ Wout, const kcf *__restrict__ tw, const int4 *__restrict__ pairtab, const int *__restrict__ posy,
                                                         const int *__restrict__ kyofpos, const float *__restrict__ lx, const float *__restrict__ ly, KSpecSym S)
{
    extern __shared__ kcf kspec_lds[];
    const int ny = PY.n, p0 = blockIdx.y;
    const bool one = gridDim.z == 2;                            // one column per block
    const int c0 = one ? blockIdx.z : 0, ncol = one ? 1 : 2;
    const int4 pt = pairtab[kspec_tile(blockIdx.x, gridDim.x)];
    const bool self = pt.w != 0;
    const int jA = pt.x, jB = pt.y, kxA = pt.z, kxB = self ? pt.w - 1 : pt.z;
    const int sstride = kspec_sstride(PY);
    const int half = ny >> 1, lg_half = PY.lg - 1;
    const int plmask = (1 << lg_pl) - 1;
    auto colat = [&](kcf *base, int c, int y) {                // y even: a float4 never straddles two pieces
        return base + (long long)(y >> lg_pl) * pstride + (((long long)p0 * nxl + (c ? jB : jA)) << lg_pl) + (y & plmask);
    };
    // fused edge stages as in k_spec_cols: phase 1 does the first forward stage on the values as they are gathered, phase 2 the last
    // inverse stage straight into the store (PY.flags bits 0 / 1)
    const bool edge = PY.m == 1 && PY.nstage > 0 && PY.radix[0] == 16;
    const int S0 = ny >> 4;
    if (phase == 1 && edge && (PY.flags & 1)) {
        kspec_stage0_fwd_from(kspec_lds, sstride, ncol, PY.lg, tw, [&](int cl, int i, int q) {
            const int c = c0 + cl, y = i + q * S0;
            return lg_rb >= 0 ? Wt[(long long)p0 * nxl * ny + kspec_wt_index(PY.lgw, lg_rb, ny >> lg_rb, nxl, y >> lg_rb, c ? jB : jA, y & ((1 << lg_rb) - 1))]
                              : *colat(W, c, y);
        });
        __syncthreads();
        kspec_fft_fwd(PY, kspec_lds, sstride, ncol, tw, 1);
        for (int idx = threadIdx.x; idx < ncol * half; idx += blockDim.x) {
            const int cl = idx >> lg_half, y = 2 * (idx & (half - 1));
            const kcf *q = kspec_lds + cl * sstride;
            const kcf c0v = q[kspec_pad(y)], c1v = q[kspec_pad(y + 1)];
            *reinterpret_cast<float4 *>(colat(W, c0 + cl, y)) = make_float4(c0v.x, c0v.y, c1v.x, c1v.y);
        }
        return;
    }
    // both phases start by filling the LDS with the block's column(s)
    for (int base = 0; base < ncol * half; base += 8 * blockDim.x) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            if (idx < ncol * half) {
                const int c = c0 + (idx >> lg_half), y = 2 * (idx & (half - 1));
                t[u] = (phase == 1 && lg_rb >= 0)
                           ? *reinterpret_cast<const float4 *>(Wt + (long long)p0 * nxl * ny + kspec_wt_index(PY.lgw, lg_rb, ny >> lg_rb, nxl, y >> lg_rb, c ? jB : jA, y & ((1 << lg_rb) - 1)))
                           : *reinterpret_cast<const float4 *>(colat(W, c, y));
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = base + u * blockDim.x + threadIdx.x;
            if (idx < ncol * half) {
                const int cl = idx >> lg_half, y = 2 * (idx & (half - 1));
                kcf *q = kspec_lds + cl * sstride;
                q[kspec_pad(y)] = make_float2(t[u].x, t[u].y);
                q[kspec_pad(y + 1)] = make_float2(t[u].z, t[u].w);
            }
        }
    }
    __syncthreads();
    kcf *dst = W;
    if (phase == 1) kspec_fft_fwd(PY, kspec_lds, sstride, ncol, tw);
    else {
        if (one) { KSPEC_NL_SWITCH(S.nlig, (kspec_cols_symbol_split1<NL>(PY, kspec_lds, self, c0, kxA, kxB, p0, W, nxl, lg_pl, pstride, jA, jB, posy, kyofpos, lx, ly, S))); }
        else { KSPEC_NL_SWITCH(S.nlig, (kspec_cols_symbol_split<NL>(PY, kspec_lds, sstride, self, kxA, kxB, p0, W, nxl, lg_pl, pstride, jA, jB, posy, kyofpos, lx, ly, S))); }
        __syncthreads();
        if (edge && (PY.flags & 2)) {
            kspec_fft_inv(PY, kspec_lds, sstride, ncol, tw, 1);
            kspec_stage0_inv_to(kspec_lds, sstride, ncol, PY.lg, tw, [&](int cl, int i, kcf (&x)[16]) {
#pragma unroll
                for (int q = 0; q < 16; q++) *colat(Wout, c0 + cl, i + q * S0) = x[q];
            });
            return;
        }
        kspec_fft_inv(PY, kspec_lds, sstride, ncol, tw);
        dst = Wout;
    }
    for (int idx = threadIdx.x; idx < ncol * half; idx += blockDim.x) {
        const int cl = idx >> lg_half, y = 2 * (idx & (half - 1));
        const kcf *q = kspec_lds + cl * sstride;
        const kcf c0v = q[kspec_pad(y)], c1v = q[kspec_pad(y + 1)];
        *reinterpret_cast<float4 *>(colat(dst, c0 + cl, y)) = make_float4(c0v.x, c0v.y, c1v.x, c1v.y);
    }
}
#endif // KSPEC_LAB_MINIMAL

#ifndef KSPEC_LAB_MINIMAL
// ---------------------------------------------------------------------------------------------
// 3-D (one rank): x rows as above over the nz*ny rows (tile-major store), then
//   k_spec3_y_fwd : one block per (pos_x, CZ consecutive z): gathers CZ columns over y from the tiles, DIF along y, stores
//                   W2[pair][pos_x][pos_y][z] (CZ consecutive z per store segment)
//   k_spec3_z<NL> : PB pairs of columns {(kx,ky), (-kx,-ky)} per block, contiguous in z: DIF, symbol, DIT inverse, in place
//   k_spec3_y_inv : reads CZ-z segments of W2, DIT inverse along y, stores W3[pair][pos_x][z*ny + y] (contiguous runs of ny)
// and the inverse x rows over W3.  HBM traffic 48 F N bytes per application (five kernels, four passes over the work arrays).
// ---------------------------------------------------------------------------------------------
// global -> LDS staging in batches of 8 items per thread: all loads of a batch are issued before the first LDS store (one memory
// latency per batch instead of one per item; the plain loop left it to the compiler, which kept them in order)
template <typename T, typename LD, typename ST>
__device__ __forceinline__ void kspec_stage_in(int total, LD ld, ST st)
{
    for (int base = 0; base < total; base += 8 * blockDim.x) {
        T t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int idx = base + u * blockDim.x + threadIdx.x; if (idx < total) t[u] = ld(idx); }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int idx = base + u * blockDim.x + threadIdx.x; if (idx < total) st(idx, t[u]); }
    }
}

__global__ void __launch_bounds__(1024) k_spec3_y_fwd(KFFTPlan PY, int nx, int nz, int lg_cz, int npair, int lg_rb, const kcf *__restrict__ Wt,
                                                      kcf *__restrict__ W2, const kcf *__restrict__ tw)
{
    extern __shared__ kcf kspec_lds[];
    const int ny = PY.n, lg_ny = PY.lg, cz = 1 << lg_cz, jx = blockIdx.y, z0 = blockIdx.x * cz;
    const int sstride = ny + (ny >> 4) + 1;
    const int rb = 1 << lg_rb;
    const long long ntiles = ((long long)ny * nz) >> lg_rb;
    const int nseq = npair * cz;
    // sequence s = p*cz + zc ; element y
    auto gather = [&](int s, int y) {
        const int p = s >> lg_cz, zc = s & (cz - 1);
        const long long R = (long long)(z0 + zc) * ny + y;
        return Wt[(((long long)p * ntiles + (R >> lg_rb)) * nx + jx) * rb + (R & (rb - 1))];
    };
    if (PY.m == 1 && PY.radix[0] == 16 && (PY.flags & 1)) {
        // first stage on the gathered values themselves (the 16 loads of a thread in flight together, no staging pass: as k_spec_cols)
        const int S0 = ny >> 4;
        kspec_stage0_fwd_from(kspec_lds, sstride, nseq, lg_ny, tw, [&](int s, int i, int q) { return gather(s, i + q * S0); });
        __syncthreads();
        kspec_fft_fwd(PY, kspec_lds, sstride, nseq, tw, 1);
    } else {
        kspec_stage_in<kcf>(nseq * ny, [&](int idx) { return gather(idx >> lg_ny, idx & (ny - 1)); },
                            [&](int idx, kcf v) { kspec_lds[(idx >> lg_ny) * sstride + kspec_pad(idx & (ny - 1))] = v; });
        __syncthreads();
        kspec_fft_fwd(PY, kspec_lds, sstride, nseq, tw);
    }
    for (int idx = threadIdx.x; idx < nseq * ny; idx += blockDim.x) {
        const int zc = idx & (cz - 1), rest = idx >> lg_cz;
        const int jy = rest & (ny - 1), p = rest >> lg_ny;
        W2[(((long long)p * nx + jx) * ny + jy) * nz + z0 + zc] = kspec_lds[(p * cz + zc) * sstride + kspec_pad(jy)];
    }
}

__global__ void __launch_bounds__(1024) k_spec3_y_inv(KFFTPlan PY, int nx, int nz, int lg_cz, int npair, const kcf *__restrict__ W2,
                                                      kcf *__restrict__ W3, const kcf *__restrict__ tw)
{
    extern __shared__ kcf kspec_lds[];
    const int ny = PY.n, lg_ny = PY.lg, cz = 1 << lg_cz, jx = blockIdx.y, z0 = blockIdx.x * cz;
    const int sstride = ny + (ny >> 4) + 1;
    const int nseq = npair * cz;
    kspec_stage_in<kcf>(nseq * ny, [&](int idx) {
        const int zc = idx & (cz - 1), rest = idx >> lg_cz;
        const int jy = rest & (ny - 1), p = rest >> lg_ny;
        return W2[(((long long)p * nx + jx) * ny + jy) * nz + z0 + zc];
    }, [&](int idx, kcf v) {
        const int zc = idx & (cz - 1), rest = idx >> lg_cz;
        const int jy = rest & (ny - 1), p = rest >> lg_ny;
        kspec_lds[(p * cz + zc) * sstride + kspec_pad(jy)] = v;
    });
    __syncthreads();
    const long long nrows = (long long)ny * nz;
    if (PY.m == 1 && PY.radix[0] == 16 && (PY.flags & 2)) {
        // last stage of the inverse straight into the store (runs of consecutive y per lane group)
        kspec_fft_inv(PY, kspec_lds, sstride, nseq, tw, 1);
        const int S0 = ny >> 4;
        kspec_stage0_inv_to(kspec_lds, sstride, nseq, lg_ny, tw, [&](int s, int i, kcf (&x)[16]) {
            const int p = s >> lg_cz, zc = s & (cz - 1);
            kcf *dst = W3 + ((long long)p * nx + jx) * nrows + (long long)(z0 + zc) * ny + i;
#pragma unroll
            for (int q = 0; q < 16; q++) dst[q * S0] = x[q];
        });
        return;
    }
    kspec_fft_inv(PY, kspec_lds, sstride, nseq, tw);
    for (int idx = threadIdx.x; idx < nseq * ny; idx += blockDim.x) {
        const int s = idx >> lg_ny, y = idx & (ny - 1);
        const int p = s >> lg_cz, zc = s & (cz - 1);
        W3[((long long)p * nx + jx) * nrows + (long long)(z0 + zc) * ny + y] = kspec_lds[s * sstride + kspec_pad(y)];
    }
}

template <int NL>
__device__ __forceinline__ void kspec3_z_symbol(const KFFTPlan &PZ, kcf *kspec_lds, int sstride, int ne, int e0, const int4 *__restrict__ pairtab, const int *__restrict__ posz,
                                                const int *__restrict__ kzofpos, const float *__restrict__ lx, const float *__restrict__ ly, const float *__restrict__ lz,
                                                const int2 *__restrict__ ztab, const KSpecSym &S)
{
    constexpr int F = NL + 1, npair = (F + 1) / 2;
    const int nz = PZ.n, half = nz >> 1;
    for (int item = threadIdx.x; item < ne * nz; item += blockDim.x) {
        const int slot = item >> PZ.lg, mpos = item & (nz - 1);
        const int4 pt = pairtab[e0 + slot];
        // ztab[pos] = (position of -kz, bits of lz[kz]), kz = the wavenumber at position pos: one coalesced load instead of the dependent
        // chain kzofpos[pos] -> posz[-kz], lz[kz] (as in k_spec_cols); only the four self-paired columns still need kz itself
        const int2 zt = ztab[mpos];
        const bool self = pt.w != 0;
        if (self && kzofpos[mpos] > half) continue;                // (A, kz) and (A, -kz) are one item
        const int m = kspec_pad(mpos), mp = kspec_pad(zt.x);
        const float L2 = lx[pt.z & 0xffff] + ly[pt.z >> 16] + __int_as_float(zt.y);
        const int sa = slot * npair * 2, cb = self ? 0 : 1;
        kcf a[npair], b[npair];
#pragma unroll
        for (int p = 0; p < npair; p++) { a[p] = kspec_lds[(sa + 2 * p) * sstride + m]; b[p] = kspec_lds[(sa + 2 * p + cb) * sstride + mp]; }
        kspec_symbol<NL>(S, L2, a, b);
#pragma unroll
        for (int p = 0; p < npair; p++) { kspec_lds[(sa + 2 * p) * sstride + m] = a[p]; kspec_lds[(sa + 2 * p + cb) * sstride + mp] = b[p]; }
    }
}

// pairtab[e] = (column A, column B, kx | ky << 16, self) with column = pos_x * ny + pos_y; self: A == B is its own partner
// Column storage as in k_spec_cols: a column consists of nz >> lg_pl pieces of 2^lg_pl elements, `pstride` elements apart (one piece
// per z-slab rank after the all-to-all; a single piece on one rank).
template <int NPAIR_T>      // as k_spec_cols: 1, 2 or 0 = run-time number of field pairs
__global__ void __launch_bounds__(1024) k_spec3_z(KFFTPlan PZ, int nent, int pb, long long ncol, int lg_pl, long long pstride, kcf *__restrict__ W2, const kcf *__restrict__ tw,
                                                  const int4 *__restrict__ pairtab, const int *__restrict__ posz, const int *__restrict__ kzofpos,
                                                  const float *__restrict__ lx, const float *__restrict__ ly, const float *__restrict__ lz, const int2 *__restrict__ ztab, KSpecSym S)
{
    extern __shared__ kcf kspec_lds[];
    const int npair = NPAIR_T ? NPAIR_T : (S.nlig + 2) / 2;
    const int nz = PZ.n;
    const int sstride = nz + (nz >> 4) + 1;
    const int e0 = blockIdx.x * pb;
    const int ne = min(pb, nent - e0);
    const int nseq = 2 * npair * ne;                              // sequence s = (slot*npair + p)*2 + c
    const int half = nz >> 1, lg_half = PZ.lg - 1;
    const int plmask = (1 << lg_pl) - 1;
    auto colptr = [&](int s) {                                     // start of the column's FIRST piece
        const int c = s & 1, p = (s >> 1) % npair, slot = (s >> 1) / npair;
        const int4 pt = pairtab[e0 + slot];
        return W2 + (((long long)p * ncol + (c ? pt.y : pt.x)) << lg_pl);
    };
    auto zoff = [&](int z) { return (long long)(z >> lg_pl) * pstride + (z & plmask); };      // z even: a float4 never straddles two pieces
    const bool edge_in = PZ.m == 1 && PZ.radix[0] == 16 && (PZ.flags & 1), edge_out = PZ.m == 1 && PZ.radix[0] == 16 && (PZ.flags & 2);
    const int S0 = nz >> 4;
    if (edge_in) {
        kspec_stage0_fwd_from(kspec_lds, sstride, nseq, PZ.lg, tw, [&](int s, int i, int q) { return colptr(s)[zoff(i + q * S0)]; });
        __syncthreads();
        kspec_fft_fwd(PZ, kspec_lds, sstride, nseq, tw, 1);
    } else {
    kspec_stage_in<float4>(nseq * half, [&](int idx) {
        const int s = idx >> lg_half, z = 2 * (idx & (half - 1));
        return *reinterpret_cast<const float4 *>(colptr(s) + zoff(z));
    }, [&](int idx, float4 t) {
        const int s = idx >> lg_half, z = 2 * (idx & (half - 1));
        kcf *q = kspec_lds + s * sstride;
        q[kspec_pad(z)] = make_float2(t.x, t.y);
        q[kspec_pad(z + 1)] = make_float2(t.z, t.w);
    });
    __syncthreads();
    kspec_fft_fwd(PZ, kspec_lds, sstride, nseq, tw);
    }
    if (NPAIR_T == 1) kspec3_z_symbol<1>(PZ, kspec_lds, sstride, ne, e0, pairtab, posz, kzofpos, lx, ly, lz, ztab, S);
    else if (NPAIR_T == 2) {
        if (S.nlig == 2) kspec3_z_symbol<2>(PZ, kspec_lds, sstride, ne, e0, pairtab, posz, kzofpos, lx, ly, lz, ztab, S);
        else kspec3_z_symbol<3>(PZ, kspec_lds, sstride, ne, e0, pairtab, posz, kzofpos, lx, ly, lz, ztab, S);
    } else { KSPEC_NL_SWITCH(S.nlig, (kspec3_z_symbol<NL>(PZ, kspec_lds, sstride, ne, e0, pairtab, posz, kzofpos, lx, ly, lz, ztab, S))); }
    __syncthreads();
    if (edge_out) {
        kspec_fft_inv(PZ, kspec_lds, sstride, nseq, tw, 1);
        kspec_stage0_inv_to(kspec_lds, sstride, nseq, PZ.lg, tw, [&](int s, int i, kcf (&x)[16]) {
            if ((s & 1) && pairtab[e0 + (s >> 1) / npair].w) return;       // the B slot of a self column is a copy
            kcf *dst = colptr(s);
#pragma unroll
            for (int q = 0; q < 16; q++) dst[zoff(i + q * S0)] = x[q];
        });
        return;
    }
    kspec_fft_inv(PZ, kspec_lds, sstride, nseq, tw);
    for (int idx = threadIdx.x; idx < nseq * half; idx += blockDim.x) {
        const int s = idx >> lg_half, z = 2 * (idx & (half - 1));
        if ((s & 1) && pairtab[e0 + (s >> 1) / npair].w) continue;      // the B slot of a self column is a copy
        const kcf *q = kspec_lds + s * sstride;
        const kcf c0 = q[kspec_pad(z)], c1 = q[kspec_pad(z + 1)];
        *reinterpret_cast<float4 *>(colptr(s) + zoff(z)) = make_float4(c0.x, c0.y, c1.x, c1.y);
    }
}

#endif // KSPEC_LAB_MINIMAL

// grid means of rho*G_rho and rho*G_Ul over the frozen coefficient planes C = [rho, G, G_rho, G_U1..] (once per step)
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_spec_means(KGeom G, const double *__restrict__ C, double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE][NL + 1];
    double acc[NL + 1];
#pragma unroll
    for (int i = 0; i <= NL; i++) acc[i] = 0.0;
    const long long off = (long long)G.ng * G.inner;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride) {
        const double rho = C[off + p];
#pragma unroll
        for (int i = 0; i <= NL; i++) acc[i] += rho * C[(long long)(2 + i) * G.plane + off + p];
    }
    const int lane = threadIdx.x & (KSFD_WAVE - 1), wv = threadIdx.x / KSFD_WAVE;
#pragma unroll
    for (int i = 0; i <= NL; i++) {
        const double s = ksfd_wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x <= NL) {
        double s = 0.0;
        for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) s += red[q][threadIdx.x];
        part[(long long)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}
