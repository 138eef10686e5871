// Stencil kernels: RHS (Derivatives.dfdt), Jacobian action, CFL velocity.
//
// Two families:
//  * "generic" (1-D / 2-D / 3-D, any size): a pointwise pass writes G (and dG) over the local
//    slab including ghost units, then a one-thread-per-point pass applies the 4th-order star
//    (5/9/13 points) with direct, L1/L2-served neighbour loads.  Simple; used for 1-D, 3-D and
//    odd sizes, and as the in-library cross-check of the fused kernels.
//  * "fused 2-D" (the 4096^2 headline path): one pass, no G array.  Each wave64 owns a strip of
//    128 columns (2 per lane, 16-B loads, 124 of them outputs) and marches down a segment of rows
//    holding a 5-row register window of rho, G, U_l; G is evaluated once per loaded point; the
//    x-neighbours come from wave shuffles, the y-neighbours from the window.  Compulsory HBM
//    traffic only: read F planes, write F planes (+4 halo rows per segment, L2 hits).
#pragma once
#include <type_traits>
#include "pointwise.hip.h"

struct KSrc {
    const double *p[KSFD_MAXL + 1];   // dense nloc-sized source planes (no ghosts) or NULL
};

// Stage-vector algebra of the Rosenbrock step folded into the RHS kernel: the input is u + sum_j ain[j]*yin[j] (formed on
// load, before the clamp), the output gets sum_j aout[j]*yout[j] added at the store.  Vectors in the ghosted layout.
struct KComb {
    int nin, nout;
    const double *yin[3], *yout[3];
    double ain[3], aout[3];
};

// Inner products of what the RHS kernel stores with up to two given vectors (ghosted layout), from the same store epilogue as ||out||^2:
// wave w leaves <out, v[q]> in normpart[(1 + q) * stride + w].  The stage guesses of ksfd_step need <b_i, b_j> of the new right-hand
// side with the two before it; as a pass of their own they cost a read of all three vectors.
struct KDots {
    int n;
    long long stride;
    const double *v[2];
};

// Smoother algebra of the multigrid preconditioner folded into the Jacobian-action epilogue (modes 5 and 6):
//   5:  r = yadd - A v  -> out ;  d = scale * Dinv r -> out2                (residual + first Chebyshev direction)
//   6:  x = (x_has_d ? d : x + d) + c1*d + c2 * Dinv (rr - A d),  v = d     (k_cheb_last without the A d round trip; x_has_d: first sweep
//       from a zero guess, x would be d itself and is neither written beforehand nor read)
// Dinv: F*F planes of the inverse point-block diagonal (row-major), all vectors in the level's ghosted layout.
template <typename TS = double>
struct KSmoothT {
    const float *dinv;
    const TS *rr;
    TS *x, *out2;
    double c1, c2, scale;
    int x_has_d;
    double *x64;         // mode 6: the updated x goes HERE in fp64 instead of back to x (last sweep of an fp32 V cycle: the result in the caller's precision)
};
typedef KSmoothT<double> KSmooth;

// ---------------------------------------------------------------------------------------------
// generic family
// ---------------------------------------------------------------------------------------------

// G (and dG = G_rho*v_rho + sum G_Ul*v_Ul) at every point of the slab incl. ghosts.
template <int NL, bool DERIV>
__global__ void __launch_bounds__(KSFD_BLOCK) k_gfield(KGeom G, KPhys P, const double *__restrict__ u,
                                                       const double *__restrict__ v, double *__restrict__ Gout,
                                                       double *__restrict__ dGout)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < G.plane; e += stride) {
        double rho = ksfd_clamp(u[e], P.rhomin);
        double U[NL], GU[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) U[l] = ksfd_clamp(u[(long long)(l + 1) * G.plane + e], P.Umin);
        double g, gr = 0.0;
        ksfd_G<NL, DERIV>(P, rho, U, g, gr, GU);
        Gout[e] = g;
        if (DERIV) {
            double d = gr * v[e];
#pragma unroll
            for (int l = 0; l < NL; l++) d += GU[l] * v[(long long)(l + 1) * G.plane + e];
            dGout[e] = d;
        }
    }
}

// neighbour addressing for the generic kernels: offsets (relative to a field plane base) of the
// point and its +-1, +-2 neighbours along axis a.
struct KNbr {
    long long c, m2, m1, p1, p2;
};
__device__ __forceinline__ long long ksfd_wrap(long long i, long long n)
{
    i %= n;
    return i < 0 ? i + n : i;
}
__device__ __forceinline__ void ksfd_decode(const KGeom &G, long long p, long long &i, long long &j, long long &k)
{
    i = p % G.nx;
    long long r = p / G.nx;
    j = (G.dim >= 2) ? r % G.ny : 0;
    k = (G.dim >= 3) ? r / G.ny : 0;
}
__device__ __forceinline__ KNbr ksfd_nbr(const KGeom &G, int a, long long i, long long j, long long k)
{
    // coordinates along the slow axis are local (0..sloc-1); ghosts sit at -2,-1 and sloc,sloc+1
    long long idx[3] = { i, j, k };
    long long ext[3] = { G.nx, G.ny, G.nz };
    long long str[3] = { 1, G.nx, G.nx * G.ny };
    const int slow = G.dim - 1;
    long long base = 0;
    for (int d = 0; d < G.dim; d++)
        if (d != a) base += (d == slow ? idx[d] + G.ng : idx[d]) * str[d];
    KNbr n;
    long long q[5];
#pragma unroll
    for (int m = -2; m <= 2; m++) {
        long long x = idx[a] + m;
        if (a == slow) x = G.wrap_slow ? ksfd_wrap(x, ext[a]) : x + G.ng;
        else x = ksfd_wrap(x, ext[a]);
        q[m + 2] = base + x * str[a];
    }
    n.m2 = q[0]; n.m1 = q[1]; n.c = q[2]; n.p1 = q[3]; n.p2 = q[4];
    return n;
}

// Derivatives.dfdt (KSFD/ksfdsym.py:902-940) given the G plane:
//   out_rho = sum_a D1a(rho) D1a(G) + rho sum_a D2a(G) + src      (:531-571, :763-812)
//   out_Ul  = -gamma U + s rho + D sum_a D2a(U) + src             (:583-613)
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_rhs_generic(KGeom G, KPhys P, const double *__restrict__ u,
                                                            const double *__restrict__ Gb, KSrc src,
                                                            double *__restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride) {
        long long i, j, k;
        ksfd_decode(G, p, i, j, k);
        double acc = 0.0, lapG = 0.0, rho0 = 0.0;
        double lapU[NL], U0[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) lapU[l] = 0.0;
        for (int a = 0; a < G.dim; a++) {
            KNbr n = ksfd_nbr(G, a, i, j, k);
            double r_m2 = ksfd_clamp(u[n.m2], P.rhomin), r_m1 = ksfd_clamp(u[n.m1], P.rhomin),
                   r_p1 = ksfd_clamp(u[n.p1], P.rhomin), r_p2 = ksfd_clamp(u[n.p2], P.rhomin);
            rho0 = ksfd_clamp(u[n.c], P.rhomin);
            double g_m2 = Gb[n.m2], g_m1 = Gb[n.m1], g_0 = Gb[n.c], g_p1 = Gb[n.p1], g_p2 = Gb[n.p2];
            acc += (KSFD_D1(r_m2, r_m1, r_p1, r_p2) * P.inv_h[a]) * (KSFD_D1(g_m2, g_m1, g_p1, g_p2) * P.inv_h[a]);
            lapG += KSFD_D2(g_m2, g_m1, g_0, g_p1, g_p2) * P.inv_h2[a];
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const double *U = u + (long long)(l + 1) * G.plane;
                double c0 = ksfd_clamp(U[n.c], P.Umin);
                U0[l] = c0;
                lapU[l] += KSFD_D2(ksfd_clamp(U[n.m2], P.Umin), ksfd_clamp(U[n.m1], P.Umin), c0,
                                   ksfd_clamp(U[n.p1], P.Umin), ksfd_clamp(U[n.p2], P.Umin)) * P.inv_h2[a];
            }
        }
        const long long o = (long long)G.ng * G.inner + p;
        double r = acc + rho0 * lapG;
        if (src.p[0]) r += src.p[0][p];
        out[o] = r;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            double w = -P.lig_gamma[l] * U0[l] + P.lig_s[l] * rho0 + P.lig_D[l] * lapU[l];
            if (src.p[l + 1]) w += src.p[l + 1][p];
            out[(long long)(l + 1) * G.plane + o] = w;
        }
    }
}

// Jacobian action at the (clamped) state u on v, given G and dG planes.  mode 0: out = J v,
// mode 1: out = shift*v - J v  (the IJacobian of KSFD/ksfdts.py:598-640 applied matrix-free).
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_jvp_generic(KGeom G, KPhys P, const double *__restrict__ u,
                                                            const double *__restrict__ v,
                                                            const double *__restrict__ Gb,
                                                            const double *__restrict__ dGb, int mode, double shift,
                                                            double *__restrict__ out, const double *__restrict__ yadd = nullptr,
                                                            double alpha = 0.0, double beta = 0.0, KSmooth sm = KSmooth{})
{
    // mode 0: J v | 1: shift v - J v | 2: yadd - (shift v - J v) | 3: alpha*yadd + beta*(shift v - J v)
    // 5 / 6: smoother algebra of the multigrid preconditioner in the epilogue (KSmooth)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride) {
        long long i, j, k;
        ksfd_decode(G, p, i, j, k);
        double acc = 0.0, lapG = 0.0, lapdG = 0.0, rho0 = 0.0, v0 = 0.0;
        double lapV[NL], V0[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) lapV[l] = 0.0;
        for (int a = 0; a < G.dim; a++) {
            KNbr n = ksfd_nbr(G, a, i, j, k);
            double r_m2 = ksfd_clamp(u[n.m2], P.rhomin), r_m1 = ksfd_clamp(u[n.m1], P.rhomin),
                   r_p1 = ksfd_clamp(u[n.p1], P.rhomin), r_p2 = ksfd_clamp(u[n.p2], P.rhomin);
            rho0 = ksfd_clamp(u[n.c], P.rhomin);
            v0 = v[n.c];
            double d1r = KSFD_D1(r_m2, r_m1, r_p1, r_p2) * P.inv_h[a];
            double d1v = KSFD_D1(v[n.m2], v[n.m1], v[n.p1], v[n.p2]) * P.inv_h[a];
            double d1g = KSFD_D1(Gb[n.m2], Gb[n.m1], Gb[n.p1], Gb[n.p2]) * P.inv_h[a];
            double d1e = KSFD_D1(dGb[n.m2], dGb[n.m1], dGb[n.p1], dGb[n.p2]) * P.inv_h[a];
            acc += d1v * d1g + d1r * d1e;
            lapG += KSFD_D2(Gb[n.m2], Gb[n.m1], Gb[n.c], Gb[n.p1], Gb[n.p2]) * P.inv_h2[a];
            lapdG += KSFD_D2(dGb[n.m2], dGb[n.m1], dGb[n.c], dGb[n.p1], dGb[n.p2]) * P.inv_h2[a];
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const double *V = v + (long long)(l + 1) * G.plane;
                V0[l] = V[n.c];
                lapV[l] += KSFD_D2(V[n.m2], V[n.m1], V[n.c], V[n.p1], V[n.p2]) * P.inv_h2[a];
            }
        }
        const long long o = (long long)G.ng * G.inner + p;
        double jr = acc + v0 * lapG + rho0 * lapdG;
        if (mode >= 5) {
            double q[NL + 1], dc[NL + 1];
            q[0] = shift * v0 - jr; dc[0] = v0;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const double ju = -P.lig_gamma[l] * V0[l] + P.lig_s[l] * v0 + P.lig_D[l] * lapV[l];
                q[l + 1] = shift * V0[l] - ju; dc[l + 1] = V0[l];
            }
#pragma unroll
            for (int c = 0; c <= NL; c++) q[c] = (mode == 5 ? yadd[(long long)c * G.plane + o] : sm.rr[(long long)c * G.plane + o]) - q[c];
#pragma unroll
            for (int a = 0; a <= NL; a++) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c <= NL; c++) s += sm.dinv[(long long)(a * (NL + 1) + c) * G.plane + o] * q[c];
                const long long oa = (long long)a * G.plane + o;
                if (mode == 5) { out[oa] = q[a]; sm.out2[oa] = sm.scale * s; }
                else sm.x[oa] = (sm.x_has_d ? dc[a] : sm.x[oa] + dc[a]) + sm.c1 * dc[a] + sm.c2 * s;       // x_has_d: x == d == v, not read
            }
            continue;
        }
        double o0 = mode ? shift * v0 - jr : jr;
        if (mode == 2) o0 = yadd[o] - o0;
        else if (mode == 3) o0 = alpha * yadd[o] + beta * o0;
        else if (mode == 4) o0 = alpha * v0 + beta * o0;
        out[o] = o0;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            double ju = -P.lig_gamma[l] * V0[l] + P.lig_s[l] * v0 + P.lig_D[l] * lapV[l];
            double ol = mode ? shift * V0[l] - ju : ju;
            if (mode == 2) ol = yadd[(long long)(l + 1) * G.plane + o] - ol;
            else if (mode == 3) ol = alpha * yadd[(long long)(l + 1) * G.plane + o] + beta * ol;
            else if (mode == 4) ol = alpha * V0[l] + beta * ol;
            out[(long long)(l + 1) * G.plane + o] = ol;
        }
    }
}

// Synthetic start values on the device (ksfdsolver2.py:580-639): rho = rho0 + smoothstep
// interpolation of the coarse samples z (KSFD/ksfdrandom.py:116, 194-214: weight f(x) = 2x^3 - 3x^2 + 1 on
// |dx|/h_coarse < 1 per axis, periodic; f(x) + f(1-x) = 1, so each fine point is a convex combination of its 2^dim coarse
// neighbours), U_l = rho * s_l / gamma_l.  z: global coarse grid, x fastest.  Same summation order as the oracle.
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_random_start(KGeom G, KPhys P, long long gslow, long long slow0,
                                                             long long nc0, long long nc1, long long nc2,
                                                             const double *__restrict__ z, double rho0, double *__restrict__ u)
{
    const int slow = G.dim - 1;
    const long long nc[3] = { nc0, nc1, nc2 };
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride) {
        long long idx[3];
        ksfd_decode(G, p, idx[0], idx[1], idx[2]);
        idx[slow] += slow0;
        const long long gext[3] = { slow == 0 ? gslow : G.nx, slow == 1 ? gslow : G.ny, slow == 2 ? gslow : G.nz };
        long long lo[3] = { 0, 0, 0 };
        double wlo[3] = { 1.0, 1.0, 1.0 }, whi[3] = { 0.0, 0.0, 0.0 };
        for (int a = 0; a < G.dim; a++) {
            const double xc = (double)idx[a] * (double)nc[a] / (double)gext[a];
            const double fl = floor(xc), fr = xc - fl, g1 = 1.0 - fr;
            lo[a] = (long long)fl;
            wlo[a] = 2.0 * fr * fr * fr - 3.0 * fr * fr + 1.0;
            whi[a] = fr == 0.0 ? 0.0 : 2.0 * g1 * g1 * g1 - 3.0 * g1 * g1 + 1.0;
        }
        double sum = 0.0;
        for (int dz = 0; dz < (G.dim > 2 ? 2 : 1); dz++)
            for (int dy = 0; dy < (G.dim > 1 ? 2 : 1); dy++)
                for (int dx = 0; dx < 2; dx++) {
                    const double w = (dx ? whi[0] : wlo[0]) * (G.dim > 1 ? (dy ? whi[1] : wlo[1]) : 1.0) *
                                     (G.dim > 2 ? (dz ? whi[2] : wlo[2]) : 1.0);
                    if (w == 0.0) continue;
                    const long long ci = ksfd_wrap(lo[0] + dx, nc[0]);
                    const long long cj = G.dim > 1 ? ksfd_wrap(lo[1] + dy, nc[1]) : 0;
                    const long long ck = G.dim > 2 ? ksfd_wrap(lo[2] + dz, nc[2]) : 0;
                    sum += w * z[ci + nc[0] * (cj + nc[1] * ck)];
                }
        const long long o = (long long)G.ng * G.inner + p;
        const double rho = rho0 + sum;
        u[o] = rho;
#pragma unroll
        for (int l = 0; l < NL; l++) u[(long long)(l + 1) * G.plane + o] = rho * (P.lig_s[l] / P.lig_gamma[l]);
    }
}

// Assembled Jacobian export: what Derivatives.Jacobian + ksfdMat.setValuesJacobian put
// into the PETSc AIJ matrix (KSFD/ksfdsym.py:814-886, cython/ksfdMat/ksfdMat.pyx:55-180), as CSR values + GLOBAL column
// indices in the reference's Vec ordering (unknown = F*point + dof, point x-fastest).  One thread per owned point,
// coefficients from the frozen planes C = [rho, G, G_rho, G_U1..] (k_jcoef, ghosts filled).  Entry order per point:
// rho row: for each dof, centre then axis by axis m = -2,-1,+1,+2;  U_l row: rho centre, U_l centre, the axes.
// Not a hot path: the entry stores are strided.
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_jac_csr(KGeom G, KPhys P, const double *__restrict__ C,
                                                        long long gslow, long long slow0,
                                                        long long *__restrict__ col, double *__restrict__ val)
{
    constexpr int F = NL + 1;
    const int npts = 4 * G.dim + 1;
    const long long per = (long long)F * npts + (long long)NL * (npts + 1);
    const int slow = G.dim - 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride) {
        long long i, j, k;
        ksfd_decode(G, p, i, j, k);
        long long gi[3] = { i, j, k };
        gi[slow] += slow0;
        const long long gext[3] = { slow == 0 ? gslow : G.nx, slow == 1 ? gslow : G.ny, slow == 2 ? gslow : G.nz };
        const long long gstr[3] = { 1, gext[0], gext[0] * gext[1] };
        const long long gp = gi[0] + gi[1] * gstr[1] + gi[2] * gstr[2];
        double d1g[3], d1r[3], lapG = 0.0, w2c = 0.0;
        long long nl_[3][4], ng_[3][4], c0 = 0;
        for (int a = 0; a < G.dim; a++) {
            KNbr n = ksfd_nbr(G, a, i, j, k);
            c0 = n.c;
            const double *Gb = C + G.plane;
            d1g[a] = KSFD_D1(Gb[n.m2], Gb[n.m1], Gb[n.p1], Gb[n.p2]) * P.inv_h[a];
            d1r[a] = KSFD_D1(C[n.m2], C[n.m1], C[n.p1], C[n.p2]) * P.inv_h[a];
            lapG += KSFD_D2(Gb[n.m2], Gb[n.m1], Gb[n.c], Gb[n.p1], Gb[n.p2]) * P.inv_h2[a];
            w2c += -2.5 * P.inv_h2[a];
            nl_[a][0] = n.m2; nl_[a][1] = n.m1; nl_[a][2] = n.p1; nl_[a][3] = n.p2;
            const int ms[4] = { -2, -1, 1, 2 };
            for (int m = 0; m < 4; m++)
                ng_[a][m] = gp + (ksfd_wrap(gi[a] + ms[m], gext[a]) - gi[a]) * gstr[a];
        }
        const double rho0 = C[c0];
        const double w1[4] = { 1.0 / 12.0, -2.0 / 3.0, 2.0 / 3.0, -1.0 / 12.0 };
        const double w2[4] = { -1.0 / 12.0, 4.0 / 3.0, 4.0 / 3.0, -1.0 / 12.0 };
        long long e = p * per;
        for (int dof = 0; dof < F; dof++) {
            const double *Gx = C + (long long)(2 + dof) * G.plane;       // G_rho, G_U1, ...
            col[e] = gp * F + dof;
            val[e++] = (dof == 0 ? lapG : 0.0) + rho0 * w2c * Gx[c0];
            for (int a = 0; a < G.dim; a++)
                for (int m = 0; m < 4; m++) {
                    const double a1 = w1[m] * P.inv_h[a], a2 = w2[m] * P.inv_h2[a];
                    col[e] = ng_[a][m] * F + dof;
                    val[e++] = (dof == 0 ? a1 * d1g[a] : 0.0) + (a1 * d1r[a] + rho0 * a2) * Gx[nl_[a][m]];
                }
        }
        for (int l = 0; l < NL; l++) {
            col[e] = gp * F;
            val[e++] = P.lig_s[l];
            col[e] = gp * F + l + 1;
            val[e++] = -P.lig_gamma[l] + P.lig_D[l] * w2c;
            for (int a = 0; a < G.dim; a++)
                for (int m = 0; m < 4; m++) {
                    col[e] = ng_[a][m] * F + l + 1;
                    val[e++] = P.lig_D[l] * w2[m] * P.inv_h2[a];
                }
        }
    }
}

// Derivatives.velocity (KSFD/ksfdsym.py:1158-1209): v_a = D1a(G).  Writes dense dim*nloc planes when
// vel != NULL and per-block per-axis max|v_a| partials (CFL_step, KSFD/ksfdts.py:302-319) when part != NULL.
__global__ void __launch_bounds__(KSFD_BLOCK) k_velocity(KGeom G, KPhys P, const double *__restrict__ Gb,
                                                         double *__restrict__ vel, double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE][3];
    double mx[3] = { 0.0, 0.0, 0.0 };
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride) {
        long long i, j, k;
        ksfd_decode(G, p, i, j, k);
        for (int a = 0; a < G.dim; a++) {
            KNbr n = ksfd_nbr(G, a, i, j, k);
            double d = KSFD_D1(Gb[n.m2], Gb[n.m1], Gb[n.p1], Gb[n.p2]) * P.inv_h[a];
            if (vel) vel[(long long)a * G.nloc + p] = d;
            mx[a] = fmax(mx[a], fabs(d));
        }
    }
    if (part) {
        for (int a = 0; a < 3; a++) {
            double m = ksfd_wave_max(mx[a]);
            if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) red[threadIdx.x / KSFD_WAVE][a] = m;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double m = 0.0;
            for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) m = fmax(m, red[q][threadIdx.x]);
            part[(long long)threadIdx.x * gridDim.x + blockIdx.x] = m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused 2-D family
// ---------------------------------------------------------------------------------------------
#define KSFD_STRIP_OUT 124          // output columns per wave strip (128 loaded, 2 halo each side)

struct KStrips {
    int nstrips;      // ceil(nx / 124), strips balanced to even widths
    int nseg;         // row segments handled by THIS launch
    int yseg;
    int nblocks;      // launch grid (multiple of 8 for the XCD remap)
    int seg0;         // first segment and stride between the segments of this launch: the interior launch uses
    int seg_stride;   // (1, 1), the boundary launch (0, nseg_total-1) = the two segments that read ghost rows
};

// blockIdx -> logical block so that each XCD (blocks are dealt round-robin over the 8 XCDs)
// works on one contiguous band of row segments: neighbouring strips/segments then share an L2.
__device__ __forceinline__ int ksfd_xcd_remap(int b, int nblocks)
{
    const int per = nblocks >> 3;
    return (b & 7) * per + (b >> 3);
}

__device__ __forceinline__ long long ksfd_rowoff(const KGeom &G, long long r)
{
    // r may be -2..sloc+1
    if (G.wrap_slow) {                   // no 64-bit modulo here: it costs ~100 instructions per row and wave
        if (r < 0) r += G.sloc;
        else if (r >= G.sloc) r -= G.sloc;
        return r * G.nx;
    }
    return (r + G.ng) * G.nx;
}

__device__ __forceinline__ double2 ksfd_ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
__device__ __forceinline__ void ksfd_st2(double *p, double a, double b)
{
    *reinterpret_cast<double2 *>(p) = make_double2(a, b);
}
// fp32 STORAGE (arithmetic stays fp64): used only inside the polynomial preconditioner, see poly_apply
__device__ __forceinline__ double2 ksfd_ld2(const float *p)
{
    const float2 t = *reinterpret_cast<const float2 *>(p);
    return make_double2((double)t.x, (double)t.y);
}
__device__ __forceinline__ void ksfd_st2(float *p, double a, double b)
{
    *reinterpret_cast<float2 *>(p) = make_float2((float)a, (float)b);
}

// x-neighbours of a 2-column pair through wave shuffles.  For the pair (a0,a1) at columns (c,c+1):
// left lane holds (c-2,c-1), right lane (c+2,c+3).
struct KX {
    double l0, l1, r0, r1;
};
// Neighbour-lane moves as DPP wavefront shifts (GFX9 dpp_ctrl wave_shr:1 = 0x138, wave_shl:1 = 0x130): a v_mov_dpp per
// 32-bit half instead of a ds_bpermute through the LDS crossbar.  Same result as __shfl_up/__shfl_down by one lane
// (lane 0 / lane 63 keep their own value: old == src, bound_ctrl off).
__device__ __forceinline__ double ksfd_lane_from_left(double a)
{
    int lo = __double2loint(a), hi = __double2hiint(a);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ksfd_lane_from_right(double a)
{
    int lo = __double2loint(a), hi = __double2hiint(a);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ KX ksfd_xnb(double a0, double a1)
{
    KX x;
    x.l0 = ksfd_lane_from_left(a0);
    x.l1 = ksfd_lane_from_left(a1);
    x.r0 = ksfd_lane_from_right(a0);
    x.r1 = ksfd_lane_from_right(a1);
    return x;
}
// h*d/dx and h^2*d2/dx2 for both points of the pair
__device__ __forceinline__ void ksfd_dx(double a0, double a1, const KX &x, double &d1_0, double &d1_1)
{
    d1_0 = KSFD_D1(x.l0, x.l1, a1, x.r0);
    d1_1 = KSFD_D1(x.l1, a0, x.r0, x.r1);
}
__device__ __forceinline__ void ksfd_dxx(double a0, double a1, const KX &x, double &d2_0, double &d2_1)
{
    d2_0 = KSFD_D2(x.l0, x.l1, a0, a1, x.r0);
    d2_1 = KSFD_D2(x.l1, a0, a1, x.r0, x.r1);
}
// The same for a vector STORED in fp32 (the multigrid cycle's level vectors): the window keeps the stored values -- half the registers,
// one DPP move per neighbour instead of two -- and every difference is formed in fp64.
struct KXf {
    float l0, l1, r0, r1;
};
__device__ __forceinline__ float ksfd_lane_from_left(float a)
{
    int v = __float_as_int(a);
    v = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
    return __int_as_float(v);
}
__device__ __forceinline__ float ksfd_lane_from_right(float a)
{
    int v = __float_as_int(a);
    v = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
    return __int_as_float(v);
}
__device__ __forceinline__ KXf ksfd_xnb(float a0, float a1)
{
    KXf x;
    x.l0 = ksfd_lane_from_left(a0);
    x.l1 = ksfd_lane_from_left(a1);
    x.r0 = ksfd_lane_from_right(a0);
    x.r1 = ksfd_lane_from_right(a1);
    return x;
}
__device__ __forceinline__ void ksfd_dx(float a0, float a1, const KXf &x, double &d1_0, double &d1_1)
{
    d1_0 = KSFD_D1((double)x.l0, (double)x.l1, (double)a1, (double)x.r0);
    d1_1 = KSFD_D1((double)x.l1, (double)a0, (double)x.r0, (double)x.r1);
}
__device__ __forceinline__ void ksfd_dxx(float a0, float a1, const KXf &x, double &d2_0, double &d2_1)
{
    d2_0 = KSFD_D2((double)x.l0, (double)x.l1, (double)a0, (double)a1, (double)x.r0);
    d2_1 = KSFD_D2((double)x.l1, (double)a0, (double)a1, (double)x.r0, (double)x.r1);
}
// a pair in its storage type
__device__ __forceinline__ double2 ksfd_ldw(const double *p) { return *reinterpret_cast<const double2 *>(p); }
__device__ __forceinline__ float2 ksfd_ldw(const float *p) { return *reinterpret_cast<const float2 *>(p); }

struct KWaveJob {
    long long wid;      // logical wave id (strip + nstrips * segment slot of this launch)
    long long c0;       // first of this lane's two columns (wrapped)
    long long r0, r1;   // row segment [r0, r1)
    bool store;
    bool valid;
    bool up;            // march direction.  Even segments go up, odd ones down, so two neighbouring segments read
                        // their shared halo rows at the same time (one HBM fetch, one L2 hit) instead of a wave
                        // lifetime apart.  All first y-derivatives enter as products of pairs: the sign cancels.
    __device__ __forceinline__ long long row(long long q) const { return up ? q : (r0 + r1 - 1 - q); }
};
__device__ __forceinline__ KWaveJob ksfd_wave_job(const KGeom &G, const KStrips &S)
{
    KWaveJob J;
    const int lane = threadIdx.x & (KSFD_WAVE - 1);
    const long long wid = (long long)ksfd_xcd_remap(blockIdx.x, S.nblocks) * (KSFD_BLOCK / KSFD_WAVE) + (threadIdx.x >> 6);
    J.wid = wid;
    J.valid = wid < (long long)S.nstrips * S.nseg;
    const int strip = (int)(wid % S.nstrips);
    const long long seg = S.seg0 + (wid / S.nstrips) * S.seg_stride;
    const long long half = G.nx >> 1;
    const long long xs = 2 * ((long long)strip * half / S.nstrips);
    const long long xe = 2 * ((long long)(strip + 1) * half / S.nstrips);
    long long c = xs - 2 + 2 * lane;
    c %= G.nx;
    if (c < 0) c += G.nx;
    J.c0 = c;
    J.store = lane >= 1 && lane <= (int)((xe - xs) >> 1);
    J.r0 = seg * S.yseg;
    J.r1 = J.r0 + S.yseg < G.sloc ? J.r0 + S.yseg : G.sloc;
    J.up = (seg & 1) == 0;
    return J;
}

// CARRY: the vectors added at the store are the SAME Y_j the stage argument is formed from (yout[j] == yin[j], every stage of
// RA34PW2 but the first).  Their output combination sum_j aout[j] Y_j is then formed from the values the window load has in
// registers anyway and waits in a four-row delay line until its row is stored -- each Y_j crosses HBM once instead of twice (the row
// is three rows behind by the time it is stored: 24 KB per wave further on, more than a CU's share of the L2 holds): 20 -> 14 vector
// passes over the four RHS launches of a step.  The delay line is a wave-private LDS ring (8 KB per wave and ligand pair): kept in
// registers it cost a wave per SIMD and most of the gain.
template <int NL, bool CARRY = false>
__global__ void __launch_bounds__(KSFD_BLOCK) k_rhs2d_fused(KGeom G, KPhys P, KStrips S, const double *__restrict__ u,
                                                            KSrc src, double *__restrict__ out, KComb cmb = KComb{},
                                                            double *__restrict__ normpart = nullptr, KDots dots = KDots{})
{
    // normpart != NULL: every wave also leaves the sum of squares of what it stored in normpart[wave id] (||out||^2 without
    // a pass of its own; finished by k_reduce_rows in a fixed order), and its share of the inner products `dots` asks for
    const KWaveJob J = ksfd_wave_job(G, S);
    if (!J.valid) return;                       // whole waves only; the kernel has no block barrier
    double nacc = 0.0, dacc[2] = { 0.0, 0.0 };
    // 5-row windows, two columns per lane: slot s <-> row (r - 2 + s)
    double rw[5][2], gw[5][2], uw[NL][5][2];
    double nr[2], nu[NL][2];                   // raw values of the row being prefetched
    // CARRY: sum_j aout[j] Y_j of a row waits in a wave-private LDS ring of four rows from the load of the row until its store three rows
    // later (in registers the 16-32 doubles cost a wave per SIMD: 178 VGPRs against 146; wave-private and in program order: no barrier)
    __shared__ double2 cring[CARRY ? KSFD_BLOCK / KSFD_WAVE : 1][CARRY ? 4 : 1][CARRY ? NL + 1 : 1][CARRY ? KSFD_WAVE : 1];
    const int cw = CARRY ? (threadIdx.x >> 6) : 0, cl = CARRY ? (threadIdx.x & (KSFD_WAVE - 1)) : 0;
    int nload = 0;                             // rows loaded so far: row r of the segment is load r - r0 + 2

    auto load_row = [&](long long r) {
        const long long o = ksfd_rowoff(G, r) + J.c0;
        double2 t = ksfd_ld2(u + o);
        nr[0] = t.x; nr[1] = t.y;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            double2 q = ksfd_ld2(u + (long long)(l + 1) * G.plane + o);
            nu[l][0] = q.x; nu[l][1] = q.y;
        }
        double2 co[NL + 1];
#pragma unroll
        for (int c = 0; c <= NL; c++) co[c] = make_double2(0.0, 0.0);
        for (int j = 0; j < cmb.nin; j++) {        // wave-uniform: stage argument u + sum a_j Y_j
            const double a = cmb.ain[j], ao = CARRY ? cmb.aout[j] : 0.0;
            double2 y = ksfd_ld2(cmb.yin[j] + o);
            nr[0] += a * y.x; nr[1] += a * y.y;
            if (CARRY) { co[0].x += ao * y.x; co[0].y += ao * y.y; }
#pragma unroll
            for (int l = 0; l < NL; l++) {
                double2 q = ksfd_ld2(cmb.yin[j] + (long long)(l + 1) * G.plane + o);
                nu[l][0] += a * q.x; nu[l][1] += a * q.y;
                if (CARRY) { co[l + 1].x += ao * q.x; co[l + 1].y += ao * q.y; }
            }
        }
        if (CARRY) {
#pragma unroll
            for (int c = 0; c <= NL; c++) cring[cw][nload & 3][c][cl] = co[c];
        }
        nload++;
    };
    auto push_row = [&]() {                     // shift windows up one row, append the prefetched row
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                rw[s][e] = rw[s + 1][e];
                gw[s][e] = gw[s + 1][e];
#pragma unroll
                for (int l = 0; l < NL; l++) uw[l][s][e] = uw[l][s + 1][e];
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            double U[NL], GU[NL], gr;
            double rho = ksfd_clamp(nr[e], P.rhomin);
#pragma unroll
            for (int l = 0; l < NL; l++) { U[l] = ksfd_clamp(nu[l][e], P.Umin); uw[l][4][e] = U[l]; }
            ksfd_G<NL, false>(P, rho, U, gw[4][e], gr, GU);
            rw[4][e] = rho;
        }
    };

    // prologue: rows r0-2 .. r0+1 fill slots 1..4 after four pushes
    for (int q = -2; q <= 1; q++) { load_row(J.row(J.r0 + q)); push_row(); }
    load_row(J.row(J.r0 + 2));
    for (long long r = J.r0; r < J.r1; r++) {
        push_row();                             // window now centred on row r
        if (r + 1 < J.r1) load_row(J.row(r + 3));      // prefetch for the next iteration
        // x neighbours of the centre row
        const KX xr = ksfd_xnb(rw[2][0], rw[2][1]);
        const KX xg = ksfd_xnb(gw[2][0], gw[2][1]);
        double d1rx[2], d1gx[2], d2gx[2];
        ksfd_dx(rw[2][0], rw[2][1], xr, d1rx[0], d1rx[1]);
        ksfd_dx(gw[2][0], gw[2][1], xg, d1gx[0], d1gx[1]);
        ksfd_dxx(gw[2][0], gw[2][1], xg, d2gx[0], d2gx[1]);
        double res[NL + 1][2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double d1ry = KSFD_D1(rw[0][e], rw[1][e], rw[3][e], rw[4][e]) * P.inv_h[1];
            const double d1gy = KSFD_D1(gw[0][e], gw[1][e], gw[3][e], gw[4][e]) * P.inv_h[1];
            const double d2gy = KSFD_D2(gw[0][e], gw[1][e], gw[2][e], gw[3][e], gw[4][e]) * P.inv_h2[1];
            res[0][e] = (d1rx[e] * P.inv_h[0]) * (d1gx[e] * P.inv_h[0]) + d1ry * d1gy +
                        rw[2][e] * (d2gx[e] * P.inv_h2[0] + d2gy);
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const KX xu = ksfd_xnb(uw[l][2][0], uw[l][2][1]);
            double d2ux[2];
            ksfd_dxx(uw[l][2][0], uw[l][2][1], xu, d2ux[0], d2ux[1]);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double d2uy = KSFD_D2(uw[l][0][e], uw[l][1][e], uw[l][2][e], uw[l][3][e], uw[l][4][e]) * P.inv_h2[1];
                res[l + 1][e] = -P.lig_gamma[l] * uw[l][2][e] + P.lig_s[l] * rw[2][e] +
                                P.lig_D[l] * (d2ux[e] * P.inv_h2[0] + d2uy);
            }
        }
        if (J.store) {
            const long long pi = J.row(r) * G.nx + J.c0;          // dense interior index
            const long long o = (long long)G.ng * G.inner + pi;   // offset inside a plane
#pragma unroll
            for (int c = 0; c <= NL; c++) {
                double a = res[c][0], b = res[c][1];
                if (src.p[c]) { double2 s = ksfd_ld2(src.p[c] + pi); a += s.x; b += s.y; }
                if (CARRY) { const double2 cc = cring[cw][(int)(r - J.r0 + 2) & 3][c][cl]; a += cc.x; b += cc.y; }
                else for (int j = 0; j < cmb.nout; j++) {
                    const double2 y = ksfd_ld2(cmb.yout[j] + (long long)c * G.plane + o);
                    a += cmb.aout[j] * y.x; b += cmb.aout[j] * y.y;
                }
                ksfd_st2(out + (long long)c * G.plane + o, a, b);
                nacc += a * a + b * b;
#pragma unroll
                for (int q = 0; q < 2; q++)
                    if (q < dots.n) { const double2 y = ksfd_ld2(dots.v[q] + (long long)c * G.plane + o); dacc[q] += a * y.x + b * y.y; }
            }
        }
    }
    if (normpart) {
        nacc = ksfd_wave_sum(nacc);
        if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) normpart[J.wid] = nacc;
        for (int q = 0; q < dots.n; q++) {
            const double t = ksfd_wave_sum(dacc[q]);
            if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) normpart[(long long)(1 + q) * dots.stride + J.wid] = t;
        }
    }
}

template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_jvp2d_fused(KGeom G, KPhys P, KStrips S, const double *__restrict__ u,
                                                            const double *__restrict__ v, int mode, double shift,
                                                            double *__restrict__ out)
{
    const KWaveJob J = ksfd_wave_job(G, S);
    if (!J.valid) return;
    double rw[5][2], gw[5][2], vw[5][2], ew[5][2], zw[NL][5][2];   // rho, G, v_rho, dG, v_U
    double nr[2], nv[2], nu[NL][2], nz[NL][2];

    auto load_row = [&](long long r) {
        const long long o = ksfd_rowoff(G, r) + J.c0;
        double2 t = ksfd_ld2(u + o), w = ksfd_ld2(v + o);
        nr[0] = t.x; nr[1] = t.y; nv[0] = w.x; nv[1] = w.y;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            double2 q = ksfd_ld2(u + (long long)(l + 1) * G.plane + o);
            double2 z = ksfd_ld2(v + (long long)(l + 1) * G.plane + o);
            nu[l][0] = q.x; nu[l][1] = q.y; nz[l][0] = z.x; nz[l][1] = z.y;
        }
    };
    auto push_row = [&]() {
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                rw[s][e] = rw[s + 1][e]; gw[s][e] = gw[s + 1][e];
                vw[s][e] = vw[s + 1][e]; ew[s][e] = ew[s + 1][e];
#pragma unroll
                for (int l = 0; l < NL; l++) zw[l][s][e] = zw[l][s + 1][e];
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            double U[NL], GU[NL], gr;
            double rho = ksfd_clamp(nr[e], P.rhomin);
#pragma unroll
            for (int l = 0; l < NL; l++) U[l] = ksfd_clamp(nu[l][e], P.Umin);
            ksfd_G<NL, true>(P, rho, U, gw[4][e], gr, GU);
            double d = gr * nv[e];
#pragma unroll
            for (int l = 0; l < NL; l++) { d += GU[l] * nz[l][e]; zw[l][4][e] = nz[l][e]; }
            rw[4][e] = rho; vw[4][e] = nv[e]; ew[4][e] = d;
        }
    };

    for (int q = -2; q <= 1; q++) { load_row(J.row(J.r0 + q)); push_row(); }
    load_row(J.row(J.r0 + 2));
    for (long long r = J.r0; r < J.r1; r++) {
        push_row();
        if (r + 1 < J.r1) load_row(J.row(r + 3));
        const KX xr = ksfd_xnb(rw[2][0], rw[2][1]);
        const KX xg = ksfd_xnb(gw[2][0], gw[2][1]);
        const KX xv = ksfd_xnb(vw[2][0], vw[2][1]);
        const KX xe = ksfd_xnb(ew[2][0], ew[2][1]);
        double d1r[2], d1g[2], d1v[2], d1e[2], d2g[2], d2e[2];
        ksfd_dx(rw[2][0], rw[2][1], xr, d1r[0], d1r[1]);
        ksfd_dx(gw[2][0], gw[2][1], xg, d1g[0], d1g[1]);
        ksfd_dx(vw[2][0], vw[2][1], xv, d1v[0], d1v[1]);
        ksfd_dx(ew[2][0], ew[2][1], xe, d1e[0], d1e[1]);
        ksfd_dxx(gw[2][0], gw[2][1], xg, d2g[0], d2g[1]);
        ksfd_dxx(ew[2][0], ew[2][1], xe, d2e[0], d2e[1]);
        double res[NL + 1][2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double ih0 = P.inv_h[0], ih1 = P.inv_h[1];
            const double yr = KSFD_D1(rw[0][e], rw[1][e], rw[3][e], rw[4][e]) * ih1;
            const double yg = KSFD_D1(gw[0][e], gw[1][e], gw[3][e], gw[4][e]) * ih1;
            const double yv = KSFD_D1(vw[0][e], vw[1][e], vw[3][e], vw[4][e]) * ih1;
            const double ye = KSFD_D1(ew[0][e], ew[1][e], ew[3][e], ew[4][e]) * ih1;
            const double lapG = d2g[e] * P.inv_h2[0] + KSFD_D2(gw[0][e], gw[1][e], gw[2][e], gw[3][e], gw[4][e]) * P.inv_h2[1];
            const double lapE = d2e[e] * P.inv_h2[0] + KSFD_D2(ew[0][e], ew[1][e], ew[2][e], ew[3][e], ew[4][e]) * P.inv_h2[1];
            const double jr = (d1v[e] * ih0) * (d1g[e] * ih0) + (d1r[e] * ih0) * (d1e[e] * ih0) + yv * yg + yr * ye +
                              vw[2][e] * lapG + rw[2][e] * lapE;
            res[0][e] = mode ? shift * vw[2][e] - jr : jr;
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const KX xz = ksfd_xnb(zw[l][2][0], zw[l][2][1]);
            double d2z[2];
            ksfd_dxx(zw[l][2][0], zw[l][2][1], xz, d2z[0], d2z[1]);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double lap = d2z[e] * P.inv_h2[0] +
                                   KSFD_D2(zw[l][0][e], zw[l][1][e], zw[l][2][e], zw[l][3][e], zw[l][4][e]) * P.inv_h2[1];
                const double ju = -P.lig_gamma[l] * zw[l][2][e] + P.lig_s[l] * vw[2][e] + P.lig_D[l] * lap;
                res[l + 1][e] = mode ? shift * zw[l][2][e] - ju : ju;
            }
        }
        if (J.store) {
            const long long o = (long long)G.ng * G.inner + J.row(r) * G.nx + J.c0;
#pragma unroll
            for (int c = 0; c <= NL; c++) ksfd_st2(out + (long long)c * G.plane + o, res[c][0], res[c][1]);
        }
    }
}

// CFL check on a 2-D grid (KSFDTS.CFL_step, KSFD/ksfdts.py:302-319): per-axis max |D1(G)| only.  One thread per pair of
// columns, rows by blockIdx.y; the x-neighbours are two more 16-B loads (L1 hits), the y-neighbours four.
__global__ void __launch_bounds__(KSFD_BLOCK) k_velmax2d(KGeom G, KPhys P, const double *__restrict__ Gb, double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE][2];
    double mx = 0.0, my = 0.0;
    const long long half = G.nx >> 1;
    const long long pc = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pc < half) {
        const long long x = 2 * pc;
        const long long xl = x >= 2 ? x - 2 : G.nx - 2, xr = x + 2 < G.nx ? x + 2 : 0;
        for (long long r = blockIdx.y; r < G.sloc; r += gridDim.y) {
            const double *row = Gb + ksfd_rowoff(G, r);
            const double2 c = ksfd_ld2(row + x), l = ksfd_ld2(row + xl), rr = ksfd_ld2(row + xr);
            const double2 m2 = ksfd_ld2(Gb + ksfd_rowoff(G, r - 2) + x), m1 = ksfd_ld2(Gb + ksfd_rowoff(G, r - 1) + x),
                          p1 = ksfd_ld2(Gb + ksfd_rowoff(G, r + 1) + x), p2 = ksfd_ld2(Gb + ksfd_rowoff(G, r + 2) + x);
            mx = fmax(mx, fmax(fabs(KSFD_D1(l.x, l.y, c.y, rr.x)), fabs(KSFD_D1(l.y, c.x, rr.x, rr.y))));
            my = fmax(my, fmax(fabs(KSFD_D1(m2.x, m1.x, p1.x, p2.x)), fabs(KSFD_D1(m2.y, m1.y, p1.y, p2.y))));
        }
    }
    mx = ksfd_wave_max(mx * P.inv_h[0]);
    my = ksfd_wave_max(my * P.inv_h[1]);
    if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) { red[threadIdx.x / KSFD_WAVE][0] = mx; red[threadIdx.x / KSFD_WAVE][1] = my; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double m = 0.0;
        if (threadIdx.x < 2) for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) m = fmax(m, red[q][threadIdx.x]);
        const long long nb = (long long)gridDim.x * gridDim.y;
        part[(long long)threadIdx.x * nb + (long long)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
}

// The same check on a 3-D grid.  One thread per (pair of x columns, y), marching along z through a segment of the slab with a
// five-plane register window, so that every plane is read from memory once (a row-by-row sweep re-reads the four z-neighbour
// planes, 2 MB apart at 512^2 and too many for the L2: 1.3 ms at 512^3 against 0.3 ms this way); y wraps inside the plane, z has
// ghost planes (or wraps on one rank).  blockIdx.x: 256 threads = consecutive (x pair, y); blockIdx.y: z segment.
__global__ void __launch_bounds__(KSFD_BLOCK) k_velmax3d(KGeom G, KPhys P, int zseg, const double *__restrict__ Gb, double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE][3];
    double mx = 0.0, my = 0.0, mz = 0.0;
    const int half = (int)(G.nx >> 1), ny = (int)G.ny, nx = (int)G.nx;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < half * ny) {
        const int y = t / half, x = 2 * (t - y * half);
        const int xl = x >= 2 ? x - 2 : nx - 2, xr = x + 2 < nx ? x + 2 : 0;
        const int ym2 = y >= 2 ? y - 2 : y - 2 + ny, ym1 = y >= 1 ? y - 1 : ny - 1, yp1 = y + 1 < ny ? y + 1 : 0, yp2 = y + 2 < ny ? y + 2 : y + 2 - ny;
        auto zoff = [&](int k) -> long long {                // plane k of the slab, k = -2 .. sloc+1
            if (G.wrap_slow) { if (k < 0) k += (int)G.sloc; else if (k >= (int)G.sloc) k -= (int)G.sloc; return (long long)k * G.inner; }
            return (long long)(k + G.ng) * G.inner;
        };
        const int z0 = blockIdx.y * zseg, z1 = min(z0 + zseg, (int)G.sloc);
        const long long yo = (long long)y * nx + x;
        double2 m2 = ksfd_ld2(Gb + zoff(z0 - 2) + yo), m1 = ksfd_ld2(Gb + zoff(z0 - 1) + yo), c = ksfd_ld2(Gb + zoff(z0) + yo), p1 = ksfd_ld2(Gb + zoff(z0 + 1) + yo);
#pragma unroll 4
        for (int z = z0; z < z1; z++) {
            const double *pl = Gb + zoff(z);
            const double2 p2 = ksfd_ld2(Gb + zoff(z + 2) + yo);
            const double2 l = ksfd_ld2(pl + (long long)y * nx + xl), rr = ksfd_ld2(pl + (long long)y * nx + xr);
            const double2 a2 = ksfd_ld2(pl + (long long)ym2 * nx + x), a1 = ksfd_ld2(pl + (long long)ym1 * nx + x),
                          b1 = ksfd_ld2(pl + (long long)yp1 * nx + x), b2 = ksfd_ld2(pl + (long long)yp2 * nx + x);
            mx = fmax(mx, fmax(fabs(KSFD_D1(l.x, l.y, c.y, rr.x)), fabs(KSFD_D1(l.y, c.x, rr.x, rr.y))));
            my = fmax(my, fmax(fabs(KSFD_D1(a2.x, a1.x, b1.x, b2.x)), fabs(KSFD_D1(a2.y, a1.y, b1.y, b2.y))));
            mz = fmax(mz, fmax(fabs(KSFD_D1(m2.x, m1.x, p1.x, p2.x)), fabs(KSFD_D1(m2.y, m1.y, p1.y, p2.y))));
            m2 = m1; m1 = c; c = p1; p1 = p2;
        }
    }
    mx = ksfd_wave_max(mx * P.inv_h[0]);
    my = ksfd_wave_max(my * P.inv_h[1]);
    mz = ksfd_wave_max(mz * P.inv_h[2]);
    if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) { red[threadIdx.x / KSFD_WAVE][0] = mx; red[threadIdx.x / KSFD_WAVE][1] = my; red[threadIdx.x / KSFD_WAVE][2] = mz; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double m = 0.0;
        for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) m = fmax(m, red[q][threadIdx.x]);
        const long long nb = (long long)gridDim.x * gridDim.y;
        part[(long long)threadIdx.x * nb + (long long)blockIdx.y * gridDim.x + blockIdx.x] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// Frozen-Jacobian path.  The Rosenbrock-W step keeps J = df/du(t_n, u_n) for all four stages and
// every GMRES iteration (~40 Jacobian actions per step), so everything in J that depends only on
// u_n is evaluated ONCE per step into a coefficient vector C of (3 + NL) planes
//      C = [ rho (clamped), G, G_rho, G_U1 .. G_UNL ]
// and the per-iteration kernel is pure stencil arithmetic: no log/tanh/divide.
// Traffic per point: (3+NL) + (1+NL) reads + (1+NL) writes  (F=2: 64 B).
// ---------------------------------------------------------------------------------------------
template <int NL>
// meanpart != NULL: per-block partial sums of rho*G_rho and rho*G_Ul over the OWNED points (the grid means the spectral preconditioner
// is built from: no pass of its own over the planes just written)
__global__ void __launch_bounds__(KSFD_BLOCK) k_jcoef(KGeom G, KPhys P, const double *__restrict__ u,
                                                      double *__restrict__ C, float *__restrict__ C32 = nullptr, double *__restrict__ meanpart = nullptr)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long own0 = (long long)G.ng * G.inner, own1 = own0 + G.nloc;
    double macc[NL + 1];
#pragma unroll
    for (int i = 0; i <= NL; i++) macc[i] = 0.0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < G.plane; e += stride) {
        double rho = ksfd_clamp(u[e], P.rhomin);
        double U[NL], GU[NL];
#pragma unroll
        for (int l = 0; l < NL; l++) U[l] = ksfd_clamp(u[(long long)(l + 1) * G.plane + e], P.Umin);
        double g, gr;
        ksfd_G<NL, true>(P, rho, U, g, gr, GU);
        C[e] = rho;
        C[G.plane + e] = g;
        C[2 * G.plane + e] = gr;
#pragma unroll
        for (int l = 0; l < NL; l++) C[(long long)(3 + l) * G.plane + e] = GU[l];
        if (C32) {                                   // copy for the polynomial preconditioner's Jacobian actions
            C32[e] = (float)rho;
            C32[G.plane + e] = (float)g;
            C32[2 * G.plane + e] = (float)gr;
#pragma unroll
            for (int l = 0; l < NL; l++) C32[(long long)(3 + l) * G.plane + e] = (float)GU[l];
        }
        if (meanpart && e >= own0 && e < own1) {
            macc[0] += rho * gr;
#pragma unroll
            for (int l = 0; l < NL; l++) macc[l + 1] += rho * GU[l];
        }
    }
    if (meanpart) {
        __shared__ double red[KSFD_BLOCK / KSFD_WAVE][NL + 1];
        const int lane = threadIdx.x & (KSFD_WAVE - 1), wv = threadIdx.x / KSFD_WAVE;
#pragma unroll
        for (int i = 0; i <= NL; i++) {
            const double t = ksfd_wave_sum(macc[i]);
            if (lane == 0) red[wv][i] = t;
        }
        __syncthreads();
        if (threadIdx.x <= NL) {
            double t = 0.0;
            for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) t += red[q][threadIdx.x];
            meanpart[(long long)threadIdx.x * gridDim.x + blockIdx.x] = t;
        }
    }
}

// dG = G_rho v_rho + sum_l G_Ul v_Ul over the whole slab (generic path helper)
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_dg_frozen(KGeom G, const double *__restrict__ C,
                                                          const double *__restrict__ v, double *__restrict__ dG)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < G.plane; e += stride) {
        double d = C[2 * G.plane + e] * v[e];
#pragma unroll
        for (int l = 0; l < NL; l++) d += C[(long long)(3 + l) * G.plane + e] * v[(long long)(l + 1) * G.plane + e];
        dG[e] = d;
    }
}

// Storage types: TC coefficient planes, TV input vector, TY the added vector of modes 2/3, TO output (double everywhere
// except inside the polynomial preconditioner, whose Horner temporaries and coefficient copy are fp32).
// PF = rows in flight ahead of the one being computed.  PF = 2 (fits for NL = 1: 226 VGPRs, still 2 waves/SIMD) was measured
// and gains nothing: the kernel is co-limited by HBM and fp64/VALU issue (~2000 cycles per row and wave), not by latency.
// Also measured and dropped: rotating the 5-row windows through a five-fold unrolled row loop (no register moves, same
// 196 VGPR): 6 % SLOWER (0.198 -> 0.210 ms), the five copies of the body cost more in instruction fetch than the ~100
// v_mov per row they save.
template <int NL, typename TC = double, typename TV = double, typename TY = double, typename TO = double, int PF = 1, bool SMOOTH = false, typename TS = double>
__global__ void __launch_bounds__(KSFD_BLOCK) k_jvp2d_frozen(KGeom G, KPhys P, KStrips S, const TC *__restrict__ C,
                                                             const TV *__restrict__ v, int mode, double shift,
                                                             TO *__restrict__ out, const TY *__restrict__ yadd = nullptr,
                                                             double alpha = 0.0, double beta = 0.0, KSmoothT<TS> sm = KSmoothT<TS>{},
                                                             double *__restrict__ normpart = nullptr)
{
    // normpart != NULL (modes 0-4): ||out||^2 partials per wave, see k_rhs2d_fused
    // mode 0: out = J v ; 1: out = shift*v - J v ; 2: out = yadd - (shift*v - J v)   (residual b - A x) ;
    // 3: out = alpha*yadd + beta*(shift*v - J v)   (one Horner step of the polynomial preconditioner) ;
    // 4: out = alpha*v + beta*(shift*v - J v)      (first Horner step: no extra plane is read)
    const KWaveJob J = ksfd_wave_job(G, S);
    if (!J.valid) return;
    typedef typename std::conditional<std::is_same<TV, float>::value, float, double>::type WV;     // window of v in its storage type
    double rw[5][2], gw[5][2], ew[5][2];   // rho, G, dG
    WV vw[5][2], zw[NL][5][2];             // v_rho, v_U
    double nr[PF][2], ng[PF][2], nq[PF][2], nc[PF][NL][2];
    WV nv[PF][2], nz[PF][NL][2];
    double nacc = 0.0;
    typedef std::integral_constant<int, 0> B0;
    typedef std::integral_constant<int, (PF > 1 ? 1 : 0)> B1;

    auto load_row = [&](auto Bc, long long r) {
        constexpr int B = decltype(Bc)::value;
        const long long o = ksfd_rowoff(G, r) + J.c0;
        double2 a = ksfd_ld2(C + o), b = ksfd_ld2(C + G.plane + o), c = ksfd_ld2(C + 2 * G.plane + o);
        const auto w = ksfd_ldw(v + o);
        nr[B][0] = a.x; nr[B][1] = a.y; ng[B][0] = b.x; ng[B][1] = b.y; nq[B][0] = c.x; nq[B][1] = c.y; nv[B][0] = w.x; nv[B][1] = w.y;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            double2 q = ksfd_ld2(C + (long long)(3 + l) * G.plane + o);
            const auto z = ksfd_ldw(v + (long long)(l + 1) * G.plane + o);
            nc[B][l][0] = q.x; nc[B][l][1] = q.y; nz[B][l][0] = z.x; nz[B][l][1] = z.y;
        }
    };
    auto push_row = [&](auto Bc) {
        constexpr int B = decltype(Bc)::value;
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                rw[s][e] = rw[s + 1][e]; gw[s][e] = gw[s + 1][e];
                vw[s][e] = vw[s + 1][e]; ew[s][e] = ew[s + 1][e];
#pragma unroll
                for (int l = 0; l < NL; l++) zw[l][s][e] = zw[l][s + 1][e];
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            double d = nq[B][e] * (double)nv[B][e];
#pragma unroll
            for (int l = 0; l < NL; l++) { d += nc[B][l][e] * (double)nz[B][l][e]; zw[l][4][e] = nz[B][l][e]; }
            rw[4][e] = nr[B][e]; gw[4][e] = ng[B][e]; vw[4][e] = nv[B][e]; ew[4][e] = d;
        }
    };

    for (int q = -2; q <= 1; q++) { load_row(B0(), J.row(J.r0 + q)); push_row(B0()); }
    load_row(B0(), J.row(J.r0 + 2));
    if (PF > 1 && J.r0 + 1 < J.r1) load_row(B1(), J.row(J.r0 + 3));
    // one row: window <- pending buffer (holds row r+2), refill that buffer with row r+2+PF, compute and store row r
    auto do_row = [&](auto Bc, long long r) {
        push_row(Bc);
        if (r + PF < J.r1) load_row(Bc, J.row(r + 2 + PF));
        // the added vector of modes 2/3 is needed only at the store: issue its loads now, behind the next row's
        decltype(ksfd_ldw(yadd)) yv[NL + 1];                 // kept in the storage type until they are used (registers)
        if ((mode == 2 || mode == 3 || (SMOOTH && mode == 5)) && J.store) {
            const long long oy = (long long)G.ng * G.inner + J.row(r) * G.nx + J.c0;
#pragma unroll
            for (int c = 0; c <= NL; c++) yv[c] = ksfd_ldw(yadd + (long long)c * G.plane + oy);
        }
        float2 dv_[SMOOTH ? (NL + 1) * (NL + 1) : 1];
        decltype(ksfd_ldw(sm.rr)) xs_[SMOOTH ? NL + 1 : 1], rs_[SMOOTH ? NL + 1 : 1];
        if constexpr (SMOOTH) {
            if (mode >= 5 && J.store) {
                const long long oy = (long long)G.ng * G.inner + J.row(r) * G.nx + J.c0;
#pragma unroll
                for (int q = 0; q < (NL + 1) * (NL + 1); q++) dv_[q] = ksfd_ldw(sm.dinv + (long long)q * G.plane + oy);
                if (mode == 6) {
#pragma unroll
                    for (int c = 0; c <= NL; c++) {
                        if (!sm.x_has_d) xs_[c] = ksfd_ldw((const TS *)sm.x + (long long)c * G.plane + oy);     // x_has_d: x == d, which is v itself
                        rs_[c] = ksfd_ldw(sm.rr + (long long)c * G.plane + oy);
                    }
                }
            }
        }
        const KX xr = ksfd_xnb(rw[2][0], rw[2][1]);
        const KX xg = ksfd_xnb(gw[2][0], gw[2][1]);
        const auto xv = ksfd_xnb(vw[2][0], vw[2][1]);
        const KX xe = ksfd_xnb(ew[2][0], ew[2][1]);
        double d1r[2], d1g[2], d1v[2], d1e[2], d2g[2], d2e[2];
        ksfd_dx(rw[2][0], rw[2][1], xr, d1r[0], d1r[1]);
        ksfd_dx(gw[2][0], gw[2][1], xg, d1g[0], d1g[1]);
        ksfd_dx(vw[2][0], vw[2][1], xv, d1v[0], d1v[1]);
        ksfd_dx(ew[2][0], ew[2][1], xe, d1e[0], d1e[1]);
        ksfd_dxx(gw[2][0], gw[2][1], xg, d2g[0], d2g[1]);
        ksfd_dxx(ew[2][0], ew[2][1], xe, d2e[0], d2e[1]);
        double res[NL + 1][2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double ih0 = P.inv_h[0], ih1 = P.inv_h[1];
            const double yr = KSFD_D1(rw[0][e], rw[1][e], rw[3][e], rw[4][e]) * ih1;
            const double yg = KSFD_D1(gw[0][e], gw[1][e], gw[3][e], gw[4][e]) * ih1;
            const double yv = KSFD_D1((double)vw[0][e], (double)vw[1][e], (double)vw[3][e], (double)vw[4][e]) * ih1;
            const double ye = KSFD_D1(ew[0][e], ew[1][e], ew[3][e], ew[4][e]) * ih1;
            const double lapG = d2g[e] * P.inv_h2[0] + KSFD_D2(gw[0][e], gw[1][e], gw[2][e], gw[3][e], gw[4][e]) * P.inv_h2[1];
            const double lapE = d2e[e] * P.inv_h2[0] + KSFD_D2(ew[0][e], ew[1][e], ew[2][e], ew[3][e], ew[4][e]) * P.inv_h2[1];
            const double jr = (d1v[e] * ih0) * (d1g[e] * ih0) + (d1r[e] * ih0) * (d1e[e] * ih0) + yv * yg + yr * ye +
                              (double)vw[2][e] * lapG + rw[2][e] * lapE;
            res[0][e] = mode ? shift * (double)vw[2][e] - jr : jr;
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const auto xz = ksfd_xnb(zw[l][2][0], zw[l][2][1]);
            double d2z[2];
            ksfd_dxx(zw[l][2][0], zw[l][2][1], xz, d2z[0], d2z[1]);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double lap = d2z[e] * P.inv_h2[0] +
                                   KSFD_D2((double)zw[l][0][e], (double)zw[l][1][e], (double)zw[l][2][e], (double)zw[l][3][e], (double)zw[l][4][e]) * P.inv_h2[1];
                const double ju = -P.lig_gamma[l] * (double)zw[l][2][e] + P.lig_s[l] * (double)vw[2][e] + P.lig_D[l] * lap;
                res[l + 1][e] = mode ? shift * (double)zw[l][2][e] - ju : ju;
            }
        }
        if (SMOOTH && mode >= 5) {
            if constexpr (SMOOTH) {
                if (J.store) {
                    const long long o = (long long)G.ng * G.inner + J.row(r) * G.nx + J.c0;
                    double q0[NL + 1], q1[NL + 1];                    // mode 5: r = yadd - A v ; mode 6: rr - A d
#pragma unroll
                    for (int c = 0; c <= NL; c++) {
                        q0[c] = (mode == 5 ? (double)yv[c].x : (double)rs_[c].x) - res[c][0];
                        q1[c] = (mode == 5 ? (double)yv[c].y : (double)rs_[c].y) - res[c][1];
                    }
#pragma unroll
                    for (int a = 0; a <= NL; a++) {
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int c = 0; c <= NL; c++) { s0 += (double)dv_[a * (NL + 1) + c].x * q0[c]; s1 += (double)dv_[a * (NL + 1) + c].y * q1[c]; }
                        if (mode == 5) {
                            ksfd_st2(out + (long long)a * G.plane + o, q0[a], q1[a]);
                            ksfd_st2(sm.out2 + (long long)a * G.plane + o, sm.scale * s0, sm.scale * s1);
                        } else {
                            const double d0 = a == 0 ? (double)vw[2][0] : (double)zw[a > 0 ? a - 1 : 0][2][0], d1 = a == 0 ? (double)vw[2][1] : (double)zw[a > 0 ? a - 1 : 0][2][1];
                            const double xn0 = (sm.x_has_d ? d0 : (double)xs_[a].x + d0) + sm.c1 * d0 + sm.c2 * s0;
                            const double xn1 = (sm.x_has_d ? d1 : (double)xs_[a].y + d1) + sm.c1 * d1 + sm.c2 * s1;
                            if (sm.x64) ksfd_st2(sm.x64 + (long long)a * G.plane + o, xn0, xn1);
                            else ksfd_st2(sm.x + (long long)a * G.plane + o, xn0, xn1);
                        }
                    }
                }
            }
        } else if (J.store) {
            const long long o = (long long)G.ng * G.inner + J.row(r) * G.nx + J.c0;
#pragma unroll
            for (int c = 0; c <= NL; c++) {
                double a = res[c][0], b = res[c][1];
                if (mode == 4) {            // yadd is the input vector itself: its centre values are in the windows
                    const double c0 = c == 0 ? vw[2][0] : zw[c > 0 ? c - 1 : 0][2][0], c1 = c == 0 ? vw[2][1] : zw[c > 0 ? c - 1 : 0][2][1];
                    a = alpha * c0 + beta * a; b = alpha * c1 + beta * b;
                } else if (mode >= 2) {
                    const double2 yy = make_double2((double)yv[c].x, (double)yv[c].y);
                    if (mode == 2) { a = yy.x - a; b = yy.y - b; } else { a = alpha * yy.x + beta * a; b = alpha * yy.y + beta * b; }
                }
                ksfd_st2(out + (long long)c * G.plane + o, a, b);
                nacc += a * a + b * b;
            }
        }
    };
    if (PF == 1) {
        for (long long r = J.r0; r < J.r1; r++) do_row(B0(), r);
    } else {
        for (long long r = J.r0; r < J.r1; r += 2) {
            do_row(B0(), r);
            if (r + 1 < J.r1) do_row(B1(), r + 1);
        }
    }
    if (normpart) {
        nacc = ksfd_wave_sum(nacc);
        if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) normpart[J.wid] = nacc;
    }
}

// ---------------------------------------------------------------------------------------------
// 3-D frozen Jacobian action: z-marching.  A wave owns a strip of 128 columns (2 per lane) of ONE
// y row and marches down a segment of z planes with 5-plane register windows of rho, G, v_rho, dG, v_U
// (z-neighbours); x-neighbours by wave shuffles; the 4 y-neighbour rows of the centre plane are read
// with 16-B loads that hit L1/L2 (the 4 waves of a block own 4 adjacent rows).  dG is a stored plane
// (k_dg_frozen) because the y-neighbours need it.  Traffic: dg pass 8N(3+2n) + this pass 8N(5+n+F) + F writes.
// ---------------------------------------------------------------------------------------------
// rows: y rows (= waves) per block.  The y-neighbour rows come through the caches, and not for free: PMC at 512^3 shows k_jvp3d_frozen
// reading 18.3 GB per launch where its operands are 7.5 GB (136 B per point instead of 56) at 6.4 TB/s of fabric traffic.  A row of plane k is
// loaded by its own wave three steps before its four neighbour waves ask for it, by which time an XCD's L2 (4 MB, ~1.9 MB of traffic per step)
// has let it go.  Round 3 tried 8 rows per block (kernels templated on ROWS: the launch bound sets the register budget): slower; a barrier per
// plane so that the four requests for a row come together: -4 %, kept (K3D.sync).  What would remove the repeated traffic is a block that
// stages the centre plane of its rows + 4 halo rows in the LDS; not built.
struct K3D {
    int nstrips, nygrp, nzseg, zseg, nblocks, rows;
    int sync;        // 1: the waves of a block march in step (a barrier per plane; needs ny % rows == 0: no wave leaves early) -- the four requests
                     // for a row of a plane then come at the same time and three of them hit the cache
};

__device__ __forceinline__ long long ksfd_planeoff(const KGeom &G, long long k)
{
    const long long pl = G.nx * G.ny;
    if (G.wrap_slow) {
        k %= G.sloc;
        if (k < 0) k += G.sloc;
        return k * pl;
    }
    return (k + G.ng) * pl;
}

// TO: storage type of `out` (float for the residual of the spectral defect correction, which only the preconditioner reads);
// normpart != NULL: sum of squares of everything this wave stored -> normpart[blockIdx.x * 4 + wave] (fixed-order second stage)
template <int NL, typename TO = double, int ROWS = 4>
__global__ void __launch_bounds__(ROWS * KSFD_WAVE) k_jvp3d_frozen(KGeom G, KPhys P, K3D S, const double *__restrict__ C,
                                                             const double *__restrict__ v, const double *__restrict__ dG,
                                                             int mode, double shift, TO *__restrict__ out,
                                                             const double *__restrict__ yadd = nullptr, double alpha = 0.0, double beta = 0.0,
                                                             double *__restrict__ normpart = nullptr)
{
    const int lane = threadIdx.x & (KSFD_WAVE - 1), wv = threadIdx.x >> 6;
    if (normpart && lane == 0) normpart[(long long)blockIdx.x * ROWS + wv] = 0.0;      // waves that leave early contribute nothing
    double nacc = 0.0;
    const long long bid = ksfd_xcd_remap(blockIdx.x, S.nblocks);
    const long long nb_valid = (long long)S.nstrips * S.nygrp * S.nzseg;
    if (bid >= nb_valid) return;
    const int strip = (int)(bid % S.nstrips);
    const long long ygrp = (bid / S.nstrips) % S.nygrp, zs = bid / ((long long)S.nstrips * S.nygrp);
    const long long y = ygrp * ROWS + wv;
    if (y >= G.ny) return;                                   // whole wave; no block barrier in this kernel
    const long long half = G.nx >> 1;
    const long long xs = 2 * ((long long)strip * half / S.nstrips), xe = 2 * ((long long)(strip + 1) * half / S.nstrips);
    long long c0 = (xs - 2 + 2 * lane) % G.nx;
    if (c0 < 0) c0 += G.nx;
    const bool store = lane >= 1 && lane <= (int)((xe - xs) >> 1);
    const long long k0 = zs * S.zseg, k1 = k0 + S.zseg < G.sloc ? k0 + S.zseg : G.sloc;
    const bool up = (zs & 1) == 0;                 // alternate marching direction: see KWaveJob::up
    auto kmap = [&](long long q) { return up ? q : (k0 + k1 - 1 - q); };
    const long long rowc = y * G.nx + c0;
    long long yo[4];                                          // row offsets of y-2, y-1, y+1, y+2 (periodic)
    {
        const int dm[4] = { -2, -1, 1, 2 };
#pragma unroll
        for (int q = 0; q < 4; q++) { long long yy = (y + dm[q]) % G.ny; if (yy < 0) yy += G.ny; yo[q] = yy * G.nx + c0; }
    }
    const double *Cr = C, *Cg = C + G.plane;
    double rw[5][2], gw[5][2], vw[5][2], ew[5][2], zw[NL][5][2];
    double nr[2], ng[2], nv[2], ne[2], nz[NL][2];
    auto load_plane = [&](long long k) {
        const long long o = ksfd_planeoff(G, k) + rowc;
        double2 a = ksfd_ld2(Cr + o), b = ksfd_ld2(Cg + o), w = ksfd_ld2(v + o), e = ksfd_ld2(dG + o);
        nr[0] = a.x; nr[1] = a.y; ng[0] = b.x; ng[1] = b.y; nv[0] = w.x; nv[1] = w.y; ne[0] = e.x; ne[1] = e.y;
#pragma unroll
        for (int l = 0; l < NL; l++) { double2 z = ksfd_ld2(v + (long long)(l + 1) * G.plane + o); nz[l][0] = z.x; nz[l][1] = z.y; }
    };
    auto push = [&]() {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                rw[s][e] = rw[s + 1][e]; gw[s][e] = gw[s + 1][e]; vw[s][e] = vw[s + 1][e]; ew[s][e] = ew[s + 1][e];
#pragma unroll
                for (int l = 0; l < NL; l++) zw[l][s][e] = zw[l][s + 1][e];
            }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            rw[4][e] = nr[e]; gw[4][e] = ng[e]; vw[4][e] = nv[e]; ew[4][e] = ne[e];
#pragma unroll
            for (int l = 0; l < NL; l++) zw[l][4][e] = nz[l][e];
        }
    };
    for (int q = -2; q <= 1; q++) { load_plane(kmap(k0 + q)); push(); }
    load_plane(kmap(k0 + 2));
    for (long long k = k0; k < k1; k++) {
        if (S.sync) __syncthreads();
        push();
        if (k + 1 < k1) load_plane(kmap(k + 3));
        const long long kc = kmap(k);
        const long long po = ksfd_planeoff(G, kc);
        // the added vector of modes 2/3 is needed only at the store: issue its loads now (see k_jvp2d_frozen)
        double2 yv_add[NL + 1];
        if ((mode == 2 || mode == 3) && store) {
            const long long oy = (long long)G.ng * G.inner + kc * G.nx * G.ny + rowc;
#pragma unroll
            for (int c = 0; c <= NL; c++) yv_add[c] = ksfd_ld2(yadd + (long long)c * G.plane + oy);
        }
        // y-neighbour rows of the centre plane
        double2 yr[4], yg[4], yv[4], ye[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            yr[q] = ksfd_ld2(Cr + po + yo[q]); yg[q] = ksfd_ld2(Cg + po + yo[q]);
            yv[q] = ksfd_ld2(v + po + yo[q]);  ye[q] = ksfd_ld2(dG + po + yo[q]);
        }
        const KX xr = ksfd_xnb(rw[2][0], rw[2][1]), xg = ksfd_xnb(gw[2][0], gw[2][1]);
        const KX xv = ksfd_xnb(vw[2][0], vw[2][1]), xe_ = ksfd_xnb(ew[2][0], ew[2][1]);
        double d1r[2], d1g[2], d1v[2], d1e[2], d2g[2], d2e[2];
        ksfd_dx(rw[2][0], rw[2][1], xr, d1r[0], d1r[1]);
        ksfd_dx(gw[2][0], gw[2][1], xg, d1g[0], d1g[1]);
        ksfd_dx(vw[2][0], vw[2][1], xv, d1v[0], d1v[1]);
        ksfd_dx(ew[2][0], ew[2][1], xe_, d1e[0], d1e[1]);
        ksfd_dxx(gw[2][0], gw[2][1], xg, d2g[0], d2g[1]);
        ksfd_dxx(ew[2][0], ew[2][1], xe_, d2e[0], d2e[1]);
        double res[NL + 1][2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double ih0 = P.inv_h[0], ih1 = P.inv_h[1], ih2 = P.inv_h[2];
#define KY(arr, q) (e ? arr[q].y : arr[q].x)
            const double yr1 = KSFD_D1(KY(yr, 0), KY(yr, 1), KY(yr, 2), KY(yr, 3)) * ih1;
            const double yg1 = KSFD_D1(KY(yg, 0), KY(yg, 1), KY(yg, 2), KY(yg, 3)) * ih1;
            const double yv1 = KSFD_D1(KY(yv, 0), KY(yv, 1), KY(yv, 2), KY(yv, 3)) * ih1;
            const double ye1 = KSFD_D1(KY(ye, 0), KY(ye, 1), KY(ye, 2), KY(ye, 3)) * ih1;
            const double yg2 = KSFD_D2(KY(yg, 0), KY(yg, 1), gw[2][e], KY(yg, 2), KY(yg, 3)) * P.inv_h2[1];
            const double ye2 = KSFD_D2(KY(ye, 0), KY(ye, 1), ew[2][e], KY(ye, 2), KY(ye, 3)) * P.inv_h2[1];
            const double zr1 = KSFD_D1(rw[0][e], rw[1][e], rw[3][e], rw[4][e]) * ih2;
            const double zg1 = KSFD_D1(gw[0][e], gw[1][e], gw[3][e], gw[4][e]) * ih2;
            const double zv1 = KSFD_D1(vw[0][e], vw[1][e], vw[3][e], vw[4][e]) * ih2;
            const double ze1 = KSFD_D1(ew[0][e], ew[1][e], ew[3][e], ew[4][e]) * ih2;
            const double zg2 = KSFD_D2(gw[0][e], gw[1][e], gw[2][e], gw[3][e], gw[4][e]) * P.inv_h2[2];
            const double ze2 = KSFD_D2(ew[0][e], ew[1][e], ew[2][e], ew[3][e], ew[4][e]) * P.inv_h2[2];
            const double lapG = d2g[e] * P.inv_h2[0] + yg2 + zg2;
            const double lapE = d2e[e] * P.inv_h2[0] + ye2 + ze2;
            const double jr = (d1v[e] * ih0) * (d1g[e] * ih0) + (d1r[e] * ih0) * (d1e[e] * ih0) + yv1 * yg1 + yr1 * ye1 +
                              zv1 * zg1 + zr1 * ze1 + vw[2][e] * lapG + rw[2][e] * lapE;
            res[0][e] = mode ? shift * vw[2][e] - jr : jr;
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const double *vl = v + (long long)(l + 1) * G.plane + po;
            double2 yz[4];
#pragma unroll
            for (int q = 0; q < 4; q++) yz[q] = ksfd_ld2(vl + yo[q]);
            const KX xz = ksfd_xnb(zw[l][2][0], zw[l][2][1]);
            double d2z[2];
            ksfd_dxx(zw[l][2][0], zw[l][2][1], xz, d2z[0], d2z[1]);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double lap = d2z[e] * P.inv_h2[0] +
                                   KSFD_D2(KY(yz, 0), KY(yz, 1), zw[l][2][e], KY(yz, 2), KY(yz, 3)) * P.inv_h2[1] +
                                   KSFD_D2(zw[l][0][e], zw[l][1][e], zw[l][2][e], zw[l][3][e], zw[l][4][e]) * P.inv_h2[2];
                const double ju = -P.lig_gamma[l] * zw[l][2][e] + P.lig_s[l] * vw[2][e] + P.lig_D[l] * lap;
                res[l + 1][e] = mode ? shift * zw[l][2][e] - ju : ju;
            }
#undef KY
        }
        if (store) {
            const long long o = (long long)G.ng * G.inner + kc * G.nx * G.ny + rowc;
#pragma unroll
            for (int c = 0; c <= NL; c++) {
                double a = res[c][0], b = res[c][1];
                if (mode == 4) {            // yadd is the input vector itself: its centre values are in the windows
                    const double c0 = c == 0 ? vw[2][0] : zw[c > 0 ? c - 1 : 0][2][0], c1 = c == 0 ? vw[2][1] : zw[c > 0 ? c - 1 : 0][2][1];
                    a = alpha * c0 + beta * a; b = alpha * c1 + beta * b;
                } else if (mode >= 2) {
                    const double2 yy = yv_add[c];
                    if (mode == 2) { a = yy.x - a; b = yy.y - b; } else { a = alpha * yy.x + beta * a; b = alpha * yy.y + beta * b; }
                }
                if constexpr (sizeof(TO) == 4) *reinterpret_cast<float2 *>(out + (long long)c * G.plane + o) = make_float2((float)a, (float)b);
                else ksfd_st2(out + (long long)c * G.plane + o, a, b);
                nacc += a * a + b * b;
            }
        }
    }
    if (normpart) {
        nacc = ksfd_wave_sum(nacc);
        if (lane == 0) normpart[(long long)blockIdx.x * ROWS + wv] = nacc;
    }
}


// ---------------------------------------------------------------------------------------------
// 3-D frozen Jacobian action, second generation (round 3): the y-neighbours come out of the LDS, dG is formed on the fly.
// PMC on k_jvp3d_frozen at 512^3: 18.3 GB read per launch for 7.5 GB of operands (the y-neighbour rows of the centre plane, asked for
// through the caches three steps after their own wave loaded them, mostly come from memory again), plus a whole pass (k_dg_frozen,
// 5.4 GB) because the y-neighbours need dG.  Here a block is R = 8 rows (waves) of one 128-column strip marching along z in step:
//   * every wave keeps the 5-plane z windows of rho, G, v_rho, dG, v_U of ITS row in registers as before, with dG = G_rho v_rho +
//     sum_l G_Ul v_Ul formed when a plane is loaded (G_rho, G_U read in place of the dG plane: no separate pass);
//   * per step it publishes its centre-plane values to the LDS; waves 0..3 also load ONE halo row each (the two rows above and the two
//     below the block: rho, G, G_rho, G_U, v -> dG) one step ahead and publish it; after one barrier every wave reads its four
//     y-neighbour rows from the LDS (double-buffered: one barrier per step).
// HBM per point: 8 (4 + 2 NL) own + (4 / R)(4 + 2 NL) halo + vectors of the mode, e.g. residual mode, one ligand: 64 + 24 B against
// 136 + 40 B measured for the old pair of kernels.  x-neighbours by DPP shifts as everywhere.  ny % R == 0 (no wave leaves early).
// ---------------------------------------------------------------------------------------------
#define KSFD_J3L_ROWS 8
template <int NL>
__host__ __device__ constexpr size_t ksfd_j3l_lds_bytes() { return (size_t)2 * (KSFD_J3L_ROWS + 4) * (4 + NL) * KSFD_WAVE * sizeof(double2); }

template <int NL, typename TO = double>
__global__ void __launch_bounds__(KSFD_J3L_ROWS * KSFD_WAVE) k_jvp3d_lds(KGeom G, KPhys P, K3D S, const double *__restrict__ C,
                                                                        const double *__restrict__ v, int mode, double shift, TO *__restrict__ out,
                                                                        const double *__restrict__ yadd = nullptr, double alpha = 0.0, double beta = 0.0,
                                                                        double *__restrict__ normpart = nullptr)
{
    constexpr int R = KSFD_J3L_ROWS, NA = 4 + NL;
    extern __shared__ double2 j3l_lds[];                     // [2][R + 4][NA][64]
    const int lane = threadIdx.x & (KSFD_WAVE - 1), wv = threadIdx.x >> 6;
    if (normpart && lane == 0) normpart[(long long)blockIdx.x * R + wv] = 0.0;
    double nacc = 0.0;
    const long long bid = ksfd_xcd_remap(blockIdx.x, S.nblocks);
    const long long nb_valid = (long long)S.nstrips * S.nygrp * S.nzseg;
    if (bid >= nb_valid) return;                             // whole block
    const int strip = (int)(bid % S.nstrips);
    const long long ygrp = (bid / S.nstrips) % S.nygrp, zs = bid / ((long long)S.nstrips * S.nygrp);
    const long long y0 = ygrp * R, y = y0 + wv;              // ny % R == 0: every wave has a row
    const long long half = G.nx >> 1;
    const long long xs = 2 * ((long long)strip * half / S.nstrips), xe = 2 * ((long long)(strip + 1) * half / S.nstrips);
    long long c0 = (xs - 2 + 2 * lane) % G.nx;
    if (c0 < 0) c0 += G.nx;
    const bool store = lane >= 1 && lane <= (int)((xe - xs) >> 1);
    const long long k0 = zs * S.zseg, k1 = k0 + S.zseg < G.sloc ? k0 + S.zseg : G.sloc;
    const bool up = (zs & 1) == 0;
    auto kmap = [&](long long q) { return up ? q : (k0 + k1 - 1 - q); };
    const long long rowc = y * G.nx + c0;
    long long hrowc = 0;                                     // waves 0..3: the halo row this wave brings in
    if (wv < 4) {
        long long hy = wv < 2 ? y0 - 2 + wv : y0 + R + (wv - 2);
        hy %= G.ny;
        if (hy < 0) hy += G.ny;
        hrowc = hy * G.nx + c0;
    }
    const double *Cr = C, *Cg = C + G.plane, *Cgr = C + 2 * G.plane;
    double rw[5][2], gw[5][2], vw[5][2], ew[5][2], zw[NL][5][2];
    double nr[2], ng[2], nv[2], ne[2], nz[NL][2];
    double2 hr, hg, hv, he, hz[NL];                          // halo row of the next centre plane (waves 0..3)
    auto row_load = [&](long long o, double2 &a, double2 &b, double2 &w, double2 &e, double2 (&z)[NL]) {
        a = ksfd_ld2(Cr + o); b = ksfd_ld2(Cg + o); w = ksfd_ld2(v + o);
        const double2 gr = ksfd_ld2(Cgr + o);
        e = make_double2(gr.x * w.x, gr.y * w.y);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const double2 gu = ksfd_ld2(C + (long long)(3 + l) * G.plane + o);
            z[l] = ksfd_ld2(v + (long long)(l + 1) * G.plane + o);
            e.x += gu.x * z[l].x; e.y += gu.y * z[l].y;
        }
    };
    auto load_plane = [&](long long k) {
        double2 a, b, w, e, z[NL];
        row_load(ksfd_planeoff(G, k) + rowc, a, b, w, e, z);
        nr[0] = a.x; nr[1] = a.y; ng[0] = b.x; ng[1] = b.y; nv[0] = w.x; nv[1] = w.y; ne[0] = e.x; ne[1] = e.y;
#pragma unroll
        for (int l = 0; l < NL; l++) { nz[l][0] = z[l].x; nz[l][1] = z[l].y; }
    };
    auto push = [&]() {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                rw[s][e] = rw[s + 1][e]; gw[s][e] = gw[s + 1][e]; vw[s][e] = vw[s + 1][e]; ew[s][e] = ew[s + 1][e];
#pragma unroll
                for (int l = 0; l < NL; l++) zw[l][s][e] = zw[l][s + 1][e];
            }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            rw[4][e] = nr[e]; gw[4][e] = ng[e]; vw[4][e] = nv[e]; ew[4][e] = ne[e];
#pragma unroll
            for (int l = 0; l < NL; l++) zw[l][4][e] = nz[l][e];
        }
    };
    auto slot = [&](int buf, int row, int a) { return j3l_lds + ((long long)((buf * (R + 4) + row) * NA + a) << 6) + lane; };
    for (int q = -2; q <= 1; q++) { load_plane(kmap(k0 + q)); push(); }
    load_plane(kmap(k0 + 2));
    if (wv < 4) row_load(ksfd_planeoff(G, kmap(k0)) + hrowc, hr, hg, hv, he, hz);
    for (long long k = k0; k < k1; k++) {
        push();
        if (k + 1 < k1) load_plane(kmap(k + 3));
        const long long kc = kmap(k);
        const int buf = (int)((k - k0) & 1);
        // centre plane of this wave's row, and of its halo row, to the LDS
        *slot(buf, wv, 0) = make_double2(rw[2][0], rw[2][1]);
        *slot(buf, wv, 1) = make_double2(gw[2][0], gw[2][1]);
        *slot(buf, wv, 2) = make_double2(vw[2][0], vw[2][1]);
        *slot(buf, wv, 3) = make_double2(ew[2][0], ew[2][1]);
#pragma unroll
        for (int l = 0; l < NL; l++) *slot(buf, wv, 4 + l) = make_double2(zw[l][2][0], zw[l][2][1]);
        if (wv < 4) {
            *slot(buf, R + wv, 0) = hr; *slot(buf, R + wv, 1) = hg; *slot(buf, R + wv, 2) = hv; *slot(buf, R + wv, 3) = he;
#pragma unroll
            for (int l = 0; l < NL; l++) *slot(buf, R + wv, 4 + l) = hz[l];
            if (k + 1 < k1) row_load(ksfd_planeoff(G, kmap(k + 1)) + hrowc, hr, hg, hv, he, hz);
        }
        // the added vector of modes 2/3 is needed only at the store: issue its loads now
        double2 yv_add[NL + 1];
        if ((mode == 2 || mode == 3) && store) {
            const long long oy = (long long)G.ng * G.inner + kc * G.nx * G.ny + rowc;
#pragma unroll
            for (int c = 0; c <= NL; c++) yv_add[c] = ksfd_ld2(yadd + (long long)c * G.plane + oy);
        }
        __syncthreads();
        // rows y-2, y-1, y+1, y+2 of the centre plane: LDS rows of the block, or the halo rows behind them
        int nbr[4];
        {
            const int dm[4] = { -2, -1, 1, 2 };
#pragma unroll
            for (int q = 0; q < 4; q++) { const int rr = wv + dm[q]; nbr[q] = rr < 0 ? R + 2 + rr : (rr >= R ? R + 2 + (rr - R) : rr); }
        }
        const double ih0 = P.inv_h[0], ih1 = P.inv_h[1], ih2 = P.inv_h[2];
        // one array at a time: four neighbour values from the LDS, the y- and z-derivatives, done
        double yr1[2], yg1[2], yv1[2], ye1[2], yg2[2], ye2[2];
        {
            double2 t[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = *slot(buf, nbr[q], 0);
            yr1[0] = KSFD_D1(t[0].x, t[1].x, t[2].x, t[3].x) * ih1; yr1[1] = KSFD_D1(t[0].y, t[1].y, t[2].y, t[3].y) * ih1;
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = *slot(buf, nbr[q], 1);
            yg1[0] = KSFD_D1(t[0].x, t[1].x, t[2].x, t[3].x) * ih1; yg1[1] = KSFD_D1(t[0].y, t[1].y, t[2].y, t[3].y) * ih1;
            yg2[0] = KSFD_D2(t[0].x, t[1].x, gw[2][0], t[2].x, t[3].x) * P.inv_h2[1]; yg2[1] = KSFD_D2(t[0].y, t[1].y, gw[2][1], t[2].y, t[3].y) * P.inv_h2[1];
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = *slot(buf, nbr[q], 2);
            yv1[0] = KSFD_D1(t[0].x, t[1].x, t[2].x, t[3].x) * ih1; yv1[1] = KSFD_D1(t[0].y, t[1].y, t[2].y, t[3].y) * ih1;
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = *slot(buf, nbr[q], 3);
            ye1[0] = KSFD_D1(t[0].x, t[1].x, t[2].x, t[3].x) * ih1; ye1[1] = KSFD_D1(t[0].y, t[1].y, t[2].y, t[3].y) * ih1;
            ye2[0] = KSFD_D2(t[0].x, t[1].x, ew[2][0], t[2].x, t[3].x) * P.inv_h2[1]; ye2[1] = KSFD_D2(t[0].y, t[1].y, ew[2][1], t[2].y, t[3].y) * P.inv_h2[1];
        }
        const KX xr = ksfd_xnb(rw[2][0], rw[2][1]), xg = ksfd_xnb(gw[2][0], gw[2][1]);
        const KX xv = ksfd_xnb(vw[2][0], vw[2][1]), xe_ = ksfd_xnb(ew[2][0], ew[2][1]);
        double d1r[2], d1g[2], d1v[2], d1e[2], d2g[2], d2e[2];
        ksfd_dx(rw[2][0], rw[2][1], xr, d1r[0], d1r[1]);
        ksfd_dx(gw[2][0], gw[2][1], xg, d1g[0], d1g[1]);
        ksfd_dx(vw[2][0], vw[2][1], xv, d1v[0], d1v[1]);
        ksfd_dx(ew[2][0], ew[2][1], xe_, d1e[0], d1e[1]);
        ksfd_dxx(gw[2][0], gw[2][1], xg, d2g[0], d2g[1]);
        ksfd_dxx(ew[2][0], ew[2][1], xe_, d2e[0], d2e[1]);
        double res[NL + 1][2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double zr1 = KSFD_D1(rw[0][e], rw[1][e], rw[3][e], rw[4][e]) * ih2;
            const double zg1 = KSFD_D1(gw[0][e], gw[1][e], gw[3][e], gw[4][e]) * ih2;
            const double zv1 = KSFD_D1(vw[0][e], vw[1][e], vw[3][e], vw[4][e]) * ih2;
            const double ze1 = KSFD_D1(ew[0][e], ew[1][e], ew[3][e], ew[4][e]) * ih2;
            const double zg2 = KSFD_D2(gw[0][e], gw[1][e], gw[2][e], gw[3][e], gw[4][e]) * P.inv_h2[2];
            const double ze2 = KSFD_D2(ew[0][e], ew[1][e], ew[2][e], ew[3][e], ew[4][e]) * P.inv_h2[2];
            const double lapG = d2g[e] * P.inv_h2[0] + yg2[e] + zg2;
            const double lapE = d2e[e] * P.inv_h2[0] + ye2[e] + ze2;
            const double jr = (d1v[e] * ih0) * (d1g[e] * ih0) + (d1r[e] * ih0) * (d1e[e] * ih0) + yv1[e] * yg1[e] + yr1[e] * ye1[e] +
                              zv1 * zg1 + zr1 * ze1 + vw[2][e] * lapG + rw[2][e] * lapE;
            res[0][e] = mode ? shift * vw[2][e] - jr : jr;
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            double2 t[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t[q] = *slot(buf, nbr[q], 4 + l);
            const KX xz = ksfd_xnb(zw[l][2][0], zw[l][2][1]);
            double d2z[2];
            ksfd_dxx(zw[l][2][0], zw[l][2][1], xz, d2z[0], d2z[1]);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double t0 = e ? t[0].y : t[0].x, t1 = e ? t[1].y : t[1].x, t2 = e ? t[2].y : t[2].x, t3 = e ? t[3].y : t[3].x;
                const double lap = d2z[e] * P.inv_h2[0] + KSFD_D2(t0, t1, zw[l][2][e], t2, t3) * P.inv_h2[1] +
                                   KSFD_D2(zw[l][0][e], zw[l][1][e], zw[l][2][e], zw[l][3][e], zw[l][4][e]) * P.inv_h2[2];
                const double ju = -P.lig_gamma[l] * zw[l][2][e] + P.lig_s[l] * vw[2][e] + P.lig_D[l] * lap;
                res[l + 1][e] = mode ? shift * zw[l][2][e] - ju : ju;
            }
        }
        if (store) {
            const long long o = (long long)G.ng * G.inner + kc * G.nx * G.ny + rowc;
#pragma unroll
            for (int c = 0; c <= NL; c++) {
                double a = res[c][0], b = res[c][1];
                if (mode == 4) {
                    const double c0v = c == 0 ? vw[2][0] : zw[c > 0 ? c - 1 : 0][2][0], c1v = c == 0 ? vw[2][1] : zw[c > 0 ? c - 1 : 0][2][1];
                    a = alpha * c0v + beta * a; b = alpha * c1v + beta * b;
                } else if (mode >= 2) {
                    const double2 yy = yv_add[c];
                    if (mode == 2) { a = yy.x - a; b = yy.y - b; } else { a = alpha * yy.x + beta * a; b = alpha * yy.y + beta * b; }
                }
                if constexpr (sizeof(TO) == 4) *reinterpret_cast<float2 *>(out + (long long)c * G.plane + o) = make_float2((float)a, (float)b);
                else ksfd_st2(out + (long long)c * G.plane + o, a, b);
                nacc += a * a + b * b;
            }
        }
    }
    if (normpart) {
        nacc = ksfd_wave_sum(nacc);
        if (lane == 0) normpart[(long long)blockIdx.x * R + wv] = nacc;
    }
}


// ---------------------------------------------------------------------------------------------
// 3-D RHS, z-marching.  Two passes by design: G needs two logs and a tanh per point (fp64), and a single-pass kernel would
// have to evaluate it for the y-halo rows of every block as well (1.5x at 8 rows per block) -- on this VALU-bound evaluation
// that costs more than writing and re-reading one G plane.  Pass 1 (k_gfield_comb) forms the stage argument
// z = u + sum_j a_j Y_j, stores it and G(z) over the slab incl. ghost units; pass 2 (this kernel) is the 13-point star with
// the same strip / register-window / wave-shuffle scheme as k_jvp3d_frozen (z-neighbours in 5-plane windows, x-neighbours by
// DPP shifts, the 4 y-neighbour rows of the centre plane by 16-B loads that hit L1/L2), and adds -sum_j c_j Y_j / h and the
// sources at the store, so a 3-D step has no separate stage-vector passes either.
// ---------------------------------------------------------------------------------------------
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_gfield_comb(KGeom G, KPhys P, const double *__restrict__ u, KComb cmb,
                                                            double *__restrict__ zout, double *__restrict__ Gout)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < G.plane; e += stride) {
        double val[NL + 1];
#pragma unroll
        for (int c = 0; c <= NL; c++) val[c] = u[(long long)c * G.plane + e];
        for (int j = 0; j < cmb.nin; j++) {
            const double a = cmb.ain[j];
#pragma unroll
            for (int c = 0; c <= NL; c++) val[c] += a * cmb.yin[j][(long long)c * G.plane + e];
        }
        if (zout) {
#pragma unroll
            for (int c = 0; c <= NL; c++) zout[(long long)c * G.plane + e] = val[c];
        }
        double U[NL], GU[NL], g, gr;
        const double rho = ksfd_clamp(val[0], P.rhomin);
#pragma unroll
        for (int l = 0; l < NL; l++) U[l] = ksfd_clamp(val[l + 1], P.Umin);
        ksfd_G<NL, false>(P, rho, U, g, gr, GU);
        Gout[e] = g;
    }
}

template <int NL, int ROWS = 4>
__global__ void __launch_bounds__(ROWS * KSFD_WAVE) k_rhs3d_strip(KGeom G, KPhys P, K3D S, const double *__restrict__ u,
                                                            const double *__restrict__ Gb, KSrc src, double *__restrict__ out,
                                                            KComb cmb)
{
    const int lane = threadIdx.x & (KSFD_WAVE - 1), wv = threadIdx.x >> 6;
    const long long bid = ksfd_xcd_remap(blockIdx.x, S.nblocks);
    const long long nb_valid = (long long)S.nstrips * S.nygrp * S.nzseg;
    if (bid >= nb_valid) return;
    const int strip = (int)(bid % S.nstrips);
    const long long ygrp = (bid / S.nstrips) % S.nygrp, zs = bid / ((long long)S.nstrips * S.nygrp);
    const long long y = ygrp * ROWS + wv;
    if (y >= G.ny) return;                                   // whole wave; no block barrier in this kernel
    const long long half = G.nx >> 1;
    const long long xs = 2 * ((long long)strip * half / S.nstrips), xe = 2 * ((long long)(strip + 1) * half / S.nstrips);
    long long c0 = (xs - 2 + 2 * lane) % G.nx;
    if (c0 < 0) c0 += G.nx;
    const bool store = lane >= 1 && lane <= (int)((xe - xs) >> 1);
    const long long k0 = zs * S.zseg, k1 = k0 + S.zseg < G.sloc ? k0 + S.zseg : G.sloc;
    const bool up = (zs & 1) == 0;
    auto kmap = [&](long long q) { return up ? q : (k0 + k1 - 1 - q); };
    const long long rowc = y * G.nx + c0;
    long long yo[4];
    {
        const int dm[4] = { -2, -1, 1, 2 };
#pragma unroll
        for (int q = 0; q < 4; q++) { long long yy = (y + dm[q]) % G.ny; if (yy < 0) yy += G.ny; yo[q] = yy * G.nx + c0; }
    }
    double rw[5][2], gw[5][2], uw[NL][5][2];
    double nr[2], ng[2], nu[NL][2];
    auto load_plane = [&](long long k) {
        const long long o = ksfd_planeoff(G, k) + rowc;
        const double2 a = ksfd_ld2(u + o), b = ksfd_ld2(Gb + o);
        nr[0] = a.x; nr[1] = a.y; ng[0] = b.x; ng[1] = b.y;
#pragma unroll
        for (int l = 0; l < NL; l++) { const double2 z = ksfd_ld2(u + (long long)(l + 1) * G.plane + o); nu[l][0] = z.x; nu[l][1] = z.y; }
    };
    auto push = [&]() {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                rw[s][e] = rw[s + 1][e]; gw[s][e] = gw[s + 1][e];
#pragma unroll
                for (int l = 0; l < NL; l++) uw[l][s][e] = uw[l][s + 1][e];
            }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            rw[4][e] = ksfd_clamp(nr[e], P.rhomin); gw[4][e] = ng[e];
#pragma unroll
            for (int l = 0; l < NL; l++) uw[l][4][e] = ksfd_clamp(nu[l][e], P.Umin);
        }
    };
    for (int q = -2; q <= 1; q++) { load_plane(kmap(k0 + q)); push(); }
    load_plane(kmap(k0 + 2));
    for (long long k = k0; k < k1; k++) {
        if (S.sync) __syncthreads();
        push();
        if (k + 1 < k1) load_plane(kmap(k + 3));
        const long long kc = kmap(k);
        const long long po = ksfd_planeoff(G, kc);
        double2 yr[4], yg[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { yr[q] = ksfd_ld2(u + po + yo[q]); yg[q] = ksfd_ld2(Gb + po + yo[q]); }
        const KX xr = ksfd_xnb(rw[2][0], rw[2][1]), xg = ksfd_xnb(gw[2][0], gw[2][1]);
        double d1r[2], d1g[2], d2g[2];
        ksfd_dx(rw[2][0], rw[2][1], xr, d1r[0], d1r[1]);
        ksfd_dx(gw[2][0], gw[2][1], xg, d1g[0], d1g[1]);
        ksfd_dxx(gw[2][0], gw[2][1], xg, d2g[0], d2g[1]);
        double res[NL + 1][2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double ih0 = P.inv_h[0], ih1 = P.inv_h[1], ih2 = P.inv_h[2];
#define KY(arr, q) (e ? arr[q].y : arr[q].x)
#define KYC(arr, q) ksfd_clamp(KY(arr, q), P.rhomin)
            const double yr1 = KSFD_D1(KYC(yr, 0), KYC(yr, 1), KYC(yr, 2), KYC(yr, 3)) * ih1;
            const double yg1 = KSFD_D1(KY(yg, 0), KY(yg, 1), KY(yg, 2), KY(yg, 3)) * ih1;
            const double yg2 = KSFD_D2(KY(yg, 0), KY(yg, 1), gw[2][e], KY(yg, 2), KY(yg, 3)) * P.inv_h2[1];
            const double zr1 = KSFD_D1(rw[0][e], rw[1][e], rw[3][e], rw[4][e]) * ih2;
            const double zg1 = KSFD_D1(gw[0][e], gw[1][e], gw[3][e], gw[4][e]) * ih2;
            const double zg2 = KSFD_D2(gw[0][e], gw[1][e], gw[2][e], gw[3][e], gw[4][e]) * P.inv_h2[2];
            res[0][e] = (d1r[e] * ih0) * (d1g[e] * ih0) + yr1 * yg1 + zr1 * zg1 + rw[2][e] * (d2g[e] * P.inv_h2[0] + yg2 + zg2);
#undef KYC
        }
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const double *ul = u + (long long)(l + 1) * G.plane + po;
            double2 yz[4];
#pragma unroll
            for (int q = 0; q < 4; q++) yz[q] = ksfd_ld2(ul + yo[q]);
            const KX xz = ksfd_xnb(uw[l][2][0], uw[l][2][1]);
            double d2z[2];
            ksfd_dxx(uw[l][2][0], uw[l][2][1], xz, d2z[0], d2z[1]);
#pragma unroll
            for (int e = 0; e < 2; e++) {
#define KYU(q) ksfd_clamp(KY(yz, q), P.Umin)
                const double lap = d2z[e] * P.inv_h2[0] + KSFD_D2(KYU(0), KYU(1), uw[l][2][e], KYU(2), KYU(3)) * P.inv_h2[1] +
                                   KSFD_D2(uw[l][0][e], uw[l][1][e], uw[l][2][e], uw[l][3][e], uw[l][4][e]) * P.inv_h2[2];
                res[l + 1][e] = -P.lig_gamma[l] * uw[l][2][e] + P.lig_s[l] * rw[2][e] + P.lig_D[l] * lap;
#undef KYU
            }
        }
#undef KY
        if (store) {
            const long long pi = kc * G.nx * G.ny + rowc;              // dense interior index
            const long long o = (long long)G.ng * G.inner + pi;
#pragma unroll
            for (int c = 0; c <= NL; c++) {
                double a = res[c][0], b = res[c][1];
                if (src.p[c]) { const double2 sv = ksfd_ld2(src.p[c] + pi); a += sv.x; b += sv.y; }
                for (int j = 0; j < cmb.nout; j++) {
                    const double2 yv = ksfd_ld2(cmb.yout[j] + (long long)c * G.plane + o);
                    a += cmb.aout[j] * yv.x; b += cmb.aout[j] * yv.y;
                }
                ksfd_st2(out + (long long)c * G.plane + o, a, b);
            }
        }
    }
}
