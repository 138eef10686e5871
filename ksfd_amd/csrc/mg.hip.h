// Geometric multigrid preconditioner for the stage systems (shift*I - J) y = b of the Rosenbrock-W step.
//
// Stands in for what `-pc_type lu -pc_factor_mat_solver_type mumps` (options84:58-60) gives the reference:
// a solve whose cost does not blow up when h*gamma*lambda_max(J) >> 1 (late-time steps h ~ 1..1e4).
// Matrix-free and rediscretised: level l uses the SAME analytic Jacobian-action kernels with the frozen
// coefficient planes [rho, G, G_rho, G_U] full-weighted onto a grid of spacing 2^l h.  Smoother: Chebyshev
// iteration preconditioned by the inverse of the point-block (F x F) diagonal of A_l.  Vertex-centred full
// coarsening on the periodic box, full-weighting restriction, bilinear prolongation, V(nu,nu) cycle, fixed
// polynomial smoothing on the coarsest grid -- every piece is a fixed linear operator, so plain
// right-preconditioned GMRES applies.  1-D, 2-D and 3-D, single rank or slab ranks.
#pragma once
#include "stencil.hip.h"

// coarse(I,J) = sum_{a,b in -1..1} w_a w_b fine(2I+a, 2J+b), w = (1/4, 1/2, 1/4).  x is periodic; the row axis either
// wraps (one rank) or reads the ghost rows at local index -1 / nyf (slab ranks: the fine vector's halo must be current
// and the slab must start on an even global row).  foff/coff = offset of local row 0 inside a plane (ng*nx).
// TF / TK: storage type of the fine / coarse vector (the V cycle runs its level vectors in fp32 when the solve tolerance allows: mg_host.hip.h)
template <typename TF = double, typename TK = double>
__global__ void __launch_bounds__(KSFD_BLOCK) k_restrict2d(int nplanes, long long nxf, long long nyf, int wrap,
                                                           const TF *__restrict__ fine, long long fplane, long long foff,
                                                           TK *__restrict__ coarse, long long cplane, long long coff)
{
    const long long nxc = nxf >> 1, nyc = nyf >> 1, nc = nxc * nyc;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < nc; p += stride) {
        const long long I = p % nxc, J = p / nxc;
        const long long i0 = 2 * I, j0 = 2 * J;
        const long long im = (i0 + nxf - 1) % nxf, ip = (i0 + 1) % nxf;
        const long long jm = wrap ? (j0 + nyf - 1) % nyf : j0 - 1, jp = wrap ? (j0 + 1) % nyf : j0 + 1;
        for (int c = 0; c < nplanes; c++) {
            const TF *f = fine + (long long)c * fplane + foff;
            const double s = 0.25 * (double)f[i0 + nxf * j0] +
                             0.125 * ((double)f[im + nxf * j0] + (double)f[ip + nxf * j0] + (double)f[i0 + nxf * jm] + (double)f[i0 + nxf * jp]) +
                             0.0625 * ((double)f[im + nxf * jm] + (double)f[ip + nxf * jm] + (double)f[im + nxf * jp] + (double)f[ip + nxf * jp]);
            coarse[(long long)c * cplane + coff + p] = (TK)s;
        }
    }
}

// fine += P coarse (bilinear interpolation); row axis wraps or reads the coarse ghost row nyc (slab ranks)
template <typename TK = double, typename TF = double>
__global__ void __launch_bounds__(KSFD_BLOCK) k_prolong_add2d(int nplanes, long long nxf, long long nyf, int wrap,
                                                              const TK *__restrict__ coarse, long long cplane, long long coff,
                                                              TF *__restrict__ fine, long long fplane, long long foff)
{
    const long long nxc = nxf >> 1, nyc = nyf >> 1, nf = nxf * nyf;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < nf; p += stride) {
        const long long i = p % nxf, j = p / nxf;
        const long long I = i >> 1, J = j >> 1;
        const long long I1 = (i & 1) ? (I + 1) % nxc : I;
        const long long J1 = (j & 1) ? (wrap ? (J + 1) % nyc : J + 1) : J;
        for (int c = 0; c < nplanes; c++) {
            const TK *q = coarse + (long long)c * cplane + coff;
            const double v = 0.25 * ((double)q[I + nxc * J] + (double)q[I1 + nxc * J] + (double)q[I + nxc * J1] + (double)q[I1 + nxc * J1]);
            fine[(long long)c * fplane + foff + p] = (TF)((double)fine[(long long)c * fplane + foff + p] + v);
        }
    }
}

// 1-D transfer operators (the slab axis is x itself): weights (1/4, 1/2, 1/4) and linear interpolation; the axis wraps
// (one rank) or reads the ghost points at local index -1 / nf (slab ranks).
__global__ void __launch_bounds__(KSFD_BLOCK) k_restrict1d(int nplanes, long long nf, int wrap,
                                                           const double *__restrict__ fine, long long fplane, long long foff,
                                                           double *__restrict__ coarse, long long cplane, long long coff)
{
    const long long nc = nf >> 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < nc; p += stride) {
        const long long i0 = 2 * p;
        const long long im = wrap ? (i0 + nf - 1) % nf : i0 - 1, ip = wrap ? (i0 + 1) % nf : i0 + 1;
        for (int c = 0; c < nplanes; c++) {
            const double *f = fine + (long long)c * fplane + foff;
            coarse[(long long)c * cplane + coff + p] = 0.5 * f[i0] + 0.25 * (f[im] + f[ip]);
        }
    }
}
__global__ void __launch_bounds__(KSFD_BLOCK) k_prolong_add1d(int nplanes, long long nf, int wrap,
                                                              const double *__restrict__ coarse, long long cplane, long long coff,
                                                              double *__restrict__ fine, long long fplane, long long foff)
{
    const long long nc = nf >> 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < nf; p += stride) {
        const long long I = p >> 1;
        const long long I1 = (p & 1) ? (wrap ? (I + 1) % nc : I + 1) : I;
        for (int c = 0; c < nplanes; c++) {
            const double *q = coarse + (long long)c * cplane + coff;
            fine[(long long)c * fplane + foff + p] += 0.5 * (q[I] + q[I1]);
        }
    }
}

// 3-D transfer operators (slab axis = z): 27-point full weighting and trilinear interpolation.  x and y wrap; z wraps
// (one rank) or reads ghost planes -1 / nzf (slab ranks).  foff/coff = offset of local plane 0 inside a field plane.
__global__ void __launch_bounds__(KSFD_BLOCK) k_restrict3d(int nplanes, long long nxf, long long nyf, long long nzf, int wrap,
                                                           const double *__restrict__ fine, long long fplane, long long foff,
                                                           double *__restrict__ coarse, long long cplane, long long coff)
{
    const long long nxc = nxf >> 1, nyc = nyf >> 1, nzc = nzf >> 1, nc = nxc * nyc * nzc;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < nc; p += stride) {
        const long long I = p % nxc, J = (p / nxc) % nyc, K = p / (nxc * nyc);
        long long xi[3], yj[3], zk[3];
        xi[1] = 2 * I; xi[0] = (xi[1] + nxf - 1) % nxf; xi[2] = (xi[1] + 1) % nxf;
        yj[1] = 2 * J; yj[0] = (yj[1] + nyf - 1) % nyf; yj[2] = (yj[1] + 1) % nyf;
        zk[1] = 2 * K; zk[0] = wrap ? (zk[1] + nzf - 1) % nzf : zk[1] - 1; zk[2] = wrap ? (zk[1] + 1) % nzf : zk[1] + 1;
        const double w[3] = { 0.25, 0.5, 0.25 };
        for (int c = 0; c < nplanes; c++) {
            const double *f = fine + (long long)c * fplane + foff;
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++)
#pragma unroll
                    for (int d = 0; d < 3; d++) s += w[a] * w[b] * w[d] * f[xi[d] + nxf * (yj[b] + nyf * zk[a])];
            coarse[(long long)c * cplane + coff + p] = s;
        }
    }
}

__global__ void __launch_bounds__(KSFD_BLOCK) k_prolong_add3d(int nplanes, long long nxf, long long nyf, long long nzf, int wrap,
                                                              const double *__restrict__ coarse, long long cplane, long long coff,
                                                              double *__restrict__ fine, long long fplane, long long foff)
{
    const long long nxc = nxf >> 1, nyc = nyf >> 1, nzc = nzf >> 1, nf = nxf * nyf * nzf;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < nf; p += stride) {
        const long long i = p % nxf, j = (p / nxf) % nyf, k = p / (nxf * nyf);
        const long long I = i >> 1, J = j >> 1, K = k >> 1;
        const long long I1 = (i & 1) ? (I + 1) % nxc : I, J1 = (j & 1) ? (J + 1) % nyc : J;
        const long long K1 = (k & 1) ? (wrap ? (K + 1) % nzc : K + 1) : K;
        for (int c = 0; c < nplanes; c++) {
            const double *q = coarse + (long long)c * cplane + coff;
            const double v = 0.125 * (q[I + nxc * (J + nyc * K)] + q[I1 + nxc * (J + nyc * K)] + q[I + nxc * (J1 + nyc * K)] +
                                      q[I1 + nxc * (J1 + nyc * K)] + q[I + nxc * (J + nyc * K1)] + q[I1 + nxc * (J + nyc * K1)] +
                                      q[I + nxc * (J1 + nyc * K1)] + q[I1 + nxc * (J1 + nyc * K1)]);
            fine[(long long)c * fplane + foff + p] += v;
        }
    }
}

// Inverse of the point-block diagonal of A = shift*I - J (F x F per point, row-major planes dinv[(r*F+c)*plane], stored in
// fp32: it only scales the smoother of a preconditioner, and it is read three times per smoothing sweep):
//   D_rr = shift - (lapG + rho*G_rho*c2),  D_rUl = -rho*G_Ul*c2,  D_Ulr = -s_l,  D_UlUl = shift + gamma_l - D_l*c2
// with c2 = sum_a (-30/12)/h_a^2 the centre weight of the 4th-order Laplacian.
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_blockdiag_inv(KGeom G, KPhys P, const double *__restrict__ C, double shift,
                                                              float *__restrict__ dinv)
{
    constexpr int F = NL + 1;
    const double *Gb = C + G.plane;
    double c2 = 0.0;
    for (int a = 0; a < G.dim; a++) c2 += -2.5 * P.inv_h2[a];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < G.nloc; p += stride) {
        long long i, j, k;
        ksfd_decode(G, p, i, j, k);
        double lapG = 0.0;
        for (int a = 0; a < G.dim; a++) {
            KNbr n = ksfd_nbr(G, a, i, j, k);
            lapG += KSFD_D2(Gb[n.m2], Gb[n.m1], Gb[n.c], Gb[n.p1], Gb[n.p2]) * P.inv_h2[a];
        }
        const long long o = (long long)G.ng * G.inner + p;       // owned point inside a (ghosted) plane
        const double rho = C[o], gr = C[2 * G.plane + o];
        double M[F][F], Inv[F][F];
#pragma unroll
        for (int r = 0; r < F; r++)
#pragma unroll
            for (int c = 0; c < F; c++) { M[r][c] = 0.0; Inv[r][c] = (r == c) ? 1.0 : 0.0; }
        M[0][0] = shift - (lapG + rho * gr * c2);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            M[0][l + 1] = -rho * C[(long long)(3 + l) * G.plane + o] * c2;
            M[l + 1][0] = -P.lig_s[l];
            M[l + 1][l + 1] = shift + P.lig_gamma[l] - P.lig_D[l] * c2;
        }
        // Gauss-Jordan without pivoting (the blocks are strongly diagonally dominated by shift + diffusion)
#pragma unroll
        for (int q = 0; q < F; q++) {
            const double ip = 1.0 / M[q][q];
#pragma unroll
            for (int c = 0; c < F; c++) { M[q][c] *= ip; Inv[q][c] *= ip; }
#pragma unroll
            for (int r = 0; r < F; r++) {
                if (r == q) continue;
                const double f = M[r][q];
#pragma unroll
                for (int c = 0; c < F; c++) { M[r][c] -= f * M[q][c]; Inv[r][c] -= f * Inv[q][c]; }
            }
        }
#pragma unroll
        for (int r = 0; r < F; r++)
#pragma unroll
            for (int c = 0; c < F; c++) dinv[(long long)(r * F + c) * G.plane + o] = (float)Inv[r][c];
    }
}

// z = scale * Dinv r   (and z2 = the same values when z2 != NULL: first Chebyshev sweep from a zero guess, x = d)
// rcopy != NULL: r itself, in the storage type of z (entry of an fp32 V cycle: the fp64 right-hand side is read once)
template <int NL, typename TR = double, typename TZ = double>
__global__ void __launch_bounds__(KSFD_BLOCK) k_dinv_apply(long long n, long long plane, const float *__restrict__ dinv,
                                                           const TR *__restrict__ r, double scale, TZ *__restrict__ z,
                                                           TZ *__restrict__ z2 = nullptr, TZ *__restrict__ rcopy = nullptr)
{
    constexpr int F = NL + 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        double rv[F];
#pragma unroll
        for (int c = 0; c < F; c++) {
            rv[c] = (double)r[(long long)c * plane + p];
            if (rcopy) rcopy[(long long)c * plane + p] = (TZ)rv[c];
        }
#pragma unroll
        for (int a = 0; a < F; a++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < F; c++) s += dinv[(long long)(a * F + c) * plane + p] * rv[c];
            z[(long long)a * plane + p] = (TZ)(scale * s);
            if (z2) z2[(long long)a * plane + p] = (TZ)(scale * s);
        }
    }
}

// Last Chebyshev sweep: x += d_old + d_new with d_new = c1*d_old + c2*Dinv (r - Ad); r and d are dead afterwards.
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_cheb_last(long long n, long long plane, const float *__restrict__ dinv,
                                                          double *__restrict__ x, const double *__restrict__ r,
                                                          const double *__restrict__ d, const double *__restrict__ Ad,
                                                          double c1, double c2, int x_has_d)
{
    constexpr int F = NL + 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        double rv[F], dv[F];
#pragma unroll
        for (int c = 0; c < F; c++) {
            const long long o = (long long)c * plane + p;
            dv[c] = d[o];
            rv[c] = r[o] - Ad[o];
        }
#pragma unroll
        for (int a = 0; a < F; a++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < F; c++) s += dinv[(long long)(a * F + c) * plane + p] * rv[c];
            const long long o = (long long)a * plane + p;
            // x_has_d: x already contains d_old (zero-guess start wrote x = d)
            x[o] += (x_has_d ? 0.0 : dv[a]) + c1 * dv[a] + c2 * s;
        }
    }
}

// One Chebyshev recurrence step after Ad = A d is known:
//   x += d ; r -= Ad ; d = c1*d + c2*(Dinv r)
template <int NL>
__global__ void __launch_bounds__(KSFD_BLOCK) k_cheb_step(long long n, long long plane, const float *__restrict__ dinv,
                                                          double *__restrict__ x, double *__restrict__ r, double *__restrict__ d,
                                                          const double *__restrict__ Ad, double c1, double c2)
{
    constexpr int F = NL + 1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        double rv[F], dv[F];
#pragma unroll
        for (int c = 0; c < F; c++) {
            const long long o = (long long)c * plane + p;
            dv[c] = d[o];
            x[o] += dv[c];
            rv[c] = r[o] - Ad[o];
            r[o] = rv[c];
        }
#pragma unroll
        for (int a = 0; a < F; a++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < F; c++) s += dinv[(long long)(a * F + c) * plane + p] * rv[c];
            d[(long long)a * plane + p] = c1 * dv[a] + c2 * s;
        }
    }
}

// deterministic pseudo-random fill for the power iteration
__global__ void __launch_bounds__(KSFD_BLOCK) k_hash_fill(long long n, double *__restrict__ v)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        unsigned long long z = (unsigned long long)p * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
        v[p] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}

// per-block max over points of 1/(shift * [Dinv]_00): ~ largest diagonal-to-shift ratio, i.e. an estimate of
// lambda_max/lambda_min of Dinv*A on a grid whose smoothest modes see only the shift
__global__ void __launch_bounds__(KSFD_BLOCK) k_ratio_est(long long n, const float *__restrict__ dinv00, double shift,
                                                          double *__restrict__ part)
{
    __shared__ double red[KSFD_BLOCK / KSFD_WAVE];
    double m = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) m = fmax(m, 1.0 / fabs(shift * dinv00[p]));
    m = ksfd_wave_max(m);
    if ((threadIdx.x & (KSFD_WAVE - 1)) == 0) red[threadIdx.x / KSFD_WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < KSFD_BLOCK / KSFD_WAVE; q++) t = fmax(t, red[q]);
        part[blockIdx.x] = t;
    }
}
