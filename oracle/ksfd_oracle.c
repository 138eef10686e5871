/*
 * oracle/ksfd_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * A plain-C, CPU restatement of the reference's algorithm for the hot path
 * (leonavery/KSFD: RHS stencil, analytic Jacobian action, CFL velocity, and the
 * PETSc TS ROSW step the reference drives through petsc4py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the shipped package never does.
 *
 * Parity status
 *   - operators (rhs / jvp / velocity / groom): PINNED against tests/golden/op_*.npz, which
 *     were produced by importing the reference's own sympy + generated-C layer
 *     (tests/golden/make_golden.py).
 *   - time step: the arithmetic of the reference's step lives in PETSc (TS ROSW + TSAdapt
 *     + LU/MUMPS), a third-party dependency that is NOT in /root/reference and whose
 *     version the reference does not pin.  ko_rosw_step_* restate PETSc's published
 *     RA34PW2 scheme (coefficients: SURVEY.md 8c) and are pinned only against
 *     tests/golden/step_*.npz = reference operators + exact sparse LU + the same tableau.
 *     Against PETSc itself: "stepper parity unpinned".
 *
 * Data layout: SoA planes, x fastest:  a[c][k][j][i] at  i + nx*(j + ny*(k + nz*c)).
 * (The reference's PETSc Vec is dof-fastest, KSFD/ksfdgrid.py:9-58; conversion is the caller's.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KO_MAXF 16

/* OpenMP is used only by the cpu_baseline timing leg; default is one thread (tests). */
static int ko_threads = 1;
void ko_set_threads(int n) { ko_threads = n < 1 ? 1 : n; }

typedef struct ko_config {
    int32_t dim, nlig, ngroups, cap_kind;      /* cap_kind: 0 tophat, 1 witch (ksfdsoln.py:150-157) */
    int64_t n[3];
    double L[3];
    double s2, rhomax, cushion, maxscale, rhomin, Umin;
    const int32_t *lig_group;                  /* [nlig] 0-based group index */
    const double *lig_w, *lig_s, *lig_gamma, *lig_D;
    const double *grp_alpha, *grp_beta;        /* [ngroups] */
} ko_config;

static inline int64_t ko_npts(const ko_config *c) { return c->n[0] * c->n[1] * c->n[2]; }
static inline int64_t wrapi(int64_t i, int64_t n) { i %= n; return i < 0 ? i + n : i; }

/* 4th-order central weights on {-2h,-h,0,h,2h}: what sympy's as_finite_difference yields in
 * KSFD/ksfdsym.py:391-436. */
static const double W1[5] = { 1.0 / 12, -2.0 / 3, 0.0, 2.0 / 3, -1.0 / 12 };
static const double W2[5] = { -1.0 / 12, 4.0 / 3, -5.0 / 2, 4.0 / 3, -1.0 / 12 };

/* KSFD/ksfdsym.py:888-900 groom: rho=max(rho,rhomin), NaN->rhomin; same for U with Umin. */
void ko_groom(const ko_config *c, double *u)
{
    int64_t N = ko_npts(c);
    for (int f = 0; f <= c->nlig; f++) {
        double lo = f == 0 ? c->rhomin : c->Umin;
        double *p = u + (int64_t)f * N;
        for (int64_t i = 0; i < N; i++) {
            double x = p[i];
            p[i] = (x != x) ? lo : (x < lo ? lo : x);   /* np.maximum propagates NaN, then NaN->lo */
        }
    }
}

/* Pointwise free energy G and (optionally) its partials.
 * G = sum_g -beta_g log(alpha_g + sum_l w_gl U_gl) + Vcap(rho) + s2 log(rho)
 *   KSFD/ksfdsym.py:983-990, ksfdligand.py:527-547,720-746, ksfdsoln.py:147-161. */
static void ko_G_point(const ko_config *c, double rho, const double *U, double *G, double *Grho, double *GU)
{
    double sums[KO_MAXF] = { 0 };
    for (int l = 0; l < c->nlig; l++) sums[c->lig_group[l]] += c->lig_w[l] * U[l];
    double g = 0.0;
    for (int q = 0; q < c->ngroups; q++) g += -c->grp_beta[q] * log(c->grp_alpha[q] + sums[q]);
    double x = (rho - c->rhomax) / c->cushion;
    double th = tanh(x);
    double ms = c->maxscale * c->s2;
    double cap, dcap;
    if (c->cap_kind == 1) {
        cap = ms * (th + 1.0) * rho / c->rhomax;
        dcap = ms * ((1.0 - th * th) * rho / (c->cushion * c->rhomax) + (th + 1.0) / c->rhomax);
    } else {
        cap = ms * (th + 1.0);
        dcap = ms * (1.0 - th * th) / c->cushion;
    }
    *G = g + cap + c->s2 * log(rho);
    if (Grho) *Grho = c->s2 / rho + dcap;
    if (GU)
        for (int l = 0; l < c->nlig; l++) {
            int q = c->lig_group[l];
            GU[l] = -c->grp_beta[q] * c->lig_w[l] / (c->grp_alpha[q] + sums[q]);
        }
}

void ko_G(const ko_config *c, const double *ug, double *G)
{
    int64_t N = ko_npts(c);
#pragma omp parallel for schedule(static) num_threads(ko_threads) if (ko_threads > 1)
    for (int64_t p = 0; p < N; p++) {
        double U[KO_MAXF];
        for (int l = 0; l < c->nlig; l++) U[l] = ug[(int64_t)(l + 1) * N + p];
        ko_G_point(c, ug[p], U, &G[p], NULL, NULL);
    }
}

/* first / second central difference of plane a along axis ax at point (i,j,k), periodic */
static inline void ko_d12(const ko_config *c, const double *a, int ax, int64_t i, int64_t j, int64_t k,
                          double *d1, double *d2)
{
    int64_t nx = c->n[0], ny = c->n[1], nz = c->n[2];
    double h = c->L[ax] / (double)c->n[ax];
    double s1 = 0.0, s2 = 0.0;
    for (int m = -2; m <= 2; m++) {
        int64_t ii = i, jj = j, kk = k;
        if (ax == 0) ii = wrapi(i + m, nx);
        else if (ax == 1) jj = wrapi(j + m, ny);
        else kk = wrapi(k + m, nz);
        double v = a[ii + nx * (jj + ny * kk)];
        s1 += W1[m + 2] * v;
        s2 += W2[m + 2] * v;
    }
    *d1 = s1 / h;
    *d2 = s2 / (h * h);
}

/* KSFD/ksfdsym.py:902-940 dfdt:  out_rho = grad(rho).grad(G) + rho lap(G) (+src)   (:531-571, :763-812)
 *                                 out_U   = -gamma U + s rho + D lap(U) (+src)     (:583-613)
 * u is groomed into a scratch copy first (the reference clamps the ghosted local copy, :922).
 * src: F plane pointers or NULL (NULL entries = no source). */
int ko_rhs(const ko_config *c, const double *u, const double *const *src, double *out)
{
    int64_t N = ko_npts(c), nx = c->n[0], ny = c->n[1], nz = c->n[2];
    int F = c->nlig + 1;
    double *ug = (double *)malloc(sizeof(double) * N * (F + 1));
    if (!ug) return 1;
    double *G = ug + (int64_t)F * N;
    memcpy(ug, u, sizeof(double) * N * F);
    ko_groom(c, ug);
    ko_G(c, ug, G);
#pragma omp parallel for schedule(static) collapse(2) num_threads(ko_threads) if (ko_threads > 1)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                int64_t p = i + nx * (j + ny * k);
                double acc = 0.0, lapG = 0.0;
                for (int ax = 0; ax < c->dim; ax++) {
                    double r1, r2, g1, g2;
                    ko_d12(c, ug, ax, i, j, k, &r1, &r2);
                    ko_d12(c, G, ax, i, j, k, &g1, &g2);
                    acc += r1 * g1;
                    lapG += g2;
                }
                double v = acc + ug[p] * lapG;
                if (src && src[0]) v += src[0][p];
                out[p] = v;
                for (int l = 0; l < c->nlig; l++) {
                    const double *U = ug + (int64_t)(l + 1) * N;
                    double lap = 0.0, d1, d2;
                    for (int ax = 0; ax < c->dim; ax++) { ko_d12(c, U, ax, i, j, k, &d1, &d2); lap += d2; }
                    double w = -c->lig_gamma[l] * U[p] + c->lig_s[l] * ug[p] + c->lig_D[l] * lap;
                    if (src && src[l + 1]) w += src[l + 1][p];
                    out[(int64_t)(l + 1) * N + p] = w;
                }
            }
    free(ug);
    return 0;
}

/* Analytic Jacobian action at the groomed state u (clamp treated as identity, as the
 * reference's assembled Jacobian does: KSFD/ksfdsym.py:675-761, 1067-1127; closed form SURVEY.md 0):
 *   dG = G_rho drho + sum_l G_Ul dU_l
 *   (Jv)_rho = sum_a D1(drho) D1(G) + D1(rho) D1(dG) + drho D2(G) + rho D2(dG)
 *   (Jv)_U   = -gamma dU + s drho + D lap(dU) */
int ko_jvp(const ko_config *c, const double *u, const double *v, double *out)
{
    int64_t N = ko_npts(c), nx = c->n[0], ny = c->n[1], nz = c->n[2];
    int F = c->nlig + 1;
    double *ug = (double *)malloc(sizeof(double) * N * (F + 2));
    if (!ug) return 1;
    double *G = ug + (int64_t)F * N, *dG = G + N;
    memcpy(ug, u, sizeof(double) * N * F);
    ko_groom(c, ug);
#pragma omp parallel for schedule(static) num_threads(ko_threads) if (ko_threads > 1)
    for (int64_t p = 0; p < N; p++) {
        double U[KO_MAXF], GU[KO_MAXF], Gr;
        for (int l = 0; l < c->nlig; l++) U[l] = ug[(int64_t)(l + 1) * N + p];
        ko_G_point(c, ug[p], U, &G[p], &Gr, GU);
        double d = Gr * v[p];
        for (int l = 0; l < c->nlig; l++) d += GU[l] * v[(int64_t)(l + 1) * N + p];
        dG[p] = d;
    }
#pragma omp parallel for schedule(static) collapse(2) num_threads(ko_threads) if (ko_threads > 1)
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                int64_t p = i + nx * (j + ny * k);
                double acc = 0.0, lapG = 0.0, lapdG = 0.0;
                for (int ax = 0; ax < c->dim; ax++) {
                    double r1, r2, g1, g2, v1, v2, e1, e2;
                    ko_d12(c, ug, ax, i, j, k, &r1, &r2);
                    ko_d12(c, G, ax, i, j, k, &g1, &g2);
                    ko_d12(c, v, ax, i, j, k, &v1, &v2);
                    ko_d12(c, dG, ax, i, j, k, &e1, &e2);
                    acc += v1 * g1 + r1 * e1;
                    lapG += g2;
                    lapdG += e2;
                }
                out[p] = acc + v[p] * lapG + ug[p] * lapdG;
                for (int l = 0; l < c->nlig; l++) {
                    const double *dU = v + (int64_t)(l + 1) * N;
                    double lap = 0.0, d1, d2;
                    for (int ax = 0; ax < c->dim; ax++) { ko_d12(c, dU, ax, i, j, k, &d1, &d2); lap += d2; }
                    out[(int64_t)(l + 1) * N + p] = -c->lig_gamma[l] * dU[p] + c->lig_s[l] * v[p] + c->lig_D[l] * lap;
                }
            }
    free(ug);
    return 0;
}

/* Assembled analytic Jacobian df/du at the groomed state u, CSR, in the unknown ordering of the reference's PETSc
 * Vec (row = F*p + dof, p = i + nx*(j + ny*k): DMDA dof-fastest, KSFD/ksfdgrid.py:388-411).  Restates
 * Derivatives.Jacobian (KSFD/ksfdsym.py:814-886; rho-row entry fields :675-761, 1067-1127; U-row constants
 * :630-673) followed by the scatter of ksfdMat.setValuesJacobian (cython/ksfdMat/ksfdMat.pyx:55-180).
 *   d f_rho(p) / d rho(p)        = lap G + rho_p (sum_a w2_0/h_a^2) G_rho(p)
 *   d f_rho(p) / d rho(p + m e_a) = w1_m/h_a D1_a(G) + (w1_m/h_a D1_a(rho) + rho_p w2_m/h_a^2) G_rho(p + m e_a)
 *   d f_rho(p) / d U_l(q)         = same with G_Ul(q) and without the D1_a(G) term
 *   d f_Ul(p)  / d rho(p) = s_l ;  d f_Ul(p) / d U_l(p + m e_a) = D_l w2_m/h_a^2 ;  centre: -gamma_l + D_l sum_a w2_0/h_a^2
 * Entries per point: F*npts (rho row) + nlig*(npts+1) (U rows), npts = 4*dim+1.  Order inside a rho row: for each
 * dof, centre then axis by axis m = -2,-1,+1,+2; inside a U_l row: rho centre, U_l centre, then the axes.
 * Columns are not sorted.  rowptr has F*N+1 entries. */
int64_t ko_jacobian_nnz(const ko_config *c)
{
    int npts = 4 * c->dim + 1, F = c->nlig + 1;
    return ko_npts(c) * ((int64_t)F * npts + (int64_t)c->nlig * (npts + 1));
}

int ko_jacobian_csr(const ko_config *c, const double *u, int64_t *rowptr, int64_t *col, double *val)
{
    int64_t N = ko_npts(c), nx = c->n[0], ny = c->n[1], nz = c->n[2];
    int F = c->nlig + 1, nl = c->nlig, npts = 4 * c->dim + 1;
    double *ug = (double *)malloc(sizeof(double) * N * (2 * F + 1));
    if (!ug) return 1;
    double *G = ug + (int64_t)F * N, *Gd = G + N;          /* Gd: F planes G_rho, G_U1.. */
    memcpy(ug, u, sizeof(double) * N * F);
    ko_groom(c, ug);
    for (int64_t p = 0; p < N; p++) {
        double U[KO_MAXF], GU[KO_MAXF], Gr;
        for (int l = 0; l < nl; l++) U[l] = ug[(int64_t)(l + 1) * N + p];
        ko_G_point(c, ug[p], U, &G[p], &Gr, GU);
        Gd[p] = Gr;
        for (int l = 0; l < nl; l++) Gd[(int64_t)(l + 1) * N + p] = GU[l];
    }
    const int64_t per = (int64_t)F * npts + (int64_t)nl * (npts + 1);
    static const int MS[4] = { -2, -1, 1, 2 };
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                int64_t p = i + nx * (j + ny * k);
                int64_t e = p * per;
                double d1g[3], d1r[3], lapG = 0.0, w2c = 0.0, ih[3], ih2[3];
                int64_t q[3][4];
                for (int ax = 0; ax < c->dim; ax++) {
                    double h = c->L[ax] / (double)c->n[ax], d2;
                    ih[ax] = 1.0 / h;
                    ih2[ax] = 1.0 / (h * h);
                    ko_d12(c, G, ax, i, j, k, &d1g[ax], &d2);
                    lapG += d2;
                    ko_d12(c, ug, ax, i, j, k, &d1r[ax], &d2);
                    w2c += W2[2] * ih2[ax];
                    for (int m = 0; m < 4; m++) {
                        int64_t ii = i, jj = j, kk = k;
                        if (ax == 0) ii = wrapi(i + MS[m], nx);
                        else if (ax == 1) jj = wrapi(j + MS[m], ny);
                        else kk = wrapi(k + MS[m], nz);
                        q[ax][m] = ii + nx * (jj + ny * kk);
                    }
                }
                const double rho0 = ug[p];
                rowptr[p * F] = e;
                for (int dof = 0; dof < F; dof++) {
                    const double *Gx = Gd + (int64_t)dof * N;
                    col[e] = p * F + dof;
                    val[e++] = (dof == 0 ? lapG : 0.0) + rho0 * w2c * Gx[p];
                    for (int ax = 0; ax < c->dim; ax++)
                        for (int m = 0; m < 4; m++) {
                            double w1 = W1[MS[m] + 2] * ih[ax], w2 = W2[MS[m] + 2] * ih2[ax];
                            col[e] = q[ax][m] * F + dof;
                            val[e++] = (dof == 0 ? w1 * d1g[ax] : 0.0) + (w1 * d1r[ax] + rho0 * w2) * Gx[q[ax][m]];
                        }
                }
                for (int l = 0; l < nl; l++) {
                    rowptr[p * F + l + 1] = e;
                    col[e] = p * F;
                    val[e++] = c->lig_s[l];
                    col[e] = p * F + l + 1;
                    val[e++] = -c->lig_gamma[l] + c->lig_D[l] * w2c;
                    for (int ax = 0; ax < c->dim; ax++)
                        for (int m = 0; m < 4; m++) {
                            col[e] = q[ax][m] * F + l + 1;
                            val[e++] = c->lig_D[l] * W2[MS[m] + 2] * ih2[ax];
                        }
                }
            }
    rowptr[N * F] = N * per;
    free(ug);
    return 0;
}

/* KSFD/ksfdsym.py:1158-1209 velocity: v_a = D1_a(G) on the groomed state; vel = dim planes. */
int ko_velocity(const ko_config *c, const double *u, double *vel)
{
    int64_t N = ko_npts(c), nx = c->n[0], ny = c->n[1], nz = c->n[2];
    int F = c->nlig + 1;
    double *ug = (double *)malloc(sizeof(double) * N * (F + 1));
    if (!ug) return 1;
    double *G = ug + (int64_t)F * N;
    memcpy(ug, u, sizeof(double) * N * F);
    ko_groom(c, ug);
    ko_G(c, ug, G);
    for (int ax = 0; ax < c->dim; ax++)
        for (int64_t k = 0; k < nz; k++)
            for (int64_t j = 0; j < ny; j++)
                for (int64_t i = 0; i < nx; i++) {
                    double d1, d2;
                    ko_d12(c, G, ax, i, j, k, &d1, &d2);
                    vel[(int64_t)ax * N + i + nx * (j + ny * k)] = d1;
                }
    free(ug);
    return 0;
}

/* KSFD/ksfdts.py:302-319 CFL_step: h_CFL = min_a spacing_a * sw / max|v_a|  (sw = 2). */
int ko_cfl(const ko_config *c, const double *u, double vmax[3], double *hcfl)
{
    int64_t N = ko_npts(c);
    double *vel = (double *)malloc(sizeof(double) * N * c->dim);
    if (!vel) return 1;
    ko_velocity(c, u, vel);
    double hm = INFINITY;
    for (int ax = 0; ax < 3; ax++) vmax[ax] = 0.0;
    for (int ax = 0; ax < c->dim; ax++) {
        double m = 0.0;
        for (int64_t p = 0; p < N; p++) { double a = fabs(vel[(int64_t)ax * N + p]); if (a > m) m = a; }
        vmax[ax] = m;
        double sp = c->L[ax] / (double)c->n[ax];
        double hh = m == 0.0 ? INFINITY : sp * 2.0 / m;
        if (hh < hm) hm = hh;
    }
    *hcfl = hm;
    free(vel);
    return 0;
}

/* ---------------- PETSc TS ROSW "ra34pw2" (default TSROSW type), restated ----------------
 * Coefficients A, Gamma, b, b2: SURVEY.md 8c (PETSc src/ts/impls/rosw/rosw.c, not in /root/reference).
 * PETSc works in transformed stage variables Y = Gamma k:  At = A Gamma^-1, bt = b Gamma^-1,
 * Zdot_i = (1/h) sum_{j<i} (Gamma^-1)_ij Y_j; with -snes_type ksponly (one linear solve per stage):
 *      (1/(gamma h) I - J(t_n,u_n)) Y_i = f(t_n + ASum_i h, u_n + sum_{j<i} At_ij Y_j) - Zdot_i
 *      u_{n+1} = u_n + sum bt_j Y_j,   err = sum (b2t_j - bt_j) Y_j
 * IFunction = udot - f(u) (KSFD/ksfdts.py:563-596), IJacobian = shift*I - J (KSFD/ksfdts.py:598-640). */
#define KO_GAM 4.3586652150845900e-01
static const double RA_A[4][4] = { { 0, 0, 0, 0 },
                                   { 8.7173304301691801e-01, 0, 0, 0 },
                                   { 8.4457060015369423e-01, -1.1299064236484185e-01, 0, 0 },
                                   { 0, 0, 1., 0 } };
static const double RA_G[4][4] = { { KO_GAM, 0, 0, 0 },
                                   { -8.7173304301691801e-01, KO_GAM, 0, 0 },
                                   { -9.0338057013044082e-01, 5.4180672388095326e-02, KO_GAM, 0 },
                                   { 2.4212380706095346e-01, -1.2232505839045147e+00, 5.4526025533510214e-01, KO_GAM } };
static const double RA_b[4] = { 2.4212380706095346e-01, -1.2232505839045147e+00, 1.5452602553351020e+00, KO_GAM };
static const double RA_b2[4] = { 3.7810903145819369e-01, -9.6042292212423178e-02, 0.5, 2.1793326075422950e-01 };

typedef struct { double At[4][4], Ginv[4][4], bt[4], b2t[4], asum[4]; } ko_tableau;

void ko_tableau_build(ko_tableau *T)
{
    /* forward substitution inverse of lower-triangular Gamma */
    memset(T, 0, sizeof(*T));
    for (int col = 0; col < 4; col++)
        for (int i = col; i < 4; i++) {
            double s = (i == col) ? 1.0 : 0.0;
            for (int k = col; k < i; k++) s -= RA_G[i][k] * T->Ginv[k][col];
            T->Ginv[i][col] = s / RA_G[i][i];
        }
    for (int i = 0; i < 4; i++) {
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += RA_A[i][k] * T->Ginv[k][j];
            T->At[i][j] = s;
            T->asum[i] += RA_A[i][j];
        }
    }
    for (int j = 0; j < 4; j++) {
        double s = 0.0, s2 = 0.0;
        for (int k = 0; k < 4; k++) { s += RA_b[k] * T->Ginv[k][j]; s2 += RA_b2[k] * T->Ginv[k][j]; }
        T->bt[j] = s;
        T->b2t[j] = s2;
    }
}

/* exported so tests can read the derived coefficients */
void ko_tableau_get(double *At16, double *Ginv16, double *bt4, double *b2t4, double *asum4)
{
    ko_tableau T;
    ko_tableau_build(&T);
    memcpy(At16, T.At, sizeof(T.At));
    memcpy(Ginv16, T.Ginv, sizeof(T.Ginv));
    memcpy(bt4, T.bt, sizeof(T.bt));
    memcpy(b2t4, T.b2t, sizeof(T.b2t));
    memcpy(asum4, T.asum, sizeof(T.asum));
}

/* TSErrorWeightedNorm (2-norm form) as TSAdaptBasic uses it:
 * sqrt(mean_i (err_i / (atol + rtol max(|u_i|,|y_i|)))^2),  y = u + err the embedded solution. */
double ko_wrms(int64_t n, const double *unew, const double *err, double atol, double rtol)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double a = fabs(unew[i]), b = fabs(unew[i] + err[i]);
        double tol = atol + rtol * (a > b ? a : b);
        double e = err[i] / tol;
        s += e * e;
    }
    return sqrt(s / (double)n);
}

/* TSAdaptChoose_Basic (PETSc src/ts/adapt/impls/basic, restated): accept iff enorm<=1 (or h already at dt_min);
 * hfac = clip(safety*enorm^(-1/order)), order=3; after two consecutive rejections safety *= reject_safety. */
double ko_adapt_basic(double h, double enorm, int *accept, double safety, double clip_lo, double clip_hi,
                      double dt_min, double dt_max, int prev_accept, double reject_safety)
{
    if (enorm > 1.0) {
        if (!prev_accept) safety *= reject_safety;
        *accept = h < (1.0 + 1.4901161193847656e-08) * dt_min;
    } else *accept = 1;
    double hfac = enorm > 0.0 ? safety * pow(enorm, -1.0 / 3.0) : INFINITY;
    if (hfac < clip_lo) hfac = clip_lo;
    if (hfac > clip_hi) hfac = clip_hi;
    double hn = h * hfac;
    if (hn < dt_min) hn = dt_min;
    if (hn > dt_max) hn = dt_max;
    return hn;
}

/* ---- linear solvers for (shift I - J) y = b ---- */
typedef struct { const ko_config *c; const double *u; double shift; double *tmp; } ko_op;

static int ko_apply(const ko_op *op, const double *x, double *y)
{
    int64_t n = ko_npts(op->c) * (op->c->nlig + 1);
    if (ko_jvp(op->c, op->u, x, y)) return 1;
    for (int64_t i = 0; i < n; i++) y[i] = op->shift * x[i] - y[i];
    return 0;
}

/* dense LU with partial pivoting (the reference's -pc_type lu, small grids only) */
static int ko_dense_factor(const ko_op *op, int64_t n, double **Aout, int64_t **pivout)
{
    double *A = (double *)malloc(sizeof(double) * n * n);   /* column major: A[r + n*c] */
    int64_t *piv = (int64_t *)malloc(sizeof(int64_t) * n);
    double *e = (double *)calloc(n, sizeof(double));
    if (!A || !piv || !e) return 1;
    for (int64_t col = 0; col < n; col++) {
        e[col] = 1.0;
        ko_apply(op, e, A + n * col);
        e[col] = 0.0;
    }
    free(e);
    for (int64_t k = 0; k < n; k++) {
        int64_t p = k;
        double m = fabs(A[k + n * k]);
        for (int64_t r = k + 1; r < n; r++) if (fabs(A[r + n * k]) > m) { m = fabs(A[r + n * k]); p = r; }
        piv[k] = p;
        if (m == 0.0) { free(A); free(piv); return 2; }
        if (p != k) for (int64_t cc = 0; cc < n; cc++) { double t = A[k + n * cc]; A[k + n * cc] = A[p + n * cc]; A[p + n * cc] = t; }
        double inv = 1.0 / A[k + n * k];
        for (int64_t r = k + 1; r < n; r++) A[r + n * k] *= inv;
        for (int64_t cc = k + 1; cc < n; cc++) {
            double a = A[k + n * cc];
            if (a != 0.0) for (int64_t r = k + 1; r < n; r++) A[r + n * cc] -= A[r + n * k] * a;
        }
    }
    *Aout = A;
    *pivout = piv;
    return 0;
}

static void ko_dense_solve(int64_t n, const double *A, const int64_t *piv, double *x)
{
    for (int64_t k = 0; k < n; k++) { int64_t p = piv[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
    for (int64_t k = 0; k < n; k++) { double a = x[k]; if (a != 0.0) for (int64_t r = k + 1; r < n; r++) x[r] -= A[r + n * k] * a; }
    for (int64_t k = n - 1; k >= 0; k--) { x[k] /= A[k + n * k]; double a = x[k]; for (int64_t r = 0; r < k; r++) x[r] -= A[r + n * k] * a; }
}

/* restarted GMRES(m), classical Gram-Schmidt with one refinement pass, x0 = 0, unpreconditioned;
 * stops when ||r|| <= max(rtol*||b||, atol).  Returns iterations via *its. */
static int ko_gmres(const ko_op *op, int64_t n, const double *b, double *x, double rtol, double atol,
                    int m, int maxit, int *its, double *resid)
{
    double *V = (double *)malloc(sizeof(double) * n * (m + 1));
    double *w = (double *)malloc(sizeof(double) * n);
    double *H = (double *)calloc((size_t)(m + 1) * m, sizeof(double));
    double *cs = (double *)calloc(m, sizeof(double)), *sn = (double *)calloc(m, sizeof(double));
    double *g = (double *)calloc(m + 1, sizeof(double)), *y = (double *)calloc(m, sizeof(double));
    if (!V || !w || !H || !cs || !sn || !g || !y) return 1;
    memset(x, 0, sizeof(double) * n);
    double bn = 0.0;
    for (int64_t i = 0; i < n; i++) bn += b[i] * b[i];
    bn = sqrt(bn);
    double tol = rtol * bn > atol ? rtol * bn : atol;
    int total = 0;
    double rn = bn;
    while (total < maxit && rn > tol) {
        /* r = b - A x */
        if (total == 0) memcpy(V, b, sizeof(double) * n);
        else {
            ko_apply(op, x, w);
            for (int64_t i = 0; i < n; i++) V[i] = b[i] - w[i];
        }
        double beta = 0.0;
        for (int64_t i = 0; i < n; i++) beta += V[i] * V[i];
        beta = sqrt(beta);
        rn = beta;
        if (rn <= tol) break;
        for (int64_t i = 0; i < n; i++) V[i] /= beta;
        memset(g, 0, sizeof(double) * (m + 1));
        g[0] = beta;
        int j = 0;
        for (; j < m && total < maxit; j++) {
            double *vj = V + n * j, *vn = V + n * (j + 1);
            ko_apply(op, vj, vn);
            double *h = H + (size_t)(m + 1) * j;
            for (int i = 0; i <= j; i++) h[i] = 0.0;
            for (int pass = 0; pass < 2; pass++) {
                double d[64];
                for (int i = 0; i <= j; i++) { const double *vi = V + n * i; double s = 0.0;
_Pragma("omp parallel for reduction(+:s) schedule(static) num_threads(ko_threads) if (ko_threads > 1)")
                    for (int64_t q = 0; q < n; q++) s += vi[q] * vn[q]; d[i] = s; }
                for (int i = 0; i <= j; i++) { const double *vi = V + n * i; double s = d[i];
_Pragma("omp parallel for schedule(static) num_threads(ko_threads) if (ko_threads > 1)")
                    for (int64_t q = 0; q < n; q++) vn[q] -= s * vi[q]; h[i] += s; }
            }
            double hn = 0.0;
            for (int64_t q = 0; q < n; q++) hn += vn[q] * vn[q];
            hn = sqrt(hn);
            h[j + 1] = hn;
            if (hn > 0.0) for (int64_t q = 0; q < n; q++) vn[q] /= hn;
            for (int i = 0; i < j; i++) { double t = cs[i] * h[i] + sn[i] * h[i + 1]; h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]; h[i] = t; }
            double den = hypot(h[j], h[j + 1]);
            cs[j] = h[j] / den; sn[j] = h[j + 1] / den;
            h[j] = den; h[j + 1] = 0.0;
            g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
            total++;
            rn = fabs(g[j + 1]);
            if (rn <= tol) { j++; break; }
        }
        for (int i = j - 1; i >= 0; i--) {
            double s = g[i];
            for (int k = i + 1; k < j; k++) s -= H[(size_t)(m + 1) * k + i] * y[k];
            y[i] = s / H[(size_t)(m + 1) * i + i];
        }
        for (int i = 0; i < j; i++) { const double *vi = V + n * i; double a = y[i]; for (int64_t q = 0; q < n; q++) x[q] += a * vi[q]; }
    }
    *its = total;
    if (resid) *resid = rn;
    free(V); free(w); free(H); free(cs); free(sn); free(g); free(y);
    return rn <= tol ? 0 : 3;
}

/* One RA34PW2 step from (t,u) with step h.
 *   solver: 0 = dense LU (reference's direct solve; small grids), 1 = matrix-free GMRES.
 *   src_stage: NULL, or 4*F plane pointers [stage][field] (NULL entries allowed) holding the source
 *              fields at the stage times t + asum[i]*h  (KSFD/ksfdsym.py:930-936 adds sources(t_stage)).
 *   Outputs: unew (F*N), err (F*N, embedded-minus-main), *wrms.  Returns 0 on success. */
int ko_rosw_step(const ko_config *c, const double *u, double h, const double *const *src_stage,
                 double atol, double rtol, int solver, double ksp_rtol, double ksp_atol, int restart, int maxit,
                 double *unew, double *err, double *wrms, int *lin_its)
{
    int F = c->nlig + 1;
    int64_t n = ko_npts(c) * F;
    ko_tableau T;
    ko_tableau_build(&T);
    double *ug = (double *)malloc(sizeof(double) * n);
    double *Y = (double *)malloc(sizeof(double) * n * 4);
    double *Z = (double *)malloc(sizeof(double) * n), *b = (double *)malloc(sizeof(double) * n);
    if (!ug || !Y || !Z || !b) return 1;
    /* KSFDTS.solve grooms the global vector before every TS step (KSFD/ksfdts.py:210, 231-237) */
    memcpy(ug, u, sizeof(double) * n);
    ko_groom(c, ug);
    ko_op op = { c, ug, 1.0 / (KO_GAM * h), NULL };
    double *A = NULL;
    int64_t *piv = NULL;
    int rc = 0, its_total = 0;
    if (solver == 0 && (rc = ko_dense_factor(&op, n, &A, &piv))) goto done;
    for (int i = 0; i < 4; i++) {
        memcpy(Z, ug, sizeof(double) * n);
        for (int j = 0; j < i; j++) { double a = T.At[i][j]; const double *y = Y + n * j; if (a != 0.0) for (int64_t q = 0; q < n; q++) Z[q] += a * y[q]; }
        if ((rc = ko_rhs(c, Z, src_stage ? src_stage + (size_t)i * F : NULL, b))) goto done;
        for (int j = 0; j < i; j++) { double a = T.Ginv[i][j] / h; const double *y = Y + n * j; if (a != 0.0) for (int64_t q = 0; q < n; q++) b[q] -= a * y[q]; }
        double *yi = Y + n * i;
        if (solver == 0) { memcpy(yi, b, sizeof(double) * n); ko_dense_solve(n, A, piv, yi); }
        else {
            int its = 0;
            rc = ko_gmres(&op, n, b, yi, ksp_rtol, ksp_atol, restart, maxit, &its, NULL);
            its_total += its;
            if (rc) goto done;
        }
    }
    memcpy(unew, ug, sizeof(double) * n);
    memset(err, 0, sizeof(double) * n);
    for (int j = 0; j < 4; j++) {
        const double *y = Y + n * j;
        double bj = T.bt[j], ej = T.b2t[j] - T.bt[j];
        for (int64_t q = 0; q < n; q++) { unew[q] += bj * y[q]; err[q] += ej * y[q]; }
    }
    *wrms = ko_wrms(n, unew, err, atol, rtol);
done:
    if (lin_its) *lin_its = its_total;
    free(ug); free(Y); free(Z); free(b); free(A); free(piv);
    return rc;
}

/* Synthetic initial condition of SURVEY.md 8d / BASELINE.md 4: coarse normal samples z (n/4 per axis,
 * supplied by the caller from numpy's default_rng so the stream matches KSFD/ksfdrandom.py:44-49)
 * interpolated to the fine grid with the separable weight f(x) = 2x^3 - 3x^2 + 1 on |dx|/h_coarse < 1,
 * periodic (KSFD/ksfdrandom.py:116, 194-214).  out[p] = sum_v z[v] * prod_a f(|x_p - x_v|_a / hc_a). */
void ko_random_function(const ko_config *c, const int64_t nc[3], const double *z, double *out)
{
    int64_t nx = c->n[0], ny = c->n[1], nz = c->n[2];
    for (int64_t k = 0; k < nz; k++)
        for (int64_t j = 0; j < ny; j++)
            for (int64_t i = 0; i < nx; i++) {
                int64_t idx[3] = { i, j, k };
                int64_t lo[3] = { 0, 0, 0 };
                double wlo[3] = { 1, 1, 1 }, whi[3] = { 0, 0, 0 };
                for (int a = 0; a < c->dim; a++) {
                    double xc = (double)idx[a] * (double)nc[a] / (double)c->n[a];   /* position in coarse units */
                    double fl = floor(xc);
                    double fr = xc - fl;
                    lo[a] = (int64_t)fl;
                    wlo[a] = 2 * fr * fr * fr - 3 * fr * fr + 1;
                    double g = 1.0 - fr;
                    whi[a] = fr == 0.0 ? 0.0 : 2 * g * g * g - 3 * g * g + 1;
                }
                double s = 0.0;
                for (int dz = 0; dz < (c->dim > 2 ? 2 : 1); dz++)
                    for (int dy = 0; dy < (c->dim > 1 ? 2 : 1); dy++)
                        for (int dx = 0; dx < 2; dx++) {
                            double w = (dx ? whi[0] : wlo[0]) * (c->dim > 1 ? (dy ? whi[1] : wlo[1]) : 1.0) *
                                       (c->dim > 2 ? (dz ? whi[2] : wlo[2]) : 1.0);
                            if (w == 0.0) continue;
                            int64_t ci = wrapi(lo[0] + dx, nc[0]);
                            int64_t cj = c->dim > 1 ? wrapi(lo[1] + dy, nc[1]) : 0;
                            int64_t ck = c->dim > 2 ? wrapi(lo[2] + dz, nc[2]) : 0;
                            s += w * z[ci + nc[0] * (cj + nc[1] * ck)];
                        }
                out[i + nx * (j + ny * k)] = s;
            }
}
