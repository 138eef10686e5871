// libksfd_hip.so -- host side: handle, C ABI (include/ksfd_hip.h), matrix-free GMRES and the
// Rosenbrock-W step that stand in for petsc4py TS.step() in the reference (KSFD/ksfdts.py:211).
// gfx950 only.  No CPU fallback: every entry point runs HIP kernels or fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/ksfd_hip.h"
#include "stencil.hip.h"
#include "mg.hip.h"
#include "transport.h"

// ------------------------------------------------------------------------------------------------
// kernel classes for the profile
enum { KC_RHS = 0, KC_JVP, KC_MULTIDOT, KC_GSUPDATE, KC_LINCOMB, KC_BASISAXPY, KC_FINISH, KC_REDUCE,
       KC_GFIELD, KC_VELOCITY, KC_MISC, KC_HALO, KC_MG };
static const char *kc_names[KSFD_NKCLASS] = { "rhs", "jvp", "multidot", "gs_update", "lincomb", "basis_axpy",
                                              "rosw_finish", "reduce", "gfield", "velocity", "misc", "halo", "mg" };
extern "C" const char *ksfd_kernel_class_name(int32_t c) { return (c >= 0 && c < KSFD_NKCLASS) ? kc_names[c] : "?"; }

static thread_local std::string g_create_error;

struct EvPair { hipEvent_t a, b; int cls; };

// one grid of the multigrid hierarchy (level 0 = the solver's own grid; see mg.hip.h)
struct MGLevel {
    KGeom G;
    KPhys P;
    KVec kv;
    int64_t vlen = 0;
    int nblk = 1;
    double *coef = nullptr;    // [rho, G, G_rho, G_U..] planes (level 0 aliases the handle's)
    double *dinv = nullptr;    // F*F planes
    double *x = nullptr, *b = nullptr, *r = nullptr, *d = nullptr, *Ad = nullptr, *dG = nullptr;
    double lam_max = 2.3;
    double ratio = 60.0;       // lambda_max/lambda_min estimate (used on the coarsest grid)
};

struct ksfd_handle {
    ksfd_config cfg;
    int32_t lig_group[KSFD_MAXL];
    double lig_w[KSFD_MAXL], lig_s[KSFD_MAXL], lig_gamma[KSFD_MAXL], lig_D[KSFD_MAXL];
    double grp_alpha[KSFD_MAXL], grp_beta[KSFD_MAXL];
    KGeom G;
    KPhys P;
    KVec kv;
    int rank = 0, size = 1, device = 0;
    int64_t slow0 = 0;               // first owned global slow index
    hipStream_t st = nullptr;
    hipStream_t st_comm = nullptr;            // halo exchange stream (overlapped with interior rows)
    hipEvent_t ev_ready = nullptr, ev_halo = nullptr;
    bool overlap = true;
    Transport *tr = nullptr;
    std::string err;

    // device vectors (each F*plane doubles)
    int64_t vlen = 0;
    double *u = nullptr, *usave = nullptr, *Z = nullptr, *bvec = nullptr, *Y = nullptr, *V = nullptr;
    double *t1 = nullptr, *t2 = nullptr, *t3 = nullptr, *errv = nullptr;
    double *Gb = nullptr, *dGb = nullptr;   // generic-path scratch planes
    double *coef = nullptr;                 // frozen-Jacobian coefficient planes [rho, G, G_rho, G_U..]
    bool use_frozen = true;
    double *flat = nullptr;                 // staging for host layouts: max(F,dim)*nloc
    double *src[4][KSFD_MAXL + 1];          // dense source planes per stage (lazy)
    double *part = nullptr;                 // block partials
    double *dres = nullptr;                 // reduced results (device)
    double *hres = nullptr;                 // pinned host mirror
    // zero-copy hand-over of reduction results: the kernel stores into hres through its device alias and raises pub_flag
    double *hres_dev = nullptr;
    unsigned long long *pub_flag = nullptr, *pub_flag_dev = nullptr, pub_seq = 0;
    unsigned int *pub_count = nullptr;
    bool zero_copy = true;
    int restart_alloc = 0;
    int nblk_vec = 0;                       // grid.x of the BLAS-1 kernels
    bool have_err = false;

    // tuning
    int use_fused = 1;
    int yseg = 35;        // rows per wave segment, RHS kernel (measured best at 4096^2)
    int yseg_jvp = 16;    // same for the Jacobian-action kernels
    int zseg = 32;        // planes per wave segment, 3-D z-marching kernel

    // profile
    bool profiling = false;
    int prof_only = -1;             // >= 0: HIP events only around launches of this kernel class
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> pool;
    ksfd_profile prof;
    double bytes_acc = 0.0;

    // pipelined GMRES (device-resident Hessenberg / Givens state, see gmres_async)
    double *gm_dev = nullptr;       // [G (m+1)^2 | H (m+1)m | cs m | sn m | g m+1 | coef MAXDOT | scale 1 | mon 2(m+1)]
    double *gm_host = nullptr;      // pinned: [mon 2(m+1) | H (m+1)m | g m+1]
    std::vector<hipEvent_t> gm_ev;
    int async_mode = 0;             // 0 off (default: measured no gain on one GPU, tools/async_bench.py), 1 whenever legal,
                                    // 2 when the local problem is small

    // polynomial (Chebyshev) preconditioner + flexible GMRES (see poly_setup / gmres)
    double *Zb = nullptr;           // preconditioned basis z_j = p(A) v_j (allocated on first use)
    double *pvec = nullptr;         // power-iteration vector for lambda_max(A)
    double lamJ = -1.0;             // running estimate of lambda_max(-J) = lambda_max(A) - shift
    int lam_age = 0, lam_period = 1;   // steps since the last estimate / re-estimate every lam_period steps (1..8, grows while stable)
    int poly_deg = 0;
    double poly_alpha[8];           // z = sum_i alpha_i (A/shift)^i v
    double poly_shift = -1.0;
    int poly_max_deg = 6;
    double mg_threshold = 60.0;      // stiffness above which pc_type 2 switches from the polynomial to multigrid
    double poly_target = 0.02;      // wanted reduction per outer iteration (picks the degree)
    float *coef32 = nullptr;        // fp32 copy of the frozen coefficient planes (2-D strip path only)
    bool fuse_stage = true;         // stage-vector algebra inside the RHS kernel (2-D strip path)
    bool poly_fp32 = true;          // Horner temporaries and coefficients of p(A) in fp32 storage (the outer A z_j stays fp64)

    // asynchronous snapshots for writers (ksfd_snapshot_begin / _wait): layout transform on the compute stream into a
    // device staging slot, D2H on a third stream into pinned memory while the stepper carries on
    hipStream_t st_io = nullptr;
    double *snap_dev[2] = { nullptr, nullptr }, *snap_host[2] = { nullptr, nullptr };
    hipEvent_t snap_ready[2] = { nullptr, nullptr }, snap_done[2] = { nullptr, nullptr };
    bool snap_busy[2] = { false, false };
    int snap_next = 0;

    // Krylov recycling across the four stage systems of one step (same matrix): see gmres()
    struct RecSpace { bool valid = false; int vb = 0, zb = 0, k = 0, pc = 0; double H[20]; };   // leading (k+1) x k raw Hessenberg, ld = k+1
    RecSpace rec[4];
    int rec_vtop = 0, rec_ztop = 0;  // first free slot of V / Zb behind the kept vectors
    int rec_mode = 1;                // 0 off, 1 selected earlier stages (default), 2 every earlier stage
    int rec_keep = 3;                // leading vectors kept per stage (<= 4)

    // multigrid preconditioner
    std::vector<MGLevel> mg;
    bool mg_ok = false;          // hierarchy exists (2-D, single rank, >= 2 levels)
    double mg_shift = -1.0;      // shift the block diagonals / eigen-bounds were built for
    bool mg_coef_valid = false;  // coarse coefficient planes match the current frozen state
    hipGraphExec_t mg_graph = nullptr;   // captured coarse part of the V cycle (levels >= 1) for mg_graph_shift/x
    double mg_graph_shift = -1.0, mg_graph_bytes = 0.0;
    double *mg_graph_x = nullptr;
    bool capturing = false, mg_use_graph = true;
    int mg_nu = 2, mg_ncoarse = 400, mg_power_its = 8;   // smoothing sweeps, cap on coarsest-grid sweeps, power iterations
    double mg_ratio = 6.0, mg_coarse_tol = 1e-2;

    // ROSW tableau (PETSc transformed form)
    double At[4][4], Ginv[4][4], bt[4], b2t[4], asum[4];
};

#define GAMMA_RA 4.3586652150845900e-01

static int fail(ksfd_handle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}
#define HIPCHK(h, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return fail(h, KSFD_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- profiling helpers -------------------------------------------------------------------------
static hipEvent_t ev_get(ksfd_handle *h)
{
    if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
struct Scope {
    ksfd_handle *h; EvPair p; bool on;
    Scope(ksfd_handle *h_, int cls, double bytes) : h(h_), on(h_->profiling && !h_->capturing && (h_->prof_only < 0 || h_->prof_only == cls))
    {
        h->bytes_acc += bytes;
        h->prof.bytes[cls] += bytes;
        h->prof.launches[cls] += 1;
        if (on) { p.a = ev_get(h); p.b = ev_get(h); p.cls = cls; hipEventRecord(p.a, h->st); }
    }
    ~Scope() { if (on) { hipEventRecord(p.b, h->st); h->pending.push_back(p); } }
};
static void prof_resolve(ksfd_handle *h)
{
    if (h->pending.empty()) return;
    hipStreamSynchronize(h->st);
    for (auto &p : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) h->prof.ms[p.cls] += ms;
        h->pool.push_back(p.a);
        h->pool.push_back(p.b);
    }
    h->pending.clear();
}

// ---- small utilities ---------------------------------------------------------------------------
static inline dim3 vgrid(const ksfd_handle *h) { return dim3(h->nblk_vec, h->G.F); }
static inline double vbytes(const ksfd_handle *h, double nvec) { return nvec * 8.0 * (double)h->G.F * (double)h->G.nloc; }

static void build_tableau(ksfd_handle *h)
{
    static const double A[4][4] = { { 0, 0, 0, 0 }, { 8.7173304301691801e-01, 0, 0, 0 },
                                    { 8.4457060015369423e-01, -1.1299064236484185e-01, 0, 0 }, { 0, 0, 1., 0 } };
    static const double Gm[4][4] = { { GAMMA_RA, 0, 0, 0 }, { -8.7173304301691801e-01, GAMMA_RA, 0, 0 },
                                     { -9.0338057013044082e-01, 5.4180672388095326e-02, GAMMA_RA, 0 },
                                     { 2.4212380706095346e-01, -1.2232505839045147e+00, 5.4526025533510214e-01, GAMMA_RA } };
    static const double b[4] = { 2.4212380706095346e-01, -1.2232505839045147e+00, 1.5452602553351020e+00, GAMMA_RA };
    static const double b2[4] = { 3.7810903145819369e-01, -9.6042292212423178e-02, 0.5, 2.1793326075422950e-01 };
    memset(h->Ginv, 0, sizeof h->Ginv);
    for (int col = 0; col < 4; col++)
        for (int i = col; i < 4; i++) {
            double s = (i == col) ? 1.0 : 0.0;
            for (int k = col; k < i; k++) s -= Gm[i][k] * h->Ginv[k][col];
            h->Ginv[i][col] = s / Gm[i][i];
        }
    for (int i = 0; i < 4; i++) {
        h->asum[i] = 0.0;
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += A[i][k] * h->Ginv[k][j];
            h->At[i][j] = s;
            h->asum[i] += A[i][j];
        }
    }
    for (int j = 0; j < 4; j++) {
        double s = 0.0, s2 = 0.0;
        for (int k = 0; k < 4; k++) { s += b[k] * h->Ginv[k][j]; s2 += b2[k] * h->Ginv[k][j]; }
        h->bt[j] = s;
        h->b2t[j] = s2;
    }
}

static int fill_phys(ksfd_handle *h, const ksfd_config *c)
{
    if (c->nlig < 1 || c->nlig > KSFD_MAXL || c->ngroups < 1 || c->ngroups > KSFD_MAXL)
        return fail(h, KSFD_EINVAL, "nlig=%d ngroups=%d outside 1..%d", c->nlig, c->ngroups, KSFD_MAXL);
    KPhys &P = h->P;
    memset(&P, 0, sizeof P);
    P.nlig = c->nlig; P.ngroups = c->ngroups; P.cap_kind = c->cap_kind;
    for (int a = 0; a < 3; a++) {
        double sp = c->L[a] / (double)c->n[a];
        P.inv_h[a] = 1.0 / sp;
        P.inv_h2[a] = 1.0 / (sp * sp);
    }
    P.s2 = c->s2; P.rhomax = c->rhomax; P.inv_cushion = 1.0 / c->cushion; P.ms = c->maxscale * c->s2;
    P.rhomin = c->rhomin; P.Umin = c->Umin; P.inv_rhomax = 1.0 / c->rhomax;
    for (int l = 0; l < c->nlig; l++) {
        if (c->lig_group[l] < 0 || c->lig_group[l] >= c->ngroups) return fail(h, KSFD_EINVAL, "lig_group[%d] out of range", l);
        P.lig_group[l] = h->lig_group[l] = c->lig_group[l];
        P.lig_w[l] = h->lig_w[l] = c->lig_w[l];
        P.lig_s[l] = h->lig_s[l] = c->lig_s[l];
        P.lig_gamma[l] = h->lig_gamma[l] = c->lig_gamma[l];
        P.lig_D[l] = h->lig_D[l] = c->lig_D[l];
    }
    for (int l = c->nlig; l < KSFD_MAXL; l++) P.lig_group[l] = -1;
    for (int q = 0; q < c->ngroups; q++) {
        P.grp_alpha[q] = h->grp_alpha[q] = c->grp_alpha[q];
        P.grp_beta[q] = h->grp_beta[q] = c->grp_beta[q];
    }
    return KSFD_OK;
}

static int alloc_d(ksfd_handle *h, double **p, int64_t n)
{
    if (hipMalloc((void **)p, sizeof(double) * (size_t)n) != hipSuccess) return fail(h, KSFD_ENOMEM, "hipMalloc of %lld doubles failed", (long long)n);
    return KSFD_OK;
}

static bool fused_ok(const ksfd_handle *h)
{
    return h->use_fused && h->G.dim == 2 && (h->G.nx % 2 == 0) && h->G.nx >= 4 && h->G.sloc >= 4 && h->P.nlig <= 4;
}

// ---- halo exchange (DMDA globalToLocal stand-in, KSFD/ksfdsym.py:919-920) -----------------------
static int halo(ksfd_handle *h, double *vec)
{
    if (h->size == 1) return KSFD_OK;
    Scope sc(h, KC_HALO, 4.0 * 2.0 * 8.0 * h->G.F * (double)h->G.inner * 2.0);
    int rc = h->tr->exchange(vec, h->G.F, h->G.plane, h->G.inner, h->G.sloc, h->G.ng, h->st);
    if (rc) return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
    return KSFD_OK;
}

// ---- reductions to the host ------------------------------------------------------------------
// part holds `rows` rows of `nblk` partials; result lands in h->hres[0..rows)
// host side of the zero-copy hand-over: spin until the kernel has raised the flag (with a look at the stream now and then,
// so that a faulted launch turns into an error instead of a hang)
static int spin_for(ksfd_handle *h, unsigned long long seq)
{
    for (unsigned long long spins = 1;; spins++) {
        if (__atomic_load_n(h->pub_flag, __ATOMIC_ACQUIRE) == seq) return KSFD_OK;
        if ((spins & 0xffff) == 0) {
            hipError_t e = hipStreamQuery(h->st);
            if (e == hipSuccess) {
                if (__atomic_load_n(h->pub_flag, __ATOMIC_ACQUIRE) == seq) return KSFD_OK;
                return fail(h, KSFD_EHIP, "reduction finished without publishing its result");
            }
            if (e != hipErrorNotReady) return fail(h, KSFD_EHIP, "stream error while waiting for a reduction: %s", hipGetErrorString(e));
        }
    }
}

static int reduce_rows(ksfd_handle *h, int rows, int nblk, int op)
{
    const bool zc = h->zero_copy && !h->capturing && rows <= 128;
    const bool zc_here = zc && h->size == 1;
    const unsigned long long seq = zc ? ++h->pub_seq : 0;
    {
        Scope sc(h, KC_REDUCE, 8.0 * rows * (double)nblk);
        if (zc_here) hipLaunchKernelGGL(k_reduce_rows, dim3(rows), dim3(KSFD_BLOCK), 0, h->st, h->part, nblk, op, h->dres, h->hres_dev, h->pub_count, h->pub_flag_dev, seq);
        else hipLaunchKernelGGL(k_reduce_rows, dim3(rows), dim3(KSFD_BLOCK), 0, h->st, h->part, nblk, op, h->dres);
    }
    if (zc_here) { HIPCHK(h, hipGetLastError()); return spin_for(h, seq); }
    if (h->size > 1) {
        int rc = h->tr->allreduce(h->dres, rows, op, h->st);
        if (rc) return fail(h, KSFD_ECOMM, "allreduce failed: %s", h->tr->error().c_str());
        if (h->tr->result_on_host()) { memcpy(h->hres, h->tr->host_result(), sizeof(double) * rows); return KSFD_OK; }
        if (zc) {
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(128), 0, h->st, (const double *)h->dres, rows, h->hres_dev, h->pub_flag_dev, seq);
            HIPCHK(h, hipGetLastError());
            return spin_for(h, seq);
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->hres, h->dres, sizeof(double) * rows, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

// ---- kernel wrappers ----------------------------------------------------------------------------
#define NL_DISPATCH(nl, CALL)                                                                      \
    switch (nl) {                                                                                  \
    case 1: { constexpr int NL = 1; CALL; } break;                                                 \
    case 2: { constexpr int NL = 2; CALL; } break;                                                 \
    case 3: { constexpr int NL = 3; CALL; } break;                                                 \
    case 4: { constexpr int NL = 4; CALL; } break;                                                 \
    case 5: { constexpr int NL = 5; CALL; } break;                                                 \
    default: { constexpr int NL = 6; CALL; } break;                                                \
    }

static KStrips make_strips(const ksfd_handle *h, bool jvp = false)
{
    KStrips S;
    S.nstrips = (int)((h->G.nx + KSFD_STRIP_OUT - 1) / KSFD_STRIP_OUT);
    S.yseg = jvp ? h->yseg_jvp : h->yseg;
    // small grids: shorter segments so that there are enough waves to fill 256 CUs (a wave costs ~1 us per row it
    // marches; the 4 halo rows per segment are L2 hits at these sizes)
    {
        const long long target = jvp ? 4096 : 6144;
        long long fit = (long long)S.nstrips * h->G.sloc / target;
        if (fit < 2) fit = 2;
        if (fit < S.yseg) S.yseg = (int)fit;
    }
    S.nseg = (int)((h->G.sloc + S.yseg - 1) / S.yseg);
    S.seg0 = 0;
    S.seg_stride = 1;
    long long waves = (long long)S.nstrips * S.nseg;
    long long nb = (waves + 3) / 4;
    nb = (nb + 7) / 8 * 8;
    S.nblocks = (int)nb;
    return S;
}

static KSrc src_of(const ksfd_handle *h, int stage)
{
    KSrc s;
    for (int c = 0; c <= KSFD_MAXL; c++) s.p[c] = (stage >= 0 && c < h->G.F) ? h->src[stage][c] : nullptr;
    return s;
}

// out = f(u) (+sources of `stage`); u must have valid ghosts when size>1
static int op_rhs(ksfd_handle *h, const double *u, int stage, double *out, const KComb *cmb = nullptr)
{
    const KGeom &G = h->G;
    KSrc S = src_of(h, stage);
    if (fused_ok(h)) {
        KStrips K = make_strips(h);
        KComb C = cmb ? *cmb : KComb{};
        Scope sc(h, KC_RHS, vbytes(h, 2 + C.nin + C.nout));
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_rhs2d_fused<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, K, u, S, out, C));
    } else {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (G.F + 1) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_gfield<NL, false>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, h->P, u, (const double *)nullptr, h->Gb, (double *)nullptr));
        }
        int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        Scope sc(h, KC_RHS, vbytes(h, 2) + 8.0 * (double)G.nloc);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_rhs_generic<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, u, h->Gb, S, out));
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// out = J(u) v (mode 0) or shift*v - J(u) v (mode 1); u and v need valid ghosts when size>1
static int op_jvp(ksfd_handle *h, const double *u, const double *v, int mode, double shift, double *out)
{
    const KGeom &G = h->G;
    if (fused_ok(h)) {
        KStrips K = make_strips(h, true);
        Scope sc(h, KC_JVP, vbytes(h, 3));
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_fused<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, K, u, v, mode, shift, out));
    } else {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (2 * G.F + 2) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_gfield<NL, true>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, h->P, u, v, h->Gb, h->dGb));
        }
        int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        Scope sc(h, KC_JVP, vbytes(h, 3) + 16.0 * (double)G.nloc);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jvp_generic<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, u, v, h->Gb, h->dGb, mode, shift, out));
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// Once per step: C = [rho, G, G_rho, G_U..] of the (ghost-filled) state u
static int op_jcoef(ksfd_handle *h, const double *u)
{
    const KGeom &G = h->G;
    int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    if (!h->coef32 && h->poly_fp32 && fused_ok(h) && h->P.nlig <= 4 && G.plane % 2 == 0 && G.inner % 2 == 0 &&
        hipMalloc((void **)&h->coef32, sizeof(float) * (size_t)(3 + h->P.nlig) * G.plane) != hipSuccess) { h->coef32 = nullptr; h->poly_fp32 = false; }
    float *c32 = h->poly_fp32 ? h->coef32 : nullptr;
    Scope sc(h, KC_GFIELD, (8.0 * (G.F + 3 + h->P.nlig) + (c32 ? 4.0 * (3 + h->P.nlig) : 0.0)) * (double)G.plane);
    NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jcoef<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, h->P, u, h->coef, c32));
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// Jacobian action from the frozen coefficients (see stencil.hip.h, "Frozen-Jacobian path")
static int op_jvp_frozen(ksfd_handle *h, const double *v, int mode, double shift, double *out,
                         const double *yadd = nullptr, double alpha = 0.0, double beta = 0.0)
{
    const KGeom &G = h->G;
    const double nplanes = (3 + h->P.nlig) + 2.0 * G.F + ((mode == 2 || mode == 3) ? G.F : 0);   // coefficients + v + out (+ yadd)
    if (fused_ok(h)) {
        KStrips K = make_strips(h, true);
        Scope sc(h, KC_JVP, 8.0 * nplanes * (double)G.nloc);
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, K, (const double *)h->coef, v, mode, shift, out, yadd, alpha, beta));
    } else if (h->use_fused && G.dim == 3 && (G.nx % 2 == 0) && G.nx >= 4 && h->P.nlig <= 4) {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (2 + h->P.nlig + G.F) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dg_frozen<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)h->coef, v, h->dGb));
        }
        K3D K;
        K.nstrips = (int)((G.nx + KSFD_STRIP_OUT - 1) / KSFD_STRIP_OUT);
        K.nygrp = (int)((G.ny + 3) / 4);
        K.zseg = h->zseg;
        {
            long long fit = (long long)K.nstrips * K.nygrp * G.sloc / 1024;      // blocks of 4 waves
            if (fit < 2) fit = 2;
            if (fit < K.zseg) K.zseg = (int)fit;
        }
        K.nzseg = (int)((G.sloc + K.zseg - 1) / K.zseg);
        long long nb3 = (long long)K.nstrips * K.nygrp * K.nzseg;
        K.nblocks = (int)((nb3 + 7) / 8 * 8);
        Scope sc(h, KC_JVP, 8.0 * (2.0 * G.F + 3 + ((mode == 2 || mode == 3) ? G.F : 0)) * (double)G.nloc);
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp3d_frozen<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, K, (const double *)h->coef, v, (const double *)h->dGb, mode, shift, out, yadd, alpha, beta));
    } else {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (2 + h->P.nlig + G.F) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dg_frozen<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)h->coef, v, h->dGb));
        }
        int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        Scope sc(h, KC_JVP, 8.0 * (2.0 * G.F + 3 + ((mode == 2 || mode == 3) ? G.F : 0)) * (double)G.nloc);
        // the generic stencil kernel reads rho from plane 0 of its `u` argument (already clamped in C) and G from C
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jvp_generic<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, (const double *)h->coef, v, (const double *)(h->coef + G.plane), (const double *)h->dGb, mode, shift, out, yadd, alpha, beta));
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// Jacobian action with the halo exchange of v hidden behind the interior rows (slab ranks, 2-D fused kernel):
//   compute stream: [interior segments]                      [two boundary segments]
//   comm stream   :   wait(v ready) -> ghost rows of v <- ring neighbours -> signal
// Interior segments read owned rows only; the first and last segment are the only readers of ghost rows.
static int op_jvp_frozen_halo(ksfd_handle *h, double *v, int mode, double shift, double *out,
                              const double *yadd = nullptr, double alpha = 0.0, double beta = 0.0)
{
    int rc;
    const KGeom &G = h->G;
    if (h->size == 1) return op_jvp_frozen(h, v, mode, shift, out, yadd, alpha, beta);
    KStrips K = make_strips(h, true);
    if (!h->overlap || !fused_ok(h) || K.nseg < 3 || h->P.nlig > 4) {
        if ((rc = halo(h, v))) return rc;
        return op_jvp_frozen(h, v, mode, shift, out, yadd, alpha, beta);
    }
    const double nplanes = (3 + h->P.nlig) + 2.0 * G.F + ((mode == 2 || mode == 3) ? G.F : 0);   // coefficients + v + out (+ yadd)
    const int nseg_total = K.nseg;
    HIPCHK(h, hipEventRecord(h->ev_ready, h->st));
    {
        KStrips Ki = K;
        Ki.seg0 = 1; Ki.seg_stride = 1; Ki.nseg = nseg_total - 2;
        long long nb = ((long long)Ki.nstrips * Ki.nseg + 3) / 4;
        Ki.nblocks = (int)((nb + 7) / 8 * 8);
        Scope sc(h, KC_JVP, 8.0 * nplanes * (double)G.nloc * (double)Ki.nseg / nseg_total);
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL>), dim3(Ki.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, Ki, (const double *)h->coef, (const double *)v, mode, shift, out, yadd, alpha, beta));
    }
    HIPCHK(h, hipStreamWaitEvent(h->st_comm, h->ev_ready, 0));
    {
        Scope sc(h, KC_HALO, 4.0 * 2.0 * 8.0 * G.F * (double)G.inner * 2.0);
        if (h->tr->exchange(v, G.F, G.plane, G.inner, G.sloc, G.ng, h->st_comm)) return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
    }
    HIPCHK(h, hipEventRecord(h->ev_halo, h->st_comm));
    HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_halo, 0));
    {
        KStrips Kb = K;
        Kb.seg0 = 0; Kb.seg_stride = nseg_total - 1; Kb.nseg = 2;
        long long nb = ((long long)Kb.nstrips * Kb.nseg + 3) / 4;
        Kb.nblocks = (int)((nb + 7) / 8 * 8);
        Scope sc(h, KC_JVP, 8.0 * nplanes * (double)G.nloc * 2.0 / nseg_total);
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL>), dim3(Kb.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, Kb, (const double *)h->coef, (const double *)v, mode, shift, out, yadd, alpha, beta));
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// The same strip kernel with mixed storage types (fp32 coefficient copy / Horner temporaries of the polynomial
// preconditioner).  Same overlap scheme as op_jvp_frozen_halo; a float vector travels through the double-typed transport
// as half as many doubles (inner and plane are even on this path).
template <typename TC, typename TV, typename TY, typename TO>
static int jvp2d_launch_t(ksfd_handle *h, const KStrips &K, double frac, const TC *C, const TV *v, int mode, double shift,
                          TO *out, const TY *yadd, double alpha, double beta)
{
    const KGeom &G = h->G;
    const double per_pt = (3.0 + h->P.nlig) * sizeof(TC) + G.F * (double)(sizeof(TV) + sizeof(TO)) + ((mode == 2 || mode == 3) ? G.F * (double)sizeof(TY) : 0.0);
    Scope sc(h, KC_JVP, per_pt * (double)G.nloc * frac);
    NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, TC, TV, TY, TO>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st,
                                                                     G, h->P, K, C, (const TV *)v, mode, shift, out, yadd, alpha, beta));
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

template <typename TC, typename TV, typename TY, typename TO>
static int jvp2d_halo_t(ksfd_handle *h, const TC *C, TV *v, int mode, double shift, TO *out, const TY *yadd, double alpha, double beta)
{
    const KGeom &G = h->G;
    KStrips K = make_strips(h, true);
    if (h->size == 1) return jvp2d_launch_t(h, K, 1.0, C, v, mode, shift, out, yadd, alpha, beta);
    const long long scale = sizeof(double) / sizeof(TV);            // 1 for double, 2 for float
    const bool ovl = h->overlap && K.nseg >= 3;
    int rc;
    if (ovl) {
        HIPCHK(h, hipEventRecord(h->ev_ready, h->st));
        KStrips Ki = K;
        Ki.seg0 = 1; Ki.seg_stride = 1; Ki.nseg = K.nseg - 2;
        long long nb = ((long long)Ki.nstrips * Ki.nseg + 3) / 4;
        Ki.nblocks = (int)((nb + 7) / 8 * 8);
        if ((rc = jvp2d_launch_t(h, Ki, (double)Ki.nseg / K.nseg, C, v, mode, shift, out, yadd, alpha, beta))) return rc;
        HIPCHK(h, hipStreamWaitEvent(h->st_comm, h->ev_ready, 0));
    }
    {
        Scope sc(h, KC_HALO, 4.0 * 2.0 * sizeof(TV) * G.F * (double)G.inner * 2.0);
        if (h->tr->exchange(reinterpret_cast<double *>(v), G.F, G.plane / scale, G.inner / scale, G.sloc, G.ng, ovl ? h->st_comm : h->st))
            return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
    }
    if (!ovl) return jvp2d_launch_t(h, K, 1.0, C, v, mode, shift, out, yadd, alpha, beta);
    HIPCHK(h, hipEventRecord(h->ev_halo, h->st_comm));
    HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_halo, 0));
    KStrips Kb = K;
    Kb.seg0 = 0; Kb.seg_stride = K.nseg - 1; Kb.nseg = 2;
    long long nb = ((long long)Kb.nstrips * Kb.nseg + 3) / 4;
    Kb.nblocks = (int)((nb + 7) / 8 * 8);
    return jvp2d_launch_t(h, Kb, 2.0 / K.nseg, C, v, mode, shift, out, yadd, alpha, beta);
}

// VW = 2 when every plane/offset/length is even (all accesses 16-byte aligned double2)
static inline bool vec2(const ksfd_handle *h) { return (h->G.nloc % 2 == 0) && (h->kv.off % 2 == 0) && (h->G.plane % 2 == 0); }
static inline dim3 vgridw(const ksfd_handle *h, int vw) { return dim3((h->nblk_vec + vw - 1) / vw, h->G.F); }
#define VW_DISPATCH(h, CALL) do { if (vec2(h)) { constexpr int VW = 2; CALL; } else { constexpr int VW = 1; CALL; } } while (0)

static int op_lincomb(ksfd_handle *h, int nt, const double *const *x, const double *a, double *out)
{
    KLin L;
    for (int t = 0; t < 6; t++) { L.x[t] = t < nt ? x[t] : nullptr; L.a[t] = t < nt ? a[t] : 0.0; }
    Scope sc(h, KC_LINCOMB, vbytes(h, nt + 1));
    switch (nt) {
    case 1: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<1, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out)); break;
    case 2: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<2, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out)); break;
    case 3: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<3, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out)); break;
    case 4: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<4, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out)); break;
    case 5: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<5, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out)); break;
    default: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<6, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out)); break;
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// d[0..k) = <w,V_i>, d[k] = <w,w>  -> h->hres
static int op_multidot(ksfd_handle *h, const double *w, const double *V, int k)
{
    const int nb = vec2(h) ? (h->nblk_vec + 1) / 2 : h->nblk_vec;
    {
        Scope sc(h, KC_MULTIDOT, vbytes(h, k + 1));
        if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot<4, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
        else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot<8, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
        else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot<16, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
        else VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot<32, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
    }
    HIPCHK(h, hipGetLastError());
    return reduce_rows(h, k + 1, nb, 0);
}

// d[0..k) = <w,V_i>, g[0..k) = <V_{k-1},V_i>, ww  -> h->hres[0..2k]
static int op_multidot_gram(ksfd_handle *h, const double *w, const double *V, int k)
{
    const int nb = vec2(h) ? (h->nblk_vec + 1) / 2 : h->nblk_vec;
    {
        Scope sc(h, KC_MULTIDOT, vbytes(h, k + 1));
        if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<4, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
        else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<8, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
        else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<16, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
        else VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<32, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, h->part));
    }
    HIPCHK(h, hipGetLastError());
    return reduce_rows(h, 2 * k + 1, nb, 0);
}

static int op_gs_update(ksfd_handle *h, double *w, const double *V, int k, const double *coef, double scale)
{
    KCoef C;
    for (int i = 0; i < KSFD_MAXDOT; i++) C.h[i] = i < k ? coef[i] : 0.0;
    Scope sc(h, KC_GSUPDATE, vbytes(h, k + 2));
    if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update<4, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, C, scale));
    else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update<8, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, C, scale));
    else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update<16, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, C, scale));
    else VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update<32, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, V, h->vlen, k, C, scale));
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

static int op_basis_axpy(ksfd_handle *h, double *x, const double *V, int k, const double *coef, double beta)
{
    KCoef C;
    for (int i = 0; i < KSFD_MAXDOT; i++) C.h[i] = i < k ? coef[i] : 0.0;
    Scope sc(h, KC_BASISAXPY, vbytes(h, k + 1 + (beta != 0.0)));
    if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<4, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta));
    else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<8, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta));
    else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<16, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta));
    else VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<32, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta));
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

static int op_copy(ksfd_handle *h, double *dst, const double *src)
{
    Scope sc(h, KC_MISC, vbytes(h, 2));
    HIPCHK(h, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)h->vlen, hipMemcpyDeviceToDevice, h->st));
    return KSFD_OK;
}

// ---- host <-> device vectors -------------------------------------------------------------------
static int upload(ksfd_handle *h, const double *host, int layout, double *dev)
{
    const KGeom &G = h->G;
    if (layout < 0 || layout > 2) return fail(h, KSFD_EINVAL, "bad layout %d", layout);
    HIPCHK(h, hipMemcpyAsync(h->flat, host, sizeof(double) * (size_t)G.F * G.nloc, hipMemcpyHostToDevice, h->st));
    Scope sc(h, KC_MISC, vbytes(h, 2));
    hipLaunchKernelGGL(k_from_host_layout, vgrid(h), dim3(KSFD_BLOCK), 0, h->st, G, layout, h->flat, dev, G.plane,
                       (long long)G.ng * G.inner);
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}
static int download(ksfd_handle *h, const double *dev, int layout, double *host)
{
    const KGeom &G = h->G;
    if (layout < 0 || layout > 2) return fail(h, KSFD_EINVAL, "bad layout %d", layout);
    {
        Scope sc(h, KC_MISC, vbytes(h, 2));
        hipLaunchKernelGGL(k_to_host_layout, vgrid(h), dim3(KSFD_BLOCK), 0, h->st, G, layout, dev, G.plane,
                           (long long)G.ng * G.inner, h->flat);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(host, h->flat, sizeof(double) * (size_t)G.F * G.nloc, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

static void mg_free(ksfd_handle *h);
static int mg_build(ksfd_handle *h);

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" const char *ksfd_last_error(const ksfd_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" void ksfd_destroy(ksfd_handle *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->st) hipStreamSynchronize(h->st);
    double *bufs[] = { h->Zb, h->pvec, h->coef, h->u, h->usave, h->Z, h->bvec, h->Y, h->V, h->t1, h->t2, h->t3, h->errv, h->Gb, h->dGb, h->flat, h->part, h->dres };
    for (double *b : bufs) if (b) hipFree(b);
    for (int s = 0; s < 4; s++) for (int c = 0; c <= KSFD_MAXL; c++) if (h->src[s][c]) hipFree(h->src[s][c]);
    if (h->coef32) hipFree(h->coef32);
    if (h->hres) hipHostFree(h->hres);
    if (h->pub_count) hipFree(h->pub_count);
    if (h->gm_host) hipHostFree(h->gm_host);
    if (h->gm_dev) hipFree(h->gm_dev);
    for (auto e : h->gm_ev) if (e) hipEventDestroy(e);
    for (auto &p : h->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : h->pool) hipEventDestroy(e);
    if (h->st_io) { hipStreamSynchronize(h->st_io); hipStreamDestroy(h->st_io); }
    for (int q = 0; q < 2; q++) {
        if (h->snap_dev[q]) hipFree(h->snap_dev[q]);
        if (h->snap_host[q]) hipHostFree(h->snap_host[q]);
        if (h->snap_ready[q]) hipEventDestroy(h->snap_ready[q]);
        if (h->snap_done[q]) hipEventDestroy(h->snap_done[q]);
    }
    mg_free(h);
    delete h->tr;
    if (h->ev_ready) hipEventDestroy(h->ev_ready);
    if (h->ev_halo) hipEventDestroy(h->ev_halo);
    if (h->st_comm) hipStreamDestroy(h->st_comm);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

extern "C" int ksfd_create(const ksfd_config *cfg, const ksfd_dist *dist, ksfd_handle **out)
{
    if (!cfg || !out) return fail(nullptr, KSFD_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->dim < 1 || cfg->dim > 3) return fail(nullptr, KSFD_EINVAL, "dim must be 1, 2 or 3");
    for (int a = 0; a < cfg->dim; a++)
        if (cfg->n[a] < 5) return fail(nullptr, KSFD_EINVAL, "n[%d]=%lld < 5: the width-2 periodic star needs >= 5 points", a, (long long)cfg->n[a]);
    ksfd_handle *h = new ksfd_handle();
    memset(h->src, 0, sizeof h->src);
    memset(&h->prof, 0, sizeof h->prof);
    h->cfg = *cfg;
    // the caller keeps ownership of the tables: the handle holds numeric copies (fill_phys), never these pointers
    h->cfg.lig_group = nullptr; h->cfg.lig_w = h->cfg.lig_s = h->cfg.lig_gamma = h->cfg.lig_D = nullptr;
    h->cfg.grp_alpha = h->cfg.grp_beta = nullptr;
    int rc = fill_phys(h, cfg);
    if (rc) { g_create_error = h->err; delete h; return rc; }
    h->rank = dist ? dist->rank : 0;
    h->size = dist ? dist->size : 1;
    h->device = dist ? dist->device : 0;
    if (h->size < 1 || h->rank < 0 || h->rank >= h->size) { delete h; return fail(nullptr, KSFD_EINVAL, "bad rank/size"); }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { delete h; return fail(nullptr, KSFD_EHIP, "no HIP device available (libksfd_hip has no CPU path)"); }
    if (h->device < 0 || h->device >= ndev) { delete h; return fail(nullptr, KSFD_EINVAL, "device %d of %d", h->device, ndev); }
#define CFAIL(code, ...) do { int rc_ = fail(nullptr, code, __VA_ARGS__); ksfd_destroy(h); return rc_; } while (0)
    if (hipSetDevice(h->device) != hipSuccess) CFAIL(KSFD_EHIP, "hipSetDevice(%d) failed", h->device);
    if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) CFAIL(KSFD_EHIP, "hipStreamCreate failed");
    if (h->size > 1 && (hipStreamCreateWithFlags(&h->st_comm, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming) != hipSuccess)) CFAIL(KSFD_EHIP, "comm stream/event creation failed");

    KGeom &G = h->G;
    G.dim = cfg->dim; G.F = cfg->nlig + 1;
    const int slow = cfg->dim - 1;
    int64_t nglob = cfg->n[slow];
    if (h->size > 1 && (nglob % h->size != 0 || nglob / h->size < 4))
        CFAIL(KSFD_EINVAL, "slab axis extent %lld must be divisible by %d ranks with >= 4 units each", (long long)nglob, h->size);
    int64_t sloc = nglob / h->size;
    h->slow0 = sloc * h->rank;
    G.ng = h->size > 1 ? 2 : 0;
    G.wrap_slow = h->size == 1;
    G.nx = cfg->n[0]; G.ny = cfg->dim >= 2 ? cfg->n[1] : 1; G.nz = cfg->dim >= 3 ? cfg->n[2] : 1;
    if (slow == 0) G.nx = sloc; else if (slow == 1) G.ny = sloc; else G.nz = sloc;
    G.inner = slow == 0 ? 1 : (slow == 1 ? G.nx : G.nx * G.ny);
    G.sloc = sloc;
    G.plane = (sloc + 2 * G.ng) * G.inner;
    G.nloc = sloc * G.inner;
    h->kv.plane = G.plane; h->kv.off = (long long)G.ng * G.inner; h->kv.nloc = G.nloc; h->kv.nf = G.F;
    h->vlen = (int64_t)G.F * G.plane;
    h->nblk_vec = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 2048);
    build_tableau(h);

    h->restart_alloc = 30;
    double **vecs[] = { &h->u, &h->usave, &h->Z, &h->bvec, &h->t1, &h->t2, &h->t3, &h->errv };
    for (double **v : vecs) {
        if (alloc_d(h, v, h->vlen)) CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
        hipMemsetAsync(*v, 0, sizeof(double) * (size_t)h->vlen, h->st);
    }
    if (alloc_d(h, &h->Y, 4 * h->vlen) || alloc_d(h, &h->V, (int64_t)(h->restart_alloc + 1) * h->vlen) ||
        alloc_d(h, &h->Gb, G.plane) || alloc_d(h, &h->dGb, G.plane) || alloc_d(h, &h->coef, (int64_t)(3 + cfg->nlig) * G.plane) ||
        alloc_d(h, &h->flat, (int64_t)std::max(G.F, 3) * G.nloc) ||
        alloc_d(h, &h->part, (int64_t)(2 * KSFD_MAXDOT + 4) * 4096) || alloc_d(h, &h->dres, 128))
        CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
    hipMemsetAsync(h->Y, 0, sizeof(double) * (size_t)(4 * h->vlen), h->st);
    hipMemsetAsync(h->V, 0, sizeof(double) * (size_t)((h->restart_alloc + 1) * h->vlen), h->st);
    if (hipHostMalloc((void **)&h->hres, sizeof(double) * 128 + 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) CFAIL(KSFD_ENOMEM, "hipHostMalloc failed");
    h->pub_flag = reinterpret_cast<unsigned long long *>(h->hres + 128);
    *h->pub_flag = 0;
    if (hipHostGetDevicePointer((void **)&h->hres_dev, h->hres, 0) != hipSuccess || hipMalloc((void **)&h->pub_count, sizeof(unsigned int)) != hipSuccess ||
        hipMemset(h->pub_count, 0, sizeof(unsigned int)) != hipSuccess) { h->zero_copy = false; h->hres_dev = nullptr; }
    else h->pub_flag_dev = reinterpret_cast<unsigned long long *>(h->hres_dev + 128);
    {
        const int m = h->restart_alloc;
        const size_t ndev = (size_t)(m + 1) * (m + 1) + (size_t)(m + 1) * m + 2 * m + (m + 1) + KSFD_MAXDOT + 1 + 2 * (m + 1);
        const size_t nhost = 2 * (m + 1) + (size_t)(m + 1) * m + (m + 1);
        if (alloc_d(h, &h->gm_dev, (int64_t)ndev)) CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
        if (hipHostMalloc((void **)&h->gm_host, sizeof(double) * nhost, hipHostMallocDefault) != hipSuccess) CFAIL(KSFD_ENOMEM, "hipHostMalloc failed");
        h->gm_ev.resize(m + 1);
        for (auto &e : h->gm_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) CFAIL(KSFD_EHIP, "hipEventCreate failed");
    }

    if (h->size > 1) {
        std::string terr;
        h->tr = make_transport(dist, G.F, G.inner, terr);
        if (!h->tr) CFAIL(KSFD_ECOMM, "transport %d: %s", dist->transport, terr.c_str());
    }
    if (mg_build(h)) CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
    if (hipStreamSynchronize(h->st) != hipSuccess) CFAIL(KSFD_EHIP, "stream sync failed in create");
#undef CFAIL
    *out = h;
    return KSFD_OK;
}

extern "C" int ksfd_rccl_unique_id(void *out128)
{
    if (!out128) return KSFD_EINVAL;
    std::string err;
    RcclApi api;
    if (!api.load(err)) return fail(nullptr, KSFD_ECOMM, "%s", err.c_str());
    auto getid = (decltype(&ncclGetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    if (!getid) return fail(nullptr, KSFD_ECOMM, "librccl lacks ncclGetUniqueId");
    ncclUniqueId id;
    ncclResult_t r = getid(&id);
    if (r != ncclSuccess) return fail(nullptr, KSFD_ECOMM, "ncclGetUniqueId: %s", api.GetErrorString(r));
    memcpy(out128, &id, sizeof id);
    return KSFD_OK;
}

extern "C" int ksfd_update_params(ksfd_handle *h, const ksfd_config *cfg)
{
    if (!h || !cfg) return KSFD_EINVAL;
    if (cfg->dim != h->cfg.dim || cfg->nlig != h->cfg.nlig) return fail(h, KSFD_EINVAL, "update_params cannot change dim/nlig");
    for (int a = 0; a < 3; a++) if (cfg->n[a] != h->cfg.n[a]) return fail(h, KSFD_EINVAL, "update_params cannot change the grid");
    h->cfg = *cfg;
    h->cfg.lig_group = nullptr; h->cfg.lig_w = h->cfg.lig_s = h->cfg.lig_gamma = h->cfg.lig_D = nullptr;
    h->cfg.grp_alpha = h->cfg.grp_beta = nullptr;
    return fill_phys(h, cfg);
}

extern "C" int ksfd_local_range(const ksfd_handle *h, int64_t *b, int64_t *e)
{
    if (!h) return KSFD_EINVAL;
    if (b) *b = h->slow0;
    if (e) *e = h->slow0 + h->G.sloc;
    return KSFD_OK;
}
extern "C" int64_t ksfd_local_size(const ksfd_handle *h) { return h ? (int64_t)h->G.F * h->G.nloc : 0; }
extern "C" double *ksfd_device_state(ksfd_handle *h) { return h ? h->u : nullptr; }
extern "C" int64_t ksfd_device_plane_stride(const ksfd_handle *h) { return h ? h->G.plane : 0; }
extern "C" int64_t ksfd_device_interior_offset(const ksfd_handle *h) { return h ? (int64_t)h->G.ng * h->G.inner : 0; }

extern "C" int ksfd_set_state(ksfd_handle *h, const double *u, int32_t layout)
{
    if (!h || !u) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int rc = upload(h, u, layout, h->u);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}
// Asynchronous read-out of the resident state for writers ("next" row f2: TimeSeries fed from the device without
// stalling the stepper).  _begin: the layout transform runs on the compute stream (ordered after everything issued so
// far, ~0.1 ms at 4096^2) into one of two device staging slots; the D2H copy into pinned memory runs on a third stream.
// _wait: blocks until that copy has landed and hands out the pinned buffer (F*nlocal doubles), valid until the slot is
// used again, i.e. until the second-next _begin.  A slot whose data nobody waited for is simply overwritten.
extern "C" int ksfd_snapshot_begin(ksfd_handle *h, int32_t layout, int32_t *slot)
{
    if (!h || !slot) return KSFD_EINVAL;
    if (layout < 0 || layout > 2) return fail(h, KSFD_EINVAL, "bad layout %d", layout);
    hipSetDevice(h->device);
    const KGeom &G = h->G;
    const size_t bytes = sizeof(double) * (size_t)G.F * G.nloc;
    if (!h->st_io) {
        if (hipStreamCreateWithFlags(&h->st_io, hipStreamNonBlocking) != hipSuccess) return fail(h, KSFD_EHIP, "hipStreamCreate (io) failed");
        for (int q = 0; q < 2; q++) {
            if (hipMalloc((void **)&h->snap_dev[q], bytes) != hipSuccess || hipHostMalloc((void **)&h->snap_host[q], bytes, hipHostMallocDefault) != hipSuccess)
                return fail(h, KSFD_ENOMEM, "snapshot staging buffers (%zu bytes each) could not be allocated", bytes);
            if (hipEventCreateWithFlags(&h->snap_ready[q], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->snap_done[q], hipEventDisableTiming) != hipSuccess) return fail(h, KSFD_EHIP, "hipEventCreate failed");
        }
    }
    const int q = h->snap_next;
    h->snap_next ^= 1;
    if (h->snap_busy[q]) HIPCHK(h, hipEventSynchronize(h->snap_done[q]));      // its previous copy must have left the staging slot
    {
        Scope sc(h, KC_MISC, vbytes(h, 2));
        hipLaunchKernelGGL(k_to_host_layout, vgrid(h), dim3(KSFD_BLOCK), 0, h->st, G, layout, (const double *)h->u, G.plane,
                           (long long)G.ng * G.inner, h->snap_dev[q]);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->snap_ready[q], h->st));
    HIPCHK(h, hipStreamWaitEvent(h->st_io, h->snap_ready[q], 0));
    HIPCHK(h, hipMemcpyAsync(h->snap_host[q], h->snap_dev[q], bytes, hipMemcpyDeviceToHost, h->st_io));
    HIPCHK(h, hipEventRecord(h->snap_done[q], h->st_io));
    h->snap_busy[q] = true;
    *slot = q;
    return KSFD_OK;
}

extern "C" int ksfd_snapshot_wait(ksfd_handle *h, int32_t slot, const double **host)
{
    if (!h || !host || slot < 0 || slot > 1) return KSFD_EINVAL;
    if (!h->snap_busy[slot]) return fail(h, KSFD_EINVAL, "snapshot slot %d holds nothing", slot);
    hipSetDevice(h->device);
    HIPCHK(h, hipEventSynchronize(h->snap_done[slot]));
    *host = h->snap_host[slot];
    return KSFD_OK;
}

extern "C" int ksfd_set_state_random(ksfd_handle *h, const int64_t *nc, const double *z, double rho0)
{
    if (!h || !nc || !z) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int64_t n = 1;
    for (int a = 0; a < 3; a++) {
        if (a < h->G.dim ? nc[a] < 1 : nc[a] != 1) return fail(h, KSFD_EINVAL, "coarse grid must be >= 1 per used axis and 1 elsewhere");
        n *= nc[a];
    }
    double *dz = nullptr;
    if (hipMalloc((void **)&dz, sizeof(double) * (size_t)n) != hipSuccess) return fail(h, KSFD_ENOMEM, "hipMalloc of the coarse samples failed");
    hipError_t e = hipMemcpyAsync(dz, z, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->st);
    const KGeom &G = h->G;
    int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 65535);
    if (e == hipSuccess) {
        Scope sc(h, KC_MISC, vbytes(h, 1));
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_random_start<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, (long long)h->cfg.n[G.dim - 1],
                                                  (long long)h->slow0, (long long)nc[0], (long long)nc[1], (long long)nc[2], (const double *)dz, rho0, h->u));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    hipFree(dz);
    if (e != hipSuccess) return fail(h, KSFD_EHIP, "random start: %s", hipGetErrorString(e));
    return KSFD_OK;
}
extern "C" int ksfd_get_state(ksfd_handle *h, double *u, int32_t layout)
{
    if (!h || !u) return KSFD_EINVAL;
    hipSetDevice(h->device);
    return download(h, h->u, layout, u);
}

extern "C" int ksfd_set_source(ksfd_handle *h, int32_t stage, int32_t field, const double *srch, int32_t layout)
{
    if (!h || field < 0 || field >= h->G.F || stage < -1 || stage > 3) return h ? fail(h, KSFD_EINVAL, "bad stage/field") : KSFD_EINVAL;
    if (layout != KSFD_LAYOUT_SOA) return fail(h, KSFD_EINVAL, "sources are single dense planes: pass layout SOA");
    hipSetDevice(h->device);
    for (int s = 0; s < 4; s++) {
        if (stage != -1 && s != stage) continue;
        if (!srch) {
            if (h->src[s][field]) { HIPCHK(h, hipStreamSynchronize(h->st)); hipFree(h->src[s][field]); h->src[s][field] = nullptr; }
            continue;
        }
        if (!h->src[s][field] && alloc_d(h, &h->src[s][field], h->G.nloc)) return KSFD_ENOMEM;
        HIPCHK(h, hipMemcpyAsync(h->src[s][field], srch, sizeof(double) * (size_t)h->G.nloc, hipMemcpyHostToDevice, h->st));
    }
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

extern "C" int ksfd_rhs(ksfd_handle *h, double t, const double *uh, double *outh, int32_t layout)
{
    (void)t;   // sources(t) are uploaded by the caller (ksfd_set_source); constants via ksfd_update_params
    if (!h || !outh) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int rc;
    double *uin = h->u;
    if (uh) { if ((rc = upload(h, uh, layout, h->t1))) return rc; uin = h->t1; }
    if ((rc = halo(h, uin))) return rc;
    if ((rc = op_rhs(h, uin, 0, h->t3))) return rc;
    return download(h, h->t3, layout, outh);
}

extern "C" int ksfd_jvp(ksfd_handle *h, const double *uh, const double *vh, double *outh, int32_t layout)
{
    if (!h || !vh || !outh) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int rc;
    double *uin = h->u;
    if (uh) { if ((rc = upload(h, uh, layout, h->t1))) return rc; uin = h->t1; }
    if ((rc = upload(h, vh, layout, h->t2))) return rc;
    if ((rc = halo(h, uin)) || (rc = halo(h, h->t2))) return rc;
    if (!uh && h->use_frozen) {
        // the path the stepper uses: coefficients of the stored state once, then the frozen-coefficient kernels
        if ((rc = op_jcoef(h, h->u))) return rc;
        h->mg_coef_valid = false; h->mg_shift = -1.0;
        if ((rc = op_jvp_frozen(h, h->t2, 0, 0.0, h->t3))) return rc;
    } else if ((rc = op_jvp(h, uin, h->t2, 0, 0.0, h->t3))) return rc;
    return download(h, h->t3, layout, outh);
}

static int velocity_common(ksfd_handle *h, const double *uin, double *vel_dev, double vmax[3])
{
    const KGeom &G = h->G;
    int rc;
    if ((rc = halo(h, (double *)uin))) return rc;
    int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    {
        Scope sc(h, KC_GFIELD, 8.0 * (G.F + 1) * (double)G.plane);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_gfield<NL, false>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, h->P, uin, (const double *)nullptr, h->Gb, (double *)nullptr));
    }
    int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 1024);
    {
        Scope sc(h, KC_VELOCITY, 8.0 * (double)G.nloc * (1 + (vel_dev ? G.dim : 0)));
        hipLaunchKernelGGL(k_velocity, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, h->Gb, vel_dev, vmax ? h->part : (double *)nullptr);
    }
    HIPCHK(h, hipGetLastError());
    if (vmax) {
        if ((rc = reduce_rows(h, 3, nb, 1))) return rc;
        for (int a = 0; a < 3; a++) vmax[a] = a < G.dim ? h->hres[a] : 0.0;
    }
    return KSFD_OK;
}

extern "C" int ksfd_velocity(ksfd_handle *h, const double *uh, double *velh, int32_t layout)
{
    if (!h || !velh) return KSFD_EINVAL;
    if (layout != KSFD_LAYOUT_SOA) return fail(h, KSFD_EINVAL, "velocity output is dim dense SoA planes: pass layout SOA for it");
    hipSetDevice(h->device);
    int rc;
    double *uin = h->u;
    if (uh) { if ((rc = upload(h, uh, layout, h->t1))) return rc; uin = h->t1; }
    if ((rc = velocity_common(h, uin, h->flat, nullptr))) return rc;
    HIPCHK(h, hipMemcpyAsync(velh, h->flat, sizeof(double) * (size_t)h->G.dim * h->G.nloc, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

extern "C" int ksfd_velocity_max(ksfd_handle *h, double vmax[3])
{
    if (!h || !vmax) return KSFD_EINVAL;
    hipSetDevice(h->device);
    return velocity_common(h, h->u, nullptr, vmax);
}

extern "C" int ksfd_groom(ksfd_handle *h)
{
    if (!h) return KSFD_EINVAL;
    hipSetDevice(h->device);
    Scope sc(h, KC_MISC, vbytes(h, 2));
    hipLaunchKernelGGL(k_groom, vgrid(h), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, h->P.rhomin, h->P.Umin);
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

extern "C" int ksfd_count_worms(ksfd_handle *h, double *total)
{
    if (!h || !total) return KSFD_EINVAL;
    hipSetDevice(h->device);
    {
        Scope sc(h, KC_MISC, 8.0 * (double)h->G.nloc);
        hipLaunchKernelGGL(k_sum_rho, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, h->part);
    }
    HIPCHK(h, hipGetLastError());
    int rc = reduce_rows(h, 1, h->nblk_vec, 0);
    if (rc) return rc;
    *total = h->hres[0];
    return KSFD_OK;
}

extern "C" int ksfd_scale_rho(ksfd_handle *h, double factor)
{
    if (!h) return KSFD_EINVAL;
    hipSetDevice(h->device);
    Scope sc(h, KC_MISC, 16.0 * (double)h->G.nloc);
    hipLaunchKernelGGL(k_mul_rho, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, (const double *)nullptr, factor);
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

extern "C" int ksfd_mul_rho(ksfd_handle *h, const double *fh)
{
    if (!h || !fh) return KSFD_EINVAL;
    hipSetDevice(h->device);
    HIPCHK(h, hipMemcpyAsync(h->flat, fh, sizeof(double) * (size_t)h->G.nloc, hipMemcpyHostToDevice, h->st));
    Scope sc(h, KC_MISC, 24.0 * (double)h->G.nloc);
    hipLaunchKernelGGL(k_mul_rho, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, (const double *)h->flat, 1.0);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

// Assembled Jacobian export (row f4 of the scope table; kernel k_jac_csr in stencil.hip.h)
extern "C" int ksfd_jacobian_nnz(ksfd_handle *h, int64_t *nrows, int64_t *nnz)
{
    if (!h) return KSFD_EINVAL;
    const int64_t npts = 4 * h->G.dim + 1;
    if (nrows) *nrows = (int64_t)h->G.F * h->G.nloc;
    if (nnz) *nnz = h->G.nloc * ((int64_t)h->G.F * npts + (int64_t)h->P.nlig * (npts + 1));
    return KSFD_OK;
}

extern "C" int ksfd_jacobian_csr(ksfd_handle *h, int64_t *rowptr, int64_t *col, double *val)
{
    if (!h || !rowptr || !col || !val) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int64_t nrows, nnz;
    ksfd_jacobian_nnz(h, &nrows, &nnz);
    int rc;
    if ((rc = halo(h, h->u))) return rc;
    if ((rc = op_jcoef(h, h->u))) return rc;                 // clamps like groom; the frozen planes are rebuilt by the next step anyway
    h->mg_coef_valid = false; h->mg_shift = -1.0;
    long long *dcol = nullptr;
    double *dval = nullptr;
    if (hipMalloc((void **)&dcol, sizeof(long long) * (size_t)nnz) != hipSuccess ||
        hipMalloc((void **)&dval, sizeof(double) * (size_t)nnz) != hipSuccess) {
        if (dcol) hipFree(dcol);
        return fail(h, KSFD_ENOMEM, "hipMalloc of the CSR staging buffers failed");
    }
    const KGeom &G = h->G;
    int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 65535);
    {
        Scope sc(h, KC_MISC, 16.0 * (double)nnz + 8.0 * (3 + h->P.nlig) * (double)G.nloc);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jac_csr<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, (const double *)h->coef,
                                                  (long long)h->cfg.n[G.dim - 1], (long long)h->slow0, dcol, dval));
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(col, dcol, sizeof(long long) * (size_t)nnz, hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(val, dval, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    hipFree(dcol);
    hipFree(dval);
    if (e != hipSuccess) return fail(h, KSFD_EHIP, "jacobian export: %s", hipGetErrorString(e));
    // row pointers are a fixed pattern: rho row F*npts entries, each U row npts+1
    const int64_t npts = 4 * G.dim + 1, F = G.F, per = F * npts + (F - 1) * (npts + 1);
    for (int64_t p = 0; p < G.nloc; p++) {
        rowptr[p * F] = p * per;
        for (int64_t l = 1; l < F; l++) rowptr[p * F + l] = p * per + F * npts + (l - 1) * (npts + 1);
    }
    rowptr[nrows] = nnz;
    return KSFD_OK;
}

// ------------------------------------------------------------------------------------------------
// multigrid preconditioner (host side; kernels and rationale in mg.hip.h)
// ------------------------------------------------------------------------------------------------
static void mg_free(ksfd_handle *h)
{
    if (h->mg_graph) { hipGraphExecDestroy(h->mg_graph); h->mg_graph = nullptr; }
    for (size_t l = 0; l < h->mg.size(); l++) {
        MGLevel &L = h->mg[l];
        double *bufs[] = { l ? L.coef : nullptr, L.dinv, l ? L.x : nullptr, l ? L.b : nullptr, L.r, L.d, L.Ad, L.dG };
        for (double *b : bufs) if (b) hipFree(b);
    }
    h->mg.clear();
    h->mg_ok = false;
}

static int mg_build(ksfd_handle *h)
{
    if (h->G.dim < 2) return KSFD_OK;
    const int dim = h->G.dim;
    int nl = h->P.nlig, F = h->G.F;
    long long nx = h->G.nx, ny = dim == 3 ? h->G.ny : 1, rows = h->G.sloc;   // rows = local slow units (y rows in 2-D, z planes in 3-D)
    // NOTE: every decision below must be identical on all ranks (the levels exchange halos): use sloc, never slow0.
    // Slab r starts at unit r*sloc; it stays on the coarse grid of level l as long as sloc is divisible by 2^l.
    KPhys P = h->P;
    for (int l = 0;; l++) {
        MGLevel L;
        L.G = h->G; L.G.nx = nx;
        if (dim == 2) { L.G.ny = rows; L.G.inner = nx; } else { L.G.ny = ny; L.G.nz = rows; L.G.inner = nx * ny; }
        L.G.sloc = rows;
        L.G.plane = (rows + 2 * L.G.ng) * L.G.inner; L.G.nloc = rows * L.G.inner;
        L.P = P;
        L.kv.plane = L.G.plane; L.kv.off = (long long)L.G.ng * L.G.inner; L.kv.nloc = L.G.nloc; L.kv.nf = F;
        L.vlen = (int64_t)F * L.G.plane;
        L.nblk = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 2048);
        if (l == 0) L.coef = h->coef;
        else if (alloc_d(h, &L.coef, (int64_t)(3 + nl) * L.G.plane) || alloc_d(h, &L.x, L.vlen) || alloc_d(h, &L.b, L.vlen)) return KSFD_ENOMEM;
        if (alloc_d(h, &L.dinv, (int64_t)F * F * L.G.plane) || alloc_d(h, &L.r, L.vlen) || alloc_d(h, &L.d, L.vlen) ||
            alloc_d(h, &L.Ad, L.vlen) || alloc_d(h, &L.dG, L.G.plane)) return KSFD_ENOMEM;
        double *zero[] = { l ? L.x : nullptr, l ? L.b : nullptr, L.r, L.d, L.Ad };
        for (double *z : zero) if (z) hipMemsetAsync(z, 0, sizeof(double) * (size_t)L.vlen, h->st);
        h->mg.push_back(L);
        // next level: every rank keeps >= 4 slow units (ghost width 2 + the 4th-order star), global grid >= 8 per axis
        const long long rows_glob = rows * h->size;
        if ((nx % 2) || (rows % 2) || nx / 2 < 8 || rows_glob / 2 < 8 || (h->size > 1 && rows / 2 < 4)) break;
        if (dim == 3 && ((ny % 2) || ny / 2 < 8)) break;
        nx /= 2; rows /= 2;
        if (dim == 3) ny /= 2;
        for (int a = 0; a < 3; a++) { P.inv_h[a] *= 0.5; P.inv_h2[a] *= 0.25; }
    }
    h->mg_ok = h->mg.size() >= 2;
    if (h->size > 1) h->mg_use_graph = false;           // collectives inside the cycle: keep eager launches
    return KSFD_OK;
}

// transfer operators, 2-D or 3-D by the level geometry
static void mg_launch_restrict(ksfd_handle *h, MGLevel &Lf, MGLevel &Lc, int np, const double *fine, double *coarse)
{
    int nb = (int)std::min<long long>((Lc.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    if (Lf.G.dim == 3)
        hipLaunchKernelGGL(k_restrict3d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.ny, Lf.G.sloc, Lf.G.wrap_slow,
                           fine, Lf.G.plane, Lf.kv.off, coarse, Lc.G.plane, Lc.kv.off);
    else
        hipLaunchKernelGGL(k_restrict2d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.sloc, Lf.G.wrap_slow,
                           fine, Lf.G.plane, Lf.kv.off, coarse, Lc.G.plane, Lc.kv.off);
}
static void mg_launch_prolong(ksfd_handle *h, MGLevel &Lf, MGLevel &Lc, int np, const double *coarse, double *fine)
{
    int nb = (int)std::min<long long>((Lf.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    if (Lf.G.dim == 3)
        hipLaunchKernelGGL(k_prolong_add3d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.ny, Lf.G.sloc, Lf.G.wrap_slow,
                           coarse, Lc.G.plane, Lc.kv.off, fine, Lf.G.plane, Lf.kv.off);
    else
        hipLaunchKernelGGL(k_prolong_add2d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.sloc, Lf.G.wrap_slow,
                           coarse, Lc.G.plane, Lc.kv.off, fine, Lf.G.plane, Lf.kv.off);
}

// ghost rows of a level vector (np field planes) from the ring neighbours
static int mg_halo(ksfd_handle *h, MGLevel &L, double *v, int np)
{
    if (h->size == 1) return KSFD_OK;
    Scope sc(h, KC_HALO, 4.0 * 8.0 * np * (double)L.G.inner * 2.0);
    if (h->tr->exchange(v, np, L.G.plane, L.G.inner, L.G.sloc, L.G.ng, h->st)) return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
    return KSFD_OK;
}

// out = J v | shift v - J v | yadd - (shift v - J v) on level L
static int mg_op(ksfd_handle *h, MGLevel &L, const double *v, int mode, double shift, double *out, const double *yadd)
{
    const KGeom &G = L.G;
    if (h->size > 1) { int rch = mg_halo(h, L, const_cast<double *>(v), G.F); if (rch) return rch; }
    const int cls = (&L == &h->mg[0]) ? KC_JVP : KC_MG;
    const double by = 8.0 * ((3 + h->P.nlig) + 2.0 * G.F + (mode == 2 ? G.F : 0)) * (double)G.nloc;
    if (G.dim == 2 && h->use_fused && (G.nx % 2 == 0) && G.nx >= 16 && h->P.nlig <= 4) {
        KStrips K;
        K.nstrips = (int)((G.nx + KSFD_STRIP_OUT - 1) / KSFD_STRIP_OUT);
        K.yseg = h->yseg_jvp;
        {
            long long fit = (long long)K.nstrips * G.sloc / 4096;
            if (fit < 2) fit = 2;
            if (fit < K.yseg) K.yseg = (int)fit;
        }
        K.nseg = (int)((G.sloc + K.yseg - 1) / K.yseg);
        K.seg0 = 0; K.seg_stride = 1;
        long long nb = ((long long)K.nstrips * K.nseg + 3) / 4;
        K.nblocks = (int)((nb + 7) / 8 * 8);
        Scope sc(h, cls, by);
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, (const double *)L.coef, v, mode, shift, out, yadd));
    } else if (G.dim == 3 && h->use_fused && (G.nx % 2 == 0) && G.nx >= 16 && h->P.nlig <= 4) {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        K3D K;
        K.nstrips = (int)((G.nx + KSFD_STRIP_OUT - 1) / KSFD_STRIP_OUT);
        K.nygrp = (int)((G.ny + 3) / 4);
        K.zseg = h->zseg;
        {
            long long fit = (long long)K.nstrips * K.nygrp * G.sloc / 1024;
            if (fit < 2) fit = 2;
            if (fit < K.zseg) K.zseg = (int)fit;
        }
        K.nzseg = (int)((G.sloc + K.zseg - 1) / K.zseg);
        long long nb3 = (long long)K.nstrips * K.nygrp * K.nzseg;
        K.nblocks = (int)((nb3 + 7) / 8 * 8);
        Scope sc(h, cls, by + 8.0 * G.plane);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dg_frozen<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)L.coef, v, L.dG));
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp3d_frozen<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, (const double *)L.coef, v, (const double *)L.dG, mode, shift, out, yadd));
    } else {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        Scope sc(h, cls, by + 8.0 * G.plane);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dg_frozen<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)L.coef, v, L.dG));
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jvp_generic<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, L.P, (const double *)L.coef, v, (const double *)(L.coef + G.plane), (const double *)L.dG, mode, shift, out, yadd));
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

static int mg_norm(ksfd_handle *h, MGLevel &L, const double *v, double *nrm)
{
    const bool v2 = (L.G.nloc % 2 == 0);
    const int nb = v2 ? (L.nblk + 1) / 2 : L.nblk;
    {
        Scope sc(h, KC_MG, 8.0 * L.vlen);
        if (v2) hipLaunchKernelGGL((k_multidot<4, 2>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.kv, v, v, L.vlen, 0, h->part);
        else hipLaunchKernelGGL((k_multidot<4, 1>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.kv, v, v, L.vlen, 0, h->part);
    }
    HIPCHK(h, hipGetLastError());
    int rc = reduce_rows(h, 1, nb, 0);
    if (rc) return rc;
    *nrm = sqrt(h->hres[0]);
    return KSFD_OK;
}

// restrict coefficient planes down the hierarchy (once per frozen state)
static int mg_restrict_coefs(ksfd_handle *h)
{
    const int np = 3 + h->P.nlig;
    int rc;
    for (size_t l = 0; l + 1 < h->mg.size(); l++) {
        MGLevel &Lf = h->mg[l], &Lc = h->mg[l + 1];
        {
            Scope sc(h, KC_MG, 8.0 * np * (Lf.G.nloc + Lc.G.nloc));
            mg_launch_restrict(h, Lf, Lc, np, Lf.coef, Lc.coef);
        }
        if ((rc = mg_halo(h, Lc, Lc.coef, np))) return rc;       // fine ghosts were valid; now the coarse ones are too
    }
    HIPCHK(h, hipGetLastError());
    h->mg_coef_valid = true;
    return KSFD_OK;
}

// block-diagonal inverses and Chebyshev upper bounds for this shift
static int mg_setup_shift(ksfd_handle *h, double shift)
{
    int rc;
    for (size_t l = 0; l < h->mg.size(); l++) {
        MGLevel &L = h->mg[l];
        const int F = L.G.F;
        int nb = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_MG, 8.0 * (3 + h->P.nlig + F * F) * L.G.nloc);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_blockdiag_inv<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G, L.P, (const double *)L.coef, shift, L.dinv));
        }
        HIPCHK(h, hipGetLastError());
        // power iteration on Dinv*A: v in L.d, A v in L.Ad, Dinv A v in L.r
        hipLaunchKernelGGL(k_hash_fill, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, (long long)L.vlen, L.d);
        double nv = 1.0, lam = 2.0;
        if ((rc = mg_norm(h, L, L.d, &nv))) return rc;
        for (int it = 0; it < h->mg_power_its; it++) {
            if ((rc = mg_op(h, L, L.d, 1, shift, L.Ad, nullptr))) return rc;
            {
                Scope sc(h, KC_MG, 8.0 * (2 * F + F * F) * L.G.nloc);
                NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const double *)(L.dinv + L.kv.off), (const double *)(L.Ad + L.kv.off), 1.0, L.r + L.kv.off));
            }
            double nw;
            if ((rc = mg_norm(h, L, L.r, &nw))) return rc;
            if (!(nw > 0.0) || !(nv > 0.0)) break;
            lam = nw / nv;
            // v <- w / |w|
            Scope sc(h, KC_MG, 16.0 * L.vlen);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const double *)(L.dinv + L.kv.off), (const double *)(L.Ad + L.kv.off), 1.0 / nw, L.d + L.kv.off));
            nv = 1.0;
        }
        L.lam_max = 1.15 * lam;
        if (l + 1 == h->mg.size()) {
            int nbr = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 256);
            hipLaunchKernelGGL(k_ratio_est, dim3(nbr), dim3(KSFD_BLOCK), 0, h->st, (long long)L.G.nloc, (const double *)(L.dinv + L.kv.off), shift, h->part);
            if ((rc = reduce_rows(h, 1, nbr, 1))) return rc;
            L.ratio = std::max(30.0, 1.5 * L.lam_max * h->hres[0]);
        }
    }
    h->mg_shift = shift;
    h->mg_graph_shift = -1.0;        // Chebyshev bounds changed: the captured coarse cycle is stale
    return KSFD_OK;
}

// Chebyshev smoothing of A x = b on level L with Dinv; nu sweeps; eigen-interval [lmax/ratio, lmax]
static int mg_smooth(ksfd_handle *h, MGLevel &L, double shift, const double *b, double *x, int nu, bool zero_init, double ratio)
{
    // Chebyshev iteration in the "direction" form:  d_0 = Dinv r_0 / theta ; x += d_k ; r -= A d_k ;
    // d_{k+1} = c1 d_k + c2 Dinv r.   nu sweeps = nu updates of x = nu-1 operator applications (+1 for a nonzero guess).
    // Fusions: a zero guess writes x = d_0 directly; the last sweep folds "x += d_old + d_new" into one kernel.
    int rc;
    const int F = L.G.F;
    const int nb = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    const double lmax = L.lam_max, lmin = lmax / ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sig1 = theta / delta;
    const long long off = L.kv.off;     // owned rows start here inside a (ghosted) plane
    const double *res = b;
    if (!zero_init) {
        if ((rc = mg_op(h, L, x, 2, shift, L.r, b))) return rc;        // r = b - A x
        res = L.r;
    }
    {
        Scope sc(h, KC_MG, 8.0 * ((zero_init ? 3 : 2) * F + F * F) * L.G.nloc);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const double *)(L.dinv + off), res + off, 1.0 / theta, L.d + off, zero_init ? x + off : (double *)nullptr));
    }
    bool x_has_d = zero_init;          // x == d_0 already
    double rho = 1.0 / sig1;
    for (int k = 1; k < nu; k++) {
        if ((rc = mg_op(h, L, L.d, 1, shift, L.Ad, nullptr))) return rc;
        const double rhon = 1.0 / (2.0 * sig1 - rho);
        const double *rsrc = (zero_init && k == 1) ? b : L.r;             // first sweep from a zero guess: r_0 = b, never copied
        if (k == nu - 1) {
            Scope sc(h, KC_MG, 8.0 * (5 * F + F * F) * L.G.nloc);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_cheb_last<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const double *)(L.dinv + off), x + off, rsrc + off, (const double *)(L.d + off), (const double *)(L.Ad + off), rhon * rho, 2.0 * rhon / delta, x_has_d ? 1 : 0));
            x_has_d = true;
        } else {
            if (rsrc != L.r) HIPCHK(h, hipMemcpyAsync(L.r, b, sizeof(double) * (size_t)L.vlen, hipMemcpyDeviceToDevice, h->st));
            if (x_has_d && k == 1) { /* x already holds d_0: the step kernel adds d to x, so undo by starting x at 0 */
                HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)L.vlen, h->st));
            }
            Scope sc(h, KC_MG, 8.0 * (7 * F + F * F) * L.G.nloc);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_cheb_step<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const double *)(L.dinv + off), x + off, L.r + off, L.d + off, (const double *)(L.Ad + off), rhon * rho, 2.0 * rhon / delta));
            x_has_d = false;
        }
        rho = rhon;
    }
    if (!x_has_d) {
        // x += d (only reached when nu == 1 with a nonzero guess, or after k_cheb_step sweeps)
        const double *xs[2] = { x, L.d };
        KLin LL;
        for (int t = 0; t < 6; t++) { LL.x[t] = t < 2 ? xs[t] : nullptr; LL.a[t] = t < 2 ? 1.0 : 0.0; }
        Scope sc(h, KC_MG, 24.0 * L.vlen);
        hipLaunchKernelGGL((k_lincomb<2, 1>), dim3(L.nblk, F), dim3(KSFD_BLOCK), 0, h->st, L.kv, LL, x);
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

static int mg_vcycle(ksfd_handle *h, size_t l, double shift, const double *b, double *x);

// coarse-grid correction of level l: restrict L.r, recurse, prolong-add into x
static int mg_coarse_correction(ksfd_handle *h, size_t l, double shift, double *x)
{
    int rc;
    MGLevel &L = h->mg[l], &Lc = h->mg[l + 1];
    if ((rc = mg_halo(h, L, L.r, L.G.F))) return rc;                 // restriction reads fine rows -1 and sloc
    {
        Scope sc(h, KC_MG, 8.0 * L.G.F * (L.G.nloc + Lc.G.nloc));
        mg_launch_restrict(h, L, Lc, L.G.F, L.r, Lc.b);
    }
    if ((rc = mg_vcycle(h, l + 1, shift, Lc.b, Lc.x))) return rc;
    if ((rc = mg_halo(h, Lc, Lc.x, L.G.F))) return rc;               // prolongation reads coarse row sloc_c
    {
        Scope sc(h, KC_MG, 8.0 * L.G.F * (2 * L.G.nloc + Lc.G.nloc));
        mg_launch_prolong(h, L, Lc, L.G.F, Lc.x, x);
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

static int mg_vcycle(ksfd_handle *h, size_t l, double shift, const double *b, double *x)
{
    int rc;
    MGLevel &L = h->mg[l];
    if (l + 1 == h->mg.size()) {
        // coarsest grid: Chebyshev over the whole spectrum, enough sweeps for a ~1e-2 reduction
        int sweeps = (int)ceil(0.5 * sqrt(L.ratio) * log(2.0 / h->mg_coarse_tol));
        sweeps = std::min(std::max(sweeps, 4), h->mg_ncoarse);
        return mg_smooth(h, L, shift, b, x, sweeps, true, L.ratio);
    }
    if ((rc = mg_smooth(h, L, shift, b, x, h->mg_nu, true, h->mg_ratio))) return rc;
    if ((rc = mg_op(h, L, x, 2, shift, L.r, b))) return rc;
    if (l == 0 && h->mg_use_graph && !h->capturing) {
        // everything below level 0 touches only fixed buffers: capture it once per shift into a hipGraph and
        // replay it (a V cycle has ~15 launches per level; on small grids they are pure launch latency)
        if (!h->mg_graph || h->mg_graph_shift != shift || h->mg_graph_x != x) {
            if (h->mg_graph) { hipGraphExecDestroy(h->mg_graph); h->mg_graph = nullptr; }
            hipGraph_t g = nullptr;
            const double b0 = h->bytes_acc;
            HIPCHK(h, hipStreamBeginCapture(h->st, hipStreamCaptureModeThreadLocal));
            h->capturing = true;
            rc = mg_coarse_correction(h, 0, shift, x);
            h->capturing = false;
            hipError_t e = hipStreamEndCapture(h->st, &g);
            if (rc) { if (g) hipGraphDestroy(g); return rc; }
            if (e != hipSuccess || !g) return fail(h, KSFD_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            e = hipGraphInstantiate(&h->mg_graph, g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
            if (e != hipSuccess) { h->mg_graph = nullptr; return fail(h, KSFD_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
            h->mg_graph_bytes = h->bytes_acc - b0;
            h->bytes_acc = b0;
            h->mg_graph_shift = shift;
            h->mg_graph_x = x;
        }
        {
            Scope sc(h, KC_MG, h->mg_graph_bytes);
            HIPCHK(h, hipGraphLaunch(h->mg_graph, h->st));
        }
    } else if ((rc = mg_coarse_correction(h, l, shift, x))) return rc;
    return mg_smooth(h, L, shift, b, x, h->mg_nu, false, h->mg_ratio);
}

// out = M^-1 in  (one V cycle)
static int mg_precond(ksfd_handle *h, double shift, const double *in, double *out)
{
    int rc;
    if (!h->mg_coef_valid && (rc = mg_restrict_coefs(h))) return rc;
    if (h->mg_shift != shift && (rc = mg_setup_shift(h, shift))) return rc;
    return mg_vcycle(h, 0, shift, in, out);
}

// ------------------------------------------------------------------------------------------------
// Polynomial preconditioner.  In the non-stiff regime (h*gamma*lambda_max(J) of order 1..10, the regime of the
// headline benchmark) plain GMRES needs ~7 iterations per stage and spends most of its time in Gram-Schmidt, whose
// traffic grows with the square of the iteration count.  z = p(A) v with p the degree-d Chebyshev approximation of
// 1/lambda on [a, b] (spectrum of A/shift: a ~ 1, b = 1 + lambda_max(-J)/shift) costs d Jacobian actions with a fused
// Horner epilogue (out = alpha*v + beta*A t, no extra pass) and cuts the outer iterations to 2-3: same number of
// Jacobian actions, a fraction of the Gram-Schmidt passes.  Used through flexible GMRES (Z basis kept), so the
// solution update needs no extra preconditioner application.
// ------------------------------------------------------------------------------------------------
static int est_lambda_max(ksfd_handle *h, double shift, int nits)
{
    int rc;
    if (!h->pvec) {
        if (alloc_d(h, &h->pvec, h->vlen)) return KSFD_ENOMEM;
        int nb = (int)std::min<long long>((h->vlen + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        hipLaunchKernelGGL(k_hash_fill, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, (long long)h->vlen, h->pvec);
        if ((rc = op_multidot(h, h->pvec, h->pvec, 0))) return rc;
        const double n0 = sqrt(h->hres[0]);
        const double *xs[1] = { h->pvec }; double a[1] = { 1.0 / n0 };
        if ((rc = op_lincomb(h, 1, xs, a, h->pvec))) return rc;
    }
    double lamA = 0.0;
    for (int it = 0; it < nits; it++) {
        if ((rc = op_jvp_frozen_halo(h, h->pvec, 1, shift, h->t3))) return rc;
        if ((rc = op_multidot(h, h->t3, h->t3, 0))) return rc;
        lamA = sqrt(h->hres[0]);
        if (!(lamA > 0.0) || lamA != lamA) return fail(h, KSFD_ENAN, "power iteration on the Jacobian broke down");
        const double *xs[1] = { h->t3 }; double a[1] = { 1.0 / lamA };
        if ((rc = op_lincomb(h, 1, xs, a, h->pvec))) return rc;
    }
    const double est = lamA - shift;
    h->lamJ = est > 0.0 ? est : 0.0;
    return KSFD_OK;
}

// coefficients of p for this shift; degree 0 = "do not precondition"
static void poly_setup(ksfd_handle *h, double shift)
{
    const double a = 0.97, b = 1.0 + 1.15 * h->lamJ / shift;      // spectrum of A/shift (power iteration converges from below: +15 %)
    h->poly_shift = shift;
    h->poly_deg = 0;
    const double kappa = b / a;
    if (kappa < 1.3) return;                                       // GMRES alone needs <= 3 iterations
    const double rc_ = (sqrt(kappa) - 1.0) / (sqrt(kappa) + 1.0);
    int d = (int)ceil(log(h->poly_target) / log(rc_)) - 1;         // residual polynomial of degree d+1: ~2 rc^(d+1) <= 2*target
    d = std::min(std::max(d, 1), std::max(h->poly_max_deg, 1));
    // r(l) = T_{d+1}(mu(l)) / T_{d+1}(mu(0)), mu(l) = m0 + m1 l;  p(l) = (1 - r(l)) / l
    const int n = d + 1;
    double m0 = (b + a) / (b - a), m1 = -2.0 / (b - a);
    double Tp[10] = { 1.0 }, Tc[10] = { m0, m1 }, Tn[10];
    int degc = 1;
    for (int k = 1; k < n; k++) {
        for (int i = 0; i < 10; i++) Tn[i] = 0.0;
        for (int i = 0; i <= degc; i++) { Tn[i] += 2.0 * m0 * Tc[i]; Tn[i + 1] += 2.0 * m1 * Tc[i]; }
        for (int i = 0; i <= degc - 1; i++) Tn[i] -= Tp[i];
        for (int i = 0; i < 10; i++) { Tp[i] = Tc[i]; Tc[i] = Tn[i]; }
        degc++;
    }
    const double t0 = Tc[0];                                       // T_n(mu(0))
    for (int i = 0; i <= d; i++) h->poly_alpha[i] = -(Tc[i + 1] / t0) / shift;   // p_i = -r_{i+1}; the 1/shift turns p(A/shift) into ~A^-1
    h->poly_deg = d;
}

// z = sum_i alpha_i (A/shift)^i v  by Horner, one fused Jacobian action per degree
static int poly_apply(ksfd_handle *h, double shift, double *v, double *z)
{
    int rc;
    const int d = h->poly_deg;
    const double *al = h->poly_alpha;
    if (h->poly_fp32 && h->coef32 && fused_ok(h)) {
        // mixed precision: the Horner temporaries and the coefficient planes live in fp32 (half the traffic of every
        // application but the arithmetic stays fp64); v is read and z written in fp64.  p(A) becomes a slightly
        // different fixed linear operator, which flexible GMRES does not care about: w_j = A z_j is computed in fp64
        // from the stored z_j, so the Arnoldi relation and the solution keep full accuracy.
        const float *C = h->coef32;
        float *tf[2] = { reinterpret_cast<float *>(h->t1), reinterpret_cast<float *>(h->t2) };
        const double *nod = nullptr;
        if (d == 1) return jvp2d_halo_t<float, double, double, double>(h, C, v, 4, shift, z, nod, al[0], al[1] / shift);
        if ((rc = jvp2d_halo_t<float, double, double, float>(h, C, v, 4, shift, tf[0], nod, al[d - 1], al[d] / shift))) return rc;
        int cur32 = 0;
        for (int i = d - 2; i >= 1; i--) {
            if ((rc = jvp2d_halo_t<float, float, double, float>(h, C, tf[cur32], 3, shift, tf[cur32 ^ 1], (const double *)v, al[i], 1.0 / shift))) return rc;
            cur32 ^= 1;
        }
        return jvp2d_halo_t<float, float, double, double>(h, C, tf[cur32], 3, shift, z, (const double *)v, al[0], 1.0 / shift);
    }
    double *tmp[2] = { h->t1, h->t2 };
    // t_{d-1} = alpha_{d-1} v + (alpha_d/shift) A v
    double *cur = (d == 1) ? z : tmp[0];
    if ((rc = op_jvp_frozen_halo(h, v, 4, shift, cur, nullptr, al[d - 1], al[d] / shift))) return rc;
    int flip = 1;
    for (int i = d - 2; i >= 0; i--) {
        double *nxt = (i == 0) ? z : tmp[flip];
        if ((rc = op_jvp_frozen_halo(h, cur, 3, shift, nxt, v, al[i], 1.0 / shift))) return rc;
        cur = nxt;
        flip ^= 1;
    }
    return KSFD_OK;
}

// ------------------------------------------------------------------------------------------------
// matrix-free GMRES(m) for (shift I - J(u)) x = b, x0 = 0  -- replaces -ksp_type preonly -pc_type lu
// (options84:58-60).  Classical Gram-Schmidt applied twice (CGS2), one fused multi-dot + one fused
// update kernel per pass; the new vector's norm comes from the second pass by Pythagoras.
// ------------------------------------------------------------------------------------------------
struct LinStats { int its; double rel; };

static void rec_reset(ksfd_handle *h)
{
    for (auto &r : h->rec) r.valid = false;
    h->rec_vtop = h->rec_ztop = 0;
}

// Least squares min ||g - H y|| for a small upper-Hessenberg H ((k+1) x k, column-major, ld = k+1); also returns H y.
static void hess_lsq(const double *H, int k, const double *g, double *y, double *Hy)
{
    double R[20], q[5];
    const int ld = k + 1;
    for (int i = 0; i < ld * k; i++) R[i] = H[i];
    for (int i = 0; i <= k; i++) q[i] = g[i];
    for (int j = 0; j < k; j++) {
        const double a = R[j * ld + j], b = R[j * ld + j + 1], den = hypot(a, b);
        const double c = den > 0.0 ? a / den : 1.0, sn = den > 0.0 ? b / den : 0.0;
        for (int l = j; l < k; l++) {
            const double t = c * R[l * ld + j] + sn * R[l * ld + j + 1];
            R[l * ld + j + 1] = -sn * R[l * ld + j] + c * R[l * ld + j + 1];
            R[l * ld + j] = t;
        }
        const double t = c * q[j] + sn * q[j + 1];
        q[j + 1] = -sn * q[j] + c * q[j + 1];
        q[j] = t;
    }
    for (int i = k - 1; i >= 0; i--) {
        double t = q[i];
        for (int l = i + 1; l < k; l++) t -= R[l * ld + i] * y[l];
        y[i] = R[i * ld + i] != 0.0 ? t / R[i * ld + i] : 0.0;
    }
    for (int i = 0; i <= k; i++) {
        double t = 0.0;
        for (int l = 0; l < k; l++) t += H[l * ld + i] * y[l];
        Hy[i] = t;
    }
}

// stage >= 0: Krylov recycling.  The four stage systems of a step share the matrix, and their right-hand sides are
// nearly linear images of one another (b_i = f(u + sum a_ij Y_j) - sum c_ij Y_j/h with f almost linear over a step), so
// the leading Arnoldi vectors of an earlier stage (A Z_s = V_s H_s, kept in place at the front of V / Zb) already span
// most of the new solution: x0 = Z_s y with y = argmin ||V_s^T r - H_s y||, r <- r - V_s H_s y, one space after the
// other, costs ~5 vector passes per kept vector and removes 1-2 of the 3-4 outer iterations of stages 2-4 (each
// (d+1) Jacobian actions + Gram-Schmidt).  The iteration then continues on the true residual with the same stopping
// test, so the result is the same to the solver tolerance.  stage < 0: plain solve from x0 = 0.
static int gmres(ksfd_handle *h, const double *ustate, double shift, const double *b, double *x,
                 const ksfd_step_opts *o, LinStats *ls, int pcmode, int stage = -1)
{
    const bool use_pc = pcmode == 1;       // multigrid, right preconditioning
    const bool use_poly = pcmode == 2;     // Chebyshev polynomial, flexible GMRES (z_j kept in Zb)
    // use_pc: right preconditioning with one multigrid V cycle, w = A (M^-1 v_j), x = M^-1 (V y)
    auto apply_A = [&](const double *vin, double *wout) -> int {
        return h->use_frozen ? op_jvp_frozen(h, vin, 1, shift, wout) : op_jvp(h, ustate, vin, 1, shift, wout);
    };
    const int m_opt = std::min(o->ksp_restart > 0 ? o->ksp_restart : 30, h->restart_alloc);
    const int maxit = o->ksp_max_it > 0 ? o->ksp_max_it : 2000;
    const int64_t vs = h->vlen;
    bool rec_on = stage >= 0 && stage < 4 && h->rec_mode > 0 && h->use_frozen;
    if (!rec_on || stage == 0 || h->restart_alloc - h->rec_vtop < 10) { if (stage != 0) rec_on = false; rec_reset(h); }
    const int vb = rec_on ? h->rec_vtop : 0, zb = rec_on ? h->rec_ztop : 0;
    const int m = std::min(m_opt, h->restart_alloc - vb);
    double *V = h->V + (int64_t)vb * vs;
    double *Zq = use_poly ? h->Zb + (int64_t)zb * vs : nullptr;
    int rc;
    std::vector<double> H((size_t)(m + 1) * m, 0.0), Hraw((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m), hcol(m + 2), d(m + 2), Gm((size_t)(m + 1) * (m + 1), 0.0);
    if ((rc = op_multidot(h, b, V, 0))) return rc;
    const double bn = sqrt(h->hres[0]);
    ls->its = 0; ls->rel = 0.0;
    if (!(bn > 0.0)) {
        if (bn != bn) return fail(h, KSFD_ENAN, "GMRES: right-hand side is not finite");
        HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)vs, h->st));
        return KSFD_OK;
    }
    const double tol = std::max(o->ksp_rtol * bn, o->ksp_atol);
    double beta = bn, rn = bn;
    int total = 0;
    bool first = true;          // the residual of the current x is at hand (rsrc, norm beta): no A x needed
    bool x_set = false;         // x holds an iterate (else it is taken as 0 and overwritten)
    bool restarted = false;
    const double *rsrc = b;
    if (rec_on && stage > 0) {
        static const int sel[4][3] = { { -1, -1, -1 }, { 0, -1, -1 }, { 0, -1, -1 }, { 0, 2, -1 } };
        for (int q = 0; q < stage; q++) {
            bool use = h->rec_mode == 2;
            for (int e = 0; e < 3; e++) use = use || sel[stage][e] == q;
            const ksfd_handle::RecSpace &S = h->rec[q];
            if (!use || !S.valid || S.pc != pcmode) continue;
            const double *Vs = h->V + (int64_t)S.vb * vs;
            const double *Zs = use_poly ? h->Zb + (int64_t)S.zb * vs : Vs;
            double gq[5], yq[4], Hy[5], neg[5];
            if ((rc = op_multidot(h, rsrc, Vs, S.k + 1))) return rc;
            for (int i = 0; i <= S.k; i++) gq[i] = h->hres[i];
            hess_lsq(S.H, S.k, gq, yq, Hy);
            if (use_pc) {
                if ((rc = op_basis_axpy(h, h->t2, Vs, S.k, yq, 0.0)) || (rc = mg_precond(h, shift, h->t2, h->t1))) return rc;
                if (!x_set) { if ((rc = op_copy(h, x, h->t1))) return rc; }
                else { const double *xs[2] = { x, h->t1 }; double a2[2] = { 1.0, 1.0 }; if ((rc = op_lincomb(h, 2, xs, a2, x))) return rc; }
            } else if ((rc = op_basis_axpy(h, x, Zs, S.k, yq, x_set ? 1.0 : 0.0))) return rc;
            x_set = true;
            for (int i = 0; i <= S.k; i++) neg[i] = -Hy[i];
            if (rsrc == b) {
                const double *xs[6] = { b }; double a[6] = { 1.0 };
                for (int i = 0; i <= S.k; i++) { xs[i + 1] = Vs + (int64_t)i * vs; a[i + 1] = neg[i]; }
                if ((rc = op_lincomb(h, S.k + 2, xs, a, h->t3))) return rc;
                rsrc = h->t3;
            } else if ((rc = op_basis_axpy(h, h->t3, Vs, S.k + 1, neg, 1.0))) return rc;
        }
        if (x_set) {
            if ((rc = op_multidot(h, rsrc, rsrc, 0))) return rc;
            beta = rn = sqrt(h->hres[0]);
            if (!(beta == beta)) return fail(h, KSFD_ENAN, "GMRES: projected residual is not finite");
        }
    }
    while (true) {
        if (first && beta <= tol) break;                       // the recycled spaces already hold the solution
        // V0 = r / beta
        if (first) { const double *xs[1] = { rsrc }; double a[1] = { 1.0 / beta }; if ((rc = op_lincomb(h, 1, xs, a, V))) return rc; }
        else {
            if ((rc = halo(h, x)) || (rc = apply_A(x, V))) return rc;     // V0 = A x
            const double *xs[2] = { b, V }; double a[2] = { 1.0, -1.0 };
            if ((rc = op_lincomb(h, 2, xs, a, V))) return rc;                                  // r = b - A x
            if ((rc = op_multidot(h, V, V, 0))) return rc;
            beta = sqrt(h->hres[0]);
            rn = beta;
            if (!(beta == beta)) return fail(h, KSFD_ENAN, "GMRES: residual is not finite");
            if (beta <= tol) break;
            const double *x1[1] = { V }; double a1[1] = { 1.0 / beta };
            if ((rc = op_lincomb(h, 1, x1, a1, V))) return rc;
        }
        std::fill(g.begin(), g.end(), 0.0);
        g[0] = beta;
        int j = 0;
        bool done = false;
        for (; j < m && total < maxit; j++) {
            double *vj = V + (int64_t)j * vs, *w = V + (int64_t)(j + 1) * vs;
            if (use_pc) {
                if ((rc = mg_precond(h, shift, vj, h->t1)) || (rc = op_jvp_frozen_halo(h, h->t1, 1, shift, w))) return rc;
            } else if (use_poly) {
                double *zj = Zq + (int64_t)j * vs;
                if ((rc = poly_apply(h, shift, vj, zj)) || (rc = op_jvp_frozen_halo(h, zj, 1, shift, w))) return rc;
            } else if (h->use_frozen) {
                if ((rc = op_jvp_frozen_halo(h, vj, 1, shift, w))) return rc;
            } else if ((rc = halo(h, vj)) || (rc = apply_A(vj, w))) return rc;
            const int k = j + 1;
            if (o->reserved == 1) {
                // classic CGS2: two Gram-Schmidt passes, each = one fused multi-dot + one fused update.
                // (One pass alone loses orthogonality like eps*(||r0||/||r_j||)^2 and stalls near 1e-8.)
                if ((rc = op_multidot(h, w, V, k))) return rc;
                for (int i = 0; i < k; i++) hcol[i] = h->hres[i];
                if (!(h->hres[k] == h->hres[k])) return fail(h, KSFD_ENAN, "GMRES: Krylov vector is not finite");
                if ((rc = op_gs_update(h, w, V, k, hcol.data(), 1.0))) return rc;
                if ((rc = op_multidot(h, w, V, k))) return rc;
                double s2 = 0.0;
                for (int i = 0; i < k; i++) { d[i] = h->hres[i]; hcol[i] += d[i]; s2 += d[i] * d[i]; }
                double hn2 = h->hres[k] - s2;          // ||w''||^2 by Pythagoras; d is O(eps) so this is accurate
                if (hn2 < 0.0) hn2 = 0.0;
                const double hn = sqrt(hn2);
                if ((rc = op_gs_update(h, w, V, k, d.data(), hn > 0.0 ? 1.0 / hn : 0.0))) return rc;
                hcol[k] = hn;
            } else {
                // CGS2 with the second projection done algebraically (halves the Gram-Schmidt traffic):
                //   d = V^T w and the Gram row g = V^T v_j come from ONE pass over V; with G = V^T V,
                //   the twice-projected coefficients are c = d + (I - G) d, and
                //   ||w - V c||^2 = ww - 2 c.d + c.G c.   One fused update pass applies c and normalises.
                if ((rc = op_multidot_gram(h, w, V, k))) return rc;
                for (int i = 0; i < k; i++) { d[i] = h->hres[i]; Gm[(size_t)i * (m + 1) + j] = Gm[(size_t)j * (m + 1) + i] = h->hres[k + i]; }
                const double ww = h->hres[2 * k];
                if (!(ww == ww)) return fail(h, KSFD_ENAN, "GMRES: Krylov vector is not finite");
                for (int i = 0; i < k; i++) {
                    double s = 0.0;
                    for (int l = 0; l < k; l++) s += ((i == l ? 1.0 : 0.0) - Gm[(size_t)i * (m + 1) + l]) * d[l];
                    hcol[i] = d[i] + s;
                }
                double cd = 0.0, cGc = 0.0;
                for (int i = 0; i < k; i++) {
                    cd += hcol[i] * d[i];
                    double s = 0.0;
                    for (int l = 0; l < k; l++) s += Gm[(size_t)i * (m + 1) + l] * hcol[l];
                    cGc += hcol[i] * s;
                }
                double hn2 = ww - 2.0 * cd + cGc;
                double hn;
                if (hn2 > 1e-8 * ww) {
                    hn = sqrt(hn2);
                    if ((rc = op_gs_update(h, w, V, k, hcol.data(), 1.0 / hn))) return rc;
                } else {
                    // heavy cancellation (||w|| >> ||w - Vc||): apply c, then measure and project once more
                    if ((rc = op_gs_update(h, w, V, k, hcol.data(), 1.0))) return rc;
                    if ((rc = op_multidot(h, w, V, k))) return rc;
                    double s2 = 0.0;
                    for (int i = 0; i < k; i++) { d[i] = h->hres[i]; hcol[i] += d[i]; s2 += d[i] * d[i]; }
                    hn2 = h->hres[k] - s2;
                    if (hn2 < 0.0) hn2 = 0.0;
                    hn = sqrt(hn2);
                    if ((rc = op_gs_update(h, w, V, k, d.data(), hn > 0.0 ? 1.0 / hn : 0.0))) return rc;
                }
                hcol[k] = hn;
            }
            double *Hc = &H[(size_t)(m + 1) * j];
            for (int i = 0; i <= k; i++) Hraw[(size_t)(m + 1) * j + i] = Hc[i] = hcol[i];
            for (int i = 0; i < j; i++) { double t = cs[i] * Hc[i] + sn[i] * Hc[i + 1]; Hc[i + 1] = -sn[i] * Hc[i] + cs[i] * Hc[i + 1]; Hc[i] = t; }
            const double den = hypot(Hc[j], Hc[j + 1]);
            cs[j] = den > 0.0 ? Hc[j] / den : 1.0;
            sn[j] = den > 0.0 ? Hc[j + 1] / den : 0.0;
            Hc[j] = den; Hc[j + 1] = 0.0;
            g[j + 1] = -sn[j] * g[j];
            g[j] = cs[j] * g[j];
            total++;
            rn = fabs(g[j + 1]);
            if (rn <= tol || hcol[k] == 0.0) { j++; done = true; break; }
        }
        for (int i = j - 1; i >= 0; i--) {
            double s = g[i];
            for (int q = i + 1; q < j; q++) s -= H[(size_t)(m + 1) * q + i] * y[q];
            y[i] = s / H[(size_t)(m + 1) * i + i];
        }
        if (use_pc) {
            if ((rc = op_basis_axpy(h, h->t2, V, j, y.data(), 0.0)) || (rc = mg_precond(h, shift, h->t2, h->t1))) return rc;
            if (!x_set) { if ((rc = op_copy(h, x, h->t1))) return rc; }
            else { const double *xs[2] = { x, h->t1 }; double a2[2] = { 1.0, 1.0 }; if ((rc = op_lincomb(h, 2, xs, a2, x))) return rc; }
        } else if ((rc = op_basis_axpy(h, x, use_poly ? Zq : V, j, y.data(), x_set ? 1.0 : 0.0))) return rc;
        x_set = true;
        if (rec_on && first && !restarted && done && j >= 1) {
            // keep the leading vectors of this stage's Arnoldi relation where they are; the next stage builds behind them
            ksfd_handle::RecSpace &S = h->rec[stage];
            S.k = std::min(j, std::min(h->rec_keep, 4));
            S.vb = vb; S.zb = zb; S.pc = pcmode;
            for (int c = 0; c < S.k; c++)
                for (int i = 0; i <= S.k; i++) S.H[c * (S.k + 1) + i] = Hraw[(size_t)(m + 1) * c + i];
            S.valid = true;
            h->rec_vtop = vb + S.k + 1;
            h->rec_ztop = zb + (use_poly ? S.k : 0);
        }
        if (!first) restarted = true;
        first = false;
        if (done || total >= maxit) break;
        restarted = true;
    }
    if (!x_set) HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)vs, h->st));
    ls->its = total;
    ls->rel = rn / bn;
    if (rn > tol) return fail(h, KSFD_ELINEAR, "GMRES did not converge: %d iterations, relative residual %.3e (tol %.3e)", total, rn / bn, tol / bn);
    return KSFD_OK;
}

// ------------------------------------------------------------------------------------------------
// Pipelined GMRES: same mathematics as gmres() (CGS2 with the algebraic second projection), but the small
// algebra of every iteration runs in a one-thread kernel on the device (k_gmres_coef) and the fused update reads its
// coefficients from device memory, so an iteration = [J action, multi-dot, reduce(+allreduce), coef, update] with NO
// host round trip.  The host polls the residual estimate one iteration behind (and exactly on time when the
// extrapolated estimate says "this one converges"), so the GPU never idles and at most one iteration is wasted.
// Pays when an iteration is latency-bound: small grids, many slab ranks.  Unpreconditioned, frozen Jacobian only.
// ------------------------------------------------------------------------------------------------
static int gmres_async(ksfd_handle *h, double shift, const double *b, double *x, const ksfd_step_opts *o, LinStats *ls)
{
    rec_reset(h);
    const int m = std::min(o->ksp_restart > 0 ? o->ksp_restart : 30, h->restart_alloc);
    const int maxit = o->ksp_max_it > 0 ? o->ksp_max_it : 2000;
    const int64_t vs = h->vlen;
    const int ld = h->restart_alloc + 1;
    double *V = h->V;
    double *dG = h->gm_dev, *dH = dG + (size_t)ld * ld, *dcs = dH + (size_t)ld * h->restart_alloc, *dsn = dcs + h->restart_alloc,
           *dg = dsn + h->restart_alloc, *dcoef = dg + ld, *dscale = dcoef + KSFD_MAXDOT, *dmon = dscale + 1;
    double *hmon = h->gm_host, *hH = hmon + 2 * ld, *hg = hH + (size_t)ld * h->restart_alloc;
    int rc;
    if ((rc = op_multidot(h, b, V, 0))) return rc;
    const double bn = sqrt(h->hres[0]);
    ls->its = 0; ls->rel = 0.0;
    if (!(bn > 0.0)) {
        if (bn != bn) return fail(h, KSFD_ENAN, "GMRES: right-hand side is not finite");
        HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)vs, h->st));
        return KSFD_OK;
    }
    const double tol = std::max(o->ksp_rtol * bn, o->ksp_atol);
    double beta = bn, rn = bn;
    int total = 0;
    bool first = true;
    std::vector<double> y(m);
    while (true) {
        if (first) { const double *xs[1] = { b }; double a[1] = { 1.0 / beta }; if ((rc = op_lincomb(h, 1, xs, a, V))) return rc; }
        else {
            if ((rc = op_jvp_frozen_halo(h, x, 1, shift, V))) return rc;
            const double *xs[2] = { b, V }; double a[2] = { 1.0, -1.0 };
            if ((rc = op_lincomb(h, 2, xs, a, V))) return rc;
            if ((rc = op_multidot(h, V, V, 0))) return rc;
            beta = sqrt(h->hres[0]);
            rn = beta;
            if (!(beta == beta)) return fail(h, KSFD_ENAN, "GMRES: residual is not finite");
            if (beta <= tol) break;
            const double *x1[1] = { V }; double a1[1] = { 1.0 / beta };
            if ((rc = op_lincomb(h, 1, x1, a1, V))) return rc;
        }
        int jc = -1, jlast = -1, checked = -1;
        double r1 = beta, r2 = -1.0;
        auto poll = [&](int upto) -> int {          // read monitors (checked, upto]; sets jc when converged
            for (int q = checked + 1; q <= upto; q++) {
                if (hipEventSynchronize(h->gm_ev[q]) != hipSuccess) return fail(h, KSFD_EHIP, "event sync failed");
                const double r = hmon[2 * q], hn = hmon[2 * q + 1];
                if (!(r == r)) return fail(h, KSFD_ENAN, "GMRES: Krylov vector is not finite");
                checked = q;
                r2 = r1; r1 = r;
                if (r <= tol || hn == 0.0) { jc = q; return KSFD_OK; }
            }
            return KSFD_OK;
        };
        for (int j = 0; j < m && total + j < maxit; j++) {
            double *vj = V + (int64_t)j * vs, *w = V + (int64_t)(j + 1) * vs;
            const int k = j + 1;
            if ((rc = op_jvp_frozen_halo(h, vj, 1, shift, w))) return rc;
            const int nb = vec2(h) ? (h->nblk_vec + 1) / 2 : h->nblk_vec;
            {
                Scope sc(h, KC_MULTIDOT, vbytes(h, k + 1));
                if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<4, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
                else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<8, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
                else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<16, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
                else VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<32, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
            }
            {
                Scope sc(h, KC_REDUCE, 8.0 * (2 * k + 1) * (double)nb);
                hipLaunchKernelGGL(k_reduce_rows, dim3(2 * k + 1), dim3(KSFD_BLOCK), 0, h->st, (const double *)h->part, nb, 0, h->dres);
            }
            if (h->size > 1 && h->tr->allreduce(h->dres, 2 * k + 1, 0, h->st)) return fail(h, KSFD_ECOMM, "allreduce failed: %s", h->tr->error().c_str());
            hipLaunchKernelGGL(k_gmres_coef, dim3(1), dim3(64), 0, h->st, j, h->restart_alloc, beta, (const double *)h->dres, dG, dH, dcs, dsn, dg, dcoef, dscale, dmon);
            {
                Scope sc(h, KC_GSUPDATE, vbytes(h, k + 2));
                if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<4, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
                else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<8, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
                else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<16, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
                else VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<32, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
            }
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(hmon + 2 * j, dmon + 2 * j, 2 * sizeof(double), hipMemcpyDeviceToHost, h->st));
            HIPCHK(h, hipEventRecord(h->gm_ev[j], h->st));
            jlast = j;
            if ((rc = poll(j - 1))) return rc;                 // one iteration behind: the GPU already has iteration j queued
            if (jc >= 0) break;
            if (r2 > 0.0 && r1 * (r1 / r2) <= 1.5 * tol) {     // extrapolation says iteration j converges: look now, queue nothing more
                if ((rc = poll(j))) return rc;
                if (jc >= 0) break;
            }
        }
        if (jc < 0 && (rc = poll(jlast))) return rc;
        const int kused = jc >= 0 ? jc + 1 : jlast + 1;
        total += jlast + 1;
        HIPCHK(h, hipMemcpyAsync(hH, dH, sizeof(double) * (size_t)ld * h->restart_alloc, hipMemcpyDeviceToHost, h->st));
        HIPCHK(h, hipMemcpyAsync(hg, dg, sizeof(double) * ld, hipMemcpyDeviceToHost, h->st));
        HIPCHK(h, hipStreamSynchronize(h->st));
        for (int i = kused - 1; i >= 0; i--) {
            double s = hg[i];
            for (int q = i + 1; q < kused; q++) s -= hH[(size_t)ld * q + i] * y[q];
            y[i] = s / hH[(size_t)ld * i + i];
        }
        if ((rc = op_basis_axpy(h, x, V, kused, y.data(), first ? 0.0 : 1.0))) return rc;
        first = false;
        rn = hmon[2 * (kused - 1)];
        if (jc >= 0 || total >= maxit) break;
    }
    ls->its = total;
    ls->rel = rn / bn;
    if (rn > tol) return fail(h, KSFD_ELINEAR, "GMRES did not converge: %d iterations, relative residual %.3e (tol %.3e)", total, rn / bn, tol / bn);
    return KSFD_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" void ksfd_default_step_opts(ksfd_step_opts *o)
{
    memset(o, 0, sizeof *o);
    o->rtol = 1e-5; o->atol = 1e-5;           // KSFD/ksfdts.py:66-67 defaults
    o->adapt = 1; o->max_reject = 10;
    o->clip_lo = 0.1; o->clip_hi = 5.0;       // -ts_adapt_clip 0.1,5
    o->dt_min = 1e-20; o->dt_max = 1e4;       // -ts_adapt_dt_min/-ts_adapt_dt_max
    o->safety = 0.9; o->reject_safety = 0.5;  // PETSc TSAdaptBasic defaults
    // relative residual 1e-6: the fields then agree with a 1e-12 solve to ~1e-10 rel-L2 over several steps
    // (tools/acc_vs_ksp.py); the north-star tolerance is 1e-8.  PETSc's own KSP default is 1e-5.
    o->ksp_rtol = 1e-6; o->ksp_atol = 1e-50;
    o->ksp_restart = 30; o->ksp_max_it = 2000;
    o->pc_type = 2;    // 0 none, 1 multigrid always, 2 multigrid when the step is stiff (2-D, single rank)
}

// One TSStep_RosW attempt loop (PETSc rosw.c restated; tableau/derivation in oracle/ksfd_oracle.c).
extern "C" int ksfd_step(ksfd_handle *h, double *t, double *hstep, const ksfd_step_opts *opts, ksfd_step_stats *stats)
{
    if (!h || !t || !hstep || !opts) return KSFD_EINVAL;
    hipSetDevice(h->device);
    ksfd_step_stats st;
    memset(&st, 0, sizeof st);
    const double bytes0 = h->bytes_acc;
    const int64_t rhs0 = h->prof.launches[KC_RHS], jvp0 = h->prof.launches[KC_JVP];
    int rc = KSFD_OK;
    const int64_t vs = h->vlen;
    double hh = *hstep;
    bool prev_accept = true;
    bool lam_done = false;
    int rejects = 0;
    const int max_rej = opts->max_reject;
    // KSFDTS.solve grooms the global vector before every TS.step (KSFD/ksfdts.py:210)
    if ((rc = ksfd_groom(h))) goto out;
    if ((rc = op_copy(h, h->usave, h->u))) goto out;
    if ((rc = halo(h, h->u))) goto out;
    if (h->use_frozen && (rc = op_jcoef(h, h->u))) goto out;
    h->mg_coef_valid = false; h->mg_shift = -1.0;
    h->poly_shift = -1.0;
    while (true) {
        const double shift = 1.0 / (GAMMA_RA * hh);
        // stiffness estimate X = h*gamma*lambda_max of the diffusion part; the multigrid preconditioner pays off above ~60
        double dmax = h->P.s2, lap = 0.0;
        for (int l = 0; l < h->P.nlig; l++) dmax = std::max(dmax, h->P.lig_D[l]);
        for (int a = 0; a < h->G.dim; a++) lap += (16.0 / 3.0) * h->P.inv_h2[a];
        const double stiff = dmax * lap / shift;
        // (measured crossover against the degree-6 polynomial: X ~ 80 on 4096^2, ~ 280 on 1024^2 where the V cycle is latency-bound)
        const double mg_from = h->mg_threshold * ((double)h->G.F * (double)h->G.nloc < 8.0e6 ? 3.0 : 1.0);
        const bool use_pc = h->mg_ok && h->use_frozen && (opts->pc_type == 1 || (opts->pc_type == 2 && stiff > mg_from));
        // pipelined solver: latency-bound iterations only (small local problem), not in the tiny-h regime where the
        // Pythagorean norm update cancels heavily (|w|^2/h_n^2 ~ 1/stiff^2) and gmres() takes its explicit second pass
        // polynomial preconditioner in the mildly stiff regime (pc_type 2 = automatic, 3 = polynomial whenever useful)
        bool use_poly = false;
        if (!use_pc && h->use_frozen && (opts->pc_type == 2 || opts->pc_type == 3) && stiff >= 0.3) {
            if (!h->Zb && alloc_d(h, &h->Zb, (int64_t)h->restart_alloc * h->vlen)) { rc = KSFD_ENOMEM; goto out; }
            if (!lam_done) {
                if (h->lamJ < 0.0 || ++h->lam_age >= h->lam_period) {
                    const double before = h->lamJ;
                    if ((rc = est_lambda_max(h, shift, before < 0.0 ? 8 : 2))) goto out;    // warm-started after the first step
                    // J changes slowly from step to step: while the estimate moves by < 2 %, look less often
                    const bool stable = before > 0.0 && fabs(h->lamJ - before) <= 0.02 * before;
                    h->lam_period = stable ? std::min(2 * h->lam_period, 8) : 1;
                    h->lam_age = 0;
                }
                lam_done = true;
            }
            if (h->poly_shift != shift) poly_setup(h, shift);
            use_poly = h->poly_deg >= 1 && h->poly_max_deg >= 1;
        }
        const bool small = (double)h->G.F * (double)h->G.nloc <= 6.0e6;
        const bool use_async = !use_pc && !use_poly && h->use_frozen && opts->reserved == 0 && stiff >= 1e-3 &&
                               (h->size == 1 || h->tr->device_allreduce()) &&
                               (h->async_mode == 1 || (h->async_mode == 2 && small));
        const bool fuse_stage = fused_ok(h) && h->P.nlig <= 4 && h->fuse_stage;
        for (int i = 0; i < 4 && !rc; i++) {
            const double *zin = h->u;
            if (fuse_stage) {
                // stage argument and Zdot term folded into the RHS kernel (no Z vector, no separate passes)
                KComb cmb = KComb{};
                for (int j = 0; j < i; j++) {
                    if (h->At[i][j] != 0.0) { cmb.yin[cmb.nin] = h->Y + (int64_t)j * vs; cmb.ain[cmb.nin++] = h->At[i][j]; }
                    if (h->Ginv[i][j] != 0.0) { cmb.yout[cmb.nout] = h->Y + (int64_t)j * vs; cmb.aout[cmb.nout++] = -h->Ginv[i][j] / hh; }
                }
                if (i > 0 && (rc = halo(h, h->Y + (int64_t)(i - 1) * vs))) break;     // ghosts of the newest stage vector (earlier ones done)
                if ((rc = op_rhs(h, h->u, i, h->bvec, &cmb))) break;
            } else {
            if (i > 0) {
                const double *xs[5]; double a[5]; int nt = 0;
                xs[nt] = h->u; a[nt++] = 1.0;
                for (int j = 0; j < i; j++) if (h->At[i][j] != 0.0) { xs[nt] = h->Y + (int64_t)j * vs; a[nt++] = h->At[i][j]; }
                if (nt > 1) {
                    if ((rc = op_lincomb(h, nt, xs, a, h->Z))) break;
                    if ((rc = halo(h, h->Z))) break;
                    zin = h->Z;
                }
            }
            if ((rc = op_rhs(h, zin, i, h->bvec))) break;
            if (i > 0) {
                const double *xs[5]; double a[5]; int nt = 0;
                xs[nt] = h->bvec; a[nt++] = 1.0;
                for (int j = 0; j < i; j++) if (h->Ginv[i][j] != 0.0) { xs[nt] = h->Y + (int64_t)j * vs; a[nt++] = -h->Ginv[i][j] / hh; }
                if (nt > 1 && (rc = op_lincomb(h, nt, xs, a, h->bvec))) break;
            }
            }
            LinStats ls;
            rc = use_async ? gmres_async(h, shift, h->bvec, h->Y + (int64_t)i * vs, opts, &ls)
                           : gmres(h, h->u, shift, h->bvec, h->Y + (int64_t)i * vs, opts, &ls, use_pc ? 1 : (use_poly ? 2 : 0), i);
            st.linear_its += ls.its;
            st.ksp_resid = ls.rel;
            if (rc == KSFD_ELINEAR && !use_pc && h->mg_ok && h->use_frozen && opts->pc_type) {
                // unpreconditioned GMRES ran out of iterations: the multigrid-preconditioned solve of the same system
                // is the remedy (the stiffness estimate above only knows the diffusion part of J)
                rc = gmres(h, h->u, shift, h->bvec, h->Y + (int64_t)i * vs, opts, &ls, 1);
                st.linear_its += ls.its;
                st.ksp_resid = ls.rel;
            }
        }
        if (rc == KSFD_ELINEAR && opts->adapt && max_rej >= 0 && rejects < max_rej && hh * 0.25 >= opts->dt_min) {
            // PETSc's -ts_adapt_scale_solve_failed (0.25): a failed solve rejects the step and quarters it.  The
            // reference disables that by setMaxSNESFailures(1) (KSFD/ksfdts.py:135) because its LU cannot fail this
            // way; an iterative solve can, and aborting a long run for it would not be a service.
            rejects++;
            st.rejections = rejects;
            prev_accept = false;
            if ((rc = op_copy(h, h->u, h->usave))) goto out;
            hh *= 0.25;
            *hstep = hh;
            if ((rc = halo(h, h->u))) goto out;
            continue;
        }
        if (rc) { op_copy(h, h->u, h->usave); hipStreamSynchronize(h->st); goto out; }
        // completion + embedded error norm
        {
            Scope sc(h, KC_FINISH, vbytes(h, 7));
            hipLaunchKernelGGL(k_rosw_finish, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, h->Y, vs,
                               h->bt[0], h->bt[1], h->bt[2], h->bt[3], h->b2t[0] - h->bt[0], h->b2t[1] - h->bt[1],
                               h->b2t[2] - h->bt[2], h->b2t[3] - h->bt[3], opts->atol, opts->rtol, h->errv, h->part);
        }
        h->have_err = true;
        if ((rc = reduce_rows(h, 1, h->nblk_vec, 0))) goto out;
        {
            double ntot = (double)h->G.F * (double)h->cfg.n[0] * (double)h->cfg.n[1] * (double)h->cfg.n[2];
            st.wrms = sqrt(h->hres[0] / ntot);
        }
        if (!(st.wrms == st.wrms) || isinf(st.wrms)) {
            op_copy(h, h->u, h->usave); hipStreamSynchronize(h->st);
            rc = fail(h, KSFD_ENAN, "non-finite error norm at t=%g h=%g", *t, hh);
            goto out;
        }
        bool accept = true;
        double hnext = hh;
        if (opts->adapt) {
            // TSAdaptChoose_Basic
            double safety = opts->safety;
            if (st.wrms > 1.0) {
                if (!prev_accept) safety *= opts->reject_safety;
                accept = hh < (1.0 + 1.4901161193847656e-08) * opts->dt_min;   // at minimum step: accept anyway
            }
            double hfac = st.wrms > 0.0 ? safety * pow(st.wrms, -1.0 / 3.0) : INFINITY;
            hfac = std::min(std::max(hfac, opts->clip_lo), opts->clip_hi);
            hnext = std::min(std::max(hh * hfac, opts->dt_min), opts->dt_max);
        }
        prev_accept = accept;
        if (accept) {
            st.accepted = 1; st.h_used = hh;
            *t += hh;
            *hstep = hnext;
            break;
        }
        rejects++;
        st.rejections = rejects;
        if ((rc = op_copy(h, h->u, h->usave))) goto out;
        hh = hnext;
        *hstep = hnext;
        if (max_rej < 0) break;                                  // single attempt: report the rejection
        if (rejects > max_rej) { rc = fail(h, KSFD_EREJECT, "step rejected %d times at t=%g", rejects, *t); break; }
        if ((rc = halo(h, h->u))) goto out;
    }
out:
    prof_resolve(h);
    st.bytes = h->bytes_acc - bytes0;
    st.rhs_evals = (int32_t)(h->prof.launches[KC_RHS] - rhs0);
    st.jvp_evals = (int32_t)(h->prof.launches[KC_JVP] - jvp0);
    if (stats) *stats = st;
    return rc;
}

extern "C" int ksfd_get_last_error_vector(ksfd_handle *h, double *eh, int32_t layout)
{
    if (!h || !eh) return KSFD_EINVAL;
    if (!h->have_err) return fail(h, KSFD_EINVAL, "no step has been attempted yet");
    hipSetDevice(h->device);
    return download(h, h->errv, layout, eh);
}

extern "C" int ksfd_set_profiling(ksfd_handle *h, int32_t on)
{
    if (!h) return KSFD_EINVAL;
    prof_resolve(h);
    if (on < 0 || on >= 2 + KSFD_NKCLASS) return KSFD_EINVAL;
    h->profiling = on != 0;
    h->prof_only = on >= 2 ? on - 2 : -1;
    return KSFD_OK;
}
extern "C" int ksfd_get_profile(ksfd_handle *h, ksfd_profile *p, int32_t reset)
{
    if (!h || !p) return KSFD_EINVAL;
    prof_resolve(h);
    *p = h->prof;
    if (reset) memset(&h->prof, 0, sizeof h->prof);
    return KSFD_OK;
}
extern "C" int ksfd_set_mg_params(ksfd_handle *h, int32_t nu, int32_t ncoarse_max, int32_t power_its, double ratio, double coarse_tol)
{
    if (!h) return KSFD_EINVAL;
    if (nu > 0) h->mg_nu = nu;
    if (ncoarse_max > 0) h->mg_ncoarse = ncoarse_max;
    if (power_its > 0) h->mg_power_its = power_its;
    if (ratio > 1.0) h->mg_ratio = ratio;
    if (coarse_tol > 0.0) h->mg_coarse_tol = coarse_tol;
    h->mg_use_graph = power_its != -7 && h->size == 1;     // power_its = -7: eager launches (debug / A-B timing); slab ranks: collectives inside the cycle
    h->mg_shift = -1.0;
    return KSFD_OK;
}
extern "C" int ksfd_set_poly_params(ksfd_handle *h, int32_t max_degree, double target, double mg_threshold)
{
    if (!h || max_degree < 0 || max_degree > 7) return KSFD_EINVAL;
    if (mg_threshold > 0.0) h->mg_threshold = mg_threshold;
    h->poly_max_deg = max_degree;           // 0 disables the polynomial preconditioner
    if (target > 0.0 && target < 1.0) h->poly_target = target;
    h->poly_shift = -1.0;
    return KSFD_OK;
}
extern "C" int ksfd_synchronize(ksfd_handle *h)
{
    if (!h) return KSFD_EINVAL;
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}
extern "C" int ksfd_set_tuning(ksfd_handle *h, int32_t use_fused, int32_t yseg, int32_t yseg_jvp)
{
    if (!h) return KSFD_EINVAL;
    if (use_fused >= 0) {
        h->use_fused = use_fused & 1; h->use_frozen = !(use_fused & 2); h->overlap = !(use_fused & 4);
        h->async_mode = (use_fused & 8) ? 1 : 0;
        h->rec_mode = (use_fused & 16) ? 0 : ((use_fused & 32) ? 2 : 1);
        if ((use_fused >> 6) & 7) h->rec_keep = std::min((use_fused >> 6) & 7, 4);
        h->poly_fp32 = !(use_fused & 512);
        h->fuse_stage = !(use_fused & 1024);
        h->zero_copy = !(use_fused & 2048) && h->hres_dev;
    }
    if (yseg > 0) h->yseg = yseg;
    if (yseg_jvp > 0) h->yseg_jvp = yseg_jvp;
    return KSFD_OK;
}

// Raw timing of one kernel class on the current state, HIP events on the compute stream.
// cls: KC_RHS, KC_JVP, KC_MULTIDOT (k = 8 basis vectors), KC_GSUPDATE (k = 8), KC_LINCOMB (3 inputs).
extern "C" int ksfd_bench_kernel(ksfd_handle *h, int32_t cls, int32_t reps, double *avg_ms, double *bytes_per_launch)
{
    if (!h || reps < 1 || !avg_ms) return KSFD_EINVAL;
    hipSetDevice(h->device);
    hipEvent_t a, b;
    HIPCHK(h, hipEventCreate(&a));
    HIPCHK(h, hipEventCreate(&b));
    const bool was = h->profiling;
    h->profiling = false;
    int rc = KSFD_OK;
    double by = 0.0;
    auto one = [&]() -> int {
        const double b0 = h->bytes_acc;
        int r = KSFD_OK;
        double coef[KSFD_MAXDOT] = { 0 };
        switch (cls) {
        case KC_RHS: r = op_rhs(h, h->u, -1, h->t3); break;
        case KC_JVP: r = h->use_frozen ? op_jvp_frozen(h, h->Y, 1, 1.0, h->t3) : op_jvp(h, h->u, h->Y, 1, 1.0, h->t3); break;
        case KC_MULTIDOT: {
            Scope sc(h, KC_MULTIDOT, vbytes(h, 9));
            VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot<8, VW>), dim3(vec2(h) ? (h->nblk_vec + 1) / 2 : h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)h->t3, (const double *)h->V, h->vlen, 8, h->part));
        } break;
        case KC_GSUPDATE: r = op_gs_update(h, h->t3, h->V, 8, coef, 1.0); break;
        case KC_LINCOMB: { const double *xs[3] = { h->u, h->Y, h->Y + h->vlen }; double aa[3] = { 1.0, 0.5, 0.25 }; r = op_lincomb(h, 3, xs, aa, h->t3); } break;
        default: r = fail(h, KSFD_EINVAL, "bench_kernel: class %d not benchable", cls);
        }
        by = h->bytes_acc - b0;
        return r;
    };
    if ((rc = halo(h, h->u))) goto done;
    if (h->use_frozen && cls == KC_JVP && (rc = op_jcoef(h, h->u))) goto done;
    for (int i = 0; i < 3 && !rc; i++) rc = one();
    if (rc) goto done;
    hipEventRecord(a, h->st);
    for (int i = 0; i < reps && !rc; i++) rc = one();
    hipEventRecord(b, h->st);
    if (hipEventSynchronize(b) != hipSuccess) rc = fail(h, KSFD_EHIP, "event sync failed");
    if (!rc) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        *avg_ms = ms / reps;
        if (bytes_per_launch) *bytes_per_launch = by;
    }
done:
    h->profiling = was;
    hipEventDestroy(a);
    hipEventDestroy(b);
    return rc;
}
