// libksfd_hip.so -- the handle: device buffers, solver state, error reporting, HIP-event profiling scopes, tableau and physics tables
// (part of the single translation unit ksfd_hip.hip; included from there in this order:
//  handle.hip.h, ops.hip.h, mg_host.hip.h, krylov.hip.h)
#pragma once
// ------------------------------------------------------------------------------------------------
// kernel classes for the profile
enum { KC_RHS = 0, KC_JVP, KC_MULTIDOT, KC_GSUPDATE, KC_LINCOMB, KC_BASISAXPY, KC_FINISH, KC_REDUCE,
       KC_GFIELD, KC_VELOCITY, KC_MISC, KC_HALO, KC_MG, KC_SPECTRAL };
static const char *kc_names[KSFD_NKCLASS] = { "rhs", "jvp", "multidot", "gs_update", "lincomb", "basis_axpy",
                                              "rosw_finish", "reduce", "gfield", "velocity", "misc", "halo", "mg", "spectral" };
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
    float *dinv = nullptr;     // F*F planes (fp32)
    double *x = nullptr, *b = nullptr, *r = nullptr, *d = nullptr, *Ad = nullptr, *dG = nullptr;
    float *x32 = nullptr, *b32 = nullptr, *r32 = nullptr, *d32 = nullptr;   // level vectors of the fp32 V cycle (f32 levels only)
    float *coef32 = nullptr;   // fp32 copy of the coefficient planes of a coarse f32 level (level 0 uses the handle's)
    bool f32 = false;          // this level can run the fp32 cycle (2-D strip kernel, not the coarsest level)
    double *pv = nullptr;      // power-iteration vector of Dinv*A, kept from one set-up to the next (warm start)
    double pv_norm = 0.0;      // its norm; 0 = no vector yet
    double lam_max = 2.3;
    double ratio = 60.0;       // lambda_max/lambda_min estimate (used on the coarsest grid)
};

// spectral preconditioner (spectral.hip.h / spectral_host.hip.h)
struct SpecState {
    bool ok = false, means_valid = false, tile_major = false;
    int cols_split = 0;                      // k_spec_cols_split in two phases (spectral.hip.h): 1 = the 2*npair columns of a pair block exceed the LDS (one field pair per block),
                                             // 2 = the two columns of one pair do too (columns of more than 8192 points: one column per block)
    KFFTPlan px, py, pz;
    int dim = 2, lg_cz = 0, pb = 1, nent = 0, lg_rb3 = 0;          // 3-D: z per block of the y kernels, column pairs per block / entries of the z kernel
    size_t lds_y3 = 0, lds_z3 = 0;
    int *posz = nullptr, *kzofpos = nullptr;
    float *lz = nullptr;
    int rb = 0, npair = 0, nyp = 0;
    size_t lds_rows = 0, lds_cols = 0;
    kcf *W = nullptr, *W2 = nullptr, *twx = nullptr, *twy = nullptr, *twz = nullptr;     // W: [pair][pos_x][y_local]; W2: after the all-to-all, [rank][pair][own pos][y_local]
    int *posy = nullptr, *kyofpos = nullptr;
    int2 *ytab = nullptr;                    // per y position: (position of -ky, bits of ly[ky]) -- symbol stage of k_spec_cols
    int lgw = -1;                            // layout of the forward work array (kspec_wt_index)
    int4 *pairtab = nullptr;          // per block of the column kernel
    int nxl = 0, nblk_cols = 0, lg_pl = 0;   // owned positions, column-kernel blocks, log2(rows per rank)
    std::vector<A2APiece> a2a_fwd_s, a2a_fwd_r, a2a_bwd_s, a2a_bwd_r;
    float *lx = nullptr, *ly = nullptr;
    double a_rr = 0.0, a_rU[KSFD_MAXL] = { 0 };
    // adaptation: steps (counted by ksfd_step calls) before which the automatic choice leaves it alone after it converged badly
    long long bad_until = 0;
    bool user_off = false;                   // ksfd_set_spectral_params(enable = 0): off until enabled again; a checkpoint restore does not bring it back
    int backoff = 8;
    // largest contraction ||r_k+1|| / ||r_k|| of a defect-correction sweep measured in the current / the previous step
    // (spec_solve: predicted last sweep)
    double rho_step = 0.0, rho_prev = 0.0;
};

struct ksfd_handle {
    ksfd_config cfg;
    int32_t lig_group[KSFD_MAXL];
    double lig_w[KSFD_MAXL], lig_s[KSFD_MAXL], lig_gamma[KSFD_MAXL], lig_D[KSFD_MAXL];
    double grp_alpha[KSFD_MAXL], grp_beta[KSFD_MAXL];
    KGeom G;
    KPhys P;
    KPhys Pst[4];                    // ps.values(t_stage) for the four stage RHS evaluations (ksfd_set_stage_params)
    bool Pst_valid[4] = { false, false, false, false };
    KVec kv;
    int rank = 0, size = 1, device = 0;
    bool ring = false;               // ghost units + transport in use: size > 1, or ONE rank that is its own ring neighbour (ksfd_dist.size = 1 with a
                                     // transport: the RCCL path executed end to end on a single GPU)
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
    double *ckpt = nullptr;                 // ksfd_checkpoint slot (allocated on first save)
    double *bstore = nullptr;               // right-hand sides of stages 0..2 of the current step (initial guesses of the spectral solves)
    bool spec_guess = true;
    // earlier stages a stage guess is built from (the most recent ones).  Measured on the pinned window at 4096^2: 3 vectors 9.80 ms per step,
    // 2 vectors 9.67 (same sweep counts, two full-vector reads and a multi-dot operand less for stage 4), 1 vector 9.75 (one more sweep)
    int guess_max = getenv("KSFD_GUESS_MAX") ? std::max(1, std::min(atoi(getenv("KSFD_GUESS_MAX")), 3)) : 2;
    bool spec_predict = true;               // predicted last sweep of the defect correction (spec_solve)
    long long n_predicted = 0;              // ... how many solves ended that way
    long long n_residual = 0;               // true residuals evaluated by the spectral defect correction
    long long n_host_sync = 0;              // host waits on the device so far (reduction results, stream synchronisations)
    struct SolverMemo { double lamJ; int lam_age, lam_period; double mg_shift_floor; int sf_dir, sf_hold; bool sf_tried_down; double sf_prev_its, sf_prev_floor;
                        long long nsteps, spec_bad_until; int spec_backoff; double spec_rho_step, spec_rho_prev; };
    SpecState spec;
    long long nsteps = 0;                   // ksfd_step calls so far
    double spec_from = 0.1;                 // stiffness above which pc_type 2 prefers the spectral solver (below: plain GMRES; measured at 4096^2, X = 0.19: 6 sweeps = 7.3 ms against 5 GMRES iterations = 7.9 ms per step)
    SolverMemo ckpt_memo;
    bool ckpt_valid = false;
    double *Gb = nullptr, *dGb = nullptr;   // generic-path scratch planes
    double *coef = nullptr;                 // frozen-Jacobian coefficient planes [rho, G, G_rho, G_U..]
    bool use_frozen = true;
    bool coef_fresh = false;                // coef (and coef32) belong to the resident state u as it is now (ensure_coef)
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
    int restart_alloc = 0;                  // basis vectors V / Zb hold (ksfd_create: 30 ... 120 by the free HBM)
    bool restart_grow = true;               // a GMRES cycle that ends without convergence is followed by one of twice the length (KSFD_TUNE bit 17 switches it off)
    int nblk_vec = 0;                       // grid.x of the BLAS-1 kernels
    bool have_err = false;

    // tuning
    int use_fused = 1;
    int yseg = 48;        // rows per wave segment, RHS kernel (tools/kbench.py at 4096^2 with the hand-rolled log/exp: 8: 0.253, 16: 0.222, 32: 0.198, 64: 0.190 ms)
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
    double mg_threshold = 75.0;      // stiffness above which pc_type 2 switches from the polynomial to multigrid
    double poly_target = 0.02;      // wanted reduction per outer iteration (picks the degree)
    float *coef32 = nullptr;        // fp32 copy of the frozen coefficient planes (2-D strip path only)
    bool rhs_carry = true;          // stage vectors that enter both sides of a stage RHS are read once (k_rhs2d_fused<NL, true>); KSFD_TUNE bit 15 switches it off
    bool fuse_stage = true;         // stage-vector algebra inside the RHS kernel (2-D strip path; 3-D: inside the G pass + strip kernel)
    bool j3l_attr_set = false, j3l_usable = false;   // k_jvp3d_lds: dynamic LDS size registered / usable (ops.hip.h: j3l_ok)
    bool rhs3d_strip = true;        // 3-D RHS: G pass + z-marching strip kernel (false: generic one-thread-per-point stencil pass)
    bool poly_fp32 = true;          // Horner temporaries and coefficients of p(A) in fp32 storage (the outer A z_j stays fp64)

    // asynchronous snapshots for writers (ksfd_snapshot_begin / _wait): layout transform on the compute stream into a
    // device staging slot, D2H on a third stream into pinned memory while the stepper carries on
    hipStream_t st_io = nullptr;
    double *snap_dev[2] = { nullptr, nullptr }, *snap_host[2] = { nullptr, nullptr };
    hipEvent_t snap_ready[2] = { nullptr, nullptr }, snap_done[2] = { nullptr, nullptr };
    bool snap_busy[2] = { false, false };
    int snap_next = 0;

    // Krylov recycling across the four stage systems of one step (same matrix): see gmres()
    struct RecSpace { bool valid = false; int vb = 0, zb = 0, k = 0, pc = 0; std::vector<double> H; };   // leading (k+1) x k raw Hessenberg, ld = k+1
    RecSpace rec[4];
    int rec_vtop = 0, rec_ztop = 0;  // first free slot of V / Zb behind the kept vectors
    int rec_mode = 1;                // 0 off, 1 selected earlier stages (default), 2 every earlier stage
    int rec_keep = 3;                // leading vectors kept per stage (<= 4)
    bool rhs_dots = true;            // inner products for the stage guesses from the RHS kernel's store epilogue (KSFD_TUNE bit 21 clears)
    bool rec_mg = false;             // multigrid-preconditioned solves: the WHOLE first cycle of every stage is kept for the later stages of the step (KSFD_TUNE bit 20 sets; measured: no gain, see gmres)

    // multigrid preconditioner
    std::vector<MGLevel> mg;
    double mg_shift_floor = 0.0;  // lower bound on the shift the multigrid hierarchy is built for (see gmres)
    // online search for that floor (ksfd_step): hill climbing in log2(floor) on the iterations per step
    int sf_dir = 0, sf_hold = 0;  // +1 doubling, -1 halving, 0 settled (and steps to wait before the next probe)
    bool sf_tried_down = false, sf_auto = true;
    double sf_prev_its = 0.0, sf_prev_floor = 0.0;
    bool mg_fp32 = true;         // V cycle with fp32 level vectors when the solve tolerance allows (KSFD_TUNE bit 19 clears); see mg_vcycle32
    bool mg_use32 = false;       // ... decided per step by ksfd_step (ksp_rtol >= 1e-7)
    bool mg_graph_f32 = false;   // precision the captured coarse cycle was recorded in
    bool mg_warm_power = true;   // power iteration of a level starts from the vector of the previous set-up (KSFD_TUNE bit 18 clears)
    bool mg_fuse = true;         // smoother algebra inside the Jacobian-action epilogues (modes 5/6)
    bool mg_ok = false;          // hierarchy exists (2-D, single rank, >= 2 levels)
    double mg_shift = -1.0;      // shift the block diagonals / eigen-bounds were built for
    bool mg_coef_valid = false;  // coarse coefficient planes match the current frozen state
    hipGraphExec_t mg_graph = nullptr;   // captured coarse part of the V cycle (levels >= 1) for mg_graph_shift/x
    double mg_graph_shift = -1.0, mg_graph_bytes = 0.0;
    const void *mg_graph_x = nullptr;
    bool capturing = false, mg_use_graph = true;
    int mg_nu = 2, mg_ncoarse = 400, mg_power_its = 8;   // smoothing sweeps, cap on coarsest-grid sweeps, power iterations
    double mg_ratio = 6.0, mg_coarse_tol = 0.3;    // coarsest-grid reduction target: 0.3 is enough (tools/mg_longrun_tune.py: 24.3 -> 19.9 ms/step at 384^2, same iterations)

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
    // alg < 0: the implementation moves exactly the algorithmic bytes
    Scope(ksfd_handle *h_, int cls, double bytes, double alg = -1.0) : h(h_), on(h_->profiling && !h_->capturing && (h_->prof_only < 0 || h_->prof_only == cls))
    {
        h->bytes_acc += bytes;
        h->prof.bytes[cls] += bytes;
        h->prof.alg_bytes[cls] += alg < 0.0 ? bytes : alg;
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

static int fill_phys(ksfd_handle *h, const ksfd_config *c, KPhys *dst = nullptr)
{
    if (c->nlig < 1 || c->nlig > KSFD_MAXL || c->ngroups < 1 || c->ngroups > KSFD_MAXL)
        return fail(h, KSFD_EINVAL, "nlig=%d ngroups=%d outside 1..%d", c->nlig, c->ngroups, KSFD_MAXL);
    KPhys &P = dst ? *dst : h->P;
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
// 3-D z-marching strip kernels (k_jvp3d_frozen, k_rhs3d_strip)
static bool strip3d_ok(const ksfd_handle *h)
{
    return h->use_fused && h->G.dim == 3 && (h->G.nx % 2 == 0) && h->G.nx >= 4 && h->P.nlig <= 4;
}
