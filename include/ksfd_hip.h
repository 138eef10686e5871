/*
 * ksfd_hip.h -- C ABI of libksfd_hip.so, the MI355X (gfx950) implementation of the hot path of
 * leonavery/KSFD: RHS stencil, analytic Jacobian action, CFL velocity and the implicit
 * (PETSc "TS ROSW ra34pw2"-equivalent) time step, for the periodic 4th-order finite-difference
 * Keller-Segel system in 1, 2 or 3 dimensions.
 *
 * The reference has no FFI for this path: the seam is the Python object
 * `implicitTS(derivs, t0, dt, tmax, maxsteps, rtol, atol)` (KSFD/ksfdts.py:500-561) whose
 * TS.step() calls back into Derivatives.dfdt / Derivatives.Jacobian (KSFD/ksfdsym.py:902-940,
 * 814-886).  Each entry point below names the reference code it stands in for.  The ctypes
 * binding a maintainer would add is shown in INTEGRATION.md; ksfd_amd/lib.py is that binding.
 *
 * Conventions: every function returns 0 on success or a KSFD_E* code; ksfd_last_error() gives
 * the message (mirrors the CHKERR -> RuntimeError convention of cython/ksfdMat/ksfdMat.pyx:17-20).
 * One host thread per handle.  The handle owns all device memory, streams and events; callers
 * own host buffers.  No callbacks into the host language except the optional halo transport.
 */
#ifndef KSFD_HIP_H
#define KSFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSFD_MAX_LIG 12  /* ligand fields after the reference's fourier_series() expansion (KSFD/ksfdligand.py:315-388 has no cap;
                          * 12 = the dispatch width of the kernels, params.h: KSFD_MAXL) */

enum {
    KSFD_OK = 0,
    KSFD_EINVAL = 1,      /* bad argument / unsupported configuration */
    KSFD_EHIP = 2,        /* a HIP runtime call failed */
    KSFD_ENOMEM = 3,
    KSFD_ELINEAR = 4,     /* linear solve did not converge (the reference's SNES failure, ksfdts.py:135) */
    KSFD_ENAN = 5,        /* non-finite error norm */
    KSFD_EREJECT = 6,     /* step rejected more than max_reject times (PETSc TS_DIVERGED_STEP_REJECTED) */
    KSFD_ECOMM = 7        /* halo / reduction transport failed */
};

/* Host-buffer layouts of an (F fields x local grid) state.  F = nlig + 1, field 0 = rho. */
enum {
    KSFD_LAYOUT_PETSC = 0, /* dof fastest: c + F*(i + nx*(j + ny*k))   KSFD/ksfdgrid.py:9-58 */
    KSFD_LAYOUT_SOA = 1,   /* device order, x fastest: i + nx*(j + ny*(k + nz*c)) */
    KSFD_LAYOUT_HDF5 = 2   /* (c,x,y[,z]) C order, last axis fastest  KSFD/ksfdtimeseries.py:497-505 */
};

/* Numeric problem description = what ps.values(t) and ps.Vgroups.ligands() give the reference's
 * operators (KSFD/ksfdsoln.py:104-161, KSFD/ksfdligand.py:306-388, 527-547). */
typedef struct ksfd_config {
    int32_t dim;            /* 1, 2, 3 */
    int32_t nlig;           /* <= KSFD_MAX_LIG */
    int32_t ngroups;
    int32_t cap_kind;       /* 0 tophat, 1 witch   (KSFD/ksfdsoln.py:150-157) */
    int64_t n[3];           /* GLOBAL grid points per axis, unused axes = 1 */
    double L[3];            /* box lengths; spacing = L/n, periodic (KSFD/ksfdgrid.py:138-149) */
    double s2, rhomax, cushion, maxscale, rhomin, Umin;
    const int32_t *lig_group;      /* [nlig] 0-based group of each ligand */
    const double *lig_w, *lig_s, *lig_gamma, *lig_D;   /* [nlig] */
    const double *grp_alpha, *grp_beta;                /* [ngroups] */
} ksfd_config;

/* Slab decomposition along the slowest spatial axis (y in 2-D, z in 3-D) -- the build's stand-in for
 * the PETSc DMDA ghost exchange of width 2 (KSFD/ksfdgrid.py:388-411; globalToLocal at
 * KSFD/ksfdsym.py:704,787,920,1203).  rank r owns slow-axis indices [r*n/size, (r+1)*n/size).
 * transport: 0 none (size must be 1), 1 RCCL (ncclSend/ncclRecv + ncclAllReduce inside the library;
 * nccl_id = the 128-byte ncclUniqueId broadcast by the launcher), 2 host callbacks (exchange /
 * allreduce run by the caller, e.g. mpi4py or torch.distributed; buffers are HOST pointers). */
typedef int (*ksfd_exchange_fn)(void *ctx, const double *send_lo, const double *send_hi,
                                double *recv_lo, double *recv_hi, int64_t count);
typedef int (*ksfd_allreduce_fn)(void *ctx, double *buf, int32_t count, int32_t op /*0 sum, 1 max*/);
/* uniform all-to-all on host buffers: block q of `send` (bytes_per_peer bytes) goes to rank q, block r of `recv` comes from
 * rank r (the own block included).  Optional: without it the spectral solver stays single-rank under transport 2. */
typedef int (*ksfd_alltoall_fn)(void *ctx, const void *send, void *recv, int64_t bytes_per_peer);
typedef struct ksfd_dist {
    int32_t rank, size;
    int32_t transport;
    int32_t device;                /* HIP device ordinal for this rank */
    const void *nccl_id;           /* transport 1 */
    ksfd_exchange_fn exchange;     /* transport 2 */
    ksfd_allreduce_fn allreduce;   /* transport 2 */
    void *ctx;
    ksfd_alltoall_fn alltoall;     /* transport 2, may be NULL */
} ksfd_dist;

/* Time-step controls = the PETSc options every shipped options file passes
 * (options84:47-67: -ts_type rosw -ts_adapt_type basic -ts_adapt_clip 0.1,5 -ts_adapt_dt_min/max)
 * plus the Krylov knobs that replace -ksp_type preonly -pc_type lu. */
typedef struct ksfd_step_opts {
    double rtol, atol;          /* TSSetTolerances (KSFD/ksfdts.py:136) */
    int32_t adapt;              /* 1 = TSAdaptBasic, 0 = -ts_adapt_type none */
    int32_t max_reject;         /* rejections tolerated inside one call (PETSc default 10); < 0: unlimited (-ts_max_reject -1) */
    double clip_lo, clip_hi;    /* 0.1, 5 */
    double dt_min, dt_max;      /* 1e-20, 1e4 */
    double safety, reject_safety; /* 0.9, 0.5 */
    double ksp_rtol, ksp_atol;  /* GMRES: stop at ||r|| <= max(ksp_rtol*||b||, ksp_atol) */
    int32_t ksp_restart, ksp_max_it;
    int32_t pc_type;            /* 0 none; 1 geometric multigrid V cycle always; 2 automatic (default): the spectral defect correction
                                 * (constant-coefficient part of shift*I - J inverted by FFT; 2-D and 3-D, extents 2^k or 3*2^k, on 1, 2,
                                 * 4 or 8 slab ranks) while it converges in a few sweeps, else multigrid when the step is stiff,
                                 * Chebyshev polynomial + flexible GMRES when mildly stiff, none when not; 3 polynomial only;
                                 * 4 spectral always */
    int32_t reserved;           /* flags.  bit 0: classic two-pass CGS2 instead of CGS2 with the algebraic second projection;
                                 * bit 1: single attempt per call -- a rejected step returns with accepted = 0 and *hstep = the
                                 * controller's proposal (callers that must refresh stage-time data per attempt);
                                 * bit 2: the previous attempt of this step was rejected (TSAdaptBasic then applies reject_safety);
                                 * bit 3: the spectral defect correction verifies EVERY solve with a final residual evaluation (no
                                 * predicted last sweep) */
} ksfd_step_opts;

typedef struct ksfd_step_stats {
    int32_t accepted;           /* 1 if the state advanced */
    int32_t rejections;
    int32_t linear_its;         /* linear iterations over all stages and attempts: GMRES iterations, or applications of the spectral
                                 * inverse (sweeps of the defect correction) where pc_used has bit 8 */
    int32_t rhs_evals, jvp_evals;
    int32_t pc_used;            /* preconditioners the stage solves of this call ran with, OR of: 1 none, 2 multigrid V cycle,
                                 * 4 Chebyshev polynomial, 8 spectral (constant-coefficient FFT) */
    double wrms;                /* error norm of the last attempt */
    double h_used;              /* step actually taken (valid when accepted) */
    double ksp_resid;           /* last relative residual */
    double bytes;               /* algorithmic HBM bytes of all kernels launched by the call */
    int32_t launches;           /* kernels / device copies the call enqueued on the compute stream */
    int32_t host_syncs;         /* times the host waited for a device result inside the call (reduction hand-overs, stream
                                 * synchronisations): the latency-bound part of a step on many slab ranks */
    int32_t residual_evals;     /* true residuals b - A x evaluated by the spectral defect correction (one Jacobian action each) */
    int32_t predicted_final;    /* stage solves whose LAST sweep was applied on the measured contraction of the earlier sweeps
                                 * instead of a residual evaluation (ksp_rtol >= 1e-8 only; opts.reserved bit 3 switches it off) */
} ksfd_step_stats;

/* Per-kernel-class timing gathered with HIP events on the library's compute stream. */
#define KSFD_NKCLASS 14
typedef struct ksfd_profile {
    double ms[KSFD_NKCLASS];       /* accumulated device time */
    double bytes[KSFD_NKCLASS];    /* accumulated bytes the IMPLEMENTATION must move (e.g. frozen-coefficient planes included) */
    int64_t launches[KSFD_NKCLASS];
    double alg_bytes[KSFD_NKCLASS];/* accumulated ALGORITHMIC bytes of the same launches (RHS 16*F*N,
                                    * Jacobian action 24*F*N, vector passes = operands * 8*F*N) */
} ksfd_profile;
const char *ksfd_kernel_class_name(int32_t cls);

typedef struct ksfd_handle ksfd_handle;

/* 128-byte ncclUniqueId from the same librccl the library resolves (rank 0 calls, the launcher broadcasts it) */
int ksfd_rccl_unique_id(void *out128);

/* -- lifetime.  dist == NULL: single GPU (device 0 or HIP_VISIBLE_DEVICES), in-kernel periodic wrap. */
int ksfd_create(const ksfd_config *cfg, const ksfd_dist *dist, ksfd_handle **out);
void ksfd_destroy(ksfd_handle *h);
const char *ksfd_last_error(const ksfd_handle *h);          /* h may be NULL: error of a failed create */
int ksfd_update_params(ksfd_handle *h, const ksfd_config *cfg);   /* time-dependent ps.values(t): Jacobian / operators at t */
/* Time-dependent parameters inside a step: the reference hands ps.values(t) at the STAGE time t_n + ASum_i*h to every RHS
 * evaluation (KSFD/ksfdsym.py:1303-1312, 1430-1439 via implicitIF, KSFD/ksfdts.py:563-596) while the Jacobian uses
 * ps.values(t_n).  stage 0..3 sets the table the i-th stage RHS of ksfd_step uses, -1 all four; cfg == NULL clears (the
 * stages then use the ksfd_update_params table).  Grid and ligand count must match the handle's. */
int ksfd_set_stage_params(ksfd_handle *h, int32_t stage, const ksfd_config *cfg);
int ksfd_local_range(const ksfd_handle *h, int64_t *slow_begin, int64_t *slow_end); /* Grid._ranges on the slab axis */
int64_t ksfd_local_size(const ksfd_handle *h);               /* F * local points: length of every host buffer */

/* -- state (the TS solution Vec u; KSFD/ksfdts.py:141-152).  Host buffers hold the LOCAL slab. */
int ksfd_set_state(ksfd_handle *h, const double *u_host, int32_t layout);
int ksfd_get_state(ksfd_handle *h, double *u_host, int32_t layout);
/* "next" row f2: asynchronous read-out for writers (TimeSeries.store / makeSaveMonitor, KSFD/ksfdtimeseries.py:484-509,
 * KSFD/ksfdts.py:466-497) -- the layout transform is ordered on the compute stream, the D2H copy runs on a third stream
 * into one of two pinned buffers owned by the handle while the stepper carries on.  _wait blocks until the copy has landed
 * and returns the buffer (ksfd_local_size doubles), valid until the second-next _begin.  Thread rule: _begin from the
 * stepping thread; _wait may be called from a writer thread. */
int ksfd_snapshot_begin(ksfd_handle *h, int32_t layout, int32_t *slot);
int ksfd_snapshot_wait(ksfd_handle *h, int32_t slot, const double **host);
/* "next" row f3: the reference's start_values (ksfdsolver2.py:580-639) evaluated on the device for this rank's slab:
 * rho = rho0 + smoothstep interpolation (KSFD/ksfdrandom.py:116,194-214) of the coarse samples z (GLOBAL coarse grid
 * nc[0..2], x fastest, periodic; the caller draws them from numpy's default_rng so the stream matches ksfdrandom.py:44-49),
 * U_l = rho*s_l/gamma_l (:636-637). */
int ksfd_set_state_random(ksfd_handle *h, const int64_t nc[3], const double *z_coarse_host, double rho0);
/* Device-side checkpoint of the state together with what the solver remembers from step to step (spectral-radius estimate,
 * preconditioner adaptation): op 0 = save, op 1 = restore.  One slot, allocated on first use.  Used by bench.py to time a
 * pinned window of steps repeatedly and by callers that want to retry from a known state without a host round trip. */
int ksfd_checkpoint(ksfd_handle *h, int32_t op);
double *ksfd_device_state(ksfd_handle *h);      /* device pointer, SoA with ghost rows; plane stride below */
int64_t ksfd_device_plane_stride(const ksfd_handle *h);
int64_t ksfd_device_interior_offset(const ksfd_handle *h);

/* -- sources(t) added by Derivatives.dfdt (KSFD/ksfdsym.py:930-936).  stage -1 sets all four stage
 *    slots; stage 0..3 sets the field used at stage time t + ASum[stage]*h.  NULL clears. */
int ksfd_set_source(ksfd_handle *h, int32_t stage, int32_t field, const double *src_host, int32_t layout);

/* -- operators on host vectors (parity tests and host-driven integrators).
 *    u_host == NULL: use the handle's state.  Inputs are clamped the way Derivatives.groom does
 *    (KSFD/ksfdsym.py:888-900) on the fly; the stored state is not modified. */
int ksfd_rhs(ksfd_handle *h, double t, const double *u_host, double *out_host, int32_t layout);   /* Derivatives.dfdt */
int ksfd_jvp(ksfd_handle *h, const double *u_host, const double *v_host, double *out_host,
             int32_t layout);                                    /* action of Derivatives.Jacobian(u) on v */
int ksfd_velocity(ksfd_handle *h, const double *u_host, double *vel_host, int32_t layout); /* Derivatives.velocity: dim planes */
int ksfd_velocity_max(ksfd_handle *h, double vmax[3]);           /* per-axis max|grad G| of the state; CFL_step, ksfdts.py:302-319 */

/* -- outer-loop helpers of KSFDTS.solve (KSFD/ksfdts.py:202-284) acting on the device state */
int ksfd_groom(ksfd_handle *h);                                  /* KSFDTS.groom :231-237 */
int ksfd_count_worms(ksfd_handle *h, double *total);             /* sum(rho) over all ranks :239-246 */
int ksfd_scale_rho(ksfd_handle *h, double factor);               /* conserve_worms :248-256 */
int ksfd_mul_rho(ksfd_handle *h, const double *factor_host);     /* add_variance: rho *= exp(sd*N(0,1)) :268-284; local SoA plane */

/* -- assembled Jacobian export ("next" row f4): the matrix Derivatives.Jacobian + ksfdMat.setValuesJacobian assemble
 *    (KSFD/ksfdsym.py:814-886, cython/ksfdMat/ksfdMat.pyx:55-180), evaluated at the resident (clamped) state, as CSR over
 *    this rank's rows.  Ordering = the reference's PETSc Vec: unknown F*point + dof, point x-fastest; rows are numbered
 *    from this rank's first owned point (rowptr starts at 0), columns are GLOBAL unknown indices (periodic wrap applied),
 *    not sorted within a row.  Row of rho: F*(4*dim+1) entries; row of U_l: 4*dim+2.  Caller allocates from _nnz. */
int ksfd_jacobian_nnz(ksfd_handle *h, int64_t *nrows_local, int64_t *nnz_local);
int ksfd_jacobian_csr(ksfd_handle *h, int64_t *rowptr /* nrows+1 */, int64_t *col /* nnz */, double *val /* nnz */);

/* -- the implicit step that replaces petsc4py TS.step() (KSFD/ksfdts.py:211):
 *    4-stage Rosenbrock-W RA34PW2 with frozen Jacobian J(t_n,u_n), shift 1/(gamma h), each stage
 *    solved matrix-free by restarted GMRES on the device; TSAdaptBasic error control.
 *    in: *t, *hstep (step to try).  out: *t advanced when accepted, *hstep = next proposed step. */
void ksfd_default_step_opts(ksfd_step_opts *o);
int ksfd_step(ksfd_handle *h, double *t, double *hstep, const ksfd_step_opts *opts, ksfd_step_stats *stats);
int ksfd_get_last_error_vector(ksfd_handle *h, double *err_host, int32_t layout); /* embedded-minus-main of last attempt */

/* -- measurement
 * on: 0 off; 1 HIP-event pair around every launch (costs ~8 us of device time per launch); 2+c events only around
 * launches of kernel class c (launch and byte counters of the other classes keep running) */
int ksfd_set_profiling(ksfd_handle *h, int32_t on);
int ksfd_get_profile(ksfd_handle *h, ksfd_profile *p, int32_t reset);
int ksfd_synchronize(ksfd_handle *h);
/* raw kernel benchmark used by bench.py/profiles: run `reps` launches of one kernel class on the state,
 * timed with HIP events on the compute stream; returns average ms per launch. */
int ksfd_bench_kernel(ksfd_handle *h, int32_t cls, int32_t reps, double *avg_ms, double *bytes_per_launch);
/* use_fused: bit0 fused kernels, bit1 set = recompute (non-frozen) Jacobian action, bit2 set = no halo/compute overlap,
 * bit3 set = pipelined GMRES with device-resident Hessenberg/Givens state (off by default);
 * bit4 set = no Krylov recycling across the four stage systems of a step, bit5 set = recycle from every earlier stage
 * (default: from the stages measured to matter: 1 for 2 and 3, 1 and 3 for 4), bits 6-8 = leading Arnoldi vectors kept
 * per stage (1..4, 0 keeps the default 3); bit9 set = keep the polynomial preconditioner's temporaries and coefficient
 * copy in fp64 (default: fp32 storage inside p(A) only; the Krylov vectors, A z_j and the solution are fp64 always);
 * bit10 set = form the stage vectors with separate passes instead of inside the RHS kernel;
 * bit11 set = fetch reduction results with a stream synchronisation + copy instead of spinning on a flag the
 * reduction kernel raises in mapped host memory;
 * bit12 set = multigrid smoother as separate kernels instead of inside the Jacobian-action epilogues;
 * bit14 set = spectral stage solves start from zero instead of from the span of the earlier stage solutions of the step;
 * bit13 set = 3-D RHS through the generic one-thread-per-point stencil pass instead of the z-marching strip kernel;
 * bit15 set = the stage RHS kernel re-reads the stage vectors it adds at the store instead of carrying their combination along from
 * the window load; bit16 set = the spectral defect correction verifies every solve with a residual evaluation (no predicted last sweep);
 * bit17 set = GMRES keeps the restart length it was given (default: a cycle that ends without convergence is followed by one of twice
 * the length, up to the 30 ... 120 basis vectors ksfd_create found room for);
 * bit18 set = every multigrid set-up estimates the Chebyshev bound of a level by a cold power iteration (default: warm start from the
 * vector of the previous set-up, at most three iterations);
 * bit21 set = the inner products of a new stage right-hand side with the earlier ones (stage guesses) are a pass of their own instead of
 * part of the RHS kernel's store epilogue;
 * bit20 set = multigrid-preconditioned stage solves keep the whole first GMRES cycle of every stage and project the later stages of the step
 * on it first (experiment, measured slower: krylov.hip.h);
 * bit19 set = the V cycle keeps its level vectors in fp64 at every tolerance (default: fp32 level vectors when ksp_rtol >= 1e-7, 2-D);
 * yseg_*: rows per wave segment; <=0 keeps */
int ksfd_set_tuning(ksfd_handle *h, int32_t use_fused, int32_t yseg_rhs, int32_t yseg_jvp);
/* multigrid knobs (<=0 keeps): smoothing sweeps per side, cap on coarsest-grid sweeps, power iterations for the
 * Chebyshev bound, smoothing interval ratio lambda_max/lambda_min, coarsest-grid reduction target */
int ksfd_set_mg_params(ksfd_handle *h, int32_t nu, int32_t ncoarse_max, int32_t power_its, double ratio, double coarse_tol);
/* The hierarchy is built for max(shift, floor)*I - J; the floor is searched online by ksfd_step when 1/(gamma h) has fallen
 * below the growth rate of the instability and the iteration count explodes.  Environment KSFD_PC_SIGMA=<x> fixes it
 * instead (0 = no floor) -- an experiment knob, not part of the ABI. */
/* Chebyshev polynomial preconditioner: highest degree (default 6, at most 7; 0 = off), the residual reduction per
 * outer iteration that picks the degree (default 0.02; <= 0 keeps), and the stiffness h*gamma*lambda_max(diffusion)
 * above which pc_type 2 hands over to multigrid (default 75; <= 0 keeps) */
int ksfd_set_poly_params(ksfd_handle *h, int32_t max_degree, double target, double mg_threshold);
/* Spectral preconditioner: z = (shift*I - J0)^-1 v with J0 the constant-coefficient part of the Jacobian at the resident state
 * (grid means of rho*G_rho, rho*G_Ul; the 4th-order star's exact symbol), three hand-written FFT kernels (csrc/spectral.hip.h).
 * _apply is the parity/test entry (host vectors; KSFD_EINVAL where the handle has no spectral solver: 1-D, extents outside
 * {2^k, 3*2^k} or 32..16384, rank counts other than 1, 2, 4, 8, a transport without an all-to-all).  _params: stiffness
 * h*gamma*lambda_max(diffusion) from which pc_type 2 prefers it (default 0.1 where the fused 2-D residual kernel runs, 0.3
 * elsewhere; <= 0 keeps) and enable (0 = never pick it automatically, 1 = default, < 0 keeps). */
int ksfd_spectral_apply(ksfd_handle *h, double shift, const double *v_host, double *out_host, int32_t layout);
int ksfd_set_spectral_params(ksfd_handle *h, double from_stiffness, int32_t enable);

#ifdef __cplusplus
}
#endif
#endif /* KSFD_HIP_H */
