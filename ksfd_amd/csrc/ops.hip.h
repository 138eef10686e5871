// libksfd_hip.so -- halo exchange, host-visible reductions, launch wrappers of every kernel class, host<->device layouts
// (part of the single translation unit ksfd_hip.hip; included from there in this order:
//  handle.hip.h, ops.hip.h, mg_host.hip.h, krylov.hip.h)
#pragma once
// ---- halo exchange (DMDA globalToLocal stand-in, KSFD/ksfdsym.py:919-920) -----------------------
static int halo(ksfd_handle *h, double *vec)
{
    if (!h->ring) return KSFD_OK;
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
    // several ranks with a device-side all-reduce (RCCL): a one-block k_publish behind the all-reduce hands the result over
    // the same way; ksfd_amd.dist.open_handle checks that path end to end on a new handle and clears zero_copy if it fails
    const bool zc = h->zero_copy && !h->capturing && rows <= 128 && (!h->ring || h->tr->device_allreduce());
    const bool zc_here = zc && !h->ring;
    h->n_host_sync++;
    const unsigned long long seq = zc ? ++h->pub_seq : 0;
    {
        Scope sc(h, KC_REDUCE, 8.0 * rows * (double)nblk);
        if (zc_here) hipLaunchKernelGGL(k_reduce_rows, dim3(rows), dim3(KSFD_BLOCK), 0, h->st, h->part, nblk, op, h->dres, h->hres_dev, h->pub_count, h->pub_flag_dev, seq);
        else hipLaunchKernelGGL(k_reduce_rows, dim3(rows), dim3(KSFD_BLOCK), 0, h->st, h->part, nblk, op, h->dres);
    }
    if (zc_here) { HIPCHK(h, hipGetLastError()); return spin_for(h, seq); }
    if (h->ring) {
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
    case 6: { constexpr int NL = 6; CALL; } break;                                                 \
    case 7: { constexpr int NL = 7; CALL; } break;                                                 \
    case 8: { constexpr int NL = 8; CALL; } break;                                                 \
    case 9: { constexpr int NL = 9; CALL; } break;                                                 \
    case 10: { constexpr int NL = 10; CALL; } break;                                               \
    case 11: { constexpr int NL = 11; CALL; } break;                                               \
    default: { constexpr int NL = 12; CALL; } break;                                               \
    }

static KStrips make_strips(const ksfd_handle *h, bool jvp = false)
{
    KStrips S;
    S.nstrips = (int)((h->G.nx + KSFD_STRIP_OUT - 1) / KSFD_STRIP_OUT);
    S.yseg = jvp ? h->yseg_jvp : h->yseg;
    // small grids: shorter segments so that there are enough waves to fill 256 CUs (a wave costs ~1 us per row it
    // marches; the 4 halo rows per segment are L2 hits at these sizes)
    {
        static const long long t_jvp = getenv("KSFD_WAVES_JVP") ? atoll(getenv("KSFD_WAVES_JVP")) : 4096;       // experiments (tools/yseg_sweep.py)
        static const long long t_rhs = getenv("KSFD_WAVES_RHS") ? atoll(getenv("KSFD_WAVES_RHS")) : 6144;
        const long long target = jvp ? t_jvp : t_rhs;
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

// launch geometry of the 3-D z-marching strip kernels
// rows of y per block of the 3-D strip kernels (K3D).  Measured at 512^3 (bench.py --dim 3, ms per step): 4 rows 110.6, 4 rows with a
// barrier per plane 107.8, 8 rows 111.9, 8 rows + barrier 115.5 -- so 4 rows marching in step; KSFD_ROWS3D=8 / KSFD_SYNC3D=0 for measurements
static int rows3d(const ksfd_handle *h, long long ny)
{
    static const int env = getenv("KSFD_ROWS3D") ? atoi(getenv("KSFD_ROWS3D")) : 0;
    (void)ny;
    return (env == 8 && h->P.nlig == 1) ? 8 : 4;
}
static K3D make_k3d(const ksfd_handle *h)
{
    const KGeom &G = h->G;
    K3D K;
    K.rows = rows3d(h, G.ny);
    {
        static const int sync_env = getenv("KSFD_SYNC3D") ? atoi(getenv("KSFD_SYNC3D")) : 1;
        K.sync = sync_env && (G.ny % K.rows == 0);
    }
    K.nstrips = (int)((G.nx + KSFD_STRIP_OUT - 1) / KSFD_STRIP_OUT);
    K.nygrp = (int)((G.ny + K.rows - 1) / K.rows);
    K.zseg = h->zseg;
    {
        long long fit = (long long)K.nstrips * K.nygrp * G.sloc / (K.rows == 8 ? 512 : 1024);      // enough blocks for 256 CUs
        if (fit < 2) fit = 2;
        if (fit < K.zseg) K.zseg = (int)fit;
    }
    K.nzseg = (int)((G.sloc + K.zseg - 1) / K.zseg);
    const long long nb3 = (long long)K.nstrips * K.nygrp * K.nzseg;
    K.nblocks = (int)((nb3 + 7) / 8 * 8);
    return K;
}

// second-generation 3-D Jacobian action (k_jvp3d_lds: y-neighbours through the LDS, dG on the fly): one or two ligands, ny a multiple of
// its 8 rows per block; KSFD_J3L=0 keeps the first generation (dG plane pass + k_jvp3d_frozen)
static bool j3l_ok(ksfd_handle *h)
{
    static const int env = getenv("KSFD_J3L") ? atoi(getenv("KSFD_J3L")) : 1;
    if (!env || !strip3d_ok(h) || h->P.nlig > 2 || h->G.ny % KSFD_J3L_ROWS) return false;
    if (!h->j3l_attr_set) {
        hipError_t e = hipSuccess;
        if (h->P.nlig == 1) {
            e = hipFuncSetAttribute((const void *)k_jvp3d_lds<1, double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ksfd_j3l_lds_bytes<1>());
            if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_jvp3d_lds<1, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ksfd_j3l_lds_bytes<1>());
        } else {
            e = hipFuncSetAttribute((const void *)k_jvp3d_lds<2, double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ksfd_j3l_lds_bytes<2>());
            if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_jvp3d_lds<2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ksfd_j3l_lds_bytes<2>());
        }
        h->j3l_attr_set = true;
        h->j3l_usable = e == hipSuccess;
        if (e != hipSuccess) hipGetLastError();
    }
    return h->j3l_usable;
}
static K3D make_k3d_lds(const ksfd_handle *h)
{
    const KGeom &G = h->G;
    K3D K;
    K.rows = KSFD_J3L_ROWS; K.sync = 0;
    K.nstrips = (int)((G.nx + KSFD_STRIP_OUT - 1) / KSFD_STRIP_OUT);
    K.nygrp = (int)(G.ny / KSFD_J3L_ROWS);
    K.zseg = h->zseg;
    {
        long long fit = (long long)K.nstrips * K.nygrp * G.sloc / 1024;      // one block per CU: four rounds of blocks at least
        if (fit < 2) fit = 2;
        if (fit < K.zseg) K.zseg = (int)fit;
    }
    K.nzseg = (int)((G.sloc + K.zseg - 1) / K.zseg);
    const long long nb3 = (long long)K.nstrips * K.nygrp * K.nzseg;
    K.nblocks = (int)((nb3 + 7) / 8 * 8);
    return K;
}
// launch of k_jvp3d_lds for 1 or 2 ligands and the storage type of the output
template <typename TO>
static void j3l_launch(ksfd_handle *h, const K3D &K, const double *v, int mode, double shift, TO *out, const double *yadd, double alpha, double beta, double *normpart)
{
    const KGeom &G = h->G;
    if (h->P.nlig == 1) hipLaunchKernelGGL((k_jvp3d_lds<1, TO>), dim3(K.nblocks), dim3(KSFD_J3L_ROWS * KSFD_WAVE), ksfd_j3l_lds_bytes<1>(), h->st, G, h->P, K, (const double *)h->coef, v, mode, shift, out, yadd, alpha, beta, normpart);
    else hipLaunchKernelGGL((k_jvp3d_lds<2, TO>), dim3(K.nblocks), dim3(KSFD_J3L_ROWS * KSFD_WAVE), ksfd_j3l_lds_bytes<2>(), h->st, G, h->P, K, (const double *)h->coef, v, mode, shift, out, yadd, alpha, beta, normpart);
}

static KSrc src_of(const ksfd_handle *h, int stage)
{
    KSrc s;
    for (int c = 0; c <= KSFD_MAXL; c++) s.p[c] = (stage >= 0 && c < h->G.F) ? h->src[stage][c] : nullptr;
    return s;
}

// out = f(u) (+sources of `stage`); u must have valid ghosts when size>1
// capacity of h->part in doubles (ksfd_create)
static inline long long part_capacity() { return (long long)(2 * KSFD_MAXDOT + 4) * 4096; }

// want_norm (fused 2-D path only): ||out||^2 lands in h->hres[0] without a pass of its own (per-wave partials in the store epilogue)
// halo_vec (slab ranks): a vector among the inputs whose ghost rows have NOT been exchanged yet (the newest stage vector).  On the
// 2-D strip path they travel on the communication stream while the interior segments are computed -- the scheme of
// op_jvp_frozen_halo: only the first and the last segment read ghost rows -- elsewhere they are exchanged first.
// ndots (with want_norm, fused 2-D path): <out, dotv + q*vlen>, q < ndots <= 2, land in h->hres[1 + q] beside ||out||^2 in h->hres[0]
static int op_rhs(ksfd_handle *h, const double *u, int stage, double *out, const KComb *cmb = nullptr, bool want_norm = false, double *halo_vec = nullptr,
                  int ndots = 0, const double *dotv = nullptr)
{
    const KGeom &G = h->G;
    KSrc S = src_of(h, stage);
    // time-dependent parameters: the reference evaluates ps.values(t) at the STAGE time of every RHS call
    const KPhys &PP = (stage >= 0 && stage < 4 && h->Pst_valid[stage]) ? h->Pst[stage] : h->P;
    if (!h->ring) halo_vec = nullptr;
    if (fused_ok(h)) {
        KStrips K = make_strips(h);
        KComb C = cmb ? *cmb : KComb{};
        const long long nwaves = (long long)K.nstrips * K.nseg;
        const bool fused_norm = want_norm && nwaves * (1 + ndots) <= part_capacity();
        KDots D = KDots{};
        if (fused_norm && ndots > 0) { D.n = std::min(ndots, 2); D.stride = nwaves; for (int q = 0; q < D.n; q++) D.v[q] = dotv + (int64_t)q * h->vlen; }
        // the vectors added at the store are the ones the stage argument is formed from: one read serves both (k_rhs2d_fused<NL, true>)
        bool carry = h->rhs_carry && C.nout > 0 && C.nout == C.nin;
        for (int j = 0; j < C.nout && carry; j++) carry = C.yin[j] == C.yout[j];
        auto launch = [&](const KStrips &Kx, double frac, double *np) {
            Scope sc(h, KC_RHS, vbytes(h, 2 + C.nin + (carry ? 0 : C.nout) + D.n) * frac, vbytes(h, 2 + C.nin + C.nout + D.n) * frac);
            if (carry) { NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_rhs2d_fused<NL, true>), dim3(Kx.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, PP, Kx, u, S, out, C, np, D)); }
            else { NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_rhs2d_fused<NL>), dim3(Kx.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, PP, Kx, u, S, out, C, np, D)); }
        };
        double *np = fused_norm ? h->part : (double *)nullptr;
        const bool ovl = halo_vec && h->overlap && K.nseg >= 3;
        if (halo_vec && !ovl) { int rc = halo(h, halo_vec); if (rc) return rc; }
        if (!ovl) launch(K, 1.0, np);
        else {
            HIPCHK(h, hipEventRecord(h->ev_ready, h->st));
            KStrips Ki = K;
            Ki.seg0 = 1; Ki.seg_stride = 1; Ki.nseg = K.nseg - 2;
            long long nb = ((long long)Ki.nstrips * Ki.nseg + 3) / 4;
            Ki.nblocks = (int)((nb + 7) / 8 * 8);
            launch(Ki, (double)Ki.nseg / K.nseg, np);
            HIPCHK(h, hipStreamWaitEvent(h->st_comm, h->ev_ready, 0));
            {
                Scope sc(h, KC_HALO, 4.0 * 2.0 * 8.0 * G.F * (double)G.inner * 2.0);
                if (h->tr->exchange(halo_vec, G.F, G.plane, G.inner, G.sloc, G.ng, h->st_comm)) return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
            }
            HIPCHK(h, hipEventRecord(h->ev_halo, h->st_comm));
            HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_halo, 0));
            KStrips Kb = K;
            Kb.seg0 = 0; Kb.seg_stride = K.nseg - 1; Kb.nseg = 2;
            nb = ((long long)Kb.nstrips * Kb.nseg + 3) / 4;
            Kb.nblocks = (int)((nb + 7) / 8 * 8);
            launch(Kb, 2.0 / K.nseg, np ? np + (long long)K.nstrips * (K.nseg - 2) : nullptr);
        }
        HIPCHK(h, hipGetLastError());
        if (fused_norm) return reduce_rows(h, 1 + D.n, (int)nwaves, 0);
        if (want_norm) return fail(h, KSFD_EINVAL, "op_rhs: fused norm needs the strip kernels");
        return KSFD_OK;
    }
    if (halo_vec) { int rc = halo(h, halo_vec); if (rc) return rc; }
    if (strip3d_ok(h) && h->rhs3d_strip) {
        // 3-D: G plane (+ the stage argument, when the stage algebra rides along), then the z-marching 13-point star
        if (want_norm) return fail(h, KSFD_EINVAL, "op_rhs: fused norm is a 2-D feature");
        KComb C = cmb ? *cmb : KComb{};
        const double *uin = u;
        const double *Gplane = h->Gb;
        const int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        // first stage of a step: the argument is the resident state itself and G(u) is plane 1 of the frozen coefficients the step has
        // just made of it (same parameters): nothing to form, no pass
        const bool reuse_G = C.nin == 0 && u == h->u && h->coef_fresh && h->use_frozen && h->coef && &PP == &h->P;
        if (reuse_G) Gplane = h->coef + G.plane;
        else {
            Scope sc(h, KC_GFIELD, 8.0 * ((1 + C.nin) * G.F + (C.nin ? G.F : 0) + 1) * (double)G.plane, C.nin ? vbytes(h, 2 + C.nin) : 0.0);
            NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_gfield_comb<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, PP, u, C, C.nin ? h->Z : (double *)nullptr, h->Gb));
        }
        if (C.nin) uin = h->Z;
        K3D K = make_k3d(h);
        Scope sc(h, KC_RHS, 8.0 * (2.0 * G.F + 1 + C.nout * G.F) * (double)G.nloc, vbytes(h, 2 + C.nout));
        if (K.rows == 8) hipLaunchKernelGGL((k_rhs3d_strip<1, 8>), dim3(K.nblocks), dim3(8 * KSFD_WAVE), 0, h->st, G, PP, K, uin, Gplane, S, out, C);
        else NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_rhs3d_strip<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, PP, K, uin, Gplane, S, out, C));
    } else {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (G.F + 1) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_gfield<NL, false>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, PP, u, (const double *)nullptr, h->Gb, (double *)nullptr));
        }
        int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        Scope sc(h, KC_RHS, vbytes(h, 2) + 8.0 * (double)G.nloc, vbytes(h, 2));
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_rhs_generic<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, PP, u, h->Gb, S, out));
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
        Scope sc(h, KC_JVP, vbytes(h, 3) + 16.0 * (double)G.nloc, vbytes(h, 3));
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jvp_generic<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, u, v, h->Gb, h->dGb, mode, shift, out));
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// Once per step: C = [rho, G, G_rho, G_U..] of the (ghost-filled) state u
// want_means: the grid means of the spectral preconditioner come out of the same launch (resident state only)
static int op_jcoef(ksfd_handle *h, const double *u, bool want_means = false)
{
    const KGeom &G = h->G;
    int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    if (!h->coef32 && h->poly_fp32 && fused_ok(h) && h->P.nlig <= 4 && G.plane % 2 == 0 && G.inner % 2 == 0 &&
        hipMalloc((void **)&h->coef32, sizeof(float) * (size_t)(3 + h->P.nlig) * G.plane) != hipSuccess) { h->coef32 = nullptr; h->poly_fp32 = false; }
    float *c32 = h->poly_fp32 ? h->coef32 : nullptr;
    want_means = want_means && (long long)(1 + h->P.nlig) * nbp <= part_capacity();
    {
    Scope sc(h, KC_GFIELD, (8.0 * (G.F + 3 + h->P.nlig) + (c32 ? 4.0 * (3 + h->P.nlig) : 0.0)) * (double)G.plane);
    NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jcoef<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, h->P, u, h->coef, c32, want_means ? h->part : (double *)nullptr));
    }
    HIPCHK(h, hipGetLastError());
    if (want_means) {
        int rc = reduce_rows(h, 1 + h->P.nlig, nbp, 0);
        if (rc) return rc;
        const double ntot = (double)h->cfg.n[0] * (double)h->cfg.n[1] * (double)h->cfg.n[2];
        h->spec.a_rr = h->hres[0] / ntot;
        for (int l = 0; l < h->P.nlig; l++) h->spec.a_rU[l] = h->hres[1 + l] / ntot;
    }
    return KSFD_OK;
}

// Coefficient planes of the RESIDENT state, computed once per state: the CFL check after a step and the next step's
// Jacobian need the same planes (the reference evaluates G twice there: velocity, KSFD/ksfdsym.py:1188-1209, and Jacobian).
// coef_fresh is cleared by everything that changes h->u.
static int ensure_coef(ksfd_handle *h, bool ghosts_done = false)
{
    if (h->coef_fresh) return KSFD_OK;
    int rc;
    if (!ghosts_done && (rc = halo(h, h->u))) return rc;
    if ((rc = op_jcoef(h, h->u, h->spec.ok))) return rc;
    h->coef_fresh = true;
    h->mg_coef_valid = false; h->mg_shift = -1.0;
    h->spec.means_valid = h->spec.ok && (long long)(1 + h->P.nlig) * std::min<long long>((h->G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096) <= part_capacity();
    return KSFD_OK;
}

// Jacobian action from the frozen coefficients (see stencil.hip.h, "Frozen-Jacobian path")
// want_norm (fused 2-D path only): ||out||^2 -> h->hres[0]
static int op_jvp_frozen(ksfd_handle *h, const double *v, int mode, double shift, double *out,
                         const double *yadd = nullptr, double alpha = 0.0, double beta = 0.0, bool want_norm = false)
{
    const KGeom &G = h->G;
    const double nplanes = (3 + h->P.nlig) + 2.0 * G.F + ((mode == 2 || mode == 3) ? G.F : 0);   // coefficients + v + out (+ yadd)
    const double alg = 8.0 * (3.0 * G.F + ((mode == 2 || mode == 3) ? G.F : 0)) * (double)G.nloc;   // read u, v, write out (+ the fused vector operand)
    if (fused_ok(h)) {
        KStrips K = make_strips(h, true);
        const long long nwaves = (long long)K.nstrips * K.nseg;
        if (want_norm && nwaves > part_capacity()) return fail(h, KSFD_EINVAL, "op_jvp_frozen: too many waves for the fused norm");
        {
            Scope sc(h, KC_JVP, 8.0 * nplanes * (double)G.nloc, alg);
            NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, K, (const double *)h->coef, v, mode, shift, out, yadd, alpha, beta, KSmooth{}, want_norm ? h->part : (double *)nullptr));
        }
        HIPCHK(h, hipGetLastError());
        return want_norm ? reduce_rows(h, 1, (int)nwaves, 0) : KSFD_OK;
    } else if (!want_norm && j3l_ok(h)) {
        K3D K = make_k3d_lds(h);
        Scope sc(h, KC_JVP, 8.0 * nplanes * (double)G.nloc, alg);
        j3l_launch<double>(h, K, v, mode, shift, out, yadd, alpha, beta, nullptr);
    } else if (h->use_fused && G.dim == 3 && (G.nx % 2 == 0) && G.nx >= 4 && h->P.nlig <= 4) {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (2 + h->P.nlig + G.F) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dg_frozen<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)h->coef, v, h->dGb));
        }
        K3D K = make_k3d(h);
        Scope sc(h, KC_JVP, 8.0 * (2.0 * G.F + 3 + ((mode == 2 || mode == 3) ? G.F : 0)) * (double)G.nloc, alg);
        if (K.rows == 8) hipLaunchKernelGGL((k_jvp3d_frozen<1, double, 8>), dim3(K.nblocks), dim3(8 * KSFD_WAVE), 0, h->st, G, h->P, K, (const double *)h->coef, v, (const double *)h->dGb, mode, shift, out, yadd, alpha, beta);
        else NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp3d_frozen<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, K, (const double *)h->coef, v, (const double *)h->dGb, mode, shift, out, yadd, alpha, beta));
    } else {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (2 + h->P.nlig + G.F) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dg_frozen<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)h->coef, v, h->dGb));
        }
        int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        Scope sc(h, KC_JVP, 8.0 * (2.0 * G.F + 3 + ((mode == 2 || mode == 3) ? G.F : 0)) * (double)G.nloc, alg);
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
    if (!h->ring) return op_jvp_frozen(h, v, mode, shift, out, yadd, alpha, beta);
    KStrips K = make_strips(h, true);
    if (!h->overlap || !fused_ok(h) || K.nseg < 3 || h->P.nlig > 4) {
        if ((rc = halo(h, v))) return rc;
        return op_jvp_frozen(h, v, mode, shift, out, yadd, alpha, beta);
    }
    const double nplanes = (3 + h->P.nlig) + 2.0 * G.F + ((mode == 2 || mode == 3) ? G.F : 0);   // coefficients + v + out (+ yadd)
    const double algp = 3.0 * G.F + ((mode == 2 || mode == 3) ? G.F : 0);
    const int nseg_total = K.nseg;
    HIPCHK(h, hipEventRecord(h->ev_ready, h->st));
    {
        KStrips Ki = K;
        Ki.seg0 = 1; Ki.seg_stride = 1; Ki.nseg = nseg_total - 2;
        long long nb = ((long long)Ki.nstrips * Ki.nseg + 3) / 4;
        Ki.nblocks = (int)((nb + 7) / 8 * 8);
        Scope sc(h, KC_JVP, 8.0 * nplanes * (double)G.nloc * (double)Ki.nseg / nseg_total, 8.0 * algp * (double)G.nloc * (double)Ki.nseg / nseg_total);
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
        Scope sc(h, KC_JVP, 8.0 * nplanes * (double)G.nloc * 2.0 / nseg_total, 8.0 * algp * (double)G.nloc * 2.0 / nseg_total);
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
                          TO *out, const TY *yadd, double alpha, double beta, double *normpart = nullptr)
{
    const KGeom &G = h->G;
    const double per_pt = (3.0 + h->P.nlig) * sizeof(TC) + G.F * (double)(sizeof(TV) + sizeof(TO)) + ((mode == 2 || mode == 3) ? G.F * (double)sizeof(TY) : 0.0);
    Scope sc(h, KC_JVP, per_pt * (double)G.nloc * frac, 8.0 * (3.0 * G.F + ((mode == 2 || mode == 3) ? G.F : 0)) * (double)G.nloc * frac);
    NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, TC, TV, TY, TO>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st,
                                                                     G, h->P, K, C, (const TV *)v, mode, shift, out, yadd, alpha, beta, KSmooth{}, normpart));
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

template <typename TC, typename TV, typename TY, typename TO>
static int jvp2d_halo_t(ksfd_handle *h, const TC *C, TV *v, int mode, double shift, TO *out, const TY *yadd, double alpha, double beta, double *normpart = nullptr);

// r32 = b - A x stored in fp32 (its only reader is the spectral preconditioner, which works in fp32 anyway) with ||r||^2 in
// fp64 from the store epilogue -> h->hres[0].  Single rank, strip kernels.
static int op_residual32(ksfd_handle *h, const double *x, double shift, const double *b, float *r32)
{
    if (h->G.dim == 3 && j3l_ok(h)) {
        const KGeom &G = h->G;
        K3D K = make_k3d_lds(h);
        const long long nwaves = (long long)K.nblocks * K.rows;
        if (nwaves > part_capacity()) return fail(h, KSFD_EINVAL, "op_residual32: too many waves for the fused norm");
        {
            Scope sc(h, KC_JVP, (8.0 * (3.0 * G.F + 3 + h->P.nlig) + 4.0 * G.F) * (double)G.nloc, 8.0 * 4.0 * G.F * (double)G.nloc);
            j3l_launch<float>(h, K, x, 2, shift, r32, b, 0.0, 0.0, h->part);
        }
        HIPCHK(h, hipGetLastError());
        return reduce_rows(h, 1, (int)nwaves, 0);
    }
    if (h->G.dim == 3) {
        // 3-D strip kernel: dG plane, then the z-marching Jacobian action in residual mode with fp32 output and the norm in its epilogue
        const KGeom &G = h->G;
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        {
            Scope sc(h, KC_GFIELD, 8.0 * (2 + h->P.nlig + G.F) * (double)G.plane);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dg_frozen<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)h->coef, x, h->dGb));
        }
        K3D K = make_k3d(h);
        const long long nwaves = (long long)K.nblocks * K.rows;
        if (nwaves > part_capacity()) return fail(h, KSFD_EINVAL, "op_residual32: too many waves for the fused norm");
        {
            Scope sc(h, KC_JVP, (8.0 * (2.0 * G.F + 3) + 4.0 * G.F) * (double)G.nloc, 8.0 * 4.0 * G.F * (double)G.nloc);
            if (K.rows == 8) hipLaunchKernelGGL((k_jvp3d_frozen<1, float, 8>), dim3(K.nblocks), dim3(8 * KSFD_WAVE), 0, h->st, G, h->P, K, (const double *)h->coef, x, (const double *)h->dGb, 2, shift, r32, b, 0.0, 0.0, h->part);
            else NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp3d_frozen<NL, float>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, h->P, K, (const double *)h->coef, x, (const double *)h->dGb, 2, shift, r32, b, 0.0, 0.0, h->part));
        }
        HIPCHK(h, hipGetLastError());
        return reduce_rows(h, 1, (int)nwaves, 0);
    }
    KStrips K = make_strips(h, true);
    const long long nwaves = (long long)K.nstrips * K.nseg;
    if (nwaves > part_capacity()) return fail(h, KSFD_EINVAL, "op_residual32: too many waves for the fused norm");
    // slab ranks: the ghost rows of x travel while the interior segments are computed (the caller has NOT exchanged them)
    int rc = h->ring ? jvp2d_halo_t<double, double, double, float>(h, (const double *)h->coef, const_cast<double *>(x), 2, shift, r32, b, 0.0, 0.0, h->part)
                         : jvp2d_launch_t<double, double, double, float>(h, K, 1.0, (const double *)h->coef, x, 2, shift, r32, b, 0.0, 0.0, h->part);
    if (rc) return rc;
    return reduce_rows(h, 1, (int)nwaves, 0);
}

template <typename TC, typename TV, typename TY, typename TO>
// normpart != NULL: per-wave partials of ||out||^2, numbered over ALL segments (interior launch first, then the two boundary segments)
static int jvp2d_halo_t(ksfd_handle *h, const TC *C, TV *v, int mode, double shift, TO *out, const TY *yadd, double alpha, double beta, double *normpart)
{
    const KGeom &G = h->G;
    KStrips K = make_strips(h, true);
    if (!h->ring) return jvp2d_launch_t(h, K, 1.0, C, v, mode, shift, out, yadd, alpha, beta, normpart);
    const long long scale = sizeof(double) / sizeof(TV);            // 1 for double, 2 for float
    const bool ovl = h->overlap && K.nseg >= 3;
    int rc;
    if (ovl) {
        HIPCHK(h, hipEventRecord(h->ev_ready, h->st));
        KStrips Ki = K;
        Ki.seg0 = 1; Ki.seg_stride = 1; Ki.nseg = K.nseg - 2;
        long long nb = ((long long)Ki.nstrips * Ki.nseg + 3) / 4;
        Ki.nblocks = (int)((nb + 7) / 8 * 8);
        if ((rc = jvp2d_launch_t(h, Ki, (double)Ki.nseg / K.nseg, C, v, mode, shift, out, yadd, alpha, beta, normpart))) return rc;
        HIPCHK(h, hipStreamWaitEvent(h->st_comm, h->ev_ready, 0));
    }
    {
        Scope sc(h, KC_HALO, 4.0 * 2.0 * sizeof(TV) * G.F * (double)G.inner * 2.0);
        if (h->tr->exchange(reinterpret_cast<double *>(v), G.F, G.plane / scale, G.inner / scale, G.sloc, G.ng, ovl ? h->st_comm : h->st))
            return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
    }
    if (!ovl) return jvp2d_launch_t(h, K, 1.0, C, v, mode, shift, out, yadd, alpha, beta, normpart);
    HIPCHK(h, hipEventRecord(h->ev_halo, h->st_comm));
    HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_halo, 0));
    KStrips Kb = K;
    Kb.seg0 = 0; Kb.seg_stride = K.nseg - 1; Kb.nseg = 2;
    long long nb = ((long long)Kb.nstrips * Kb.nseg + 3) / 4;
    Kb.nblocks = (int)((nb + 7) / 8 * 8);
    return jvp2d_launch_t(h, Kb, 2.0 / K.nseg, C, v, mode, shift, out, yadd, alpha, beta, normpart ? normpart + (long long)K.nstrips * (K.nseg - 2) : nullptr);
}

// VW = 2 when every plane/offset/length is even (all accesses 16-byte aligned double2)
static inline bool vec2(const ksfd_handle *h) { return (h->G.nloc % 2 == 0) && (h->kv.off % 2 == 0) && (h->G.plane % 2 == 0); }
static inline dim3 vgridw(const ksfd_handle *h, int vw) { return dim3((h->nblk_vec + vw - 1) / vw, h->G.F); }
#define VW_DISPATCH(h, CALL) do { if (vec2(h)) { constexpr int VW = 2; CALL; } else { constexpr int VW = 1; CALL; } } while (0)

// want_norm: ||out||^2 lands in h->hres[0] (one reduction instead of a pass of its own)
static int op_lincomb(ksfd_handle *h, int nt, const double *const *x, const double *a, double *out, bool want_norm = false)
{
    double *part = want_norm ? h->part : nullptr;
    KLin L;
    for (int t = 0; t < 6; t++) { L.x[t] = t < nt ? x[t] : nullptr; L.a[t] = t < nt ? a[t] : 0.0; }
    Scope sc(h, KC_LINCOMB, vbytes(h, nt + 1));
    switch (nt) {
    case 1: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<1, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out, part)); break;
    case 2: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<2, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out, part)); break;
    case 3: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<3, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out, part)); break;
    case 4: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<4, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out, part)); break;
    case 5: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<5, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out, part)); break;
    default: VW_DISPATCH(h, hipLaunchKernelGGL((k_lincomb<6, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, L, out, part)); break;
    }
    HIPCHK(h, hipGetLastError());
    if (want_norm) { const dim3 gr = vgridw(h, vec2(h) ? 2 : 1); return reduce_rows(h, 1, (int)(gr.x * gr.y), 0); }
    return KSFD_OK;
}

// d[0..k) = <w,V_i>, d[k] = <w,w>  -> h->hres
static int op_multidot(ksfd_handle *h, const double *w, const double *V, int k)
{
    if (k > 32) {
        // more basis vectors than one launch takes (restart lengths beyond 32: gmres() grows the restart when a cycle stagnates): chunks
        // of 32, each with its own reduction; <w,w> from the last one
        std::vector<double> acc((size_t)k + 1);
        for (int c0 = 0; c0 < k; c0 += 32) {
            const int kc = std::min(32, k - c0);
            int rc = op_multidot(h, w, V + (int64_t)c0 * h->vlen, kc);
            if (rc) return rc;
            for (int i = 0; i < kc; i++) acc[c0 + i] = h->hres[i];
            acc[k] = h->hres[kc];
        }
        if (k + 1 > 128) return fail(h, KSFD_EINVAL, "op_multidot: %d vectors exceed the result buffer", k);
        for (int i = 0; i <= k; i++) h->hres[i] = acc[i];
        return KSFD_OK;
    }
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
    if (k > 32) {                        // chunks of 32 basis vectors; the scaling rides in the last one
        for (int c0 = 0; c0 < k; c0 += 32) {
            const int kc = std::min(32, k - c0);
            int rc = op_gs_update(h, w, V + (int64_t)c0 * h->vlen, kc, coef + c0, c0 + kc == k ? scale : 1.0);
            if (rc) return rc;
        }
        return KSFD_OK;
    }
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

static int op_basis_axpy(ksfd_handle *h, double *x, const double *V, int k, const double *coef, double beta, bool want_norm = false)
{
    if (k > 32) {                        // chunks of 32 basis vectors: beta applies to the first, the norm comes with the last
        for (int c0 = 0; c0 < k; c0 += 32) {
            const int kc = std::min(32, k - c0);
            int rc = op_basis_axpy(h, x, V + (int64_t)c0 * h->vlen, kc, coef + c0, c0 == 0 ? beta : 1.0, want_norm && c0 + kc == k);
            if (rc) return rc;
        }
        return KSFD_OK;
    }
    double *part = want_norm ? h->part : nullptr;
    KCoef C;
    int nread = 0;                                      // vectors with a zero coefficient are not loaded
    for (int i = 0; i < KSFD_MAXDOT; i++) { C.h[i] = i < k ? coef[i] : 0.0; nread += C.h[i] != 0.0; }
    Scope sc(h, KC_BASISAXPY, vbytes(h, nread + 1 + (beta != 0.0)));
    if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<4, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta, part));
    else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<8, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta, part));
    else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<16, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta, part));
    else VW_DISPATCH(h, hipLaunchKernelGGL((k_basis_axpy<32, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, x, V, h->vlen, k, C, beta, part));
    HIPCHK(h, hipGetLastError());
    if (want_norm) { const dim3 gr = vgridw(h, vec2(h) ? 2 : 1); return reduce_rows(h, 1, (int)(gr.x * gr.y), 0); }
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
