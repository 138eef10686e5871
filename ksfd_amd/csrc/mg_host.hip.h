// libksfd_hip.so -- host side of the geometric multigrid preconditioner (kernels and rationale: mg.hip.h)
// (part of the single translation unit ksfd_hip.hip; included from there in this order:
//  handle.hip.h, ops.hip.h, mg_host.hip.h, krylov.hip.h)
#pragma once
// ------------------------------------------------------------------------------------------------
// multigrid preconditioner (host side; kernels and rationale in mg.hip.h)
// ------------------------------------------------------------------------------------------------
static void mg_free(ksfd_handle *h)
{
    if (h->mg_graph) { hipGraphExecDestroy(h->mg_graph); h->mg_graph = nullptr; }
    for (size_t l = 0; l < h->mg.size(); l++) {
        MGLevel &L = h->mg[l];
        if (L.dinv) hipFree(L.dinv);
        double *bufs[] = { l ? L.coef : nullptr, l ? L.x : nullptr, l ? L.b : nullptr, L.r, L.d, L.Ad, L.dG, L.pv };
        for (double *b : bufs) if (b) hipFree(b);
        float *fb[] = { L.x32, L.b32, L.r32, L.d32, L.coef32 };
        for (float *b : fb) if (b) hipFree(b);
    }
    h->mg.clear();
    h->mg_ok = false;
}

static int mg_build(ksfd_handle *h)
{
    const int dim = h->G.dim;
    int nl = h->P.nlig, F = h->G.F;
    long long nx = h->G.nx, ny = dim == 3 ? h->G.ny : 1, rows = h->G.sloc;   // rows = local slow units (y rows in 2-D, z planes in 3-D)
    // NOTE: every decision below must be identical on all ranks (the levels exchange halos): use sloc, never slow0.
    // Slab r starts at unit r*sloc; it stays on the coarse grid of level l as long as sloc is divisible by 2^l.
    KPhys P = h->P;
    for (int l = 0;; l++) {
        MGLevel L;
        L.G = h->G; L.G.nx = nx;
        if (dim == 1) { L.G.nx = rows; L.G.inner = 1; }                      // 1-D: the slab axis is x itself
        else if (dim == 2) { L.G.ny = rows; L.G.inner = nx; } else { L.G.ny = ny; L.G.nz = rows; L.G.inner = nx * ny; }
        L.G.sloc = rows;
        L.G.plane = (rows + 2 * L.G.ng) * L.G.inner; L.G.nloc = rows * L.G.inner;
        L.P = P;
        L.kv.plane = L.G.plane; L.kv.off = (long long)L.G.ng * L.G.inner; L.kv.nloc = L.G.nloc; L.kv.nf = F;
        L.vlen = (int64_t)F * L.G.plane;
        L.nblk = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 2048);
        if (l == 0) L.coef = h->coef;
        else if (alloc_d(h, &L.coef, (int64_t)(3 + nl) * L.G.plane) || alloc_d(h, &L.x, L.vlen) || alloc_d(h, &L.b, L.vlen)) return KSFD_ENOMEM;
        if (hipMalloc((void **)&L.dinv, sizeof(float) * (size_t)F * F * L.G.plane) != hipSuccess || alloc_d(h, &L.r, L.vlen) || alloc_d(h, &L.d, L.vlen) ||
            alloc_d(h, &L.Ad, L.vlen) || alloc_d(h, &L.dG, L.G.plane) || alloc_d(h, &L.pv, L.vlen)) return KSFD_ENOMEM;
        double *zero[] = { l ? L.x : nullptr, l ? L.b : nullptr, L.r, L.d, L.Ad, L.pv };
        for (double *z : zero) if (z) hipMemsetAsync(z, 0, sizeof(double) * (size_t)L.vlen, h->st);
        h->mg.push_back(L);
        // next level: every rank keeps >= 4 slow units (ghost width 2 + the 4th-order star), global grid >= 8 per axis
        const long long rows_glob = rows * h->size;
        if ((dim > 1 && ((nx % 2) || nx / 2 < 8)) || (rows % 2) || rows_glob / 2 < 8 || (h->ring && rows / 2 < 4)) break;
        if (dim == 3 && ((ny % 2) || ny / 2 < 8)) break;
        nx /= 2; rows /= 2;
        if (dim == 3) ny /= 2;
        for (int a = 0; a < 3; a++) { P.inv_h[a] *= 0.5; P.inv_h2[a] *= 0.25; }
    }
    h->mg_ok = h->mg.size() >= 2;
    if (h->ring) h->mg_use_graph = false;           // collectives inside the cycle: keep eager launches
    // fp32 level vectors (mg_vcycle32): 2-D, levels the strip kernel serves, never the coarsest one (its many Chebyshev sweeps
    // stay in fp64 with the kernels they have)
    if (h->mg_ok && dim == 2 && h->use_fused && nl <= 4) {
        for (size_t l = 0; l + 1 < h->mg.size(); l++) {
            MGLevel &L = h->mg[l];
            if ((L.G.nx % 2) || L.G.nx < 16) break;
            const size_t nb = sizeof(float) * (size_t)L.vlen;
            if (hipMalloc((void **)&L.x32, nb) != hipSuccess || hipMalloc((void **)&L.b32, nb) != hipSuccess ||
                hipMalloc((void **)&L.r32, nb) != hipSuccess || hipMalloc((void **)&L.d32, nb) != hipSuccess) { (void)hipGetLastError(); break; }
            float *zero[] = { L.x32, L.b32, L.r32, L.d32 };
            for (float *z : zero) hipMemsetAsync(z, 0, nb, h->st);
            if (l > 0 && hipMalloc((void **)&L.coef32, sizeof(float) * (size_t)(3 + nl) * L.G.plane) != hipSuccess) { (void)hipGetLastError(); L.coef32 = nullptr; }
            L.f32 = true;
        }
    }
    return KSFD_OK;
}

// transfer operators, 2-D or 3-D by the level geometry
static void mg_launch_restrict(ksfd_handle *h, MGLevel &Lf, MGLevel &Lc, int np, const double *fine, double *coarse)
{
    int nb = (int)std::min<long long>((Lc.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    if (Lf.G.dim == 1)
        hipLaunchKernelGGL(k_restrict1d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.sloc, Lf.G.wrap_slow,
                           fine, Lf.G.plane, Lf.kv.off, coarse, Lc.G.plane, Lc.kv.off);
    else if (Lf.G.dim == 3)
        hipLaunchKernelGGL(k_restrict3d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.ny, Lf.G.sloc, Lf.G.wrap_slow,
                           fine, Lf.G.plane, Lf.kv.off, coarse, Lc.G.plane, Lc.kv.off);
    else
        hipLaunchKernelGGL((k_restrict2d<double, double>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.sloc, Lf.G.wrap_slow,
                           fine, Lf.G.plane, Lf.kv.off, coarse, Lc.G.plane, Lc.kv.off);
}
static void mg_launch_prolong(ksfd_handle *h, MGLevel &Lf, MGLevel &Lc, int np, const double *coarse, double *fine)
{
    int nb = (int)std::min<long long>((Lf.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    if (Lf.G.dim == 1)
        hipLaunchKernelGGL(k_prolong_add1d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.sloc, Lf.G.wrap_slow,
                           coarse, Lc.G.plane, Lc.kv.off, fine, Lf.G.plane, Lf.kv.off);
    else if (Lf.G.dim == 3)
        hipLaunchKernelGGL(k_prolong_add3d, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.ny, Lf.G.sloc, Lf.G.wrap_slow,
                           coarse, Lc.G.plane, Lc.kv.off, fine, Lf.G.plane, Lf.kv.off);
    else
        hipLaunchKernelGGL((k_prolong_add2d<double, double>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.sloc, Lf.G.wrap_slow,
                           coarse, Lc.G.plane, Lc.kv.off, fine, Lf.G.plane, Lf.kv.off);
}

// ghost rows of a level vector (np field planes) from the ring neighbours
static int mg_halo(ksfd_handle *h, MGLevel &L, double *v, int np)
{
    if (!h->ring) return KSFD_OK;
    Scope sc(h, KC_HALO, 4.0 * 8.0 * np * (double)L.G.inner * 2.0);
    if (h->tr->exchange(v, np, L.G.plane, L.G.inner, L.G.sloc, L.G.ng, h->st)) return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
    return KSFD_OK;
}

// out = J v | shift v - J v | yadd - (shift v - J v) on level L
// sm != NULL: modes 5 / 6, smoother algebra in the epilogue (2-D strip kernel and generic kernel only: see mg_can_fuse)
// ... of an fp32 level vector: it travels through the double-typed transport as half as many doubles (nx is even on these levels)
static int mg_halo32(ksfd_handle *h, MGLevel &L, float *v, int np)
{
    if (!h->ring) return KSFD_OK;
    Scope sc(h, KC_HALO, 4.0 * 4.0 * np * (double)L.G.inner * 2.0);
    if (h->tr->exchange(reinterpret_cast<double *>(v), np, L.G.plane / 2, L.G.inner / 2, L.G.sloc, L.G.ng, h->st)) return fail(h, KSFD_ECOMM, "halo exchange failed: %s", h->tr->error().c_str());
    return KSFD_OK;
}

static bool mg_can_fuse(const ksfd_handle *h, const MGLevel &L)
{
    const KGeom &G = L.G;
    const bool strip3d = G.dim == 3 && h->use_fused && (G.nx % 2 == 0) && G.nx >= 16 && h->P.nlig <= 4;
    return h->mg_fuse && !strip3d;
}

static int mg_op(ksfd_handle *h, MGLevel &L, const double *v, int mode, double shift, double *out, const double *yadd,
                 const KSmooth *sm = nullptr)
{
    const KGeom &G = L.G;
    if (h->ring) { int rch = mg_halo(h, L, const_cast<double *>(v), G.F); if (rch) return rch; }
    const int cls = (&L == &h->mg[0]) ? KC_JVP : KC_MG;
    // planes moved: coefficients + v, plus per mode: 1/2: out (+ yadd); 5: yadd, Dinv, r, d; 6: Dinv, rr, x in and out
    const double by = 8.0 * ((3 + h->P.nlig) + G.F + (mode == 5 ? 3.0 * G.F + 0.5 * G.F * G.F : mode == 6 ? 3.0 * G.F + 0.5 * G.F * G.F : G.F + (mode == 2 ? G.F : 0))) * (double)G.nloc;
    const KSmooth S = sm ? *sm : KSmooth{};
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
        // level 0 reads the fp32 copy of the coefficient planes when there is one (the V cycle is a preconditioner: see poly_apply)
        const float *c32 = (&L == &h->mg[0] && h->poly_fp32) ? h->coef32 : nullptr;
        Scope sc(h, cls, by - (c32 ? 4.0 * (3 + h->P.nlig) * (double)G.nloc : 0.0));
        if (sm && c32) {
            NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, float, double, double, double, 1, true>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, c32, v, mode, shift, out, yadd, 0.0, 0.0, S));
        } else if (sm) {
            NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, double, double, double, double, 1, true>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, (const double *)L.coef, v, mode, shift, out, yadd, 0.0, 0.0, S));
        } else if (c32) {
            NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, float, double, double, double>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, c32, v, mode, shift, out, yadd));
        } else
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, (const double *)L.coef, v, mode, shift, out, yadd));
    } else if (G.dim == 3 && h->use_fused && (G.nx % 2 == 0) && G.nx >= 16 && h->P.nlig <= 4) {
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        K3D K;
        K.rows = 4; K.sync = 0;
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
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jvp_generic<NL>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, L.P, (const double *)L.coef, v, (const double *)(L.coef + G.plane), (const double *)L.dG, mode, shift, out, yadd, 0.0, 0.0, S));
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
        if (Lc.coef32) {
            // the fp32 cycle reads an fp32 copy (the same full weighting of the fine fp64 planes, rounded once)
            int nb = (int)std::min<long long>((Lc.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
            {
                Scope sc(h, KC_MG, np * (8.0 * Lf.G.nloc + 4.0 * Lc.G.nloc));
                hipLaunchKernelGGL((k_restrict2d<double, float>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, np, Lf.G.nx, Lf.G.sloc, Lf.G.wrap_slow,
                                   (const double *)Lf.coef, Lf.G.plane, Lf.kv.off, Lc.coef32, Lc.G.plane, Lc.kv.off);
            }
            if ((rc = mg_halo32(h, Lc, Lc.coef32, np))) return rc;
        }
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
        // power iteration on Dinv*A: v and w = Dinv A v / |v| alternate between L.pv and L.r, A v in L.Ad.  The vector is kept
        // from one set-up to the next (L.pv): the shift and the frozen state move a little from step to step and the dominant
        // vector with them, so a warm start needs 2-3 iterations where the cold one from a hash fill takes mg_power_its
        // (4096^2 x 3 fields: 9.4 -> 2.9 ms of set-up per step).
        double *v = L.pv, *w = L.r;
        double nv = L.pv_norm, lam = 2.0, lam_prev = 0.0;
        const bool warm = nv > 0.0 && h->mg_warm_power;
        if (!warm) {
            hipLaunchKernelGGL(k_hash_fill, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, (long long)L.vlen, v);
            if ((rc = mg_norm(h, L, v, &nv))) return rc;
        }
        const int its = warm ? std::min(h->mg_power_its, 3) : h->mg_power_its;
        for (int it = 0; it < its; it++) {
            if (!(nv > 0.0) || nv != nv) break;
            if ((rc = mg_op(h, L, v, 1, shift, L.Ad, nullptr))) return rc;
            {
                Scope sc(h, KC_MG, 8.0 * (2 * F + 0.5 * F * F) * L.G.nloc);
                NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const float *)(L.dinv + L.kv.off), (const double *)(L.Ad + L.kv.off), 1.0 / nv, w + L.kv.off));
            }
            double nw;
            if ((rc = mg_norm(h, L, w, &nw))) return rc;
            if (!(nw > 0.0)) break;
            lam_prev = lam; lam = nw;                              // |Dinv A v| / |v|
            std::swap(v, w); nv = nw;
            if (warm && it >= 1 && fabs(lam - lam_prev) <= 0.01 * lam) break;
        }
        if (v != L.pv) HIPCHK(h, hipMemcpyAsync(L.pv, v, sizeof(double) * (size_t)L.vlen, hipMemcpyDeviceToDevice, h->st));
        L.pv_norm = (nv > 0.0 && nv == nv) ? nv : 0.0;
        L.lam_max = 1.15 * lam;
        if (l + 1 == h->mg.size()) {
            int nbr = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 256);
            hipLaunchKernelGGL(k_ratio_est, dim3(nbr), dim3(KSFD_BLOCK), 0, h->st, (long long)L.G.nloc, (const float *)(L.dinv + L.kv.off), shift, h->part);
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
    if (nu == 2 && mg_can_fuse(h, L)) {
        // the default V(2,2) sweeps with the smoother algebra in the Jacobian-action epilogues: 2 launches instead of 3 (zero guess)
        // or 4 (correction), and the residual / A d round trips through memory disappear (modes 5 and 6, KSmooth)
        const double rho0 = 1.0 / sig1, rhon = 1.0 / (2.0 * sig1 - rho0);
        KSmooth S = KSmooth{};
        S.dinv = L.dinv; S.x = x; S.c1 = rhon * rho0; S.c2 = 2.0 * rhon / delta;
        if (zero_init) {
            {
                Scope sc(h, KC_MG, 8.0 * (2 * F + 0.5 * F * F) * L.G.nloc);
                NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const float *)(L.dinv + off), b + off, 1.0 / theta, L.d + off));
            }
            S.rr = b; S.x_has_d = 1;
            return mg_op(h, L, L.d, 6, shift, nullptr, nullptr, &S);
        }
        S.out2 = L.d; S.scale = 1.0 / theta;
        if ((rc = mg_op(h, L, x, 5, shift, L.r, b, &S))) return rc;
        S.rr = L.r; S.x_has_d = 0;
        return mg_op(h, L, L.d, 6, shift, nullptr, nullptr, &S);
    }
    const double *res = b;
    if (!zero_init) {
        if ((rc = mg_op(h, L, x, 2, shift, L.r, b))) return rc;        // r = b - A x
        res = L.r;
    }
    {
        Scope sc(h, KC_MG, 8.0 * ((zero_init ? 3 : 2) * F + 0.5 * F * F) * L.G.nloc);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const float *)(L.dinv + off), res + off, 1.0 / theta, L.d + off, zero_init ? x + off : (double *)nullptr));
    }
    bool x_has_d = zero_init;          // x == d_0 already
    double rho = 1.0 / sig1;
    for (int k = 1; k < nu; k++) {
        if ((rc = mg_op(h, L, L.d, 1, shift, L.Ad, nullptr))) return rc;
        const double rhon = 1.0 / (2.0 * sig1 - rho);
        const double *rsrc = (zero_init && k == 1) ? b : L.r;             // first sweep from a zero guess: r_0 = b, never copied
        if (k == nu - 1) {
            Scope sc(h, KC_MG, 8.0 * (5 * F + 0.5 * F * F) * L.G.nloc);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_cheb_last<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const float *)(L.dinv + off), x + off, rsrc + off, (const double *)(L.d + off), (const double *)(L.Ad + off), rhon * rho, 2.0 * rhon / delta, x_has_d ? 1 : 0));
            x_has_d = true;
        } else {
            if (rsrc != L.r) HIPCHK(h, hipMemcpyAsync(L.r, b, sizeof(double) * (size_t)L.vlen, hipMemcpyDeviceToDevice, h->st));
            if (x_has_d && k == 1) { /* x already holds d_0: the step kernel adds d to x, so undo by starting x at 0 */
                HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)L.vlen, h->st));
            }
            Scope sc(h, KC_MG, 8.0 * (7 * F + 0.5 * F * F) * L.G.nloc);
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_cheb_step<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const float *)(L.dinv + off), x + off, L.r + off, L.d + off, (const double *)(L.Ad + off), rhon * rho, 2.0 * rhon / delta));
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

// everything below level 0 touches only fixed buffers: capture it once per shift into a hipGraph and replay it (a V cycle has ~15
// launches per level; on small grids they are pure launch latency).  body = the coarse-grid correction of level 0 in the precision in use.
template <typename Body>
static int mg_coarse_graph(ksfd_handle *h, double shift, const void *xkey, bool f32, Body body)
{
    int rc;
    if (!h->mg_graph || h->mg_graph_shift != shift || h->mg_graph_x != xkey || h->mg_graph_f32 != f32) {
        if (h->mg_graph) { hipGraphExecDestroy(h->mg_graph); h->mg_graph = nullptr; }
        hipGraph_t g = nullptr;
        const double b0 = h->bytes_acc;
        HIPCHK(h, hipStreamBeginCapture(h->st, hipStreamCaptureModeThreadLocal));
        h->capturing = true;
        rc = body();
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
        h->mg_graph_x = xkey;
        h->mg_graph_f32 = f32;
    }
    Scope sc(h, KC_MG, h->mg_graph_bytes);
    HIPCHK(h, hipGraphLaunch(h->mg_graph, h->st));
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
        if ((rc = mg_coarse_graph(h, shift, x, false, [&]() { return mg_coarse_correction(h, 0, shift, x); }))) return rc;
    } else if ((rc = mg_coarse_correction(h, l, shift, x))) return rc;
    return mg_smooth(h, L, shift, b, x, h->mg_nu, false, h->mg_ratio);
}

// ------------------------------------------------------------------------------------------------
// The same V(2,2) cycle with fp32 LEVEL VECTORS (x, b, r, d of every f32 level; the arithmetic inside the kernels stays fp64).
// A V cycle is a preconditioner: GMRES sees the true fp64 residual of the real system whatever the cycle returns, and the cycle is
// bandwidth-bound -- at 4096^2 x 3 fields an iteration moves ~82 planes of 134 MB through level 0 alone, two thirds of them level
// vectors.  Used when the step's ksp_rtol >= 1e-7 (ksfd_step; the parity tests at 1e-11 keep the fp64 cycle), 2-D, one rank or slab
// ranks (a float plane travels through the double-typed transport as half as many doubles), V(2,2) with the fused smoother.  The fp64 right-hand side is read once (k_dinv_apply leaves its fp32 copy), the last smoothing
// kernel writes the result in fp64 (KSmoothT::x64); levels below the last f32 one run the fp64 code above, the transfer kernels
// convert at that border.
// ------------------------------------------------------------------------------------------------
static int mg_op32(ksfd_handle *h, MGLevel &L, const float *v, int mode, double shift, float *out, const float *yadd, const KSmoothT<float> *sm)
{
    const KGeom &G = L.G;
    if (h->ring) { int rch = mg_halo32(h, L, const_cast<float *>(v), G.F); if (rch) return rch; }
    const int cls = (&L == &h->mg[0]) ? KC_JVP : KC_MG;
    const float *c32 = (&L == &h->mg[0]) ? (h->poly_fp32 ? h->coef32 : nullptr) : L.coef32;
    // planes moved (in units of 8 B per point): coefficients, v, per mode: 2: yadd + out; 5: yadd, Dinv, r, d; 6: Dinv, rr, x in and out
    const double by = ((c32 ? 4.0 : 8.0) * (3 + h->P.nlig) + 4.0 * G.F + (mode == 2 ? 8.0 * G.F : 12.0 * G.F + 4.0 * G.F * G.F + ((sm && sm->x64) ? 4.0 * G.F : 0.0))) * (double)G.nloc;
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
    const KSmoothT<float> S = sm ? *sm : KSmoothT<float>{};
    Scope sc(h, cls, by);
    if (sm && c32) {
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, float, float, float, float, 1, true, float>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, c32, v, mode, shift, out, yadd, 0.0, 0.0, S));
    } else if (sm) {
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, double, float, float, float, 1, true, float>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, (const double *)L.coef, v, mode, shift, out, yadd, 0.0, 0.0, S));
    } else if (c32) {
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, float, float, float, float>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, c32, v, mode, shift, out, yadd));
    } else {
        NL_DISPATCH(h->P.nlig, if constexpr (NL <= 4) hipLaunchKernelGGL((k_jvp2d_frozen<NL, double, float, float, float>), dim3(K.nblocks), dim3(KSFD_BLOCK), 0, h->st, G, L.P, K, (const double *)L.coef, v, mode, shift, out, yadd));
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

static int mg_vcycle32(ksfd_handle *h, size_t l, double shift, const double *b64, double *x64);

// coarse-grid correction of an f32 level: restrict r32, recurse in the precision of the next level, prolong-add into x32
static int mg_coarse_correction32(ksfd_handle *h, size_t l, double shift)
{
    int rc;
    MGLevel &L = h->mg[l], &Lc = h->mg[l + 1];
    const int F = L.G.F;
    const int nbr = (int)std::min<long long>((Lc.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    const int nbp = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    if ((rc = mg_halo32(h, L, L.r32, F))) return rc;                    // restriction reads fine rows -1 and sloc
    {
        Scope sc(h, KC_MG, F * (4.0 * L.G.nloc + (Lc.f32 ? 4.0 : 8.0) * Lc.G.nloc));
        if (Lc.f32) hipLaunchKernelGGL((k_restrict2d<float, float>), dim3(nbr), dim3(KSFD_BLOCK), 0, h->st, F, L.G.nx, L.G.sloc, L.G.wrap_slow, (const float *)L.r32, L.G.plane, L.kv.off, Lc.b32, Lc.G.plane, Lc.kv.off);
        else hipLaunchKernelGGL((k_restrict2d<float, double>), dim3(nbr), dim3(KSFD_BLOCK), 0, h->st, F, L.G.nx, L.G.sloc, L.G.wrap_slow, (const float *)L.r32, L.G.plane, L.kv.off, Lc.b, Lc.G.plane, Lc.kv.off);
    }
    if (Lc.f32) rc = mg_vcycle32(h, l + 1, shift, nullptr, nullptr);
    else rc = mg_vcycle(h, l + 1, shift, Lc.b, Lc.x);
    if (rc) return rc;
    if ((rc = Lc.f32 ? mg_halo32(h, Lc, Lc.x32, F) : mg_halo(h, Lc, Lc.x, F))) return rc;      // prolongation reads coarse row sloc_c
    {
        Scope sc(h, KC_MG, F * (8.0 * L.G.nloc + (Lc.f32 ? 4.0 : 8.0) * Lc.G.nloc));
        if (Lc.f32) hipLaunchKernelGGL((k_prolong_add2d<float, float>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, F, L.G.nx, L.G.sloc, L.G.wrap_slow, (const float *)Lc.x32, Lc.G.plane, Lc.kv.off, L.x32, L.G.plane, L.kv.off);
        else hipLaunchKernelGGL((k_prolong_add2d<double, float>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, F, L.G.nx, L.G.sloc, L.G.wrap_slow, (const double *)Lc.x, Lc.G.plane, Lc.kv.off, L.x32, L.G.plane, L.kv.off);
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

// level l of the fp32 cycle.  l = 0: right-hand side b64 and result x64 in fp64 (the caller's vectors); l > 0: L.b32 -> L.x32.
static int mg_vcycle32(ksfd_handle *h, size_t l, double shift, const double *b64, double *x64)
{
    int rc;
    MGLevel &L = h->mg[l];
    const int F = L.G.F;
    const int nb = (int)std::min<long long>((L.G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
    const double lmax = L.lam_max, lmin = lmax / h->mg_ratio;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sig1 = theta / delta;
    const double rho0 = 1.0 / sig1, rhon = 1.0 / (2.0 * sig1 - rho0);
    const long long off = L.kv.off;
    KSmoothT<float> S = KSmoothT<float>{};
    S.dinv = L.dinv; S.x = L.x32; S.c1 = rhon * rho0; S.c2 = 2.0 * rhon / delta;
    // pre-smoothing from a zero guess: d0 = Dinv b / theta (+ the fp32 copy of b on level 0); then x = d0 + d1 in the epilogue of A d0
    {
        Scope sc(h, KC_MG, ((b64 ? 8.0 + 4.0 : 4.0) * F + 4.0 * F + 4.0 * F * F) * (double)L.G.nloc);
        if (b64) {
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL, double, float>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const float *)(L.dinv + off), b64 + off, 1.0 / theta, L.d32 + off, (float *)nullptr, L.b32 + off));
        } else {
            NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_dinv_apply<NL, float, float>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, L.G.nloc, L.G.plane, (const float *)(L.dinv + off), (const float *)(L.b32 + off), 1.0 / theta, L.d32 + off));
        }
    }
    S.rr = L.b32; S.x_has_d = 1;
    if ((rc = mg_op32(h, L, L.d32, 6, shift, nullptr, nullptr, &S))) return rc;
    if ((rc = mg_op32(h, L, L.x32, 2, shift, L.r32, L.b32, nullptr))) return rc;
    if (l == 0 && h->mg_use_graph && !h->capturing) {
        if ((rc = mg_coarse_graph(h, shift, L.x32, true, [&]() { return mg_coarse_correction32(h, 0, shift); }))) return rc;
    } else if ((rc = mg_coarse_correction32(h, l, shift))) return rc;
    // post-smoothing: r = b - A x and d0 = Dinv r / theta in one launch, x += d0 + d1 in the next (level 0: into the caller's fp64 vector)
    S.out2 = L.d32; S.scale = 1.0 / theta;
    if ((rc = mg_op32(h, L, L.x32, 5, shift, L.r32, L.b32, &S))) return rc;
    S.rr = L.r32; S.x_has_d = 0; S.x64 = x64;
    return mg_op32(h, L, L.d32, 6, shift, nullptr, nullptr, &S);
}

// out = M^-1 in  (one V cycle)
static int mg_precond(ksfd_handle *h, double shift, const double *in, double *out)
{
    int rc;
    if (!h->mg_coef_valid && (rc = mg_restrict_coefs(h))) return rc;
    if (h->mg_shift != shift && (rc = mg_setup_shift(h, shift))) return rc;
    if (h->mg_use32 && h->mg_fp32 && h->mg[0].f32 && h->mg_nu == 2 && mg_can_fuse(h, h->mg[0])) return mg_vcycle32(h, 0, shift, in, out);
    return mg_vcycle(h, 0, shift, in, out);
}
