// libksfd_hip.so -- host side of the spectral preconditioner (kernels and rationale: spectral.hip.h)
// (part of the single translation unit ksfd_hip.hip; included after ops.hip.h)
#pragma once

static bool spec_plan(long long n, KFFTPlan &P)
{
    if (n < 32 || n > 16384 || (n & (n - 1))) return false;
    int lg = 0;
    while ((1LL << lg) < n) lg++;
    P.n = (int)n; P.lg = lg; P.nstage = 0;
    int left = lg;
    static const int lgmax = getenv("KSFD_SPEC_RADIX") ? (atoi(getenv("KSFD_SPEC_RADIX")) == 4 ? 2 : (atoi(getenv("KSFD_SPEC_RADIX")) == 8 ? 3 : 4)) : 4;   // experiment knob
    while (left >= lgmax && P.nstage < KSPEC_MAXSTAGE) { P.radix[P.nstage++] = 1 << lgmax; left -= lgmax; }
    if (left && P.nstage < KSPEC_MAXSTAGE) { P.radix[P.nstage++] = 1 << left; left = 0; }
    return left == 0;
}

// position of frequency k in the output of the DIF stages (see spectral.hip.h): pos = q0*(n/r0) + pos'(k / r0), q0 = k % r0
static int spec_pos(const KFFTPlan &P, int k)
{
    int pos = 0, n = P.n;
    for (int s = 0; s < P.nstage; s++) {
        const int r = P.radix[s];
        pos += (k % r) * (n / r);
        k /= r;
        n /= r;
    }
    return pos;
}

static void spec_free(ksfd_handle *h)
{
    SpecState &S = h->spec;
    void *bufs[] = { S.W, S.twx, S.twy, S.posx, S.posy, S.kyofpos, S.lx, S.ly };
    for (void *b : bufs) if (b) hipFree(b);
    S = SpecState();
}

template <typename T> static bool spec_upload(T **dev, const std::vector<T> &host)
{
    return hipMalloc((void **)dev, sizeof(T) * host.size()) == hipSuccess &&
           hipMemcpy(*dev, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice) == hipSuccess;
}

// Builds plans, tables and the work array if this handle can use the spectral preconditioner; leaves spec.ok = false otherwise.
static void spec_build(ksfd_handle *h)
{
    SpecState &S = h->spec;
    const KGeom &G = h->G;
    S.ok = false;
    if (G.dim != 2 || h->size != 1 || G.ng != 0) return;
    if (!spec_plan(G.nx, S.px) || !spec_plan(G.ny, S.py)) return;
    S.npair = (G.F + 1) / 2;
    const size_t lds_max = 160 * 1024;
    const size_t row_bytes = sizeof(kcf) * (size_t)(G.nx + (G.nx >> 4) + 1);
    int rb = (int)std::min<size_t>((lds_max - 1024) / row_bytes, 16);
    while (rb > 1 && (G.ny % rb)) rb--;                              // power-of-two ny: rb ends up a power of two
    if (rb < 1) return;
    // rows per block: as many as the LDS holds (wider store segments of the transposed write), but keep >= 2 tiles per CU
    while (rb > 2 && G.ny / rb < 512) rb >>= 1;
    if (getenv("KSFD_SPEC_RB")) rb = std::max(1, std::min(rb, atoi(getenv("KSFD_SPEC_RB"))));
    S.rb = rb;
    S.lds_rows = row_bytes * rb;
    S.lds_cols = sizeof(kcf) * (size_t)(G.ny + (G.ny >> 4) + 1) * 2 * S.npair;
    if (S.lds_cols > lds_max - 1024) return;
    if (hipFuncSetAttribute((const void *)k_spec_rows_fwd<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_spec_rows_fwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_spec_rows_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows) != hipSuccess ||
        0) { hipGetLastError(); return; }
    {
        hipError_t e = hipSuccess;
        NL_DISPATCH(h->P.nlig, e = hipFuncSetAttribute((const void *)k_spec_cols<NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_cols));
        if (e != hipSuccess) { hipGetLastError(); return; }
    }
    auto twiddles = [](int n) {
        std::vector<kcf> t(n);
        for (int k = 0; k < n; k++) { const double a = -2.0 * M_PI * k / n; t[k] = make_float2((float)cos(a), (float)sin(a)); }
        return t;
    };
    auto positions = [](const KFFTPlan &P) { std::vector<int> p(P.n); for (int k = 0; k < P.n; k++) p[k] = spec_pos(P, k); return p; };
    auto inverse = [](const std::vector<int> &p) { std::vector<int> q(p.size()); for (size_t k = 0; k < p.size(); k++) q[p[k]] = (int)k; return q; };
    auto symbol = [](int n, double inv_h2) {
        std::vector<float> l(n);
        for (int k = 0; k < n; k++) { const double th = 2.0 * M_PI * k / n; l[k] = (float)((-30.0 + 32.0 * cos(th) - 2.0 * cos(2.0 * th)) / 12.0 * inv_h2); }
        return l;
    };
    // columns of W are padded by 256 B: with a power-of-two column stride every block of the column kernel (and every store
    // segment of the transposed write) would walk the HBM channels in lockstep
    S.nyp = (int)G.ny + (getenv("KSFD_SPEC_PAD") ? 32 : 0);      // (measured: a 256-B pad makes all three kernels ~5 % slower; no channel lockstep to break)
    if (hipMalloc((void **)&S.W, sizeof(kcf) * (size_t)S.npair * G.nx * S.nyp) != hipSuccess ||
        !spec_upload(&S.twx, twiddles(S.px.n)) || !spec_upload(&S.twy, twiddles(S.py.n)) ||
        !spec_upload(&S.posx, positions(S.px)) || !spec_upload(&S.posy, positions(S.py)) || !spec_upload(&S.kyofpos, inverse(positions(S.py))) ||
        !spec_upload(&S.lx, symbol(S.px.n, h->P.inv_h2[0])) || !spec_upload(&S.ly, symbol(S.py.n, h->P.inv_h2[1]))) { hipGetLastError(); spec_free(h); return; }
    S.ok = true;
}

// means of rho*G_rho, rho*G_Ul over the frozen coefficient planes (call after op_jcoef, once per step)
static int spec_means(ksfd_handle *h)
{
    SpecState &S = h->spec;
    const KGeom &G = h->G;
    const int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 2048);
    {
        Scope sc(h, KC_SPECTRAL, 8.0 * (2 + h->P.nlig) * (double)G.nloc);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_spec_means<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, (const double *)h->coef, h->part));
    }
    HIPCHK(h, hipGetLastError());
    int rc = reduce_rows(h, 1 + h->P.nlig, nb, 0);
    if (rc) return rc;
    const double ntot = (double)h->cfg.n[0] * (double)h->cfg.n[1] * (double)h->cfg.n[2];
    S.a_rr = h->hres[0] / ntot;
    for (int l = 0; l < h->P.nlig; l++) S.a_rU[l] = h->hres[1 + l] / ntot;
    S.means_valid = true;
    return KSFD_OK;
}

// z = M^-1 v, M = shift*I - J0 (constant-coefficient part of the frozen Jacobian)
// xadd != NULL: z = xadd + M^-1 v (one Richardson update without a vector pass of its own)
// v32 != NULL: the input is that fp32 copy (F planes of floats, same geometry) instead of v
static int spec_apply(ksfd_handle *h, double shift, const double *v, double *z, const double *xadd = nullptr, const float *v32 = nullptr)
{
    SpecState &S = h->spec;
    const KGeom &G = h->G;
    if (!S.ok || !S.means_valid) return fail(h, KSFD_EINVAL, "spectral preconditioner not available for this handle");
    KSpecSym Y;
    memset(&Y, 0, sizeof Y);
    Y.nlig = h->P.nlig;
    Y.shift = (float)shift; Y.a_rr = (float)S.a_rr;
    Y.scale = (float)(1.0 / ((double)G.nx * (double)G.ny));
    Y.den_floor = (float)(0.02 * shift);
    for (int l = 0; l < h->P.nlig; l++) { Y.a_rU[l] = (float)S.a_rU[l]; Y.s[l] = (float)h->P.lig_s[l]; Y.gam[l] = (float)h->P.lig_gamma[l]; Y.D[l] = (float)h->P.lig_D[l]; }
    const int ntiles = (int)(G.ny / S.rb);
    // timing-only diagnostics (wrong results): KSFD_SPEC_DIAG bit0/1/2 = skip the FFT stages of the rows-fwd / cols / rows-inv kernel, bit3 = rows-fwd stores tile-major (contiguous)
    static const int diag = getenv("KSFD_SPEC_DIAG") ? atoi(getenv("KSFD_SPEC_DIAG")) : 0;
    KFFTPlan px_f = S.px, py_c = S.py, px_i = S.px;
    if (diag & 1) px_f.nstage = 0;
    if (diag & 2) py_c.nstage = 0;
    if (diag & 4) px_i.nstage = 0;
    int thr_rows = (int)std::min<long long>(1024, std::max<long long>(256, (long long)S.rb * G.nx / 16));
    if (getenv("KSFD_SPEC_THRR")) thr_rows = atoi(getenv("KSFD_SPEC_THRR"));
    int thr_cols = (int)std::min<long long>(512, std::max<long long>(128, (long long)2 * S.npair * G.ny / 16));
    if (getenv("KSFD_SPEC_THRC")) thr_cols = atoi(getenv("KSFD_SPEC_THRC"));
    const double fn = (double)G.F * (double)G.nloc, pn = 8.0 * S.npair * (double)G.nloc;
    {
        Scope sc(h, KC_SPECTRAL, (v32 ? 4.0 : 8.0) * fn + pn, 8.0 * fn);        // read v | write W
        if (v32) hipLaunchKernelGGL(k_spec_rows_fwd<float>, dim3(ntiles, S.npair), dim3(thr_rows), S.lds_rows, h->st, px_f, S.nyp, S.rb, (diag & 8) ? -ntiles : ntiles, G.F, v32, G.plane, S.W, (const kcf *)S.twx);
        else hipLaunchKernelGGL(k_spec_rows_fwd<double>, dim3(ntiles, S.npair), dim3(thr_rows), S.lds_rows, h->st, px_f, S.nyp, S.rb, (diag & 8) ? -ntiles : ntiles, G.F, v, G.plane, S.W, (const kcf *)S.twx);
    }
    {
        Scope sc(h, KC_SPECTRAL, 2.0 * pn, 0.0);                                 // W in place
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_spec_cols<NL>), dim3((unsigned)(G.nx / 2)), dim3(thr_cols), S.lds_cols, h->st, py_c, (int)G.nx, S.nyp, S.W, (const kcf *)S.twy,
                           (const int *)S.posx, (const int *)S.posy, (const int *)S.kyofpos, (const float *)S.lx, (const float *)S.ly, Y));
    }
    {
        Scope sc(h, KC_SPECTRAL, pn + (xadd ? 16.0 : 8.0) * fn, (xadd ? 16.0 : 8.0) * fn);     // read W (+ x) | write z
        hipLaunchKernelGGL(k_spec_rows_inv, dim3(ntiles, S.npair), dim3(thr_rows), S.lds_rows, h->st, px_i, S.nyp, S.rb, ntiles, G.F, (const kcf *)S.W, z, G.plane, (const kcf *)S.twx, xadd);
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}
