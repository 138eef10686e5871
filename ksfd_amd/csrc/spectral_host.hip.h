// libksfd_hip.so -- host side of the spectral preconditioner (kernels and rationale: spectral.hip.h)
// (part of the single translation unit ksfd_hip.hip; included after ops.hip.h)
#pragma once
#include "spectral_plan.h"

static void spec_free(ksfd_handle *h)
{
    SpecState &S = h->spec;
    void *bufs[] = { S.W, S.W2, S.twx, S.twy, S.twz != S.twy ? S.twz : nullptr, S.posy, S.kyofpos, S.pairtab, S.lx, S.ly, S.posz, S.kzofpos, S.lz, S.ytab };
    for (void *b : bufs) if (b) hipFree(b);
    S = SpecState();
}

template <typename T> static bool spec_upload(T **dev, const std::vector<T> &host)
{
    return hipMalloc((void **)dev, sizeof(T) * host.size()) == hipSuccess &&
           hipMemcpy(*dev, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice) == hipSuccess;
}

// 3-D (see spectral.hip.h): plans per axis, the column-pair table of the z kernel, two work arrays.  z-slab ranks (P = 2, 4, 8): the
// x and y transforms are local (a rank owns whole z planes); for the z transforms every rank needs whole z columns, so
// W[pair][pos_x][pos_y][z_local] is redistributed by an all-to-all over pos_x exactly as in 2-D (same digit ownership: the columns
// (kx, ky) and (-kx, -ky) land on one rank), a received column consisting of P pieces of nz/P elements.
static void spec_build3d(ksfd_handle *h)
{
    SpecState &S = h->spec;
    const KGeom &G = h->G;
    const int P = h->size;
    if (getenv("KSFD_SPEC_NO3D")) return;
    const bool ring = h->ring;                                     // slab layout + all-to-alls (also a ring of ONE rank: every piece is its own)
    if (!ring && G.ng != 0) return;
    if (ring && (!h->tr || !h->tr->has_alltoall() || (P != 1 && P != 2 && P != 4 && P != 8) || getenv("KSFD_SPEC_SINGLE"))) return;
    const long long nzg = h->cfg.n[2], nzl = G.sloc;               // global / local z planes
    if (!spec_plan(G.nx, S.px) || !spec_plan(G.ny, S.py) || !spec_plan(nzg, S.pz)) return;
    if (S.px.m != 1 || S.py.m != 1 || S.pz.m != 1) return;           // (3 * 2^k extents: 2-D, one rank so far)
    if (S.px.radix[0] != 16 || G.nx > 32768 || G.ny > 32768 || (nzl & (nzl - 1)) || nzl < 2) return;
    S.dim = 3;
    S.npair = (G.F + 1) / 2;
    const size_t lds_max = 160 * 1024 - 1024;
    const long long nrows = G.ny * nzl;
    const size_t row_bytes = sizeof(kcf) * (size_t)(G.nx + (G.nx >> 4) + 1);
    int rb = 16;
    while (rb > 1 && row_bytes * rb > lds_max) rb >>= 1;
    while (rb > 2 && nrows / rb < 512) rb >>= 1;
    if (row_bytes * rb > lds_max || nrows % rb || G.ny % rb) return;      // a tile must not straddle two z planes' y ranges unevenly
    S.rb = rb;
    S.nyp = (int)nrows;                                            // column stride of the array the inverse row kernel reads
    S.lg_rb3 = 0; while ((1 << S.lg_rb3) < rb) S.lg_rb3++;
    S.lds_rows = row_bytes * rb;
    const size_t ycol = sizeof(kcf) * (size_t)(G.ny + (G.ny >> 4) + 1) * S.npair, zcol = sizeof(kcf) * (size_t)(nzg + (nzg >> 4) + 1) * 2 * S.npair;
    int cz = 16;
    while (cz > 1 && (ycol * cz > lds_max / 2 || nzl % cz)) cz >>= 1;      // <= half the LDS: two blocks per CU
    if (getenv("KSFD_SPEC_CZ")) cz = std::max(1, std::min(cz, atoi(getenv("KSFD_SPEC_CZ"))));        // experiment knob
    if (ycol * cz > lds_max) return;
    S.lg_cz = 0; while ((1 << S.lg_cz) < cz) S.lg_cz++;
    S.lds_y3 = ycol * cz;
    int pb = 16;
    while (pb > 1 && zcol * pb > lds_max / 8) pb >>= 1;                     // small blocks, many per CU: their load / transform / store phases overlap (512^3: pb 8 -> 2, 1.27 -> 1.12 ms)
    if (getenv("KSFD_SPEC_PB")) pb = std::max(1, std::min(pb, atoi(getenv("KSFD_SPEC_PB"))));        // experiment knob
    if (zcol * pb > lds_max) return;
    S.pb = pb;
    S.lds_z3 = zcol * pb;
    {
        hipError_t e = hipFuncSetAttribute((const void *)k_spec_rows_fwd<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_spec_rows_fwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_spec_rows_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_spec3_y_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_y3);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_spec3_y_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_y3);
        if (e == hipSuccess) e = S.npair == 1 ? hipFuncSetAttribute((const void *)k_spec3_z<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_z3)
                               : S.npair == 2 ? hipFuncSetAttribute((const void *)k_spec3_z<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_z3)
                                              : hipFuncSetAttribute((const void *)k_spec3_z<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_z3);
        if (e != hipSuccess) { hipGetLastError(); return; }
    }
    // ownership of the x positions (as in 2-D): top digit -> (rank, index of the digit in that rank's list)
    const int nx = (int)G.nx, ny = (int)G.ny, nx16 = nx / 16, ndig = 16 / P;
    int dig_rank[16], dig_idx[16];
    for (int q = 0; q < P; q++) for (int di = 0; di < ndig; di++) { const int d = spec_digit_order[q * ndig + di]; dig_rank[d] = q; dig_idx[d] = di; }
    S.nxl = nx / P;
    S.lg_pl = 0;
    while ((1LL << S.lg_pl) < nzl) S.lg_pl++;
    auto local_index = [&](int j) { return !ring ? j : dig_idx[j / nx16] * nx16 + (j % nx16); };
    // pairs of columns {(kx,ky), (-kx,-ky)}; a column is addressed by its (local) position pair
    const std::vector<int> posx = spec_positions(S.px), posy = spec_positions(S.py);
    std::vector<int4> ent;
    ent.reserve((size_t)S.nxl * ny / 2 + 4);
    for (int kx = 0; kx <= nx / 2; kx++) {
        if (dig_rank[posx[kx] / nx16] != h->rank) continue;
        for (int ky = 0; ky < ny; ky++) {
            const int kxm = (nx - kx) % nx, kym = (ny - ky) % ny;
            const bool self = kxm == kx && kym == ky;
            if (!self && (kxm < kx || (kxm == kx && kym < ky))) continue;       // the partner comes first: it owns the pair
            const int ca = local_index(posx[kx]) * ny + posy[ky], cb = local_index(posx[kxm]) * ny + posy[kym];
            ent.push_back(make_int4(ca, cb, kx | (ky << 16), self ? 1 : 0));
        }
    }
    std::sort(ent.begin(), ent.end(), [](const int4 &a, const int4 &b) { return a.x < b.x; });
    S.nent = (int)ent.size();
    const size_t wbytes = sizeof(kcf) * (size_t)S.npair * G.nx * nrows;
    if (hipMalloc((void **)&S.W, wbytes) != hipSuccess || hipMalloc((void **)&S.W2, wbytes) != hipSuccess ||
        !spec_upload(&S.twx, spec_twiddles(S.px)) || !spec_upload(&S.twy, spec_twiddles(S.py)) ||
        !spec_upload(&S.posz, spec_positions(S.pz)) || !spec_upload(&S.kzofpos, spec_inverse(spec_positions(S.pz))) || !spec_upload(&S.pairtab, ent) ||
        !spec_upload(&S.lx, spec_symbol_table(S.px.n, h->P.inv_h2[0])) || !spec_upload(&S.ly, spec_symbol_table(S.py.n, h->P.inv_h2[1])) ||
        !spec_upload(&S.lz, spec_symbol_table(S.pz.n, h->P.inv_h2[2])) ||
        !spec_upload(&S.ytab, spec_partner_table(S.pz, spec_symbol_table(S.pz.n, h->P.inv_h2[2])))) { hipGetLastError(); spec_free(h); return; }
    // the z transforms need their own twiddle table when nz differs from ny: kept behind twy in one allocation is not worth it
    if (nzg != G.ny) {
        kcf *tz = nullptr;
        if (!spec_upload(&tz, spec_twiddles(S.pz))) { hipGetLastError(); spec_free(h); return; }
        S.twz = tz;
    } else S.twz = S.twy;
    if (ring) {
        // pieces of the two all-to-alls: one per (peer, pair, top digit of the receiver) = nx/16 positions x ny x nzl, contiguous on both sides
        const size_t pel = (size_t)nx16 * ny * nzl, pbytes = sizeof(kcf) * pel;
        for (int q = 0; q < P; q++)
            for (int p = 0; p < S.npair; p++)
                for (int di = 0; di < ndig; di++) {
                    const int dq = spec_digit_order[q * ndig + di];
                    kcf *mine_for_q = S.W + ((size_t)p * nx + (size_t)dq * nx16) * ny * nzl;                          // my z planes of q's columns
                    kcf *from_q = S.W2 + (((size_t)q * S.npair + p) * S.nxl + (size_t)di * nx16) * ny * nzl;          // q's z planes of my columns
                    S.a2a_fwd_s.push_back({ q, mine_for_q, pbytes });
                    S.a2a_fwd_r.push_back({ q, from_q, pbytes });
                    S.a2a_bwd_s.push_back({ q, from_q, pbytes });
                    S.a2a_bwd_r.push_back({ q, mine_for_q, pbytes });
                }
    }
    S.tile_major = true;
    S.ok = true;
}

// Builds plans, tables and the work arrays if this handle can use the spectral solver; leaves spec.ok = false otherwise.
static void spec_build(ksfd_handle *h)
{
    SpecState &S = h->spec;
    const KGeom &G = h->G;
    const int P = h->size;
    S.ok = false;
    if (G.dim == 3) { spec_build3d(h); return; }
    if (G.dim != 2) return;
    const bool ring = h->ring;                                       // slab layout + all-to-alls (also a ring of ONE rank: every piece is its own)
    if (ring && (!h->tr || !h->tr->has_alltoall() || (P != 1 && P != 2 && P != 4 && P != 8) || getenv("KSFD_SPEC_SINGLE"))) return;
    const long long ny = h->cfg.n[1], nyl = G.sloc;                  // global / local rows
    if (!spec_plan(G.nx, S.px) || !spec_plan(ny, S.py)) return;
    const bool pow2 = S.px.m == 1 && S.py.m == 1;
    if (ring && (!pow2 || S.px.radix[0] != 16 || (nyl & (nyl - 1)))) return;      // slab ownership goes by the top radix-16 digit
    if (nyl < 4) return;
    S.npair = (G.F + 1) / 2;
    const size_t lds_max = 160 * 1024;
    const size_t row_bytes = sizeof(kcf) * spec_sstride(S.px);
    int rb = (int)std::min<size_t>((lds_max - 1024) / row_bytes, 16);
    while (rb & (rb - 1)) rb &= rb - 1;                              // a power of two ...
    while (rb > 1 && (nyl % rb)) rb >>= 1;                           // ... that divides the rows
    if (rb < 1) return;
    // rows per block: as many as the LDS holds (wider store segments of the transposed write), but keep >= 2 tiles per CU
    while (rb > 2 && nyl / rb < 512) rb >>= 1;
    if (getenv("KSFD_SPEC_RB")) rb = std::max(1, std::min(rb, atoi(getenv("KSFD_SPEC_RB"))));
    S.rb = rb;
    S.lds_rows = row_bytes * rb;
    S.lds_cols = sizeof(kcf) * spec_sstride(S.py) * 2 * S.npair;
    const int split_env = getenv("KSFD_SPEC_SPLIT") ? atoi(getenv("KSFD_SPEC_SPLIT")) : 0;        // (the knob: tests of the split paths on small grids)
    S.cols_split = (S.lds_cols > lds_max - 1024 || (split_env == 1 && S.npair > 1) || split_env == 2) ? 1 : 0;
    if (S.cols_split) S.lds_cols = sizeof(kcf) * spec_sstride(S.py) * 2;                            // one field pair per block
    if (S.cols_split && (S.lds_cols > lds_max - 1024 || split_env == 2)) {                          // ... one column per block (more than 8192 points)
        S.cols_split = 2;
        S.lds_cols = sizeof(kcf) * spec_sstride(S.py);
    }
    if (S.lds_cols > lds_max - 1024) return;
    if (S.cols_split && !pow2) return;                               // (the two-phase column kernel is power-of-two only)
    if (hipFuncSetAttribute((const void *)k_spec_rows_fwd<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_spec_rows_fwd<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_spec_rows_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_rows) != hipSuccess) { hipGetLastError(); return; }
    {
        hipError_t e = hipSuccess;
        e = S.cols_split ? hipFuncSetAttribute((const void *)k_spec_cols_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_cols)
                         : S.npair == 1 ? hipFuncSetAttribute((const void *)k_spec_cols<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_cols)
                         : S.npair == 2 ? hipFuncSetAttribute((const void *)k_spec_cols<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_cols)
                                        : hipFuncSetAttribute((const void *)k_spec_cols<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_cols);
        if (e != hipSuccess) { hipGetLastError(); return; }
    }
    auto positions = [](const KFFTPlan &Q) { std::vector<int> p(Q.n); for (int k = 0; k < Q.n; k++) p[k] = spec_pos(Q, k); return p; };
    auto inverse = [](const std::vector<int> &p) { std::vector<int> q(p.size()); for (size_t k = 0; k < p.size(); k++) q[p[k]] = (int)k; return q; };
    auto symbol = [](int n, double inv_h2) {
        std::vector<float> l(n);
        for (int k = 0; k < n; k++) { const double th = 2.0 * M_PI * k / n; l[k] = (float)((-30.0 + 32.0 * cos(th) - 2.0 * cos(2.0 * th)) / 12.0 * inv_h2); }
        return l;
    };
    // ownership of spectral positions: top digit -> (rank, index of the digit in that rank's list)
    const int nx = (int)G.nx, nx16 = nx / 16, ndig = 16 / P;
    int dig_rank[16], dig_idx[16];
    for (int q = 0; q < P; q++) for (int di = 0; di < ndig; di++) { const int d = spec_digit_order[q * ndig + di]; dig_rank[d] = q; dig_idx[d] = di; }
    S.nxl = nx / P;
    S.lg_pl = 0;
    while ((1LL << S.lg_pl) < nyl) S.lg_pl++;
    const std::vector<int> posx = positions(S.px);
    // among the owner's positions (one rank: the work array is used in place, positions are their own index)
    auto local_index = [&](int j) { return !ring ? j : dig_idx[j / nx16] * nx16 + (j % nx16); };
    std::vector<int4> pairs;
    for (int kx = 0; kx <= nx / 2; kx++) {
        const int kxm = (nx - kx) % nx, j = posx[kx], jm = posx[kxm];
        if (dig_rank[j / nx16] != h->rank) continue;
        if (kx == 0) pairs.push_back(make_int4(local_index(posx[0]), local_index(posx[nx / 2]), 0, nx / 2 + 1));     // the two self-paired columns share a block
        else if (kx != nx / 2) pairs.push_back(make_int4(local_index(j), local_index(jm), kx, 0));
    }
    // neighbouring blocks of the column kernel should own neighbouring positions (they share the 128-B lines of the tile-major
    // array, and of the transposed one through the inverse row kernel's tiles)
    std::sort(pairs.begin(), pairs.end(), [](const int4 &a, const int4 &b) { return a.x < b.x; });
    S.nblk_cols = (int)pairs.size();
    if (S.nblk_cols != S.nxl / 2) return;                            // (cannot happen for P in {1, 2, 4, 8}: the digit pairs keep kx and -kx together)
    // column stride of W: the rows of this rank, rounded up to the power of two the column kernel addresses pieces with (equal for 2^k rows)
    const long long cstride = ring ? nyl : (1LL << S.lg_pl);
    const size_t wbytes = sizeof(kcf) * (size_t)S.npair * G.nx * cstride;
    S.nyp = (int)cstride;
    S.tile_major = !ring && rb >= 2 && !getenv("KSFD_SPEC_TRANSPOSED");
    if (hipMalloc((void **)&S.W, wbytes) != hipSuccess || ((ring || S.tile_major || S.cols_split) && hipMalloc((void **)&S.W2, wbytes) != hipSuccess) ||
        !spec_upload(&S.twx, spec_twiddles(S.px)) || !spec_upload(&S.twy, spec_twiddles(S.py)) ||
        !spec_upload(&S.posy, positions(S.py)) || !spec_upload(&S.kyofpos, inverse(positions(S.py))) || !spec_upload(&S.pairtab, pairs) ||
        !spec_upload(&S.lx, symbol(S.px.n, h->P.inv_h2[0])) || !spec_upload(&S.ly, symbol(S.py.n, h->P.inv_h2[1])) ||
        !spec_upload(&S.ytab, spec_partner_table(S.py, symbol(S.py.n, h->P.inv_h2[1])))) { hipGetLastError(); spec_free(h); return; }
    // one rank: the forward work array is tile-major; KSFD_SPEC_LGW=<k> makes it position-group-major with groups of 2^k (kspec_wt_index;
    // k = 2 at 4 rows per tile: one 128-B line per (group, tile).  Measured with the kernel laboratory at 4096^2 (tools/spec_lab.hip, median
    // of 7 batches): row kernel 70 -> 74 us, column kernel 94 -> 96 us -- no gain once the symbol stage stopped waiting on table look-ups)
    S.lgw = -1;
    if (S.tile_major) {
        int lg_rb = 0;
        while ((1 << lg_rb) < rb) lg_rb++;
        if (getenv("KSFD_SPEC_LGW")) S.lgw = std::max(-1, std::min(atoi(getenv("KSFD_SPEC_LGW")), 8));
        if (S.lgw >= 0 && (S.nxl & ((1 << S.lgw) - 1))) S.lgw = -1;
    }
    if (ring) {
        // pieces of the two all-to-alls: one per (peer, pair, top digit of the receiver) = nx/16 columns x nyl rows, contiguous on both sides
        const size_t pbytes = sizeof(kcf) * (size_t)nx16 * nyl;
        for (int q = 0; q < P; q++)
            for (int p = 0; p < S.npair; p++)
                for (int di = 0; di < ndig; di++) {
                    const int dq = spec_digit_order[q * ndig + di], dme = spec_digit_order[h->rank * ndig + di];
                    kcf *mine_for_q = S.W + ((size_t)p * nx + (size_t)dq * nx16) * nyl;                               // my rows of q's columns
                    kcf *from_q = S.W2 + (((size_t)q * S.npair + p) * S.nxl + (size_t)di * nx16) * nyl;               // q's rows of my columns
                    kcf *back_mine = S.W + ((size_t)p * nx + (size_t)dq * nx16) * nyl;
                    (void)dme;
                    S.a2a_fwd_s.push_back({ q, mine_for_q, pbytes });
                    S.a2a_fwd_r.push_back({ q, from_q, pbytes });
                    if (S.cols_split) {
                        // phase 2 of the split column kernel leaves its result in W (used as scratch in the layout of W2); it comes
                        // home into W2 (in the layout of W), which is where the inverse row kernel then reads
                        S.a2a_bwd_s.push_back({ q, S.W + (from_q - S.W2), pbytes });
                        S.a2a_bwd_r.push_back({ q, S.W2 + (back_mine - S.W), pbytes });
                    } else {
                    S.a2a_bwd_s.push_back({ q, from_q, pbytes });
                    S.a2a_bwd_r.push_back({ q, back_mine, pbytes });
                    }
                }
    }
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
// guess != NULL (with v): z = sum_j c_j Y_j + M^-1 (v - sum_j c_j b_j), the first sweep from an initial guess in span{Y_j}
struct SpecGuess { int n; const double *Y[3], *b[3]; double c[3]; };
static int spec_apply(ksfd_handle *h, double shift, const double *v, double *z, const double *xadd = nullptr, const float *v32 = nullptr, const SpecGuess *guess = nullptr)
{
    SpecState &S = h->spec;
    const KGeom &G = h->G;
    if (!S.ok || !S.means_valid) return fail(h, KSFD_EINVAL, "spectral preconditioner not available for this handle");
    KSpecSym Y;
    memset(&Y, 0, sizeof Y);
    Y.nlig = h->P.nlig;
    Y.shift = (float)shift; Y.a_rr = (float)S.a_rr;
    Y.scale = (float)(1.0 / ((double)G.nx * (double)h->cfg.n[1] * (double)h->cfg.n[2]));
    Y.den_floor = (float)(0.02 * shift);
    for (int l = 0; l < h->P.nlig; l++) { Y.a_rU[l] = (float)S.a_rU[l]; Y.s[l] = (float)h->P.lig_s[l]; Y.gam[l] = (float)h->P.lig_gamma[l]; Y.D[l] = (float)h->P.lig_D[l]; }
    const bool d3 = S.dim == 3;
    const int ntiles = (int)((d3 ? G.ny * G.sloc : G.sloc) / S.rb);     // 3-D: the x rows are the nz*ny rows of the box
    const long long goff = (long long)G.ng * G.inner;                // the row kernels address owned rows only
    KSpecLin ex, add;
    memset(&ex, 0, sizeof ex); memset(&add, 0, sizeof add);
    if (xadd) { add.n = 1; add.p[0] = xadd + goff; add.a[0] = 1.0; }
    if (guess && !v32) {
        for (int j = 0; j < guess->n && j < 3; j++) {
            ex.p[ex.n] = guess->b[j] + goff; ex.a[ex.n++] = -guess->c[j];
            add.p[add.n] = guess->Y[j] + goff; add.a[add.n++] = guess->c[j];
        }
    }
    const long long ny_glob = h->cfg.n[1];
    // timing-only diagnostics (wrong results): KSFD_SPEC_DIAG bit0/1/2 = skip the FFT stages of the rows-fwd / cols / rows-inv kernel
    static const int diag = getenv("KSFD_SPEC_DIAG") ? atoi(getenv("KSFD_SPEC_DIAG")) : 0;
    KFFTPlan px_f = S.px, py_c = S.py, px_i = S.px;
    if (diag & 1) px_f.nstage = 0;
    if (diag & 2) py_c.nstage = 0;
    if (diag & 4) px_i.nstage = 0;
    static const int fuse_env = getenv("KSFD_SPEC_FUSE") ? atoi(getenv("KSFD_SPEC_FUSE")) : 7;
    // bit 3 (the column kernel stores tile-major for the inverse row kernel): one rank, 2-D, plain column kernel -- set from KSFD_SPEC_FUSE as the others
    const int fuse = (d3 || h->ring || S.cols_split || !S.tile_major) ? (fuse_env & 7) : fuse_env;
    px_f.flags = py_c.flags = px_i.flags = fuse;
    px_f.lgw = py_c.lgw = d3 ? -1 : S.lgw;
    int thr_rows = (int)std::min<long long>(1024, std::max<long long>(256, (long long)S.rb * G.nx / 16));
    if (getenv("KSFD_SPEC_THRR")) thr_rows = atoi(getenv("KSFD_SPEC_THRR"));
    int thr_cols = (int)std::min<long long>(512, std::max<long long>(128, (long long)(S.cols_split == 2 ? 1 : 2) * (S.cols_split ? 1 : S.npair) * ny_glob / 16));
    if (getenv("KSFD_SPEC_THRC")) thr_cols = atoi(getenv("KSFD_SPEC_THRC"));
    const double fn = (double)G.F * (double)G.nloc, pn = 8.0 * S.npair * (double)G.nloc;
    {
        Scope sc(h, KC_SPECTRAL, (v32 ? 4.0 : 8.0) * fn + 8.0 * ex.n * fn + pn, 8.0 * (1 + ex.n) * fn);        // read v (+ guess vectors) | write W
        if (v32) hipLaunchKernelGGL(k_spec_rows_fwd<float>, dim3(ntiles, S.npair), dim3(thr_rows), S.lds_rows, h->st, px_f, S.nyp, S.rb, S.tile_major ? -ntiles : ntiles, G.F, v32 + goff, G.plane, S.tile_major ? S.W2 : S.W, (const kcf *)S.twx, ex);
        else hipLaunchKernelGGL(k_spec_rows_fwd<double>, dim3(ntiles, S.npair), dim3(thr_rows), S.lds_rows, h->st, px_f, S.nyp, S.rb, S.tile_major ? -ntiles : ntiles, G.F, v + goff, G.plane, S.tile_major ? S.W2 : S.W, (const kcf *)S.twx, ex);
    }
    if (d3) {
        const int cz = 1 << S.lg_cz;
        const long long nzg = h->cfg.n[2], nzl = G.sloc;
        const int thr_y = (int)std::min<long long>(1024, std::max<long long>(256, (long long)S.npair * cz * G.ny / 16));
        int thr_z = (int)std::min<long long>(1024, std::max<long long>(128, (long long)2 * S.npair * S.pb * nzg / 16));
        if (getenv("KSFD_SPEC_THRZ")) thr_z = atoi(getenv("KSFD_SPEC_THRZ"));
        // fused edge stages of the y and z kernels (bit 0: first forward stage on the loaded values, bit 1: last inverse stage into the store)
        static const int fuse3 = getenv("KSFD_SPEC_FUSE3") ? atoi(getenv("KSFD_SPEC_FUSE3")) : 3;
        KFFTPlan py3 = S.py, pz3 = S.pz;
        py3.flags = pz3.flags = fuse3 & 3;
        {
            Scope sc(h, KC_SPECTRAL, 2.0 * pn, 0.0);
            hipLaunchKernelGGL(k_spec3_y_fwd, dim3((unsigned)(nzl / cz), (unsigned)G.nx), dim3(thr_y), S.lds_y3, h->st, py3, (int)G.nx, (int)nzl, S.lg_cz, S.npair, S.lg_rb3,
                               (const kcf *)S.W2, S.W, (const kcf *)S.twy);
        }
        kcf *Wz = S.W;
        if (h->ring) {                                                // z planes of everybody's columns -> whole z columns of mine
            Scope sc(h, KC_HALO, pn);
            if (h->tr->alltoall(S.a2a_fwd_s, S.a2a_fwd_r, h->st)) return fail(h, KSFD_ECOMM, "spectral all-to-all failed: %s", h->tr->error().c_str());
            Wz = S.W2;
        }
        {
            Scope sc(h, KC_SPECTRAL, 2.0 * pn, 0.0);
            const long long pstride = (long long)S.npair * S.nxl * G.ny << S.lg_pl;
#define KSPEC_Z_LAUNCH(NP) hipLaunchKernelGGL(k_spec3_z<NP>, dim3((unsigned)((S.nent + S.pb - 1) / S.pb)), dim3(thr_z), S.lds_z3, h->st, pz3, S.nent, S.pb, (long long)S.nxl * G.ny, S.lg_pl, pstride, Wz, \
                               (const kcf *)S.twz, (const int4 *)S.pairtab, (const int *)S.posz, (const int *)S.kzofpos, (const float *)S.lx, (const float *)S.ly, (const float *)S.lz, (const int2 *)S.ytab, Y)
            if (S.npair == 1) KSPEC_Z_LAUNCH(1); else if (S.npair == 2) KSPEC_Z_LAUNCH(2); else KSPEC_Z_LAUNCH(0);
#undef KSPEC_Z_LAUNCH
        }
        if (h->ring) {
            Scope sc(h, KC_HALO, pn);
            if (h->tr->alltoall(S.a2a_bwd_s, S.a2a_bwd_r, h->st)) return fail(h, KSFD_ECOMM, "spectral all-to-all failed: %s", h->tr->error().c_str());
        }
        {
            Scope sc(h, KC_SPECTRAL, 2.0 * pn, 0.0);
            hipLaunchKernelGGL(k_spec3_y_inv, dim3((unsigned)(nzl / cz), (unsigned)G.nx), dim3(thr_y), S.lds_y3, h->st, py3, (int)G.nx, (int)nzl, S.lg_cz, S.npair,
                               (const kcf *)S.W, S.W2, (const kcf *)S.twy);
        }
        // the inverse x rows read W3 = S.W2 ([pair][pos_x][z*ny + y]) below
    } else {
    kcf *Wc = S.W;
    if (h->ring) {                                                    // rows of everybody's columns -> whole columns of mine
        Scope sc(h, KC_HALO, pn);
        if (h->tr->alltoall(S.a2a_fwd_s, S.a2a_fwd_r, h->st)) return fail(h, KSFD_ECOMM, "spectral all-to-all failed: %s", h->tr->error().c_str());
        Wc = S.W2;
    }
    {
        Scope sc(h, KC_SPECTRAL, S.cols_split == 2 ? (3.0 + 2.0 * S.npair) * pn : S.cols_split ? (4.0 + S.npair) * pn : 2.0 * pn, 0.0);      // work array in place (split: 2 + (npair + 2) passes; by column: 2 + (2 npair + 1))
        const long long pstride = (long long)S.npair * S.nxl << S.lg_pl;
        int lg_rb = 0;
        while ((1 << lg_rb) < S.rb) lg_rb++;
        if (S.cols_split) {
            // one rank: tiles (W2) -> spectrum (W) -> result (W2); slab ranks: in place in W2, then result into W as scratch
            kcf *spec = h->ring ? S.W2 : S.W, *res = h->ring ? S.W : S.W2;
            for (int phase = 1; phase <= 2; phase++)
                hipLaunchKernelGGL(k_spec_cols_split, dim3((unsigned)S.nblk_cols, (unsigned)S.npair, S.cols_split == 2 ? 2u : 1u), dim3(thr_cols), S.lds_cols, h->st, phase, py_c, S.nxl, S.lg_pl, pstride, spec,
                                   (const kcf *)(S.tile_major ? S.W2 : nullptr), S.tile_major ? lg_rb : -1, res, (const kcf *)S.twy,
                                   (const int4 *)S.pairtab, (const int *)S.posy, (const int *)S.kyofpos, (const float *)S.lx, (const float *)S.ly, Y);
        } else {
#define KSPEC_COLS_LAUNCH(NP) hipLaunchKernelGGL(k_spec_cols<NP>, dim3((unsigned)S.nblk_cols), dim3(thr_cols), S.lds_cols, h->st, py_c, S.nxl, S.lg_pl, pstride, Wc, \
                           (const kcf *)(S.tile_major ? S.W2 : nullptr), S.tile_major ? lg_rb : -1, (const kcf *)S.twy, \
                           (const int4 *)S.pairtab, (const int *)S.posy, (const int *)S.kyofpos, (const float *)S.lx, (const float *)S.ly, (const int2 *)S.ytab, Y)
            if (S.npair == 1) KSPEC_COLS_LAUNCH(1); else if (S.npair == 2) KSPEC_COLS_LAUNCH(2); else KSPEC_COLS_LAUNCH(0);
#undef KSPEC_COLS_LAUNCH
        }
    }
    if (h->ring) {
        Scope sc(h, KC_HALO, pn);
        if (h->tr->alltoall(S.a2a_bwd_s, S.a2a_bwd_r, h->st)) return fail(h, KSFD_ECOMM, "spectral all-to-all failed: %s", h->tr->error().c_str());
    }
    }
    {
        Scope sc(h, KC_SPECTRAL, pn + 8.0 * (1 + add.n) * fn, 8.0 * (1 + add.n) * fn);     // read W (+ x / guess vectors) | write z
        hipLaunchKernelGGL(k_spec_rows_inv, dim3(ntiles, S.npair), dim3(thr_rows), S.lds_rows, h->st, px_i, S.nyp, S.rb, ntiles, G.F, (const kcf *)((d3 || S.cols_split) ? S.W2 : S.W), z + goff, G.plane, (const kcf *)S.twx, add);
    }
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}
