// Variants of the spectral kernels under test (included by tools/spec_lab.hip after the Lab helpers).
#pragma once

template <typename REP>
static void lab_variants(Lab &L, int reps, const double *xref, REP report)
{
    const double N = (double)L.n * L.n;
    // layout of the forward work array (KFFTPlan.lgw): tile-major (-1) against position groups of 2^lgw
    for (int lgw : { -1, -1 }) {
        L.lgw = lgw;
        char nm[96];
        snprintf(nm, sizeof nm, "fwd, lgw %d", lgw);
        report(nm, timeit(L, reps, [&] { base_fwd(L); }), 16.0 * N);
        snprintf(nm, sizeof nm, "cols, lgw %d", lgw);
        report(nm, timeit(L, reps, [&] { base_cols(L); }), 16.0 * N);
        CK(hipMemcpy(L.x, L.x0, sizeof(double) * L.F * L.plane, hipMemcpyDeviceToDevice));
        base_fwd(L); base_cols(L); base_inv(L);
        CK(hipStreamSynchronize(L.st));
        printf("    max rel diff vs the first result %.3e\n", maxdiff(L, L.x, xref));
    }
    L.lgw = -1;
}
