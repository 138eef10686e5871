// Variants of the spectral kernels under test (included by tools/spec_lab.hip after the Lab helpers).
#pragma once

template <typename REP>
static void lab_variants(Lab &L, int reps, const double *xref, REP report)
{
    const double N = (double)L.n * L.n;
    // KFFTPlan.flags bit 3: the column kernel stores tile-major for the inverse row kernel
    for (int fl : { 7, 15, 7, 15, 7, 15 }) {
        char nm[96];
        snprintf(nm, sizeof nm, "cols, flags %d", fl);
        base_fwd(L);
        report(nm, timeit(L, reps, [&] { base_cols(L, fl); }), 16.0 * N);
        snprintf(nm, sizeof nm, "inv (x += z), flags %d", fl);
        report(nm, timeit(L, reps, [&] { base_inv(L, fl); }), 40.0 * N);
        snprintf(nm, sizeof nm, "whole application, flags %d", fl);
        report(nm, timeit(L, reps, [&] { base_fwd(L, fl); base_cols(L, fl); base_inv(L, fl); }), 72.0 * N);
        CK(hipMemcpy(L.x, L.x0, sizeof(double) * L.F * L.plane, hipMemcpyDeviceToDevice));
        base_fwd(L, fl); base_cols(L, fl); base_inv(L, fl);
        CK(hipStreamSynchronize(L.st));
        printf("    max rel diff vs the first result %.3e\n", maxdiff(L, L.x, xref));
    }
}
