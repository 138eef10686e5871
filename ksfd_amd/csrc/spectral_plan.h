// libksfd_hip.so -- FFT plans and small host-side tables of the spectral preconditioner (no handle, no device code): shared by
// spectral_host.hip.h and the stand-alone kernel laboratory tools/spec_lab.hip
#pragma once
#include <math.h>
#include <vector>

static bool spec_plan(long long n, KFFTPlan &P)
{
    if (n < 32 || n > 16384 || (n & (n - 1))) return false;
    int lg = 0;
    while ((1LL << lg) < n) lg++;
    P.n = (int)n; P.lg = lg; P.nstage = 0; P.flags = 0;
    int left = lg;
    // radix 16 from the top, then one smaller stage -- except that a trailing [16, 2] becomes [8, 4]: a radix-2 stage costs a full LDS
    // pass and a barrier for a quarter of the work (512 = 16*8*4, 8192 = 16*16*8*4).  kspec_stage_any relies on exactly these shapes.
    // (n = 32 stays [16, 2]: the row kernels and the slab ownership want a leading radix 16)
    if (lg == 5) { P.radix[P.nstage++] = 16; P.radix[P.nstage++] = 2; left = 0; }
    while (left >= 4 && left != 5 && P.nstage < KSPEC_MAXSTAGE) { P.radix[P.nstage++] = 16; left -= 4; }
    if (left == 5) { P.radix[P.nstage++] = 8; P.radix[P.nstage++] = 4; left = 0; }
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

// Slab ranks (P = 2, 4, 8): the x-transforms are local (a rank owns whole rows); for the y-transforms every rank needs whole
// columns, so the transposed work array is redistributed by an all-to-all: rank q gets the spectral positions whose TOP digit
// (= lowest radix-16 digit of kx) is in its share of the list below.  kx and -kx have top digits d and (16 - d) % 16, so with
// the digits handed out in these pairs both columns of every {kx, -kx} pair land on one rank.
static const int spec_digit_order[16] = { 0, 8, 1, 15, 2, 14, 3, 13, 4, 12, 5, 11, 6, 10, 7, 9 };

static std::vector<kcf> spec_twiddles(int n)
{
    std::vector<kcf> t(n);
    for (int k = 0; k < n; k++) { const double a = -2.0 * M_PI * k / n; t[k] = make_float2((float)cos(a), (float)sin(a)); }
    return t;
}
static std::vector<int> spec_positions(const KFFTPlan &Q) { std::vector<int> p(Q.n); for (int k = 0; k < Q.n; k++) p[k] = spec_pos(Q, k); return p; }
static std::vector<int> spec_inverse(const std::vector<int> &p) { std::vector<int> q(p.size()); for (size_t k = 0; k < p.size(); k++) q[p[k]] = (int)k; return q; }
static std::vector<float> spec_symbol_table(int n, double inv_h2)
{
    std::vector<float> l(n);
    for (int k = 0; k < n; k++) { const double th = 2.0 * M_PI * k / n; l[k] = (float)((-30.0 + 32.0 * cos(th) - 2.0 * cos(2.0 * th)) / 12.0 * inv_h2); }
    return l;
}

