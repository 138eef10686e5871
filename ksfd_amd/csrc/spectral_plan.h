// libksfd_hip.so -- FFT plans and small host-side tables of the spectral preconditioner (no handle, no device code): shared by
// spectral_host.hip.h and the stand-alone kernel laboratory tools/spec_lab.hip
#pragma once
#include <math.h>
#include <string.h>
#include <vector>

// n = 2^lg (32 ... 16384) or 3 * 2^lg (48 ... 12288): the power-of-two part is radix 16 from the top with one smaller last stage; a
// factor 3 is a radix-3 stage in front of it (spectral.hip.h: KFFTPlan.m)
static bool spec_plan(long long n, KFFTPlan &P)
{
    int m = 1;
    long long n2 = n;
    if (n2 > 0 && n2 % 3 == 0) { m = 3; n2 /= 3; }
    if (n2 < (m == 3 ? 16 : 32) || n2 > (m == 3 ? 4096 : 16384) || (n2 & (n2 - 1))) return false;
    int lg = 0;
    while ((1LL << lg) < n2) lg++;
    P.n = (int)n; P.lg = lg; P.m = m; P.nstage = 0; P.flags = 0; P.lgw = -1;
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

// elements of one sequence in the LDS, padding included (device side: kspec_sstride)
static size_t spec_sstride(const KFFTPlan &P) { const size_t n2 = (size_t)1 << P.lg; return (size_t)P.m * (n2 + (n2 >> 4) + 1); }

// position of frequency k in the output of the DIF stages (see spectral.hip.h): pos = q0*(n/r0) + pos'(k / r0), q0 = k % r0
static int spec_pos(const KFFTPlan &P, int k)
{
    int pos = 0, n = P.n;
    if (P.m == 3) { pos = (k % 3) * (n / 3); k /= 3; n /= 3; }
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

// exp(-2 pi i j / 2^lg), j < 2^lg, for the power-of-two stages; behind it, for 3 * 2^lg, exp(-2 pi i j / n), j < 2^lg (radix-3 stage)
static std::vector<kcf> spec_twiddles(const KFFTPlan &P)
{
    const int n2 = 1 << P.lg;
    std::vector<kcf> t((size_t)n2 * (P.m == 3 ? 2 : 1));
    for (int k = 0; k < n2; k++) { const double a = -2.0 * M_PI * k / n2; t[k] = make_float2((float)cos(a), (float)sin(a)); }
    if (P.m == 3) for (int k = 0; k < n2; k++) { const double a = -2.0 * M_PI * k / P.n; t[n2 + k] = make_float2((float)cos(a), (float)sin(a)); }
    return t;
}
static std::vector<int> spec_positions(const KFFTPlan &Q) { std::vector<int> p(Q.n); for (int k = 0; k < Q.n; k++) p[k] = spec_pos(Q, k); return p; }
static std::vector<int> spec_inverse(const std::vector<int> &p) { std::vector<int> q(p.size()); for (size_t k = 0; k < p.size(); k++) q[p[k]] = (int)k; return q; }
// per POSITION of the transform's output: (position of the wavenumber -k, bit pattern of the symbol l[k]), k = the wavenumber at that position
static std::vector<int2> spec_partner_table(const KFFTPlan &Q, const std::vector<float> &l)
{
    const std::vector<int> pos = spec_positions(Q), kof = spec_inverse(pos);
    std::vector<int2> t(Q.n);
    for (int m = 0; m < Q.n; m++) {
        const int k = kof[m];
        int bits;
        memcpy(&bits, &l[k], sizeof bits);
        t[m] = make_int2(pos[(Q.n - k) % Q.n], bits);
    }
    return t;
}
static std::vector<float> spec_symbol_table(int n, double inv_h2);
static std::vector<float> spec_symbol_table(int n, double inv_h2)
{
    std::vector<float> l(n);
    for (int k = 0; k < n; k++) { const double th = 2.0 * M_PI * k / n; l[k] = (float)((-30.0 + 32.0 * cos(th) - 2.0 * cos(2.0 * th)) / 12.0 * inv_h2); }
    return l;
}

