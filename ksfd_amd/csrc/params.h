// Kernel-argument structs shared by all gfx950 kernels of libksfd_hip.so.
#pragma once
#include <stdint.h>

#define KSFD_MAXL 12

// Geometry of the local slab as the kernels see it.
// Device layout of a vector: F field planes; a plane is (sloc + 2*ng) "slow units" of `inner`
// doubles (1-D: unit = 1 point, 2-D: unit = a row of nx, 3-D: unit = an xy plane of nx*ny),
// x fastest.  ng = 0 on a single GPU (periodic wrap is done with modular indices in the kernels),
// ng = 2 ghost units per side under slab decomposition (filled by the halo exchange).
struct KGeom {
    int dim;
    int F;            // nlig + 1
    int ng;           // ghost slow-units per side
    int wrap_slow;    // 1: slow axis wraps inside the kernel (single rank)
    long long nx, ny, nz;   // local extents; the slow axis holds sloc
    long long inner;  // doubles per slow unit
    long long sloc;   // owned slow units
    long long plane;  // doubles per field plane incl. ghosts
    long long nloc;   // owned points per field = sloc*inner
};

// Physics constants (ps.values(t) of the reference, numerically folded).
struct KPhys {
    int nlig, ngroups, cap_kind, pad;
    double inv_h[3], inv_h2[3];
    double s2, rhomax, inv_cushion, ms /* maxscale*s2 */, rhomin, Umin, inv_rhomax;
    int lig_group[KSFD_MAXL];
    double lig_w[KSFD_MAXL], lig_s[KSFD_MAXL], lig_gamma[KSFD_MAXL], lig_D[KSFD_MAXL];
    double grp_alpha[KSFD_MAXL], grp_beta[KSFD_MAXL];
};

// 4th-order central differences on {-2,-1,0,1,2} (KSFD/ksfdsym.py:391-436):
//   f'  = ( 8 (f[+1]-f[-1]) - (f[+2]-f[-2]) ) / (12 h)
//   f'' = ( 16 (f[+1]+f[-1]) - (f[+2]+f[-2]) - 30 f[0] ) / (12 h^2)
#define KSFD_D1(fm2, fm1, fp1, fp2) ((8.0 * ((fp1) - (fm1)) - ((fp2) - (fm2))) * (1.0 / 12.0))
#define KSFD_D2(fm2, fm1, f0, fp1, fp2) ((16.0 * ((fp1) + (fm1)) - ((fp2) + (fm2)) - 30.0 * (f0)) * (1.0 / 12.0))
