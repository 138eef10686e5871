// libksfd_hip.so -- Chebyshev polynomial preconditioner, flexible/recycled GMRES(m), pipelined GMRES
// (part of the single translation unit ksfd_hip.hip; included from there in this order:
//  handle.hip.h, ops.hip.h, mg_host.hip.h, krylov.hip.h)
#pragma once
// ------------------------------------------------------------------------------------------------
// Polynomial preconditioner.  In the non-stiff regime (h*gamma*lambda_max(J) of order 1..10, the regime of the
// headline benchmark) plain GMRES needs ~7 iterations per stage and spends most of its time in Gram-Schmidt, whose
// traffic grows with the square of the iteration count.  z = p(A) v with p the degree-d Chebyshev approximation of
// 1/lambda on [a, b] (spectrum of A/shift: a ~ 1, b = 1 + lambda_max(-J)/shift) costs d Jacobian actions with a fused
// Horner epilogue (out = alpha*v + beta*A t, no extra pass) and cuts the outer iterations to 2-3: same number of
// Jacobian actions, a fraction of the Gram-Schmidt passes.  Used through flexible GMRES (Z basis kept), so the
// solution update needs no extra preconditioner application.
// ------------------------------------------------------------------------------------------------
static int est_lambda_max(ksfd_handle *h, double shift, int nits)
{
    int rc;
    if (!h->pvec) {
        if (alloc_d(h, &h->pvec, h->vlen)) return KSFD_ENOMEM;
        int nb = (int)std::min<long long>((h->vlen + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        hipLaunchKernelGGL(k_hash_fill, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, (long long)h->vlen, h->pvec);
        if ((rc = op_multidot(h, h->pvec, h->pvec, 0))) return rc;
        const double n0 = sqrt(h->hres[0]);
        const double *xs[1] = { h->pvec }; double a[1] = { 1.0 / n0 };
        if ((rc = op_lincomb(h, 1, xs, a, h->pvec))) return rc;
    }
    double lamA = 0.0;
    for (int it = 0; it < nits; it++) {
        if ((rc = op_jvp_frozen_halo(h, h->pvec, 1, shift, h->t3))) return rc;
        if ((rc = op_multidot(h, h->t3, h->t3, 0))) return rc;
        lamA = sqrt(h->hres[0]);
        if (!(lamA > 0.0) || lamA != lamA) return fail(h, KSFD_ENAN, "power iteration on the Jacobian broke down");
        const double *xs[1] = { h->t3 }; double a[1] = { 1.0 / lamA };
        if ((rc = op_lincomb(h, 1, xs, a, h->pvec))) return rc;
    }
    const double est = lamA - shift;
    h->lamJ = est > 0.0 ? est : 0.0;
    return KSFD_OK;
}

// coefficients of p for this shift; degree 0 = "do not precondition"
static void poly_setup(ksfd_handle *h, double shift)
{
    const double a = 0.97, b = 1.0 + 1.15 * h->lamJ / shift;      // spectrum of A/shift (power iteration converges from below: +15 %)
    h->poly_shift = shift;
    h->poly_deg = 0;
    const double kappa = b / a;
    if (kappa < 1.3) return;                                       // GMRES alone needs <= 3 iterations
    const double rc_ = (sqrt(kappa) - 1.0) / (sqrt(kappa) + 1.0);
    int d = (int)ceil(log(h->poly_target) / log(rc_)) - 1;         // residual polynomial of degree d+1: ~2 rc^(d+1) <= 2*target
    d = std::min(std::max(d, 1), std::max(h->poly_max_deg, 1));
    // r(l) = T_{d+1}(mu(l)) / T_{d+1}(mu(0)), mu(l) = m0 + m1 l;  p(l) = (1 - r(l)) / l
    const int n = d + 1;
    double m0 = (b + a) / (b - a), m1 = -2.0 / (b - a);
    double Tp[10] = { 1.0 }, Tc[10] = { m0, m1 }, Tn[10];
    int degc = 1;
    for (int k = 1; k < n; k++) {
        for (int i = 0; i < 10; i++) Tn[i] = 0.0;
        for (int i = 0; i <= degc; i++) { Tn[i] += 2.0 * m0 * Tc[i]; Tn[i + 1] += 2.0 * m1 * Tc[i]; }
        for (int i = 0; i <= degc - 1; i++) Tn[i] -= Tp[i];
        for (int i = 0; i < 10; i++) { Tp[i] = Tc[i]; Tc[i] = Tn[i]; }
        degc++;
    }
    const double t0 = Tc[0];                                       // T_n(mu(0))
    for (int i = 0; i <= d; i++) h->poly_alpha[i] = -(Tc[i + 1] / t0) / shift;   // p_i = -r_{i+1}; the 1/shift turns p(A/shift) into ~A^-1
    h->poly_deg = d;
}

// z = sum_i alpha_i (A/shift)^i v  by Horner, one fused Jacobian action per degree
static int poly_apply(ksfd_handle *h, double shift, double *v, double *z)
{
    int rc;
    const int d = h->poly_deg;
    const double *al = h->poly_alpha;
    if (h->poly_fp32 && h->coef32 && fused_ok(h)) {
        // mixed precision: the Horner temporaries and the coefficient planes live in fp32 (half the traffic of every
        // application but the arithmetic stays fp64); v is read and z written in fp64.  p(A) becomes a slightly
        // different fixed linear operator, which flexible GMRES does not care about: w_j = A z_j is computed in fp64
        // from the stored z_j, so the Arnoldi relation and the solution keep full accuracy.
        const float *C = h->coef32;
        float *tf[2] = { reinterpret_cast<float *>(h->t1), reinterpret_cast<float *>(h->t2) };
        const double *nod = nullptr;
        if (d == 1) return jvp2d_halo_t<float, double, double, double>(h, C, v, 4, shift, z, nod, al[0], al[1] / shift);
        if ((rc = jvp2d_halo_t<float, double, double, float>(h, C, v, 4, shift, tf[0], nod, al[d - 1], al[d] / shift))) return rc;
        int cur32 = 0;
        for (int i = d - 2; i >= 1; i--) {
            if ((rc = jvp2d_halo_t<float, float, double, float>(h, C, tf[cur32], 3, shift, tf[cur32 ^ 1], (const double *)v, al[i], 1.0 / shift))) return rc;
            cur32 ^= 1;
        }
        return jvp2d_halo_t<float, float, double, double>(h, C, tf[cur32], 3, shift, z, (const double *)v, al[0], 1.0 / shift);
    }
    double *tmp[2] = { h->t1, h->t2 };
    // t_{d-1} = alpha_{d-1} v + (alpha_d/shift) A v
    double *cur = (d == 1) ? z : tmp[0];
    if ((rc = op_jvp_frozen_halo(h, v, 4, shift, cur, nullptr, al[d - 1], al[d] / shift))) return rc;
    int flip = 1;
    for (int i = d - 2; i >= 0; i--) {
        double *nxt = (i == 0) ? z : tmp[flip];
        if ((rc = op_jvp_frozen_halo(h, cur, 3, shift, nxt, v, al[i], 1.0 / shift))) return rc;
        cur = nxt;
        flip ^= 1;
    }
    return KSFD_OK;
}

// ------------------------------------------------------------------------------------------------
// matrix-free GMRES(m) for (shift I - J(u)) x = b, x0 = 0  -- replaces -ksp_type preonly -pc_type lu
// (options84:58-60).  Classical Gram-Schmidt applied twice (CGS2), one fused multi-dot + one fused
// update kernel per pass; the new vector's norm comes from the second pass by Pythagoras.
// ------------------------------------------------------------------------------------------------
struct LinStats { int its; double rel; };

static void rec_reset(ksfd_handle *h)
{
    for (auto &r : h->rec) r.valid = false;
    h->rec_vtop = h->rec_ztop = 0;
}

// Least squares min ||g - H y|| for a small upper-Hessenberg H ((k+1) x k, column-major, ld = k+1); also returns H y.
static void hess_lsq(const double *H, int k, const double *g, double *y, double *Hy)
{
    const int ld = k + 1;
    std::vector<double> R((size_t)ld * k), q((size_t)k + 1);
    for (int i = 0; i < ld * k; i++) R[i] = H[i];
    for (int i = 0; i <= k; i++) q[i] = g[i];
    for (int j = 0; j < k; j++) {
        const double a = R[j * ld + j], b = R[j * ld + j + 1], den = hypot(a, b);
        const double c = den > 0.0 ? a / den : 1.0, sn = den > 0.0 ? b / den : 0.0;
        for (int l = j; l < k; l++) {
            const double t = c * R[l * ld + j] + sn * R[l * ld + j + 1];
            R[l * ld + j + 1] = -sn * R[l * ld + j] + c * R[l * ld + j + 1];
            R[l * ld + j] = t;
        }
        const double t = c * q[j] + sn * q[j + 1];
        q[j + 1] = -sn * q[j] + c * q[j + 1];
        q[j] = t;
    }
    for (int i = k - 1; i >= 0; i--) {
        double t = q[i];
        for (int l = i + 1; l < k; l++) t -= R[l * ld + i] * y[l];
        y[i] = R[i * ld + i] != 0.0 ? t / R[i * ld + i] : 0.0;
    }
    for (int i = 0; i <= k; i++) {
        double t = 0.0;
        for (int l = 0; l < k; l++) t += H[l * ld + i] * y[l];
        Hy[i] = t;
    }
}

// stage >= 0: Krylov recycling.  The four stage systems of a step share the matrix, and their right-hand sides are
// nearly linear images of one another (b_i = f(u + sum a_ij Y_j) - sum c_ij Y_j/h with f almost linear over a step), so
// the leading Arnoldi vectors of an earlier stage (A Z_s = V_s H_s, kept in place at the front of V / Zb) already span
// most of the new solution: x0 = Z_s y with y = argmin ||V_s^T r - H_s y||, r <- r - V_s H_s y, one space after the
// other, costs ~5 vector passes per kept vector and removes 1-2 of the 3-4 outer iterations of stages 2-4 (each
// (d+1) Jacobian actions + Gram-Schmidt).  The iteration then continues on the true residual with the same stopping
// test, so the result is the same to the solver tolerance.  stage < 0: plain solve from x0 = 0.
static int gmres(ksfd_handle *h, const double *ustate, double shift, const double *b, double *x,
                 const ksfd_step_opts *o, LinStats *ls, int pcmode, int stage = -1, double tol_abs = -1.0)
{
    // tol_abs > 0: stop at that absolute residual norm (the caller solves a correction equation A d = b - A x0 and wants the
    // tolerance of the original system)
    const bool use_pc = pcmode == 1;       // multigrid, right preconditioning
    // the hierarchy may be built for a LARGER shift than the system's (h->mg_shift_floor): when 1/(gamma h) falls below the growth
    // rate of the chemotactic instability, shift*I - J is indefinite and a V cycle of it is no contraction; the V cycle of the
    // positively shifted operator still is, and GMRES (true residual of the real system) takes care of the difference
    const double shift_pc = std::max(shift, h->mg_shift_floor);
    const bool use_poly = pcmode == 2 || pcmode == 3;     // flexible GMRES (z_j = M^-1 v_j kept in Zb): 2 Chebyshev polynomial p(A), 3 spectral (spectral_host.hip.h)
    // use_pc: right preconditioning with one multigrid V cycle, w = A (M^-1 v_j), x = M^-1 (V y)
    auto apply_A = [&](const double *vin, double *wout) -> int {
        return h->use_frozen ? op_jvp_frozen(h, vin, 1, shift, wout) : op_jvp(h, ustate, vin, 1, shift, wout);
    };
    const int m_opt = std::min(o->ksp_restart > 0 ? o->ksp_restart : 30, h->restart_alloc);
    const int maxit = o->ksp_max_it > 0 ? o->ksp_max_it : 2000;
    const int64_t vs = h->vlen;
    // With the multigrid preconditioner the LEADING vectors of an earlier stage buy nothing (measured on the 600-step 384^2 run:
    // 20.4 s with, 18.9 s without).  Round 3 tried the other end (rec_full, KSFD_TUNE bit 20, off by default): keep the WHOLE first
    // cycle of every stage -- the slow modes of shift*I - J that take the iterations late in a run sit in the tail of the Krylov
    // space -- and project a later stage's right-hand side on A M^-1 V_k = V_k+1 H_k first (one multi-dot, one basis combination,
    // ONE V cycle for all spaces together).  Measured: it does NOT pay -- aggregated state at 4096^2 x 3 (h = 4.4) 25 instead of 26
    // iterations per step but 142 instead of 125 ms; indefinite tail of the 384^2 run (h = 400, tools/late_phase.py) 142 instead of
    // 135 iterations per step.  What a later stage still has to resolve is not in the span of what an earlier one built.
    const bool rec_full = use_pc && h->rec_mg && stage >= 0 && stage < 4 && h->use_frozen && h->rec_mode > 0;
    bool rec_on = stage >= 0 && stage < 4 && h->rec_mode > 0 && h->use_frozen && (!use_pc || rec_full);
    if (!rec_on || stage == 0 || h->restart_alloc - h->rec_vtop < (rec_full ? 16 : 6)) { if (stage != 0) rec_on = false; rec_reset(h); }
    const int vb = rec_on ? h->rec_vtop : 0, zb = rec_on ? h->rec_ztop : 0;
    // Restart length.  The first cycle runs with ksp_restart (30: PETSc's default); a cycle that ends without convergence is followed by
    // one of twice the length, up to what ksfd_create could allocate (restart_alloc, <= 120).  Restarted GMRES loses most on exactly the
    // systems where it needs many iterations -- shift*I - J indefinite late in a run, 30-50 iterations per stage system -- and a longer
    // basis costs little next to the V cycle and Jacobian actions of an iteration there.  Host arrays are sized for the longest cycle.
    const int m = h->restart_alloc - vb;
    int m_cur = (rec_full && rec_on) ? m : std::min(m_opt, m);          // rec_full: no restart inside the space that is going to be kept
    double *V = h->V + (int64_t)vb * vs;
    double *Zq = use_poly ? h->Zb + (int64_t)zb * vs : nullptr;
    int rc;
    std::vector<double> H((size_t)(m + 1) * m, 0.0), Hraw((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m), hcol(m + 2), d(m + 2), Gm((size_t)(m + 1) * (m + 1), 0.0);
    // ||b||: with a recycled space to project on, the projection's multi-dot of b returns <b,b> as well -- one pass, one
    // reduction and one host round trip less per stage
    static const int sel[4][3] = { { -1, -1, -1 }, { 0, -1, -1 }, { 0, -1, -1 }, { 0, 2, -1 } };
    int first_space = -1;
    std::vector<double> g_first((size_t)h->restart_alloc + 3, 0.0);
    if (rec_on && stage > 0)
        for (int q = 0; q < stage && first_space < 0; q++) {
            bool use = h->rec_mode == 2 || rec_full;
            for (int e = 0; e < 3; e++) use = use || sel[stage][e] == q;
            if (use && h->rec[q].valid && h->rec[q].pc == pcmode) first_space = q;
        }
    if (first_space >= 0) {
        const ksfd_handle::RecSpace &S = h->rec[first_space];
        if ((rc = op_multidot(h, b, h->V + (int64_t)S.vb * vs, S.k + 1))) return rc;
        for (int i = 0; i <= S.k + 1; i++) g_first[i] = h->hres[i];
    } else if ((rc = op_multidot(h, b, V, 0))) return rc;
    const double bn = sqrt(first_space >= 0 ? g_first[h->rec[first_space].k + 1] : h->hres[0]);
    ls->its = 0; ls->rel = 0.0;
    if (!(bn > 0.0)) {
        if (bn != bn) return fail(h, KSFD_ENAN, "GMRES: right-hand side is not finite");
        HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)vs, h->st));
        return KSFD_OK;
    }
    const double tol = tol_abs > 0.0 ? tol_abs : std::max(o->ksp_rtol * bn, o->ksp_atol);
    double beta = bn, rn = bn;
    int total = 0;
    bool first = true;          // the residual of the current x is at hand (rsrc, norm beta): no A x needed
    bool x_set = false;         // x holds an iterate (else it is taken as 0 and overwritten)
    bool restarted = false;
    const double *rsrc = b;
    // x0 from the recycled spaces is not written on its own: its coefficients wait (indexed by absolute slot of Zb / V) and
    // ride in the solution update of the first cycle -- one pass over x instead of two
    const bool defer_x0 = rec_on && !use_pc;
    std::vector<double> xcoef((size_t)std::max(h->restart_alloc + 2, KSFD_MAXDOT), 0.0);
    bool x0_pending = false;
    double *const Xbase = use_poly ? h->Zb : h->V;          // slot 0 of the basis the solution is expanded in
    const int xslot0 = use_poly ? zb : vb;                  // first slot of this solve's own vectors
    bool pc_x0_pending = false;
    double *const rbuf = (x == h->t3) ? V : h->t3;           // projected residual (a correction solve has its x in t3: this solve's slot 0 then, scaled in place below)
    if (rec_on && stage > 0) {
        int last_space = -1;                                  // its residual update also returns the norm of the result
        for (int q = 0; q < stage; q++) {
            bool use = h->rec_mode == 2 || rec_full;
            for (int e = 0; e < 3; e++) use = use || sel[stage][e] == q;
            if (use && h->rec[q].valid && h->rec[q].pc == pcmode) last_space = q;
        }
        bool have_norm = false;
        for (int q = 0; q < stage; q++) {
            bool use = h->rec_mode == 2 || rec_full;
            for (int e = 0; e < 3; e++) use = use || sel[stage][e] == q;
            const ksfd_handle::RecSpace &S = h->rec[q];
            if (!use || !S.valid || S.pc != pcmode) continue;
            const double *Vs = h->V + (int64_t)S.vb * vs;
            const double *Zs = use_poly ? h->Zb + (int64_t)S.zb * vs : Vs;
            std::vector<double> gq_((size_t)S.k + 2), yq_((size_t)S.k + 1), Hy_((size_t)S.k + 2), neg_((size_t)S.k + 2);
            double *gq = gq_.data(), *yq = yq_.data(), *Hy = Hy_.data(), *neg = neg_.data();
            if (q == first_space && rsrc == b) {
                for (int i = 0; i <= S.k; i++) gq[i] = g_first[i];                 // already computed together with ||b||
            } else {
                if ((rc = op_multidot(h, rsrc, Vs, S.k + 1))) return rc;
                for (int i = 0; i <= S.k; i++) gq[i] = h->hres[i];
            }
            hess_lsq(S.H.data(), S.k, gq, yq, Hy);
            if (use_pc) {
                // x0 = M^-1 (sum over the spaces of V_s y_s): the combinations gather in t2, one V cycle behind the loop
                if ((rc = op_basis_axpy(h, h->t2, Vs, S.k, yq, pc_x0_pending ? 1.0 : 0.0))) return rc;
                pc_x0_pending = true;
            } else if (defer_x0) {
                for (int i = 0; i < S.k; i++) xcoef[(use_poly ? S.zb : S.vb) + i] += yq[i];
                x0_pending = true;
            } else {
                if ((rc = op_basis_axpy(h, x, Zs, S.k, yq, x_set ? 1.0 : 0.0))) return rc;
                x_set = true;
            }
            for (int i = 0; i <= S.k; i++) neg[i] = -Hy[i];
            if (rsrc == b && S.k + 2 <= 6) {
                const double *xs[6] = { b }; double a[6] = { 1.0 };
                for (int i = 0; i <= S.k; i++) { xs[i + 1] = Vs + (int64_t)i * vs; a[i + 1] = neg[i]; }
                if ((rc = op_lincomb(h, S.k + 2, xs, a, rbuf, q == last_space))) return rc;
                rsrc = rbuf;
                have_norm = q == last_space;
            } else {
                if (rsrc == b) { if ((rc = op_copy(h, rbuf, b))) return rc; rsrc = rbuf; }
                const bool nrm = q == last_space;
                if ((rc = op_basis_axpy(h, rbuf, Vs, S.k + 1, neg, 1.0, nrm))) return rc;
                have_norm = nrm;
            }
        }
        if (pc_x0_pending) {
            if ((rc = mg_precond(h, shift_pc, h->t2, h->t1)) || (rc = op_copy(h, x, h->t1))) return rc;
            x_set = true;
        }
        if (x_set || x0_pending) {
            if (!have_norm && (rc = op_multidot(h, rsrc, rsrc, 0))) return rc;
            beta = rn = sqrt(h->hres[0]);
            if (!(beta == beta)) return fail(h, KSFD_ENAN, "GMRES: projected residual is not finite");
        }
    }
    while (true) {
        if (first && beta <= tol) {                            // the recycled spaces already hold the solution
            if (x0_pending) {
                if ((rc = op_basis_axpy(h, x, Xbase, xslot0, xcoef.data(), x_set ? 1.0 : 0.0))) return rc;
                x0_pending = false; x_set = true;
            }
            break;
        }
        // V0 = r / beta
        if (first) { const double *xs[1] = { rsrc }; double a[1] = { 1.0 / beta }; if ((rc = op_lincomb(h, 1, xs, a, V))) return rc; }
        else {
            if ((rc = halo(h, x)) || (rc = apply_A(x, V))) return rc;     // V0 = A x
            const double *xs[2] = { b, V }; double a[2] = { 1.0, -1.0 };
            if ((rc = op_lincomb(h, 2, xs, a, V))) return rc;                                  // r = b - A x
            if ((rc = op_multidot(h, V, V, 0))) return rc;
            beta = sqrt(h->hres[0]);
            rn = beta;
            if (!(beta == beta)) return fail(h, KSFD_ENAN, "GMRES: residual is not finite");
            if (beta <= tol) break;
            const double *x1[1] = { V }; double a1[1] = { 1.0 / beta };
            if ((rc = op_lincomb(h, 1, x1, a1, V))) return rc;
        }
        std::fill(g.begin(), g.end(), 0.0);
        g[0] = beta;
        int j = 0;
        bool done = false;
        for (; j < m_cur && total < maxit; j++) {
            double *vj = V + (int64_t)j * vs, *w = V + (int64_t)(j + 1) * vs;
            if (use_pc) {
                if ((rc = mg_precond(h, shift_pc, vj, h->t1)) || (rc = op_jvp_frozen_halo(h, h->t1, 1, shift, w))) return rc;
            } else if (use_poly) {
                double *zj = Zq + (int64_t)j * vs;
                if ((rc = pcmode == 3 ? spec_apply(h, shift, vj, zj) : poly_apply(h, shift, vj, zj)) || (rc = op_jvp_frozen_halo(h, zj, 1, shift, w))) return rc;
            } else if (h->use_frozen) {
                if ((rc = op_jvp_frozen_halo(h, vj, 1, shift, w))) return rc;
            } else if ((rc = halo(h, vj)) || (rc = apply_A(vj, w))) return rc;
            const int k = j + 1;
            if ((o->reserved & 1) || k > 30) {       // (the fused multi-dot + Gram-row kernel returns 2k + 1 numbers, sized for the 30 vectors of the default restart: longer cycles finish classically)
                // classic CGS2: two Gram-Schmidt passes, each = one fused multi-dot + one fused update.
                // (One pass alone loses orthogonality like eps*(||r0||/||r_j||)^2 and stalls near 1e-8.)
                if ((rc = op_multidot(h, w, V, k))) return rc;
                for (int i = 0; i < k; i++) hcol[i] = h->hres[i];
                if (!(h->hres[k] == h->hres[k])) return fail(h, KSFD_ENAN, "GMRES: Krylov vector is not finite");
                if ((rc = op_gs_update(h, w, V, k, hcol.data(), 1.0))) return rc;
                if ((rc = op_multidot(h, w, V, k))) return rc;
                double s2 = 0.0;
                for (int i = 0; i < k; i++) { d[i] = h->hres[i]; hcol[i] += d[i]; s2 += d[i] * d[i]; }
                double hn2 = h->hres[k] - s2;          // ||w''||^2 by Pythagoras; d is O(eps) so this is accurate
                if (hn2 < 0.0) hn2 = 0.0;
                const double hn = sqrt(hn2);
                if ((rc = op_gs_update(h, w, V, k, d.data(), hn > 0.0 ? 1.0 / hn : 0.0))) return rc;
                hcol[k] = hn;
            } else {
                // CGS2 with the second projection done algebraically (halves the Gram-Schmidt traffic):
                //   d = V^T w and the Gram row g = V^T v_j come from ONE pass over V; with G = V^T V,
                //   the twice-projected coefficients are c = d + (I - G) d, and
                //   ||w - V c||^2 = ww - 2 c.d + c.G c.   One fused update pass applies c and normalises.
                if ((rc = op_multidot_gram(h, w, V, k))) return rc;
                for (int i = 0; i < k; i++) { d[i] = h->hres[i]; Gm[(size_t)i * (m + 1) + j] = Gm[(size_t)j * (m + 1) + i] = h->hres[k + i]; }
                const double ww = h->hres[2 * k];
                if (!(ww == ww)) return fail(h, KSFD_ENAN, "GMRES: Krylov vector is not finite");
                for (int i = 0; i < k; i++) {
                    double s = 0.0;
                    for (int l = 0; l < k; l++) s += ((i == l ? 1.0 : 0.0) - Gm[(size_t)i * (m + 1) + l]) * d[l];
                    hcol[i] = d[i] + s;
                }
                double cd = 0.0, cGc = 0.0;
                for (int i = 0; i < k; i++) {
                    cd += hcol[i] * d[i];
                    double s = 0.0;
                    for (int l = 0; l < k; l++) s += Gm[(size_t)i * (m + 1) + l] * hcol[l];
                    cGc += hcol[i] * s;
                }
                double hn2 = ww - 2.0 * cd + cGc;
                double hn;
                if (hn2 > 1e-8 * ww) {
                    hn = sqrt(hn2);
                    if ((rc = op_gs_update(h, w, V, k, hcol.data(), 1.0 / hn))) return rc;
                } else {
                    // heavy cancellation (||w|| >> ||w - Vc||): apply c, then measure and project once more
                    if ((rc = op_gs_update(h, w, V, k, hcol.data(), 1.0))) return rc;
                    if ((rc = op_multidot(h, w, V, k))) return rc;
                    double s2 = 0.0;
                    for (int i = 0; i < k; i++) { d[i] = h->hres[i]; hcol[i] += d[i]; s2 += d[i] * d[i]; }
                    hn2 = h->hres[k] - s2;
                    if (hn2 < 0.0) hn2 = 0.0;
                    hn = sqrt(hn2);
                    if ((rc = op_gs_update(h, w, V, k, d.data(), hn > 0.0 ? 1.0 / hn : 0.0))) return rc;
                }
                hcol[k] = hn;
            }
            double *Hc = &H[(size_t)(m + 1) * j];
            for (int i = 0; i <= k; i++) Hraw[(size_t)(m + 1) * j + i] = Hc[i] = hcol[i];
            for (int i = 0; i < j; i++) { double t = cs[i] * Hc[i] + sn[i] * Hc[i + 1]; Hc[i + 1] = -sn[i] * Hc[i] + cs[i] * Hc[i + 1]; Hc[i] = t; }
            const double den = hypot(Hc[j], Hc[j + 1]);
            cs[j] = den > 0.0 ? Hc[j] / den : 1.0;
            sn[j] = den > 0.0 ? Hc[j + 1] / den : 0.0;
            Hc[j] = den; Hc[j + 1] = 0.0;
            g[j + 1] = -sn[j] * g[j];
            g[j] = cs[j] * g[j];
            total++;
            rn = fabs(g[j + 1]);
            if (rn <= tol || hcol[k] == 0.0) { j++; done = true; break; }
        }
        for (int i = j - 1; i >= 0; i--) {
            double s = g[i];
            for (int q = i + 1; q < j; q++) s -= H[(size_t)(m + 1) * q + i] * y[q];
            y[i] = s / H[(size_t)(m + 1) * i + i];
        }
        if (use_pc) {
            if ((rc = op_basis_axpy(h, h->t2, V, j, y.data(), 0.0)) || (rc = mg_precond(h, shift_pc, h->t2, h->t1))) return rc;
            if (!x_set) { if ((rc = op_copy(h, x, h->t1))) return rc; }
            else { const double *xs[2] = { x, h->t1 }; double a2[2] = { 1.0, 1.0 }; if ((rc = op_lincomb(h, 2, xs, a2, x))) return rc; }
        } else if (x0_pending) {
            for (int i = 0; i < j; i++) xcoef[xslot0 + i] = y[i];
            if ((rc = op_basis_axpy(h, x, Xbase, xslot0 + j, xcoef.data(), x_set ? 1.0 : 0.0))) return rc;
            x0_pending = false;
        } else if ((rc = op_basis_axpy(h, x, use_poly ? Zq : V, j, y.data(), x_set ? 1.0 : 0.0))) return rc;
        x_set = true;
        if (rec_on && first && !restarted && done && j >= 1) {
            // keep the leading vectors of this stage's Arnoldi relation where they are; the next stage builds behind them
            ksfd_handle::RecSpace &S = h->rec[stage];
            S.k = rec_full ? j : std::min(j, std::min(h->rec_keep, 4));
            S.vb = vb; S.zb = zb; S.pc = pcmode;
            S.H.assign((size_t)(S.k + 1) * S.k, 0.0);
            for (int c = 0; c < S.k; c++)
                for (int i = 0; i <= S.k; i++) S.H[c * (S.k + 1) + i] = Hraw[(size_t)(m + 1) * c + i];
            S.valid = true;
            h->rec_vtop = vb + S.k + 1;
            h->rec_ztop = zb + (use_poly ? S.k : 0);
        }
        if (!first) restarted = true;
        first = false;
        if (done || total >= maxit) break;
        restarted = true;
        if (h->restart_grow) m_cur = std::min(2 * m_cur, m);
    }
    if (!x_set) HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)vs, h->st));
    ls->its = total;
    ls->rel = rn / ((tol_abs > 0.0 && o->ksp_rtol > 0.0) ? tol_abs / o->ksp_rtol : bn);      // correction equation: relative to the original right-hand side
    if (rn > tol) return fail(h, KSFD_ELINEAR, "GMRES did not converge: %d iterations, relative residual %.3e (tol %.3e)", total, rn / bn, tol / bn);
    return KSFD_OK;
}

// ------------------------------------------------------------------------------------------------
// Stage solve with the spectral preconditioner.  M = shift*I - J0 is so close to A = shift*I - J on near-uniform states
// (||I - A M^-1|| ~ 0.01: tests/experiments/fft_pc_experiment.py) that plain defect correction
//      x <- x + M^-1 (b - A x)
// converges as fast as GMRES would (3-4 iterations to 1e-6) and needs NO Krylov vectors: an iteration is one spectral
// application (its last kernel adds into x) and one Jacobian action in residual mode whose store epilogue also leaves
// ||r||^2 -- no BLAS-1 pass at all, where GMRES spends ~40 % of such a step in Gram-Schmidt.  Stops on the TRUE residual,
// ||b - A x|| <= max(ksp_rtol ||b||, ksp_atol), like every other solver here.  If the contraction is worse than 0.25 per
// sweep (coefficients vary too much for the constant-coefficient inverse), the remaining correction A d = r is handed to
// flexible GMRES with the same preconditioner, and if that fails too the caller falls back to the V cycle.
// bnorm2 >= 0: ||b||^2 if the caller already has it (RHS kernel epilogue), else it is computed here.
// ------------------------------------------------------------------------------------------------
// guess != NULL: the first sweep starts from x0 = sum_j c_j Y_j (earlier stage solutions of the same step, A Y_j = b_j):
//   x = x0 + M^-1 (b - sum_j c_j b_j)   -- no extra Jacobian action, the vectors ride in the row kernels of the application
static int spec_solve(ksfd_handle *h, double shift, const double *b, double *x, const ksfd_step_opts *o, LinStats *ls, double bnorm2, int gmres_cap,
                      const SpecGuess *guess = nullptr)
{
    int rc;
    const int64_t vs = h->vlen;
    ls->its = 0; ls->rel = 0.0;
    rec_reset(h);
    if (bnorm2 < 0.0) {
        if ((rc = op_multidot(h, b, b, 0))) return rc;
        bnorm2 = h->hres[0];
    }
    const double bn = sqrt(bnorm2);
    if (!(bn > 0.0)) {
        if (bn != bn) return fail(h, KSFD_ENAN, "spectral solve: right-hand side is not finite");
        HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)vs, h->st));
        return KSFD_OK;
    }
    const double tol = std::max(o->ksp_rtol * bn, o->ksp_atol);
    const int maxit = std::min(o->ksp_max_it > 0 ? o->ksp_max_it : 2000, 24);
    double *r = h->Z;                              // residual (the fused-stage path leaves Z unused)
    float *r32 = reinterpret_cast<float *>(h->Z);  // ... kept in fp32 while only the preconditioner reads it
    double rn = bn, rprev = bn;
    // fp32 residual with the norm from the store epilogue: the 2-D strip kernel, or the 3-D one when its wave count fits the partial buffer
    const bool fused = fused_ok(h) || (strip3d_ok(h) && (j3l_ok(h) ? (long long)make_k3d_lds(h).nblocks * KSFD_J3L_ROWS : (long long)make_k3d(h).nblocks * make_k3d(h).rows) <= part_capacity());
    bool slow = false;
    // Predicted last sweep.  Every sweep multiplies the residual by (I - A M^-1); the four stage systems of a step share that
    // operator, so its contraction has been MEASURED by the time a solve is about to finish: rho_hat = the largest ratio
    // ||r_k+1|| / ||r_k|| of any sweep (but the first one from zero, see below) of this step and the previous one.  When SAFETY * rho_hat * ||r_k|| <= tol the update
    // x += M^-1 r_k is applied and the solve returns WITHOUT evaluating the residual of the result (a Jacobian action, 40 % of a
    // sweep, whose only purpose would be to confirm a number 1-2 decades below the tolerance).  Every other residual is the true
    // fp64 one, as before.  Off for tight tolerances (ksp_rtol < 1e-8: the parity tests verify every solve), with opts.reserved
    // bit 3, and with KSFD_SPEC_VERIFY=1, which also evaluates and reports the residual the prediction stood in for.
    static const int verify_env = getenv("KSFD_SPEC_VERIFY") ? atoi(getenv("KSFD_SPEC_VERIFY")) : 0;
    const bool predict = h->spec_predict && !(o->reserved & 8) && o->ksp_rtol >= 1e-8 && verify_env != 2;
    const double PRED_SAFETY = 4.0;
    bool final_pending = false;
    for (int k = 0; k < maxit; k++) {
        if (k == 0) rc = spec_apply(h, shift, b, x, nullptr, nullptr, guess);
        else rc = fused ? spec_apply(h, shift, nullptr, x, x, r32) : spec_apply(h, shift, r, x, x);
        if (rc) return rc;
        ls->its++;
        if (final_pending) {
            const double rho_hat = std::max(h->spec.rho_step, h->spec.rho_prev);
            ls->rel = PRED_SAFETY * rho_hat * rn / bn;                            // the bound the decision was made on
            h->n_predicted++;
            if (verify_env == 1) {
                if (h->ring && !(fused && h->G.dim == 2) && (rc = halo(h, x))) return rc;
                if (fused) { if ((rc = op_residual32(h, x, shift, b, r32))) return rc; }
                else if ((rc = op_jvp_frozen(h, x, 2, shift, r, b)) || (rc = op_multidot(h, r, r, 0))) return rc;
                const double rv = sqrt(h->hres[0]);
                fprintf(stderr, "[spec] predicted final sweep %d: bound %.3e true %.3e tol %.3e%s\n", k, ls->rel, rv / bn, tol / bn, rv <= tol ? "" : "  <-- ABOVE TOLERANCE");
                if (!(rv <= tol)) return fail(h, KSFD_ELINEAR, "predicted final sweep missed the tolerance: true %.3e, bound %.3e, tol %.3e", rv / bn, ls->rel, tol / bn);
            }
            return KSFD_OK;
        }
        // the spectral application wrote owned rows only; the 2-D fused residual exchanges the ghost rows itself, behind its interior rows
        if (h->ring && !(fused && h->G.dim == 2) && (rc = halo(h, x))) return rc;
        if (fused) {
            if ((rc = op_residual32(h, x, shift, b, r32))) return rc;                          // r = b - A x (fp32 copy), ||r||^2 -> hres[0]
        } else {
            if ((rc = op_jvp_frozen(h, x, 2, shift, r, b)) || (rc = op_multidot(h, r, r, 0))) return rc;
        }
        h->n_residual++;
        rn = sqrt(h->hres[0]);
        {
            static const bool trace = getenv("KSFD_SPEC_TRACE") != nullptr;        // residual history of the defect correction (diagnostics)
            if (trace) fprintf(stderr, "[spec] sweep %d rel %.3e%s\n", k, rn / bn, (k == 0 && guess) ? " (guess)" : "");
        }
        if (!(rn == rn)) return fail(h, KSFD_ENAN, "spectral solve: residual is not finite");
        if (rn <= tol) { ls->rel = rn / bn; return KSFD_OK; }
        if (rn > 0.25 * rprev) { slow = true; break; }
        // A contraction ratio of the sweep operator, taken between residuals of iterates that have been through a sweep or come from an
        // initial guess.  The first sweep from zero is a class of its own and is neither recorded nor predicted from: b = f(u) is
        // smooth, (I - A M^-1) b = dJ M^-1 b is several times larger relative to b (3-6 % on the bench window) than the same operator
        // makes of the rough residuals that follow (0.6-1 %), and after one sweep a residual never is that smooth again.
        if (k >= 1) h->spec.rho_step = std::max(h->spec.rho_step, rn / rprev);
        rprev = rn;
        const double rho_hat = std::max(h->spec.rho_step, h->spec.rho_prev);
        if (predict && rho_hat > 0.0 && (k >= 1 || guess) && k + 1 < maxit && PRED_SAFETY * rho_hat * rn <= tol) final_pending = true;
    }
    if (fused && (rc = op_jvp_frozen(h, x, 2, shift, r, b))) return rc;       // the correction solve below wants the residual in fp64
    if (!slow && rn > tol) slow = true;
    // hand the correction equation A d = r to flexible GMRES with the same preconditioner (absolute tolerance = ours)
    ksfd_step_opts go = *o;
    go.ksp_rtol = 1e-30; go.ksp_atol = tol;
    if (gmres_cap > 0) go.ksp_max_it = gmres_cap;
    const bool restart_from_zero = !(rn < bn);       // the sweeps made it worse: forget x
    LinStats g2;
    rc = gmres(h, h->u, shift, restart_from_zero ? b : r, h->t3, &go, &g2, 3);
    ls->its += g2.its;
    if (rc) { ls->rel = rn / bn; return rc; }
    if (restart_from_zero) { if ((rc = op_copy(h, x, h->t3))) return rc; }
    else { const double *xs[2] = { x, h->t3 }; double a2[2] = { 1.0, 1.0 }; if ((rc = op_lincomb(h, 2, xs, a2, x))) return rc; }
    ls->rel = g2.rel * (restart_from_zero ? 1.0 : rn / bn);
    return KSFD_OK;
}

// ------------------------------------------------------------------------------------------------
// Pipelined GMRES: same mathematics as gmres() (CGS2 with the algebraic second projection), but the small
// algebra of every iteration runs in a one-thread kernel on the device (k_gmres_coef) and the fused update reads its
// coefficients from device memory, so an iteration = [J action, multi-dot, reduce(+allreduce), coef, update] with NO
// host round trip.  The host polls the residual estimate one iteration behind (and exactly on time when the
// extrapolated estimate says "this one converges"), so the GPU never idles and at most one iteration is wasted.
// Pays when an iteration is latency-bound: small grids, many slab ranks.  Unpreconditioned, frozen Jacobian only.
// ------------------------------------------------------------------------------------------------
static int gmres_async(ksfd_handle *h, double shift, const double *b, double *x, const ksfd_step_opts *o, LinStats *ls)
{
    rec_reset(h);
    const int m = std::min(o->ksp_restart > 0 ? o->ksp_restart : 30, h->restart_alloc);
    const int maxit = o->ksp_max_it > 0 ? o->ksp_max_it : 2000;
    const int64_t vs = h->vlen;
    const int ld = h->restart_alloc + 1;
    double *V = h->V;
    double *dG = h->gm_dev, *dH = dG + (size_t)ld * ld, *dcs = dH + (size_t)ld * h->restart_alloc, *dsn = dcs + h->restart_alloc,
           *dg = dsn + h->restart_alloc, *dcoef = dg + ld, *dscale = dcoef + KSFD_MAXDOT, *dmon = dscale + 1;
    double *hmon = h->gm_host, *hH = hmon + 2 * ld, *hg = hH + (size_t)ld * h->restart_alloc;
    int rc;
    if ((rc = op_multidot(h, b, V, 0))) return rc;
    const double bn = sqrt(h->hres[0]);
    ls->its = 0; ls->rel = 0.0;
    if (!(bn > 0.0)) {
        if (bn != bn) return fail(h, KSFD_ENAN, "GMRES: right-hand side is not finite");
        HIPCHK(h, hipMemsetAsync(x, 0, sizeof(double) * (size_t)vs, h->st));
        return KSFD_OK;
    }
    const double tol = std::max(o->ksp_rtol * bn, o->ksp_atol);
    double beta = bn, rn = bn;
    int total = 0;
    bool first = true;
    std::vector<double> y(m);
    while (true) {
        if (first) { const double *xs[1] = { b }; double a[1] = { 1.0 / beta }; if ((rc = op_lincomb(h, 1, xs, a, V))) return rc; }
        else {
            if ((rc = op_jvp_frozen_halo(h, x, 1, shift, V))) return rc;
            const double *xs[2] = { b, V }; double a[2] = { 1.0, -1.0 };
            if ((rc = op_lincomb(h, 2, xs, a, V))) return rc;
            if ((rc = op_multidot(h, V, V, 0))) return rc;
            beta = sqrt(h->hres[0]);
            rn = beta;
            if (!(beta == beta)) return fail(h, KSFD_ENAN, "GMRES: residual is not finite");
            if (beta <= tol) break;
            const double *x1[1] = { V }; double a1[1] = { 1.0 / beta };
            if ((rc = op_lincomb(h, 1, x1, a1, V))) return rc;
        }
        int jc = -1, jlast = -1, checked = -1;
        double r1 = beta, r2 = -1.0;
        auto poll = [&](int upto) -> int {          // read monitors (checked, upto]; sets jc when converged
            for (int q = checked + 1; q <= upto; q++) {
                if (hipEventSynchronize(h->gm_ev[q]) != hipSuccess) return fail(h, KSFD_EHIP, "event sync failed");
                const double r = hmon[2 * q], hn = hmon[2 * q + 1];
                if (!(r == r)) return fail(h, KSFD_ENAN, "GMRES: Krylov vector is not finite");
                checked = q;
                r2 = r1; r1 = r;
                if (r <= tol || hn == 0.0) { jc = q; return KSFD_OK; }
            }
            return KSFD_OK;
        };
        for (int j = 0; j < m && total + j < maxit; j++) {
            double *vj = V + (int64_t)j * vs, *w = V + (int64_t)(j + 1) * vs;
            const int k = j + 1;
            if ((rc = op_jvp_frozen_halo(h, vj, 1, shift, w))) return rc;
            const int nb = vec2(h) ? (h->nblk_vec + 1) / 2 : h->nblk_vec;
            {
                Scope sc(h, KC_MULTIDOT, vbytes(h, k + 1));
                if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<4, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
                else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<8, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
                else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<16, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
                else VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot_gram<32, VW>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)w, (const double *)V, h->vlen, k, h->part));
            }
            {
                Scope sc(h, KC_REDUCE, 8.0 * (2 * k + 1) * (double)nb);
                hipLaunchKernelGGL(k_reduce_rows, dim3(2 * k + 1), dim3(KSFD_BLOCK), 0, h->st, (const double *)h->part, nb, 0, h->dres);
            }
            if (h->ring && h->tr->allreduce(h->dres, 2 * k + 1, 0, h->st)) return fail(h, KSFD_ECOMM, "allreduce failed: %s", h->tr->error().c_str());
            hipLaunchKernelGGL(k_gmres_coef, dim3(1), dim3(64), 0, h->st, j, h->restart_alloc, beta, (const double *)h->dres, dG, dH, dcs, dsn, dg, dcoef, dscale, dmon);
            {
                Scope sc(h, KC_GSUPDATE, vbytes(h, k + 2));
                if (k <= 4) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<4, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
                else if (k <= 8) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<8, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
                else if (k <= 16) VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<16, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
                else VW_DISPATCH(h, hipLaunchKernelGGL((k_gs_update_dev<32, VW>), vgridw(h, VW), dim3(KSFD_BLOCK), 0, h->st, h->kv, w, (const double *)V, h->vlen, k, (const double *)dcoef, (const double *)dscale));
            }
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(hmon + 2 * j, dmon + 2 * j, 2 * sizeof(double), hipMemcpyDeviceToHost, h->st));
            HIPCHK(h, hipEventRecord(h->gm_ev[j], h->st));
            jlast = j;
            if ((rc = poll(j - 1))) return rc;                 // one iteration behind: the GPU already has iteration j queued
            if (jc >= 0) break;
            if (r2 > 0.0 && r1 * (r1 / r2) <= 1.5 * tol) {     // extrapolation says iteration j converges: look now, queue nothing more
                if ((rc = poll(j))) return rc;
                if (jc >= 0) break;
            }
        }
        if (jc < 0 && (rc = poll(jlast))) return rc;
        const int kused = jc >= 0 ? jc + 1 : jlast + 1;
        total += jlast + 1;
        HIPCHK(h, hipMemcpyAsync(hH, dH, sizeof(double) * (size_t)ld * h->restart_alloc, hipMemcpyDeviceToHost, h->st));
        HIPCHK(h, hipMemcpyAsync(hg, dg, sizeof(double) * ld, hipMemcpyDeviceToHost, h->st));
        HIPCHK(h, hipStreamSynchronize(h->st));
        for (int i = kused - 1; i >= 0; i--) {
            double s = hg[i];
            for (int q = i + 1; q < kused; q++) s -= hH[(size_t)ld * q + i] * y[q];
            y[i] = s / hH[(size_t)ld * i + i];
        }
        if ((rc = op_basis_axpy(h, x, V, kused, y.data(), first ? 0.0 : 1.0))) return rc;
        first = false;
        rn = hmon[2 * (kused - 1)];
        if (jc >= 0 || total >= maxit) break;
    }
    ls->its = total;
    ls->rel = rn / bn;
    if (rn > tol) return fail(h, KSFD_ELINEAR, "GMRES did not converge: %d iterations, relative residual %.3e (tol %.3e)", total, rn / bn, tol / bn);
    return KSFD_OK;
}
