// libksfd_hip.so -- C ABI (include/ksfd_hip.h) and the Rosenbrock-W step; handle, launch wrappers, multigrid and Krylov
// solvers live in handle.hip.h / ops.hip.h / mg_host.hip.h / krylov.hip.h (one translation unit).  Together they
// stand in for petsc4py TS.step() in the reference (KSFD/ksfdts.py:211).
// gfx950 only.  No CPU fallback: every entry point runs HIP kernels or fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/ksfd_hip.h"
#include "stencil.hip.h"
#include "mg.hip.h"
#include "spectral.hip.h"
#include "transport.h"


#include "handle.hip.h"
#include "ops.hip.h"
#include "spectral_host.hip.h"
#include "mg_host.hip.h"
#include "krylov.hip.h"

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" const char *ksfd_last_error(const ksfd_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" void ksfd_destroy(ksfd_handle *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->st) hipStreamSynchronize(h->st);
    double *bufs[] = { h->bstore, h->ckpt, h->Zb, h->pvec, h->coef, h->u, h->usave, h->Z, h->bvec, h->Y, h->V, h->t1, h->t2, h->t3, h->errv, h->Gb, h->dGb, h->flat, h->part, h->dres };
    for (double *b : bufs) if (b) hipFree(b);
    for (int s = 0; s < 4; s++) for (int c = 0; c <= KSFD_MAXL; c++) if (h->src[s][c]) hipFree(h->src[s][c]);
    if (h->coef32) hipFree(h->coef32);
    if (h->hres) hipHostFree(h->hres);
    if (h->pub_count) hipFree(h->pub_count);
    if (h->gm_host) hipHostFree(h->gm_host);
    if (h->gm_dev) hipFree(h->gm_dev);
    for (auto e : h->gm_ev) if (e) hipEventDestroy(e);
    for (auto &p : h->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : h->pool) hipEventDestroy(e);
    if (h->st_io) { hipStreamSynchronize(h->st_io); hipStreamDestroy(h->st_io); }
    for (int q = 0; q < 2; q++) {
        if (h->snap_dev[q]) hipFree(h->snap_dev[q]);
        if (h->snap_host[q]) hipHostFree(h->snap_host[q]);
        if (h->snap_ready[q]) hipEventDestroy(h->snap_ready[q]);
        if (h->snap_done[q]) hipEventDestroy(h->snap_done[q]);
    }
    mg_free(h);
    spec_free(h);
    delete h->tr;
    if (h->ev_ready) hipEventDestroy(h->ev_ready);
    if (h->ev_halo) hipEventDestroy(h->ev_halo);
    if (h->st_comm) hipStreamDestroy(h->st_comm);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

extern "C" int ksfd_create(const ksfd_config *cfg, const ksfd_dist *dist, ksfd_handle **out)
{
    if (!cfg || !out) return fail(nullptr, KSFD_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->dim < 1 || cfg->dim > 3) return fail(nullptr, KSFD_EINVAL, "dim must be 1, 2 or 3");
    for (int a = 0; a < cfg->dim; a++)
        if (cfg->n[a] < 5) return fail(nullptr, KSFD_EINVAL, "n[%d]=%lld < 5: the width-2 periodic star needs >= 5 points", a, (long long)cfg->n[a]);
    ksfd_handle *h = new ksfd_handle();
    memset(h->src, 0, sizeof h->src);
    memset(&h->prof, 0, sizeof h->prof);
    h->cfg = *cfg;
    // the caller keeps ownership of the tables: the handle holds numeric copies (fill_phys), never these pointers
    h->cfg.lig_group = nullptr; h->cfg.lig_w = h->cfg.lig_s = h->cfg.lig_gamma = h->cfg.lig_D = nullptr;
    h->cfg.grp_alpha = h->cfg.grp_beta = nullptr;
    int rc = fill_phys(h, cfg);
    if (rc) { g_create_error = h->err; delete h; return rc; }
    h->rank = dist ? dist->rank : 0;
    h->size = dist ? dist->size : 1;
    h->device = dist ? dist->device : 0;
    if (h->size < 1 || h->rank < 0 || h->rank >= h->size) { delete h; return fail(nullptr, KSFD_EINVAL, "bad rank/size"); }
    // one rank WITH a transport = a ring of one: ghost units, halo exchange with itself, all-reduce over one rank, the own-piece path of
    // the all-to-alls -- the whole multi-rank code path on a single GPU (how the RCCL transport is exercised on a one-GPU box)
    h->ring = h->size > 1 || (dist && dist->transport != 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { delete h; return fail(nullptr, KSFD_EHIP, "no HIP device available (libksfd_hip has no CPU path)"); }
    if (h->device < 0 || h->device >= ndev) { delete h; return fail(nullptr, KSFD_EINVAL, "device %d of %d", h->device, ndev); }
#define CFAIL(code, ...) do { int rc_ = fail(nullptr, code, __VA_ARGS__); ksfd_destroy(h); return rc_; } while (0)
    if (hipSetDevice(h->device) != hipSuccess) CFAIL(KSFD_EHIP, "hipSetDevice(%d) failed", h->device);
    if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) CFAIL(KSFD_EHIP, "hipStreamCreate failed");
    if (h->ring && (hipStreamCreateWithFlags(&h->st_comm, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming) != hipSuccess)) CFAIL(KSFD_EHIP, "comm stream/event creation failed");

    KGeom &G = h->G;
    G.dim = cfg->dim; G.F = cfg->nlig + 1;
    const int slow = cfg->dim - 1;
    int64_t nglob = cfg->n[slow];
    if (h->size > 1 && (nglob % h->size != 0 || nglob / h->size < 4))
        CFAIL(KSFD_EINVAL, "slab axis extent %lld must be divisible by %d ranks with >= 4 units each", (long long)nglob, h->size);
    int64_t sloc = nglob / h->size;
    h->slow0 = sloc * h->rank;
    G.ng = h->ring ? 2 : 0;
    G.wrap_slow = !h->ring;
    G.nx = cfg->n[0]; G.ny = cfg->dim >= 2 ? cfg->n[1] : 1; G.nz = cfg->dim >= 3 ? cfg->n[2] : 1;
    if (slow == 0) G.nx = sloc; else if (slow == 1) G.ny = sloc; else G.nz = sloc;
    G.inner = slow == 0 ? 1 : (slow == 1 ? G.nx : G.nx * G.ny);
    G.sloc = sloc;
    G.plane = (sloc + 2 * G.ng) * G.inner;
    G.nloc = sloc * G.inner;
    h->kv.plane = G.plane; h->kv.off = (long long)G.ng * G.inner; h->kv.nloc = G.nloc; h->kv.nf = G.F;
    h->vlen = (int64_t)G.F * G.plane;
    h->nblk_vec = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 2048);
    build_tableau(h);

    // Krylov storage is what scales with the restart length: V (m+1 vectors) + Zb (m).  Size it to the HBM that is free:
    // ~30 other full-size vectors (state, stage vectors, temporaries, coefficient planes, multigrid level 0 + coarse levels)
    // must fit first.  m = 30 on anything up to ~12k^2 x 2 fields on a 288 GB MI355X; larger grids run with a shorter restart.
    // Round 3: where HBM is plentiful the basis may be longer than the first cycle's 30 vectors -- gmres() doubles the restart length after
    // a cycle that did not converge (indefinite stage matrices late in a run), up to 120 and to 40 % of the free memory for V + Zb.
    h->restart_alloc = 30;
    int fit_local = 30;
    {
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && mfree > 0) {
            const double vecbytes = 8.0 * (double)h->vlen;
            const double room = 0.92 * (double)mfree / vecbytes - 30.0;
            const int fit_room = (int)floor(std::min((room - 1.0) / 2.0, 1.0e6));
            const int fit_40 = (int)floor(std::min(0.4 * (double)mfree / vecbytes / 2.0, 1.0e6));
            static const int cap = getenv("KSFD_RESTART_MAX") ? std::max(8, std::min(atoi(getenv("KSFD_RESTART_MAX")), 120)) : 120;
            fit_local = std::max(std::min(std::min(cap, fit_room), fit_40), std::min(30, fit_room));
        }
    }
    if (alloc_d(h, &h->dres, 128)) CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
    if (h->ring) {
        std::string terr;
        h->tr = make_transport(dist, G.F, G.inner, terr);
        if (!h->tr) CFAIL(KSFD_ECOMM, "transport %d: %s", dist->transport, terr.c_str());
        // every rank must run with the SAME restart length (the multi-dot row counts and restart points are part of the
        // collective pattern): take the minimum over the ranks of what each one's free HBM allows
        double v = -(double)fit_local;
        if (hipMemcpyAsync(h->dres, &v, sizeof v, hipMemcpyHostToDevice, h->st) != hipSuccess || h->tr->allreduce(h->dres, 1, 1, h->st))
            CFAIL(KSFD_ECOMM, "agreeing on the restart length failed: %s", h->tr->error().c_str());
        if (h->tr->result_on_host()) v = h->tr->host_result()[0];
        else if (hipMemcpyAsync(&v, h->dres, sizeof v, hipMemcpyDeviceToHost, h->st) != hipSuccess || hipStreamSynchronize(h->st) != hipSuccess)
            CFAIL(KSFD_EHIP, "reading the agreed restart length failed");
        fit_local = (int)(-v + 0.5);
    }
    if (fit_local != 30) {
        if (fit_local < 8) CFAIL(KSFD_ENOMEM, "grid too large for this device: a vector is %.2f GB, the solver needs ~47 of them (restart length that fits: %d)", 8.0 * (double)h->vlen / 1e9, fit_local);
        h->restart_alloc = fit_local;
    }
    double **vecs[] = { &h->u, &h->usave, &h->Z, &h->bvec, &h->t1, &h->t2, &h->t3, &h->errv };
    for (double **v : vecs) {
        if (alloc_d(h, v, h->vlen)) CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
        hipMemsetAsync(*v, 0, sizeof(double) * (size_t)h->vlen, h->st);
    }
    if (alloc_d(h, &h->Y, 4 * h->vlen) || alloc_d(h, &h->V, (int64_t)(h->restart_alloc + 1) * h->vlen) ||
        alloc_d(h, &h->Gb, G.plane) || alloc_d(h, &h->dGb, G.plane) || alloc_d(h, &h->coef, (int64_t)(3 + cfg->nlig) * G.plane) ||
        alloc_d(h, &h->flat, (int64_t)std::max(G.F, 3) * G.nloc) ||
        alloc_d(h, &h->part, (int64_t)(2 * KSFD_MAXDOT + 4) * 4096))
        CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
    // flexible-GMRES / spectral-fallback basis: allocated here when the device has room (the sizing above counted it), so that no step
    // pays a multi-GB hipMalloc; if it does not fit now it is tried again on first use
    if (hipMalloc((void **)&h->Zb, sizeof(double) * (size_t)h->restart_alloc * (size_t)h->vlen) != hipSuccess) { h->Zb = nullptr; hipGetLastError(); }
    hipMemsetAsync(h->Y, 0, sizeof(double) * (size_t)(4 * h->vlen), h->st);
    hipMemsetAsync(h->V, 0, sizeof(double) * (size_t)((h->restart_alloc + 1) * h->vlen), h->st);
    if (hipHostMalloc((void **)&h->hres, sizeof(double) * 128 + 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) CFAIL(KSFD_ENOMEM, "hipHostMalloc failed");
    h->pub_flag = reinterpret_cast<unsigned long long *>(h->hres + 128);
    *h->pub_flag = 0;
    if (hipHostGetDevicePointer((void **)&h->hres_dev, h->hres, 0) != hipSuccess || hipMalloc((void **)&h->pub_count, sizeof(unsigned int)) != hipSuccess ||
        hipMemset(h->pub_count, 0, sizeof(unsigned int)) != hipSuccess) { h->zero_copy = false; h->hres_dev = nullptr; }
    else h->pub_flag_dev = reinterpret_cast<unsigned long long *>(h->hres_dev + 128);
    {
        const int m = h->restart_alloc;
        const size_t ndev = (size_t)(m + 1) * (m + 1) + (size_t)(m + 1) * m + 2 * m + (m + 1) + KSFD_MAXDOT + 1 + 2 * (m + 1);
        const size_t nhost = 2 * (m + 1) + (size_t)(m + 1) * m + (m + 1);
        if (alloc_d(h, &h->gm_dev, (int64_t)ndev)) CFAIL(KSFD_ENOMEM, "%s", h->err.c_str());
        if (hipHostMalloc((void **)&h->gm_host, sizeof(double) * nhost, hipHostMallocDefault) != hipSuccess) CFAIL(KSFD_ENOMEM, "hipHostMalloc failed");
        h->gm_ev.resize(m + 1);
        for (auto &e : h->gm_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) CFAIL(KSFD_EHIP, "hipEventCreate failed");
    }

    spec_build(h);                                            // leaves spec.ok = false where it does not apply (3-D, non power-of-two extents, ...)
    if (mg_build(h)) { mg_free(h); h->mg_ok = false; }        // out of memory for the hierarchy: run without the multigrid preconditioner
    if ((h->spec.ok || h->mg_ok) && alloc_d(h, &h->bstore, 3 * h->vlen)) { h->bstore = nullptr; h->spec_guess = false; h->err.clear(); }
    if (h->ring) {
        // which solvers exist decides the sequence of collectives of every step: all ranks must agree (an allocation that failed
        // on one rank only would otherwise leave the others waiting in an all-reduce)
        double flags[3] = { h->spec.ok ? 1.0 : 0.0, h->mg_ok ? 1.0 : 0.0, (h->spec_guess && h->bstore) ? 1.0 : 0.0 };
        for (double &f : flags) f = -f;                       // MIN through the transport's MAX
        if (hipMemcpyAsync(h->dres, flags, sizeof flags, hipMemcpyHostToDevice, h->st) != hipSuccess || h->tr->allreduce(h->dres, 3, 1, h->st))
            CFAIL(KSFD_ECOMM, "agreeing on the available solvers failed: %s", h->tr->error().c_str());
        if (h->tr->result_on_host()) memcpy(flags, h->tr->host_result(), sizeof flags);
        else if (hipMemcpyAsync(flags, h->dres, sizeof flags, hipMemcpyDeviceToHost, h->st) != hipSuccess || hipStreamSynchronize(h->st) != hipSuccess)
            CFAIL(KSFD_EHIP, "reading the agreed solver flags failed");
        if (flags[0] > -0.5 && h->spec.ok) spec_free(h);
        if (flags[1] > -0.5 && h->mg_ok) { mg_free(h); h->mg_ok = false; }
        if (flags[2] > -0.5) h->spec_guess = false;
    }
    if (hipStreamSynchronize(h->st) != hipSuccess) CFAIL(KSFD_EHIP, "stream sync failed in create");
#undef CFAIL
    *out = h;
    return KSFD_OK;
}

extern "C" int ksfd_rccl_unique_id(void *out128)
{
    if (!out128) return KSFD_EINVAL;
    std::string err;
    RcclApi api;
    if (!api.load(err)) return fail(nullptr, KSFD_ECOMM, "%s", err.c_str());
    auto getid = (decltype(&ncclGetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    if (!getid) return fail(nullptr, KSFD_ECOMM, "librccl lacks ncclGetUniqueId");
    ncclUniqueId id;
    ncclResult_t r = getid(&id);
    if (r != ncclSuccess) return fail(nullptr, KSFD_ECOMM, "ncclGetUniqueId: %s", api.GetErrorString(r));
    memcpy(out128, &id, sizeof id);
    return KSFD_OK;
}

extern "C" int ksfd_update_params(ksfd_handle *h, const ksfd_config *cfg)
{
    if (!h || !cfg) return KSFD_EINVAL;
    if (cfg->dim != h->cfg.dim || cfg->nlig != h->cfg.nlig) return fail(h, KSFD_EINVAL, "update_params cannot change dim/nlig");
    for (int a = 0; a < 3; a++) if (cfg->n[a] != h->cfg.n[a]) return fail(h, KSFD_EINVAL, "update_params cannot change the grid");
    h->cfg = *cfg;
    h->cfg.lig_group = nullptr; h->cfg.lig_w = h->cfg.lig_s = h->cfg.lig_gamma = h->cfg.lig_D = nullptr;
    h->cfg.grp_alpha = h->cfg.grp_beta = nullptr;
    h->coef_fresh = false;                                   // G, G_rho, G_U depend on the parameters
    return fill_phys(h, cfg);
}

extern "C" int ksfd_set_stage_params(ksfd_handle *h, int32_t stage, const ksfd_config *cfg)
{
    if (!h || stage < -1 || stage > 3) return KSFD_EINVAL;
    if (!cfg) {                                            // clear: the stages use the handle's parameters again
        for (int s = 0; s < 4; s++) if (stage == -1 || s == stage) h->Pst_valid[s] = false;
        return KSFD_OK;
    }
    if (cfg->dim != h->cfg.dim || cfg->nlig != h->cfg.nlig) return fail(h, KSFD_EINVAL, "set_stage_params cannot change dim/nlig");
    for (int a = 0; a < 3; a++) if (cfg->n[a] != h->cfg.n[a] || cfg->L[a] != h->cfg.L[a]) return fail(h, KSFD_EINVAL, "set_stage_params cannot change the grid");
    for (int s = 0; s < 4; s++) {
        if (stage != -1 && s != stage) continue;
        int rc = fill_phys(h, cfg, &h->Pst[s]);
        if (rc) return rc;
        h->Pst_valid[s] = true;
    }
    return KSFD_OK;
}

extern "C" int ksfd_local_range(const ksfd_handle *h, int64_t *b, int64_t *e)
{
    if (!h) return KSFD_EINVAL;
    if (b) *b = h->slow0;
    if (e) *e = h->slow0 + h->G.sloc;
    return KSFD_OK;
}
extern "C" int64_t ksfd_local_size(const ksfd_handle *h) { return h ? (int64_t)h->G.F * h->G.nloc : 0; }
extern "C" double *ksfd_device_state(ksfd_handle *h) { return h ? h->u : nullptr; }
extern "C" int64_t ksfd_device_plane_stride(const ksfd_handle *h) { return h ? h->G.plane : 0; }
extern "C" int64_t ksfd_device_interior_offset(const ksfd_handle *h) { return h ? (int64_t)h->G.ng * h->G.inner : 0; }

extern "C" int ksfd_set_state(ksfd_handle *h, const double *u, int32_t layout)
{
    if (!h || !u) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int rc = upload(h, u, layout, h->u);
    h->coef_fresh = false;
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}
// Asynchronous read-out of the resident state for writers ("next" row f2: TimeSeries fed from the device without
// stalling the stepper).  _begin: the layout transform runs on the compute stream (ordered after everything issued so
// far, ~0.1 ms at 4096^2) into one of two device staging slots; the D2H copy into pinned memory runs on a third stream.
// _wait: blocks until that copy has landed and hands out the pinned buffer (F*nlocal doubles), valid until the slot is
// used again, i.e. until the second-next _begin.  A slot whose data nobody waited for is simply overwritten.
extern "C" int ksfd_snapshot_begin(ksfd_handle *h, int32_t layout, int32_t *slot)
{
    if (!h || !slot) return KSFD_EINVAL;
    if (layout < 0 || layout > 2) return fail(h, KSFD_EINVAL, "bad layout %d", layout);
    hipSetDevice(h->device);
    const KGeom &G = h->G;
    const size_t bytes = sizeof(double) * (size_t)G.F * G.nloc;
    if (!h->st_io) {
        if (hipStreamCreateWithFlags(&h->st_io, hipStreamNonBlocking) != hipSuccess) return fail(h, KSFD_EHIP, "hipStreamCreate (io) failed");
        for (int q = 0; q < 2; q++) {
            if (hipMalloc((void **)&h->snap_dev[q], bytes) != hipSuccess || hipHostMalloc((void **)&h->snap_host[q], bytes, hipHostMallocDefault) != hipSuccess)
                return fail(h, KSFD_ENOMEM, "snapshot staging buffers (%zu bytes each) could not be allocated", bytes);
            if (hipEventCreateWithFlags(&h->snap_ready[q], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->snap_done[q], hipEventDisableTiming) != hipSuccess) return fail(h, KSFD_EHIP, "hipEventCreate failed");
        }
    }
    const int q = h->snap_next;
    h->snap_next ^= 1;
    if (h->snap_busy[q]) HIPCHK(h, hipEventSynchronize(h->snap_done[q]));      // its previous copy must have left the staging slot
    {
        Scope sc(h, KC_MISC, vbytes(h, 2));
        hipLaunchKernelGGL(k_to_host_layout, vgrid(h), dim3(KSFD_BLOCK), 0, h->st, G, layout, (const double *)h->u, G.plane,
                           (long long)G.ng * G.inner, h->snap_dev[q]);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->snap_ready[q], h->st));
    HIPCHK(h, hipStreamWaitEvent(h->st_io, h->snap_ready[q], 0));
    HIPCHK(h, hipMemcpyAsync(h->snap_host[q], h->snap_dev[q], bytes, hipMemcpyDeviceToHost, h->st_io));
    HIPCHK(h, hipEventRecord(h->snap_done[q], h->st_io));
    h->snap_busy[q] = true;
    *slot = q;
    return KSFD_OK;
}

extern "C" int ksfd_snapshot_wait(ksfd_handle *h, int32_t slot, const double **host)
{
    if (!h || !host || slot < 0 || slot > 1) return KSFD_EINVAL;
    if (!h->snap_busy[slot]) return fail(h, KSFD_EINVAL, "snapshot slot %d holds nothing", slot);
    hipSetDevice(h->device);
    HIPCHK(h, hipEventSynchronize(h->snap_done[slot]));
    *host = h->snap_host[slot];
    return KSFD_OK;
}

extern "C" int ksfd_checkpoint(ksfd_handle *h, int32_t op)
{
    if (!h || op < 0 || op > 1) return KSFD_EINVAL;
    hipSetDevice(h->device);
    if (op == 0) {
        if (!h->ckpt && alloc_d(h, &h->ckpt, h->vlen)) return KSFD_ENOMEM;
        HIPCHK(h, hipMemcpyAsync(h->ckpt, h->u, sizeof(double) * (size_t)h->vlen, hipMemcpyDeviceToDevice, h->st));
        h->ckpt_memo = { h->lamJ, h->lam_age, h->lam_period, h->mg_shift_floor, h->sf_dir, h->sf_hold, h->sf_tried_down, h->sf_prev_its, h->sf_prev_floor,
                         h->nsteps, h->spec.bad_until, h->spec.backoff, h->spec.rho_step, h->spec.rho_prev };
        for (MGLevel &L : h->mg) L.pv_norm = 0.0;       // the step after a save and the step after a restore both set the hierarchy up cold
        h->ckpt_valid = true;
        return KSFD_OK;
    }
    if (!h->ckpt_valid) return fail(h, KSFD_EINVAL, "no checkpoint has been saved");
    HIPCHK(h, hipMemcpyAsync(h->u, h->ckpt, sizeof(double) * (size_t)h->vlen, hipMemcpyDeviceToDevice, h->st));
    h->coef_fresh = false;
    const ksfd_handle::SolverMemo &m = h->ckpt_memo;
    h->lamJ = m.lamJ; h->lam_age = m.lam_age; h->lam_period = m.lam_period; h->mg_shift_floor = m.mg_shift_floor;
    h->sf_dir = m.sf_dir; h->sf_hold = m.sf_hold; h->sf_tried_down = m.sf_tried_down; h->sf_prev_its = m.sf_prev_its; h->sf_prev_floor = m.sf_prev_floor;
    h->nsteps = m.nsteps; h->spec.bad_until = m.spec_bad_until; h->spec.backoff = m.spec_backoff; h->spec.means_valid = false;
    h->spec.rho_step = m.spec_rho_step; h->spec.rho_prev = m.spec_rho_prev;
    h->mg_coef_valid = false; h->mg_shift = -1.0; h->poly_shift = -1.0; h->have_err = false;
    for (MGLevel &L : h->mg) L.pv_norm = 0.0;
    rec_reset(h);
    return KSFD_OK;
}

extern "C" int ksfd_set_state_random(ksfd_handle *h, const int64_t *nc, const double *z, double rho0)
{
    if (!h || !nc || !z) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int64_t n = 1;
    for (int a = 0; a < 3; a++) {
        if (a < h->G.dim ? nc[a] < 1 : nc[a] != 1) return fail(h, KSFD_EINVAL, "coarse grid must be >= 1 per used axis and 1 elsewhere");
        n *= nc[a];
    }
    double *dz = nullptr;
    h->coef_fresh = false;
    if (hipMalloc((void **)&dz, sizeof(double) * (size_t)n) != hipSuccess) return fail(h, KSFD_ENOMEM, "hipMalloc of the coarse samples failed");
    hipError_t e = hipMemcpyAsync(dz, z, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, h->st);
    const KGeom &G = h->G;
    int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 65535);
    if (e == hipSuccess) {
        Scope sc(h, KC_MISC, vbytes(h, 1));
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_random_start<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, (long long)h->cfg.n[G.dim - 1],
                                                  (long long)h->slow0, (long long)nc[0], (long long)nc[1], (long long)nc[2], (const double *)dz, rho0, h->u));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    hipFree(dz);
    if (e != hipSuccess) return fail(h, KSFD_EHIP, "random start: %s", hipGetErrorString(e));
    return KSFD_OK;
}
extern "C" int ksfd_get_state(ksfd_handle *h, double *u, int32_t layout)
{
    if (!h || !u) return KSFD_EINVAL;
    hipSetDevice(h->device);
    return download(h, h->u, layout, u);
}

extern "C" int ksfd_set_source(ksfd_handle *h, int32_t stage, int32_t field, const double *srch, int32_t layout)
{
    if (!h || field < 0 || field >= h->G.F || stage < -1 || stage > 3) return h ? fail(h, KSFD_EINVAL, "bad stage/field") : KSFD_EINVAL;
    if (layout != KSFD_LAYOUT_SOA) return fail(h, KSFD_EINVAL, "sources are single dense planes: pass layout SOA");
    hipSetDevice(h->device);
    for (int s = 0; s < 4; s++) {
        if (stage != -1 && s != stage) continue;
        if (!srch) {
            if (h->src[s][field]) { HIPCHK(h, hipStreamSynchronize(h->st)); hipFree(h->src[s][field]); h->src[s][field] = nullptr; }
            continue;
        }
        if (!h->src[s][field] && alloc_d(h, &h->src[s][field], h->G.nloc)) return KSFD_ENOMEM;
        HIPCHK(h, hipMemcpyAsync(h->src[s][field], srch, sizeof(double) * (size_t)h->G.nloc, hipMemcpyHostToDevice, h->st));
    }
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

extern "C" int ksfd_rhs(ksfd_handle *h, double t, const double *uh, double *outh, int32_t layout)
{
    (void)t;   // sources(t) are uploaded by the caller (ksfd_set_source); constants via ksfd_update_params
    if (!h || !outh) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int rc;
    double *uin = h->u;
    if (uh) { if ((rc = upload(h, uh, layout, h->t1))) return rc; uin = h->t1; }
    if ((rc = halo(h, uin))) return rc;
    if ((rc = op_rhs(h, uin, 0, h->t3))) return rc;
    return download(h, h->t3, layout, outh);
}

extern "C" int ksfd_jvp(ksfd_handle *h, const double *uh, const double *vh, double *outh, int32_t layout)
{
    if (!h || !vh || !outh) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int rc;
    double *uin = h->u;
    if (uh) { if ((rc = upload(h, uh, layout, h->t1))) return rc; uin = h->t1; }
    if ((rc = upload(h, vh, layout, h->t2))) return rc;
    if ((rc = halo(h, uin)) || (rc = halo(h, h->t2))) return rc;
    if (!uh && h->use_frozen) {
        // the path the stepper uses: coefficients of the stored state once, then the frozen-coefficient kernels
        if ((rc = ensure_coef(h, true))) return rc;
        if ((rc = op_jvp_frozen(h, h->t2, 0, 0.0, h->t3))) return rc;
    } else if ((rc = op_jvp(h, uin, h->t2, 0, 0.0, h->t3))) return rc;
    return download(h, h->t3, layout, outh);
}

static int velocity_common(ksfd_handle *h, const double *uin, double *vel_dev, double vmax[3])
{
    const KGeom &G = h->G;
    int rc;
    const double *Gplane = h->Gb;
    if (uin == h->u && h->use_frozen) {
        // the G plane of the resident state is plane 1 of the frozen coefficient planes; the next step reuses them
        if ((rc = ensure_coef(h))) return rc;
        Gplane = h->coef + G.plane;
    } else {
        if ((rc = halo(h, (double *)uin))) return rc;
        int nbp = (int)std::min<long long>((G.plane + KSFD_BLOCK - 1) / KSFD_BLOCK, 4096);
        Scope sc(h, KC_GFIELD, 8.0 * (G.F + 1) * (double)G.plane);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_gfield<NL, false>), dim3(nbp), dim3(KSFD_BLOCK), 0, h->st, G, h->P, uin, (const double *)nullptr, h->Gb, (double *)nullptr));
    }
    int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 1024);
    if (vmax && !vel_dev && G.dim == 2 && (G.nx % 2 == 0) && G.nx >= 4) {
        // CFL check: max|dG/dx|, max|dG/dy| only -- row-structured kernel, 16-B loads, no 64-bit divisions
        const int bx = (int)((G.nx / 2 + KSFD_BLOCK - 1) / KSFD_BLOCK), by = (int)std::min<long long>(G.sloc, std::max(1, 2048 / bx));
        nb = bx * by;
        Scope sc(h, KC_VELOCITY, 8.0 * (double)G.nloc);
        hipLaunchKernelGGL(k_velmax2d, dim3(bx, by), dim3(KSFD_BLOCK), 0, h->st, G, h->P, Gplane, h->part);
    } else if (vmax && !vel_dev && G.dim == 3 && (G.nx % 2 == 0) && G.nx >= 4 && G.ny >= 4 && G.nx * G.ny < (1LL << 30)) {
        const int bx = (int)((G.nx / 2 * G.ny + KSFD_BLOCK - 1) / KSFD_BLOCK);
        int zseg = (int)G.sloc;                               // enough blocks to fill the chip, segments of >= 8 planes (4 extra loads each)
        while (zseg >= 32 && (long long)bx * ((G.sloc + zseg - 1) / zseg) < 8192) zseg = (zseg + 1) / 2;
        const int by = (int)((G.sloc + zseg - 1) / zseg);
        nb = bx * by;
        if (3LL * nb > part_capacity()) return fail(h, KSFD_EINVAL, "velocity_max: grid too large for the partial-result buffer");
        Scope sc(h, KC_VELOCITY, 8.0 * (double)G.nloc);
        hipLaunchKernelGGL(k_velmax3d, dim3(bx, by), dim3(KSFD_BLOCK), 0, h->st, G, h->P, zseg, Gplane, h->part);
    } else {
        Scope sc(h, KC_VELOCITY, 8.0 * (double)G.nloc * (1 + (vel_dev ? G.dim : 0)));
        hipLaunchKernelGGL(k_velocity, dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, Gplane, vel_dev, vmax ? h->part : (double *)nullptr);
    }
    HIPCHK(h, hipGetLastError());
    if (vmax) {
        if ((rc = reduce_rows(h, 3, nb, 1))) return rc;
        for (int a = 0; a < 3; a++) vmax[a] = a < G.dim ? h->hres[a] : 0.0;
    }
    return KSFD_OK;
}

extern "C" int ksfd_velocity(ksfd_handle *h, const double *uh, double *velh, int32_t layout)
{
    if (!h || !velh) return KSFD_EINVAL;
    if (layout != KSFD_LAYOUT_SOA) return fail(h, KSFD_EINVAL, "velocity output is dim dense SoA planes: pass layout SOA for it");
    hipSetDevice(h->device);
    int rc;
    double *uin = h->u;
    if (uh) { if ((rc = upload(h, uh, layout, h->t1))) return rc; uin = h->t1; }
    if ((rc = velocity_common(h, uin, h->flat, nullptr))) return rc;
    HIPCHK(h, hipMemcpyAsync(velh, h->flat, sizeof(double) * (size_t)h->G.dim * h->G.nloc, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

extern "C" int ksfd_velocity_max(ksfd_handle *h, double vmax[3])
{
    if (!h || !vmax) return KSFD_EINVAL;
    hipSetDevice(h->device);
    return velocity_common(h, h->u, nullptr, vmax);
}

extern "C" int ksfd_groom(ksfd_handle *h)
{
    if (!h) return KSFD_EINVAL;
    hipSetDevice(h->device);
    Scope sc(h, KC_MISC, vbytes(h, 2));
    hipLaunchKernelGGL(k_groom, vgrid(h), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, h->P.rhomin, h->P.Umin);
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

extern "C" int ksfd_count_worms(ksfd_handle *h, double *total)
{
    if (!h || !total) return KSFD_EINVAL;
    hipSetDevice(h->device);
    {
        Scope sc(h, KC_MISC, 8.0 * (double)h->G.nloc);
        hipLaunchKernelGGL(k_sum_rho, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, h->part);
    }
    HIPCHK(h, hipGetLastError());
    int rc = reduce_rows(h, 1, h->nblk_vec, 0);
    if (rc) return rc;
    *total = h->hres[0];
    return KSFD_OK;
}

extern "C" int ksfd_scale_rho(ksfd_handle *h, double factor)
{
    if (!h) return KSFD_EINVAL;
    hipSetDevice(h->device);
    h->coef_fresh = false;
    Scope sc(h, KC_MISC, 16.0 * (double)h->G.nloc);
    hipLaunchKernelGGL(k_mul_rho, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, (const double *)nullptr, factor);
    HIPCHK(h, hipGetLastError());
    return KSFD_OK;
}

extern "C" int ksfd_mul_rho(ksfd_handle *h, const double *fh)
{
    if (!h || !fh) return KSFD_EINVAL;
    hipSetDevice(h->device);
    HIPCHK(h, hipMemcpyAsync(h->flat, fh, sizeof(double) * (size_t)h->G.nloc, hipMemcpyHostToDevice, h->st));
    h->coef_fresh = false;
    Scope sc(h, KC_MISC, 24.0 * (double)h->G.nloc);
    hipLaunchKernelGGL(k_mul_rho, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, (const double *)h->flat, 1.0);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}

// Assembled Jacobian export (row f4 of the scope table; kernel k_jac_csr in stencil.hip.h)
extern "C" int ksfd_jacobian_nnz(ksfd_handle *h, int64_t *nrows, int64_t *nnz)
{
    if (!h) return KSFD_EINVAL;
    const int64_t npts = 4 * h->G.dim + 1;
    if (nrows) *nrows = (int64_t)h->G.F * h->G.nloc;
    if (nnz) *nnz = h->G.nloc * ((int64_t)h->G.F * npts + (int64_t)h->P.nlig * (npts + 1));
    return KSFD_OK;
}

extern "C" int ksfd_jacobian_csr(ksfd_handle *h, int64_t *rowptr, int64_t *col, double *val)
{
    if (!h || !rowptr || !col || !val) return KSFD_EINVAL;
    hipSetDevice(h->device);
    int64_t nrows, nnz;
    ksfd_jacobian_nnz(h, &nrows, &nnz);
    int rc;
    if ((rc = ensure_coef(h))) return rc;                    // clamps like groom
    long long *dcol = nullptr;
    double *dval = nullptr;
    if (hipMalloc((void **)&dcol, sizeof(long long) * (size_t)nnz) != hipSuccess ||
        hipMalloc((void **)&dval, sizeof(double) * (size_t)nnz) != hipSuccess) {
        if (dcol) hipFree(dcol);
        return fail(h, KSFD_ENOMEM, "hipMalloc of the CSR staging buffers failed");
    }
    const KGeom &G = h->G;
    int nb = (int)std::min<long long>((G.nloc + KSFD_BLOCK - 1) / KSFD_BLOCK, 65535);
    {
        Scope sc(h, KC_MISC, 16.0 * (double)nnz + 8.0 * (3 + h->P.nlig) * (double)G.nloc);
        NL_DISPATCH(h->P.nlig, hipLaunchKernelGGL((k_jac_csr<NL>), dim3(nb), dim3(KSFD_BLOCK), 0, h->st, G, h->P, (const double *)h->coef,
                                                  (long long)h->cfg.n[G.dim - 1], (long long)h->slow0, dcol, dval));
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(col, dcol, sizeof(long long) * (size_t)nnz, hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(val, dval, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    hipFree(dcol);
    hipFree(dval);
    if (e != hipSuccess) return fail(h, KSFD_EHIP, "jacobian export: %s", hipGetErrorString(e));
    // row pointers are a fixed pattern: rho row F*npts entries, each U row npts+1
    const int64_t npts = 4 * G.dim + 1, F = G.F, per = F * npts + (F - 1) * (npts + 1);
    for (int64_t p = 0; p < G.nloc; p++) {
        rowptr[p * F] = p * per;
        for (int64_t l = 1; l < F; l++) rowptr[p * F + l] = p * per + F * npts + (l - 1) * (npts + 1);
    }
    rowptr[nrows] = nnz;
    return KSFD_OK;
}


// ------------------------------------------------------------------------------------------------
extern "C" void ksfd_default_step_opts(ksfd_step_opts *o)
{
    memset(o, 0, sizeof *o);
    o->rtol = 1e-5; o->atol = 1e-5;           // KSFD/ksfdts.py:66-67 defaults
    o->adapt = 1; o->max_reject = 10;
    o->clip_lo = 0.1; o->clip_hi = 5.0;       // -ts_adapt_clip 0.1,5
    o->dt_min = 1e-20; o->dt_max = 1e4;       // -ts_adapt_dt_min/-ts_adapt_dt_max
    o->safety = 0.9; o->reject_safety = 0.5;  // PETSc TSAdaptBasic defaults
    // relative residual 1e-6: the fields then agree with a 1e-12 solve to ~1e-10 rel-L2 over several steps
    // (tools/acc_vs_ksp.py); the north-star tolerance is 1e-8.  PETSc's own KSP default is 1e-5.
    o->ksp_rtol = 1e-6; o->ksp_atol = 1e-50;
    o->ksp_restart = 30; o->ksp_max_it = 2000;
    o->pc_type = 2;    // 0 none, 1 multigrid always, 2 multigrid when the step is stiff (2-D, single rank)
}

// One TSStep_RosW attempt loop (PETSc rosw.c restated; tableau/derivation in oracle/ksfd_oracle.c).
extern "C" int ksfd_step(ksfd_handle *h, double *t, double *hstep, const ksfd_step_opts *opts, ksfd_step_stats *stats)
{
    if (!h || !t || !hstep || !opts) return KSFD_EINVAL;
    hipSetDevice(h->device);
    if (getenv("KSFD_PC_SIGMA")) { h->mg_shift_floor = atof(getenv("KSFD_PC_SIGMA")); h->sf_auto = false; }      // experiment knob: fixed floor
    ksfd_step_stats st;
    memset(&st, 0, sizeof st);
    const double bytes0 = h->bytes_acc;
    const int64_t rhs0 = h->prof.launches[KC_RHS], jvp0 = h->prof.launches[KC_JVP];
    int64_t launches0 = 0;
    for (int c = 0; c < KSFD_NKCLASS; c++) launches0 += h->prof.launches[c];
    const long long sync0 = h->n_host_sync, pred0 = h->n_predicted, resid0 = h->n_residual;
    int rc = KSFD_OK;
    const int64_t vs = h->vlen;
    double hh = *hstep;
    bool prev_accept = !(opts->reserved & 4);           // bit 2: the caller's previous attempt of this step was rejected
    bool lam_done = false;
    int rejects = 0;
    const int max_rej = opts->max_reject < 0 ? 0x7fffffff : opts->max_reject;      // PETSc: -ts_max_reject -1 = unlimited
    const bool single = (opts->reserved & 2) != 0;                                   // one attempt per call (the caller owns the reject loop)
    // KSFDTS.solve grooms the global vector before every TS.step (KSFD/ksfdts.py:210)
    {
        // groom + roll-back copy of the owned points in one pass (ghost units of usave are never used: every roll-back is followed by a halo exchange)
        Scope sc(h, KC_MISC, vbytes(h, 3));
        hipLaunchKernelGGL(k_groom, vgrid(h), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, h->P.rhomin, h->P.Umin, h->usave);
        if (hipGetLastError() != hipSuccess) { rc = fail(h, KSFD_EHIP, "k_groom launch failed"); goto out; }
    }
    if ((rc = halo(h, h->u))) goto out;
    if (h->use_frozen && (rc = ensure_coef(h, true))) goto out;      // usually already there: the CFL check after the last step made them
    h->poly_shift = -1.0;
    h->nsteps++;
    if (h->spec.rho_step > 0.0) h->spec.rho_prev = h->spec.rho_step;      // contraction memory of the spectral sweeps: this step's maximum replaces the last one's
    h->spec.rho_step = 0.0;
    while (true) {
        const double shift = 1.0 / (GAMMA_RA * hh);
        // stiffness estimate X = h*gamma*lambda_max of the diffusion part; the multigrid preconditioner pays off above ~60
        double dmax = h->P.s2, lap = 0.0;
        for (int l = 0; l < h->P.nlig; l++) dmax = std::max(dmax, h->P.lig_D[l]);
        for (int a = 0; a < h->G.dim; a++) lap += (16.0 / 3.0) * h->P.inv_h2[a];
        const double stiff = dmax * lap / shift;
        // (measured crossover against the degree-6 polynomial: X ~ 80 on 4096^2, ~ 280 on 1024^2 where the V cycle is latency-bound)
        const double mg_from = h->mg_threshold * ((double)h->G.F * (double)h->G.nloc < 8.0e6 ? 3.0 : 1.0);
        // spectral preconditioner (constant-coefficient part of shift*I - J inverted by FFT): nearly exact while the state is a
        // smooth perturbation of a uniform one, at any stiffness; pc_type 2 uses it until it converges badly (see below), 4 always
        bool use_spec = h->spec.ok && h->use_frozen && (opts->pc_type == 4 || (opts->pc_type == 2 && stiff >= (fused_ok(h) ? h->spec_from : std::max(h->spec_from, 0.3)) && h->nsteps > h->spec.bad_until && !h->spec.user_off));      // (without the fused 2-D residual kernel a sweep costs more: 3-D at X = 0.29, 80 ms plain GMRES against 83 ms)
        if (use_spec) {
            if (!h->Zb && alloc_d(h, &h->Zb, (int64_t)h->restart_alloc * h->vlen)) { rc = KSFD_ENOMEM; goto out; }
            if (!h->spec.means_valid && (rc = spec_means(h))) goto out;
        }
        const bool use_pc = !use_spec && h->mg_ok && h->use_frozen && (opts->pc_type == 1 || (opts->pc_type == 2 && stiff > mg_from));
        // pipelined solver: latency-bound iterations only (small local problem), not in the tiny-h regime where the
        // Pythagorean norm update cancels heavily (|w|^2/h_n^2 ~ 1/stiff^2) and gmres() takes its explicit second pass
        // polynomial preconditioner in the mildly stiff regime (pc_type 2 = automatic, 3 = polynomial whenever useful)
        bool use_poly = false;
        if (!use_spec && !use_pc && h->use_frozen && (opts->pc_type == 2 || opts->pc_type == 3) && stiff >= 0.3) {
            if (!h->Zb && alloc_d(h, &h->Zb, (int64_t)h->restart_alloc * h->vlen)) { rc = KSFD_ENOMEM; goto out; }
            if (!lam_done) {
                if (h->lamJ < 0.0 || ++h->lam_age >= h->lam_period) {
                    const double before = h->lamJ;
                    if ((rc = est_lambda_max(h, shift, before < 0.0 ? 8 : 2))) goto out;    // warm-started after the first step
                    // J changes slowly from step to step: while the estimate moves by < 2 %, look less often
                    const bool stable = before > 0.0 && fabs(h->lamJ - before) <= 0.02 * before;
                    h->lam_period = stable ? std::min(2 * h->lam_period, 8) : 1;
                    h->lam_age = 0;
                }
                lam_done = true;
            }
            if (h->poly_shift != shift) poly_setup(h, shift);
            use_poly = h->poly_deg >= 1 && h->poly_max_deg >= 1;
        }
        const bool small = (double)h->G.F * (double)h->G.nloc <= 6.0e6;
        const bool use_async = !use_spec && !use_pc && !use_poly && h->use_frozen && !(opts->reserved & 1) && stiff >= 1e-3 &&
                               (!h->ring || h->tr->device_allreduce()) &&
                               (h->async_mode == 1 || (h->async_mode == 2 && small));
        h->mg_use32 = opts->ksp_rtol >= 1e-7;          // fp32 level vectors inside the V cycle (mg_vcycle32); tight tolerances keep fp64
        const bool fuse_stage = (fused_ok(h) || (strip3d_ok(h) && h->rhs3d_strip)) && h->P.nlig <= 4 && h->fuse_stage;
        const int its_before = st.linear_its;
        bool spec_failed = false;
        // Initial guesses for the spectral stage solves from the earlier stages of the step (A Y_j = b_j is known): the right-hand
        // sides of a step are nearly dependent -- b_1 = c b_0 to ~1e-3, later ones to a few per cent (CPU experiment with the oracle)
        // -- so x0 = sum c_j Y_j, c = argmin ||b_i - sum c_j b_j||, starts the defect correction 1-3 digits ahead for one small
        // multi-dot.  The b_j are kept in bstore (three vectors, allocated on first use); gb = their Gram matrix.
        const bool guess_on = (use_spec || use_pc) && fuse_stage && h->spec_guess && h->bstore;
        double gb[4][4];
        for (int i = 0; i < 4 && !rc; i++) {
            const double *zin = h->u;
            double bnorm2 = -1.0;                 // ||b||^2 when the RHS kernel's epilogue delivered it
            bool rhs_dots_done = false;           // ... and <b_i, b_j> for the stage guess with it
            double rhs_dot[2] = { 0.0, 0.0 };
            double *bcur = (guess_on && i < 3) ? h->bstore + (int64_t)i * vs : h->bvec;
            if (fuse_stage) {
                // stage argument and Zdot term folded into the RHS kernel (no Z vector, no separate passes)
                KComb cmb = KComb{};
                for (int j = 0; j < i; j++) {
                    if (h->At[i][j] != 0.0) { cmb.yin[cmb.nin] = h->Y + (int64_t)j * vs; cmb.ain[cmb.nin++] = h->At[i][j]; }
                    if (h->Ginv[i][j] != 0.0) { cmb.yout[cmb.nout] = h->Y + (int64_t)j * vs; cmb.aout[cmb.nout++] = -h->Ginv[i][j] / hh; }
                }
                // ||b||^2 from the store epilogue (2-D strip kernel), and with it the inner products of b_i with the right-hand sides the stage
                // guess is built from (the multi-dot below would read all of them again)
                const int gdot_j0 = std::max(0, i - h->guess_max), gdot_n = (guess_on && i > 0 && h->rhs_dots) ? i - gdot_j0 : 0;
                const bool rhs_norm = use_spec && fused_ok(h) && (!(guess_on && i > 0) || (gdot_n > 0 && gdot_n <= 2 && (long long)make_strips(h).nstrips * make_strips(h).nseg * (1 + gdot_n) <= part_capacity()));
                rhs_dots_done = rhs_norm && gdot_n > 0;
                // ghosts of the newest stage vector (earlier ones done): exchanged behind the interior rows of the RHS (op_rhs)
                if ((rc = op_rhs(h, h->u, i, bcur, &cmb, rhs_norm, i > 0 ? h->Y + (int64_t)(i - 1) * vs : nullptr, rhs_dots_done ? gdot_n : 0,
                                 rhs_dots_done ? h->bstore + (int64_t)gdot_j0 * vs : nullptr))) break;
                if (rhs_norm) bnorm2 = h->hres[0];
                if (rhs_dots_done) for (int j = 0; j < gdot_n; j++) rhs_dot[j] = h->hres[1 + j];
            } else {
            if (i > 0) {
                const double *xs[5]; double a[5]; int nt = 0;
                xs[nt] = h->u; a[nt++] = 1.0;
                for (int j = 0; j < i; j++) if (h->At[i][j] != 0.0) { xs[nt] = h->Y + (int64_t)j * vs; a[nt++] = h->At[i][j]; }
                if (nt > 1) {
                    if ((rc = op_lincomb(h, nt, xs, a, h->Z))) break;
                    if ((rc = halo(h, h->Z))) break;
                    zin = h->Z;
                }
            }
            if ((rc = op_rhs(h, zin, i, h->bvec))) break;
            if (i > 0) {
                const double *xs[5]; double a[5]; int nt = 0;
                xs[nt] = h->bvec; a[nt++] = 1.0;
                for (int j = 0; j < i; j++) if (h->Ginv[i][j] != 0.0) { xs[nt] = h->Y + (int64_t)j * vs; a[nt++] = -h->Ginv[i][j] / hh; }
                if (nt > 1 && (rc = op_lincomb(h, nt, xs, a, h->bvec))) break;
            }
            }
            LinStats ls;
            SpecGuess sg;
            sg.n = 0;
            {
                if (guess_on) {
                    if (i == 0) {
                        if (bnorm2 < 0.0) { if ((rc = op_multidot(h, bcur, bcur, 0))) break; bnorm2 = h->hres[0]; }
                        gb[0][0] = bnorm2;
                    }
                    else {
                        // <b_i, b_j> (j0 <= j < i) and <b_i, b_i> in one pass; least squares on the (ill-conditioned but tiny) Gram system.
                        // j0 > 0 (h->guess_max): only the most recent stages enter -- every vector of the guess costs two more full-vector
                        // reads in the first sweep (b_j in the forward row kernel, Y_j in the inverse one) and one in this multi-dot
                        const int j0 = std::max(0, i - h->guess_max), ng = i - j0;
                        if (rhs_dots_done) {
                            for (int j = 0; j < ng; j++) gb[i][j0 + j] = gb[j0 + j][i] = rhs_dot[j];
                            gb[i][i] = bnorm2;
                        } else {
                            if ((rc = op_multidot(h, bcur, h->bstore + (int64_t)j0 * vs, ng))) break;
                            for (int j = 0; j < ng; j++) gb[i][j0 + j] = gb[j0 + j][i] = h->hres[j];
                            gb[i][i] = bnorm2 = h->hres[ng];
                        }
                        double M[3][4];
                        for (int a = 0; a < ng; a++) { for (int c = 0; c < ng; c++) M[a][c] = gb[j0 + a][j0 + c]; M[a][ng] = gb[i][j0 + a]; M[a][a] *= 1.0 + 1e-13; }
                        bool okls = ng > 0;
                        for (int c = 0; c < ng && okls; c++) {              // Gaussian elimination with partial pivoting
                            int pv = c;
                            for (int a = c + 1; a < ng; a++) if (fabs(M[a][c]) > fabs(M[pv][c])) pv = a;
                            if (!(fabs(M[pv][c]) > 0.0)) { okls = false; break; }
                            for (int q = 0; q <= ng; q++) std::swap(M[c][q], M[pv][q]);
                            for (int a = c + 1; a < ng; a++) { const double f = M[a][c] / M[c][c]; for (int q = c; q <= ng; q++) M[a][q] -= f * M[c][q]; }
                        }
                        double cf[3] = { 0, 0, 0 };
                        for (int a = ng - 1; a >= 0 && okls; a--) { double t = M[a][ng]; for (int q = a + 1; q < ng; q++) t -= M[a][q] * cf[q]; cf[a] = t / M[a][a]; }
                        double pred = gb[i][i];                                // ||b_i - sum c_j b_j||^2 = b.b - 2 c.g + c.G c
                        for (int a = 0; a < ng; a++) { pred -= 2.0 * cf[a] * gb[i][j0 + a]; for (int c = 0; c < ng; c++) pred += cf[a] * cf[c] * gb[j0 + a][j0 + c]; }
                        if (okls && pred == pred && pred < 0.09 * gb[i][i]) {
                            for (int j = 0; j < ng; j++) if (cf[j] != 0.0) { sg.Y[sg.n] = h->Y + (int64_t)(j0 + j) * vs; sg.b[sg.n] = h->bstore + (int64_t)(j0 + j) * vs; sg.c[sg.n++] = cf[j]; }
                        }
                    }
                }
            }
            if (use_spec) {
                // defect correction with M^-1 (no Krylov vectors), flexible GMRES for the rest if it contracts slowly; the attempt
                // is capped so that a state it does not suit costs little, then the V cycle / plain GMRES takes over
                // (automatic choice: once a stage of this step has failed, the remaining stages go straight to the fallback, and a
                //  re-trial after a back-off period gets a short leash -- a state the preconditioner does not suit then costs one
                //  cheap attempt instead of four expensive ones: 440 -> 176 ms for such a step at 4096^2 x 3 fields)
                const bool skip = spec_failed && opts->pc_type == 2;
                if (!skip) {
                    rc = spec_solve(h, shift, bcur, h->Y + (int64_t)i * vs, opts, &ls, bnorm2, opts->pc_type == 2 ? (h->spec.backoff > 8 ? 8 : 40) : 0, sg.n ? &sg : nullptr);
                    st.pc_used |= 8;
                }
                if (skip || (rc == KSFD_ELINEAR && opts->pc_type == 2)) {
                    if (!skip) st.linear_its += ls.its;
                    spec_failed = true;
                    const bool mg_here = h->mg_ok && stiff > 0.3;
                    rc = gmres(h, h->u, shift, bcur, h->Y + (int64_t)i * vs, opts, &ls, mg_here ? 1 : 0);
                    st.pc_used |= mg_here ? 2 : 1;
                }
            } else if (use_pc && sg.n) {
                // multigrid regime, same initial guess: x0 = sum c_j Y_j, TRUE residual r0 = b - A x0 (one Jacobian action), then the
                // correction A d = r0 to the tolerance of the original system and x = x0 + d
                double *xi = h->Y + (int64_t)i * vs;
                const double *xs[3]; double a[3];
                for (int j = 0; j < sg.n; j++) { xs[j] = sg.Y[j]; a[j] = sg.c[j]; }
                if ((rc = op_lincomb(h, sg.n, xs, a, xi)) || (rc = halo(h, xi)) || (rc = op_jvp_frozen(h, xi, 2, shift, h->Z, bcur))) break;
                const double tol = std::max(opts->ksp_rtol * sqrt(bnorm2), opts->ksp_atol);
                rc = gmres(h, h->u, shift, h->Z, h->t3, opts, &ls, 1, i, tol);        // stage index: the Krylov spaces of the earlier stages are projected out first
                if (!rc) { const double *x2[2] = { xi, h->t3 }; double a2[2] = { 1.0, 1.0 }; rc = op_lincomb(h, 2, x2, a2, xi); }
                st.pc_used |= 2;
            } else {
            rc = use_async ? gmres_async(h, shift, bcur, h->Y + (int64_t)i * vs, opts, &ls)
                           : gmres(h, h->u, shift, bcur, h->Y + (int64_t)i * vs, opts, &ls, use_pc ? 1 : (use_poly ? 2 : 0), i);
            st.pc_used |= use_pc ? 2 : (use_poly ? 4 : 1);
            }
            st.linear_its += ls.its;
            st.ksp_resid = ls.rel;
            {
                static const bool stage_trace = getenv("KSFD_STAGE_TRACE") != nullptr;     // iterations per stage system (diagnostics)
                if (stage_trace) fprintf(stderr, "[stage %d] its %d rel %.2e guess %d\n", i, ls.its, ls.rel, sg.n);
            }
            if (rc == KSFD_ELINEAR && !use_pc && h->mg_ok && h->use_frozen && opts->pc_type) {
                // unpreconditioned GMRES ran out of iterations: the multigrid-preconditioned solve of the same system
                // is the remedy (the stiffness estimate above only knows the diffusion part of J)
                rc = gmres(h, h->u, shift, bcur, h->Y + (int64_t)i * vs, opts, &ls, 1);
                st.pc_used |= 2;
                st.linear_its += ls.its;
                st.ksp_resid = ls.rel;
            }
        }
        if (use_spec && opts->pc_type == 2) {
            // adaptation: > 12 iterations per stage system (or a failed attempt) means the coefficients vary too much for the
            // constant-coefficient inverse; leave it alone for a while (doubling) and let the polynomial / V cycle work
            if (spec_failed || st.linear_its - its_before > 48) {
                h->spec.bad_until = h->nsteps + h->spec.backoff;
                h->spec.backoff = std::min(2 * h->spec.backoff, 512);
            } else h->spec.backoff = 8;
        }
        if (!rc && use_pc && h->sf_auto) {
            // Shift floor of the multigrid hierarchy (gmres(): shift_pc = max(shift, floor)).  Once 1/(gamma h) has fallen below the
            // growth rate of the chemotactic instability the V cycle of shift*I - J stops contracting and the iteration count explodes
            // (options81 run, h ~ 350: 480 iterations per step; with a floor of 0.2: 120).  The right floor is a property of J we do
            // not know, so it is searched online: when a step needs > 64 iterations, double the floor while that pays (> 5 % fewer
            // iterations), else go back and try halving, else settle for 25 steps.
            const double its = (double)(st.linear_its - its_before);
            if (h->sf_dir == 0) {
                if (h->sf_hold > 0) h->sf_hold--;
                else if (its > 64.0) {
                    h->sf_prev_its = its; h->sf_prev_floor = h->mg_shift_floor;
                    h->mg_shift_floor = 2.0 * std::max(h->mg_shift_floor, shift);
                    h->sf_dir = 1; h->sf_tried_down = false;
                }
            } else if (its < 0.95 * h->sf_prev_its) {
                h->sf_prev_its = its; h->sf_prev_floor = h->mg_shift_floor;
                h->mg_shift_floor = h->sf_dir > 0 ? 2.0 * h->mg_shift_floor : 0.5 * h->mg_shift_floor;
                if (h->mg_shift_floor <= shift) { h->sf_dir = 0; h->sf_hold = 25; }          // floor no longer active
            } else {
                h->mg_shift_floor = h->sf_prev_floor;
                if (h->sf_dir > 0 && !h->sf_tried_down && 0.5 * h->sf_prev_floor > shift) {
                    h->sf_dir = -1; h->sf_tried_down = true;
                    h->mg_shift_floor = 0.5 * h->sf_prev_floor;
                } else { h->sf_dir = 0; h->sf_hold = 25; }
            }
        }
        if (rc == KSFD_ELINEAR && opts->adapt && !single && rejects < max_rej && hh * 0.25 >= opts->dt_min) {
            // PETSc's -ts_adapt_scale_solve_failed (0.25): a failed solve rejects the step and quarters it.  The
            // reference disables that by setMaxSNESFailures(1) (KSFD/ksfdts.py:135) because its LU cannot fail this
            // way; an iterative solve can, and aborting a long run for it would not be a service.
            rejects++;
            st.rejections = rejects;
            prev_accept = false;
            if ((rc = op_copy(h, h->u, h->usave))) goto out;
            hh *= 0.25;
            *hstep = hh;
            if ((rc = halo(h, h->u))) goto out;
            continue;
        }
        if (rc) { op_copy(h, h->u, h->usave); hipStreamSynchronize(h->st); goto out; }
        // completion + embedded error norm
        {
            Scope sc(h, KC_FINISH, vbytes(h, 7));
            hipLaunchKernelGGL(k_rosw_finish, dim3(h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, h->u, h->Y, vs,
                               h->bt[0], h->bt[1], h->bt[2], h->bt[3], h->b2t[0] - h->bt[0], h->b2t[1] - h->bt[1],
                               h->b2t[2] - h->bt[2], h->b2t[3] - h->bt[3], opts->atol, opts->rtol, h->errv, h->part);
        }
        h->have_err = true;
        h->coef_fresh = false;                                   // u <- u_new (a rollback below makes the planes current again)
        if ((rc = reduce_rows(h, 1, h->nblk_vec, 0))) goto out;
        {
            double ntot = (double)h->G.F * (double)h->cfg.n[0] * (double)h->cfg.n[1] * (double)h->cfg.n[2];
            st.wrms = sqrt(h->hres[0] / ntot);
        }
        if (!(st.wrms == st.wrms) || isinf(st.wrms)) {
            op_copy(h, h->u, h->usave); hipStreamSynchronize(h->st);
            h->coef_fresh = h->use_frozen;
            rc = fail(h, KSFD_ENAN, "non-finite error norm at t=%g h=%g", *t, hh);
            goto out;
        }
        bool accept = true;
        double hnext = hh;
        if (opts->adapt) {
            // TSAdaptChoose_Basic
            double safety = opts->safety;
            if (st.wrms > 1.0) {
                if (!prev_accept) safety *= opts->reject_safety;
                accept = hh < (1.0 + 1.4901161193847656e-08) * opts->dt_min;   // at minimum step: accept anyway
            }
            double hfac = st.wrms > 0.0 ? safety * pow(st.wrms, -1.0 / 3.0) : INFINITY;
            hfac = std::min(std::max(hfac, opts->clip_lo), opts->clip_hi);
            hnext = std::min(std::max(hh * hfac, opts->dt_min), opts->dt_max);
        }
        prev_accept = accept;
        if (accept) {
            st.accepted = 1; st.h_used = hh;
            *t += hh;
            *hstep = hnext;
            break;
        }
        rejects++;
        st.rejections = rejects;
        if ((rc = op_copy(h, h->u, h->usave))) goto out;
        h->coef_fresh = h->use_frozen;                           // the planes were made from exactly this state
        hh = hnext;
        *hstep = hnext;
        if (single) break;                                       // single attempt: report the rejection
        if (rejects > max_rej) { rc = fail(h, KSFD_EREJECT, "step rejected %d times at t=%g", rejects, *t); break; }
        if ((rc = halo(h, h->u))) goto out;
    }
out:
    prof_resolve(h);
    st.bytes = h->bytes_acc - bytes0;
    st.rhs_evals = (int32_t)(h->prof.launches[KC_RHS] - rhs0);
    st.jvp_evals = (int32_t)(h->prof.launches[KC_JVP] - jvp0);
    {
        int64_t l1 = 0;
        for (int c = 0; c < KSFD_NKCLASS; c++) l1 += h->prof.launches[c];
        st.launches = (int32_t)(l1 - launches0);
    }
    st.host_syncs = (int32_t)(h->n_host_sync - sync0);
    st.predicted_final = (int32_t)(h->n_predicted - pred0);
    st.residual_evals = (int32_t)(h->n_residual - resid0);
    if (stats) *stats = st;
    return rc;
}

extern "C" int ksfd_get_last_error_vector(ksfd_handle *h, double *eh, int32_t layout)
{
    if (!h || !eh) return KSFD_EINVAL;
    if (!h->have_err) return fail(h, KSFD_EINVAL, "no step has been attempted yet");
    hipSetDevice(h->device);
    return download(h, h->errv, layout, eh);
}

extern "C" int ksfd_set_profiling(ksfd_handle *h, int32_t on)
{
    if (!h) return KSFD_EINVAL;
    prof_resolve(h);
    if (on < 0 || on >= 2 + KSFD_NKCLASS) return KSFD_EINVAL;
    h->profiling = on != 0;
    h->prof_only = on >= 2 ? on - 2 : -1;
    return KSFD_OK;
}
extern "C" int ksfd_get_profile(ksfd_handle *h, ksfd_profile *p, int32_t reset)
{
    if (!h || !p) return KSFD_EINVAL;
    prof_resolve(h);
    *p = h->prof;
    if (reset) memset(&h->prof, 0, sizeof h->prof);
    return KSFD_OK;
}
extern "C" int ksfd_set_mg_params(ksfd_handle *h, int32_t nu, int32_t ncoarse_max, int32_t power_its, double ratio, double coarse_tol)
{
    if (!h) return KSFD_EINVAL;
    if (nu > 0) h->mg_nu = nu;
    if (ncoarse_max > 0) h->mg_ncoarse = ncoarse_max;
    if (power_its > 0) h->mg_power_its = power_its;
    if (ratio > 1.0) h->mg_ratio = ratio;
    if (coarse_tol > 0.0) h->mg_coarse_tol = coarse_tol;
    h->mg_use_graph = power_its != -7 && !h->ring;     // power_its = -7: eager launches (debug / A-B timing); slab ranks: collectives inside the cycle
    h->mg_shift = -1.0;
    return KSFD_OK;
}
extern "C" int ksfd_set_poly_params(ksfd_handle *h, int32_t max_degree, double target, double mg_threshold)
{
    if (!h || max_degree < 0 || max_degree > 7) return KSFD_EINVAL;
    if (mg_threshold > 0.0) h->mg_threshold = mg_threshold;
    h->poly_max_deg = max_degree;           // 0 disables the polynomial preconditioner
    if (target > 0.0 && target < 1.0) h->poly_target = target;
    h->poly_shift = -1.0;
    return KSFD_OK;
}
extern "C" int ksfd_spectral_apply(ksfd_handle *h, double shift, const double *vh, double *outh, int32_t layout)
{
    if (!h || !vh || !outh || !(shift > 0.0)) return KSFD_EINVAL;
    hipSetDevice(h->device);
    if (!h->spec.ok) return fail(h, KSFD_EINVAL, "spectral preconditioner not available for this handle (needs power-of-two extents of 32 ... 16384 points; 1, 2, 4 or 8 slab ranks with an all-to-all transport)");
    int rc;
    if ((rc = upload(h, vh, layout, h->t2))) return rc;
    if ((rc = ensure_coef(h))) return rc;
    if ((rc = spec_means(h)) || (rc = spec_apply(h, shift, h->t2, h->t3))) return rc;
    return download(h, h->t3, layout, outh);
}
extern "C" int ksfd_set_spectral_params(ksfd_handle *h, double from_stiffness, int32_t enable)
{
    if (!h) return KSFD_EINVAL;
    if (from_stiffness > 0.0) h->spec_from = from_stiffness;
    if (enable == 0) h->spec.user_off = true;
    else if (enable > 0) { h->spec.user_off = false; h->spec.bad_until = 0; h->spec.backoff = 8; }
    return KSFD_OK;
}
extern "C" int ksfd_synchronize(ksfd_handle *h)
{
    if (!h) return KSFD_EINVAL;
    hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->st));
    return KSFD_OK;
}
extern "C" int ksfd_set_tuning(ksfd_handle *h, int32_t use_fused, int32_t yseg, int32_t yseg_jvp)
{
    if (!h) return KSFD_EINVAL;
    if (use_fused >= 0) {
        h->use_fused = use_fused & 1; h->use_frozen = !(use_fused & 2); h->overlap = !(use_fused & 4);
        h->async_mode = (use_fused & 8) ? 1 : 0;
        h->rec_mode = (use_fused & 16) ? 0 : ((use_fused & 32) ? 2 : 1);
        if ((use_fused >> 6) & 7) h->rec_keep = std::min((use_fused >> 6) & 7, 4);
        h->poly_fp32 = !(use_fused & 512);
        h->fuse_stage = !(use_fused & 1024);
        h->zero_copy = !(use_fused & 2048) && h->hres_dev;
        h->rhs3d_strip = !(use_fused & 8192);
        h->spec_guess = !(use_fused & 16384);
        h->rhs_carry = !(use_fused & 32768);
        h->spec_predict = !(use_fused & 65536);
        h->restart_grow = !(use_fused & 131072);
        h->mg_warm_power = !(use_fused & 262144);
        h->mg_fp32 = !(use_fused & 524288);
        h->rec_mg = (use_fused & 1048576) != 0;
        h->rhs_dots = !(use_fused & 2097152);
        if (h->mg_fuse != !(use_fused & 4096)) { h->mg_fuse = !(use_fused & 4096); h->mg_shift = -1.0; if (h->mg_graph) { hipGraphExecDestroy(h->mg_graph); h->mg_graph = nullptr; } }
    }
    if (yseg > 0) h->yseg = yseg;
    if (yseg_jvp > 0) h->yseg_jvp = yseg_jvp;
    return KSFD_OK;
}

// Raw timing of one kernel class on the current state, HIP events on the compute stream.
// cls: KC_RHS, KC_JVP, KC_MULTIDOT (k = 8 basis vectors), KC_GSUPDATE (k = 8), KC_LINCOMB (3 inputs).
extern "C" int ksfd_bench_kernel(ksfd_handle *h, int32_t cls, int32_t reps, double *avg_ms, double *bytes_per_launch)
{
    if (!h || reps < 1 || !avg_ms) return KSFD_EINVAL;
    hipSetDevice(h->device);
    hipEvent_t a, b;
    HIPCHK(h, hipEventCreate(&a));
    HIPCHK(h, hipEventCreate(&b));
    const bool was = h->profiling;
    h->profiling = false;
    int rc = KSFD_OK;
    double by = 0.0;
    auto one = [&]() -> int {
        const double b0 = h->bytes_acc;
        int r = KSFD_OK;
        double coef[KSFD_MAXDOT] = { 0 };
        switch (cls) {
        case KC_RHS: r = op_rhs(h, h->u, -1, h->t3); break;
        case KC_JVP: r = h->use_frozen ? op_jvp_frozen(h, h->Y, 1, 1.0, h->t3) : op_jvp(h, h->u, h->Y, 1, 1.0, h->t3); break;
        case KC_MULTIDOT: {
            Scope sc(h, KC_MULTIDOT, vbytes(h, 9));
            VW_DISPATCH(h, hipLaunchKernelGGL((k_multidot<8, VW>), dim3(vec2(h) ? (h->nblk_vec + 1) / 2 : h->nblk_vec), dim3(KSFD_BLOCK), 0, h->st, h->kv, (const double *)h->t3, (const double *)h->V, h->vlen, 8, h->part));
        } break;
        case KC_GSUPDATE: r = op_gs_update(h, h->t3, h->V, 8, coef, 1.0); break;
        case KC_SPECTRAL: r = spec_apply(h, 10.0, h->Y, h->t3); break;
        case KC_LINCOMB: { const double *xs[3] = { h->u, h->Y, h->Y + h->vlen }; double aa[3] = { 1.0, 0.5, 0.25 }; r = op_lincomb(h, 3, xs, aa, h->t3); } break;
        default: r = fail(h, KSFD_EINVAL, "bench_kernel: class %d not benchable", cls);
        }
        by = h->bytes_acc - b0;
        return r;
    };
    if ((rc = halo(h, h->u))) goto done;
    if (h->use_frozen && (cls == KC_JVP || cls == KC_SPECTRAL) && (rc = ensure_coef(h, true))) goto done;
    if (cls == KC_SPECTRAL && (rc = spec_means(h))) goto done;
    for (int i = 0; i < 3 && !rc; i++) rc = one();
    if (rc) goto done;
    hipEventRecord(a, h->st);
    for (int i = 0; i < reps && !rc; i++) rc = one();
    hipEventRecord(b, h->st);
    if (hipEventSynchronize(b) != hipSuccess) rc = fail(h, KSFD_EHIP, "event sync failed");
    if (!rc) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        *avg_ms = ms / reps;
        if (bytes_per_launch) *bytes_per_launch = by;
    }
done:
    h->profiling = was;
    hipEventDestroy(a);
    hipEventDestroy(b);
    return rc;
}
