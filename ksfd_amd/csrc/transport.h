// Halo / reduction transports behind the slab decomposition (stand-in for the PETSc DMDA ghost
// exchange and MPI allreduce of the reference: KSFD/ksfdgrid.py:388-411, KSFD/ksfdts.py:310-313).
//
//  * RcclTransport: ncclSend/ncclRecv ring-neighbour exchange + ncclAllReduce over xGMI, issued on the
//    library's compute stream.  librccl is resolved at run time with dlopen (first the copy already
//    loaded into the process, e.g. by torch), so the single-GPU path carries no RCCL dependency.
//  * CallbackTransport: the host language does the exchange (mpi4py, torch.distributed/gloo, ...) on
//    pinned host staging buffers.  Slower, but works with any launcher and on one shared GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string>
#include <string.h>
#include <vector>

#include "../../include/ksfd_hip.h"

// one contiguous piece of an all-to-all: the k-th piece sent to a peer lands in the k-th piece that peer receives from us
struct A2APiece { int peer; void *ptr; size_t bytes; };

struct Transport {
    virtual ~Transport() {}
    // personalised all-to-all of device memory (transposes of the slab-distributed spectral solver); every rank passes the same
    // number and sizes of pieces per peer; pieces addressed to the own rank are copied on the device
    virtual bool has_alltoall() const { return false; }
    virtual int alltoall(const std::vector<A2APiece> &sends, const std::vector<A2APiece> &recvs, hipStream_t st) { (void)sends; (void)recvs; (void)st; err = "all-to-all not available on this transport"; return 1; }
    // fill the 2*ng ghost units of every field plane of vec from the ring neighbours
    virtual int exchange(double *vec, int F, long long plane, long long inner, long long sloc, int ng, hipStream_t st) = 0;
    virtual int allreduce(double *dev, int n, int op, hipStream_t st) = 0;   // op 0 sum, 1 max; in place
    virtual bool result_on_host() const { return false; }
    virtual bool device_allreduce() const { return true; }   // allreduce is stream-ordered and leaves the result on the device
    virtual const double *host_result() const { return nullptr; }
    const std::string &error() const { return err; }
    std::string err;
};

// ------------------------------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string &err)
    {
        const char *names[] = { "librccl.so.1", "librccl.so" };
        for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }
        if (!lib) for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
#define KSFD_SYM(f) f = (decltype(f))dlsym(lib, "nccl" #f); if (!f) { err = "librccl lacks nccl" #f; return false; }
        KSFD_SYM(CommInitRank) KSFD_SYM(CommDestroy) KSFD_SYM(GroupStart) KSFD_SYM(GroupEnd)
        KSFD_SYM(Send) KSFD_SYM(Recv) KSFD_SYM(AllReduce) KSFD_SYM(GetErrorString)
#undef KSFD_SYM
        return true;
    }
};

struct RcclTransport : Transport {
    RcclApi api;
    ncclComm_t comm = nullptr;
    int rank, size;
    bool init(const ksfd_dist *d)
    {
        rank = d->rank; size = d->size;
        if (!d->nccl_id) { err = "transport 1 needs the 128-byte ncclUniqueId"; return false; }
        if (!api.load(err)) return false;
        ncclUniqueId id;
        memcpy(&id, d->nccl_id, sizeof id);
        ncclResult_t r = api.CommInitRank(&comm, size, id, rank);
        if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + api.GetErrorString(r); comm = nullptr; return false; }
        return true;
    }
    ~RcclTransport() override { if (comm) api.CommDestroy(comm); }
    int chk(ncclResult_t r, const char *what)
    {
        if (r == ncclSuccess) return 0;
        err = std::string(what) + ": " + api.GetErrorString(r);
        return 1;
    }
    int exchange(double *vec, int F, long long plane, long long inner, long long sloc, int ng, hipStream_t st) override
    {
        const int lo = (rank + size - 1) % size, hi = (rank + 1) % size;
        const size_t cnt = (size_t)ng * inner;
        if (chk(api.GroupStart(), "ncclGroupStart")) return 1;
        for (int c = 0; c < F; c++) {
            double *p = vec + (long long)c * plane;
            // my low rows -> lo neighbour's high ghost; my high rows -> hi neighbour's low ghost.
            // Receives are posted hi-ghost first so that with size == 2 (lo == hi) the two messages
            // of the single peer pair up in issue order.
            if (chk(api.Send(p + (long long)ng * inner, cnt, ncclDouble, lo, comm, st), "ncclSend")) return 1;
            if (chk(api.Send(p + sloc * inner, cnt, ncclDouble, hi, comm, st), "ncclSend")) return 1;
            if (chk(api.Recv(p + (sloc + ng) * inner, cnt, ncclDouble, hi, comm, st), "ncclRecv")) return 1;
            if (chk(api.Recv(p, cnt, ncclDouble, lo, comm, st), "ncclRecv")) return 1;
        }
        return chk(api.GroupEnd(), "ncclGroupEnd");
    }
    int allreduce(double *dev, int n, int op, hipStream_t st) override
    {
        return chk(api.AllReduce(dev, dev, (size_t)n, ncclDouble, op ? ncclMax : ncclSum, comm, st), "ncclAllReduce");
    }
    bool has_alltoall() const override { return true; }
    int alltoall(const std::vector<A2APiece> &sends, const std::vector<A2APiece> &recvs, hipStream_t st) override
    {
        // own pieces: device copies in order; everything else: one group of point-to-point transfers (xGMI links in parallel)
        size_t ks = 0, kr = 0;
        while (true) {
            while (ks < sends.size() && sends[ks].peer != rank) ks++;
            while (kr < recvs.size() && recvs[kr].peer != rank) kr++;
            if (ks >= sends.size() || kr >= recvs.size()) break;
            if (sends[ks].bytes != recvs[kr].bytes) { err = "all-to-all: own pieces do not pair up"; return 1; }
            if (hipMemcpyAsync(recvs[kr].ptr, sends[ks].ptr, sends[ks].bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) { err = "all-to-all: device copy failed"; return 1; }
            ks++; kr++;
        }
        if (chk(api.GroupStart(), "ncclGroupStart")) return 1;
        for (const A2APiece &p : sends) if (p.peer != rank && chk(api.Send(p.ptr, p.bytes, ncclChar, p.peer, comm, st), "ncclSend")) return 1;
        for (const A2APiece &p : recvs) if (p.peer != rank && chk(api.Recv(p.ptr, p.bytes, ncclChar, p.peer, comm, st), "ncclRecv")) return 1;
        return chk(api.GroupEnd(), "ncclGroupEnd");
    }
};

// ------------------------------------------------------------------------------------------------
struct CallbackTransport : Transport {
    ksfd_exchange_fn ex = nullptr;
    ksfd_allreduce_fn ar = nullptr;
    ksfd_alltoall_fn a2a = nullptr;
    void *ctx = nullptr;
    int nranks = 1;
    char *a2a_send = nullptr, *a2a_recv = nullptr;   // pinned, grown on demand
    size_t a2a_cap = 0;
    double *stage = nullptr;      // pinned: [send_lo | send_hi | recv_lo | recv_hi], each F*ng*inner
    double *hred = nullptr;       // pinned, 128 doubles (as the handle's result buffers)
    size_t chunk = 0;
    bool init(const ksfd_dist *d, int F, long long inner)
    {
        ex = d->exchange; ar = d->allreduce; a2a = d->alltoall; ctx = d->ctx; nranks = d->size;
        if (!ex || !ar) { err = "transport 2 needs exchange and allreduce callbacks"; return false; }
        chunk = (size_t)(F + 2) * 2 * inner;                  // F fields (+2: the 3+n coefficient planes of a coarse level fit too)
        if (hipHostMalloc((void **)&stage, sizeof(double) * chunk * 4, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&hred, sizeof(double) * 128, hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc failed"; return false; }
        return true;
    }
    ~CallbackTransport() override { if (stage) hipHostFree(stage); if (hred) hipHostFree(hred); if (a2a_send) hipHostFree(a2a_send); if (a2a_recv) hipHostFree(a2a_recv); }
    bool has_alltoall() const override { return a2a != nullptr; }
    int alltoall(const std::vector<A2APiece> &sends, const std::vector<A2APiece> &recvs, hipStream_t st) override
    {
        // pack the pieces per peer into pinned host memory (peer-major, piece order kept), let the host language move the
        // blocks, unpack.  Slow by construction (two PCIe crossings): rehearsal / generic-launcher path
        size_t per_peer = 0;
        for (const A2APiece &p : sends) if (p.peer == 0) per_peer += p.bytes;
        const size_t total = per_peer * (size_t)nranks;
        if (total > a2a_cap) {
            if (a2a_send) hipHostFree(a2a_send);
            if (a2a_recv) hipHostFree(a2a_recv);
            a2a_send = a2a_recv = nullptr; a2a_cap = 0;
            if (hipHostMalloc((void **)&a2a_send, total, hipHostMallocDefault) != hipSuccess || hipHostMalloc((void **)&a2a_recv, total, hipHostMallocDefault) != hipSuccess) { err = "all-to-all: hipHostMalloc failed"; return 1; }
            a2a_cap = total;
        }
        std::vector<size_t> off((size_t)nranks, 0);
        for (const A2APiece &p : sends) {
            if (off[p.peer] + p.bytes > per_peer) { err = "all-to-all: uneven pieces"; return 1; }
            if (hipMemcpyAsync(a2a_send + (size_t)p.peer * per_peer + off[p.peer], p.ptr, p.bytes, hipMemcpyDeviceToHost, st) != hipSuccess) { err = "all-to-all D2H failed"; return 1; }
            off[p.peer] += p.bytes;
        }
        if (hipStreamSynchronize(st) != hipSuccess) { err = "all-to-all: stream sync failed"; return 1; }
        if (a2a(ctx, a2a_send, a2a_recv, (int64_t)per_peer)) { err = "all-to-all callback reported failure"; return 1; }
        std::fill(off.begin(), off.end(), 0);
        for (const A2APiece &p : recvs) {
            if (hipMemcpyAsync(p.ptr, a2a_recv + (size_t)p.peer * per_peer + off[p.peer], p.bytes, hipMemcpyHostToDevice, st) != hipSuccess) { err = "all-to-all H2D failed"; return 1; }
            off[p.peer] += p.bytes;
        }
        return 0;
    }
    int exchange(double *vec, int F, long long plane, long long inner, long long sloc, int ng, hipStream_t st) override
    {
        const size_t w = sizeof(double) * (size_t)ng * inner;     // bytes per field per side
        const size_t cnt = (size_t)F * ng * inner;                // doubles per side in THIS call (coarse multigrid levels are smaller)
        if (cnt > chunk) { err = "halo larger than the staging buffers"; return 1; }
        double *slo = stage, *shi = stage + chunk, *rlo = stage + 2 * chunk, *rhi = stage + 3 * chunk;
        hipError_t e;
        e = hipMemcpy2DAsync(slo, w, vec + (long long)ng * inner, sizeof(double) * plane, w, F, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpy2DAsync(shi, w, vec + sloc * inner, sizeof(double) * plane, w, F, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { err = std::string("halo D2H: ") + hipGetErrorString(e); return 1; }
        if (ex(ctx, slo, shi, rlo, rhi, (int64_t)cnt)) { err = "exchange callback reported failure"; return 1; }
        e = hipMemcpy2DAsync(vec, sizeof(double) * plane, rlo, w, w, F, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpy2DAsync(vec + (sloc + ng) * inner, sizeof(double) * plane, rhi, w, w, F, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { err = std::string("halo H2D: ") + hipGetErrorString(e); return 1; }
        return 0;
    }
    int allreduce(double *dev, int n, int op, hipStream_t st) override
    {
        if (n > 128) { err = "allreduce of more than 128 doubles"; return 1; }
        hipError_t e = hipMemcpyAsync(hred, dev, sizeof(double) * n, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { err = std::string("allreduce D2H: ") + hipGetErrorString(e); return 1; }
        if (ar(ctx, hred, n, op)) { err = "allreduce callback reported failure"; return 1; }
        return 0;
    }
    bool result_on_host() const override { return true; }
    bool device_allreduce() const override { return false; }
    const double *host_result() const override { return hred; }
};

static Transport *make_transport(const ksfd_dist *d, int F, long long inner, std::string &err)
{
    if (d->transport == 1) {
        RcclTransport *t = new RcclTransport();
        if (!t->init(d)) { err = t->err; delete t; return nullptr; }
        return t;
    }
    if (d->transport == 2) {
        CallbackTransport *t = new CallbackTransport();
        if (!t->init(d, F, inner)) { err = t->err; delete t; return nullptr; }
        return t;
    }
    err = "size > 1 needs transport 1 (RCCL) or 2 (callbacks)";
    return nullptr;
}
