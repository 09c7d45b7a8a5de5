// Row-partitioned Lanczos over the GPUs of one node from ONE host process: the C-callable form of
// the multi-GPU path.  The reference has no multi-GPU code (its GPU solver is one device, one
// stream, src/lanczos_cuda.c:364); what this must agree with is the single-GPU result.
//
// The matrix is cut into contiguous row slabs, one engine per slab.  Per iteration the slabs
// exchange the current Lanczos vector -- the entries the neighbouring slabs need are received
// straight into the tail of the vector the SpMV reads (banded matrices), or every slab gathers the
// whole vector (anything else) -- and all-reduce two small buffers: alpha, then the
// reorthogonalisation coefficients together with w.w.  Without reorthogonalisation the basis stays
// unnormalised and ONE reduction of two doubles per iteration is enough.
//
// Two transports behind the same iteration code:
//   RCCL      one communicator per device (ncclCommInitAll), grouped ncclSend/ncclRecv for the
//             halo on a second stream per device so that it runs beside the interior rows'
//             product, ncclAllReduce on the engine's stream.  librccl is opened at run time: a
//             process that never asks for several GPUs does not need it.
//   loopback  every slab on one device and one stream, copies and a summing kernel in place of
//             the collectives: the partition, column remap and iteration order on a one-GPU box.
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace {

// ---------------------------------------------------------------------------
// librccl, resolved on first use
// ---------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_rccl(Rccl &r) {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
    }
    if (!r.lib) return false;
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(r.lib, n);
        ok = ok && p != nullptr;
        return p;
    };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(r.lib);
        r.lib = nullptr;
    }
    return ok;
}

// the library's entry points, or nullptr when librccl cannot be opened (looked up once, thread safe)
Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { load_rccl(r); });
    return r.lib ? &r : nullptr;
}

#define M_HIP(call)                                                                                   \
    do {                                                                                              \
        hipError_t st__ = (call);                                                                     \
        if (st__ != hipSuccess)                                                                       \
            return lz::fail("HIP API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,        \
                            hipGetErrorString(st__), (int)st__);                                      \
    } while (0)
// the reference's CHECK_* style (src/lanczos_cuda.h:76-124): message to stdout, EXIT_FAILURE
#define M_RCCL(call)                                                                                  \
    do {                                                                                              \
        ncclResult_t st__ = (call);                                                                   \
        if (st__ != ncclSuccess)                                                                      \
            return lz::fail("RCCL API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,       \
                            rccl()->GetErrorString(st__), (int)st__);                                 \
    } while (0)
#define M_TRY(call)                \
    do {                           \
        if (int rc__ = (call)) return rc__; \
    } while (0)

int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Every entry point walks over the slabs' devices; the caller gets back the device (and the slab
// note) it came in with, whichever slab was touched last.
struct DeviceRestore {
    int dev = -1, slab = -1;
    DeviceRestore() {
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        slab = lz::tl_current_slab;
    }
    ~DeviceRestore() {
        if (dev >= 0) (void)hipSetDevice(dev);
        lz::tl_current_slab = slab;
    }
};

// loopback all-reduce: bufs[p][i] <- sum_q bufs[q][i] for every p, slabs added in ascending order
__global__ void __launch_bounds__(256) k_sum_slabs(double *const *__restrict__ bufs, int slabs, int count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int q = 0; q < slabs; ++q) s += bufs[q][i];
    for (int q = 0; q < slabs; ++q) bufs[q][i] = s;
}

}  // namespace

// ---------------------------------------------------------------------------
// the transport on its own (engine.h): what the block variant's row slabs use
// ---------------------------------------------------------------------------
namespace {
constexpr int kMaxSlabs = 64;
struct SlabPtrs { double *p[kMaxSlabs]; };
__global__ void __launch_bounds__(256) k_sum_slab_args(SlabPtrs bufs, int slabs, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = 0.0;
    for (int q = 0; q < slabs; ++q) s += bufs.p[q][i];
    for (int q = 0; q < slabs; ++q) bufs.p[q][i] = s;
}
}  // namespace

namespace lz {

struct SlabComm {
    int P = 0;
    bool loopback = false;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;
};

int slab_comm_create(SlabComm **out, int slabs, const int *devices, bool loopback) {
    *out = nullptr;
    if (slabs < 1 || slabs > kMaxSlabs || !devices) return fail("slab transport: %d slabs", slabs);
    SlabComm *c = new SlabComm();
    c->P = slabs;
    c->loopback = loopback;
    c->dev.assign(devices, devices + slabs);
    if (!loopback) {
        if (!rccl()) {
            delete c;
            return fail("slab transport: librccl.so could not be opened (several GPUs need it; one GPU can rehearse with the "
                        "loopback transport)");
        }
        c->comm.assign((size_t)slabs, nullptr);
        const ncclResult_t st = rccl()->CommInitAll(c->comm.data(), slabs, c->dev.data());
        if (st != ncclSuccess) {
            c->comm.clear();
            delete c;
            return fail("slab transport: ncclCommInitAll over %d devices: %s", slabs, rccl()->GetErrorString(st));
        }
    }
    *out = c;
    return 0;
}

void slab_comm_destroy(SlabComm *c) {
    if (!c) return;
    for (ncclComm_t x : c->comm)
        if (x && rccl()) (void)rccl()->CommDestroy(x);
    delete c;
}

int slab_all_reduce(SlabComm *c, double *const *bufs, size_t count, const hipStream_t *streams) {
    if (count == 0) return 0;
    if (c->loopback) {
        if (c->P == 1) return 0;
        SlabPtrs a;
        for (int p = 0; p < c->P; ++p) a.p[p] = bufs[p];
        hipLaunchKernelGGL(k_sum_slab_args, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, streams[0], a, c->P, count);
        M_HIP(hipGetLastError());
        return 0;
    }
    M_RCCL(rccl()->GroupStart());
    for (int p = 0; p < c->P; ++p) {
        M_HIP(hipSetDevice(c->dev[(size_t)p]));
        M_RCCL(rccl()->AllReduce(bufs[p], bufs[p], count, ncclDouble, ncclSum, c->comm[(size_t)p], streams[p]));
    }
    M_RCCL(rccl()->GroupEnd());
    return 0;
}

int slab_all_gather(SlabComm *c, double *const *tables, size_t stride, size_t count, const hipStream_t *streams) {
    if (count > stride) return fail("slab transport: piece of %zu doubles in places of %zu", count, stride);
    if (c->loopback) {
        for (int p = 0; p < c->P; ++p)
            for (int q = 0; q < c->P; ++q)
                if (q != p)
                    M_HIP(hipMemcpyAsync(tables[p] + (size_t)q * stride, tables[q] + (size_t)q * stride, sizeof(double) * count,
                                         hipMemcpyDeviceToDevice, streams[0]));
        return 0;
    }
    M_RCCL(rccl()->GroupStart());
    for (int p = 0; p < c->P; ++p) {
        M_HIP(hipSetDevice(c->dev[(size_t)p]));
        M_RCCL(rccl()->AllGather(tables[p] + (size_t)p * stride, tables[p], stride, ncclDouble, c->comm[(size_t)p], streams[p]));
    }
    M_RCCL(rccl()->GroupEnd());
    return 0;
}

}  // namespace lz

struct lanczos_hip_multi {
    int P = 0;
    bool loopback = false;
    bool halo = false;       // neighbour exchange into the vectors' tails; else all-gather into x_ext
    bool overlap = false;    // the exchange runs beside the interior rows' product
    int64_t n = 0, nnz = 0;
    int max_iter = 0, reorth = 0, nvec = 0;
    int k = 0;
    int closed_k = 0;        // three-vector mode: iterations whose beta is recorded
    int reorth_steps = 0;
    std::vector<int> dev;
    std::vector<int64_t> bounds;                      // slab p = rows [bounds[p], bounds[p+1])
    std::vector<lz::Engine *> eng;
    std::vector<hipStream_t> stream, cstream;         // compute, exchange (RCCL transport only)
    std::vector<hipEvent_t> ev_ready, ev_done;
    std::vector<double *> basis, x_ext, red;          // device memory owned here
    std::vector<int64_t> ld, base, halo_lo, halo_hi, send_lo, send_hi, in_lo, in_hi;
    int64_t pad = 0;                                  // all-gather: entries per slab in x_ext
    std::vector<ncclComm_t> comm;    // all-reduce / all-gather, on the compute streams
    std::vector<ncclComm_t> xcomm;   // halo sends and receives, on the exchange streams: a communicator of
                                     // their own, so that operations issued on two streams of one device
                                     // never share a communicator
    double **d_red_ptrs = nullptr;                    // loopback: the slabs' reduction buffers
    lanczos_hip_omega *omega = nullptr;               // selective mode: one tracker decides for all slabs
    double *h_sel = nullptr;                          // pinned: alpha_k, |w|^2 for that decision
    double last_beta = 0.0;
    double build_sec = 0.0;  // column remap, upload and engine set-up of all slabs
    int build_threads = 1;   // host threads that did it

    ~lanczos_hip_multi();
    // slab p's device becomes current (and is noted as the current slab: lz::check_affinity)
    int use(int p) const {
        if (hipSetDevice(dev[(size_t)p]) != hipSuccess) return lz::fail("hipSetDevice(%d) failed", dev[(size_t)p]);
        lz::tl_current_slab = p;
        return 0;
    }
    double *vec(int p, int j) const {
        return basis[(size_t)p] + (size_t)(reorth == LANCZOS_HIP_REORTH_NONE ? j % 3 : j) * (size_t)ld[(size_t)p];
    }
    int build(int ngpu, const int *devices, int64_t n_, int64_t nnz_, const int64_t *rp, const int64_t *ci,
              const double *va, int max_iter_, const lanczos_hip_options *opt, int flags);
    int set_start(const double *v0);
    int all_reduce(int count);
    int exchange_begin();
    int exchange_end();
    int step(bool close_beta);
    int close_lazy();
    int run(int steps, double *alpha_out, double *beta_out);
};

lanczos_hip_multi::~lanczos_hip_multi() {
    for (size_t p = 0; p < eng.size(); ++p) {
        if (p < dev.size()) (void)hipSetDevice(dev[p]);
        if (p < stream.size() && stream[p]) (void)hipStreamSynchronize(stream[p]);
        if (p < cstream.size() && cstream[p]) (void)hipStreamSynchronize(cstream[p]);
        delete eng[p];
    }
    for (size_t p = 0; p < comm.size(); ++p)
        if (comm[p] && rccl()) (void)rccl()->CommDestroy(comm[p]);
    for (size_t p = 0; p < xcomm.size(); ++p)
        if (xcomm[p] && rccl()) (void)rccl()->CommDestroy(xcomm[p]);
    for (size_t p = 0; p < dev.size(); ++p) {
        (void)hipSetDevice(dev[p]);
        if (p < basis.size()) (void)hipFree(basis[p]);
        if (p < x_ext.size()) (void)hipFree(x_ext[p]);
        if (p < red.size()) (void)hipFree(red[p]);
        if (p < ev_ready.size() && ev_ready[p]) (void)hipEventDestroy(ev_ready[p]);
        if (p < ev_done.size() && ev_done[p]) (void)hipEventDestroy(ev_done[p]);
        if (p < cstream.size() && cstream[p]) (void)hipStreamDestroy(cstream[p]);
        if (p < stream.size() && stream[p] && (!loopback || p == 0)) (void)hipStreamDestroy(stream[p]);
    }
    if (d_red_ptrs) (void)hipFree(d_red_ptrs);
    if (h_sel) (void)hipHostFree(h_sel);
    lanczos_hip_omega_destroy(omega);
}

// Partition, per-slab column remap, engines, transport.
int lanczos_hip_multi::build(int ngpu, const int *devices, int64_t n_, int64_t nnz_, const int64_t *rp,
                             const int64_t *ci, const double *va, int max_iter_, const lanczos_hip_options *opt_in,
                             int flags) {
    if (ngpu < 1 || ngpu > 64) return lz::fail("lanczos_hip_multi_create: ngpu=%d", ngpu);
    if (n_ <= 0 || nnz_ < 0 || !rp || (nnz_ > 0 && (!ci || !va)) || max_iter_ < 2)
        return lz::fail("lanczos_hip_multi_create: bad matrix or max_iter");
    if (n_ < ngpu) return lz::fail("lanczos_hip_multi_create: fewer rows (%lld) than slabs (%d)", (long long)n_, ngpu);
    if (rp[0] != 0 || rp[n_] != nnz_) return lz::fail("lanczos_hip_multi_create: row_head_indexes do not span nnz");
    P = ngpu;
    loopback = (flags & LANCZOS_HIP_MULTI_LOOPBACK) != 0;
    n = n_;
    nnz = nnz_;
    max_iter = max_iter_;
    lanczos_hip_options opt;
    memset(&opt, 0, sizeof opt);
    if (opt_in) memcpy(&opt, opt_in, std::min(sizeof opt, opt_in->struct_size));
    reorth = opt.reorth;
    if (reorth == LANCZOS_HIP_REORTH_MGS)
        return lz::fail("lanczos_hip_multi_create: the one-vector-at-a-time order (mgs) would take one reduction over "
                        "the slabs per basis vector; use full or full2");
    nvec = reorth == LANCZOS_HIP_REORTH_NONE ? 3 : max_iter;
    int ndev = 0;
    M_HIP(hipGetDeviceCount(&ndev));
    dev.resize((size_t)P);
    for (int p = 0; p < P; ++p) {
        dev[(size_t)p] = loopback ? (devices ? devices[0] : (opt.device >= 0 ? opt.device : 0)) : (devices ? devices[p] : p);
        if (dev[(size_t)p] < 0 || dev[(size_t)p] >= ndev)
            return lz::fail("lanczos_hip_multi_create: device %d of %d visible (slab %d); LANCZOS_HIP_NGPU asks for more "
                            "GPUs than this process sees", dev[(size_t)p], ndev, p);
    }
    bounds.resize((size_t)P + 1);
    for (int p = 0; p <= P; ++p) bounds[(size_t)p] = (int64_t)(((__int128)p * n) / P);

    // ---- how far each slab's columns reach beyond it: the halo form needs every remote column in
    // an adjacent slab (and few enough of them), on every slab
    halo_lo.assign((size_t)P, 0);
    halo_hi.assign((size_t)P, 0);
    bool halo_ok = true;
    for (int p = 0; p < P; ++p) {
        const int64_t r0 = bounds[(size_t)p], r1 = bounds[(size_t)p + 1];
        int64_t lo = r0, hi = r1 - 1;
        for (int64_t j = rp[r0]; j < rp[r1]; ++j) {
            if (ci[j] < 0 || ci[j] >= n) return lz::fail("lanczos_hip_multi_create: column index out of range [0, %lld)", (long long)n);
            lo = std::min(lo, ci[j]);
            hi = std::max(hi, ci[j]);
        }
        const int64_t hl = r0 - lo, hh = hi - (r1 - 1);
        halo_lo[(size_t)p] = hl;
        halo_hi[(size_t)p] = hh;
        if (hl && (p == 0 || hl > r0 - bounds[(size_t)p - 1])) halo_ok = false;
        if (hh && (p == P - 1 || hh > bounds[(size_t)p + 2] - r1)) halo_ok = false;
        if (hl + hh > (r1 - r0) / 2 + 1) halo_ok = false;
    }
    if (getenv("LANCZOS_HIP_MULTI_ALLGATHER") && atoi(getenv("LANCZOS_HIP_MULTI_ALLGATHER"))) halo_ok = false;
    halo = halo_ok;
    send_lo.assign((size_t)P, 0);
    send_hi.assign((size_t)P, 0);
    if (halo)
        for (int p = 0; p < P; ++p) {
            if (p > 0) send_lo[(size_t)p] = halo_hi[(size_t)p - 1];      // my head, for the slab before me
            if (p < P - 1) send_hi[(size_t)p] = halo_lo[(size_t)p + 1];  // my tail, for the slab after me
        }
    pad = 0;
    for (int p = 0; p < P; ++p) pad = std::max(pad, bounds[(size_t)p + 1] - bounds[(size_t)p]);

    // ---- streams, buffers, engines
    eng.assign((size_t)P, nullptr);
    stream.assign((size_t)P, nullptr);
    cstream.assign((size_t)P, nullptr);
    ev_ready.assign((size_t)P, nullptr);
    ev_done.assign((size_t)P, nullptr);
    basis.assign((size_t)P, nullptr);
    x_ext.assign((size_t)P, nullptr);
    red.assign((size_t)P, nullptr);
    ld.assign((size_t)P, 0);
    base.assign((size_t)P, 0);
    in_lo.assign((size_t)P, 0);
    in_hi.assign((size_t)P, 0);
    // Streams and events first, one slab after the other (with the loopback transport every slab shares slab 0's
    // stream); then every slab's column remap, upload and engine in a host thread of its own, so that the matrix
    // of a whole node (16 GB of CRS arrays for 512^3) does not go through one thread and one copy queue.
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        if (loopback && p > 0) {
            stream[(size_t)p] = stream[0];
        } else {
            M_HIP(hipStreamCreateWithFlags(&stream[(size_t)p], hipStreamNonBlocking));
        }
        if (!loopback) {
            M_HIP(hipStreamCreateWithFlags(&cstream[(size_t)p], hipStreamNonBlocking));
            M_HIP(hipEventCreateWithFlags(&ev_ready[(size_t)p], hipEventDisableTiming));
            M_HIP(hipEventCreateWithFlags(&ev_done[(size_t)p], hipEventDisableTiming));
        }
    }
    std::vector<char> can_split((size_t)P, 1);
    auto build_slab = [&](int p) -> int {
        M_TRY(use(p));
        const int64_t r0 = bounds[(size_t)p], r1 = bounds[(size_t)p + 1], nl = r1 - r0;
        const int64_t e0 = rp[r0], cnt = rp[r1] - rp[r0];
        M_HIP(hipMalloc(&red[(size_t)p], sizeof(double) * (size_t)(max_iter + 8)));
        M_HIP(hipMemsetAsync(red[(size_t)p], 0, sizeof(double) * (size_t)(max_iter + 8), stream[(size_t)p]));
        std::vector<int64_t> cols, head;
        try {
            cols.resize((size_t)std::max<int64_t>(cnt, 1));
            head.resize((size_t)nl + 1);
        } catch (const std::bad_alloc &) {
            return lz::fail("lanczos_hip_multi_create: out of host memory for slab %d", p);
        }
        for (int64_t i = 0; i <= nl; ++i) head[(size_t)i] = rp[r0 + i] - e0;
        lanczos_hip_options o = opt;
        o.struct_size = sizeof o;
        o.device = dev[(size_t)p];
        o.stream = stream[(size_t)p];
        o.n_global = n;
        o.row_begin = r0;
        o.x_ext = nullptr;
        o.basis = nullptr;
        int64_t first_in = 0, end_in = nl;
        if (halo) {
            // x = [own slab | zero padding | tail of the slab before (halo_lo) | head of the slab after (halo_hi)]
            const int64_t b = lanczos_hip_gather_base(nl), hl = halo_lo[(size_t)p], hh = halo_hi[(size_t)p];
            base[(size_t)p] = b;
            int64_t last_lo_row = -1, first_hi_row = nl;
            for (int64_t i = 0; i < nl; ++i)
                for (int64_t j = rp[r0 + i]; j < rp[r0 + i + 1]; ++j) {
                    const int64_t c = ci[j];
                    int64_t m;
                    if (c < r0) {
                        m = b + (c - (r0 - hl));
                        last_lo_row = i;
                    } else if (c >= r1) {
                        m = b + hl + (c - r1);
                        first_hi_row = std::min(first_hi_row, i);
                    } else {
                        m = c - r0;
                    }
                    cols[(size_t)(j - e0)] = m;
                }
            first_in = last_lo_row + 1;
            end_in = first_hi_row;
            // worth computing the boundary rows twice only when they are few
            if (first_in > end_in || first_in + (nl - end_in) > nl / 4) can_split[(size_t)p] = 0;
            o.n_ext = b + hl + hh;
            ld[(size_t)p] = round_up(o.n_ext, 2048);
            M_HIP(hipMalloc(&basis[(size_t)p], sizeof(double) * (size_t)ld[(size_t)p] * (size_t)nvec));
            o.basis = basis[(size_t)p];
            o.basis_ld = ld[(size_t)p];
        } else {
            // x_ext = the whole vector in slab order, every slab padded to the longest one
            for (int64_t j = 0; j < cnt; ++j) {
                const int64_t c = ci[e0 + j];
                const int owner = (int)(std::upper_bound(bounds.begin() + 1, bounds.end(), c) - (bounds.begin() + 1));
                cols[(size_t)j] = (int64_t)owner * pad + (c - bounds[(size_t)owner]);
            }
            o.n_ext = pad * P;
            M_HIP(hipMalloc(&x_ext[(size_t)p], sizeof(double) * (size_t)o.n_ext));
            M_HIP(hipMemsetAsync(x_ext[(size_t)p], 0, sizeof(double) * (size_t)o.n_ext, stream[(size_t)p]));
            o.x_ext = x_ext[(size_t)p];
        }
        in_lo[(size_t)p] = first_in;
        in_hi[(size_t)p] = end_in;
        M_TRY(lz::Engine::create(&eng[(size_t)p], nl, cnt, head.data(), cols.data(), va + e0, max_iter, &o));
        eng[(size_t)p]->affinity_id = p;
        if (halo && !eng[(size_t)p]->can_split_product()) can_split[(size_t)p] = 0;
        return 0;
    };
    const auto t_build = std::chrono::steady_clock::now();
    // LANCZOS_HIP_MULTI_BUILD_THREADS: 0 one slab after the other, 1 a thread per slab (the default on several
    // devices; with the loopback transport the slabs share a device and a stream and are built in turn)
    const char *thr_env = getenv("LANCZOS_HIP_MULTI_BUILD_THREADS");
    const bool threaded = P > 1 && (thr_env ? atoi(thr_env) != 0 : !loopback);
    if (threaded) {
        std::vector<int> rcs((size_t)P, 0);
        std::vector<std::string> msgs((size_t)P);
        std::vector<std::thread> workers;
        try {
            for (int p = 0; p < P; ++p)
                workers.emplace_back([&, p] {
                    rcs[(size_t)p] = lz::guarded("lanczos_hip_multi_create", [&]() -> int { return build_slab(p); });
                    if (rcs[(size_t)p]) msgs[(size_t)p] = lz::last_error();  // the message is the failing thread's own
                });
        } catch (const std::exception &) {  // no more threads to be had: the slabs not started are built here
            for (size_t p = workers.size(); p < (size_t)P; ++p) {
                rcs[p] = build_slab((int)p);
                if (rcs[p]) msgs[p] = lz::last_error();
            }
        }
        for (auto &w : workers) w.join();
        for (int p = 0; p < P; ++p)
            if (rcs[(size_t)p]) return lz::fail("%s", msgs[(size_t)p].c_str());
    } else {
        for (int p = 0; p < P; ++p) M_TRY(build_slab(p));
    }
    build_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
    build_threads = threaded ? P : 1;
    bool all_split = halo;
    for (int p = 0; p < P; ++p) all_split = all_split && can_split[(size_t)p];
    overlap = all_split && P > 1 && !(getenv("LANCZOS_HIP_MULTI_NO_OVERLAP") && atoi(getenv("LANCZOS_HIP_MULTI_NO_OVERLAP")));
    if (loopback) {
        M_TRY(use(0));
        M_HIP(hipMalloc(&d_red_ptrs, sizeof(double *) * (size_t)P));
        M_HIP(hipMemcpy(d_red_ptrs, red.data(), sizeof(double *) * (size_t)P, hipMemcpyHostToDevice));
    } else {
        if (!rccl()) return lz::fail("lanczos_hip_multi_create: librccl.so.1 could not be opened (%s)", dlerror());
        // RCCL prints a version banner on stdout when a process makes its first communicator; stdout
        // carries the solver's own lines (src/lanczos.c:134-138), so the banner goes to stderr
        fflush(stdout);
        const int keep_out = dup(1);
        if (keep_out >= 0) (void)dup2(2, 1);
        comm.assign((size_t)P, nullptr);
        ncclResult_t st = rccl()->CommInitAll(comm.data(), P, dev.data());
        if (st == ncclSuccess && halo && P > 1) {
            xcomm.assign((size_t)P, nullptr);
            st = rccl()->CommInitAll(xcomm.data(), P, dev.data());
        }
        fflush(stdout);
        if (keep_out >= 0) {
            (void)dup2(keep_out, 1);
            close(keep_out);
        }
        if (st != ncclSuccess)
            return lz::fail("RCCL API failed: ncclCommInitAll over %d devices: %s (%d)", P, rccl()->GetErrorString(st), (int)st);
    }
    if (reorth == LANCZOS_HIP_REORTH_SELECTIVE) {
        omega = lanczos_hip_omega_create(max_iter);
        if (!omega) return lz::fail("lanczos_hip_multi_create: no memory for the orthogonality tracker");
        M_TRY(use(0));
        M_HIP(hipHostMalloc((void **)&h_sel, sizeof(double) * 4, hipHostMallocDefault));
    }
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_HIP(hipStreamSynchronize(stream[(size_t)p]));
    }
    return 0;
}

// v0: the global start vector; normalised over all slabs (src/lanczos.c:95-98)
int lanczos_hip_multi::set_start(const double *v0) {
    if (!v0) return lz::fail("lanczos_hip_multi_set_start: NULL vector");
    double norm2 = 0.0;
    for (int64_t i = 0; i < n; ++i) norm2 += v0[i] * v0[i];
    if (!(norm2 > 0.0)) return lz::fail("lanczos_hip_multi_set_start: zero start vector");
    const double scale = 1.0 / sqrt(norm2);
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_TRY(eng[(size_t)p]->set_start(v0 + bounds[(size_t)p], false, scale));
    }
    k = 0;
    closed_k = 0;
    reorth_steps = 0;
    last_beta = 0.0;
    if (omega) lanczos_hip_omega_reset(omega);
    return 0;
}

// red[p][0..count) <- the sum over slabs, in place, on every slab's compute stream
int lanczos_hip_multi::all_reduce(int count) {
    if (loopback) {
        if (P == 1) return 0;
        M_TRY(use(0));
        hipLaunchKernelGGL(k_sum_slabs, dim3((count + 255) / 256), dim3(256), 0, stream[0], (double *const *)d_red_ptrs, P, count);
        M_HIP(hipGetLastError());
        return 0;
    }
    M_RCCL(rccl()->GroupStart());
    for (int p = 0; p < P; ++p) {
        const ncclResult_t st = rccl()->AllReduce(red[(size_t)p], red[(size_t)p], (size_t)count, ncclDouble, ncclSum,
                                                  comm[(size_t)p], stream[(size_t)p]);
        if (st != ncclSuccess) {
            (void)rccl()->GroupEnd();
            return lz::fail("RCCL API failed: ncclAllReduce: %s", rccl()->GetErrorString(st));
        }
    }
    M_RCCL(rccl()->GroupEnd());
    return 0;
}

// The exchange of vector k.  Halo form: slab p's head goes to the tail region of slab p-1's vector, its
// tail to slab p+1's; the transfers are ordered behind everything enqueued so far on the compute
// streams and run on the exchange streams.  All-gather form: complete on return (compute streams).
int lanczos_hip_multi::exchange_begin() {
    if (!halo) {
        for (int p = 0; p < P; ++p) {
            M_TRY(use(p));
            const int64_t nl = bounds[(size_t)p + 1] - bounds[(size_t)p];
            if (loopback) {
                for (int q = 0; q < P; ++q)
                    M_HIP(hipMemcpyAsync(x_ext[(size_t)q] + (size_t)p * (size_t)pad, eng[(size_t)p]->vec(k), sizeof(double) * (size_t)nl,
                                         hipMemcpyDeviceToDevice, stream[0]));
            } else {
                M_HIP(hipMemcpyAsync(x_ext[(size_t)p] + (size_t)p * (size_t)pad, eng[(size_t)p]->vec(k), sizeof(double) * (size_t)nl,
                                     hipMemcpyDeviceToDevice, stream[(size_t)p]));
            }
        }
        if (!loopback && P > 1) {
            M_RCCL(rccl()->GroupStart());
            for (int p = 0; p < P; ++p) {
                const ncclResult_t st = rccl()->AllGather(x_ext[(size_t)p] + (size_t)p * (size_t)pad, x_ext[(size_t)p], (size_t)pad, ncclDouble,
                                                          comm[(size_t)p], stream[(size_t)p]);
                if (st != ncclSuccess) {
                    (void)rccl()->GroupEnd();
                    return lz::fail("RCCL API failed: ncclAllGather: %s", rccl()->GetErrorString(st));
                }
            }
            M_RCCL(rccl()->GroupEnd());
        }
        return 0;
    }
    if (P == 1) return 0;
    if (loopback) {
        for (int p = 0; p < P; ++p) {
            const int64_t nl = bounds[(size_t)p + 1] - bounds[(size_t)p];
            if (send_lo[(size_t)p])  // my head -> the "after" region of slab p-1
                M_HIP(hipMemcpyAsync(vec(p - 1, k) + base[(size_t)p - 1] + halo_lo[(size_t)p - 1], vec(p, k),
                                     sizeof(double) * (size_t)send_lo[(size_t)p], hipMemcpyDeviceToDevice, stream[0]));
            if (send_hi[(size_t)p])  // my tail -> the "before" region of slab p+1
                M_HIP(hipMemcpyAsync(vec(p + 1, k) + base[(size_t)p + 1], vec(p, k) + (nl - send_hi[(size_t)p]),
                                     sizeof(double) * (size_t)send_hi[(size_t)p], hipMemcpyDeviceToDevice, stream[0]));
        }
        return 0;
    }
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_HIP(hipEventRecord(ev_ready[(size_t)p], stream[(size_t)p]));
        M_HIP(hipStreamWaitEvent(cstream[(size_t)p], ev_ready[(size_t)p], 0));
    }
    M_RCCL(rccl()->GroupStart());
    ncclResult_t st = ncclSuccess;
    for (int p = 0; p < P && st == ncclSuccess; ++p) {
        const int64_t nl = bounds[(size_t)p + 1] - bounds[(size_t)p];
        double *x = vec(p, k);
        if (st == ncclSuccess && send_lo[(size_t)p])
            st = rccl()->Send(x, (size_t)send_lo[(size_t)p], ncclDouble, p - 1, xcomm[(size_t)p], cstream[(size_t)p]);
        if (st == ncclSuccess && halo_hi[(size_t)p])
            st = rccl()->Recv(x + base[(size_t)p] + halo_lo[(size_t)p], (size_t)halo_hi[(size_t)p], ncclDouble, p + 1,
                              xcomm[(size_t)p], cstream[(size_t)p]);
        if (st == ncclSuccess && send_hi[(size_t)p])
            st = rccl()->Send(x + (nl - send_hi[(size_t)p]), (size_t)send_hi[(size_t)p], ncclDouble, p + 1, xcomm[(size_t)p],
                              cstream[(size_t)p]);
        if (st == ncclSuccess && halo_lo[(size_t)p])
            st = rccl()->Recv(x + base[(size_t)p], (size_t)halo_lo[(size_t)p], ncclDouble, p - 1, xcomm[(size_t)p],
                              cstream[(size_t)p]);
    }
    if (st != ncclSuccess) {
        (void)rccl()->GroupEnd();
        return lz::fail("RCCL API failed: halo exchange: %s", rccl()->GetErrorString(st));
    }
    M_RCCL(rccl()->GroupEnd());
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_HIP(hipEventRecord(ev_done[(size_t)p], cstream[(size_t)p]));
    }
    return 0;
}

// what is enqueued on the compute streams from here on sees the received entries
int lanczos_hip_multi::exchange_end() {
    if (!halo || loopback || P == 1) return 0;
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_HIP(hipStreamWaitEvent(stream[(size_t)p], ev_done[(size_t)p], 0));
    }
    return 0;
}

// three-vector mode: beta of the last iteration is the norm of the vector it left behind
int lanczos_hip_multi::close_lazy() {
    if (reorth != LANCZOS_HIP_REORTH_NONE || k == 0 || closed_k == k) return 0;
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_TRY(eng[(size_t)p]->phase_lazy_norm(red[(size_t)p]));
    }
    M_TRY(all_reduce(1));
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_TRY(eng[(size_t)p]->phase_lazy_close(red[(size_t)p]));
    }
    closed_k = k;
    return 0;
}

// One iteration (src/lanczos.c:109-190, split at the reductions over slabs).
int lanczos_hip_multi::step(bool close_beta) {
#define EACH(expr)                    \
    for (int p = 0; p < P; ++p) {     \
        M_TRY(use(p));                \
        lz::Engine *e = eng[(size_t)p]; \
        double *r = red[(size_t)p];   \
        (void)r;                      \
        M_TRY(expr);                  \
    }
    M_TRY(exchange_begin());
    if (overlap) {
        EACH(e->phase_spmv_begin(in_lo[(size_t)p], in_hi[(size_t)p]));
        M_TRY(exchange_end());
        EACH(e->phase_spmv_end(r));
    } else {
        M_TRY(exchange_end());
        EACH(e->phase_spmv(r));
    }
    if (reorth == LANCZOS_HIP_REORTH_NONE) {
        M_TRY(all_reduce(2));  // u_k . A u_k and |u_k|^2 together
        EACH(e->phase_recur_lazy(r));
        k += 1;
        if (close_beta) M_TRY(close_lazy());
        return 0;
    }
    M_TRY(all_reduce(1));  // alpha
    if (omega) {
        M_TRY(use(0));
        M_HIP(hipMemcpyAsync(h_sel, red[0], sizeof(double), hipMemcpyDeviceToHost, stream[0]));
    }
    int n_red = 0, is_norm = 0;
    EACH(e->phase_recur(r, &n_red, &is_norm));
    M_TRY(all_reduce(n_red));  // reorthogonalisation coefficients and w.w together, or beta^2
    if (omega) {
        // the one host round trip per step: alpha_k and |w|^2 for the decision rule (the same on every slab)
        M_TRY(use(0));
        M_HIP(hipMemcpyAsync(h_sel + 1, red[0], sizeof(double), hipMemcpyDeviceToHost, stream[0]));
        M_HIP(hipStreamSynchronize(stream[0]));
        double norm2 = h_sel[1];
        if (k > 0) lanczos_hip_omega_set_beta(omega, k - 1, last_beta);
        const int need = lanczos_hip_omega_step(omega, k, h_sel[0], sqrt(norm2), nullptr);
        if (need < 0) return lz::fail("lanczos_hip_multi: orthogonality tracker out of step at iteration %d", k);
        if (need) {
            reorth_steps += 1;
            int n2 = 0;
            EACH(e->phase_dots(r, &n2));
            M_TRY(all_reduce(n2));
            EACH(e->phase_update(r));
            M_TRY(all_reduce(1));  // beta^2
            M_TRY(use(0));
            M_HIP(hipMemcpyAsync(h_sel + 1, red[0], sizeof(double), hipMemcpyDeviceToHost, stream[0]));
            M_HIP(hipStreamSynchronize(stream[0]));
            norm2 = h_sel[1];
        }
        last_beta = sqrt(norm2);
    } else if (!is_norm) {
        reorth_steps += 1;
        EACH(e->phase_update(r));  // red[0] = beta^2 over all slabs (from the reduced w.w)
        if (reorth == LANCZOS_HIP_REORTH_FULL2) {
            int n2 = 0;
            EACH(e->phase_dots(r, &n2));
            M_TRY(all_reduce(n2));
            EACH(e->phase_update(r));
            M_TRY(all_reduce(1));  // beta^2 after the second pass
        }
    }
    EACH(e->phase_scale(r));
    k += 1;
    return 0;
#undef EACH
}

int lanczos_hip_multi::run(int steps, double *alpha_out, double *beta_out) {
    if (steps < 0) return lz::fail("lanczos_hip_multi_run: steps=%d", steps);
    const int k0 = k;
    for (int s = 0; s < steps; ++s) M_TRY(step(false));
    M_TRY(close_lazy());
    for (int p = 0; p < P; ++p) {
        M_TRY(use(p));
        M_HIP(hipStreamSynchronize(stream[(size_t)p]));
        eng[(size_t)p]->harvest_profile();
    }
    M_TRY(use(0));
    if (alpha_out && steps)
        M_HIP(hipMemcpy(alpha_out, eng[0]->d_alpha + k0, sizeof(double) * (size_t)steps, hipMemcpyDeviceToHost));
    if (beta_out && steps)
        M_HIP(hipMemcpy(beta_out, eng[0]->d_beta + k0, sizeof(double) * (size_t)steps, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------
extern "C" {

int lanczos_hip_multi_create(lanczos_hip_multi **out, int ngpu, const int *devices, int64_t n, int64_t nnz,
                             const int64_t *row_head_indexes, const int64_t *column_index, const double *values,
                             int max_iter, const lanczos_hip_options *opt, int flags) {
    return lz::guarded("lanczos_hip_multi_create", [&]() -> int {
    if (!out) return lz::fail("lanczos_hip_multi_create: NULL out");
    *out = nullptr;
    lanczos_hip_multi *m = new (std::nothrow) lanczos_hip_multi();
    if (!m) return lz::fail("lanczos_hip_multi_create: out of host memory");
    DeviceRestore keep;
    const int rc = m->build(ngpu, devices, n, nnz, row_head_indexes, column_index, values, max_iter, opt, flags);
    if (rc) {
        delete m;
        return rc;
    }
    *out = m;
    return 0;
    });
}

void lanczos_hip_multi_destroy(lanczos_hip_multi *m) {
    DeviceRestore keep;
    delete m;
}

int lanczos_hip_multi_set_start(lanczos_hip_multi *m, const double *v0_host) {
    return lz::guarded("lanczos_hip_multi_set_start", [&]() -> int {
    if (!m) return lz::fail("lanczos_hip_multi_set_start: NULL handle");
    DeviceRestore keep;
    return m->set_start(v0_host);
    });
}

int lanczos_hip_multi_set_start_seed(lanczos_hip_multi *m, uint64_t seed) {
    return lz::guarded("lanczos_hip_multi_set_start_seed", [&]() -> int {
    if (!m) return lz::fail("lanczos_hip_multi_set_start_seed: NULL handle");
    std::vector<double> v;
    try {
        v.resize((size_t)m->n);
    } catch (const std::bad_alloc &) {
        return lz::fail("lanczos_hip_multi_set_start_seed: out of host memory");
    }
    lanczos_hip_gaussian_fill(seed, 0, m->n, v.data());
    DeviceRestore keep;
    return m->set_start(v.data());
    });
}

int lanczos_hip_multi_run(lanczos_hip_multi *m, int steps, double *alpha_out, double *beta_out) {
    return lz::guarded("lanczos_hip_multi_run", [&]() -> int {
    if (!m) return lz::fail("lanczos_hip_multi_run: NULL handle");
    DeviceRestore keep;
    return m->run(steps, alpha_out, beta_out);
    });
}

int lanczos_hip_multi_step(lanczos_hip_multi *m) {
    return lz::guarded("lanczos_hip_multi_step", [&]() -> int {
    if (!m) return lz::fail("lanczos_hip_multi_step: NULL handle");
    DeviceRestore keep;
    return m->step(true);
    });
}

lanczos_hip_engine *lanczos_hip_multi_engine(lanczos_hip_multi *m, int slab) {
    if (!m || slab < 0 || slab >= m->P) return nullptr;
    return reinterpret_cast<lanczos_hip_engine *>(m->eng[(size_t)slab]);
}

int lanczos_hip_multi_build_stats(const lanczos_hip_multi *m, double *seconds, int *host_threads) {
    if (!m) return lz::fail("lanczos_hip_multi_build_stats: NULL handle");
    if (seconds) *seconds = m->build_sec;
    if (host_threads) *host_threads = m->build_threads;
    return 0;
}

int lanczos_hip_multi_info(const lanczos_hip_multi *m, int *slabs, int *halo_exchange, int *overlapped, int *iterations,
                           int *reorth_steps) {
    if (!m) return lz::fail("lanczos_hip_multi_info: NULL handle");
    if (slabs) *slabs = m->P;
    if (halo_exchange) *halo_exchange = m->halo;
    if (overlapped) *overlapped = m->overlap;
    if (iterations) *iterations = m->k;
    if (reorth_steps) *reorth_steps = m->reorth_steps;
    return 0;
}

}  // extern "C"
