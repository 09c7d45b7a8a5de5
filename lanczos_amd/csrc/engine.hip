// Host side of the HIP Lanczos engine: device memory, kernel launches, the
// iteration phases and the extern "C" entry points declared in
// include/lanczos_hip.h.  One engine = one GPU = one stream; nothing here
// synchronises with the host inside an iteration.
#include "engine.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "balance.hip.h"
#include "kernels.hip.h"

namespace lz {

static thread_local char g_err[512] = "";

int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    printf("%s\n", g_err);  // stdout, as the reference's CHECK_*_GOTO macros (src/lanczos_cuda.h:76-124)
    return EXIT_FAILURE;
}

const char *last_error() { return g_err; }

thread_local int tl_current_slab = -1;

int check_affinity(const Engine *e, const char *where) {
    if (e->affinity_id < 0 || tl_current_slab == e->affinity_id) return 0;
    const char *s = getenv("LANCZOS_HIP_CHECK_AFFINITY");  // read per call: only slabs under the wrong device get here
    if (!s || atoi(s) == 0) return 0;
    return fail("%s: slab %d's stream used while slab %d is current (its device would not be the current one on a "
                "node with one GPU per slab)", where, e->affinity_id, tl_current_slab);
}

#define HIP_TRY(call)                                                                         \
    do {                                                                                      \
        hipError_t st__ = (call);                                                             \
        if (st__ != hipSuccess)                                                               \
            return fail("HIP API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,    \
                        hipGetErrorString(st__), (int)st__);                                  \
    } while (0)

static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------
struct ProfScope {
    Engine *e;
    int cls;
    bool timed;
    ProfScope(Engine *eng, int c, double bytes) : e(eng), cls(c), timed(false) {
        if (!e->profile) return;
        e->prof_bytes[cls] += bytes;
        e->prof_launches[cls] += 1;
        // two events per launch cost a run of short kernels several per cent: only the selected classes are timed
        timed = ((e->prof_mask >> cls) & 1u) != 0;
        if (!timed) return;
        hipEvent_t a = e->take_event();
        (void)hipEventRecord(a, e->stream);
        e->prof_pending.push_back({cls, a, nullptr});
    }
    ~ProfScope() {
        if (!timed) return;
        hipEvent_t b = e->take_event();
        (void)hipEventRecord(b, e->stream);
        e->prof_pending.back().stop = b;
    }
};

hipEvent_t Engine::take_event() {
    if (!event_pool.empty()) {
        hipEvent_t ev = event_pool.back();
        event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
}

void Engine::harvest_profile() {
    for (auto &p : prof_pending) {
        float ms = 0.f;
        if (p.start && p.stop && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess)
            prof_ms[p.cls] += ms;
        if (p.start) event_pool.push_back(p.start);
        if (p.stop) event_pool.push_back(p.stop);
    }
    prof_pending.clear();
}

// The events of launches that have finished, oldest first; stops at the first launch still running.
// For a host loop that runs alongside the stream: the list of live events stays as short as the
// number of launches in flight.
void Engine::harvest_completed() {
    size_t done = 0;
    for (; done < prof_pending.size(); ++done) {
        auto &p = prof_pending[done];
        if (!p.start || !p.stop || hipEventQuery(p.stop) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) prof_ms[p.cls] += ms;
        event_pool.push_back(p.start);
        event_pool.push_back(p.stop);
    }
    (void)hipGetLastError();  // hipErrorNotReady from the query is not an error
    prof_pending.erase(prof_pending.begin(), prof_pending.begin() + (long)done);
}

// ---------------------------------------------------------------------------
// creation / destruction
// ---------------------------------------------------------------------------
static int pick_group(double mean_nnz) {
    int g = 2;
    while (g < 64 && g < mean_nnz) g <<= 1;
    return g;
}

static int read_options(lanczos_hip_options *opt, const lanczos_hip_options *opt_in) {
    memset(opt, 0, sizeof(*opt));
    opt->device = -1;
    if (opt_in) memcpy(opt, opt_in, std::min(sizeof(*opt), opt_in->struct_size));
    if (opt->reorth < 0 || opt->reorth > LANCZOS_HIP_REORTH_SELECTIVE)
        return fail("lanczos_hip_create: bad reorth mode %d", opt->reorth);
    if (opt->spmv_kernel < 0 || opt->spmv_kernel > LANCZOS_HIP_SPMV_MERGE)
        return fail("lanczos_hip_create: bad spmv_kernel %d", opt->spmv_kernel);
    return 0;
}

int Engine::create(Engine **out, int64_t n, int64_t nnz, const int64_t *rp, const int64_t *ci,
                   const double *va, int max_iter, const lanczos_hip_options *opt_in) {
    *out = nullptr;
    lanczos_hip_options opt;
    if (int rc = read_options(&opt, opt_in)) return rc;
    if (n <= 0 || nnz < 0 || !rp || (nnz > 0 && (!ci || !va)))
        return fail("lanczos_hip_create: NULL or empty matrix (n=%lld, nnz=%lld)", (long long)n, (long long)nnz);
    if (n > (int64_t)INT32_MAX)
        return fail("Matrix dimension too large for HIP CRS (dim=%lld)", (long long)n);
    if (max_iter < 2) return fail("max_iter must be >= 2");
    const bool dist = opt.x_ext != nullptr || opt.basis != nullptr;
    const int64_t n_ext = dist ? opt.n_ext : n;
    if (dist && (n_ext <= 0 || n_ext > (int64_t)INT32_MAX))
        return fail("lanczos_hip_create: bad n_ext=%lld", (long long)n_ext);
    if (opt.basis) {
        const int64_t base = lanczos_hip_gather_base(n);
        if (opt.x_ext) return fail("lanczos_hip_create: options.basis and options.x_ext exclude each other");
        if (opt.basis_ld < base || opt.basis_ld < n_ext || (opt.basis_ld & 1) || ((uintptr_t)opt.basis & 15))
            return fail("lanczos_hip_create: basis_ld=%lld must be even and >= max(%lld, n_ext=%lld), basis 16-byte aligned",
                        (long long)opt.basis_ld, (long long)base, (long long)n_ext);
        // a column index may address the slab [0, n) or the tail [base, n_ext), never the zero padding between
        int gap = 0;
        for (int64_t j = 0; j < nnz; j++) gap |= (ci[j] >= n && ci[j] < base);
        if (gap) return fail("lanczos_hip_create: column index inside the padding [%lld, %lld)", (long long)n, (long long)base);
    }
    // The kernels trust these arrays; check them here, once, on the host.
    if (rp[0] != 0 || rp[n] != nnz)
        return fail("lanczos_hip_create: row_head_indexes[0]=%lld, [n]=%lld, nnz=%lld", (long long)rp[0],
                    (long long)rp[n], (long long)nnz);
    int bad = 0;
    for (int64_t i = 0; i < n; i++) bad |= (rp[i + 1] < rp[i]);
    if (bad) return fail("lanczos_hip_create: row_head_indexes not monotone");
    for (int64_t j = 0; j < nnz; j++) bad |= (ci[j] < 0 || ci[j] >= n_ext);
    if (bad) return fail("lanczos_hip_create: column index out of range [0, %lld)", (long long)n_ext);

    Engine *e = new Engine();
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto lap = [&](int i) {
        const auto t1 = clk::now();
        e->create_sec[i] = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    };
    int rc = e->setup_device(opt, n, n_ext, nnz, max_iter);
    lap(0);
    if (rc == 0) rc = e->upload_crs(rp, ci, va);
    lap(1);
    if (rc == 0) rc = e->choose_spmv(opt.spmv_kernel);
    lap(2);
    if (rc == 0) rc = e->alloc_state();
    lap(3);
    if (rc != 0) {
        delete e;
        return rc;
    }
    *out = e;
    return 0;
}

int Engine::setup_device(const lanczos_hip_options &opt, int64_t n_, int64_t n_ext_, int64_t nnz_, int max_iter_) {
    n = n_;
    n_ext = n_ext_;
    nnz = nnz_;
    max_iter = max_iter_;
    reorth = opt.reorth;
    profile = opt.profile != 0;
    x_ext = opt.x_ext;
    slab = opt.x_ext != nullptr || opt.basis != nullptr;
    if (opt.basis) {
        d_V = opt.basis;
        ldv = opt.basis_ld;
        own_V = false;
    }
    n_global = opt.n_global ? opt.n_global : n;
    row_begin = opt.row_begin;
    if (opt.device >= 0) {
        HIP_TRY(hipSetDevice(opt.device));
        device = opt.device;
    } else {
        HIP_TRY(hipGetDevice(&device));
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("lanczos_hip: device %d is %s; this library holds gfx950 code only", device, prop.gcnArchName);
    cus = prop.multiProcessorCount;
    max_blocks = cus * 8;
    max_blocks -= max_blocks % 8;
    if (opt.stream) {
        stream = (hipStream_t)opt.stream;
        own_stream = false;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        own_stream = true;
    }
    rp64 = nnz >= ((int64_t)1 << 31);
    // a few entries of padding: the rows SpMV loads aligned granules of four entries
    HIP_TRY(hipMalloc(&d_val, sizeof(double) * (std::max<int64_t>(nnz, 1) + 8)));
    HIP_TRY(hipMalloc(&d_col, sizeof(int32_t) * (std::max<int64_t>(nnz, 1) + 8)));
    HIP_TRY(hipMalloc(&d_rp, (rp64 ? 8 : 4) * (n + 1)));
    return 0;
}

// The caller's array page-locked for the duration of the upload, as the reference does around its
// copies (src/lanczos_cuda.c:144, 168): the copy engine then reads it directly.  When the
// registration is refused the copy goes through the runtime's pageable path instead.
struct HostPin {
    void *ptr = nullptr;
    hipError_t why = hipSuccess;  // what the registration answered when it was refused
    HostPin(const void *p, size_t bytes, bool enable) {
        if (!enable || !bytes) return;
        why = hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault);
        if (why == hipSuccess)
            ptr = const_cast<void *>(p);
        else
            (void)hipGetLastError();
    }
    ~HostPin() {
        if (ptr) (void)hipHostUnregister(ptr);
    }
};

// Matrix upload: values as they are, columns and (when they fit 32 bits) row offsets narrowed on
// the device from a staging buffer.
int Engine::upload_crs(const int64_t *rp, const int64_t *ci, const double *va) {
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    const char *pin_env = getenv("LANCZOS_HIP_PIN_UPLOAD");
    const bool pin = !(pin_env && *pin_env && atoi(pin_env) == 0);
    HostPin pin_va(va, sizeof(double) * (size_t)nnz, pin), pin_ci(ci, sizeof(int64_t) * (size_t)nnz, pin),
        pin_rp(rp, sizeof(int64_t) * (size_t)(n + 1), pin);
    upload_pinned = pin_va.ptr != nullptr || nnz == 0;
    snprintf(upload_pin_note, sizeof(upload_pin_note), "%s",
             !pin ? "off" : (pin_va.why == hipSuccess ? "ok" : hipGetErrorName(pin_va.why)));
    const int64_t chunk = (int64_t)1 << 26;  // 512 MiB of int64 per piece
    int64_t *stage = nullptr;
    HIP_TRY(hipMalloc(&stage, sizeof(int64_t) * std::min(chunk, std::max(nnz, n + 1))));
    auto narrow = [&](const int64_t *src, int32_t *dst, int64_t count) -> hipError_t {
        for (int64_t off = 0; off < count; off += chunk) {
            const int64_t cnt = std::min(chunk, count - off);
            hipError_t st = hipMemcpyAsync(stage, src + off, sizeof(int64_t) * cnt, hipMemcpyHostToDevice, stream);
            if (st != hipSuccess) return st;
            const int blocks = (int)std::min<int64_t>(max_blocks, (cnt + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(k_narrow_i64, dim3(blocks), dim3(kBlock), 0, stream, stage, dst + off, cnt);
        }
        return hipGetLastError();
    };
    hipError_t st = hipSuccess;
    if (nnz > 0) {
        st = hipMemcpyAsync(d_val, va, sizeof(double) * nnz, hipMemcpyHostToDevice, stream);
        if (st == hipSuccess) st = narrow(ci, d_col, nnz);
    }
    if (st == hipSuccess) {
        if (rp64)
            st = hipMemcpyAsync(d_rp, rp, 8 * (n + 1), hipMemcpyHostToDevice, stream);
        else
            st = narrow(rp, (int32_t *)d_rp, n + 1);
    }
    if (st == hipSuccess) st = hipStreamSynchronize(stream);
    (void)hipFree(stage);
    if (st != hipSuccess) return fail("lanczos_hip: matrix upload failed: %s", hipGetErrorString(st));
    upload_bytes = 16.0 * (double)nnz + 8.0 * (double)(n + 1);
    upload_sec = std::chrono::duration<double>(clk::now() - t0).count();
    return 0;
}

// Pick the SpMV form from the matrix as it sits on the device.
int Engine::choose_spmv(int requested) {
    const double mean = (double)nnz / (double)n;
    spmv_kind = requested;
    // largest nnz span of a 64-row group, and how often neighbouring rows hold
    // neighbouring columns (sampled): what the row-wise gather form needs / likes
    unsigned long long *d_stats = nullptr, stats[5] = {0, 0, 0, 0, 0};
    HIP_TRY(hipMalloc(&d_stats, sizeof(stats)));
    HIP_TRY(hipMemsetAsync(d_stats, 0, sizeof(stats), stream));
    const int64_t stride = std::max<int64_t>(1, (n - 1) / 4096);
    const int blocks = (int)std::min<int64_t>(max_blocks, ((n + 63) / 64 + kBlock - 1) / kBlock);
    if (rp64)
        hipLaunchKernelGGL((k_row_stats<int64_t>), dim3(blocks), dim3(kBlock), 0, stream, n, (const int64_t *)d_rp,
                           d_col, stride, d_stats);
    else
        hipLaunchKernelGGL((k_row_stats<int32_t>), dim3(blocks), dim3(kBlock), 0, stream, n, (const int32_t *)d_rp,
                           d_col, stride, d_stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipFree(d_stats));
    const int64_t max_span = (int64_t)stats[0];
    if (spmv_kind == LANCZOS_HIP_SPMV_ROWS && max_span > kRowsMaxSpan)
        return fail("lanczos_hip_create: the rows SpMV needs every 64-row group to hold <= %d entries (max %lld)",
                    kRowsMaxSpan, (long long)max_span);
    // wildly uneven rows: the fullest 64-row group holds 16 times the average
    const bool uneven = (double)max_span > 16.0 * 64.0 * std::max(mean, 1.0);
    bool bits_given = false;
    if (const char *s = getenv("LANCZOS_HIP_PANEL_BITS"))
        if (atoi(s) >= 6 && atoi(s) <= 24) {
            panel_bits = atoi(s);
            bits_given = true;
        }
    // narrow panels for power-law graphs: their tiles are small and the workgroups do not sweep in step,
    // so what a panel's slice of x has to fit is a CU's L1, not the L2 (R-MAT, 2^24 / 2^26 rows, ms per
    // SpMV: 2^14 3.52 / 15.3, 2^12 3.13 / 14.0, 2^10 3.04 / 13.7, 2^8 3.02 / 13.5)
    if (!bits_given && uneven) panel_bits = 10;
    const bool no_dict = getenv("LANCZOS_HIP_NO_DICT") && atoi(getenv("LANCZOS_HIP_NO_DICT")) != 0;
    if (spmv_kind == LANCZOS_HIP_SPMV_AUTO && !no_dict && stats[1] > 0 && 2 * stats[2] > stats[1]) {
        // neighbouring rows hold neighbouring columns: rows are likely to repeat as patterns
        if (int rc = build_dict()) return rc;
        // a dictionary in which a pattern serves fewer than four rows on average (values that do not repeat, up to
        // 65535 rows) is the matrix over again: left to the row-wise forms
        if (dict_patterns > 0 && (int64_t)dict_patterns * 4 > n) free_dict();
        if (dict_patterns > 0) spmv_kind = LANCZOS_HIP_SPMV_DICT;
    }
    if (spmv_kind == LANCZOS_HIP_SPMV_DICT && dict_patterns == 0) {
        if (int rc = build_dict()) return rc;
        if (dict_patterns == 0)
            return fail("lanczos_hip_create: the dictionary SpMV needs rows of at most 32 entries that repeat as at "
                        "most 65535 distinct (column offset, value) patterns");
    }
    if (spmv_kind == LANCZOS_HIP_SPMV_AUTO) {
        spmv_kind = LANCZOS_HIP_SPMV_PIPE;
        if (max_span <= kRowsMaxSpan && stats[1] > 0 && 2 * stats[2] > stats[1]) {
            spmv_kind = LANCZOS_HIP_SPMV_ROWS;
        } else {
            // rows that reach across more of x than the L2s hold (the sampled rows' mean column
            // extent against 16 MiB of doubles, half the L2s): gathers would each fetch a line from memory
            // With uneven rows the share-of-entries form (below) is the faster one while x is a few MiB; the
            // tiled form takes over from about 6 MiB of x (measured on R-MAT graphs, merge against tiled:
            // 0.079 / 0.123 ms at n = 2^18, 0.143 / 0.137 at 2^19, 0.337 / 0.250 at 2^20, 1.43 / 0.94 at 2^22,
            // 8.4 / 3.8 at 2^24, 45 / 17.7 at 2^26), provided the sampled rows reach across 2 MiB of it and more
            // (half the rows of such a graph hold the diagonal alone and count with extent 0).
            const double extent = stats[3] ? (double)stats[4] / (double)stats[3] : 0.0;
            const bool tiled = uneven ? extent * 8.0 > 2.0 * 1048576.0 && 8.0 * (double)n_ext > 6.0 * 1048576.0
                                      : extent * 8.0 > 16.0 * 1048576.0;
            if (tiled && mean >= 2.0) {
                // The regrouping needs about 20 bytes per entry of temporary device memory beside the 12-14
                // it keeps; when that does not fit, the choice falls back to the row-wise forms below.
                if (build_panels(panel_bits) != 0) (void)hipGetLastError();
                if (tile_panels > 0) spmv_kind = LANCZOS_HIP_SPMV_PANEL;
            }
        }
    } else if (spmv_kind == LANCZOS_HIP_SPMV_PANEL) {
        if (int rc = build_panels(panel_bits)) return rc;
        if (tile_panels == 0)
            return fail("lanczos_hip_create: the panel SpMV needs nnz < 2^32 and (row blocks x panels) < 2^32");
    }
    group = pick_group(mean);
    // rows per wavefront group of the pipe kernel: the largest of 64,32,16,8,4,1
    // whose typical nnz span still fits one tile
    log2r = 6;
    while (log2r > 0 && mean * (double)(1 << log2r) > 0.9 * kTile) log2r--;
    if (log2r == 1) log2r = 0;
    // uneven rows that the tiled form did not take: shares of entries
    if (spmv_kind == LANCZOS_HIP_SPMV_PIPE && requested == LANCZOS_HIP_SPMV_AUTO && uneven)
        spmv_kind = LANCZOS_HIP_SPMV_MERGE;
    if (spmv_kind == LANCZOS_HIP_SPMV_MERGE) {
        const int64_t tiles = (nnz + kTile - 1) / kTile;
        HIP_TRY(hipMalloc(&d_carry_row, sizeof(int) * 2 * std::max<int64_t>(tiles, 1)));
        HIP_TRY(hipMalloc(&d_carry_val, sizeof(double) * 2 * std::max<int64_t>(tiles, 1)));
        HIP_TRY(hipMalloc(&d_tile_rows, sizeof(int) * 2 * std::max<int64_t>(tiles, 1)));
        const int tb = (int)std::min<int64_t>(max_blocks, (2 * tiles + kBlock - 1) / kBlock);
        if (rp64)
            hipLaunchKernelGGL((k_merge_tile_rows<int64_t>), dim3(tb), dim3(kBlock), 0, stream, n, nnz,
                               (const int64_t *)d_rp, d_tile_rows);
        else
            hipLaunchKernelGGL((k_merge_tile_rows<int32_t>), dim3(tb), dim3(kBlock), 0, stream, n, nnz,
                               (const int32_t *)d_rp, d_tile_rows);
        HIP_TRY(hipGetLastError());
    }
    if (spmv_kind == LANCZOS_HIP_SPMV_PIPE || spmv_kind == LANCZOS_HIP_SPMV_PIPE_NT) {
        const int64_t groups = (n + (1 << log2r) - 1) >> log2r;
        const int64_t chunks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
        if (!d_xcd_split) HIP_TRY(hipMalloc(&d_xcd_split, sizeof(long long) * 9));
        if (rp64)
            hipLaunchKernelGGL((k_balance_split<int64_t>), dim3(1), dim3(64), 0, stream, n, (const int64_t *)d_rp,
                               (int64_t)kWavesPerBlock << log2r, chunks, d_xcd_split);
        else
            hipLaunchKernelGGL((k_balance_split<int32_t>), dim3(1), dim3(64), 0, stream, n, (const int32_t *)d_rp,
                               (int64_t)kWavesPerBlock << log2r, chunks, d_xcd_split);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
    }
    pdict = false;
    if (spmv_kind == LANCZOS_HIP_SPMV_ROWS && !slab && nnz > 0) {
        // stencil-like rows whose values do not repeat: their column offsets may (a grid with variable
        // coefficients); the product then walks the grid's planes with the values read from the CRS array
        // (an optimisation only: when the dictionary's temporaries, about 3 x 8 n bytes beside the sort's,
        // do not fit, the plain row-wise kernel runs)
        if (build_dict(false) != 0) {
            (void)hipGetLastError();
            free_dict();
        } else {
            pdict = pdict_usable();
            if (!pdict) free_dict();
        }
    }
    if (spmv_kind == LANCZOS_HIP_SPMV_MERGE)
        snprintf(spmv_name, sizeof(spmv_name), "merge%s", rp64 ? "_rp64" : "");
    else if (spmv_kind == LANCZOS_HIP_SPMV_DICT)
        snprintf(spmv_name, sizeof(spmv_name), "dict%d%s", dict_patterns,  // p: plane walk, a: loads a step ahead
                 dict_form() == 2 ? "p" : dict_form() == 1 ? "a" : "");
    else if (spmv_kind == LANCZOS_HIP_SPMV_PANEL)
        snprintf(spmv_name, sizeof(spmv_name), "panel%d%s", tile_bits, rp64 ? "_rp64" : "");
    else if (spmv_kind == LANCZOS_HIP_SPMV_ROWS)
        snprintf(spmv_name, sizeof(spmv_name), "rows64%s%s", rp64 ? "_rp64" : "", pdict ? "p" : "");  // p: plane walk
    else if (spmv_kind == LANCZOS_HIP_SPMV_PIPE || spmv_kind == LANCZOS_HIP_SPMV_PIPE_NT)
        snprintf(spmv_name, sizeof(spmv_name), "pipe%d%s%s", 1 << log2r,
                 spmv_kind == LANCZOS_HIP_SPMV_PIPE_NT ? "_nt" : "", rp64 ? "_rp64" : "");
    else if (spmv_kind == LANCZOS_HIP_SPMV_STREAM)
        snprintf(spmv_name, sizeof(spmv_name), "stream%s", rp64 ? "_rp64" : "");
    else
        snprintf(spmv_name, sizeof(spmv_name), "vector%d%s", group, rp64 ? "_rp64" : "");
    return 0;
}

int Engine::alloc_state() {
    // ---- Krylov basis: every vector padded with zeros to whole tiles
    npad = round_up(n, 2 * kTileV2);
    nvec = (reorth == LANCZOS_HIP_REORTH_NONE) ? 3 : max_iter;
    if (own_V) {
        ldv = npad;
        HIP_TRY(hipMalloc(&d_V, sizeof(double) * (size_t)ldv * (size_t)nvec));
    }
    HIP_TRY(hipMemsetAsync(d_V, 0, sizeof(double) * (size_t)ldv * (size_t)nvec, stream));

    // ---- small buffers
    HIP_TRY(hipMalloc(&d_partials, sizeof(double) * 2 * max_blocks));  // one SpMV, or one plus the edge rows
    HIP_TRY(hipMalloc(&d_norm_parts, sizeof(double) * 2 * max_blocks));
    HIP_TRY(hipMemsetAsync(d_norm_parts, 0, sizeof(double) * 2 * max_blocks, stream));
    lazy = (reorth == LANCZOS_HIP_REORTH_NONE || reorth == LANCZOS_HIP_REORTH_SELECTIVE) && !slab;
    lazy_slab = reorth == LANCZOS_HIP_REORTH_NONE && slab;
    hstride = max_blocks;
    const int hl = std::min(max_iter, kMaxDotsSegment);
    HIP_TRY(hipMalloc(&d_hpart, sizeof(double) * (size_t)hstride * (size_t)hl));
    HIP_TRY(hipMalloc(&d_red, sizeof(double) * (size_t)std::max(max_iter, 8)));
    HIP_TRY(hipMalloc(&d_scal, sizeof(double) * 8));
    HIP_TRY(hipMemsetAsync(d_scal, 0, sizeof(double) * 8, stream));
    hist_cap = (reorth == LANCZOS_HIP_REORTH_NONE) ? std::max(max_iter, 1 << 16) : max_iter;
    HIP_TRY(hipMalloc(&d_alpha, sizeof(double) * hist_cap));
    HIP_TRY(hipMalloc(&d_beta, sizeof(double) * hist_cap));
    HIP_TRY(hipMemsetAsync(d_alpha, 0, sizeof(double) * hist_cap, stream));
    HIP_TRY(hipMemsetAsync(d_beta, 0, sizeof(double) * hist_cap, stream));
    if (reorth == LANCZOS_HIP_REORTH_SELECTIVE) {
        omega = lanczos_hip_omega_create(max_iter);
        if (!omega) return fail("lanczos_hip_create: no memory for the orthogonality tracker");
        if (int rc = enable_host_history()) return rc;
        HIP_TRY(hipHostMalloc((void **)&h_sel, sizeof(double) * 8, hipHostMallocDefault));
        memset(h_sel, 0, sizeof(double) * 8);
        HIP_TRY(hipEventCreateWithFlags(&sel_event, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    k = 0;
    started = false;
    return 0;
}

Engine::~Engine() {
    if (device >= 0) (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    harvest_profile();
    for (auto ev : event_pool) (void)hipEventDestroy(ev);
    free_panels();
    free_dict();
    (void)hipFree(d_xcd_split);
    (void)hipFree(d_xcd_split_in);
    (void)hipFree(d_tile_rows);
    (void)hipFree(d_carry_row);
    (void)hipFree(d_carry_val);
    (void)hipFree(d_val);
    (void)hipFree(d_col);
    (void)hipFree(d_rp);
    if (own_V) (void)hipFree(d_V);
    (void)hipFree(d_partials);
    (void)hipFree(d_norm_parts);
    (void)hipFree(d_hpart);
    (void)hipFree(d_red);
    (void)hipFree(d_scal);
    (void)hipFree(d_alpha);
    (void)hipFree(d_beta);
    if (h_ext_x) (void)hipHostFree(h_ext_x);
    if (h_ext_w) (void)hipHostFree(h_ext_w);
    if (h_hist) (void)hipHostFree(h_hist);
    if (h_sel) (void)hipHostFree(h_sel);
    if (sel_event) (void)hipEventDestroy(sel_event);
    lanczos_hip_omega_destroy(omega);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------
int Engine::vec_blocks() const {
    const int64_t n2 = npad / 2;
    return (int)std::min<int64_t>(max_blocks, (n2 + kBlock - 1) / kBlock);
}

// The recurrence with deferred normalisation writes the vector the next product reads.  While a vector fits the L2s that
// store is best left cacheable; from 32 MiB on it is written past the caches, which also spares the next kernel the
// write-back of those lines: 256^3, SpMV + vector step on the dictionary 5990 -> 6310 it/s with both kernels faster
// (0.0718 -> 0.0672 and 0.0829 -> 0.0789 ms), 512^3 791 -> 845; 100^3 2 % slower, 128^3 equal.  On the row-wise CRS
// kernels, which gather x through the caches, the recurrence gains 4.8 us and the product loses 2.6 (2296 -> 2308 it/s).
// The same store in that SpMV's own y changed nothing (0.3425 ms either way) and is not there.
// LANCZOS_HIP_NT_STORES = 0 / 1 overrides, read at every launch.
bool Engine::nt_stores() const {
    const char *env = getenv("LANCZOS_HIP_NT_STORES");
    return env && *env ? atoi(env) != 0 : 8.0 * (double)n >= 32.0 * 1048576.0;
}

// how many basis vectors the fused recurrence + dot products kernel takes (beyond: two kernels): 32, the break-even on
// vectors in HBM; any number while a vector is at most 2 MiB (10 000 rows, 100 steps: 3.91 -> 3.77 ms, 64^3: 8.63 -> 8.47)
int Engine::fuse_limit() const {
    if (const char *env = getenv("LANCZOS_HIP_FUSE_MAX")) return atoi(env);
    return 8.0 * (double)npad <= 2.0 * 1048576.0 ? kMaxDotsSegment : 32;
}

using RecurLazyKernel = void (*)(double2 *, const double2 *, const double2 *, int64_t, int, const double *, int, const double *, int,
                                 double *, double *, double *, double *);
RecurLazyKernel Engine::recur_lazy_kernel() const { return nt_stores() ? k_recur_lazy<true> : k_recur_lazy<false>; }

// grid of the recurrence and scaling kernels: whole tiles, two workgroups per CU (512^3, k_recur_lazy with
// 2 / 4 / 8 per CU: 0.741 / 0.785 / 0.766 ms; LANCZOS_HIP_VEC_WG_PER_CU for measurements)
int Engine::recur_blocks() const {
    static const int per_cu = [] {
        const char *s = getenv("LANCZOS_HIP_VEC_WG_PER_CU");
        const int v = s ? atoi(s) : 0;
        return v >= 1 && v <= 8 ? v : 2;
    }();
    const int64_t tiles = (npad / 2) / kTileV2;
    return (int)std::min<int64_t>((int64_t)cus * per_cu, std::max<int64_t>(tiles, 1));
}

int Engine::tile_blocks() const {
    // 4 workgroups per CU: measured best for the basis-streaming kernels
    const int64_t tiles = (npad / 2) / kTileV2;
    return (int)std::min<int64_t>(max_blocks / 2, std::max<int64_t>(tiles, 1));
}

double Engine::spmv_bytes() const {
    if (spmv_kind == LANCZOS_HIP_SPMV_DICT)  // what this format reads and writes: pattern numbers, x, y
        return 2.0 * (double)n + 8.0 * (double)n_ext + 8.0 * (double)n + 12.0 * (double)dict_patterns * dict_stride;
    if (spmv_kind == LANCZOS_HIP_SPMV_ROWS && pdict)  // values and row offsets of the CRS arrays, pattern numbers, x, y
        return 8.0 * (double)nnz + 4.0 * (double)(n + 1) + 2.0 * (double)n + 8.0 * (double)n_ext + 8.0 * (double)n;
    return 12.0 * (double)nnz + (rp64 ? 8.0 : 4.0) * (double)(n + 1) + 8.0 * (double)n_ext + 8.0 * (double)n;
}

// Rows [r_a, r_a + rows) of the product: the row offsets, y and xdot shifted, x and the entry arrays as
// they are (column indices are absolute).  The merge and pipe forms hold precomputed splits of the
// whole matrix and only take r_a = 0, rows = n (the pipe form also takes the one sub-range whose XCD ranges
// phase_spmv_begin has prepared in d_xcd_split_in).
template <typename RP>
static void launch_spmv_t(Engine *e, const double *x, const double *xdot_all, double *y_all, int *nparts,
                          int64_t r_a, int64_t rows, double *parts) {
    const RP *rp = (const RP *)e->d_rp + r_a;
    const double *xdot = xdot_all + r_a;
    double *y = y_all + r_a;
    if (e->spmv_kind == LANCZOS_HIP_SPMV_MERGE) {
        const int64_t tiles = (e->nnz + kTile - 1) / kTile;
        const int fix = 64;
        int blocks = (int)std::min<int64_t>(e->max_blocks - fix, round_up((tiles + kWavesPerBlock - 1) / kWavesPerBlock, 8));
        *nparts = blocks + fix;
        (void)hipMemsetAsync(y, 0, sizeof(double) * e->n, e->stream);
        hipLaunchKernelGGL((k_spmv_merge<RP>), dim3(blocks), dim3(kBlock), 0, e->stream, e->n, e->nnz, rp, e->d_col,
                           e->d_val, x, xdot, y, e->d_partials, (const int *)e->d_tile_rows, e->d_carry_row, e->d_carry_val);
        hipLaunchKernelGGL(k_merge_fixup, dim3(fix), dim3(kBlock), 0, e->stream, 2 * tiles, (const int *)e->d_carry_row,
                           (const double *)e->d_carry_val, xdot, y, e->d_partials + blocks);
        return;
    }
    if (e->spmv_kind == LANCZOS_HIP_SPMV_ROWS) {
        const int64_t groups = (rows + 63) / 64;
        const int64_t chunks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
        // two workgroups per CU: the kernel saturates the memory pipeline with 8 wavefronts per CU, and
        // fewer, longer-running wavefronts spend less of their life in the pipeline's prologue (measured
        // 0.363 ms against 0.370 ms with eight workgroups per CU on the 256^3 Laplacian)
        int blocks = (int)std::min<int64_t>(e->cus * 2 / 8 * 8, round_up(chunks, 8));
        *nparts = blocks;
        // LANCZOS_HIP_ROWS_MASK=0: the tile loads without the descriptors that end at the group's last entry
        static const bool mask = !(getenv("LANCZOS_HIP_ROWS_MASK") && atoi(getenv("LANCZOS_HIP_ROWS_MASK")) == 0);
        // LANCZOS_HIP_ROWS_DMA=1: the tiles loaded straight into LDS (read at every launch: a measurement switch)
        const char *dma = getenv("LANCZOS_HIP_ROWS_DMA");
        if (dma && atoi(dma) != 0)
            hipLaunchKernelGGL((k_spmv_rows_dma<RP>), dim3(blocks), dim3(kBlock), 0, e->stream, rows, e->nnz, rp,
                               e->d_col, e->d_val, x, xdot, y, parts);
        else if (mask)
            hipLaunchKernelGGL((k_spmv_rows<RP, true>), dim3(blocks), dim3(kBlock), 0, e->stream, rows, e->nnz, rp,
                               e->d_col, e->d_val, x, xdot, y, parts);
        else
            hipLaunchKernelGGL((k_spmv_rows<RP, false>), dim3(blocks), dim3(kBlock), 0, e->stream, rows, e->nnz, rp,
                               e->d_col, e->d_val, x, xdot, y, parts);
        return;
    }
    if (e->spmv_kind == LANCZOS_HIP_SPMV_PIPE || e->spmv_kind == LANCZOS_HIP_SPMV_PIPE_NT) {
        const int lr = e->log2r;
        const int64_t groups = (rows + (1 << lr) - 1) >> lr;
        const int64_t chunks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
        int blocks = (int)std::min<int64_t>(e->max_blocks, round_up(chunks, 8));
        *nparts = blocks;
        const bool nt = e->spmv_kind == LANCZOS_HIP_SPMV_PIPE_NT;
        // the XCD ranges of the rows in hand: the whole matrix, or the rows of a split product
        const long long *split = (r_a == 0 && rows == e->n) ? e->d_xcd_split : e->d_xcd_split_in;
#define LZ_PIPE_CASE(LR)                                                                                       \
    case LR:                                                                                                   \
        if (nt)                                                                                                \
            hipLaunchKernelGGL((k_spmv_pipe<RP, LR, true>), dim3(blocks), dim3(kBlock), 0, e->stream, rows,   \
                               rp, e->d_col, e->d_val, x, xdot, y, parts, split);                              \
        else                                                                                                   \
            hipLaunchKernelGGL((k_spmv_pipe<RP, LR, false>), dim3(blocks), dim3(kBlock), 0, e->stream, rows,  \
                               rp, e->d_col, e->d_val, x, xdot, y, parts, split);                              \
        break;
        switch (lr) {
            LZ_PIPE_CASE(6)
            LZ_PIPE_CASE(5)
            LZ_PIPE_CASE(4)
            LZ_PIPE_CASE(3)
            LZ_PIPE_CASE(2)
            default:
                LZ_PIPE_CASE(0)
        }
#undef LZ_PIPE_CASE
        return;
    }
    if (e->spmv_kind == LANCZOS_HIP_SPMV_STREAM) {
        const int64_t groups = (rows + 63) / 64;
        const int64_t chunks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
        int blocks = (int)std::min<int64_t>(e->max_blocks, round_up(chunks, 8));
        *nparts = blocks;
        hipLaunchKernelGGL((k_spmv_stream<RP>), dim3(blocks), dim3(kBlock), 0, e->stream, rows, rp, e->d_col,
                           e->d_val, x, xdot, y, parts);
        return;
    }
#define LZ_VEC_CASE(G)                                                                                        \
    case G: {                                                                                                 \
        const int64_t chunks = (rows + (kBlock / G) - 1) / (kBlock / G);                                      \
        int blocks = (int)std::min<int64_t>(e->max_blocks, round_up(chunks, 8));                              \
        *nparts = blocks;                                                                                     \
        hipLaunchKernelGGL((k_spmv_vector<G, RP>), dim3(blocks), dim3(kBlock), 0, e->stream, rows, rp,       \
                           e->d_col, e->d_val, x, xdot, y, parts);                                            \
    } break;
    switch (e->group) {
        LZ_VEC_CASE(2)
        LZ_VEC_CASE(4)
        LZ_VEC_CASE(8)
        LZ_VEC_CASE(16)
        LZ_VEC_CASE(32)
        default:
            LZ_VEC_CASE(64)
    }
#undef LZ_VEC_CASE
}

int Engine::launch_spmv(const double *x, const double *xdot, double *y, int *nparts) {
    ProfScope ps(this, LANCZOS_HIP_K_SPMV, spmv_bytes());
    if (nnz == 0) {  // the zero matrix: y = 0, alpha = 0 (the kernels assume one valid entry)
        HIP_TRY(hipMemsetAsync(y, 0, sizeof(double) * n, stream));
        HIP_TRY(hipMemsetAsync(d_partials, 0, sizeof(double) * 8, stream));
        *nparts = 8;
        return 0;
    }
    if (spmv_kind == LANCZOS_HIP_SPMV_DICT) return launch_spmv_dict(x, xdot, y, nparts);
    if (spmv_kind == LANCZOS_HIP_SPMV_PANEL) return launch_spmv_panels(x, xdot, y, nparts);
    if (spmv_kind == LANCZOS_HIP_SPMV_ROWS && pdict) return launch_spmv_pdict(x, xdot, y, nparts);
    if (rp64)
        launch_spmv_t<int64_t>(this, x, xdot, y, nparts, 0, n, d_partials);
    else
        launch_spmv_t<int32_t>(this, x, xdot, y, nparts, 0, n, d_partials);
    HIP_TRY(hipGetLastError());
    return 0;
}

int Engine::launch_finalize(const double *parts, int nparts, double *out) {
    ProfScope ps(this, LANCZOS_HIP_K_SMALL, 8.0 * nparts);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(kBlock), 0, stream, parts, nparts, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int Engine::launch_dots(int l0, int l1, double *red_out) {
    // red_out[l - l0] = V_l . w  for l in [l0, l1), w = vec(k+1)
    const int blocks = tile_blocks();
    for (int s0 = l0; s0 < l1; s0 += kMaxDotsSegment) {
        const int s1 = std::min(l1, s0 + kMaxDotsSegment);
        const int nl = s1 - s0;
        // long vectors and more than the 32 vectors the fused sweep takes: k_dots_big (LANCZOS_HIP_DOTS_BIG=0 / 1
        // forces the choice, read per launch)
        constexpr int kBigP = 32;
        const int64_t big_groups = (npad / 2) / kTileV2 / (kBigP / kEPT2);
        const char *big_env = getenv("LANCZOS_HIP_DOTS_BIG");
        const bool big = big_env && *big_env ? atoi(big_env) != 0 && big_groups >= 1 : big_groups >= 2 * (int64_t)cus && nl > 32;
        const int grid = big ? (int)std::min<int64_t>(blocks, big_groups) : blocks;
        {
            ProfScope ps(this, LANCZOS_HIP_K_DOTS, 8.0 * (double)n * nl + 8.0 * (double)n);
            if (big)
                hipLaunchKernelGGL(k_dots_big<kBigP>, dim3(grid), dim3(kBlock), sizeof(double) * kWavesPerBlock * nl, stream,
                                   (const double2 *)d_V, ldv / 2, s0, s1, (const double2 *)vec(k + 1), npad / 2,
                                   d_hpart, hstride);
            else
                hipLaunchKernelGGL(k_dots, dim3(grid), dim3(kBlock), sizeof(double) * kWavesPerBlock * nl, stream,
                                   (const double2 *)d_V, ldv / 2, s0, s1, (const double2 *)vec(k + 1), npad / 2,
                                   d_hpart, hstride);
        }
        {
            ProfScope ps(this, LANCZOS_HIP_K_SMALL, 8.0 * (double)grid * nl);
            hipLaunchKernelGGL(k_hreduce, dim3(nl), dim3(kBlock), 0, stream, d_hpart, hstride, grid,
                               red_out + (s0 - l0));
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The recurrence of iteration k (k > 0) and red_out[l - l0] = V_l . w for l in [l0, l1) in one sweep over w; the
// first kMaxDotsSegment basis vectors ride with the recurrence, any further ones take plain k_dots launches.
int Engine::launch_recur_dots(const double *alpha_parts, int alpha_cnt, int l0, int l1, double *red_out) {
    const int blocks = tile_blocks();
    double *ahost = h_hist ? h_hist + 2 * (size_t)k : nullptr;
    const int s1 = std::min(l1, l0 + kMaxDotsSegment);
    const int nl = s1 - l0;
    {
        ProfScope ps(this, LANCZOS_HIP_K_DOTS, 8.0 * (double)n * nl + 32.0 * (double)n);
        hipLaunchKernelGGL(k_recur_dots, dim3(blocks), dim3(kBlock), sizeof(double) * (kWavesPerBlock * nl + kWavesPerBlock),
                           stream, (const double2 *)d_V, ldv / 2, l0, s1, (double2 *)vec(k + 1), (const double2 *)vec(k),
                           (const double2 *)vec(k - 1), npad / 2, alpha_parts, alpha_cnt, (const double *)(d_scal + 1),
                           d_hpart, hstride, d_alpha + k, ahost);
    }
    {
        ProfScope ps(this, LANCZOS_HIP_K_SMALL, 8.0 * (double)blocks * nl);
        hipLaunchKernelGGL(k_hreduce, dim3(nl), dim3(kBlock), 0, stream, d_hpart, hstride, blocks, red_out);
    }
    HIP_TRY(hipGetLastError());
    if (s1 < l1) return launch_dots(s1, l1, red_out + nl);
    return 0;
}

int Engine::launch_update(int l0, int l1, const double *h, int *nparts) {
    int blocks = tile_blocks();
    ProfScope ps(this, LANCZOS_HIP_K_UPDATE, 8.0 * (double)n * (l1 - l0) + 16.0 * (double)n);
    // long vectors and a pass of some length: 128 KiB of every basis vector per workgroup at a time (k_update_big;
    // LANCZOS_HIP_UPDATE_BIG=0 / 1 forces the choice, read per launch: tests compare the two)
    constexpr int kBigP = 32;
    const int64_t big_groups = (npad / 2) / kTileV2 / (kBigP / kEPT2);
    const char *big_env = getenv("LANCZOS_HIP_UPDATE_BIG");
    const bool big = big_env && *big_env ? atoi(big_env) != 0 && big_groups >= 1 : big_groups >= 2 * (int64_t)cus && l1 - l0 >= 8;
    if (big) {
        blocks = (int)std::min<int64_t>(blocks, std::max<int64_t>(big_groups, 1));
        hipLaunchKernelGGL(k_update_big<kBigP>, dim3(blocks), dim3(kBlock), 0, stream, (const double2 *)d_V, ldv / 2, l0, l1,
                           (double2 *)vec(k + 1), npad / 2, h, norm_parts(k));
    } else {
        hipLaunchKernelGGL(k_update, dim3(blocks), dim3(kBlock), 0, stream, (const double2 *)d_V, ldv / 2, l0, l1,
                           (double2 *)vec(k + 1), npad / 2, h, norm_parts(k));
    }
    *nparts = blocks;
    norm_count = blocks;
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// start vector
// ---------------------------------------------------------------------------
int Engine::set_start(const double *v0, bool normalise, double scale) {
    if (!v0) return fail("lanczos_hip_set_start: NULL vector");
    if (int rc = check_affinity(this, "set_start")) return rc;
    HIP_TRY(hipSetDevice(device));
    // clear every slot: a second run must not see the previous basis
    HIP_TRY(hipMemsetAsync(d_V, 0, sizeof(double) * (size_t)ldv * (size_t)nvec, stream));
    HIP_TRY(hipMemcpyAsync(vec(0), v0, sizeof(double) * n, hipMemcpyHostToDevice, stream));
    const int blocks = vec_blocks();
    if (normalise) {
        hipLaunchKernelGGL(k_dot, dim3(blocks), dim3(kBlock), 0, stream, (const double2 *)vec(0),
                           (const double2 *)vec(0), npad / 2, d_partials);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(kBlock), 0, stream, (const double *)d_partials, blocks, d_red);
        hipLaunchKernelGGL(k_scal_mul, dim3(blocks), dim3(kBlock), 0, stream, (double2 *)vec(0), npad / 2,
                           (const double *)d_red, 0.0);
    } else {
        hipLaunchKernelGGL(k_scal_mul, dim3(blocks), dim3(kBlock), 0, stream, (double2 *)vec(0), npad / 2,
                           (const double *)nullptr, scale);
    }
    HIP_TRY(hipMemsetAsync(d_scal, 0, sizeof(double) * 8, stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));  // v0 may be pageable host memory
    k = 0;
    lazy_flushed_k = 0;
    spec_spmv_k = -1;
    norm_count = 0;
    reorth_steps = 0;
    if (omega) lanczos_hip_omega_reset(omega);
    started = true;
    return 0;
}

// ---------------------------------------------------------------------------
// iteration phases
// ---------------------------------------------------------------------------
int Engine::check_can_step() {
    if (int rc = check_affinity(this, "iteration")) return rc;
    if (!started) return fail("lanczos_hip: no start vector set");
    if (reorth != LANCZOS_HIP_REORTH_NONE && k >= max_iter - 1)
        return fail("lanczos_hip: basis full (%d iterations of max_iter=%d done)", k, max_iter);
    if (k >= hist_cap)
        return fail("lanczos_hip: history full: the three-vector mode records at most max(max_iter, 65536) = %d "
                    "iterations per run", hist_cap);
    return 0;
}

void Engine::reorth_range(int *l0, int *l1) const {
    *l0 = 0;
    switch (reorth) {
        case LANCZOS_HIP_REORTH_NONE: *l1 = 0; break;
        case LANCZOS_HIP_REORTH_CUDA_COMPAT: *l1 = k - 2; break;
        case LANCZOS_HIP_REORTH_SELECTIVE: *l1 = k + 1; break;
        default: *l1 = k; break;
    }
    if (*l1 < 0) *l1 = 0;
}

int Engine::phase_spmv(double *red) {
    if (int rc = check_can_step()) return rc;
    const double *x = x_ext ? x_ext : vec(k);
    walk_pending = lazy_slab && two_pass_ok();
    if (walk_pending) {  // the first of two plane walks: u_k . A u_k, nothing written
        ProfScope ps(this, LANCZOS_HIP_K_SPMV, 10.0 * (double)n);
        if (int rc = launch_spmv_dict(vec(k), vec(k), nullptr, &alpha_count, 0, -1, 1, nullptr, nullptr)) return rc;
        return finalize_spmv(red);
    }
    if (int rc = launch_spmv(x, vec(k), vec(k + 1), &alpha_count)) return rc;
    return finalize_spmv(red);
}

// red[0] = this slab's u_k . w; with deferred normalisation the slab's part of |u_k|^2 rides along
// in red[1]: one reduction over ranks per iteration
int Engine::finalize_spmv(double *red) {
    if (!lazy_slab) return launch_finalize(d_partials, alpha_count, red);
    ProfScope ps(this, LANCZOS_HIP_K_SMALL, 8.0 * (alpha_count + norm_count));
    hipLaunchKernelGGL(k_finalize2, dim3(1), dim3(kBlock), 0, stream, (const double *)d_partials, alpha_count,
                       (const double *)norm_parts(k - 1), k > 0 ? norm_count : 0, red);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Rows [0, lo) and [hi, n) of the product through the plain row-wise form (src/crs.c:167-177), one
// launch; parts_out receives the partials of xdot . y.
int Engine::launch_spmv_edges(int64_t lo, int64_t hi, double *parts_out, int *nparts) {
    *nparts = 0;
    const int64_t m = lo + (n - hi);
    if (m <= 0) return 0;
    const double *x = vec(k);
    double *y = vec(k + 1);
    ProfScope ps(this, LANCZOS_HIP_K_SPMV, 0.0);  // these rows' bytes are part of the whole product's figure
#define LZ_EDGE_CASE(G)                                                                                       \
    case G: {                                                                                                 \
        const int64_t chunks = (m + (kBlock / G) - 1) / (kBlock / G);                                         \
        const int blocks = (int)std::min<int64_t>(max_blocks, chunks);                                        \
        *nparts = blocks;                                                                                     \
        if (rp64)                                                                                             \
            hipLaunchKernelGGL((k_spmv_edges<G, int64_t>), dim3(blocks), dim3(kBlock), 0, stream, lo, hi, n,  \
                               (const int64_t *)d_rp, d_col, d_val, x, x, y, parts_out);                      \
        else                                                                                                  \
            hipLaunchKernelGGL((k_spmv_edges<G, int32_t>), dim3(blocks), dim3(kBlock), 0, stream, lo, hi, n,  \
                               (const int32_t *)d_rp, d_col, d_val, x, x, y, parts_out);                      \
    } break;
    switch (group) {
        LZ_EDGE_CASE(2)
        LZ_EDGE_CASE(4)
        LZ_EDGE_CASE(8)
        LZ_EDGE_CASE(16)
        LZ_EDGE_CASE(32)
        default:
            LZ_EDGE_CASE(64)
    }
#undef LZ_EDGE_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

// The product while the remote entries are still on their way: every row is computed, but only rows
// [b_lo, b_hi) -- those that read nothing remote -- count; the others are done again by phase_spmv_end.
int Engine::phase_spmv_begin(int64_t b_lo, int64_t b_hi) {
    if (int rc = check_can_step()) return rc;
    if (!can_split_product())
        return fail("lanczos_hip_phase_spmv_begin: needs a caller-owned basis and the dictionary, rows, pipe, stream or "
                    "vector SpMV");
    if (b_lo < 0 || b_hi > n || b_lo > b_hi) return fail("lanczos_hip_phase_spmv_begin: bad row range");
    walk_pending = lazy_slab && two_pass_ok();
    ProfScope ps(this, LANCZOS_HIP_K_SPMV, walk_pending ? 10.0 * (double)n : spmv_bytes());
    if (walk_pending) {
        // the first of two plane walks: the partial sums of u_k . A u_k over the rows that read nothing remote, nothing
        // written; phase_spmv_end adds the edge rows' share, phase_recur_lazy is the second walk
        if (int rc = launch_spmv_dict(vec(k), vec(k), nullptr, &alpha_count, b_lo, b_hi, 1, nullptr, nullptr)) return rc;
    } else if (spmv_kind == LANCZOS_HIP_SPMV_DICT) {
        // every row at once; the edge rows are done again later and do not count towards the dot product
        if (int rc = launch_spmv_dict(vec(k), vec(k), vec(k + 1), &alpha_count, b_lo, b_hi)) return rc;
    } else {
        // the interior rows alone, from a multiple of 64 (the rows form's 64-row groups were sized on those)
        b_lo = std::min(round_up(b_lo, 64), b_hi);
        alpha_count = 0;
        const bool pipe = spmv_kind == LANCZOS_HIP_SPMV_PIPE || spmv_kind == LANCZOS_HIP_SPMV_PIPE_NT;
        if (pipe && b_hi > b_lo && (b_lo != split_in_lo || b_hi != split_in_hi)) {
            // XCD ranges of equal entry counts for these rows (the same range every iteration: computed once)
            const int64_t rows_in = b_hi - b_lo;
            const int64_t groups = (rows_in + (1 << log2r) - 1) >> log2r;
            const int64_t chunks = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
            if (!d_xcd_split_in) HIP_TRY(hipMalloc(&d_xcd_split_in, sizeof(long long) * 9));
            if (rp64)
                hipLaunchKernelGGL((k_balance_split<int64_t>), dim3(1), dim3(64), 0, stream, rows_in,
                                   (const int64_t *)d_rp + b_lo, (int64_t)kWavesPerBlock << log2r, chunks, d_xcd_split_in);
            else
                hipLaunchKernelGGL((k_balance_split<int32_t>), dim3(1), dim3(64), 0, stream, rows_in,
                                   (const int32_t *)d_rp + b_lo, (int64_t)kWavesPerBlock << log2r, chunks, d_xcd_split_in);
            HIP_TRY(hipGetLastError());
            split_in_lo = b_lo;
            split_in_hi = b_hi;
        }
        if (b_hi > b_lo) {
            if (rp64)
                launch_spmv_t<int64_t>(this, vec(k), vec(k), vec(k + 1), &alpha_count, b_lo, b_hi - b_lo, d_partials);
            else
                launch_spmv_t<int32_t>(this, vec(k), vec(k), vec(k + 1), &alpha_count, b_lo, b_hi - b_lo, d_partials);
            HIP_TRY(hipGetLastError());
        }
    }
    pending_lo = b_lo;
    pending_hi = b_hi;
    return 0;
}

bool Engine::can_split_product() const {
    return slab && !x_ext && nnz > 0 &&
           (spmv_kind == LANCZOS_HIP_SPMV_DICT || spmv_kind == LANCZOS_HIP_SPMV_ROWS ||
            spmv_kind == LANCZOS_HIP_SPMV_STREAM || spmv_kind == LANCZOS_HIP_SPMV_VECTOR ||
            spmv_kind == LANCZOS_HIP_SPMV_PIPE || spmv_kind == LANCZOS_HIP_SPMV_PIPE_NT);
}

// The remote entries have arrived: the rows that read them, then the reductions as phase_spmv.
int Engine::phase_spmv_end(double *red) {
    if (int rc = check_affinity(this, "phase_spmv_end")) return rc;
    if (pending_lo < 0) return fail("lanczos_hip_phase_spmv_end: no product pending");
    int np = 0;
    if (int rc = launch_spmv_edges(pending_lo, pending_hi, d_partials + alpha_count, &np)) return rc;
    alpha_count += np;
    pending_lo = pending_hi = -1;
    return finalize_spmv(red);
}

// Recurrence with deferred normalisation on a row slab: red[0] = u_k . A u_k and red[1] = |u_k|^2,
// both summed over ranks.  Leaves the slab's partials of |u_{k+1}|^2 for the next phase_spmv.
int Engine::phase_recur_lazy(double *red) {
    if (int rc = check_affinity(this, "phase_recur_lazy")) return rc;
    if (!lazy_slab) return fail("lanczos_hip_phase_recur_lazy: needs reorth=none on a row-partitioned engine");
    if (walk_pending) {
        // the second plane walk: A u_k again (the remote planes have arrived) with the recurrence in its epilogue
        walk_pending = false;
        ProfScope ps(this, LANCZOS_HIP_K_RECUR, 26.0 * (double)n);
        const DictRecur rc{k > 0 ? vec(k - 1) : vec(k), k, red, 1, red + 1, 1, d_alpha, d_beta, h_hist};
        int blocks = 0;
        if (int r = launch_spmv_dict(vec(k), vec(k), vec(k + 1), &blocks, 0, -1, 2, &rc, norm_parts(k))) return r;
        norm_count = blocks;
        k += 1;
        return 0;
    }
    const int blocks = recur_blocks();
    {
        ProfScope ps(this, LANCZOS_HIP_K_RECUR, 32.0 * (double)n);
        hipLaunchKernelGGL(recur_lazy_kernel(), dim3(blocks), dim3(kBlock), 0, stream, (double2 *)vec(k + 1),
                           (const double2 *)vec(k), (const double2 *)(k > 0 ? vec(k - 1) : nullptr), npad / 2, k,
                           (const double *)red, 1, (const double *)(red + 1), 1, norm_parts(k), d_alpha, d_beta,
                           h_hist);
    }
    norm_count = blocks;
    HIP_TRY(hipGetLastError());
    k += 1;
    return 0;
}

// beta of the last deferred iteration: red[0] = this slab's |u_k|^2 ...
int Engine::phase_lazy_norm(double *red) {
    if (int rc = check_affinity(this, "phase_lazy_norm")) return rc;
    if (!lazy_slab || k == 0) return fail("lanczos_hip_phase_lazy_norm: nothing deferred");
    return launch_finalize(norm_parts(k - 1), norm_count, red);
}

// ... and, summed over ranks, beta_{k-1} = sqrt(red[0]) into the history
int Engine::phase_lazy_close(double *red) {
    if (int rc = check_affinity(this, "phase_lazy_close")) return rc;
    if (!lazy_slab || k == 0) return fail("lanczos_hip_phase_lazy_close: nothing deferred");
    double *bhost = h_hist ? h_hist + 2 * (size_t)(k - 1) + 1 : nullptr;
    hipLaunchKernelGGL(k_lazy_flush, dim3(1), dim3(kBlock), 0, stream, (const double *)red, 1, d_beta + (k - 1), bhost);
    HIP_TRY(hipGetLastError());
    return 0;
}

// w <- (w - beta v_{k-1}) - alpha v_k ; alpha = sum of alpha_parts[0..alpha_cnt)
int Engine::launch_recur(const double *alpha_parts, int alpha_cnt) {
    const int blocks = recur_blocks();
    ProfScope ps(this, LANCZOS_HIP_K_RECUR, 32.0 * (double)n);
    double *ahost = h_hist ? h_hist + 2 * (size_t)k : nullptr;
    hipLaunchKernelGGL(k_recur, dim3(blocks), dim3(kBlock), 0, stream, (double2 *)vec(k + 1),
                       (const double2 *)vec(k), (const double2 *)(k > 0 ? vec(k - 1) : nullptr), npad / 2, alpha_parts,
                       alpha_cnt, (const double *)(d_scal + 1), norm_parts(k), d_alpha + k, ahost);
    norm_count = blocks;
    HIP_TRY(hipGetLastError());
    return 0;
}

// v_{k+1} = w / sqrt(sum of parts[0..cnt)) ; closes iteration k
int Engine::launch_scale(const double *parts, int cnt) {
    const int blocks = recur_blocks();
    {
        ProfScope ps(this, LANCZOS_HIP_K_SCALE, 16.0 * (double)n);
        double *bhost = h_hist ? h_hist + 2 * (size_t)k + 1 : nullptr;
        hipLaunchKernelGGL(k_scale, dim3(blocks), dim3(kBlock), 0, stream, (double2 *)vec(k + 1), npad / 2, parts, cnt,
                           d_scal + 1, d_beta + k, bhost);
    }
    HIP_TRY(hipGetLastError());
    k += 1;
    return 0;
}

int Engine::phase_recur(double *red, int *n_red, int *is_norm) {
    if (int rc = check_affinity(this, "phase_recur")) return rc;
    if (lazy_slab) return fail("lanczos_hip_phase_recur: reorth=none on a row slab runs phase_recur_lazy");
    if (reorth == LANCZOS_HIP_REORTH_MGS)
        return fail("lanczos_hip_phase_recur: the one-vector-at-a-time order (mgs) is not available phase by phase "
                    "(it would need one reduction over ranks per basis vector); use full or full2");
    // red[0] holds alpha (already reduced over ranks by the caller)
    if (int rc = launch_recur(red, 1)) return rc;
    int l0, l1;
    reorth_range(&l0, &l1);
    if (l1 - l0 <= 0 || reorth == LANCZOS_HIP_REORTH_SELECTIVE) {  // selective: the caller decides about dots
        *n_red = 1;
        *is_norm = 1;
        red_has_norm = false;
        return launch_finalize(norm_parts(k), norm_count, red);
    }
    // red[0..nl) = local V_l . w, red[nl] = local w.w: one reduction over ranks covers both, and
    // phase_update derives the norm after the pass from them
    *n_red = l1 - l0 + 1;
    *is_norm = 0;
    red_has_norm = true;
    if (int rc = launch_dots(l0, l1, red)) return rc;
    return launch_finalize(norm_parts(k), norm_count, red + (l1 - l0));
}

int Engine::phase_dots(double *red, int *n_red) {
    if (int rc = check_affinity(this, "phase_dots")) return rc;
    int l0, l1;
    reorth_range(&l0, &l1);
    *n_red = l1 - l0;
    if (*n_red <= 0) return fail("lanczos_hip_phase_dots: empty range");
    red_has_norm = false;
    return launch_dots(l0, l1, red);
}

int Engine::phase_update(double *red) {
    if (int rc = check_affinity(this, "phase_update")) return rc;
    int l0, l1;
    reorth_range(&l0, &l1);
    if (l1 - l0 <= 0) return fail("lanczos_hip_phase_update: empty range");
    // k_update reads the coefficients from `red`; the finaliser that then
    // overwrites red[0] is ordered behind it on the stream
    int nparts = 0;
    if (int rc = launch_update(l0, l1, red, &nparts)) return rc;
    if (red_has_norm) {  // red came from phase_recur, reduced over ranks: the global norm follows from it
        red_has_norm = false;
        ProfScope ps(this, LANCZOS_HIP_K_SMALL, 8.0 * (l1 - l0 + 1));
        hipLaunchKernelGGL(k_pyth_norm, dim3(1), dim3(kBlock), 0, stream, (const double *)red, l1 - l0,
                           (const double *)(red + (l1 - l0)), red);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    return launch_finalize(norm_parts(k), norm_count, red);
}

int Engine::phase_scale(double *red) {
    if (int rc = check_affinity(this, "phase_scale")) return rc;
    return launch_scale(red, 1);
}

// Record beta of the last deferred-normalisation iteration (its w.w partials are
// otherwise only consumed by the next iteration's recurrence).
int Engine::flush_lazy() {
    if (!lazy || k == 0 || lazy_flushed_k == k) return 0;
    double *bhost = h_hist ? h_hist + 2 * (size_t)(k - 1) + 1 : nullptr;
    hipLaunchKernelGGL(k_lazy_flush, dim3(1), dim3(kBlock), 0, stream, (const double *)norm_parts(k - 1), norm_count,
                       d_beta + (k - 1), bhost);
    HIP_TRY(hipGetLastError());
    lazy_flushed_k = k;
    return 0;
}

// The caller's operator: v_k to the host, the plugin, the product back; the alpha partials are
// then those of a plain dot product.  The stream is waited on once per product.
int Engine::set_operator(lanczos_hip_matvec_fn *fn, const void *mat) {
    if (!fn) return fail("lanczos_hip_set_operator: NULL function");
    if (slab) return fail("lanczos_hip_set_operator: not available on a row-partitioned engine");
    HIP_TRY(hipSetDevice(device));
    if (!h_ext_x) HIP_TRY(hipHostMalloc((void **)&h_ext_x, sizeof(double) * (size_t)n, hipHostMallocDefault));
    if (!h_ext_w) HIP_TRY(hipHostMalloc((void **)&h_ext_w, sizeof(double) * (size_t)n, hipHostMallocDefault));
    ext_matvec = fn;
    ext_mat = mat;
    snprintf(spmv_name, sizeof(spmv_name), "host_operator");
    return 0;
}

int Engine::launch_product(int kk) {
    if (!ext_matvec) return launch_spmv(vec(kk), vec(kk), vec(kk + 1), &alpha_count);
    HIP_TRY(hipMemcpyAsync(h_ext_x, vec(kk), sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    ext_matvec(ext_mat, h_ext_x, h_ext_w);  // overwrites all of dist, as both reference operators do
    HIP_TRY(hipMemcpyAsync(vec(kk + 1), h_ext_w, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream));
    const int blocks = vec_blocks();
    ProfScope ps(this, LANCZOS_HIP_K_SPMV, 32.0 * (double)n);
    hipLaunchKernelGGL(k_dot, dim3(blocks), dim3(kBlock), 0, stream, (const double2 *)vec(kk),
                       (const double2 *)vec(kk + 1), npad / 2, d_partials);
    alpha_count = blocks;
    HIP_TRY(hipGetLastError());
    return 0;
}

// An iteration without reorthogonalisation on a grid's dictionary form as two plane walks: the first only reads
// and leaves the partial sums of u_k . A u_k, the second forms A u_k again and writes u_{k+1} by the
// deferred-normalisation recurrence in its epilogue (dict.hip, MODE 1 and 2).  36 bytes per row in place of the
// 50 of product + k_recur_lazy, and one launch that writes in place of two.  LANCZOS_HIP_TWO_PASS=0 / 1 forces
// the choice (read per iteration: tests compare the two).
bool Engine::two_pass_ok() {
    // one GPU, or a row slab stepping phase by phase (phase_spmv* takes the first walk, phase_recur_lazy the second)
    // (with the selective rule on one GPU the recurrence is the same deferred-normalisation one, and the product launched
    // ahead while the host decides is then the first walk of the next iteration)
    const bool mode_ok = reorth == LANCZOS_HIP_REORTH_NONE || (reorth == LANCZOS_HIP_REORTH_SELECTIVE && lazy);
    if (!(lazy || lazy_slab) || !mode_ok || x_ext || ext_matvec || nnz == 0 || spmv_kind != LANCZOS_HIP_SPMV_DICT ||
        dict_form() != 2)
        return false;
    const char *env = getenv("LANCZOS_HIP_TWO_PASS");
    if (env && *env) return atoi(env) != 0;
    // from 128^3: below, a step is the sum of its launches' fixed costs and the recurrence kernel is the lighter second launch
    // (it/s, product + recurrence against two walks: 96^3 56 500 / 49 700, 128^3 36 900 / 38 700, 160^3 21 900 / 23 300,
    // 256^3 7 100 / 7 950, 512^3 855 / 1 005)
    // and from about a hundred planes: every segment of a walk loads a plane before and a plane after its own, which two
    // walks pay twice (slabs of a 512^3 grid, ms per step, two launches / two walks: 256 planes 0.592 / 0.538, 128 planes
    // 0.306 / 0.301, 64 planes 0.165 / 0.167; of a 256^3 grid: 128 planes 0.0841 / 0.0802, 32 planes 0.0368 / 0.0385)
    const int64_t planes = dict_plane_f > 0 ? (n + dict_plane_f - 1) / dict_plane_f : 0;
    return 8.0 * (double)n >= 16.0 * 1048576.0 && planes >= 100;
}

// first walk: the partial sums of u_kk . A u_kk, nothing written
int Engine::launch_walk1(int kk) {
    ProfScope ps(this, LANCZOS_HIP_K_SPMV, 10.0 * (double)n);  // pattern numbers and x
    return launch_spmv_dict(vec(kk), vec(kk), nullptr, &alpha_count, 0, -1, 1, nullptr, nullptr);
}

// second walk: A u_k again with the recurrence in its epilogue; leaves the partial sums of |u_{k+1}|^2
int Engine::launch_walk2() {
    ProfScope ps(this, LANCZOS_HIP_K_RECUR, 26.0 * (double)n);  // pattern numbers, x, u_{k-1} read, u_{k+1} written
    const DictRecur rc{k > 0 ? vec(k - 1) : vec(k), k, d_partials, alpha_count, norm_parts(k - 1), norm_count,
                       d_alpha,            d_beta, h_hist};
    int blocks = 0;
    if (int r = launch_spmv_dict(vec(k), vec(k), vec(k + 1), &blocks, 0, -1, 2, &rc, norm_parts(k))) return r;
    norm_count = blocks;
    return 0;
}

// One whole iteration on a single GPU: the reductions' last stages are folded
// into the consuming kernels, so no finaliser runs and nothing waits for the host.
int Engine::enqueue_iteration() {
    if (int rc = check_can_step()) return rc;
    const bool walks = !slab && two_pass_ok();
    // unless launched ahead by the previous iteration, on the vector that stood and in the form this iteration takes
    if (spec_spmv_k != k || spec_walk != walks)
        if (int rc = walks ? launch_walk1(k) : launch_product(k)) return rc;
    spec_spmv_k = -1;
    if (lazy) {
        if (walks) {
            if (int rc = launch_walk2()) return rc;
        } else {
            const int blocks = recur_blocks();
            ProfScope ps(this, LANCZOS_HIP_K_RECUR, 32.0 * (double)n);
            hipLaunchKernelGGL(recur_lazy_kernel(), dim3(blocks), dim3(kBlock), 0, stream, (double2 *)vec(k + 1),
                               (const double2 *)vec(k), (const double2 *)(k > 0 ? vec(k - 1) : nullptr), npad / 2, k,
                               (const double *)d_partials, alpha_count, (const double *)norm_parts(k - 1), norm_count,
                               norm_parts(k), d_alpha, d_beta, h_hist);
            norm_count = blocks;
        }
        HIP_TRY(hipGetLastError());
        if (reorth == LANCZOS_HIP_REORTH_SELECTIVE)
            if (int rc = selective_lazy_tail(walks)) return rc;
        k += 1;
        if (h_hist) return flush_lazy();  // a host loop is reading beta_k as soon as the iteration is done
        return 0;
    }
    int l0, l1;
    reorth_range(&l0, &l1);
    // recurrence and the pass's dot products in one sweep over w whenever a blocked pass follows the recurrence
    // (LANCZOS_HIP_FUSE_RECUR=0: two launches; same bits either way)
    const char *fuse_env = getenv("LANCZOS_HIP_FUSE_RECUR");  // read per iteration: tests compare the two forms
    const bool fuse = !(fuse_env && atoi(fuse_env) == 0);
    const bool blocked = reorth == LANCZOS_HIP_REORTH_FULL || reorth == LANCZOS_HIP_REORTH_FULL2 ||
                         reorth == LANCZOS_HIP_REORTH_CUDA_COMPAT;
    // Only while the pass is short: at k = 10 the one sweep saves 0.026 ms of 0.30 (256^3), but with 150 basis
    // vectors behind it the stores of w inside the sweep cost the basis stream 0.15 ms more than k_recur's own
    // pass took (3.18 ms against 2.95 + 0.08); measured break-even near 30 vectors.
    // Vectors that sit in the L2s are another matter: there a step is the sum of its launches (about 7 us each), and one
    // launch less is worth having at any k (fuse_limit).
    if (fuse && blocked && k > 0 && l1 > l0 && l1 - l0 <= fuse_limit()) {
        reorth_steps += 1;
        int nparts = 0;
        if (int rc = launch_recur_dots(d_partials, alpha_count, l0, l1, d_red)) return rc;
        if (int rc = launch_update(l0, l1, d_red, &nparts)) return rc;
        if (reorth == LANCZOS_HIP_REORTH_FULL2) {
            if (int rc = launch_dots(l0, l1, d_red)) return rc;
            if (int rc = launch_update(l0, l1, d_red, &nparts)) return rc;
        }
        return launch_scale(norm_parts(k), norm_count);
    }
    if (int rc = launch_recur(d_partials, alpha_count)) return rc;
    if (reorth == LANCZOS_HIP_REORTH_SELECTIVE) return selective_tail();
    if (l1 > l0) reorth_steps += 1;
    if (reorth == LANCZOS_HIP_REORTH_MGS) {
        // exact order of src/lanczos.c:156-167: one basis vector at a time
        int nparts = 0;
        for (int l = l0; l < l1; ++l) {
            if (int rc = launch_dots(l, l + 1, d_red)) return rc;
            if (int rc = launch_update(l, l + 1, d_red, &nparts)) return rc;
        }
    } else if (l1 > l0) {
        const int passes = reorth == LANCZOS_HIP_REORTH_FULL2 ? 2 : 1;
        int nparts = 0;
        for (int p = 0; p < passes; ++p) {
            if (int rc = launch_dots(l0, l1, d_red)) return rc;
            if (int rc = launch_update(l0, l1, d_red, &nparts)) return rc;
        }
    }
    return launch_scale(norm_parts(k), norm_count);
}

// The rest of an iteration under selective reorthogonalisation.  The host needs
// alpha_k and ||w|| to advance its orthogonality estimates, so the stream is
// waited on once per step (the wait covers one small kernel, not the scaling).
int Engine::selective_tail() {
    {
        ProfScope ps(this, LANCZOS_HIP_K_SMALL, 8.0 * norm_count);
        hipLaunchKernelGGL(k_norm_host, dim3(1), dim3(kBlock), 0, stream, (const double *)norm_parts(k), norm_count,
                           d_scal + 2, h_sel);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sel_event, stream));
    HIP_TRY(hipEventSynchronize(sel_event));
    if (k > 0) lanczos_hip_omega_set_beta(omega, k - 1, h_hist[2 * (size_t)(k - 1) + 1]);
    const int need = lanczos_hip_omega_step(omega, k, h_hist[2 * (size_t)k], h_sel[0], nullptr);
    if (need < 0) return fail("lanczos_hip: orthogonality tracker out of step at iteration %d", k);
    if (!need) return launch_scale(d_scal + 2, 1);
    reorth_steps += 1;
    int nparts = 0;
    if (int rc = launch_dots(0, k + 1, d_red)) return rc;
    if (int rc = launch_update(0, k + 1, d_red, &nparts)) return rc;
    return launch_scale(norm_parts(k), norm_count);
}

// The same decision on one GPU, where the basis is kept unnormalised (u_j = beta_{j-1} v_j, no
// scaling pass).  While the host waits for ||w|| the next product is already running on the vector
// as it stands; in the few steps that do reorthogonalise, that product is taken again.
int Engine::selective_lazy_tail(bool walks) {
    {
        ProfScope ps(this, LANCZOS_HIP_K_SMALL, 8.0 * norm_count);
        hipLaunchKernelGGL(k_norm_host, dim3(1), dim3(kBlock), 0, stream, (const double *)norm_parts(k), norm_count,
                           d_scal + 2, h_sel);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sel_event, stream));
    if (!ext_matvec && k + 2 < nvec && k + 1 < hist_cap && k + 1 < max_iter - 1) {
        if (int rc = walks ? launch_walk1(k + 1) : launch_spmv(vec(k + 1), vec(k + 1), vec(k + 2), &alpha_count)) return rc;
        spec_spmv_k = k + 1;
        spec_walk = walks;
    }
    HIP_TRY(hipEventSynchronize(sel_event));
    if (k > 0) lanczos_hip_omega_set_beta(omega, k - 1, h_hist[2 * (size_t)(k - 1) + 1]);
    const int need = lanczos_hip_omega_step(omega, k, h_hist[2 * (size_t)k], h_sel[0], nullptr);
    if (need < 0) return fail("lanczos_hip: orthogonality tracker out of step at iteration %d", k);
    if (!need) return 0;
    reorth_steps += 1;
    int nparts = 0;
    if (int rc = launch_dots(0, k + 1, d_red)) return rc;
    {
        ProfScope ps(this, LANCZOS_HIP_K_SMALL, 16.0 * (k + 1));
        hipLaunchKernelGGL(k_lazy_coeff, dim3((k + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_red, k + 1,
                           (const double *)d_beta);
    }
    if (int rc = launch_update(0, k + 1, d_red, &nparts)) return rc;
    spec_spmv_k = -1;  // the vector changed under the product launched ahead
    return 0;
}

// out_dev[i * out_ld + .] = sum_{l < m} coef_host[i * m + l] * v_l for i < count: eight at a time,
// one pass over the basis each.
int Engine::combine(int m, int count, const double *coef_host, double *out_dev, int64_t out_ld) {
    constexpr int J = 8;
    double *d_coef = nullptr;
    HIP_TRY(hipMalloc(&d_coef, sizeof(double) * (size_t)m * J));
    std::vector<double> c((size_t)m * J);
    int rc = 0;
    for (int i0 = 0; i0 < count && rc == 0; i0 += J) {
        const int nj = std::min(J, count - i0);
        for (int l = 0; l < m; ++l)
            for (int j = 0; j < J; ++j) c[(size_t)l * J + j] = j < nj ? coef_host[(size_t)(i0 + j) * m + l] : 0.0;
        hipError_t st = hipMemcpyAsync(d_coef, c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice, stream);
        if (st == hipSuccess) {
            ProfScope ps(this, LANCZOS_HIP_K_UPDATE, 8.0 * (double)n * (m + nj));
            hipLaunchKernelGGL((k_combine<J>), dim3(tile_blocks()), dim3(kBlock), 0, stream, (const double2 *)d_V,
                               ldv / 2, m, (const double *)d_coef, (double2 *)(out_dev + (size_t)i0 * out_ld),
                               out_ld / 2, nj, npad / 2);
            st = hipGetLastError();
        }
        if (st == hipSuccess) st = hipStreamSynchronize(stream);  // c is reused by the next group
        if (st != hipSuccess) rc = fail("lanczos_hip: combine failed: %s", hipGetErrorString(st));
    }
    (void)hipFree(d_coef);
    return rc;
}

// Thick restart: the basis v_0..v_k becomes y_0..y_{p-1}, v_k with y_i = sum_l coef[i][l] v_l, and
// the run continues at iteration p.  The next recurrence subtracts last_coupling * y_{p-1} where it
// would subtract beta * v_{k-1}; the couplings to the other y_i are removed by the
// reorthogonalisation, which therefore must cover every l < k.
int Engine::restart(int p, const double *coef_host, double last_coupling) {
    if (!started) return fail("lanczos_hip_restart: no run to restart");
    if (reorth != LANCZOS_HIP_REORTH_FULL && reorth != LANCZOS_HIP_REORTH_FULL2 && reorth != LANCZOS_HIP_REORTH_MGS)
        return fail("lanczos_hip_restart: needs reorthogonalisation against the whole basis (full, full2 or mgs)");
    if (slab) return fail("lanczos_hip_restart: not available on a row-partitioned engine");
    if (p < 1 || p >= k || !coef_host) return fail("lanczos_hip_restart: keep=%d of %d iterations", p, k);
    HIP_TRY(hipSetDevice(device));
    double *d_tmp = nullptr;
    HIP_TRY(hipMalloc(&d_tmp, sizeof(double) * (size_t)ldv * (size_t)p));
    int rc = combine(k, p, coef_host, d_tmp, ldv);
    hipError_t st = hipSuccess;
    if (rc == 0) st = hipMemcpyAsync(d_V, d_tmp, sizeof(double) * (size_t)ldv * (size_t)p, hipMemcpyDeviceToDevice, stream);
    if (rc == 0 && st == hipSuccess)
        st = hipMemcpyAsync(vec(p), vec(k), sizeof(double) * (size_t)ldv, hipMemcpyDeviceToDevice, stream);
    if (rc == 0 && st == hipSuccess)
        st = hipMemcpyAsync(d_scal + 1, &last_coupling, sizeof(double), hipMemcpyHostToDevice, stream);
    if (rc == 0 && st == hipSuccess) st = hipStreamSynchronize(stream);
    (void)hipFree(d_tmp);
    if (rc) return rc;
    if (st != hipSuccess) return fail("lanczos_hip_restart: %s", hipGetErrorString(st));
    harvest_profile();
    k = p;
    return 0;
}

int Engine::run(int steps, double *alpha_out, double *beta_out) {
    if (steps < 0) return fail("lanczos_hip_run: steps=%d", steps);
    HIP_TRY(hipSetDevice(device));
    const int k0 = k;
    for (int s = 0; s < steps; ++s)
        if (int rc = enqueue_iteration()) return rc;
    if (int rc = flush_lazy()) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    harvest_profile();
    if (alpha_out && steps)
        HIP_TRY(hipMemcpy(alpha_out, d_alpha + k0, sizeof(double) * steps, hipMemcpyDeviceToHost));
    if (beta_out && steps)
        HIP_TRY(hipMemcpy(beta_out, d_beta + k0, sizeof(double) * steps, hipMemcpyDeviceToHost));
    return 0;
}

int Engine::spmv_host(const double *x_host, double *y_host) {
    if (!x_host || !y_host) return fail("lanczos_hip_spmv: NULL argument");
    HIP_TRY(hipSetDevice(device));
    double *dx = nullptr, *dy = nullptr;
    HIP_TRY(hipMalloc(&dx, sizeof(double) * n_ext));
    HIP_TRY(hipMalloc(&dy, sizeof(double) * n));
    HIP_TRY(hipMemcpyAsync(dx, x_host, sizeof(double) * n_ext, hipMemcpyHostToDevice, stream));
    int nparts = 0;
    // xdot only feeds the alpha partial; the first n entries of x will do here
    int rc = launch_spmv(dx, dx, dy, &nparts);
    if (rc == 0) {
        hipError_t st = hipMemcpyAsync(y_host, dy, sizeof(double) * n, hipMemcpyDeviceToHost, stream);
        if (st == hipSuccess) st = hipStreamSynchronize(stream);
        if (st != hipSuccess) rc = fail("lanczos_hip_spmv: %s", hipGetErrorString(st));
    }
    harvest_profile();
    (void)hipFree(dx);
    (void)hipFree(dy);
    return rc;
}

int Engine::make_current() {
    HIP_TRY(hipSetDevice(device));
    tl_current_slab = affinity_id;
    return 0;
}

int Engine::record_event(hipEvent_t ev) {
    if (int rc = check_affinity(this, "record_event")) return rc;
    HIP_TRY(hipEventRecord(ev, stream));
    return 0;
}

int Engine::enable_host_history() {
    if (h_hist) return 0;
    if (int rc = check_affinity(this, "enable_host_history")) return rc;
    HIP_TRY(hipHostMalloc((void **)&h_hist, sizeof(double) * 2 * (size_t)hist_cap, hipHostMallocDefault));
    memset(h_hist, 0, sizeof(double) * 2 * (size_t)hist_cap);
    return 0;
}

}  // namespace lz

// ---------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------
using lz::Engine;

static Engine *E(lanczos_hip_engine *e) { return reinterpret_cast<Engine *>(e); }
static const Engine *E(const lanczos_hip_engine *e) { return reinterpret_cast<const Engine *>(e); }

extern "C" {

int lanczos_hip_create(lanczos_hip_engine **out, int64_t n, int64_t nnz, const int64_t *row_head_indexes,
                       const int64_t *column_index, const double *values, int max_iter,
                       const lanczos_hip_options *opt) {
    return lz::guarded("lanczos_hip_create", [&]() -> int {
    if (!out) return lz::fail("lanczos_hip_create: NULL out");
    Engine *e = nullptr;
    int rc = Engine::create(&e, n, nnz, row_head_indexes, column_index, values, max_iter, opt);
    *out = reinterpret_cast<lanczos_hip_engine *>(e);
    return rc;
    });
}

void lanczos_hip_destroy(lanczos_hip_engine *e) { delete E(e); }

int lanczos_hip_set_start(lanczos_hip_engine *e, const double *v0_host) {
    if (!e) return lz::fail("lanczos_hip_set_start: NULL engine");
    return E(e)->set_start(v0_host, true, 1.0);
}

int lanczos_hip_set_start_scaled(lanczos_hip_engine *e, const double *v0_host, double scale) {
    if (!e) return lz::fail("lanczos_hip_set_start_scaled: NULL engine");
    return E(e)->set_start(v0_host, false, scale);
}

int lanczos_hip_set_start_seed(lanczos_hip_engine *e, uint64_t seed) {
    return lz::guarded("lanczos_hip_set_start_seed", [&]() -> int {
    if (!e) return lz::fail("lanczos_hip_set_start_seed: NULL engine");
    Engine *g = E(e);
    if (g->slab) return lz::fail("lanczos_hip_set_start_seed: row-partitioned engines need a global norm; use lanczos_hip_gaussian_fill + lanczos_hip_set_start_scaled");
    std::vector<double> v((size_t)g->n);
    lanczos_hip_gaussian_fill(seed, g->row_begin, g->n, v.data());
    return g->set_start(v.data(), true, 1.0);
    });
}

int lanczos_hip_run(lanczos_hip_engine *e, int steps, double *alpha_out, double *beta_out) {
    return lz::guarded("lanczos_hip_run", [&]() -> int {
    if (!e) return lz::fail("lanczos_hip_run: NULL engine");
    if (E(e)->slab) return lz::fail("lanczos_hip_run: row-partitioned engines are driven phase by phase");
    return E(e)->run(steps, alpha_out, beta_out);
    });
}

int lanczos_hip_set_operator(lanczos_hip_engine *e, lanczos_hip_matvec_fn *matvec, const void *mat) {
    if (!e) return lz::fail("lanczos_hip_set_operator: NULL engine");
    return E(e)->set_operator(matvec, mat);
}

int lanczos_hip_iterations(const lanczos_hip_engine *e) { return e ? E(e)->k : 0; }

int lanczos_hip_reorth_steps(const lanczos_hip_engine *e) { return e ? E(e)->reorth_steps : 0; }

int lanczos_hip_spmv(lanczos_hip_engine *e, const double *x_host, double *y_host) {
    return lz::guarded("lanczos_hip_spmv", [&]() -> int {
    if (!e) return lz::fail("lanczos_hip_spmv: NULL engine");
    return E(e)->spmv_host(x_host, y_host);
    });
}

int lanczos_hip_get_vector(lanczos_hip_engine *e, int k, double *out_host) {
    if (!e || !out_host) return lz::fail("lanczos_hip_get_vector: NULL argument");
    Engine *g = E(e);
    if (k < 0 || k > g->k || (g->reorth == LANCZOS_HIP_REORTH_NONE && k + 2 < g->k))
        return lz::fail("lanczos_hip_get_vector: vector %d is not resident (iteration %d)", k, g->k);
    if (hipSetDevice(g->device) != hipSuccess) return lz::fail("lanczos_hip_get_vector: hipSetDevice failed");
    if (g->flush_lazy() != 0) return EXIT_FAILURE;
    hipError_t st = hipMemcpyAsync(out_host, g->vec(k), sizeof(double) * g->n, hipMemcpyDeviceToHost, g->stream);
    double beta_prev = 1.0;
    if (st == hipSuccess && g->lazy && k > 0)
        st = hipMemcpyAsync(&beta_prev, g->d_beta + (k - 1), sizeof(double), hipMemcpyDeviceToHost, g->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(g->stream);
    if (st != hipSuccess) return lz::fail("lanczos_hip_get_vector: %s", hipGetErrorString(st));
    if (g->lazy && k > 0)  // the three-vector mode keeps u_k = beta_{k-1} v_k
        for (int64_t i = 0; i < g->n; ++i) out_host[i] /= beta_prev;
    return 0;
}

int lanczos_hip_profile(const lanczos_hip_engine *e, int cls, double *ms, int64_t *launches, double *bytes) {
    if (!e || cls < 0 || cls >= LANCZOS_HIP_K_COUNT) return lz::fail("lanczos_hip_profile: bad argument");
    if (ms) *ms = E(e)->prof_ms[cls];
    if (launches) *launches = E(e)->prof_launches[cls];
    if (bytes) *bytes = E(e)->prof_bytes[cls];
    return 0;
}

void lanczos_hip_profile_reset(lanczos_hip_engine *e) {
    if (!e) return;
    Engine *g = E(e);
    g->harvest_profile();
    for (int i = 0; i < LANCZOS_HIP_K_COUNT; ++i) {
        g->prof_ms[i] = 0;
        g->prof_launches[i] = 0;
        g->prof_bytes[i] = 0;
    }
}

int lanczos_hip_profile_select(lanczos_hip_engine *e, unsigned class_mask) {
    if (!e) return lz::fail("lanczos_hip_profile_select: NULL engine");
    E(e)->prof_mask = class_mask;
    return 0;
}

int lanczos_hip_ritz_vectors(lanczos_hip_engine *eh, int steps, int m, double *eig, double *vectors_host,
                             double *residual_est) {
    return lz::guarded("lanczos_hip_ritz_vectors", [&]() -> int {
    if (!eh || !eig || steps < 1 || m < 1) return lz::fail("lanczos_hip_ritz_vectors: bad argument");
    Engine *e = E(eh);
    if (e->reorth == LANCZOS_HIP_REORTH_NONE || e->slab)
        return lz::fail("lanczos_hip_ritz_vectors: needs the whole basis resident on one GPU (not reorth=none, not a row slab)");
    if (steps > e->k) return lz::fail("lanczos_hip_ritz_vectors: only %d iterations done", e->k);
    if (m > steps) m = steps;
    if (hipSetDevice(e->device) != hipSuccess) return lz::fail("lanczos_hip_ritz_vectors: hipSetDevice failed");
    std::vector<double> a((size_t)steps), b((size_t)steps), S((size_t)m * steps);
    hipError_t st = hipStreamSynchronize(e->stream);
    if (st == hipSuccess) st = hipMemcpy(a.data(), e->d_alpha, sizeof(double) * steps, hipMemcpyDeviceToHost);
    if (st == hipSuccess) st = hipMemcpy(b.data(), e->d_beta, sizeof(double) * steps, hipMemcpyDeviceToHost);
    if (st != hipSuccess) return lz::fail("lanczos_hip_ritz_vectors: %s", hipGetErrorString(st));
    if (lanczos_hip_tridiag_eigs(a.data(), b.data(), steps, m, eig) != 0 ||
        lanczos_hip_tridiag_eigvecs(a.data(), b.data(), steps, m, eig, S.data()) != 0)
        return lz::fail("lanczos_hip_ritz_vectors: tridiagonal eigen-solve failed");
    if (residual_est)  // |beta_{steps-1} * last component|: the Lanczos bound on ||A y - theta y||
        for (int i = 0; i < m; ++i) residual_est[i] = fabs(b[(size_t)steps - 1] * S[(size_t)i * steps + steps - 1]);
    if (!vectors_host) return 0;
    if (e->lazy)  // the resident basis is u_j = beta_{j-1} v_j
        for (int i = 0; i < m; ++i)
            for (int j = 1; j < steps; ++j) S[(size_t)i * steps + j] /= b[(size_t)j - 1];
    // y_i = sum_j S[i][j] v_j, eight at a time with one pass over the basis each
    double *d_y = nullptr;
    if (hipMalloc(&d_y, sizeof(double) * (size_t)e->ldv * (size_t)m) != hipSuccess)
        return lz::fail("lanczos_hip_ritz_vectors: out of device memory");
    int rc = e->combine(steps, m, S.data(), d_y, e->ldv);
    if (rc == 0 && hipMemcpy2D(vectors_host, sizeof(double) * e->n, d_y, sizeof(double) * e->ldv, sizeof(double) * e->n,
                               (size_t)m, hipMemcpyDeviceToHost) != hipSuccess)
        rc = lz::fail("lanczos_hip_ritz_vectors: copy to the host failed");
    (void)hipFree(d_y);
    return rc;
    });
}

int lanczos_hip_restart(lanczos_hip_engine *e, int keep, const double *coef_host, double last_coupling) {
    return lz::guarded("lanczos_hip_restart", [&]() -> int {
    if (!e) return lz::fail("lanczos_hip_restart: NULL handle");
    return E(e)->restart(keep, coef_host, last_coupling);
    });
}

const char *lanczos_hip_spmv_name(const lanczos_hip_engine *e) { return e ? E(e)->spmv_name : ""; }

const char *lanczos_hip_last_error(void) { return lz::last_error(); }

int lanczos_hip_phase_spmv(lanczos_hip_engine *e, double *red) {
    if (!e || !red) return lz::fail("lanczos_hip_phase_spmv: NULL argument");
    return E(e)->phase_spmv(red);
}
int lanczos_hip_phase_recur(lanczos_hip_engine *e, double *red, int *n_red, int *is_norm) {
    if (!e || !red || !n_red || !is_norm) return lz::fail("lanczos_hip_phase_recur: NULL argument");
    return E(e)->phase_recur(red, n_red, is_norm);
}
int lanczos_hip_phase_update(lanczos_hip_engine *e, double *red) {
    if (!e || !red) return lz::fail("lanczos_hip_phase_update: NULL argument");
    return E(e)->phase_update(red);
}
int lanczos_hip_phase_scale(lanczos_hip_engine *e, double *red) {
    if (!e || !red) return lz::fail("lanczos_hip_phase_scale: NULL argument");
    return E(e)->phase_scale(red);
}
int lanczos_hip_phase_dots(lanczos_hip_engine *e, double *red, int *n_red) {
    if (!e || !red || !n_red) return lz::fail("lanczos_hip_phase_dots: NULL argument");
    return E(e)->phase_dots(red, n_red);
}
int lanczos_hip_phase_spmv_begin(lanczos_hip_engine *e, int64_t first_interior_row, int64_t end_interior_row) {
    if (!e) return lz::fail("lanczos_hip_phase_spmv_begin: NULL engine");
    return E(e)->phase_spmv_begin(first_interior_row, end_interior_row);
}
int lanczos_hip_phase_spmv_end(lanczos_hip_engine *e, double *red) {
    if (!e || !red) return lz::fail("lanczos_hip_phase_spmv_end: NULL argument");
    return E(e)->phase_spmv_end(red);
}
int lanczos_hip_phase_spmv_can_overlap(const lanczos_hip_engine *e) {
    if (!e) return 0;
    return E(e)->can_split_product();
}
int lanczos_hip_phase_recur_lazy(lanczos_hip_engine *e, double *red) {
    if (!e || !red) return lz::fail("lanczos_hip_phase_recur_lazy: NULL argument");
    return E(e)->phase_recur_lazy(red);
}
int lanczos_hip_phase_lazy_norm(lanczos_hip_engine *e, double *red) {
    if (!e || !red) return lz::fail("lanczos_hip_phase_lazy_norm: NULL argument");
    return E(e)->phase_lazy_norm(red);
}
int lanczos_hip_phase_lazy_close(lanczos_hip_engine *e, double *red) {
    if (!e || !red) return lz::fail("lanczos_hip_phase_lazy_close: NULL argument");
    return E(e)->phase_lazy_close(red);
}
int64_t lanczos_hip_gather_base(int64_t n) { return lz::round_up(n, 2 * lz::kTileV2); }
int lanczos_hip_export_vector(lanczos_hip_engine *e, int k, double *dst_device) {
    if (!e || !dst_device || k < 0) return lz::fail("lanczos_hip_export_vector: bad argument");
    Engine *g = E(e);
    if (hipMemcpyAsync(dst_device, g->vec(k), sizeof(double) * g->n, hipMemcpyDeviceToDevice, g->stream) != hipSuccess)
        return lz::fail("lanczos_hip_export_vector: copy failed");
    return 0;
}
double *lanczos_hip_local_vector(lanczos_hip_engine *e, int k) {
    if (!e || k < 0) return nullptr;
    return E(e)->vec(k);
}
int lanczos_hip_history(lanczos_hip_engine *e, double *alpha_out, double *beta_out, int count) {
    if (!e) return lz::fail("lanczos_hip_history: NULL engine");
    Engine *g = E(e);
    if (count < 0 || count > g->k) return lz::fail("lanczos_hip_history: count=%d of %d", count, g->k);
    if (hipSetDevice(g->device) != hipSuccess) return lz::fail("lanczos_hip_history: hipSetDevice failed");
    if (g->flush_lazy() != 0) return EXIT_FAILURE;
    hipError_t st = hipStreamSynchronize(g->stream);
    g->harvest_profile();
    if (st == hipSuccess && alpha_out && count)
        st = hipMemcpy(alpha_out, g->d_alpha, sizeof(double) * count, hipMemcpyDeviceToHost);
    if (st == hipSuccess && beta_out && count)
        st = hipMemcpy(beta_out, g->d_beta, sizeof(double) * count, hipMemcpyDeviceToHost);
    if (st != hipSuccess) return lz::fail("lanczos_hip_history: %s", hipGetErrorString(st));
    return 0;
}

}  // extern "C"
