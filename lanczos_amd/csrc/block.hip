// Block-Lanczos variant (block width 16) with full reorthogonalisation.
//
// The reference iterates on one vector (src/lanczos.c:104-191); it has no block
// form.  This variant carries 16 vectors at a time so that the
// reorthogonalisation against the Krylov basis becomes a tall-skinny dense
// contraction for the fp64 matrix cores, and the sparse product gathers whole
// 128-byte rows.  One block step j (W = slot j + 1 of the basis):
//     W  = A Q_j                                   k_spmm16
//     H  = [Q_{j-1} Q_j]^T W ; A_j = sym(H_j)      k_gram16, k_gram_reduce
//     W -= [Q_{j-1} Q_j] H                         k_bupdate16   (three-term part)
//     H  = [Q_0 .. Q_j]^T W ; W -= [Q_0 .. Q_j] H  k_gram16, k_bupdate16 (full pass)
//     W  = Q_{j+1} B_j by CholQR, twice            k_gram16, k_chol16, k_bscale16
// T is block tridiagonal with A_j on and B_j below the diagonal.  Everything
// stays on the device and on one stream; the host reads T at the end.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "block_kernels.hip.h"
#include "engine.h"
#include "spmm_hub.h"

#define HIP_TRY(call)                                                                             \
    do {                                                                                          \
        hipError_t st__ = (call);                                                                 \
        if (st__ != hipSuccess)                                                                   \
            return lz::fail("HIP API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,    \
                            hipGetErrorString(st__), (int)st__);                                  \
    } while (0)

namespace {

using namespace lzb;

struct BlockEngine {
    lz::Engine *base = nullptr;  // owns the device matrix, the stream and the device choice
    int64_t n = 0, n_pad = 0, blk_stride = 0;
    int max_steps = 0;
    int steps = 0;
    bool started = false;
    double *d_Q = nullptr;      // [max_steps + 1][n_pad][16]
    double *d_hpart = nullptr;  // [groups][hstride]
    int64_t hstride = 0;
    int groups_gram = 0, groups_update = 0, groups_spmm = 0;
    double *d_H = nullptr;      // [max_steps + 1][256]
    double *d_small = nullptr;  // G, Rinv, R1, R2: 4 x 256
    double *d_TA = nullptr, *d_TB = nullptr;
    int *d_status = nullptr;
    int *d_med_rows = nullptr, *d_long_rows = nullptr;  // rows over kMedRow / over kLongRow entries
    long long *d_xcd_split = nullptr;                   // [9] row octets of the eight XCDs' ranges (k_spmm16)
    int n_med = 0, n_long = 0;
    int med_row = kMedRow;      // rows of more entries leave k_spmm16
    HubStream hub;              // those rows' entries regrouped by column class (power-law graphs); empty: row-wise kernels
    int64_t hot_degree_floor = 0;
    bool profile = false;
    double prof_ms[LANCZOS_HIP_BK_COUNT] = {0};
    double prof_bytes[LANCZOS_HIP_BK_COUNT] = {0};
    int64_t prof_launches[LANCZOS_HIP_BK_COUNT] = {0};
    struct Pending { int cls; hipEvent_t a, b; };
    std::vector<Pending> pending;

    double *slot(int t) const { return d_Q + (size_t)t * (size_t)blk_stride; }
    hipStream_t stream() const { return base->stream; }

    ~BlockEngine() {
        if (base) {
            (void)hipSetDevice(base->device);
            (void)hipStreamSynchronize(base->stream);
        }
        harvest();
        (void)hipFree(d_Q);
        (void)hipFree(d_hpart);
        (void)hipFree(d_H);
        (void)hipFree(d_small);
        (void)hipFree(d_TA);
        (void)hipFree(d_TB);
        (void)hipFree(d_status);
        (void)hipFree(d_long_rows);
        (void)hipFree(d_med_rows);
        (void)hipFree(d_xcd_split);
        hub_free(&hub);
        delete base;
    }

    void harvest() {
        for (auto &p : pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) prof_ms[p.cls] += ms;
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        pending.clear();
    }

    struct Scope {
        BlockEngine *e;
        Scope(BlockEngine *eng, int cls, double bytes) : e(eng) {
            if (!e->profile) return;
            e->prof_bytes[cls] += bytes;
            e->prof_launches[cls] += 1;
            Pending p{cls, nullptr, nullptr};
            (void)hipEventCreate(&p.a);
            (void)hipEventCreate(&p.b);
            (void)hipEventRecord(p.a, e->stream());
            e->pending.push_back(p);
        }
        ~Scope() {
            if (e->profile) (void)hipEventRecord(e->pending.back().b, e->stream());
        }
    };

    int alloc(int max_steps_, bool profile_) {
        n = base->n;
        n_pad = (n + 127) / 128 * 128;
        blk_stride = n_pad * kW;
        max_steps = max_steps_;
        profile = profile_;
        const int cus = base->cus;
        groups_spmm = cus * 8;
        groups_update = cus * 4;
        groups_gram = cus * 2;
        const int64_t tiles4 = (n_pad / kTileRows + 3) / 4;
        groups_update = (int)std::min<int64_t>(groups_update, tiles4);
        groups_gram = (int)std::min<int64_t>(groups_gram, tiles4);
        hstride = (int64_t)(max_steps + 1) * 256;
        const size_t qbytes = sizeof(double) * (size_t)blk_stride * (size_t)(max_steps + 1);
        HIP_TRY(hipMalloc(&d_Q, qbytes));
        HIP_TRY(hipMemsetAsync(d_Q, 0, qbytes, stream()));
        HIP_TRY(hipMalloc(&d_hpart, sizeof(double) * (size_t)groups_gram * (size_t)hstride));
        HIP_TRY(hipMalloc(&d_H, sizeof(double) * (size_t)hstride));
        HIP_TRY(hipMalloc(&d_small, sizeof(double) * 4 * 256));
        HIP_TRY(hipMalloc(&d_TA, sizeof(double) * 256 * (size_t)max_steps));
        HIP_TRY(hipMalloc(&d_TB, sizeof(double) * 256 * (size_t)max_steps));
        HIP_TRY(hipMalloc(&d_status, sizeof(int) * 4));
        HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int) * 4, stream()));
        // rows too long for eight lanes get a wavefront or a workgroup each
        if (int rc = find_long_rows()) return rc;
        HIP_TRY(hipMalloc(&d_xcd_split, sizeof(long long) * 9));
        // Rows of more than kMedRow entries holding a good part of the matrix (a power-law graph): their entries are
        // regrouped by column class so that the eight L2s cache disjoint parts of X (spmm_hub.hip).  A handful of
        // long rows in an otherwise regular matrix keep the row-wise kernels, and so does LANCZOS_HIP_SPMM_HUB=0.
        const char *hub_env = getenv("LANCZOS_HIP_SPMM_HUB");
        const bool hub_off = hub_env && atoi(hub_env) == 0, hub_forced = hub_env && atoi(hub_env) == 2;
        if (!hub_off && n_med + n_long > 0 && (hub_forced || (int64_t)(n_med + n_long) * 64 > n / 16)) {
            // the regrouped stream takes every row of more than kHubRow entries (LANCZOS_HIP_SPMM_HUB_ROW: 4..64)
            int keep = kHubRow;
            if (const char *s = getenv("LANCZOS_HIP_SPMM_HUB_ROW"))
                if (atoi(s) >= 4 && atoi(s) <= 64) keep = atoi(s);
            med_row = keep;
            if (int rc = find_long_rows()) return rc;
            if (int rc = hub_build(base, d_med_rows, n_med, d_long_rows, n_long, hot_degree_floor, &hub)) return rc;
            if (!hub.ready()) {  // no memory for it: back to the row-wise kernels and their threshold
                med_row = kMedRow;
                if (int rc = find_long_rows()) return rc;
            }
        }
        return row_cost_split(base, med_row, 8, n_pad / 8, d_xcd_split);
    }

    // d_med_rows / d_long_rows: the rows of more than med_row / kLongRow entries
    int find_long_rows() {
        (void)hipFree(d_med_rows);
        (void)hipFree(d_long_rows);
        d_med_rows = d_long_rows = nullptr;
        // at most nnz / threshold rows can qualify
        const int med_cap = (int)std::min<int64_t>(n, base->nnz / med_row + 1);
        const int lng_cap = (int)std::min<int64_t>(n, base->nnz / kLongRow + 1);
        HIP_TRY(hipMalloc(&d_med_rows, sizeof(int) * (size_t)med_cap));
        HIP_TRY(hipMalloc(&d_long_rows, sizeof(int) * (size_t)lng_cap));
        HIP_TRY(hipMemsetAsync(d_status + 1, 0, sizeof(int) * 2, stream()));
        const int fb = (int)std::min<int64_t>(groups_spmm, (n + kThreads - 1) / kThreads);
        if (base->rp64)
            hipLaunchKernelGGL((k_find_long_rows<int64_t>), dim3(fb), dim3(kThreads), 0, stream(), n,
                               (const int64_t *)base->d_rp, med_row, d_med_rows, med_cap, d_long_rows, lng_cap,
                               d_status + 1);
        else
            hipLaunchKernelGGL((k_find_long_rows<int32_t>), dim3(fb), dim3(kThreads), 0, stream(), n,
                               (const int32_t *)base->d_rp, med_row, d_med_rows, med_cap, d_long_rows, lng_cap,
                               d_status + 1);
        HIP_TRY(hipGetLastError());
        int counts[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(counts, d_status + 1, sizeof(counts), hipMemcpyDeviceToHost, stream()));
        HIP_TRY(hipStreamSynchronize(stream()));
        n_med = counts[0];
        n_long = counts[1];
        if (n_med > med_cap || n_long > lng_cap)
            return lz::fail("lanczos_hip_block_create: long-row list overflow (%d > %d or %d > %d)", n_med, med_cap,
                            n_long, lng_cap);
        return 0;
    }

    double spmm_bytes() const {
        return 12.0 * (double)base->nnz + (base->rp64 ? 8.0 : 4.0) * (double)(n + 1) + 2.0 * 8.0 * kW * (double)n;
    }

    int spmm(const double *X, double *Y) {
        Scope ps(this, LANCZOS_HIP_BK_SPMM, spmm_bytes());
        if (base->rp64)
            hipLaunchKernelGGL((k_spmm16<int64_t>), dim3(groups_spmm), dim3(kThreads), 0, stream(), n, n_pad,
                               (const int64_t *)base->d_rp, base->d_col, base->d_val, X, Y, (const long long *)d_xcd_split, med_row);
        else
            hipLaunchKernelGGL((k_spmm16<int32_t>), dim3(groups_spmm), dim3(kThreads), 0, stream(), n, n_pad,
                               (const int32_t *)base->d_rp, base->d_col, base->d_val, X, Y, (const long long *)d_xcd_split, med_row);
        if (hub.ready()) {
            if (int rc = hub_spmm(hub, stream(), base->cus, X, Y)) return rc;
            return 0;
        }
        if (n_med > 0) {
            const int mb = (n_med + 3) / 4;
            if (base->rp64)
                hipLaunchKernelGGL((k_spmm16_medium<int64_t>), dim3(mb), dim3(kThreads), 0, stream(),
                                   (const int *)d_med_rows, n_med, (const int64_t *)base->d_rp, base->d_col,
                                   base->d_val, X, Y);
            else
                hipLaunchKernelGGL((k_spmm16_medium<int32_t>), dim3(mb), dim3(kThreads), 0, stream(),
                                   (const int *)d_med_rows, n_med, (const int32_t *)base->d_rp, base->d_col,
                                   base->d_val, X, Y);
        }
        if (n_long > 0) {
            if (base->rp64)
                hipLaunchKernelGGL((k_spmm16_long<int64_t>), dim3(n_long), dim3(kThreads), 0, stream(),
                                   (const int *)d_long_rows, (const int64_t *)base->d_rp, base->d_col, base->d_val, X, Y);
            else
                hipLaunchKernelGGL((k_spmm16_long<int32_t>), dim3(n_long), dim3(kThreads), 0, stream(),
                                   (const int *)d_long_rows, (const int32_t *)base->d_rp, base->d_col, base->d_val, X, Y);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // Hout[t - t0] = Q_t^T W, t0 <= t < t1
    int gram(int t0, int t1, const double *W, double *Hout) {
        const int nb = t1 - t0;
        {
            const int segs = (nb + kGramSeg - 1) / kGramSeg;  // W is read once per segment
            Scope ps(this, LANCZOS_HIP_BK_GRAM, 8.0 * kW * (double)n * (nb + segs));
            hipLaunchKernelGGL(k_gram16, dim3(groups_gram), dim3(kThreads), 0, stream(), (const double *)d_Q,
                               blk_stride, t0, t1, W, n_pad, d_hpart, hstride);
        }
        {
            const int count = nb * 256;
            Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * (double)count * groups_gram);
            hipLaunchKernelGGL(k_gram_reduce, dim3((count + kThreads - 1) / kThreads), dim3(kThreads), 0, stream(),
                               (const double *)d_hpart, groups_gram, hstride, count, Hout);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // W -= sum_t Q_t H_t; with gram_out the kernel also leaves W^T W of the result there (no extra pass)
    int update(int t0, int t1, double *W, const double *H, double *gram_out) {
        {
            Scope ps(this, LANCZOS_HIP_BK_UPDATE, 8.0 * kW * (double)n * (t1 - t0 + 2));
            hipLaunchKernelGGL(k_bupdate16, dim3(groups_update), dim3(kThreads), 0, stream(), (const double *)d_Q,
                               blk_stride, t0, t1, W, n_pad, H, gram_out ? d_hpart : (double *)nullptr);
        }
        if (gram_out) reduce_own_gram(gram_out);
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // G = sum over the update / scale workgroups' parts (one 16 x 16 block, fragment order in)
    void reduce_own_gram(double *G) {
        Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * 256 * groups_update);
        hipLaunchKernelGGL(k_gram_reduce, dim3(1), dim3(kThreads), 0, stream(), (const double *)d_hpart, groups_update,
                           (int64_t)256, 256, G);
    }

    // slot t <- orthonormal basis of its columns; Bout (optional) = the factor: old = new * Bout
    // have_gram: d_small already holds W^T W of the block (left by the update that produced it)
    int cholqr2(int t, double *Bout, bool have_gram) {
        double *G = d_small, *Rinv = d_small + 256, *R1 = d_small + 512, *R2 = d_small + 768;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 0 && !have_gram)
                if (int rc = gram(t, t + 1, slot(t), G)) return rc;
            {
                Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * 256 * 3);
                hipLaunchKernelGGL(k_chol16, dim3(1), dim3(kThreads), 0, stream(), (const double *)G, Rinv,
                                   pass == 0 ? R1 : R2, pass == 0 ? (const double *)nullptr : (const double *)R1,
                                   pass == 0 ? (double *)nullptr : Bout, d_status, t + 1);
            }
            {
                Scope ps(this, LANCZOS_HIP_BK_SCALE, 2.0 * 8.0 * kW * (double)n);
                // the first scaling leaves the Gram matrix of its result for the second factorisation
                hipLaunchKernelGGL(k_bscale16, dim3(groups_update), dim3(kThreads), 0, stream(), slot(t), n_pad,
                                   (const double *)Rinv, pass == 0 ? d_hpart : (double *)nullptr);
            }
            if (pass == 0) reduce_own_gram(G);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }

    int set_start(const double *x0_host) {
        HIP_TRY(hipSetDevice(base->device));
        HIP_TRY(hipMemsetAsync(d_Q, 0, sizeof(double) * (size_t)blk_stride * (size_t)(max_steps + 1), stream()));
        HIP_TRY(hipMemcpyAsync(slot(0), x0_host, sizeof(double) * (size_t)n * kW, hipMemcpyHostToDevice, stream()));
        HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int), stream()));
        if (int rc = cholqr2(0, nullptr, false)) return rc;
        HIP_TRY(hipStreamSynchronize(stream()));
        harvest();
        steps = 0;
        started = true;
        return check_status();
    }

    int check_status() {
        int st[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(st, d_status, sizeof(st), hipMemcpyDeviceToHost));
        if (st[0] != 0)
            return lz::fail("lanczos_hip_block: block %d is rank deficient (an invariant subspace was reached, or the "
                            "start block has dependent columns)", st[0] - 1);
        return 0;
    }

    int step() {
        if (!started) return lz::fail("lanczos_hip_block: no start block set");
        if (steps >= max_steps) return lz::fail("lanczos_hip_block: basis full (%d block steps)", steps);
        const int j = steps;
        double *W = slot(j + 1);
        if (int rc = spmm(slot(j), W)) return rc;
        const int ta = std::max(j - 1, 0);
        if (int rc = gram(ta, j + 1, W, d_H)) return rc;
        {
            Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * 512);
            hipLaunchKernelGGL(k_symm16, dim3(1), dim3(kThreads), 0, stream(), (const double *)(d_H + (j - ta) * 256),
                               d_TA + (size_t)j * 256);
        }
        if (int rc = update(ta, j + 1, W, d_H, nullptr)) return rc;
        if (int rc = gram(0, j + 1, W, d_H)) return rc;
        if (int rc = update(0, j + 1, W, d_H, d_small)) return rc;
        if (int rc = cholqr2(j + 1, d_TB + (size_t)j * 256, true)) return rc;
        steps += 1;
        return 0;
    }

    int run(int count) {
        HIP_TRY(hipSetDevice(base->device));
        for (int s = 0; s < count; ++s)
            if (int rc = step()) return rc;
        HIP_TRY(hipStreamSynchronize(stream()));
        harvest();
        return check_status();
    }
};

BlockEngine *B(lanczos_hip_block *b) { return reinterpret_cast<BlockEngine *>(b); }

}  // namespace

extern "C" {

int lanczos_hip_block_create(lanczos_hip_block **out, int64_t n, int64_t nnz, const int64_t *row_head_indexes,
                             const int64_t *column_index, const double *values, int max_steps,
                             const lanczos_hip_options *opt_in) {
    return lz::guarded("lanczos_hip_block_create", [&]() -> int {
    if (!out) return lz::fail("lanczos_hip_block_create: NULL handle");
    *out = nullptr;
    if (max_steps < 1) return lz::fail("lanczos_hip_block_create: max_steps must be >= 1");
    if (n < kW) return lz::fail("lanczos_hip_block_create: dimension %lld below the block width %d", (long long)n, kW);
    lanczos_hip_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    if (opt_in) memcpy(&opt, opt_in, std::min(sizeof opt, opt_in->struct_size));
    if (opt.x_ext || opt.basis) return lz::fail("lanczos_hip_block_create: row-partitioned matrices are not supported");
    const int want_profile = opt.profile;
    opt.struct_size = sizeof opt;
    opt.reorth = LANCZOS_HIP_REORTH_NONE;  // the base engine only holds the matrix:
    opt.spmv_kernel = LANCZOS_HIP_SPMV_VECTOR;  // no format of its own to build
    opt.profile = 0;
    lz::Engine *base = nullptr;
    if (int rc = lz::Engine::create(&base, n, nnz, row_head_indexes, column_index, values, 2, &opt)) return rc;
    BlockEngine *b = new BlockEngine();
    b->base = base;
    {
        // the columns meant to stay in the L2s while the hub rows are multiplied: the most popular ones, about
        // three quarters of the eight L2s' lines (LANCZOS_HIP_SPMM_HOT_COLUMNS; 0: no such split).  A column's
        // popularity is taken to be its row's length (the matrix is symmetric).
        int64_t hot_columns = 0;
        if (const char *s = getenv("LANCZOS_HIP_SPMM_HOT_COLUMNS")) hot_columns = atoll(s);
        b->hot_degree_floor = 0;
        if (hot_columns > 0) {
            constexpr int64_t cap = 1 << 20;
            std::vector<int64_t> hist((size_t)cap + 1, 0);
            for (int64_t i = 0; i < n; ++i) hist[(size_t)std::min(row_head_indexes[i + 1] - row_head_indexes[i], cap)] += 1;
            int64_t acc = 0, floor_deg = cap + 1;
            for (int64_t d = cap; d >= 1 && acc + hist[(size_t)d] <= hot_columns; --d) {
                acc += hist[(size_t)d];
                floor_deg = d;
            }
            b->hot_degree_floor = floor_deg;
        }
    }
    if (int rc = b->alloc(max_steps, want_profile != 0)) {
        delete b;
        return rc;
    }
    *out = reinterpret_cast<lanczos_hip_block *>(b);
    return 0;
    });
}

void lanczos_hip_block_destroy(lanczos_hip_block *b) { delete B(b); }

int lanczos_hip_block_set_start(lanczos_hip_block *b, const double *x0_host) {
    if (!b || !x0_host) return lz::fail("lanczos_hip_block_set_start: NULL argument");
    return B(b)->set_start(x0_host);
}

int lanczos_hip_block_set_start_seed(lanczos_hip_block *b, uint64_t seed) {
    return lz::guarded("lanczos_hip_block_set_start_seed", [&]() -> int {
    if (!b) return lz::fail("lanczos_hip_block_set_start_seed: NULL handle");
    std::vector<double> x((size_t)B(b)->n * kW);
    lanczos_hip_gaussian_fill(seed, 0, (int64_t)x.size(), x.data());
    return B(b)->set_start(x.data());
    });
}

int lanczos_hip_block_run(lanczos_hip_block *b, int steps) {
    return lz::guarded("lanczos_hip_block_run", [&]() -> int {
    if (!b || steps < 0) return lz::fail("lanczos_hip_block_run: bad argument");
    return B(b)->run(steps);
    });
}

int lanczos_hip_block_steps(const lanczos_hip_block *b) {
    return b ? reinterpret_cast<const BlockEngine *>(b)->steps : 0;
}

int lanczos_hip_block_tmatrix(lanczos_hip_block *bh, int steps, double *a_out, double *b_out) {
    return lz::guarded("lanczos_hip_block_tmatrix", [&]() -> int {
    BlockEngine *b = B(bh);
    if (!b || steps < 0 || steps > b->steps) return lz::fail("lanczos_hip_block_tmatrix: bad argument");
    HIP_TRY(hipSetDevice(b->base->device));
    HIP_TRY(hipStreamSynchronize(b->stream()));
    if (steps && a_out) HIP_TRY(hipMemcpy(a_out, b->d_TA, sizeof(double) * 256 * steps, hipMemcpyDeviceToHost));
    if (steps && b_out) HIP_TRY(hipMemcpy(b_out, b->d_TB, sizeof(double) * 256 * steps, hipMemcpyDeviceToHost));
    return 0;
    });
}

int lanczos_hip_block_ritz(lanczos_hip_block *bh, int steps, int nth, double *eig_out) {
    return lz::guarded("lanczos_hip_block_ritz", [&]() -> int {
    BlockEngine *b = B(bh);
    if (!b || !eig_out || nth < 1 || steps < 1 || steps > b->steps)
        return lz::fail("lanczos_hip_block_ritz: bad argument");
    std::vector<double> ta((size_t)steps * 256), tb((size_t)steps * 256);
    if (int rc = lanczos_hip_block_tmatrix(bh, steps, ta.data(), tb.data())) return rc;
    const int N = steps * kW;
    std::vector<double> t((size_t)N * N, 0.0);
    for (int s = 0; s < steps; ++s)
        for (int i = 0; i < kW; ++i)
            for (int j = 0; j < kW; ++j) {
                t[(size_t)(s * kW + i) * N + s * kW + j] = ta[(size_t)s * 256 + i * kW + j];
                if (s + 1 < steps) {
                    const double v = tb[(size_t)s * 256 + i * kW + j];  // B_s sits below the diagonal
                    t[(size_t)((s + 1) * kW + i) * N + s * kW + j] = v;
                    t[(size_t)(s * kW + j) * N + (s + 1) * kW + i] = v;
                }
            }
    if (lanczos_hip_sym_lowest_eigs(t.data(), N, std::min(nth, N), eig_out) != 0)
        return lz::fail("lanczos_hip_block_ritz: eigen-solve failed");
    for (int i = N; i < nth; ++i) eig_out[i] = 0.0;
    return 0;
    });
}

int lanczos_hip_block_get_block(lanczos_hip_block *bh, int t, double *out_host) {
    BlockEngine *b = B(bh);
    if (!b || !out_host || t < 0 || t > b->max_steps) return lz::fail("lanczos_hip_block_get_block: bad argument");
    HIP_TRY(hipSetDevice(b->base->device));
    HIP_TRY(hipStreamSynchronize(b->stream()));
    HIP_TRY(hipMemcpy(out_host, b->slot(t), sizeof(double) * (size_t)b->n * kW, hipMemcpyDeviceToHost));
    return 0;
}

int lanczos_hip_block_set_block(lanczos_hip_block *bh, int t, const double *in_host) {
    BlockEngine *b = B(bh);
    if (!b || !in_host || t < 0 || t > b->max_steps) return lz::fail("lanczos_hip_block_set_block: bad argument");
    HIP_TRY(hipSetDevice(b->base->device));
    HIP_TRY(hipStreamSynchronize(b->stream()));
    HIP_TRY(hipMemcpy(b->slot(t), in_host, sizeof(double) * (size_t)b->n * kW, hipMemcpyHostToDevice));
    return 0;
}

int lanczos_hip_block_spmm(lanczos_hip_block *bh, int t_in, int t_out) {
    BlockEngine *b = B(bh);
    if (!b || t_in < 0 || t_out < 0 || t_in > b->max_steps || t_out > b->max_steps || t_in == t_out)
        return lz::fail("lanczos_hip_block_spmm: bad argument");
    HIP_TRY(hipSetDevice(b->base->device));
    if (int rc = b->spmm(b->slot(t_in), b->slot(t_out))) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream()));
    b->harvest();
    return 0;
}

int lanczos_hip_block_project(lanczos_hip_block *bh, int t0, int t1, int t_w, double *h_out_host) {
    BlockEngine *b = B(bh);
    if (!b || t0 < 0 || t1 <= t0 || t1 > b->max_steps + 1 || t_w < 0 || t_w > b->max_steps || (t_w >= t0 && t_w < t1))
        return lz::fail("lanczos_hip_block_project: bad argument");
    HIP_TRY(hipSetDevice(b->base->device));
    if (int rc = b->gram(t0, t1, b->slot(t_w), b->d_H)) return rc;
    if (int rc = b->update(t0, t1, b->slot(t_w), b->d_H, nullptr)) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream()));
    b->harvest();
    if (h_out_host)
        HIP_TRY(hipMemcpy(h_out_host, b->d_H, sizeof(double) * 256 * (size_t)(t1 - t0), hipMemcpyDeviceToHost));
    return 0;
}

int lanczos_hip_block_orthonormalise(lanczos_hip_block *bh, int t, double *factor_out_host) {
    BlockEngine *b = B(bh);
    if (!b || t < 0 || t > b->max_steps) return lz::fail("lanczos_hip_block_orthonormalise: bad argument");
    HIP_TRY(hipSetDevice(b->base->device));
    if (int rc = b->cholqr2(t, b->d_H, false)) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream()));
    b->harvest();
    if (factor_out_host) HIP_TRY(hipMemcpy(factor_out_host, b->d_H, sizeof(double) * 256, hipMemcpyDeviceToHost));
    return b->check_status();
}

int lanczos_hip_block_profile(const lanczos_hip_block *bh, int cls, double *ms, int64_t *launches, double *bytes) {
    const BlockEngine *b = reinterpret_cast<const BlockEngine *>(bh);
    if (!b || cls < 0 || cls >= LANCZOS_HIP_BK_COUNT) return lz::fail("lanczos_hip_block_profile: bad class %d", cls);
    if (ms) *ms = b->prof_ms[cls];
    if (launches) *launches = b->prof_launches[cls];
    if (bytes) *bytes = b->prof_bytes[cls];
    return 0;
}

void lanczos_hip_block_profile_reset(lanczos_hip_block *bh) {
    BlockEngine *b = B(bh);
    if (!b) return;
    b->harvest();
    for (int c = 0; c < LANCZOS_HIP_BK_COUNT; ++c) {
        b->prof_ms[c] = 0;
        b->prof_bytes[c] = 0;
        b->prof_launches[c] = 0;
    }
}

}  // extern "C"
