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
    double *d_small = nullptr;  // G, Rinv, R1, R2, Rinv of the second factorisation: 5 x 256
    double *d_Hc = nullptr;     // the three-term part's coefficients of the current step: 2 x 256
    bool fold_passes = true;
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
        (void)hipFree(d_Hc);
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
        // Two workgroups a CU for the kernels that write W: any mix of reads and writes runs at 5.2-5.4 TB/s on this chip
        // whatever the kernel's shape (scripts/probe/mix_probe.hip), and fewer, longer streams sit at the upper end
        // (256^3, ms per launch at 1 / 2 / 3 / 4 / 8 a CU: update 2.25 / 2.05 / 2.10 / 2.13 / 2.13, scaling 0.85 / 0.81 / 0.82 / 0.85 / 0.85).
        groups_update = cus * 2;
        if (const char *g = getenv("LANCZOS_HIP_BLOCK_UPDATE_GROUPS"))
            if (atoi(g) >= 1 && atoi(g) <= 8 && atoi(g) <= 2 * (max_steps + 1)) groups_update = cus * atoi(g);
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
        HIP_TRY(hipMalloc(&d_small, sizeof(double) * 5 * 256));
        HIP_TRY(hipMalloc(&d_Hc, sizeof(double) * 2 * 256));
        if (const char *f = getenv("LANCZOS_HIP_BLOCK_FUSE")) fold_passes = atoi(f) != 0;  // 0: the three-term part and the first scaling as passes of their own that write the block
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

    // Hout[t - t0] = Q_t^T W' for W' = W - sum_{c0 <= c < c1} Q_c Hc[c - c0], W' never stored (c1 - c0 <= 2)
    int gram_corrected(int t0, int t1, const double *W, double *Hout, int c0, int c1, const double *Hc) {
        const int nb = t1 - t0;
        {
            // W once per segment; the correction's blocks are among the blocks of the first segment's products (their second
            // read there comes from L2 and is not charged) and are read again by every further segment
            const int segs = (nb + kGramSeg - 1) / kGramSeg;
            Scope ps(this, LANCZOS_HIP_BK_GRAM, 8.0 * kW * (double)n * (nb + segs + (segs - 1) * (c1 - c0)));
            hipLaunchKernelGGL(k_gram16_corr, dim3(groups_gram), dim3(kThreads), 0, stream(), (const double *)d_Q,
                               blk_stride, t0, t1, W, n_pad, d_hpart, hstride, c0, c1, Hc);
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

    // The three-term part of step j, from d_H = [Q_ta Q_j]^T W: T's diagonal block, and the coefficients kept in d_Hc
    // for the pass that applies them together with the full pass's.
    int keep_three_term(int ta, int j) {
        Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * 512 * 2);
        hipLaunchKernelGGL(k_symm16, dim3(1), dim3(kThreads), 0, stream(), (const double *)(d_H + (j - ta) * 256),
                           d_TA + (size_t)j * 256);
        HIP_TRY(hipMemcpyAsync(d_Hc, d_H, sizeof(double) * 256 * (size_t)(j + 1 - ta), hipMemcpyDeviceToDevice, stream()));
        return 0;
    }

    // d_H[ta .. j] += d_Hc: W - Q (H + Hc) is then both parts in one update
    int add_three_term(int ta, int j) {
        const int count = 256 * (j + 1 - ta);
        Scope ps(this, LANCZOS_HIP_BK_SMALL, 24.0 * count);
        hipLaunchKernelGGL(k_add_coeff, dim3((count + kThreads - 1) / kThreads), dim3(kThreads), 0, stream(),
                           d_H + (size_t)ta * 256, (const double *)d_Hc, count);
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
    void reduce_own_gram(double *G, int groups = 0) {
        if (groups == 0) groups = groups_update;
        Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * 256 * groups);
        hipLaunchKernelGGL(k_gram_reduce, dim3(1), dim3(kThreads), 0, stream(), (const double *)d_hpart, groups,
                           (int64_t)256, 256, G);
    }

    // slot t <- orthonormal basis of its columns; Bout (optional) = the factor: old = new * Bout
    // have_gram: d_small already holds W^T W of the block (left by the update that produced it)
    int cholqr2(int t, double *Bout, bool have_gram) {
        if (!have_gram)
            if (int rc = gram(t, t + 1, slot(t), d_small)) return rc;
        for (int pass = 0; pass < 2; ++pass) chol_pass(pass, t, Bout);
        HIP_TRY(hipGetLastError());
        return 0;
    }

    // One of the two passes of cholqr2 on slot t, from the Gram matrix in d_small (of the whole block: summed over the
    // slabs when the rows are partitioned): factorise, scale this engine's rows; the first pass leaves the Gram
    // matrix of ITS rows' result in d_small for the second.
    void chol_pass(int pass, int t, double *Bout) {
        double *G = d_small, *Rinv = d_small + 256, *R1 = d_small + 512, *R2 = d_small + 768, *Rinv2 = d_small + 1024;
        {
            Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * 256 * 3);
            hipLaunchKernelGGL(k_chol16, dim3(1), dim3(kThreads), 0, stream(), (const double *)G,
                               pass == 1 && fold_passes ? Rinv2 : Rinv, pass == 0 ? R1 : R2,
                               pass == 0 ? (const double *)nullptr : (const double *)R1,
                               pass == 0 ? (double *)nullptr : Bout, d_status, t + 1);
        }
        if (fold_passes) {
            // the block is not written between the two factorisations: the first pass only reads it (the Gram matrix of
            // W Rinv1, formed in registers), the second writes (W Rinv1) Rinv2 -- the same bits as two scalings
            if (pass == 0) {
                Scope ps(this, LANCZOS_HIP_BK_SCALE, 8.0 * kW * (double)n);
                hipLaunchKernelGGL(k_bscale16_gram, dim3(groups_gram), dim3(kThreads), 0, stream(), (const double *)slot(t),
                                   n_pad, (const double *)Rinv, d_hpart);
            } else {
                Scope ps(this, LANCZOS_HIP_BK_SCALE, 2.0 * 8.0 * kW * (double)n);
                hipLaunchKernelGGL(k_bscale16_chain, dim3(groups_update), dim3(kThreads), 0, stream(), slot(t), n_pad,
                                   (const double *)Rinv, (const double *)Rinv2);
            }
        } else {
            Scope ps(this, LANCZOS_HIP_BK_SCALE, 2.0 * 8.0 * kW * (double)n);
            // the first scaling leaves the Gram matrix of its result for the second factorisation
            hipLaunchKernelGGL(k_bscale16, dim3(groups_update), dim3(kThreads), 0, stream(), slot(t), n_pad,
                               (const double *)Rinv, pass == 0 ? d_hpart : (double *)nullptr);
        }
        if (pass == 0) reduce_own_gram(G, fold_passes ? groups_gram : groups_update);
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
        if (fold_passes) {
            // the three-term part is not written out: the full pass's products are taken of W less that part, formed in
            // registers, and one update applies both sets of coefficients
            if (int rc = keep_three_term(ta, j)) return rc;
            if (int rc = gram_corrected(0, j + 1, W, d_H, ta, j + 1, d_Hc)) return rc;
            if (int rc = add_three_term(ta, j)) return rc;
        } else {
            {
                Scope ps(this, LANCZOS_HIP_BK_SMALL, 8.0 * 512);
                hipLaunchKernelGGL(k_symm16, dim3(1), dim3(kThreads), 0, stream(), (const double *)(d_H + (j - ta) * 256),
                                   d_TA + (size_t)j * 256);
            }
            if (int rc = update(ta, j + 1, W, d_H, nullptr)) return rc;
            if (int rc = gram(0, j + 1, W, d_H)) return rc;
        }
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

// The block variant on row slabs, driven by ONE host process (the reference has neither a block form nor several
// GPUs; what this must agree with is the one-GPU block engine).  Slab p holds rows [bounds[p], bounds[p+1]) of A with
// its columns renumbered into a gathered table of X (slab q's rows at q * pad), the same rows of every basis block,
// and its own copy of the small matrices.  Per block step: the slabs' rows of Q_j are gathered into every table
// (all-gather of n x 16 doubles), then the step of the one-GPU engine with the 16 x 16 (x blocks) matrices summed over
// the slabs wherever that engine sums over its rows: H = Q^T W twice, the Gram matrix of the new block twice.
struct BlockMulti {
    int P = 0;
    bool loopback = false;
    int64_t n = 0, pad = 0;
    int max_steps = 0, steps = 0;
    bool started = false;
    std::vector<int> dev;
    std::vector<int64_t> bounds;
    std::vector<hipStream_t> stream;
    std::vector<BlockEngine *> eng;
    std::vector<double *> xtab;     // [P * pad][16] on every slab's device
    std::vector<double *> bufH, bufG;
    lz::SlabComm *comm = nullptr;

    ~BlockMulti() {
        for (size_t p = 0; p < eng.size(); ++p) {
            if (p < dev.size()) (void)hipSetDevice(dev[p]);
            delete eng[p];
            if (p < xtab.size()) (void)hipFree(xtab[p]);
        }
        lz::slab_comm_destroy(comm);
        for (size_t p = 0; p < stream.size(); ++p)
            if (stream[p] && (!loopback || p == 0)) {
                (void)hipSetDevice(dev[p]);
                (void)hipStreamDestroy(stream[p]);
            }
    }
    int use(int p) const {
        if (hipSetDevice(dev[(size_t)p]) != hipSuccess) return lz::fail("hipSetDevice(%d) failed", dev[(size_t)p]);
        return 0;
    }
    int reduce(std::vector<double *> &bufs, size_t count) { return lz::slab_all_reduce(comm, bufs.data(), count, stream.data()); }

    int build(int ngpu, const int *devices, int64_t n_, int64_t nnz, const int64_t *rp, const int64_t *ci, const double *va,
              int max_steps_, const lanczos_hip_options &opt_in, int flags) {
        P = ngpu;
        loopback = (flags & LANCZOS_HIP_MULTI_LOOPBACK) != 0;
        n = n_;
        max_steps = max_steps_;
        int ndev = 0;
        HIP_TRY(hipGetDeviceCount(&ndev));
        dev.resize((size_t)P);
        for (int p = 0; p < P; ++p) {
            dev[(size_t)p] = loopback ? (devices ? devices[0] : (opt_in.device >= 0 ? opt_in.device : 0)) : (devices ? devices[p] : p);
            if (dev[(size_t)p] < 0 || dev[(size_t)p] >= ndev)
                return lz::fail("lanczos_hip_block_multi_create: device %d of %d visible (slab %d)", dev[(size_t)p], ndev, p);
        }
        bounds.resize((size_t)P + 1);
        for (int p = 0; p <= P; ++p) bounds[(size_t)p] = (int64_t)(((__int128)p * n) / P);
        for (int p = 0; p < P; ++p) {
            if (bounds[(size_t)p + 1] - bounds[(size_t)p] < kW)
                return lz::fail("lanczos_hip_block_multi_create: slab %d would hold fewer rows than the block width", p);
            pad = std::max(pad, bounds[(size_t)p + 1] - bounds[(size_t)p]);
        }
        pad = (pad + 127) / 128 * 128;
        if ((int64_t)P * pad > (int64_t)INT32_MAX) return lz::fail("lanczos_hip_block_multi_create: gathered table too long");
        for (int64_t j = 0; j < nnz; ++j)
            if (ci[j] < 0 || ci[j] >= n) return lz::fail("lanczos_hip_block_multi_create: column index out of range [0, %lld)", (long long)n);
        stream.assign((size_t)P, nullptr);
        eng.assign((size_t)P, nullptr);
        xtab.assign((size_t)P, nullptr);
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            if (loopback && p > 0) stream[(size_t)p] = stream[0];
            else HIP_TRY(hipStreamCreateWithFlags(&stream[(size_t)p], hipStreamNonBlocking));
        }
        if (int rc = lz::slab_comm_create(&comm, P, dev.data(), loopback)) return rc;
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            const int64_t r0 = bounds[(size_t)p], r1 = bounds[(size_t)p + 1], nl = r1 - r0;
            const int64_t e0 = rp[r0], cnt = rp[r1] - e0;
            std::vector<int64_t> head((size_t)nl + 1), cols((size_t)std::max<int64_t>(cnt, 1));
            for (int64_t i = 0; i <= nl; ++i) head[(size_t)i] = rp[r0 + i] - e0;
            for (int64_t j = 0; j < cnt; ++j) {
                const int64_t c = ci[e0 + j];
                const int q = (int)(std::upper_bound(bounds.begin(), bounds.end(), c) - bounds.begin()) - 1;
                cols[(size_t)j] = (int64_t)q * pad + (c - bounds[(size_t)q]);
            }
            const size_t tab_bytes = sizeof(double) * kW * (size_t)P * (size_t)pad;
            HIP_TRY(hipMalloc(&xtab[(size_t)p], tab_bytes));
            HIP_TRY(hipMemsetAsync(xtab[(size_t)p], 0, tab_bytes, stream[(size_t)p]));
            lanczos_hip_options o = opt_in;
            o.struct_size = sizeof o;
            o.device = dev[(size_t)p];
            o.stream = stream[(size_t)p];
            o.reorth = LANCZOS_HIP_REORTH_NONE;      // the base engine only holds the slab's matrix
            o.spmv_kernel = LANCZOS_HIP_SPMV_VECTOR;
            o.profile = 0;
            o.n_global = n;
            o.row_begin = r0;
            o.n_ext = (int64_t)P * pad;
            o.x_ext = xtab[(size_t)p];               // never read as a vector: the table holds 16 doubles per column
            o.basis = nullptr;
            lz::Engine *base = nullptr;
            if (int rc = lz::Engine::create(&base, nl, cnt, head.data(), cols.data(), va + e0, 2, &o)) return rc;
            BlockEngine *b = new BlockEngine();
            b->base = base;
            eng[(size_t)p] = b;
            if (int rc = b->alloc(max_steps, false)) return rc;
        }
        bufH.resize((size_t)P);
        bufG.resize((size_t)P);
        for (int p = 0; p < P; ++p) {
            bufH[(size_t)p] = eng[(size_t)p]->d_H;
            bufG[(size_t)p] = eng[(size_t)p]->d_small;
        }
        return 0;
    }

    // Q_0 <- an orthonormal basis of x0's columns (n x 16 row-major on the host)
    int set_start(const double *x0) {
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            BlockEngine *b = eng[(size_t)p];
            HIP_TRY(hipMemsetAsync(b->d_Q, 0, sizeof(double) * (size_t)b->blk_stride * (size_t)(max_steps + 1), b->stream()));
            HIP_TRY(hipMemcpyAsync(b->slot(0), x0 + (size_t)bounds[(size_t)p] * kW, sizeof(double) * (size_t)b->n * kW,
                                   hipMemcpyHostToDevice, b->stream()));
            HIP_TRY(hipMemsetAsync(b->d_status, 0, sizeof(int), b->stream()));
            if (int rc = b->gram(0, 1, b->slot(0), b->d_small)) return rc;
        }
        if (int rc = orthonormalise(0, (const size_t *)nullptr)) return rc;
        steps = 0;
        started = true;
        return sync_and_check();
    }

    // the two passes of the one-GPU engine's cholqr2 with the Gram matrices (the slabs' parts are in d_small) summed over the slabs
    int orthonormalise(int t, const size_t *tb_offset) {
        for (int pass = 0; pass < 2; ++pass) {
            if (int rc = reduce(bufG, 256)) return rc;
            for (int p = 0; p < P; ++p) {
                if (int rc = use(p)) return rc;
                BlockEngine *b = eng[(size_t)p];
                b->chol_pass(pass, t, tb_offset ? b->d_TB + *tb_offset : nullptr);
            }
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }

    int sync_and_check() {
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            HIP_TRY(hipStreamSynchronize(stream[(size_t)p]));
            if (int rc = eng[(size_t)p]->check_status()) return rc;
        }
        return 0;
    }

    int step() {
        if (!started) return lz::fail("lanczos_hip_block_multi: no start block set");
        if (steps >= max_steps) return lz::fail("lanczos_hip_block_multi: basis full (%d block steps)", steps);
        const int j = steps;
        const int ta = std::max(j - 1, 0);
        // every slab's rows of Q_j into its place of its own table, then into everybody's
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            BlockEngine *b = eng[(size_t)p];
            HIP_TRY(hipMemcpyAsync(xtab[(size_t)p] + (size_t)p * (size_t)pad * kW, b->slot(j), sizeof(double) * (size_t)b->n * kW,
                                   hipMemcpyDeviceToDevice, b->stream()));
        }
        if (int rc = lz::slab_all_gather(comm, xtab.data(), (size_t)pad * kW, (size_t)pad * kW, stream.data())) return rc;
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            BlockEngine *b = eng[(size_t)p];
            if (int rc = b->spmm(xtab[(size_t)p], b->slot(j + 1))) return rc;
            if (int rc = b->gram(ta, j + 1, b->slot(j + 1), b->d_H)) return rc;
        }
        if (int rc = reduce(bufH, (size_t)(j + 1 - ta) * 256)) return rc;
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            BlockEngine *b = eng[(size_t)p];
            if (b->fold_passes) {
                if (int rc = b->keep_three_term(ta, j)) return rc;
                if (int rc = b->gram_corrected(0, j + 1, b->slot(j + 1), b->d_H, ta, j + 1, b->d_Hc)) return rc;
            } else {
                hipLaunchKernelGGL(k_symm16, dim3(1), dim3(kThreads), 0, b->stream(), (const double *)(b->d_H + (j - ta) * 256),
                                   b->d_TA + (size_t)j * 256);
                if (int rc = b->update(ta, j + 1, b->slot(j + 1), b->d_H, nullptr)) return rc;
                if (int rc = b->gram(0, j + 1, b->slot(j + 1), b->d_H)) return rc;
            }
        }
        if (int rc = reduce(bufH, (size_t)(j + 1) * 256)) return rc;
        for (int p = 0; p < P; ++p) {
            if (int rc = use(p)) return rc;
            BlockEngine *b = eng[(size_t)p];
            if (b->fold_passes)
                if (int rc = b->add_three_term(ta, j)) return rc;
            if (int rc = b->update(0, j + 1, b->slot(j + 1), b->d_H, b->d_small)) return rc;
        }
        const size_t tb = (size_t)j * 256;
        if (int rc = orthonormalise(j + 1, &tb)) return rc;
        steps += 1;
        return 0;
    }

    int run(int count) {
        for (int s = 0; s < count; ++s)
            if (int rc = step()) return rc;
        return sync_and_check();
    }
};

struct DeviceGuard {
    int dev = -1;
    DeviceGuard() {
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    }
    ~DeviceGuard() {
        if (dev >= 0) (void)hipSetDevice(dev);
    }
};

BlockMulti *BM(lanczos_hip_block_multi *m) { return reinterpret_cast<BlockMulti *>(m); }

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

// ---- the block variant on row slabs -----------------------------------------------------------------------------

int lanczos_hip_block_multi_create(lanczos_hip_block_multi **out, int ngpu, const int *devices, int64_t n, int64_t nnz,
                                   const int64_t *row_head_indexes, const int64_t *column_index, const double *values,
                                   int max_steps, const lanczos_hip_options *opt_in, int flags) {
    return lz::guarded("lanczos_hip_block_multi_create", [&]() -> int {
    if (!out) return lz::fail("lanczos_hip_block_multi_create: NULL handle");
    *out = nullptr;
    if (ngpu < 1 || ngpu > 64) return lz::fail("lanczos_hip_block_multi_create: ngpu=%d", ngpu);
    if (max_steps < 1) return lz::fail("lanczos_hip_block_multi_create: max_steps must be >= 1");
    if (n <= 0 || nnz < 0 || !row_head_indexes || (nnz > 0 && (!column_index || !values)) || row_head_indexes[0] != 0 ||
        row_head_indexes[n] != nnz)
        return lz::fail("lanczos_hip_block_multi_create: bad matrix");
    lanczos_hip_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    if (opt_in) memcpy(&opt, opt_in, std::min(sizeof opt, opt_in->struct_size));
    if (opt.x_ext || opt.basis) return lz::fail("lanczos_hip_block_multi_create: pass the whole matrix; the slabs are cut here");
    DeviceGuard guard;
    BlockMulti *m = new BlockMulti();
    if (int rc = m->build(ngpu, devices, n, nnz, row_head_indexes, column_index, values, max_steps, opt, flags)) {
        delete m;
        return rc;
    }
    *out = reinterpret_cast<lanczos_hip_block_multi *>(m);
    return 0;
    });
}

void lanczos_hip_block_multi_destroy(lanczos_hip_block_multi *m) {
    DeviceGuard guard;
    delete BM(m);
}

int lanczos_hip_block_multi_set_start(lanczos_hip_block_multi *m, const double *x0_host) {
    return lz::guarded("lanczos_hip_block_multi_set_start", [&]() -> int {
    if (!m || !x0_host) return lz::fail("lanczos_hip_block_multi_set_start: NULL argument");
    DeviceGuard guard;
    return BM(m)->set_start(x0_host);
    });
}

int lanczos_hip_block_multi_set_start_seed(lanczos_hip_block_multi *m, uint64_t seed) {
    return lz::guarded("lanczos_hip_block_multi_set_start_seed", [&]() -> int {
    if (!m) return lz::fail("lanczos_hip_block_multi_set_start_seed: NULL handle");
    std::vector<double> x((size_t)BM(m)->n * kW);
    lanczos_hip_gaussian_fill(seed, 0, (int64_t)x.size(), x.data());
    DeviceGuard guard;
    return BM(m)->set_start(x.data());
    });
}

int lanczos_hip_block_multi_run(lanczos_hip_block_multi *m, int steps) {
    return lz::guarded("lanczos_hip_block_multi_run", [&]() -> int {
    if (!m || steps < 0) return lz::fail("lanczos_hip_block_multi_run: bad argument");
    DeviceGuard guard;
    return BM(m)->run(steps);
    });
}

int lanczos_hip_block_multi_steps(const lanczos_hip_block_multi *m) {
    return m ? reinterpret_cast<const BlockMulti *>(m)->steps : 0;
}

int lanczos_hip_block_multi_slabs(const lanczos_hip_block_multi *m) {
    return m ? reinterpret_cast<const BlockMulti *>(m)->P : 0;
}

int lanczos_hip_block_multi_tmatrix(lanczos_hip_block_multi *mh, int steps, double *a_out, double *b_out) {
    return lz::guarded("lanczos_hip_block_multi_tmatrix", [&]() -> int {
    BlockMulti *m = BM(mh);
    if (!m || steps < 0 || steps > m->steps) return lz::fail("lanczos_hip_block_multi_tmatrix: bad argument");
    DeviceGuard guard;
    if (int rc = m->sync_and_check()) return rc;
    // every slab holds the same T: slab 0's copy
    m->eng[0]->steps = m->steps;
    return lanczos_hip_block_tmatrix(reinterpret_cast<lanczos_hip_block *>(m->eng[0]), steps, a_out, b_out);
    });
}

int lanczos_hip_block_multi_ritz(lanczos_hip_block_multi *mh, int steps, int nth, double *eig_out) {
    return lz::guarded("lanczos_hip_block_multi_ritz", [&]() -> int {
    BlockMulti *m = BM(mh);
    if (!m || !eig_out || nth < 1 || steps < 1 || steps > m->steps) return lz::fail("lanczos_hip_block_multi_ritz: bad argument");
    DeviceGuard guard;
    if (int rc = m->sync_and_check()) return rc;
    m->eng[0]->steps = m->steps;
    return lanczos_hip_block_ritz(reinterpret_cast<lanczos_hip_block *>(m->eng[0]), steps, nth, eig_out);
    });
}

int lanczos_hip_block_multi_get_block(lanczos_hip_block_multi *mh, int t, double *out_host) {
    return lz::guarded("lanczos_hip_block_multi_get_block", [&]() -> int {
    BlockMulti *m = BM(mh);
    if (!m || !out_host || t < 0 || t > m->max_steps) return lz::fail("lanczos_hip_block_multi_get_block: bad argument");
    DeviceGuard guard;
    for (int p = 0; p < m->P; ++p)
        if (int rc = lanczos_hip_block_get_block(reinterpret_cast<lanczos_hip_block *>(m->eng[(size_t)p]), t,
                                                 out_host + (size_t)m->bounds[(size_t)p] * kW))
            return rc;
    return 0;
    });
}

}  // extern "C"
