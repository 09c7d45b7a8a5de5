// SpMV by tiles, for matrices whose rows reach all over a vector that is too
// large for the caches (random sparsity: the row-wise product of
// src/crs.c:155-178 then fetches one memory line per entry, and the fabric's
// request rate, not its bandwidth, sets the time).  The entries are regrouped
// once, on the device, by (row block, column panel): a row block is 4096 rows,
// whose part of y a workgroup keeps in LDS; a panel is 2^bits columns, whose
// slice of x fits an XCD's L2.  A workgroup owns one row block and sweeps the
// panels in order; the workgroups of a launch start together and carry equal
// work per panel, so they sweep in step and the gathers hit L2.  Each entry is
// 12 bytes as in CRS (value + row-in-block and column-in-panel packed into 32
// bits).  Inside a tile the entries are sorted by row; a 256-entry chunk is
// multiplied lane per entry, and the first lane of every run of equal rows adds
// the run to the row's LDS slot, chunks separated by barriers: no atomics, the
// same sums on every run.
#include <hipcub/hipcub.hpp>

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "engine.h"

namespace lz {

static constexpr int kBlock = 256;
static constexpr int kRowBits = 12;             // rows per block = 4096 (32 KiB of y in LDS)
static constexpr int kRowsPerBlock = 1 << kRowBits;
static constexpr int kColField = 20;            // bits of the packed word that hold the column in its panel
static constexpr int kBlocksPerLaunch = 1024;   // 4 workgroups per CU: all of a launch are resident together

template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_tile_keys(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col, int bits, int jbits,
            uint32_t *__restrict__ keys, uint32_t *__restrict__ idx, int32_t *__restrict__ rowid) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += step)
        for (int64_t e = (int64_t)rp[r]; e < (int64_t)rp[r + 1]; ++e) {
            keys[e] = ((uint32_t)(r >> kRowBits) << jbits) | ((uint32_t)col[e] >> bits);
            idx[e] = (uint32_t)e;
            rowid[e] = (int32_t)r;
        }
}

// position q of the tile order holds entry idx[q]
__global__ void __launch_bounds__(kBlock)
k_tile_place(int64_t nnz, const uint32_t *__restrict__ idx, const int32_t *__restrict__ rowid,
             const int32_t *__restrict__ col, const double *__restrict__ val, int bits,
             uint32_t *__restrict__ packed, double *__restrict__ pval) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    const uint32_t cmask = (1u << bits) - 1u;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < nnz; q += step) {
        const uint32_t e = idx[q];
        packed[q] = ((uint32_t)(rowid[e] & (kRowsPerBlock - 1)) << kColField) | ((uint32_t)col[e] & cmask);
        pval[q] = val[e];
    }
}

// tile_ptr[t] = first position whose key is >= key(t), t = K * panels + J, for t = 0..tiles
__global__ void __launch_bounds__(kBlock)
k_tile_bounds(int64_t nnz, const uint32_t *__restrict__ keys, int64_t tiles, int panels, int jbits,
              uint32_t *__restrict__ tile_ptr) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t <= tiles; t += step) {
        const uint64_t want = ((uint64_t)(t / panels) << jbits) | (uint64_t)(t % panels);
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((uint64_t)keys[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        tile_ptr[t] = (uint32_t)lo;
    }
}

// One workgroup = one block of 4096 rows: y block in LDS, panels swept in order.  partials[b]
// receives (first launch) or accumulates (later launches, stream ordered) xdot . y of the block.
__global__ void __launch_bounds__(kBlock)
k_spmv_tiles(int64_t n, int64_t block0, int64_t nblocks, int panels, int bits,
             const uint32_t *__restrict__ tile_ptr, const uint32_t *__restrict__ packed,
             const double *__restrict__ pval, const double *__restrict__ x, const double *__restrict__ xdot,
             double *__restrict__ y, double *__restrict__ partials, int accumulate) {
    __shared__ double ylds[kRowsPerBlock];
    __shared__ double sp[kBlock];
    __shared__ int sr[kBlock];
    __shared__ double red[kBlock / 64];
    const int tid = threadIdx.x;
    const int64_t K = block0 + blockIdx.x;
    double acc = 0.0;
    if (K < nblocks) {
        for (int i = tid; i < kRowsPerBlock; i += kBlock) ylds[i] = 0.0;
        __syncthreads();
        const uint32_t *tp = tile_ptr + K * panels;
        for (int J = 0; J < panels; ++J) {
            const uint32_t a = tp[J], b = tp[J + 1];
            const double *xp = x + ((int64_t)J << bits);
            for (uint32_t q0 = a; q0 < b; q0 += kBlock) {
                const uint32_t q = q0 + tid;
                int r = -1;
                double prod = 0.0;
                if (q < b) {
                    const uint32_t pk = __builtin_nontemporal_load(packed + q);
                    prod = __builtin_nontemporal_load(pval + q) * xp[pk & ((1u << kColField) - 1u)];
                    r = (int)(pk >> kColField);
                }
                sp[tid] = prod;
                sr[tid] = r;
                __syncthreads();
                if (r >= 0 && (tid == 0 || sr[tid - 1] != r)) {
                    double sum = prod;
                    for (int j = tid + 1; j < kBlock && sr[j] == r; ++j) sum += sp[j];
                    ylds[r] += sum;
                }
                __syncthreads();
            }
        }
        const int64_t row0 = K << kRowBits;
        for (int i = tid; i < kRowsPerBlock; i += kBlock)
            if (row0 + i < n) {
                const double v = ylds[i];
                y[row0 + i] = v;
                acc += xdot[row0 + i] * v;
            }
    }
    // workgroup sum in a fixed order
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        const double t = (red[0] + red[1]) + (red[2] + red[3]);
        partials[blockIdx.x] = accumulate ? partials[blockIdx.x] + t : t;
    }
}

void Engine::free_panels() {
    (void)hipFree(d_packed);
    (void)hipFree(d_pval);
    (void)hipFree(d_tile_ptr);
    d_packed = nullptr;
    d_pval = nullptr;
    d_tile_ptr = nullptr;
    tile_panels = 0;
}

// Regroup the device CRS by (row block, column panel).  tile_panels > 0 afterwards when the form
// was built; 0 when the matrix does not qualify (too many entries for 32-bit positions, or a key
// wider than 32 bits).  Non-zero return on errors.
int Engine::build_panels(int bits) {
    free_panels();
    if (bits > kColField) bits = kColField;
    if (nnz == 0 || nnz >= ((int64_t)1 << 32)) return 0;
    const int panels = (int)(((n_ext - 1) >> bits) + 1);
    const int64_t nblocks = (n + kRowsPerBlock - 1) >> kRowBits;
    int jbits = 1, kbits = 1;
    while ((1 << jbits) < panels) jbits++;
    while (((int64_t)1 << kbits) < nblocks) kbits++;
    if (jbits + kbits > 32) return 0;
    const int64_t tiles = nblocks * panels;
    uint32_t *d_keys[2] = {nullptr, nullptr}, *d_idx[2] = {nullptr, nullptr};
    int32_t *d_rowid = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_keys[0]);
        (void)hipFree(d_keys[1]);
        (void)hipFree(d_idx[0]);
        (void)hipFree(d_idx[1]);
        (void)hipFree(d_rowid);
        (void)hipFree(d_tmp);
    };
#define PN_TRY(call)                                                                                   \
    do {                                                                                               \
        hipError_t st__ = (call);                                                                      \
        if (st__ != hipSuccess) {                                                                      \
            cleanup();                                                                                 \
            free_panels();                                                                             \
            return fail("HIP API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,            \
                        hipGetErrorString(st__), (int)st__);                                           \
        }                                                                                              \
    } while (0)
    for (int b = 0; b < 2; ++b) {
        PN_TRY(hipMalloc(&d_keys[b], sizeof(uint32_t) * nnz));
        PN_TRY(hipMalloc(&d_idx[b], sizeof(uint32_t) * nnz));
    }
    PN_TRY(hipMalloc(&d_rowid, sizeof(int32_t) * nnz));
    const int rblocks = (int)std::min<int64_t>(max_blocks, (n + kBlock - 1) / kBlock);
    if (rp64)
        hipLaunchKernelGGL((k_tile_keys<int64_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int64_t *)d_rp,
                           (const int32_t *)d_col, bits, jbits, d_keys[0], d_idx[0], d_rowid);
    else
        hipLaunchKernelGGL((k_tile_keys<int32_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int32_t *)d_rp,
                           (const int32_t *)d_col, bits, jbits, d_keys[0], d_idx[0], d_rowid);
    PN_TRY(hipGetLastError());
    hipcub::DoubleBuffer<uint32_t> keys(d_keys[0], d_keys[1]);
    hipcub::DoubleBuffer<uint32_t> vals(d_idx[0], d_idx[1]);
    size_t tmp_bytes = 0;
    PN_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, vals, (unsigned int)nnz, 0, jbits + kbits,
                                              stream));
    PN_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    PN_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, keys, vals, (unsigned int)nnz, 0, jbits + kbits,
                                              stream));
    PN_TRY(hipMalloc(&d_packed, sizeof(uint32_t) * nnz));
    PN_TRY(hipMalloc(&d_pval, sizeof(double) * nnz));
    PN_TRY(hipMalloc(&d_tile_ptr, sizeof(uint32_t) * (tiles + 1)));
    const int eblocks = (int)std::min<int64_t>(max_blocks, (nnz + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_tile_place, dim3(eblocks), dim3(kBlock), 0, stream, nnz, (const uint32_t *)vals.Current(),
                       (const int32_t *)d_rowid, (const int32_t *)d_col, (const double *)d_val, bits, d_packed, d_pval);
    const int tblocks = (int)std::min<int64_t>(max_blocks, (tiles + 1 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_tile_bounds, dim3(tblocks), dim3(kBlock), 0, stream, nnz, (const uint32_t *)keys.Current(),
                       tiles, panels, jbits, d_tile_ptr);
    PN_TRY(hipGetLastError());
    PN_TRY(hipStreamSynchronize(stream));
    cleanup();
#undef PN_TRY
    tile_panels = panels;
    tile_bits = bits;
    tile_blocks_n = nblocks;
    return 0;
}

int Engine::launch_spmv_panels(const double *x, const double *xdot, double *y, int *nparts) {
    int launches = 0;
    for (int64_t b0 = 0; b0 < tile_blocks_n; b0 += kBlocksPerLaunch, ++launches)
        hipLaunchKernelGGL(k_spmv_tiles, dim3(kBlocksPerLaunch), dim3(kBlock), 0, stream, n, b0, tile_blocks_n,
                           tile_panels, tile_bits, (const uint32_t *)d_tile_ptr, (const uint32_t *)d_packed,
                           (const double *)d_pval, x, xdot, y, d_partials, launches > 0 ? 1 : 0);
    *nparts = kBlocksPerLaunch;
    const hipError_t st = hipGetLastError();
    if (st != hipSuccess) return fail("lanczos_hip: %s", hipGetErrorString(st));
    return 0;
}

}  // namespace lz
