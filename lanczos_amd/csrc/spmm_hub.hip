// Sparse product of the block variant, Y = A X with X and Y n x 16 row-major (the block form of
// src/crs.c:155-178), for the "hub" rows of a power-law graph: the rows of more than a few entries (16 by
// default), a quarter of the rows and nine tenths of the entries.
//
// Every entry of such a row gathers one 128-byte row of X from a table far larger than the caches, and the
// row-wise kernels move one line per entry across the fabric (counted on an R-MAT graph of 2^24 vertices:
// 4.4e8 L2 misses for 4.6e8 gathers, 12.3 of the product's 17 ms).  What can be reused is the popular
// columns: a few hundred thousand of them collect most of the entries, but an XCD's L2 holds 32768 lines and
// every one of the eight L2s caches the same lines.  Here the hub rows' entries are regrouped once, on the
// device, by COLUMN CLASS -- a hash of the column, eight classes -- and class x is multiplied by workgroups
// b with b % 8 == x, which the hardware places on one XCD: the eight L2s then hold disjoint eighths of X.
// Optionally (degree_floor > 0; off by default, it measured no faster) the columns are also split by popularity:
// the entries of the popular columns are multiplied first, in a launch of their own, the others afterwards.
//
// A (class, hub row) pair is a run of consecutive entries of the stream.  Eight lanes walk a slice of 256
// entries and sum run by run; a run that lies inside its slice is stored as a partial row at once, a run
// that crosses slices leaves its parts in carry slots which a second kernel adds in slice order, and a third
// adds the up to sixteen partial rows of a hub row in class order: no atomics, the same bits on every run.
#include <hipcub/hipcub.hpp>

#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "engine.h"
#include "spmm_hub.h"

namespace lzb {
namespace {

constexpr int kT = 256;
constexpr int kCols = 16;  // block width
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int col_class(int c) { return (int)(((unsigned)c * 0x9E3779B1u) >> 29); }

template <typename RP>
__global__ void __launch_bounds__(kT)
k_hub_lens(const int *__restrict__ rows, int64_t H, const RP *__restrict__ rp, unsigned long long *__restrict__ lens) {
    const int64_t step = (int64_t)gridDim.x * kT;
    for (int64_t h = (int64_t)blockIdx.x * kT + threadIdx.x; h < H; h += step) {
        const int r = rows[h];
        lens[h] = (unsigned long long)((int64_t)rp[r + 1] - (int64_t)rp[r]);
    }
}

// keys[q] = (class of the entry's column) << hbits | hub row, idx[q] = the entry's place in the CRS arrays; q runs
// over the hub rows' entries in row order.  One workgroup per hub row.
template <typename RP>
__global__ void __launch_bounds__(kT)
k_hub_keys(const int *__restrict__ rows, int64_t H, const RP *__restrict__ rp, const int32_t *__restrict__ col,
           const unsigned long long *__restrict__ hoff, int64_t degree_floor, int hbits, uint32_t *__restrict__ keys,
           uint32_t *__restrict__ idx) {
    for (int64_t h = blockIdx.x; h < H; h += gridDim.x) {
        const int r = rows[h];
        const int64_t e0 = (int64_t)rp[r], len = (int64_t)rp[r + 1] - e0;
        const int64_t q0 = (int64_t)hoff[h];
        for (int64_t j = threadIdx.x; j < len; j += kT) {
            const int c = col[e0 + j];
            // a column's popularity is its row's length (symmetric matrices); never looked up without the split: the
            // columns of a row slab are places in a gathered table, not rows of this matrix
            uint32_t vc = (uint32_t)col_class(c);
            if (degree_floor > 0 && (int64_t)rp[c + 1] - (int64_t)rp[c] < degree_floor) vc += 8u;
            keys[q0 + j] = (vc << hbits) | (uint32_t)h;
            idx[q0 + j] = (uint32_t)(e0 + j);
        }
    }
}

__global__ void __launch_bounds__(kT)
k_hub_place(int64_t count, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ idx, uint32_t hmask,
            const int32_t *__restrict__ col, const double *__restrict__ val, HubEntry *__restrict__ ent) {
    const int64_t step = (int64_t)gridDim.x * kT;
    for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < count; i += step) {
        const uint32_t e = idx[i];
        HubEntry t;
        t.col = col[e];
        t.h = (int32_t)(keys[i] & hmask);
        t.val = val[e];
        ent[i] = t;
    }
}

// vptr[vc * H + h] = number of sorted keys below (vc << hbits | h); vptr[classes H] = count
__global__ void __launch_bounds__(kT)
k_hub_vptr(const uint32_t *__restrict__ keys, int64_t count, int64_t H, int classes, int hbits,
           uint32_t *__restrict__ vptr) {
    const int64_t total = (int64_t)classes * H;
    const int64_t step = (int64_t)gridDim.x * kT;
    for (int64_t v = (int64_t)blockIdx.x * kT + threadIdx.x; v <= total; v += step) {
        int64_t lo = 0;
        if (v == total) {
            lo = count;
        } else {
            const uint64_t key = ((uint64_t)(v / H) << hbits) | (uint64_t)(v % H);
            int64_t hi = count;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((uint64_t)keys[mid] < key) lo = mid + 1;
                else hi = mid;
            }
        }
        vptr[v] = (uint32_t)lo;
    }
}

template <bool COLD>
__device__ __forceinline__ d2v load_x(const double *p) {
    if (COLD) return __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    return *reinterpret_cast<const d2v *>(p);
}

// One launch multiplies the classes of one kind (hot: classes 0..7, cold: 8..15); workgroup b takes class
// b % 8 of them.  Eight lanes (lane c: columns 2c, 2c + 1 of X and Y) own a slice of kSlice entries.
template <bool COLD>
__global__ void __launch_bounds__(kT)
k_spmm16_hub(HubLayout lay, int64_t H, int slice_log2, int nt, const HubEntry *__restrict__ ent, const double *__restrict__ X,
             double *__restrict__ part, int *__restrict__ carry_h, double *__restrict__ carry_val) {
    // nt & 1: the stream is read past the caches' keep-lists, nt & 2: the partial rows are written that way: neither
    // is read again by this kernel, and every line they would occupy in an L2 is a row of X less
    auto load_entry = [&](int64_t i) -> HubEntry {
        if (nt & 1) return __builtin_bit_cast(HubEntry, __builtin_nontemporal_load(reinterpret_cast<const d2v *>(ent + i)));
        return ent[i];
    };
    auto store_row = [&](double *p, d2v v) {
        if (nt & 2) __builtin_nontemporal_store(v, reinterpret_cast<d2v *>(p));
        else *reinterpret_cast<d2v *>(p) = v;
    };
    const int64_t kSlice = (int64_t)1 << slice_log2;
    const int lane = threadIdx.x & 63, c = lane & 7, g = lane >> 3, wave = threadIdx.x >> 6;
    const int vc = (int)(blockIdx.x & 7) + (COLD ? 8 : 0);
    const int64_t e_lo = lay.e_lo[vc], e_hi = lay.e_lo[vc + 1];
    const int64_t s_base = lay.s_lo[vc];
    const int64_t nsl = (e_hi - e_lo + kSlice - 1) >> slice_log2;
    const int64_t stride = (int64_t)(gridDim.x >> 3) * 32;
    for (int64_t s = ((int64_t)(blockIdx.x >> 3) * 4 + wave) * 8 + g; s < nsl; s += stride) {
        const int64_t a = e_lo + (s << slice_log2);
        const int64_t b = a + kSlice < e_hi ? a + kSlice : e_hi;
        // a run goes on in the neighbouring slice when that slice's nearest entry belongs to the same hub row
        const int h_before = a > e_lo ? ent[a - 1].h : -1;
        const int h_after = b < e_hi ? ent[b].h : -1;
        const int64_t slot = 2 * (s_base + s);
        int cur = -1, carry0 = -1, carry1 = -1;
        bool first = true;
        d2v acc = d2v{0.0, 0.0};
        auto emit = [&](bool last) {
            const bool head = first && cur == h_before;
            const bool tail = last && cur == h_after;
            if (head || tail) {
                const int w = head ? 0 : 1;
                if (w) carry1 = cur;
                else carry0 = cur;
                store_row(carry_val + (slot + w) * kCols + 2 * c, acc);
            } else {
                store_row(part + ((int64_t)vc * H + cur) * kCols + 2 * c, acc);
            }
            first = false;
        };
        HubEntry nxt = HubEntry{0, -1, 0.0};
        if (a + c < b) nxt = load_entry(a + c);
        for (int64_t p = a; p < b; p += 8) {
            const HubEntry now = nxt;
            nxt = HubEntry{0, -1, 0.0};
            if (p + 8 + c < b) nxt = load_entry(p + 8 + c);
            d2v x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cj = __shfl(now.col, j, 8);
                x[j] = load_x<COLD>(X + (int64_t)cj * kCols + 2 * c);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int hj = __shfl(now.h, j, 8);
                const double vj = __shfl(now.val, j, 8);
                if (hj >= 0) {  // the same for the eight lanes of the group
                    if (hj != cur) {
                        if (cur >= 0) emit(false);
                        cur = hj;
                        acc = d2v{0.0, 0.0};
                    }
                    acc.x += vj * x[j].x;
                    acc.y += vj * x[j].y;
                }
            }
        }
        emit(true);
        if (c == 0) {
            carry_h[slot] = carry0;
            carry_h[slot + 1] = carry1;
        }
    }
}

// The runs that cross slices: a run begins as the last run of some slice (odd carry slot) and goes on as the first
// run of the following slices (even slots) for as long as those name the same hub row; its parts are added in
// slice order.  Eight lanes per run.
__global__ void __launch_bounds__(kT)
k_hub_fixup(HubLayout lay, int64_t H, int64_t slices, const int *__restrict__ carry_h,
            const double *__restrict__ carry_val, double *__restrict__ part) {
    const int c = threadIdx.x & 7;
    const int64_t groups = (int64_t)gridDim.x * (kT / 8);
    for (int64_t s = (int64_t)blockIdx.x * (kT / 8) + (threadIdx.x >> 3); s < slices; s += groups) {
        const int h = carry_h[2 * s + 1];
        if (h < 0) continue;
        int vc = 0;
        while (vc + 1 < kHubClasses && (int64_t)lay.s_lo[vc + 1] <= s) ++vc;
        const int64_t s_end = lay.s_lo[vc + 1];
        d2v sum = *reinterpret_cast<const d2v *>(carry_val + (2 * s + 1) * kCols + 2 * c);
        for (int64_t t = s + 1; t < s_end && carry_h[2 * t] == h; ++t) {
            const d2v v = *reinterpret_cast<const d2v *>(carry_val + (2 * t) * kCols + 2 * c);
            sum.x += v.x;
            sum.y += v.y;
        }
        *reinterpret_cast<d2v *>(part + ((int64_t)vc * H + h) * kCols + 2 * c) = sum;
    }
}

// Y[hub row] = the sum of its partial rows, in class order (classes without an entry of the row have none)
__global__ void __launch_bounds__(kT)
k_hub_combine(int64_t H, int classes, const int *__restrict__ rows, const uint32_t *__restrict__ vptr,
              const double *__restrict__ part, double *__restrict__ Y) {
    const int c = threadIdx.x & 7;
    const int64_t groups = (int64_t)gridDim.x * (kT / 8);
    for (int64_t h = (int64_t)blockIdx.x * (kT / 8) + (threadIdx.x >> 3); h < H; h += groups) {
        d2v sum = d2v{0.0, 0.0};
#pragma unroll 4
        for (int vc = 0; vc < classes; ++vc) {
            const int64_t v = (int64_t)vc * H + h;
            if (vptr[v + 1] > vptr[v]) {
                const d2v p = *reinterpret_cast<const d2v *>(part + v * kCols + 2 * c);
                sum.x += p.x;
                sum.y += p.y;
            }
        }
        __builtin_nontemporal_store(sum, reinterpret_cast<d2v *>(Y + (int64_t)rows[h] * kCols + 2 * c));
    }
}

// w[r] = what row r costs the row-wise kernel: its offsets and its 128 bytes of Y (2) plus one gather per entry;
// rows of more than `long_row` entries are left to other kernels and cost it a glance (1)
template <typename RP>
__global__ void __launch_bounds__(kT)
k_row_weights(int64_t n, const RP *__restrict__ rp, int long_row, unsigned long long *__restrict__ w) {
    const int64_t step = (int64_t)gridDim.x * kT;
    for (int64_t r = (int64_t)blockIdx.x * kT + threadIdx.x; r <= n; r += step) {
        unsigned long long v = 0;
        if (r < n) {
            const int64_t len = (int64_t)rp[r + 1] - (int64_t)rp[r];
            v = len > long_row ? 1ull : 2ull + (unsigned long long)len;
        }
        w[r] = v;
    }
}

// out[x] = first unit (of unit_rows rows) of the x-th of eight ranges of equal weight; wsum[r] = weight of rows < r
__global__ void k_weight_split(int64_t n, const unsigned long long *__restrict__ wsum, int64_t unit_rows, int64_t units,
                               long long *__restrict__ out) {
    const int x = threadIdx.x;
    if (x > 8) return;
    if (x == 0 || x == 8) {
        out[x] = x == 0 ? 0 : units;
        return;
    }
    const double want = (double)wsum[n] * x / 8.0;
    int64_t lo = 0, hi = units;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t r = mid * unit_rows < n ? mid * unit_rows : n;
        if ((double)wsum[r] < want) lo = mid + 1;
        else hi = mid;
    }
    out[x] = lo;
}

__global__ void k_hub_layout(const uint32_t *__restrict__ vptr, int64_t H, int classes, uint32_t *__restrict__ out) {
    if (threadIdx.x <= kHubClasses) out[threadIdx.x] = vptr[(int64_t)min((int)threadIdx.x, classes) * H];
}

}  // namespace

// d_out[0..8]: the rows cut into eight ranges of whole units whose rows cost the row-wise kernel the same
int row_cost_split(lz::Engine *base, int long_row, int64_t unit_rows, int64_t units, long long *d_out) {
    const int64_t n = base->n;
    hipStream_t st = base->stream;
    unsigned long long *d_w = nullptr, *d_sum = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t rc = hipMalloc(&d_w, sizeof(unsigned long long) * (size_t)(n + 1));
    if (rc == hipSuccess) rc = hipMalloc(&d_sum, sizeof(unsigned long long) * (size_t)(n + 1));
    if (rc == hipSuccess) {
        const int blocks = (int)std::min<int64_t>(base->max_blocks, (n + kT) / kT);
        if (base->rp64)
            hipLaunchKernelGGL((k_row_weights<int64_t>), dim3(blocks), dim3(kT), 0, st, n, (const int64_t *)base->d_rp,
                               long_row, d_w);
        else
            hipLaunchKernelGGL((k_row_weights<int32_t>), dim3(blocks), dim3(kT), 0, st, n, (const int32_t *)base->d_rp,
                               long_row, d_w);
        rc = hipGetLastError();
    }
    if (rc == hipSuccess) rc = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_w, d_sum, (int)(n + 1), st);
    if (rc == hipSuccess) rc = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1);
    if (rc == hipSuccess) rc = hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_w, d_sum, (int)(n + 1), st);
    if (rc == hipSuccess) {
        hipLaunchKernelGGL(k_weight_split, dim3(1), dim3(64), 0, st, n, (const unsigned long long *)d_sum, unit_rows, units,
                           d_out);
        rc = hipGetLastError();
    }
    if (rc == hipSuccess) rc = hipStreamSynchronize(st);
    (void)hipFree(d_w);
    (void)hipFree(d_sum);
    (void)hipFree(d_tmp);
    if (rc != hipSuccess) return lz::fail("lanczos_hip_block_create: row ranges for the XCDs: %s", hipGetErrorString(rc));
    return 0;
}

void hub_free(HubStream *hs) {
    (void)hipFree(hs->d_rows);
    (void)hipFree(hs->d_ent);
    (void)hipFree(hs->d_vptr);
    (void)hipFree(hs->d_part);
    (void)hipFree(hs->d_carry_h);
    (void)hipFree(hs->d_carry_val);
    *hs = HubStream();
}

int hub_build(lz::Engine *base, const int *d_rows_a, int count_a, const int *d_rows_b, int count_b, int64_t degree_floor,
              HubStream *out) {
    *out = HubStream();
    const int64_t H = (int64_t)count_a + count_b;
    if (H <= 0) return 0;
    if (base->nnz >= ((int64_t)1 << 32)) return 0;  // stream positions are 32-bit
    int hbits = 1;
    while (((int64_t)1 << hbits) < H) ++hbits;
    if (hbits + 4 > 32) return 0;
    hipStream_t st = base->stream;
    HubStream hs;
    hs.H = H;
    hs.slice_log2 = 8;  // 256 entries per slice (LANCZOS_HIP_SPMM_SLICE_LOG2: 6..12, for measurements)
    if (const char *e = getenv("LANCZOS_HIP_SPMM_SLICE_LOG2"))
        if (atoi(e) >= 6 && atoi(e) <= 12) hs.slice_log2 = atoi(e);
    const uint32_t kSlice = 1u << hs.slice_log2;
    int *d_unsorted = nullptr;
    unsigned long long *d_lens = nullptr, *d_hoff = nullptr;
    uint32_t *d_keys[2] = {nullptr, nullptr}, *d_idx[2] = {nullptr, nullptr}, *d_lay = nullptr;
    void *d_tmp = nullptr;
    auto drop_temps = [&]() {
        (void)hipFree(d_unsorted);
        (void)hipFree(d_lens);
        (void)hipFree(d_hoff);
        (void)hipFree(d_keys[0]);
        (void)hipFree(d_keys[1]);
        (void)hipFree(d_idx[0]);
        (void)hipFree(d_idx[1]);
        (void)hipFree(d_lay);
        (void)hipFree(d_tmp);
        d_tmp = nullptr;
    };
    // a failed step (device memory, above all) leaves the caller with its row-wise kernels: not an error
#define HUB_TRY(call)                      \
    do {                                   \
        if ((call) != hipSuccess) {        \
            (void)hipGetLastError();       \
            drop_temps();                  \
            hub_free(&hs);                 \
            return 0;                      \
        }                                  \
    } while (0)
    const int maxb = base->max_blocks;
    // ---- the hub rows in ascending order
    HUB_TRY(hipMalloc(&d_unsorted, sizeof(int) * (size_t)H));
    HUB_TRY(hipMalloc(&hs.d_rows, sizeof(int) * (size_t)H));
    if (count_a) HUB_TRY(hipMemcpyAsync(d_unsorted, d_rows_a, sizeof(int) * (size_t)count_a, hipMemcpyDeviceToDevice, st));
    if (count_b)
        HUB_TRY(hipMemcpyAsync(d_unsorted + count_a, d_rows_b, sizeof(int) * (size_t)count_b, hipMemcpyDeviceToDevice, st));
    size_t tmp_bytes = 0;
    HUB_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_unsorted, hs.d_rows, (unsigned int)H, 0, 32, st));
    HUB_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    HUB_TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_unsorted, hs.d_rows, (unsigned int)H, 0, 32, st));
    // ---- their lengths, where each one's entries start in the stream
    HUB_TRY(hipMalloc(&d_lens, sizeof(unsigned long long) * (size_t)(H + 1)));
    HUB_TRY(hipMalloc(&d_hoff, sizeof(unsigned long long) * (size_t)(H + 1)));
    HUB_TRY(hipMemsetAsync(d_lens, 0, sizeof(unsigned long long) * (size_t)(H + 1), st));
    const int hb = (int)std::min<int64_t>(maxb, (H + kT - 1) / kT);
    if (base->rp64)
        hipLaunchKernelGGL((k_hub_lens<int64_t>), dim3(hb), dim3(kT), 0, st, (const int *)hs.d_rows, H,
                           (const int64_t *)base->d_rp, d_lens);
    else
        hipLaunchKernelGGL((k_hub_lens<int32_t>), dim3(hb), dim3(kT), 0, st, (const int *)hs.d_rows, H,
                           (const int32_t *)base->d_rp, d_lens);
    HUB_TRY(hipGetLastError());
    (void)hipFree(d_tmp);
    d_tmp = nullptr;
    tmp_bytes = 0;
    HUB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_lens, d_hoff, (int)(H + 1), st));
    HUB_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    HUB_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_lens, d_hoff, (int)(H + 1), st));
    unsigned long long total = 0;
    HUB_TRY(hipMemcpyAsync(&total, d_hoff + H, sizeof(total), hipMemcpyDeviceToHost, st));
    HUB_TRY(hipStreamSynchronize(st));
    if (total == 0 || total >= ((unsigned long long)1 << 32)) {
        drop_temps();
        hub_free(&hs);
        return 0;
    }
    const int64_t E = (int64_t)total;
    hs.entries = E;
    // ---- (class, hub row) keys, stable sort, the stream
    for (int b = 0; b < 2; ++b) {
        HUB_TRY(hipMalloc(&d_keys[b], sizeof(uint32_t) * (size_t)E));
        HUB_TRY(hipMalloc(&d_idx[b], sizeof(uint32_t) * (size_t)E));
    }
    const int kb = (int)std::min<int64_t>(H, (int64_t)maxb * 4);
    if (base->rp64)
        hipLaunchKernelGGL((k_hub_keys<int64_t>), dim3(kb), dim3(kT), 0, st, (const int *)hs.d_rows, H,
                           (const int64_t *)base->d_rp, (const int32_t *)base->d_col, (const unsigned long long *)d_hoff,
                           degree_floor, hbits, d_keys[0], d_idx[0]);
    else
        hipLaunchKernelGGL((k_hub_keys<int32_t>), dim3(kb), dim3(kT), 0, st, (const int *)hs.d_rows, H,
                           (const int32_t *)base->d_rp, (const int32_t *)base->d_col, (const unsigned long long *)d_hoff,
                           degree_floor, hbits, d_keys[0], d_idx[0]);
    HUB_TRY(hipGetLastError());
    (void)hipFree(d_tmp);
    d_tmp = nullptr;
    tmp_bytes = 0;
    hipcub::DoubleBuffer<uint32_t> keys(d_keys[0], d_keys[1]);
    hipcub::DoubleBuffer<uint32_t> vals(d_idx[0], d_idx[1]);
    HUB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, vals, (unsigned int)E, 0, hbits + 4, st));
    HUB_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    HUB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, keys, vals, (unsigned int)E, 0, hbits + 4, st));
    HUB_TRY(hipMalloc(&hs.d_ent, sizeof(HubEntry) * (size_t)(E + 8)));
    HUB_TRY(hipMemsetAsync(hs.d_ent, 0, sizeof(HubEntry) * (size_t)(E + 8), st));
    const int eb = (int)std::min<int64_t>(maxb, (E + kT - 1) / kT);
    hipLaunchKernelGGL(k_hub_place, dim3(eb), dim3(kT), 0, st, E, (const uint32_t *)keys.Current(),
                       (const uint32_t *)vals.Current(), (uint32_t)(((uint64_t)1 << hbits) - 1),
                       (const int32_t *)base->d_col, (const double *)base->d_val, hs.d_ent);
    hs.classes = degree_floor > 0 ? kHubClasses : 8;
    const int64_t nv = (int64_t)hs.classes * H;
    HUB_TRY(hipMalloc(&hs.d_vptr, sizeof(uint32_t) * (size_t)(nv + 1)));
    const int vb = (int)std::min<int64_t>(maxb, (nv + 1 + kT - 1) / kT);
    hipLaunchKernelGGL(k_hub_vptr, dim3(vb), dim3(kT), 0, st, (const uint32_t *)keys.Current(), E, H, hs.classes, hbits,
                       hs.d_vptr);
    HUB_TRY(hipMalloc(&d_lay, sizeof(uint32_t) * (kHubClasses + 1)));
    hipLaunchKernelGGL(k_hub_layout, dim3(1), dim3(64), 0, st, (const uint32_t *)hs.d_vptr, H, hs.classes, d_lay);
    HUB_TRY(hipGetLastError());
    HUB_TRY(hipMemcpyAsync(hs.lay.e_lo, d_lay, sizeof(uint32_t) * (kHubClasses + 1), hipMemcpyDeviceToHost, st));
    HUB_TRY(hipStreamSynchronize(st));
    drop_temps();
    d_unsorted = nullptr;
    d_lens = d_hoff = nullptr;
    d_keys[0] = d_keys[1] = d_idx[0] = d_idx[1] = d_lay = nullptr;
    uint32_t s = 0;
    for (int vc = 0; vc < kHubClasses; ++vc) {
        hs.lay.s_lo[vc] = s;
        s += (hs.lay.e_lo[vc + 1] - hs.lay.e_lo[vc] + kSlice - 1) / kSlice;
    }
    hs.lay.s_lo[kHubClasses] = s;
    hs.slices = s;
    hs.hot_entries = hs.lay.e_lo[8];
    // ---- what a product writes besides Y
    HUB_TRY(hipMalloc(&hs.d_part, sizeof(double) * kCols * (size_t)nv));
    HUB_TRY(hipMalloc(&hs.d_carry_h, sizeof(int) * 2 * (size_t)std::max<int64_t>(hs.slices, 1)));
    HUB_TRY(hipMalloc(&hs.d_carry_val, sizeof(double) * kCols * 2 * (size_t)std::max<int64_t>(hs.slices, 1)));
#undef HUB_TRY
    *out = hs;
    return 0;
}

int hub_spmm(const HubStream &hs, hipStream_t st, int cus, const double *X, double *Y) {
    if (!hs.ready()) return lz::fail("lanczos_hip_block: no hub stream");
    int nt = 0;
    if (const char *e = getenv("LANCZOS_HIP_SPMM_NT")) nt = atoi(e);
    const int blocks = std::max(cus, 8) / 8 * 8 * 8;  // eight workgroups per CU, a multiple of eight
    if (hs.lay.e_lo[8] > hs.lay.e_lo[0])
        hipLaunchKernelGGL((k_spmm16_hub<false>), dim3(blocks), dim3(kT), 0, st, hs.lay, hs.H, hs.slice_log2, nt, (const HubEntry *)hs.d_ent, X,
                           hs.d_part, hs.d_carry_h, hs.d_carry_val);
    if (hs.lay.e_lo[16] > hs.lay.e_lo[8])
        hipLaunchKernelGGL((k_spmm16_hub<true>), dim3(blocks), dim3(kT), 0, st, hs.lay, hs.H, hs.slice_log2, nt, (const HubEntry *)hs.d_ent, X,
                           hs.d_part, hs.d_carry_h, hs.d_carry_val);
    const int fb = (int)std::min<int64_t>((int64_t)cus * 8, (hs.slices + kT / 8 - 1) / (kT / 8));
    hipLaunchKernelGGL(k_hub_fixup, dim3(std::max(fb, 1)), dim3(kT), 0, st, hs.lay, hs.H, hs.slices,
                       (const int *)hs.d_carry_h, (const double *)hs.d_carry_val, hs.d_part);
    const int cb = (int)std::min<int64_t>((int64_t)cus * 8, (hs.H + kT / 8 - 1) / (kT / 8));
    hipLaunchKernelGGL(k_hub_combine, dim3(std::max(cb, 1)), dim3(kT), 0, st, hs.H, hs.classes, (const int *)hs.d_rows,
                       (const uint32_t *)hs.d_vptr, (const double *)hs.d_part, Y);
    if (hipGetLastError() != hipSuccess) return lz::fail("lanczos_hip_block: launch of the hub-row product failed");
    return 0;
}

}  // namespace lzb
