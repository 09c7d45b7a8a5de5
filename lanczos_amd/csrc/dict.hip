// SpMV through a dictionary of row patterns, for matrices whose rows repeat
// (stencils on structured grids: the reference's own test matrices are 1-D, 2-D
// and 3-D Laplacians).  A row's pattern is its entries written relative to the
// row: ((col - row), value) in order.  The 256^3 7-point Laplacian has 27
// distinct patterns among 16.7 M rows.  The matrix then is one 16-bit pattern
// number per row plus the dictionary, and
//     y[r] = sum_j dict_val[p][j] * x[r + dict_off[p][j]],  p = pattern[r]
// reads 2 bytes per row instead of 12 bytes per entry (src/crs.c:155-178 reads
// 16).  The entries are taken in their CRS order, so y is bit for bit what the
// row-wise kernel computes.  The dictionary is found on the device (hash per
// row, sort, unique) and every row is then compared entry by entry with its
// pattern: a hash collision cannot slip through.
//
// Three forms of the product, chosen per dictionary (Engine::dict_form): k_spmv_dict (loads inside the
// workgroup step; any dictionary), k_spmv_dict_ahead (loads one step ahead, at most two distinct far
// offsets), k_spmv_dict_planes (grids in three dimensions, walked plane by plane; also the row-wise CRS
// product of a grid whose values do not repeat, over a dictionary of column offsets alone).
#include <hipcub/hipcub.hpp>

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "engine.h"

namespace lz {

static constexpr int kBlock = 256;
static constexpr int kDictMaxLen = 32;       // longest row a pattern may have
static constexpr int kDictMaxPatterns = 65535;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// with_values = 0: a row's pattern is its column offsets alone (the values then stay in the CRS array)
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_row_hash(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col, const double *__restrict__ val,
           uint64_t *__restrict__ hash, int *__restrict__ too_long, int with_values) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += step) {
        const int64_t a = (int64_t)rp[r], b = (int64_t)rp[r + 1];
        if (b - a > kDictMaxLen) {
            *too_long = 1;
            hash[r] = 0;
            continue;
        }
        uint64_t h = mix64((uint64_t)(b - a));
        for (int64_t e = a; e < b; ++e) {
            h = mix64(h ^ (uint64_t)(uint32_t)(col[e] - (int32_t)r));
            if (with_values) h = mix64(h ^ (uint64_t)__double_as_longlong(val[e]));
        }
        hash[r] = h;
    }
}

// pattern[r] = index of hash[r] among the sorted distinct hashes; rep[p] = lowest row with pattern p,
// rep[count + p] = highest one
__global__ void __launch_bounds__(kBlock)
k_dict_assign(int64_t n, const uint64_t *__restrict__ hash, const uint64_t *__restrict__ uniq, int count,
              uint16_t *__restrict__ pattern, int *__restrict__ rep) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += step) {
        const uint64_t h = hash[r];
        int lo = 0, hi = count;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < h) lo = mid + 1;
            else hi = mid;
        }
        pattern[r] = (uint16_t)lo;
        // rep only ever decreases: a plain look first keeps 16 M atomics off a handful of addresses
        if (__builtin_nontemporal_load(&rep[lo]) > (int)r) atomicMin(&rep[lo], (int)r);
        if (__builtin_nontemporal_load(&rep[count + lo]) < (int)r) atomicMax(&rep[count + lo], (int)r);
    }
}

template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_dict_fill(int count, const int *__restrict__ rep, const RP *__restrict__ rp, const int32_t *__restrict__ col,
            const double *__restrict__ val, int stride, int32_t *__restrict__ dlen, int32_t *__restrict__ doff,
            double *__restrict__ dval) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= count) return;
    const int r = rep[p];
    const int64_t a = (int64_t)rp[r];
    const int len = (int)((int64_t)rp[r + 1] - a);
    dlen[p] = len;
    for (int j = 0; j < stride; ++j) {
        doff[p * stride + j] = j < len ? col[a + j] - r : 0;
        dval[p * stride + j] = j < len ? val[a + j] : 0.0;
    }
}

// every row against its pattern, entry by entry
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_dict_verify(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col, const double *__restrict__ val,
              const uint16_t *__restrict__ pattern, int stride, const int32_t *__restrict__ dlen,
              const int32_t *__restrict__ doff, const double *__restrict__ dval, int *__restrict__ mismatch,
              int with_values) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += step) {
        const int p = pattern[r];
        const int64_t a = (int64_t)rp[r];
        const int len = (int)((int64_t)rp[r + 1] - a);
        bool ok = len == dlen[p];
        for (int j = 0; ok && j < len; ++j)
            ok = (col[a + j] - (int32_t)r == doff[p * stride + j]) &&
                 (!with_values || __double_as_longlong(val[a + j]) == __double_as_longlong(dval[p * stride + j]));
        if (!ok) *mismatch = 1;
    }
}

// One lane per row, 64 consecutive rows per wavefront, kDictRows rows per thread.  The dictionary
// sits in LDS when it fits (LDS_DICT).  A workgroup step covers kSpan consecutive rows and first
// copies x[first row - halo .. last row + halo] into LDS with coalesced loads, each memory line once;
// entries whose column offset lies within the halo read that window, only the far ones (the
// +-N^2 neighbours of a 3-D grid) go to memory, again one line once per step.  This matters more
// than the bytes: the vector L1 handles requests in order and stalls on a second touch of a line
// whose fill is still in flight, which is what the overlapping gathers of a stencil produce
// (measured: the L1 sat in that stall 70 % of the time with direct gathers).
// Workgroup b serves the b % 8-th eighth of the rows (b and b + 8 share an L2).
// partials[b] = this workgroup's part of xdot . y over the rows [dot_lo, dot_hi).
constexpr int kDictRows = 4;
constexpr int kDictLdsEntries = 1024;  // dictionary entries (pattern x position) copied to LDS (12 KiB)
constexpr int kDictSpan = kBlock * kDictRows;
constexpr int kDictMaxHalo = 2048;

template <bool LDS_DICT>
__global__ void __launch_bounds__(kBlock)
k_spmv_dict(int64_t n, int64_t n_ext, const uint16_t *__restrict__ pattern, int patterns, int stride, int halo,
            const int32_t *__restrict__ dlen, const int32_t *__restrict__ doff, const double *__restrict__ dval,
            const double *__restrict__ x, const double *__restrict__ xdot, double *__restrict__ y,
            double *__restrict__ partials, int64_t dot_lo, int64_t dot_hi) {
    extern __shared__ double win[];  // kDictSpan + 2 * halo doubles, then (LDS_DICT) the dictionary
    __shared__ double red[kBlock / 64];
    double *s_val = win + kDictSpan + 2 * halo;
    int32_t *s_off = reinterpret_cast<int32_t *>(s_val + patterns * stride);
    int32_t *s_len = s_off + patterns * stride;
    if (LDS_DICT) {
        for (int i = threadIdx.x; i < patterns * stride; i += kBlock) {
            s_val[i] = dval[i];
            s_off[i] = doff[i];
        }
        for (int i = threadIdx.x; i < patterns; i += kBlock) s_len[i] = dlen[i];
    }
    const double *tv = LDS_DICT ? s_val : dval;
    const int32_t *to = LDS_DICT ? s_off : doff;
    const int32_t *tl = LDS_DICT ? s_len : dlen;
    const int wlen = kDictSpan + 2 * halo;
    const int64_t chunks = (n + kDictSpan - 1) / kDictSpan;
    const int64_t per_xcd = (chunks + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    int64_t c_end = (xcd + 1) * per_xcd;
    if (c_end > chunks) c_end = chunks;
    double acc = 0.0;
    for (int64_t c = xcd * per_xcd + slot; c < c_end; c += slots) {
        const int64_t row0 = c * kDictSpan;
        __syncthreads();  // the previous step's readers are done (and the dictionary copy is complete)
        for (int i = threadIdx.x; i < wlen; i += kBlock) {
            const int64_t g = row0 - halo + i;
            win[i] = (g >= 0 && g < n_ext) ? x[g] : 0.0;
        }

        int p[kDictRows], len[kDictRows];
#pragma unroll
        for (int u = 0; u < kDictRows; ++u) {
            const int64_t r = row0 + u * kBlock + threadIdx.x;
            p[u] = r < n ? (int)pattern[r] : -1;
        }
        __syncthreads();
        int maxlen = 0;
#pragma unroll
        for (int u = 0; u < kDictRows; ++u) {
            len[u] = p[u] >= 0 ? tl[p[u]] : 0;
            maxlen = max(maxlen, len[u]);
        }
        double sum[kDictRows];
#pragma unroll
        for (int u = 0; u < kDictRows; ++u) sum[u] = 0.0;
        for (int j = 0; j < maxlen; ++j) {
#pragma unroll
            for (int u = 0; u < kDictRows; ++u)
                if (j < len[u]) {
                    const int off = to[p[u] * stride + j];
                    const int local = u * kBlock + (int)threadIdx.x;
                    const double xv = (off >= -halo && off <= halo) ? win[local + halo + off] : x[row0 + local + off];
                    sum[u] += tv[p[u] * stride + j] * xv;
                }
        }
#pragma unroll
        for (int u = 0; u < kDictRows; ++u)
            if (p[u] >= 0) {
                const int64_t r = row0 + u * kBlock + threadIdx.x;
                __builtin_nontemporal_store(sum[u], y + r);
                // rows outside [dot_lo, dot_hi) are recomputed later (their remote entries were still
                // on their way): whatever they hold now must not reach the sum
                const double t = (xdot == x ? win[u * kBlock + (int)threadIdx.x + halo] : xdot[r]) * sum[u];
                acc += (r >= dot_lo && r < dot_hi) ? t : 0.0;  // a select: a NaN in t goes no further
            }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The same product with every load from memory issued one workgroup step ahead of its use and next to
// no arithmetic besides the sums (the form above spends 24 vector instructions per entry on address
// arithmetic, range tests and selects, and is bound by their issue rate, not by memory).  While a step's
// rows are summed out of LDS, the next step's window, pattern numbers and far entries are on their way
// into registers, through buffer descriptors (32-bit offsets, no clamping: what lies outside x reads as
// zero and is never used).  The far entries are fetched for every row at the (at most kDictFar) distinct
// far offsets the dictionary holds, whatever the row's pattern turns out to be, so they do not wait for
// the pattern numbers; they go to LDS regions of their own behind the window, and a region of zeros
// stands in for the entries a short row does not have.  The dictionary is rewritten once per workgroup
// into (value, LDS index relative to the row) pairs, so an entry costs three LDS reads, one add and one
// multiply-add.  Register sets swap roles from step to step (a copy of a register whose load is in
// flight would make the compiler wait for it).  WPT = window doubles per thread.
constexpr int kDictFar = 2;
typedef double dict_d2 __attribute__((ext_vector_type(2)));
typedef int dict_i4 __attribute__((ext_vector_type(4)));
static_assert(kDictRows == 4 && kDictFar == 2, "the register sets below are sized for four rows and two far offsets");

typedef unsigned int dict_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int dict_u4 __attribute__((ext_vector_type(4)));

static size_t dict_ahead_lds_bytes(int halo, int patterns, int stride) {
    return sizeof(double) * (size_t)(kDictSpan + 2 * halo + (kDictFar + 1) * kDictSpan) + (size_t)patterns * stride * 12 +
           sizeof(uint16_t) * kDictSpan + 8;
}

// WIDE (far offsets and the halo even, so that no 16-byte load straddles the start of x): the window
// and the far entries are loaded two doubles, the pattern numbers four rows per instruction, and the
// sums of neighbouring lanes are paired for 16-byte stores -- ten memory instructions per thread and step
// in place of twenty-two (this kernel's loads are bound by the rate at which the addresser takes
// instructions, 64 addresses each whatever their width).  The pattern numbers go through LDS to the
// lanes that sum their rows.
struct DictRegs {
    dict_i4 p;            // narrow: pattern numbers of this thread's four rows
    dict_d2 xf[4];        // narrow: [row](far0, far1); wide: [half * 2 + far](rows 2i, 2i + 1)
    dict_u2 pq;           // wide: pattern numbers of rows 4 tid .. 4 tid + 3
};

template <int WPT, bool WIDE, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_spmv_dict_ahead(int64_t n, int64_t n_ext, const uint16_t *__restrict__ pattern, int patterns, int stride, int halo,
                  const int32_t *__restrict__ dlen, const int32_t *__restrict__ doff, const double *__restrict__ dval,
                  int far0, int far1, const double *__restrict__ x, const double *__restrict__ xdot,
                  double *__restrict__ y, double *__restrict__ partials, int64_t dot_lo, int64_t dot_hi) {
    extern __shared__ double win[];  // window | far entries (kDictFar regions) | zeros | dictionary values | indices | patterns
    constexpr int ROWS = kDictSpan / BLOCK;   // rows per thread: 4 (256 threads) or 2 (512 threads)
    static_assert(ROWS == 2 || ROWS == 4, "a step covers kDictSpan rows with 256 or 512 threads");
    __shared__ double red[BLOCK / 64];
    const int tid = threadIdx.x;
    const int wlen = kDictSpan + 2 * halo;
    double *s_far = win + wlen;
    double *s_zero = s_far + kDictFar * kDictSpan;
    double *s_val = s_zero + kDictSpan;
    int32_t *s_idx = reinterpret_cast<int32_t *>(s_val + patterns * stride);
    uint16_t *s_pat = reinterpret_cast<uint16_t *>(s_idx + patterns * stride + ((patterns * stride) & 1));
    for (int i = tid; i < patterns * stride; i += BLOCK) {
        const int off = doff[i];
        const bool absent = i % stride >= dlen[i / stride];
        int e;
        if (absent) e = (int)(s_zero - win);                                     // no such entry: zero times zero
        else if (off >= -halo && off <= halo) e = halo + off;                    // in the window
        else e = wlen + (off == far0 ? 0 : kDictSpan);                           // in a far region
        s_idx[i] = e;
        s_val[i] = absent ? -0.0 : dval[i];   // (-0.0) x (+0.0) added to a sum leaves it as it is, signed zeros included
    }
    for (int i = tid; i < kDictSpan; i += BLOCK) s_zero[i] = 0.0;
    const int64_t chunks = (n + kDictSpan - 1) / kDictSpan;
    const int64_t per_xcd = (chunks + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    int64_t c_end = (xcd + 1) * per_xcd;
    if (c_end > chunks) c_end = chunks;
    const int64_t c_first = xcd * per_xcd + slot;
    // n_ext < 2^29 and n < 2^31 (checked by the launcher): byte offsets fit 32 bits.  The pattern array's
    // descriptor ends at the next multiple of four bytes (inside the allocation).
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)(uint32_t)((uint64_t)n_ext << 3), 0x00020000);
    const __amdgpu_buffer_rsrc_t rpat = __builtin_amdgcn_make_buffer_rsrc(
        (void *)pattern, 0, (int)(uint32_t)((((uint64_t)n << 1) + 3) & ~(uint64_t)3), 0x00020000);
    double acc = 0.0;
    dict_d2 W[WPT / 2];

    auto load_x = [&](int64_t index) -> double {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, (int)((uint32_t)index << 3), 0, 0));
    };
    auto load_x2 = [&](int64_t index) -> dict_d2 {
        return __builtin_bit_cast(dict_d2, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((uint32_t)index << 3), 0, 0));
    };
    auto load_ahead = [&](int64_t row0, DictRegs &R) {
        if constexpr (WIDE) {
#pragma unroll
            for (int k = 0; k < WPT / 2; ++k) W[k] = load_x2(row0 - halo + 2 * (k * BLOCK + tid));
            if constexpr (ROWS == 4)
                R.pq = __builtin_amdgcn_raw_buffer_load_b64(rpat, (int)((uint32_t)(row0 + 4 * tid) << 1), 0, 0);
            else
                R.pq.x = __builtin_amdgcn_raw_buffer_load_b32(rpat, (int)((uint32_t)(row0 + 2 * tid) << 1), 0, 0);
#pragma unroll
            for (int h = 0; h < ROWS / 2; ++h) {
                const int64_t r = row0 + 2 * (h * BLOCK + tid);
                R.xf[h * 2 + 0] = load_x2(r + far0);
                R.xf[h * 2 + 1] = load_x2(r + far1);
            }
        } else {
#pragma unroll
            for (int k = 0; k < WPT / 2; ++k)
                W[k] = dict_d2{load_x(row0 - halo + (2 * k) * BLOCK + tid), load_x(row0 - halo + (2 * k + 1) * BLOCK + tid)};
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const int64_t r = row0 + u * BLOCK + tid;
                R.p[u] = (int)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rpat, (int)((uint32_t)r << 1), 0, 0);
                R.xf[u] = dict_d2{load_x(r + far0), load_x(r + far1)};
            }
        }
    };
    auto step = [&](int64_t c, DictRegs &R, DictRegs &Rn) {
        const int64_t row0 = c * kDictSpan;
        __syncthreads();  // the previous step's readers are done (and the dictionary is in place)
        if constexpr (WIDE) {
#pragma unroll
            for (int k = 0; k < WPT / 2; ++k)
                if (2 * (k * BLOCK + tid) < wlen) *reinterpret_cast<dict_d2 *>(win + 2 * (k * BLOCK + tid)) = W[k];
            if constexpr (ROWS == 4) *reinterpret_cast<dict_u2 *>(s_pat + 4 * tid) = R.pq;
            else *reinterpret_cast<uint32_t *>(s_pat + 2 * tid) = R.pq.x;
#pragma unroll
            for (int h = 0; h < ROWS / 2; ++h) {
                *reinterpret_cast<dict_d2 *>(s_far + 2 * (h * BLOCK + tid)) = R.xf[h * 2 + 0];
                *reinterpret_cast<dict_d2 *>(s_far + kDictSpan + 2 * (h * BLOCK + tid)) = R.xf[h * 2 + 1];
            }
        } else {
#pragma unroll
            for (int k = 0; k < WPT / 2; ++k) {
                if ((2 * k) * BLOCK + tid < wlen) win[(2 * k) * BLOCK + tid] = W[k].x;
                if ((2 * k + 1) * BLOCK + tid < wlen) win[(2 * k + 1) * BLOCK + tid] = W[k].y;
            }
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                s_far[u * BLOCK + tid] = R.xf[u].x;
                s_far[kDictSpan + u * BLOCK + tid] = R.xf[u].y;
            }
        }
        // the next step's loads (the last step loads its own rows again)
        const int64_t cn = c + slots < c_end ? c + slots : c;
        load_ahead(cn * kDictSpan, Rn);
        __syncthreads();
        int base[ROWS];
        double sum[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            base[u] = (WIDE ? (int)s_pat[u * BLOCK + tid] : R.p[u]) * stride;
            sum[u] = 0.0;
        }
        for (int j = 0; j < stride; ++j) {
#pragma unroll
            for (int u = 0; u < ROWS; ++u)
                sum[u] += s_val[base[u] + j] * win[u * BLOCK + tid + s_idx[base[u] + j]];
        }
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            const int64_t r = row0 + u * BLOCK + tid;
            if (r < n) {
                if constexpr (!WIDE) __builtin_nontemporal_store(sum[u], y + r);
                double xd = win[u * BLOCK + tid + halo];
                if (xdot != x) xd = xdot[r];
                const double t = xd * sum[u];
                acc += (r >= dot_lo && r < dot_hi) ? t : 0.0;  // a select: a NaN in t goes no further
            }
        }
        if constexpr (WIDE) {
            // lanes t and t ^ 1 pair their sums: the even lane stores rows (t, t + 1) of the first half of the
            // row sets, the odd lane those of the second half
            const bool odd = tid & 1;
#pragma unroll
            for (int q = 0; q < ROWS / 2; ++q) {
                const double mine = odd ? sum[ROWS / 2 + q] : sum[q];
                const double got = __shfl_xor(odd ? sum[q] : sum[ROWS / 2 + q], 1);
                const dict_d2 s2 = odd ? dict_d2{got, mine} : dict_d2{mine, got};
                const int64_t r2 = row0 + ((odd ? ROWS / 2 : 0) + q) * BLOCK + (tid & ~1);
                if (r2 + 1 < n) __builtin_nontemporal_store(s2, reinterpret_cast<dict_d2 *>(y + r2));
                else if (r2 < n) __builtin_nontemporal_store(s2.x, y + r2);
            }
        }
    };

    if (c_first < c_end) {
        DictRegs RA, RB;
        load_ahead(c_first * kDictSpan, RA);
        int64_t c = c_first;
        for (; c + slots < c_end; c += 2 * slots) {
            step(c, RA, RB);
            step(c + slots, RB, RA);
        }
        if (c < c_end) step(c, RA, RB);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = (red[0] + red[1]) + (red[2] + red[3]);
        if constexpr (BLOCK == 512) t += (red[4] + red[5]) + (red[6] + red[7]);
        partials[blockIdx.x] = t;
    }
}

// Grids in three dimensions: the far offsets lie at -F and +F (one plane of the grid back and ahead) or within
// the window's halo of them.  A workgroup that walks from plane to plane -- rows [c, c + kDictSpan) of plane z, then the same rows
// of plane z + 1, F rows further -- finds its far entries in the windows of the step before and the
// step after.  Three windows stay in LDS (a ring), every step loads
// one, and x is fetched 1.5 times per product (the windows' halos) in place of 3.5 times; the form above
// is bound by the rate of L1 misses, which this one halves.  The planes are cut into segments so that
// every CU has its workgroups; a segment loads one plane before and one after its own.  Index tables
// for the three positions of the ring are built once per workgroup; the step is instantiated per position.
constexpr int kPlanesBlock = 512;

// Sum of parts[0 .. count) over the workgroup, the same in every thread, in the order in which the recurrence and
// closing kernels add such partial sums up (kernels.hip.h: sum_partials over 256 threads -- a thread's entries 256
// apart, a butterfly over the wavefront, the four wavefronts' sums in pairs): beta_{k-1} as this walk derives it and
// as k_lazy_flush records it at the end of a run are then the same bits, and a run continues bit for bit across
// its parts.
template <int BLOCK>
__device__ __forceinline__ double planes_sum_partials(const double *__restrict__ parts, int count, double *red) {
    double s = 0.0;
    if (threadIdx.x < 256)
        for (int i = threadIdx.x; i < count; i += 256) s += parts[i];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

static size_t dict_planes_lds_bytes(int halo, int patterns, int stride, int span = kDictSpan) {
    return sizeof(double) * (size_t)(3 * (span + 2 * halo) + span) + (size_t)patterns * stride * (8 + 3 * 4) +
           sizeof(uint16_t) * span + 16;
}

// CRS_VALUES: the row-wise CRS product as a plane walk.  When the rows' column OFFSETS repeat as patterns although
// their values do not (a stencil with variable coefficients), the column array is replaced by one pattern number
// per row and x is taken from the ring of windows just the same; the values are read from the CRS array where
// they lie, through the row offsets (src/crs.c:155-178 reads values, column indices and row offsets: 16 bytes
// per entry; this form 8 bytes per entry plus 6 per row).  Rows of up to eight entries; sums left to right.
// CACHE (dictionary form only): 0 every row reads its pattern's values and window offsets from the LDS tables; 1 a
// wavefront whose rows share one pattern keeps the values in registers; 2 the values and the offsets of the three ring
// positions (40 registers more: two workgroups per CU where the other forms have three).
// SPAN: rows of a plane per step, 1024 or 2048, two per thread.  A step loads SPAN + 2 halo doubles of x for SPAN rows:
// the longer span re-reads less (512^3: 1.5 times x in place of 2), at the price of LDS (one workgroup of 1024
// threads per CU where the short span has two or three of 512).
// MODE: 0 the product (y and the partial sums of xdot . y); 1 the partial sums only, nothing written (a launch that
// only reads); 2 the product with the recurrence in its epilogue: y never exists, what is written is
// u_{k+1} = (s_k A u_k - c_p u_{k-1}) - c_k u_k, and the partial sums are those of ||u_{k+1}||^2.  1 then 2 make an
// iteration without reorthogonalisation out of 36 bytes per row where the product followed by k_recur_lazy moves 50,
// and out of one launch that writes where those are two.
template <int WPT, bool CRS_VALUES, int CACHE, int SPAN, int MODE = 0>
__global__ void __launch_bounds__(SPAN / 2)
k_spmv_dict_planes(int64_t n, int64_t n_ext, int64_t nnz, const int32_t *__restrict__ rp, const double *__restrict__ val, const uint16_t *__restrict__ pattern, int patterns, int stride, int halo,
                   const int32_t *__restrict__ dlen, const int32_t *__restrict__ doff, const double *__restrict__ dval,
                   int far_a, int far_b, int plane_rows, int cols, int planes, int seg_len, int segs,
                   const double *__restrict__ x,
                   const double *__restrict__ xdot, double *__restrict__ y, double *__restrict__ partials,
                   int64_t dot_lo, int64_t dot_hi, DictRecur rc) {
    constexpr int BLOCK = SPAN / 2, ROWS = SPAN / BLOCK;
    static_assert(ROWS == 2, "two rows per thread");
    static_assert(MODE == 0 || (!CRS_VALUES && CACHE == 0), "the recurrence rides with the plain dictionary form");
    extern __shared__ double win[];  // three windows | zeros | dictionary values | three index tables | patterns
    __shared__ double red[BLOCK / 64];
    const int tid = threadIdx.x;
    const int wlen = SPAN + 2 * halo;
    const int tab = patterns * stride;
    double *s_zero = win + 3 * wlen;
    double *s_val = s_zero + SPAN;
    int32_t *s_idx = reinterpret_cast<int32_t *>(s_val + tab);
    uint16_t *s_pat = reinterpret_cast<uint16_t *>(s_idx + 3 * tab + ((3 * tab) & 1));
    int32_t *s_len = reinterpret_cast<int32_t *>(s_pat + SPAN);   // CRS_VALUES only
    if (CRS_VALUES)
        for (int i = tid; i < patterns; i += BLOCK) s_len[i] = dlen[i];
    for (int i = tid; i < tab; i += BLOCK) {
        const int off = doff[i];
        const bool absent = i % stride >= dlen[i / stride];
        s_val[i] = absent ? -0.0 : dval[i];   // (-0.0) x (+0.0) added to a sum leaves it as it is, signed zeros included
#pragma unroll
        for (int rot = 0; rot < 3; ++rot) {
            // ring slot `rot` holds this step's window, the slot before it the previous plane's, the slot after it the next one's
            int e;
            if (absent) e = 3 * wlen;
            else if (off >= -halo && off <= halo) e = rot * wlen + halo + off;
            else if (off == far_a) e = ((rot + 2) % 3) * wlen + halo;                       // a slab's remote plane before
            else if (off == far_b) e = ((rot + 1) % 3) * wlen + halo;                       // ... and after
            else if (off < 0) e = ((rot + 2) % 3) * wlen + halo + (off + plane_rows);       // around -F: the window F back
            else e = ((rot + 1) % 3) * wlen + halo + (off - plane_rows);                    // around +F: the window F ahead
            s_idx[rot * tab + i] = e;
        }
    }
    for (int i = tid; i < SPAN; i += BLOCK) s_zero[i] = 0.0;
    // work item = (column of steps, segment of planes); XCD b % 8 takes a contiguous eighth of the items
    const int64_t items = (int64_t)cols * segs;
    const int64_t per_xcd = (items + 7) >> 3;
    const int64_t item = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const bool live = (int64_t)(blockIdx.x >> 3) < per_xcd && item < items;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)(uint32_t)((uint64_t)n_ext << 3), 0x00020000);
    const __amdgpu_buffer_rsrc_t rpat = __builtin_amdgcn_make_buffer_rsrc(
        (void *)pattern, 0, (int)(uint32_t)((((uint64_t)n << 1) + 3) & ~(uint64_t)3), 0x00020000);
    // nnz < 2^29 (checked by the launcher): the values' byte offsets fit 32 bits; what lies past the last entry
    // reads as zero
    const __amdgpu_buffer_rsrc_t rval =
        __builtin_amdgcn_make_buffer_rsrc((void *)val, 0, (int)(uint32_t)((uint64_t)nnz << 3), 0x00020000);
    double acc = 0.0;
    double s_k = 1.0, c_p = 0.0, c_k = 0.0;  // MODE 2: the recurrence's scalars, as k_recur_lazy derives them
    if constexpr (MODE == 2) {
        const double dot = planes_sum_partials<BLOCK>(rc.dot_parts, rc.dot_count, red);
        double beta_km1 = 0.0, s_km1 = 1.0;
        if (rc.k > 0) {
            beta_km1 = sqrt(planes_sum_partials<BLOCK>(rc.norm_parts, rc.norm_count, red));
            s_k = 1.0 / beta_km1;
            if (rc.k > 1) s_km1 = 1.0 / rc.beta_hist[rc.k - 2];
        }
        const double alpha = s_k * s_k * dot;
        c_p = beta_km1 * s_km1;
        c_k = alpha * s_k;
        if (blockIdx.x == 0 && tid == 0) {
            rc.alpha_hist[rc.k] = alpha;
            if (rc.host_hist) rc.host_hist[2 * rc.k] = alpha;
            if (rc.k > 0) {
                rc.beta_hist[rc.k - 1] = beta_km1;
                if (rc.host_hist) rc.host_hist[2 * (rc.k - 1) + 1] = beta_km1;
            }
        }
    }
    if (live) {
        const int col = (int)(item % cols), seg = (int)(item / cols);
        const int zb = seg * seg_len, ze = min(zb + seg_len, planes);
        dict_d2 W[WPT / 2];
        uint32_t pq;
        int32_t head[ROWS];   // where this thread's rows of the next step begin in the value array
        // the pattern this wavefront's rows shared last (its place in the tables) and its values
        int c_base = -1;
        double c_val[CACHE ? 8 : 1];
        int c_idx[CACHE == 2 ? 3 : 1][CACHE == 2 ? 8 : 1];
        const bool use_cache = !CRS_VALUES && CACHE != 0 && stride <= 8;
        // a slab's remote planes: x[r + far_a] for the rows r of plane 0, x[r + far_b] for those of the last plane
        // (a plane need not be a whole number of steps: its last column of steps has fewer rows, `lim`)
        const int lim = min(SPAN, plane_rows - col * SPAN);
        auto row_of = [&](int z) -> int64_t {
            int64_t row = (int64_t)z * plane_rows + (int64_t)col * SPAN;
            if (z < 0 && far_a != INT32_MIN) row = far_a + (int64_t)col * SPAN;
            if (z >= planes && far_b != INT32_MIN) row = far_b + (int64_t)(planes - 1) * plane_rows + (int64_t)col * SPAN;
            return row;
        };
        // a plane outside the grid, or rows past the end, read as zeros (or as rows no entry refers to)
        auto load_window = [&](int z) {
            const int64_t row0 = row_of(z);
#pragma unroll
            for (int k = 0; k < WPT / 2; ++k)
                W[k] = __builtin_bit_cast(dict_d2, __builtin_amdgcn_raw_buffer_load_b128(
                                                       rx, (int)((uint32_t)(row0 - halo + 2 * (k * BLOCK + tid)) << 3), 0, 0));
        };
        auto store_window = [&](int slot) {
#pragma unroll
            for (int k = 0; k < WPT / 2; ++k)
                if (2 * (k * BLOCK + tid) < wlen)
                    *reinterpret_cast<dict_d2 *>(win + slot * wlen + 2 * (k * BLOCK + tid)) = W[k];
        };
        auto load_patterns = [&](int z) {
            const int64_t row0 = row_of(z);
            pq = __builtin_amdgcn_raw_buffer_load_b32(rpat, (int)((uint32_t)(row0 + 2 * tid) << 1), 0, 0);
            if constexpr (CRS_VALUES) {
#pragma unroll
                for (int u = 0; u < ROWS; ++u) {
                    const int64_t r = row0 + u * BLOCK + tid;
                    head[u] = rp[r < n ? r : n];
                }
            }
        };
        auto step = [&](int z, auto rot_tag) {
            constexpr int ROT = decltype(rot_tag)::value;
            const int64_t row0 = row_of(z);
            __syncthreads();  // the previous step's readers are done
            store_window((ROT + 1) % 3);
            *reinterpret_cast<uint32_t *>(s_pat + 2 * tid) = pq;
            // CRS_VALUES: this step's values; a row's entries lie side by side in the CRS array, two per load
            dict_d2 V[ROWS][4];
            if constexpr (CRS_VALUES) {
#pragma unroll
                for (int u = 0; u < ROWS; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        V[u][q] = __builtin_bit_cast(dict_d2, __builtin_amdgcn_raw_buffer_load_b128(
                                                                  rval, (int)(((uint32_t)head[u] + 2 * q) << 3), 0, 0));
            }
            if (z + 1 < ze) {
                load_window(z + 2);
                load_patterns(z + 1);
            }
            double pv[ROWS];  // MODE 2: this thread's rows of u_{k-1}
            if constexpr (MODE == 2) {
#pragma unroll
                for (int u = 0; u < ROWS; ++u) {
                    const int64_t r = row0 + u * BLOCK + tid;
                    pv[u] = r < n ? __builtin_nontemporal_load(rc.uprev + r) : 0.0;
                }
            }
            __syncthreads();
            const int32_t *idx = s_idx + ROT * tab;
            int base[ROWS], len[ROWS];
            double sum[ROWS];
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const int pid = (int)s_pat[u * BLOCK + tid];
                base[u] = pid * stride;
                len[u] = CRS_VALUES ? s_len[pid] : 0;
                sum[u] = 0.0;
            }
            if constexpr (CRS_VALUES) {
                // up to eight entries per row (checked by the launcher), left to right; an entry the row does not
                // have multiplies -0.0 by the zeros' region
#pragma unroll
                for (int u = 0; u < ROWS; ++u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const double v = j < len[u] ? (j & 1 ? V[u][j >> 1].y : V[u][j >> 1].x) : -0.0;
                        const int e = j < stride ? idx[base[u] + j] : 3 * wlen;
                        sum[u] += v * win[u * BLOCK + tid + e];
                    }
                }
            } else {
                // Away from the grid's faces every row of a wavefront has the same pattern, and it is the pattern the
                // wavefront met one plane earlier: its values and window offsets then stay in registers (c_val, c_idx)
                // and a row costs its seven reads of x from LDS, not those plus fourteen reads of the tables.  Same
                // products in the same order: the entries past a pattern's end multiply -0.0 by the zeros' region here
                // as in the tables.
                const int p0 = __builtin_amdgcn_readfirstlane(base[0]);
                if (CACHE != 0 && use_cache && __all(base[0] == p0 && base[1] == p0)) {
                    if (p0 != c_base) {
                        c_base = p0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const bool in = j < stride;
                            if constexpr (CACHE != 0) c_val[j] = in ? s_val[p0 + j] : -0.0;
                            if constexpr (CACHE == 2) {
#pragma unroll
                                for (int rot = 0; rot < 3; ++rot) c_idx[rot][j] = in ? s_idx[rot * tab + p0 + j] : 3 * wlen;
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        int e;
                        if constexpr (CACHE == 2) e = c_idx[ROT][j];
                        else e = j < stride ? idx[p0 + j] : 3 * wlen;   // one address for all lanes: one bank, no conflict
                        const double v = c_val[CACHE ? j : 0];
#pragma unroll
                        for (int u = 0; u < ROWS; ++u) sum[u] += v * win[u * BLOCK + tid + e];
                    }
                } else {
                    for (int j = 0; j < stride; ++j) {
#pragma unroll
                        for (int u = 0; u < ROWS; ++u) sum[u] += s_val[base[u] + j] * win[u * BLOCK + tid + idx[base[u] + j]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < ROWS; ++u) {
                const int64_t r = row0 + u * BLOCK + tid;
                if constexpr (MODE == 2) {
                    const double xk = win[ROT * wlen + halo + u * BLOCK + tid];
                    sum[u] = (s_k * sum[u] - c_p * pv[u]) - c_k * xk;
                    acc += (r < n && u * BLOCK + tid < lim) ? sum[u] * sum[u] : 0.0;
                } else if (r < n && u * BLOCK + tid < lim) {
                    double xd = win[ROT * wlen + halo + u * BLOCK + tid];
                    if (xdot != x) xd = xdot[r];
                    const double t = xd * sum[u];
                    acc += (r >= dot_lo && r < dot_hi) ? t : 0.0;  // a select: a NaN in t goes no further
                }
            }
            if constexpr (MODE == 1) return;  // the sums were all this launch is for
            // lanes t and t ^ 1 pair their sums: the even lane stores rows (t, t + 1) of the first row set, the odd
            // lane those of the second
            const bool odd = tid & 1;
            const double mine = odd ? sum[1] : sum[0];
            const double got = __shfl_xor(odd ? sum[0] : sum[1], 1);
            const dict_d2 s2 = odd ? dict_d2{got, mine} : dict_d2{mine, got};
            const int l2 = (odd ? BLOCK : 0) + (tid & ~1);
            const int64_t r2 = row0 + l2;
            if (r2 + 1 < n && l2 + 1 < lim) __builtin_nontemporal_store(s2, reinterpret_cast<dict_d2 *>(y + r2));
            else if (r2 < n && l2 < lim) __builtin_nontemporal_store(s2.x, y + r2);
        };
        // the planes before and of the first step go to ring slots 0 and 1, the one after it waits in registers
        load_window(zb - 1);
        store_window(0);
        load_window(zb);
        store_window(1);
        load_window(zb + 1);
        load_patterns(zb);
        for (int z = zb; z < ze; z += 3) {
            step(z, std::integral_constant<int, 1>{});
            if (z + 1 < ze) step(z + 1, std::integral_constant<int, 2>{});
            if (z + 2 < ze) step(z + 2, std::integral_constant<int, 0>{});
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
        if constexpr (BLOCK == 1024) t += ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
        partials[blockIdx.x] = t;
    }
}

void Engine::free_dict() {
    (void)hipFree(d_pattern);
    (void)hipFree(d_dlen);
    (void)hipFree(d_doff);
    (void)hipFree(d_dval);
    d_pattern = nullptr;
    d_dlen = nullptr;
    d_doff = nullptr;
    d_dval = nullptr;
    dict_patterns = 0;
    dict_blocks = 0;
    dict_span = 0;
}

// dict_patterns > 0 afterwards when the matrix is a dictionary of at most 65535 row patterns of at
// most 32 entries; 0 when it is not.  Non-zero return on errors.
int Engine::build_dict(bool with_values) {
    dict_values = with_values;
    free_dict();
    if (nnz == 0) return 0;
    uint64_t *d_hash = nullptr, *d_sorted = nullptr, *d_uniq = nullptr;
    int *d_flags = nullptr, *d_count = nullptr, *d_rep = nullptr;
    void *d_tmp = nullptr, *d_tmp2 = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_hash);
        (void)hipFree(d_sorted);
        (void)hipFree(d_uniq);
        (void)hipFree(d_flags);
        (void)hipFree(d_count);
        (void)hipFree(d_rep);
        (void)hipFree(d_tmp);
        (void)hipFree(d_tmp2);
    };
#define DC_TRY(call)                                                                                   \
    do {                                                                                               \
        hipError_t st__ = (call);                                                                      \
        if (st__ != hipSuccess) {                                                                      \
            cleanup();                                                                                 \
            free_dict();                                                                               \
            return fail("HIP API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,            \
                        hipGetErrorString(st__), (int)st__);                                           \
        }                                                                                              \
    } while (0)
    DC_TRY(hipMalloc(&d_hash, sizeof(uint64_t) * n));
    DC_TRY(hipMalloc(&d_flags, sizeof(int) * 2));
    DC_TRY(hipMemsetAsync(d_flags, 0, sizeof(int) * 2, stream));
    const int rblocks = (int)std::min<int64_t>(max_blocks, (n + kBlock - 1) / kBlock);
    if (rp64)
        hipLaunchKernelGGL((k_row_hash<int64_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int64_t *)d_rp,
                           (const int32_t *)d_col, (const double *)d_val, d_hash, d_flags, (int)with_values);
    else
        hipLaunchKernelGGL((k_row_hash<int32_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int32_t *)d_rp,
                           (const int32_t *)d_col, (const double *)d_val, d_hash, d_flags, (int)with_values);
    DC_TRY(hipGetLastError());
    int flags[2] = {0, 0};
    DC_TRY(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
    DC_TRY(hipStreamSynchronize(stream));
    if (flags[0]) {  // a row longer than a pattern may be
        cleanup();
        return 0;
    }
    DC_TRY(hipMalloc(&d_sorted, sizeof(uint64_t) * n));
    DC_TRY(hipMalloc(&d_uniq, sizeof(uint64_t) * n));
    DC_TRY(hipMalloc(&d_count, sizeof(int)));
    size_t tmp_bytes = 0;
    DC_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_hash, d_sorted, (int)n, 0, 64, stream));
    DC_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    DC_TRY(hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_hash, d_sorted, (int)n, 0, 64, stream));
    size_t tmp2_bytes = 0;
    DC_TRY(hipcub::DeviceSelect::Unique(nullptr, tmp2_bytes, d_sorted, d_uniq, d_count, (int)n, stream));
    DC_TRY(hipMalloc(&d_tmp2, tmp2_bytes ? tmp2_bytes : 1));
    DC_TRY(hipcub::DeviceSelect::Unique(d_tmp2, tmp2_bytes, d_sorted, d_uniq, d_count, (int)n, stream));
    int count = 0;
    DC_TRY(hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, stream));
    DC_TRY(hipStreamSynchronize(stream));
    if (count < 1 || count > kDictMaxPatterns) {
        cleanup();
        return 0;
    }
    DC_TRY(hipMalloc(&d_pattern, sizeof(uint16_t) * n));
    DC_TRY(hipMalloc(&d_rep, sizeof(int) * 2 * count));
    DC_TRY(hipMemsetAsync(d_rep, 0x7F, sizeof(int) * count, stream));
    DC_TRY(hipMemsetAsync(d_rep + count, 0xFF, sizeof(int) * count, stream));
    hipLaunchKernelGGL(k_dict_assign, dim3(rblocks), dim3(kBlock), 0, stream, n, (const uint64_t *)d_hash,
                       (const uint64_t *)d_uniq, count, d_pattern, d_rep);
    // dictionary rows as long as the longest row of the matrix (known from the 64-row spans only as
    // a bound, so measured here: the largest length among the representatives)
    std::vector<int> rep_host((size_t)2 * count);
    DC_TRY(hipMemcpyAsync(rep_host.data(), d_rep, sizeof(int) * 2 * count, hipMemcpyDeviceToHost, stream));
    DC_TRY(hipStreamSynchronize(stream));
    int stride = 1;
    for (int p = 0; p < count; ++p) {
        int64_t ab[2] = {0, 0};
        if (rp64) {
            DC_TRY(hipMemcpy(ab, (const int64_t *)d_rp + rep_host[(size_t)p], sizeof(ab), hipMemcpyDeviceToHost));
        } else {
            int32_t t[2] = {0, 0};
            DC_TRY(hipMemcpy(t, (const int32_t *)d_rp + rep_host[(size_t)p], sizeof(t), hipMemcpyDeviceToHost));
            ab[0] = t[0];
            ab[1] = t[1];
        }
        stride = std::max(stride, (int)(ab[1] - ab[0]));
    }
    DC_TRY(hipMalloc(&d_dlen, sizeof(int32_t) * count));
    DC_TRY(hipMalloc(&d_doff, sizeof(int32_t) * (size_t)count * stride));
    DC_TRY(hipMalloc(&d_dval, sizeof(double) * (size_t)count * stride));
    const int pblocks = (count + kBlock - 1) / kBlock;
    if (rp64) {
        hipLaunchKernelGGL((k_dict_fill<int64_t>), dim3(pblocks), dim3(kBlock), 0, stream, count, (const int *)d_rep,
                           (const int64_t *)d_rp, (const int32_t *)d_col, (const double *)d_val, stride, d_dlen, d_doff,
                           d_dval);
        hipLaunchKernelGGL((k_dict_verify<int64_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int64_t *)d_rp,
                           (const int32_t *)d_col, (const double *)d_val, (const uint16_t *)d_pattern, stride,
                           (const int32_t *)d_dlen, (const int32_t *)d_doff, (const double *)d_dval, d_flags + 1,
                           (int)with_values);
    } else {
        hipLaunchKernelGGL((k_dict_fill<int32_t>), dim3(pblocks), dim3(kBlock), 0, stream, count, (const int *)d_rep,
                           (const int32_t *)d_rp, (const int32_t *)d_col, (const double *)d_val, stride, d_dlen, d_doff,
                           d_dval);
        hipLaunchKernelGGL((k_dict_verify<int32_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int32_t *)d_rp,
                           (const int32_t *)d_col, (const double *)d_val, (const uint16_t *)d_pattern, stride,
                           (const int32_t *)d_dlen, (const int32_t *)d_doff, (const double *)d_dval, d_flags + 1,
                           (int)with_values);
    }
    DC_TRY(hipGetLastError());
    DC_TRY(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
    // the dictionary's column offsets within reach set the halo of the LDS window; the far ones
    // (the +-N^2 neighbours of a 3-D grid) are gathered from memory (LDS windows for them were tried
    // and were slower: more LDS traffic and fewer resident workgroups)
    std::vector<int32_t> hlen((size_t)count), hoff((size_t)count * stride);
    DC_TRY(hipMemcpyAsync(hlen.data(), d_dlen, sizeof(int32_t) * count, hipMemcpyDeviceToHost, stream));
    DC_TRY(hipMemcpyAsync(hoff.data(), d_doff, sizeof(int32_t) * hoff.size(), hipMemcpyDeviceToHost, stream));
    DC_TRY(hipStreamSynchronize(stream));
    cleanup();
#undef DC_TRY
    dict_halo = 0;
    for (int p = 0; p < count; ++p)
        for (int j = 0; j < hlen[(size_t)p]; ++j) {
            const int off = hoff[(size_t)p * stride + j], mag = off < 0 ? -off : off;
            if (mag <= kDictMaxHalo) dict_halo = std::max(dict_halo, mag);
        }
    dict_halo = (dict_halo + 1) & ~1;  // even: the window then starts at an even index of x
    if (flags[1]) {  // two different rows behind one hash: keep the CRS kernels
        free_dict();
        return 0;
    }
    dict_patterns = count;
    dict_stride = stride;
    // the distinct offsets beyond the window: at most two of them let the SpMV fetch them ahead of the
    // pattern numbers
    dict_far_count = 0;
    dict_far[0] = dict_far[1] = 0;
    for (int p = 0; p < count && dict_far_count >= 0; ++p)
        for (int j = 0; j < hlen[(size_t)p]; ++j) {
            const int off = hoff[(size_t)p * stride + j], mag = off < 0 ? -off : off;
            if (mag <= dict_halo) continue;
            int f = 0;
            while (f < dict_far_count && dict_far[f] != off) ++f;
            if (f < dict_far_count) continue;
            if (dict_far_count == 2) {
                dict_far_count = -1;  // more than the ahead-of-time form holds
                break;
            }
            dict_far[dict_far_count++] = off;
        }
    // Plane walk: the far offsets lie within the window's halo of -F and +F for some even F (a 7-point grid
    // has -F and +F themselves, a 27-point one nine offsets around each; for an odd plane size F is the
    // next even number): x[r + o] is then found in the window of the rows F back or F ahead.  A row slab
    // of a partitioned 7-point grid holds up to two more far offsets: the rows of its first plane reach the
    // previous slab's last plane at x[r + A], those of its last plane the next slab's first plane at
    // x[r + B] (both stored behind the slab's own entries of x).  They are taken as a plane before the first
    // and a plane after the last, provided only rows of the first / last plane use them.
    // (an odd plane of a 27-point grid needs a halo one or two entries wider than the near offsets ask for)
    auto plan_planes = [&](int halo_w) -> int {   // returns the halo the far offsets would need, 0 if none will do
        dict_plane_f = 0;
        dict_plane_a = dict_plane_b = INT32_MIN;
        int need = 0;
        std::vector<int32_t> far;
        bool ok = true;
        for (int p = 0; p < count && ok; ++p)
            for (int j = 0; j < hlen[(size_t)p]; ++j) {
                const int off = hoff[(size_t)p * stride + j], mag = off < 0 ? -off : off;
                if (mag <= halo_w || std::find(far.begin(), far.end(), off) != far.end()) continue;
                far.push_back(off);
                if (far.size() > 64) ok = false;
            }
        int64_t lo = 0, hi = 0;
        int negatives = 0;
        for (int32_t o : far)
            if (o < 0) {
                lo = negatives ? std::min<int64_t>(lo, o) : o;
                hi = negatives ? std::max<int64_t>(hi, o) : o;
                ++negatives;
            }
        int64_t F = negatives ? ((-(lo + hi) / 2) + 1) & ~(int64_t)1 : 0;
        // No far offsets at all (grids in one or two dimensions, small ones in three): the row-wise CRS
        // product over a dictionary of offsets walks "planes" of 64 steps, or of a third of the rows, whose
        // neighbouring windows no entry refers to.  (The dictionary with values keeps its form without far loads.)
        if (far.empty() && !dict_values)
            F = std::min<int64_t>(64 * kDictSpan, n / 3 / kDictSpan * kDictSpan);
        if (ok && F > 0 && F * 3 <= n && F < INT32_MAX) {
            std::vector<int32_t> others;
            bool exact = negatives == 1 && -lo == F;
            for (int32_t o : far) {
                if (o < 0) {
                    need = std::max<int>(need, (int)std::min<int64_t>(std::abs((int64_t)o + F), INT32_MAX / 2));
                    if (std::abs((int64_t)o + F) > halo_w) ok = false;
                } else if (std::abs((int64_t)o - F) <= halo_w + 2) {
                    need = std::max<int>(need, (int)std::abs((int64_t)o - F));
                    if (std::abs((int64_t)o - F) > halo_w) ok = false;
                    exact = exact && o == F;
                } else {
                    others.push_back(o);
                }
            }
            if (!others.empty() && (!exact || others.size() > 2 || n % F != 0)) ok = false;
            for (size_t q = 0; q < others.size() && ok; ++q) {
                const int32_t o = others[q];
                int lowest = INT32_MAX, highest = -1;  // rows that hold this offset
                for (int p = 0; p < count; ++p)
                    for (int j = 0; j < hlen[(size_t)p]; ++j)
                        if (hoff[(size_t)p * stride + j] == o) {
                            lowest = std::min(lowest, rep_host[(size_t)p]);
                            highest = std::max(highest, rep_host[(size_t)count + p]);
                        }
                if (highest < F && dict_plane_a == INT32_MIN) dict_plane_a = o;
                else if (lowest >= n - F && dict_plane_b == INT32_MIN) dict_plane_b = o;
                else ok = false;
            }
            if (ok) dict_plane_f = (int32_t)F;
        }
        if (!dict_plane_f) dict_plane_a = dict_plane_b = INT32_MIN;
        return dict_plane_f ? 0 : need;
    };
    const int need = plan_planes(dict_halo);
    if (!dict_plane_f && need > dict_halo && need <= dict_halo + 2 && ((need + 1) & ~1) <= kDictMaxHalo) {
        const int keep = dict_halo;
        dict_halo = (need + 1) & ~1;
        plan_planes(dict_halo);
        if (!dict_plane_f) dict_halo = keep;
    }
    return 0;
}

// Which form of the dictionary SpMV this engine's dictionary runs on: 2 = plane walk, 1 = loads one step
// ahead, 0 = loads inside the step (see the kernels above; LANCZOS_HIP_DICT_PLANES=0 / LANCZOS_HIP_DICT_AHEAD=0
// switch the first two off for comparison).
int Engine::dict_form() const {
    static const bool ahead_env = !(getenv("LANCZOS_HIP_DICT_AHEAD") && atoi(getenv("LANCZOS_HIP_DICT_AHEAD")) == 0);
    static const bool planes_env = !(getenv("LANCZOS_HIP_DICT_PLANES") && atoi(getenv("LANCZOS_HIP_DICT_PLANES")) == 0);
    const bool fits = ahead_env && dict_patterns > 0 && dict_patterns * dict_stride <= kDictLdsEntries && n_ext > 0 &&
                      n_ext < ((int64_t)1 << 29) && n < ((int64_t)1 << 31);
    if (!fits) return 0;
    if (planes_env && dict_plane_f > 0 && !(dict_halo & 1) &&
        dict_planes_lds_bytes(dict_halo, dict_patterns, dict_stride) <= 65536 &&
        (n + dict_plane_f - 1) / dict_plane_f >= 3 && (dict_plane_f + kDictSpan - 1) / kDictSpan <= max_blocks)
        return 2;
    if (dict_far_count >= 0 && dict_ahead_lds_bytes(dict_halo, dict_patterns, dict_stride) <= 65536) return 1;
    return 0;
}

// The instantiations of a form share one signature, so the choice among them is a function pointer.
using DictAheadKernel = void (*)(int64_t, int64_t, const uint16_t *, int, int, int, const int32_t *, const int32_t *,
                                 const double *, int, int, const double *, const double *, double *, double *, int64_t,
                                 int64_t);
using DictPlanesKernel = void (*)(int64_t, int64_t, int64_t, const int32_t *, const double *, const uint16_t *, int, int,
                                  int, const int32_t *, const int32_t *, const double *, int, int, int, int, int, int, int,
                                  const double *, const double *, double *, double *, int64_t, int64_t, DictRecur);

// The plane walk's two knobs when the environment does not set them, from measurements on 256^3 and 512^3 (ms per SpMV,
// span 1024 / 2048 x tables / registers): 256^3 0.0713 0.0784 | 0.0718 0.0772; 512^3 0.534 0.515 | 0.510 0.539.
// Long grid lines take the long span (less of x read twice), nobody takes the register form by default.
static int kDictCacheDefault(int halo) { (void)halo; return 0; }
static int dict_span_default(int halo) { return halo >= 384 ? 2048 : kDictSpan; }

// window doubles per thread (even: loaded in pairs) for a window of `wlen` doubles and `block` threads
static int dict_window_per_thread(int wlen, int block) { return 2 * ((wlen / 2 + block - 1) / block); }

template <bool WIDE>
static DictAheadKernel dict_ahead_kernel(int block, int w) {
    if (block == 512)
        return w <= 4 ? k_spmv_dict_ahead<4, WIDE, 512> : w <= 6 ? k_spmv_dict_ahead<6, WIDE, 512> : k_spmv_dict_ahead<10, WIDE, 512>;
    return w <= 6    ? k_spmv_dict_ahead<6, WIDE, 256>
           : w <= 8  ? k_spmv_dict_ahead<8, WIDE, 256>
           : w <= 12 ? k_spmv_dict_ahead<12, WIDE, 256>
                     : k_spmv_dict_ahead<20, WIDE, 256>;
}

// workgroups of `kernel` that fit a CU at once
template <typename K>
static int dict_resident(K kernel, int block, size_t lds) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block, lds) != hipSuccess || occ < 1) occ = 1;
    return occ;
}

static size_t pdict_lds_bytes(int halo, int patterns, int stride) {
    return dict_planes_lds_bytes(halo, patterns, stride) + sizeof(int32_t) * (size_t)patterns + 8;
}

// Whether the row-wise CRS product of this engine can run as a plane walk over a dictionary of column
// offsets (build_dict(false) has run): rows of up to eight entries, 32-bit row offsets, values within
// 32-bit byte offsets, no remote planes (row slabs take the product in parts, which this form does not).
bool Engine::pdict_usable() const {
    const bool env = !(getenv("LANCZOS_HIP_ROWS_PLANES") && atoi(getenv("LANCZOS_HIP_ROWS_PLANES")) == 0);  // read per engine
    return env && dict_patterns > 0 && !dict_values && dict_stride <= 8 && !rp64 && nnz < ((int64_t)1 << 29) &&
           n_ext == n && dict_plane_a == INT32_MIN && dict_plane_b == INT32_MIN && dict_form() == 2 &&
           pdict_lds_bytes(dict_halo, dict_patterns, dict_stride) <= 65536;
}

int Engine::launch_spmv_pdict(const double *x, const double *xdot, double *y, int *nparts) {
    const int wlen = kDictSpan + 2 * dict_halo;
    const int cols = (dict_plane_f + kDictSpan - 1) / kDictSpan;
    const int planes = (int)((n + dict_plane_f - 1) / dict_plane_f);
    const size_t lds = pdict_lds_bytes(dict_halo, dict_patterns, dict_stride);
    const int w = dict_window_per_thread(wlen, kPlanesBlock);
    const DictPlanesKernel kernel = w <= 4   ? k_spmv_dict_planes<4, true, 0, kDictSpan>
                                 : w <= 6 ? k_spmv_dict_planes<6, true, 0, kDictSpan>
                                          : k_spmv_dict_planes<10, true, 0, kDictSpan>;
    if (dict_blocks == 0) {
        const int64_t want = std::min<int64_t>((int64_t)cus * dict_resident(kernel, kPlanesBlock, lds), max_blocks);
        dict_segs = (int)std::max<int64_t>(1, std::min<int64_t>(planes / 3, want / cols));
        dict_seg_len = (planes + dict_segs - 1) / dict_segs;
        dict_segs = (planes + dict_seg_len - 1) / dict_seg_len;
        dict_blocks = (int)(((int64_t)cols * dict_segs + 7) / 8 * 8);
    }
    *nparts = dict_blocks;
    hipLaunchKernelGGL(kernel, dim3(dict_blocks), dim3(kPlanesBlock), lds, stream, n, n_ext, nnz, (const int32_t *)d_rp,
                       (const double *)d_val, (const uint16_t *)d_pattern, dict_patterns, dict_stride, dict_halo,
                       (const int32_t *)d_dlen, (const int32_t *)d_doff, (const double *)d_dval, (int)dict_plane_a,
                       (int)dict_plane_b, (int)dict_plane_f, cols, planes, dict_seg_len, dict_segs, x, xdot, y, d_partials,
                       (int64_t)0, n, DictRecur{});
    const hipError_t st = hipGetLastError();
    if (st != hipSuccess) return fail("lanczos_hip: %s", hipGetErrorString(st));
    return 0;
}

// mode 0: y = A x and the partial sums of xdot . y in d_partials.  Modes 1 and 2 (the plane walk only, see
// k_spmv_dict_planes): 1 the partial sums alone; 2 y = u_{k+1} by the recurrence in `recur`, the partial sums of its
// squared norm in `parts_out`.
int Engine::launch_spmv_dict(const double *x, const double *xdot, double *y, int *nparts, int64_t dot_lo, int64_t dot_hi,
                             int mode, const void *recur, double *parts_out) {
    if (dot_hi < 0) dot_hi = n;
    const int64_t chunks = (n + kDictSpan - 1) / kDictSpan;
    const int wlen = kDictSpan + 2 * dict_halo;
    const uint16_t *pat = (const uint16_t *)d_pattern;
    const int32_t *dl = (const int32_t *)d_dlen, *dof = (const int32_t *)d_doff;
    const double *dv = (const double *)d_dval;
    const int form = dict_form();
    if (form == 2) {
        // a grid in three dimensions, walked plane by plane; the grid is one round of resident workgroups, the
        // planes cut into as many segments as that takes (at least three planes each)
        if (dict_span == 0) {
            // rows per step, chosen once per engine: LANCZOS_HIP_DICT_SPAN = 1024 / 2048, else by the in-plane reach
            const char *span_env = getenv("LANCZOS_HIP_DICT_SPAN");
            dict_span = span_env && atoi(span_env) == 2048 ? 2048 : span_env && atoi(span_env) == 1024 ? 1024 : dict_span_default(dict_halo);
            if (dict_planes_lds_bytes(dict_halo, dict_patterns, dict_stride, dict_span) > 160 * 1024 - 512) dict_span = kDictSpan;
            // a wavefront whose rows share one pattern can keep that pattern's values and window offsets in registers:
            // LANCZOS_HIP_DICT_CACHE = 0 / 1 / 2 (tables / values / values and offsets; the same bits)
            const char *cache_env = getenv("LANCZOS_HIP_DICT_CACHE");
            dict_cache = cache_env && *cache_env ? std::min(std::max(atoi(cache_env), 0), 2) : kDictCacheDefault(dict_halo);
        }
        const int span = dict_span, block = span / 2, cache = dict_cache;
        const int wlen_p = span + 2 * dict_halo;
        const int cols = (dict_plane_f + span - 1) / span;
        const int planes = (int)((n + dict_plane_f - 1) / dict_plane_f);
        const size_t lds = dict_planes_lds_bytes(dict_halo, dict_patterns, dict_stride, span);
        const int w = dict_window_per_thread(wlen_p, block);
#define LZ_PLANES_C(W_, S_) \
    (mode == 2 ? k_spmv_dict_planes<W_, false, 0, S_, 2> : mode == 1 ? k_spmv_dict_planes<W_, false, 0, S_, 1> : \
     cache == 2 ? k_spmv_dict_planes<W_, false, 2, S_> : cache == 1 ? k_spmv_dict_planes<W_, false, 1, S_> : k_spmv_dict_planes<W_, false, 0, S_>)
#define LZ_PLANES(W_) (span == 2048 ? LZ_PLANES_C(W_, 2048) : LZ_PLANES_C(W_, 1024))
        const DictPlanesKernel kernel = w <= 4 ? LZ_PLANES(4) : w <= 6 ? LZ_PLANES(6) : LZ_PLANES(10);
#undef LZ_PLANES
#undef LZ_PLANES_C
        const DictRecur rc = recur ? *static_cast<const DictRecur *>(recur) : DictRecur{};
        if (dict_blocks == 0) {
            const int64_t want = std::min<int64_t>((int64_t)cus * dict_resident(kernel, block, lds), max_blocks);
            dict_segs = (int)std::max<int64_t>(1, std::min<int64_t>(planes / 3, want / cols));
            dict_seg_len = (planes + dict_segs - 1) / dict_segs;
            dict_segs = (planes + dict_seg_len - 1) / dict_seg_len;
            dict_blocks = (int)(((int64_t)cols * dict_segs + 7) / 8 * 8);
        }
        *nparts = dict_blocks;
        hipLaunchKernelGGL(kernel, dim3(dict_blocks), dim3(block), lds, stream, n, n_ext, nnz,
                           (const int32_t *)nullptr, (const double *)nullptr, pat, dict_patterns, dict_stride, dict_halo,
                           dl, dof, dv, (int)dict_plane_a, (int)dict_plane_b, (int)dict_plane_f,
                           cols, planes, dict_seg_len, dict_segs, x, xdot, y, mode == 2 ? parts_out : d_partials, dot_lo,
                           dot_hi, rc);
    } else if (form == 1) {
        if (mode != 0) return fail("lanczos_hip: the recurrence rides with the plane-walking form only");
        // loads one step ahead; 16-byte loads when no pair of doubles can straddle the start of x
        // (LANCZOS_HIP_DICT_WIDE=0: never); 512 threads, two rows per thread (LANCZOS_HIP_DICT_BLOCK=256: four)
        static const bool wide_env = !(getenv("LANCZOS_HIP_DICT_WIDE") && atoi(getenv("LANCZOS_HIP_DICT_WIDE")) == 0);
        static const int block = getenv("LANCZOS_HIP_DICT_BLOCK") && atoi(getenv("LANCZOS_HIP_DICT_BLOCK")) == 256 ? 256 : 512;
        const bool wide = wide_env && !(dict_far[0] & 1) && !(dict_far[1] & 1) && !(dict_halo & 1);
        const size_t lds = dict_ahead_lds_bytes(dict_halo, dict_patterns, dict_stride);
        const int w = dict_window_per_thread(wlen, block);
        const DictAheadKernel kernel = wide ? dict_ahead_kernel<true>(block, w) : dict_ahead_kernel<false>(block, w);
        if (dict_blocks == 0)
            dict_blocks = (int)std::min<int64_t>(
                std::min<int64_t>((int64_t)cus * dict_resident(kernel, block, lds), max_blocks), (chunks + 7) / 8 * 8);
        *nparts = dict_blocks;
        hipLaunchKernelGGL(kernel, dim3(dict_blocks), dim3(block), lds, stream, n, n_ext, pat, dict_patterns, dict_stride,
                           dict_halo, dl, dof, dv, (int)dict_far[0], (int)dict_far[1], x, xdot, y, d_partials, dot_lo,
                           dot_hi);
    } else {
        // loads inside the step; the dictionary in LDS when it fits
        const int blocks = (int)std::min<int64_t>(max_blocks, (chunks + 7) / 8 * 8);
        *nparts = blocks;
        const bool in_lds = dict_patterns * dict_stride <= kDictLdsEntries;
        const size_t lds = sizeof(double) * (size_t)wlen +
                           (in_lds ? (size_t)dict_patterns * dict_stride * 12 + (size_t)dict_patterns * 4 : 0);
        hipLaunchKernelGGL(in_lds ? k_spmv_dict<true> : k_spmv_dict<false>, dim3(blocks), dim3(kBlock), lds, stream, n,
                           n_ext, pat, dict_patterns, dict_stride, dict_halo, dl, dof, dv, x, xdot, y, d_partials, dot_lo,
                           dot_hi);
    }
    const hipError_t st = hipGetLastError();
    if (st != hipSuccess) return fail("lanczos_hip: %s", hipGetErrorString(st));
    return 0;
}

}  // namespace lz
