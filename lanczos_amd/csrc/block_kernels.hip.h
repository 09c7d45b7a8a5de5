// Device kernels of the block-Lanczos variant (block width 16) for gfx950.
//
// A block of Lanczos vectors is stored row-major, n_pad x 16 doubles: one row
// is one 128-byte line, so the SpMM gathers whole lines and the tall-skinny
// contractions against the basis
//     H = Q^T W   (16m x 16)         W -= Q H
// feed v_mfma_f64_16x16x4_f64 with coalesced loads.  Both are bandwidth-bound
// (about 4 flop per byte of basis streamed); the matrix cores are used because
// the contraction is a dense one, not to reach their peak.
//
// MFMA operand maps (f64 16x16x4): lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; result register r of lane l is
// D[row = (l >> 4) + 4 r][col = l & 15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lzb {

constexpr int kW = 16;          // block width
constexpr int kThreads = 256;   // 4 wavefronts
constexpr int kTileRows = 32;   // rows of a wavefront's tile in the contraction kernels
constexpr int kGramSeg = 8;     // basis blocks whose accumulators a wavefront holds at once
constexpr int kLdsRow = 17;  // doubles per LDS tile row (16 + 1 against bank conflicts)

typedef double d4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4v mfma16(double a, double b, d4v c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double ld_nt(const double *p) { return __builtin_nontemporal_load(p); }

// ---------------------------------------------------------------------------
// SpMM  Y = A X  on row-major n_pad x 16 blocks (the block form of
// src/crs.c:155-178).  Eight lanes own one row, lane c its columns 2c, 2c+1;
// the group loads 8 (col, val) pairs at a time with one coalesced load each and
// hands them round with ds_bpermute, so every gather reads one full 128-byte
// row of X.  Row sums run left to right in entry order.  Rows n..n_pad-1 are
// written as zeros.  Workgroup b serves the b % 8-th of eight contiguous row ranges
// (workgroups b and b + 8 share an XCD and its L2).
// ---------------------------------------------------------------------------
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int kMedRow = 64;     // rows with more entries leave the main kernel: one wavefront each (k_spmm16_medium)
constexpr int kLongRow = 4096;  // ... and above this one workgroup each (k_spmm16_long)
// (on power-law graphs the main kernel keeps rows of up to kHubRow entries only and the rest goes through the
// regrouped stream of spmm_hub.hip: eight lanes walking a row of 40 entries take five dependent rounds of
// loads while the seven other rows of their wavefront wait)
constexpr int kHubRow = 16;

template <typename RP>
__global__ void __launch_bounds__(kThreads)
k_spmm16(int64_t n, int64_t n_pad, const RP *__restrict__ rp, const int32_t *__restrict__ col,
         const double *__restrict__ val, const double *__restrict__ X, double *__restrict__ Y,
         const long long *__restrict__ xcd_split, int med_row) {
    // eight lanes own one row, lane c its columns 2c and 2c + 1; a wavefront works on 8 rows at once
    const int lane = threadIdx.x & 63, c = lane & 7, g = lane >> 3, wave = threadIdx.x >> 6;
    // this XCD's range of row octets (ranges of equal cost: equal row counts leave seven XCDs waiting for the one
    // that got the heavy rows of a power-law graph)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int64_t q_lo = xcd_split[xcd], q_hi = xcd_split[xcd + 1];
    const int64_t step = (int64_t)per * 4;
    int64_t q = q_lo + (int64_t)slot * 4 + wave;
    // Three rounds are in flight per wavefront: the row offsets of round i + 2 and the first eight entries of
    // each row of round i + 1 are requested before the gathers of round i, so no round waits for an offset
    // before it can ask for its entries (queue order: offsets, entries, gathers; the gathers' wait covers all).
    auto fetch_offsets = [&](int64_t qq, int64_t &e0o, int &leno) {
        e0o = 0;
        leno = 0;
        const int64_t r = qq * 8 + g;
        if (qq < q_hi && r < n) {
            e0o = (int64_t)rp[r];
            const int64_t full = (int64_t)rp[r + 1] - e0o;
            leno = full > med_row ? -1 : (int)full;  // -1: left to the kernels for long rows
        }
    };
    auto fetch_entries = [&](int64_t e0i, int leni, int &colo, double &valo) {
        colo = 0;
        valo = 0.0;
        if (c < leni) {
            colo = __builtin_nontemporal_load(col + e0i + c);
            valo = __builtin_nontemporal_load(val + e0i + c);
        }
    };
    int64_t e0 = 0, e0_n = 0, e0_nn = 0;
    int len = 0, len_n = 0, len_nn = 0, mycol = 0, mycol_n = 0;
    double myval = 0.0, myval_n = 0.0;
    fetch_offsets(q, e0, len);
    fetch_entries(e0, len, mycol, myval);
    fetch_offsets(q + step, e0_n, len_n);
    for (; q < q_hi; q += step) {
        const int64_t r = q * 8 + g;
        const int64_t ce0 = e0;
        const int clen = len;
        int ccol = mycol;
        double cval = myval;
        fetch_offsets(q + 2 * step, e0_nn, len_nn);
        fetch_entries(e0_n, len_n, mycol_n, myval_n);
        // Each group of eight lanes walks its own row: a group whose row has ended issues no gathers while its
        // neighbours go on (padding every row to the longest of the wavefront's eight cost five L1 accesses per
        // useful one on a power-law graph, where an octet of rows often holds one of 40 entries and seven of 2).
        d2v acc = d2v{0.0, 0.0};
        for (int p = 0; p < clen; p += 8) {
            if (p > 0) {
                ccol = 0;
                cval = 0.0;
                if (p + c < clen) {
                    ccol = col[ce0 + p + c];
                    cval = val[ce0 + p + c];
                }
            }
            const int cnt = min(8, clen - p);  // the same for the eight lanes of a group
            d2v x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cj = __shfl(ccol, j, 8);
                x[j] = d2v{0.0, 0.0};
                if (j < cnt) x[j] = *reinterpret_cast<const d2v *>(X + (int64_t)cj * kW + 2 * c);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double vj = __shfl(cval, j, 8);  // 0 past the row's end
                acc.x += vj * x[j].x;
                acc.y += vj * x[j].y;
            }
        }
        if (clen >= 0) __builtin_nontemporal_store(acc, reinterpret_cast<d2v *>(Y + r * kW + 2 * c));
        e0 = e0_n;
        len = len_n;
        mycol = mycol_n;
        myval = myval_n;
        e0_n = e0_nn;
        len_n = len_nn;
    }
}

// med[0 .. count[0]) = the rows with med_row < entries <= kLongRow, lng[0 .. count[1]) those with more
// (any order)
template <typename RP>
__global__ void __launch_bounds__(kThreads)
k_find_long_rows(int64_t n, const RP *__restrict__ rp, int med_row, int *__restrict__ med, int med_cap,
                 int *__restrict__ lng, int lng_cap, int *__restrict__ count) {
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < n; r += stride) {
        const int64_t len = (int64_t)rp[r + 1] - (int64_t)rp[r];
        if (len > kLongRow) {
            const int at = atomicAdd(count + 1, 1);
            if (at < lng_cap) lng[at] = (int)r;
        } else if (len > med_row) {
            const int at = atomicAdd(count, 1);
            if (at < med_cap) med[at] = (int)r;
        }
    }
}

// One wavefront per medium row: the 64 lanes load 64 (col, val) pairs at a time, each group of
// eight lanes takes every eighth of them, and the eight partial rows are added by butterfly
// exchange, the same way on every run.
template <typename RP>
__global__ void __launch_bounds__(kThreads)
k_spmm16_medium(const int *__restrict__ rows, int count, const RP *__restrict__ rp, const int32_t *__restrict__ col,
                const double *__restrict__ val, const double *__restrict__ X, double *__restrict__ Y) {
    const int lane = threadIdx.x & 63, c = lane & 7, g = lane >> 3;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= count) return;
    const int r = rows[i];
    const int64_t e0 = (int64_t)rp[r], e1 = (int64_t)rp[r + 1];
    d2v acc = d2v{0.0, 0.0};
    for (int64_t p = e0; p < e1; p += 64) {  // uniform over the wavefront
        int mycol = 0;
        double myval = 0.0;
        if (p + lane < e1) {
            mycol = col[p + lane];
            myval = val[p + lane];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cj = __shfl(mycol, 8 * j + g);
            const double vj = __shfl(myval, 8 * j + g);
            const d2v x = *reinterpret_cast<const d2v *>(X + (int64_t)cj * kW + 2 * c);
            acc.x += vj * x.x;
            acc.y += vj * x.y;
        }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
    }
    if (g == 0) *reinterpret_cast<d2v *>(Y + (int64_t)r * kW + 2 * c) = acc;
}

// One workgroup per long row (a hub of a power-law graph): its 32 groups of eight lanes take the
// entries 8 at a time in turn, and the 32 partial rows are added in group order.
template <typename RP>
__global__ void __launch_bounds__(kThreads)
k_spmm16_long(const int *__restrict__ rows, const RP *__restrict__ rp, const int32_t *__restrict__ col,
              const double *__restrict__ val, const double *__restrict__ X, double *__restrict__ Y) {
    __shared__ double part[32][kW];
    const int r = rows[blockIdx.x];
    const int c = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const int64_t e0 = (int64_t)rp[r], e1 = (int64_t)rp[r + 1];
    d2v acc = d2v{0.0, 0.0};
    for (int64_t p = e0; p < e1; p += kThreads) {  // uniform over the workgroup
        int mycol = 0;
        double myval = 0.0;
        if (p + threadIdx.x < e1) {
            mycol = col[p + threadIdx.x];
            myval = val[p + threadIdx.x];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cj = __shfl(mycol, j, 8);
            const double vj = __shfl(myval, j, 8);
            const d2v x = *reinterpret_cast<const d2v *>(X + (int64_t)cj * kW + 2 * c);
            acc.x += vj * x.x;
            acc.y += vj * x.y;
        }
    }
    part[grp][2 * c] = acc.x;
    part[grp][2 * c + 1] = acc.y;
    __syncthreads();
    if (threadIdx.x < kW) {
        double s = 0.0;
        for (int g = 0; g < 32; ++g) s += part[g][threadIdx.x];
        Y[(int64_t)r * kW + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------
// hpart[wg][(t - t0) * 256 + e] = this workgroup's part of Q_t^T W in fragment
// order (e = 64 r + lane), for basis blocks t0 <= t < t1.  A wavefront owns
// 32-row tiles, keeps the W tile as eight B operands and one accumulator per
// basis block of the current segment; the basis streams past once.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_gram16(const double *__restrict__ Q, int64_t blk_stride, int t0, int t1, const double *__restrict__ W,
         int64_t n_pad, double *__restrict__ hpart, int64_t hstride) {
    __shared__ double red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    const int64_t tiles = n_pad / kTileRows;
    for (int s0 = t0; s0 < t1; s0 += kGramSeg) {
        const int ns = min(kGramSeg, t1 - s0);
        d4v acc[kGramSeg];
#pragma unroll
        for (int t = 0; t < kGramSeg; ++t) acc[t] = d4v{0.0, 0.0, 0.0, 0.0};
        for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
            const int64_t off = (tile * kTileRows + kk) * kW + cc;
            double bw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bw[u] = W[off + u * 64];
#pragma unroll
            for (int t = 0; t < kGramSeg; ++t) {
                if (t < ns) {
                    const double *q = Q + (int64_t)(s0 + t) * blk_stride + off;
                    double a[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = ld_nt(q + u * 64);
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[t] = mfma16(a[u], bw[u], acc[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kGramSeg; ++t) {
            if (t < ns) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][r * 64 + lane] = acc[t][r];
                __syncthreads();
                const int e = threadIdx.x;
                hpart[(int64_t)blockIdx.x * hstride + (int64_t)(s0 + t - t0) * 256 + e] =
                    (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            }
        }
    }
}

// The same for W' = W - sum_{c0 <= c < c1} Q_c Hc[c - c0] (c1 - c0 <= 2), with W' formed tile by tile in registers
// and never stored: the three-term part of a block step then costs no pass that writes W (a launch that writes
// runs at 5.3 TB/s on this chip, one that only reads at 6.8), its coefficients are added to those of the full pass
// and one update applies both.  The correction is the update kernel's arithmetic (Q_c's tile through the
// wavefront's LDS tile into the A operand); its result registers are rows 4r..4r+3 of the tile in the layout the
// B operand of Q_t^T W' wants.
__global__ void __launch_bounds__(kThreads)
k_gram16_corr(const double *__restrict__ Q, int64_t blk_stride, int t0, int t1, const double *__restrict__ W,
              int64_t n_pad, double *__restrict__ hpart, int64_t hstride, int c0, int c1,
              const double *__restrict__ Hc) {
    __shared__ double red[4][256];
    __shared__ double lds[4][kTileRows * kLdsRow];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    double *lt = lds[wave];
    const int64_t tiles = n_pad / kTileRows;
    double hb[2][4];  // minus the correction's coefficient blocks in B-operand layout
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) hb[c][s] = c0 + c < c1 ? -Hc[c * 256 + (4 * s + kk) * kW + cc] : 0.0;
    for (int s0 = t0; s0 < t1; s0 += kGramSeg) {
        const int ns = min(kGramSeg, t1 - s0);
        d4v acc[kGramSeg];
#pragma unroll
        for (int t = 0; t < kGramSeg; ++t) acc[t] = d4v{0.0, 0.0, 0.0, 0.0};
        for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
            const int64_t off = (tile * kTileRows + kk) * kW + cc;
            d4v w2[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) w2[sub][r] = W[off + (sub * 16 + 4 * r) * kW];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (c0 + c < c1) {
                    const double *q = Q + (int64_t)(c0 + c) * blk_stride + off;
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = q[u * 64];  // plain loads: the Gram part below reads the tile again
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int u = 0; u < 8; ++u) lt[(4 * u + kk) * kLdsRow + cc] = x[u];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            w2[sub] = mfma16(lt[(sub * 16 + cc) * kLdsRow + 4 * s + kk], hb[c][s], w2[sub]);
                }
            }
#pragma unroll
            for (int t = 0; t < kGramSeg; ++t) {
                if (t < ns) {
                    const double *q = Q + (int64_t)(s0 + t) * blk_stride + off;
                    double a[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = ld_nt(q + u * 64);
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[t] = mfma16(a[u], w2[u >> 2][u & 3], acc[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kGramSeg; ++t) {
            if (t < ns) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][r * 64 + lane] = acc[t][r];
                __syncthreads();
                const int e = threadIdx.x;
                hpart[(int64_t)blockIdx.x * hstride + (int64_t)(s0 + t - t0) * 256 + e] =
                    (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
            }
        }
    }
}

// H[i] += Hc[i], i < count (the three-term coefficients onto those of the full pass)
__global__ void __launch_bounds__(kThreads)
k_add_coeff(double *__restrict__ H, const double *__restrict__ Hc, int count) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < count) H[i] += Hc[i];
}

// H[t][i][j] (row-major 16 x 16 per block) = sum over workgroups, in workgroup order
__global__ void __launch_bounds__(kThreads)
k_gram_reduce(const double *__restrict__ hpart, int groups, int64_t hstride, int count, double *__restrict__ H) {
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= count) return;
    double s = 0.0;
#pragma unroll 8
    for (int g = 0; g < groups; ++g) s += hpart[(int64_t)g * hstride + idx];
    const int t = idx >> 8, e = idx & 255, r = e >> 6, lane = e & 63;
    const int i = (lane >> 4) + 4 * r, j = lane & 15;
    H[t * 256 + i * kW + j] = s;
}

// ---------------------------------------------------------------------------
// W -= sum_t Q_t H_t for t0 <= t < t1.  The contraction runs over the columns
// of Q_t, which are the fast index in memory, so a tile of Q_t is loaded
// coalesced (lane = 4 rows x 16 columns per load), parked in the wavefront's
// own LDS tile and read back as the A operand (16 rows x 4 columns per MFMA).
// The LDS tile is private to a wavefront, whose LDS operations execute in
// order: no workgroup barrier is needed.
// ---------------------------------------------------------------------------

// gpart[wg][e] = the workgroup's sum of its four wavefronts' accumulators, fragment order
// (e = 64 r + lane), added in wavefront order.  `scratch` is >= 1024 doubles of LDS no longer in use.
__device__ __forceinline__ void emit_gram_part(d4v g, double *scratch, double *__restrict__ gpart) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) scratch[wave * 256 + r * 64 + lane] = g[r];
    __syncthreads();
    const int e = threadIdx.x;
    gpart[(int64_t)blockIdx.x * 256 + e] = (scratch[e] + scratch[256 + e]) + (scratch[512 + e] + scratch[768 + e]);
}

__global__ void __launch_bounds__(kThreads)
k_bupdate16(const double *__restrict__ Q, int64_t blk_stride, int t0, int t1, double *__restrict__ W,
            int64_t n_pad, const double *__restrict__ H, double *__restrict__ gpart) {
    __shared__ double lds[4][kTileRows * kLdsRow];
    d4v gacc = d4v{0.0, 0.0, 0.0, 0.0};  // this wavefront's part of W^T W of the result (gpart != NULL)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    double *lt = lds[wave];
    const int64_t tiles = n_pad / kTileRows;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t off = (tile * kTileRows + kk) * kW + cc;
        d4v acc[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[sub][r] = W[off + (sub * 16 + 4 * r) * kW];
        double x[8];
        {
            const double *q = Q + (int64_t)t0 * blk_stride + off;
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = ld_nt(q + u * 64);
        }
        for (int t = t0; t < t1; ++t) {
            double hb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) hb[s] = -H[(t - t0) * 256 + (4 * s + kk) * kW + cc];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 8; ++u) lt[(4 * u + kk) * kLdsRow + cc] = x[u];
            __builtin_amdgcn_wave_barrier();
            if (t + 1 < t1) {  // next block's tile is in flight while this one is multiplied
                const double *q = Q + (int64_t)(t + 1) * blk_stride + off;
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = ld_nt(q + u * 64);
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double a = lt[(sub * 16 + cc) * kLdsRow + 4 * s + kk];
                    acc[sub] = mfma16(a, hb[s], acc[sub]);
                }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                W[off + (sub * 16 + 4 * r) * kW] = acc[sub][r];
                // a result register is rows 4r..4r+3 of the tile in exactly the operand layout
                if (gpart) gacc = mfma16(acc[sub][r], acc[sub][r], gacc);
            }
    }
    if (gpart) emit_gram_part(gacc, reinterpret_cast<double *>(lds), gpart);
}

// W <- W M for a 16 x 16 matrix M (row-major): the triangular solve of CholQR
// applied as a product with the inverse factor.  Same operand handling as
// k_bupdate16 with the tile itself as the left operand.
__global__ void __launch_bounds__(kThreads)
k_bscale16(double *__restrict__ W, int64_t n_pad, const double *__restrict__ M, double *__restrict__ gpart) {
    __shared__ double lds[4][kTileRows * kLdsRow];
    d4v gacc = d4v{0.0, 0.0, 0.0, 0.0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    double *lt = lds[wave];
    double mb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) mb[s] = M[(4 * s + kk) * kW + cc];
    const int64_t tiles = n_pad / kTileRows;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t off = (tile * kTileRows + kk) * kW + cc;
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = W[off + u * 64];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) lt[(4 * u + kk) * kLdsRow + cc] = x[u];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            d4v acc = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(lt[(sub * 16 + cc) * kLdsRow + 4 * s + kk], mb[s], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                W[off + (sub * 16 + 4 * r) * kW] = acc[r];
                if (gpart) gacc = mfma16(acc[r], acc[r], gacc);
            }
        }
    }
    if (gpart) emit_gram_part(gacc, reinterpret_cast<double *>(lds), gpart);
}

// CholQR twice without writing the block in between.  k_bscale16_gram: gpart = the workgroup's part of
// (W M1)^T (W M1), W M1 formed tile by tile in registers and not stored (a launch that only reads).
// k_bscale16_chain: W <- (W M1) M2, the first product formed exactly as the other kernel formed it, handed to the
// second through the wavefront's LDS tile: bit for bit what k_bscale16 with M1 followed by k_bscale16 with M2
// leaves, with one pass that writes instead of two.
__global__ void __launch_bounds__(kThreads)
k_bscale16_gram(const double *__restrict__ W, int64_t n_pad, const double *__restrict__ M1, double *__restrict__ gpart) {
    __shared__ double lds[4][kTileRows * kLdsRow];
    d4v gacc = d4v{0.0, 0.0, 0.0, 0.0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    double *lt = lds[wave];
    double mb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) mb[s] = M1[(4 * s + kk) * kW + cc];
    const int64_t tiles = n_pad / kTileRows;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t off = (tile * kTileRows + kk) * kW + cc;
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = W[off + u * 64];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) lt[(4 * u + kk) * kLdsRow + cc] = x[u];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            d4v acc = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(lt[(sub * 16 + cc) * kLdsRow + 4 * s + kk], mb[s], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) gacc = mfma16(acc[r], acc[r], gacc);
        }
    }
    emit_gram_part(gacc, reinterpret_cast<double *>(lds), gpart);
}

__global__ void __launch_bounds__(kThreads)
k_bscale16_chain(double *__restrict__ W, int64_t n_pad, const double *__restrict__ M1, const double *__restrict__ M2) {
    __shared__ double lds[4][kTileRows * kLdsRow];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    double *lt = lds[wave];
    double m1[4], m2[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        m1[s] = M1[(4 * s + kk) * kW + cc];
        m2[s] = M2[(4 * s + kk) * kW + cc];
    }
    const int64_t tiles = n_pad / kTileRows;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t off = (tile * kTileRows + kk) * kW + cc;
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = W[off + u * 64];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) lt[(4 * u + kk) * kLdsRow + cc] = x[u];
        __builtin_amdgcn_wave_barrier();
        d4v first[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            first[sub] = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) first[sub] = mfma16(lt[(sub * 16 + cc) * kLdsRow + 4 * s + kk], m1[s], first[sub]);
        }
        // a result register is rows 4r..4r+3 of its half of the tile, the layout the tile was loaded in
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) lt[(4 * (4 * sub + r) + kk) * kLdsRow + cc] = first[sub][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            d4v acc = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(lt[(sub * 16 + cc) * kLdsRow + 4 * s + kk], m2[s], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) W[off + (sub * 16 + 4 * r) * kW] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------
// 16 x 16 Cholesky G = R^T R (R upper triangular), its inverse, and optionally
// the product R * Rprev (the off-diagonal block of T after a second CholQR
// pass).  One workgroup; thread (i, j) = (tid / 16, tid % 16).  A pivot that
// below 1e-14 of its diagonal entry (a condition number near 1e7, the edge of
// what CholQR twice can repair) marks the block rank deficient: status[0]
// receives `tag` (first occurrence wins) and the pivot is replaced by 1 so
// nothing turns NaN.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_chol16(const double *__restrict__ G, double *__restrict__ Rinv, double *__restrict__ R,
         const double *__restrict__ Rprev, double *__restrict__ Bout, int *__restrict__ status, int tag) {
    __shared__ double A[16][17], X[16][17];
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    A[i][j] = 0.5 * (G[i * 16 + j] + G[j * 16 + i]);
    __syncthreads();
    __shared__ double diag0[16];
    if (threadIdx.x < 16) diag0[threadIdx.x] = A[threadIdx.x][threadIdx.x];
    __syncthreads();
    for (int k = 0; k < 16; ++k) {
        if (threadIdx.x == 0) {
            double d = A[k][k];
            if (!(d > 1e-14 * diag0[k])) {
                if (status[0] == 0) status[0] = tag;
                d = 1.0;
            }
            A[k][k] = sqrt(d);
        }
        __syncthreads();
        if (i == k && j > k) A[k][j] /= A[k][k];
        __syncthreads();
        if (i > k && j >= i) A[i][j] -= A[k][i] * A[k][j];
        __syncthreads();
    }
    if (j < i) A[i][j] = 0.0;
    X[i][j] = 0.0;
    __syncthreads();
    if (threadIdx.x < 16) {  // column c of the inverse by back substitution
        const int c = threadIdx.x;
        X[c][c] = 1.0 / A[c][c];
        for (int r = c - 1; r >= 0; --r) {
            double s = 0.0;
            for (int m = r + 1; m <= c; ++m) s += A[r][m] * X[m][c];
            X[r][c] = -s / A[r][r];
        }
    }
    __syncthreads();
    Rinv[i * 16 + j] = X[i][j];
    R[i * 16 + j] = A[i][j];
    if (Bout) {
        double s = 0.0;
        for (int m = 0; m < 16; ++m) s += A[i][m] * Rprev[m * 16 + j];
        Bout[i * 16 + j] = s;
    }
}

// out = 0.5 (H + H^T) of one 16 x 16 block (the diagonal block of T)
__global__ void __launch_bounds__(kThreads)
k_symm16(const double *__restrict__ H, double *__restrict__ out) {
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    out[i * 16 + j] = 0.5 * (H[i * 16 + j] + H[j * 16 + i]);
}

}  // namespace lzb
