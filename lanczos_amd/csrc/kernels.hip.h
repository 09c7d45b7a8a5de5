// Device kernels of the Lanczos engine, written for gfx950 (CDNA4): 64-lane
// wavefronts, 256-thread workgroups, 8 XCDs with private L2s.
//
// Every kernel here is bandwidth-bound fp64 streaming work; none is shaped as a
// GEMM.  Reductions are two-stage and order-fixed (DPP inside a wavefront, LDS
// across the 4 wavefronts of a workgroup, one partial per workgroup, a
// single-workgroup finaliser), so results are bitwise reproducible run to run
// for a fixed grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lz {

constexpr int kBlock = 256;        // threads per workgroup (4 wavefronts)
constexpr int kWavesPerBlock = 4;
// The vector kernels walk tiles of kTileV2 consecutive double2 elements (2048 doubles, 16 KiB per
// stream), one workgroup per tile at a time, every thread kEPT2 elements kBlock apart: all the loads
// of a tile are issued before the first is used.  Vectors are zero padded to whole tiles.
constexpr int kEPT2 = 4;                       // double2 per thread per tile
constexpr int kTileV2 = kBlock * kEPT2;        // double2 per tile (2048 doubles)

// Streamed-once data (the basis in the reorthogonalisation kernels) is loaded
// with the non-temporal hint: on MI355X a pure read stream runs ~14 % faster with
// it (7.0 vs 6.2 TB/s measured on k_dots) because it does not churn L2/MALL.
typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 ld_stream(const double2 *p) {
    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    return make_double2(t.x, t.y);
}

// The matching store, for k_update's result (-4 % on that kernel: the trickle of w tiles no longer
// goes through the L2s between the basis reads).  Not for a vector the next SpMV multiplies with:
// the recurrence's output stays in the memory-side cache for it (dictionary SpMV +9 % otherwise).
__device__ __forceinline__ void st_stream(double2 *p, const double2 &v) {
    d2v t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<d2v *>(p));
}

// Lane L's value as a wave-uniform (scalar-register) value
template <int L>
__device__ __forceinline__ int32_t uniform_lane(int32_t v) {
    return __builtin_amdgcn_readlane(v, L);
}
template <int L>
__device__ __forceinline__ int64_t uniform_lane(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, L);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), L);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ---------------------------------------------------------------------------
// Wavefront reductions on DPP
// ---------------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], CTRL, 0xF, 0xF, false);
    b.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], CTRL, 0xF, 0xF, false);
    return b.d;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    b.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return b.d;
}

// Sum over aligned groups of G lanes (G a power of two, 1..64); every lane of a
// group ends up holding the group's sum.
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    if constexpr (G >= 2) v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp_f64<0x141>(v);   // row_half_mirror
    if constexpr (G >= 16) v += dpp_f64<0x140>(v);  // row_mirror
    if constexpr (G >= 32) v += __shfl_xor(v, 16);
    if constexpr (G >= 64) v += __shfl_xor(v, 32);
    return v;
}

// Sum over the whole wavefront; the result is wave-uniform.
__device__ __forceinline__ double wave_sum(double v) {
    v = group_sum<16>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Sum over the workgroup; valid in every thread.  `red` is 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *red) {
    const double s = wave_sum(v);
    __syncthreads();  // protect `red` against a previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Sum of prod[a..b) over each lane's own slice of a tile.  A slice of 64 or more
// entries (a long row) is summed by the whole wavefront instead of by its one
// lane, so a hub row of a power-law graph costs the wavefront tile/64 steps per
// tile rather than tile steps.
__device__ __forceinline__ double slice_sum(const double *prod, int a, int b, int lane) {
    double s = 0.0;
    const bool is_long = (b - a) >= 64;
    if (!is_long)
        for (int j = a; j < b; ++j) s += prod[j];
    unsigned long long todo = __ballot(is_long);
    while (todo) {  // wave-uniform
        const int r = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int ra = __shfl(a, r), rb = __shfl(b, r);
        double p = 0.0;
        for (int j = ra + lane; j < rb; j += 64) p += prod[j];
        p = wave_sum(p);
        if (lane == r) s += p;
    }
    return s;
}

// ---------------------------------------------------------------------------
// SpMV, "vector" form: G = 2^g lanes cooperate on one row
// ---------------------------------------------------------------------------
//
// y[row] = sum_j val[j] * x[col[j]]   (reference src/crs.c:155-178,
//                                      src/lanczos_cuda.c:562-577)
// fused with the per-workgroup partial of alpha = xdot . y
// (src/lanczos.c:112, src/lanczos_cuda.c:580).
//
// Workgroups are dealt round-robin over the 8 XCDs, so workgroups b and b+8
// share an L2.  Rows are cut into 8 contiguous ranges, one per b % 8, and the
// workgroups of one range walk it side by side: the x entries a banded matrix
// gathers stay inside one XCD's L2.
template <int G, typename RP>
__global__ void __launch_bounds__(kBlock)
k_spmv_vector(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col,
              const double *__restrict__ val, const double *__restrict__ x,
              const double *__restrict__ xdot, double *__restrict__ y,
              double *__restrict__ partials) {
    __shared__ double red[kWavesPerBlock];
    constexpr int RPB = kBlock / G;  // rows per workgroup per pass
    const int g_lane = threadIdx.x & (G - 1);
    const int g_id = threadIdx.x / G;
    const int64_t total = (n + RPB - 1) / RPB;
    const int64_t per_xcd = (total + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7;
    const int64_t slot = blockIdx.x >> 3;
    const int64_t slots = gridDim.x >> 3;
    int64_t c_end = (xcd + 1) * per_xcd;
    if (c_end > total) c_end = total;
    double acc = 0.0;
    for (int64_t c = xcd * per_xcd + slot; c < c_end; c += slots) {
        const int64_t row = c * RPB + g_id;
        double sum = 0.0;
        if (row < n) {
            const RP s = rp[row], e = rp[row + 1];
            for (RP j = s + g_lane; j < e; j += G) sum += val[j] * x[col[j]];
        }
        sum = group_sum<G>(sum);
        if (g_lane == 0 && row < n) {
            y[row] = sum;
            acc += sum * xdot[row];
        }
    }
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// The same for the two edge ranges of a row slab in one launch: rows [0, lo) and [hi, n), the rows
// that read entries received from the neighbouring slabs (m = lo + n - hi rows in all).
template <int G, typename RP>
__global__ void __launch_bounds__(kBlock)
k_spmv_edges(int64_t lo, int64_t hi, int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col,
             const double *__restrict__ val, const double *__restrict__ x, const double *__restrict__ xdot,
             double *__restrict__ y, double *__restrict__ partials) {
    __shared__ double red[kWavesPerBlock];
    constexpr int RPB = kBlock / G;
    const int g_lane = threadIdx.x & (G - 1);
    const int g_id = threadIdx.x / G;
    const int64_t m = lo + (n - hi);
    double acc = 0.0;
    for (int64_t c = blockIdx.x; c * RPB < m; c += gridDim.x) {
        const int64_t e = c * RPB + g_id;
        const int64_t row = e < lo ? e : e - lo + hi;
        double sum = 0.0;
        if (e < m) {
            const RP s = rp[row], t = rp[row + 1];
            for (RP j = s + g_lane; j < t; j += G) sum += val[j] * x[col[j]];
        }
        sum = group_sum<G>(sum);
        if (g_lane == 0 && e < m) {
            y[row] = sum;
            acc += sum * xdot[row];
        }
    }
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// ---------------------------------------------------------------------------
// SpMV, "stream" form: one wavefront per group of 64 rows
// ---------------------------------------------------------------------------
//
// Lane r of a wavefront owns row g0 + r of its group and loads that row's two
// offsets (coalesced).  The wavefront then streams the group's nnz span
// [rp[g0], rp[g0+64]) in tiles of kTile entries: all col/val loads of a tile
// are issued back to back on clamped (always valid) indices, then all x
// gathers, so one wavefront keeps 24 loads in flight with no branch between
// them; the products are parked in LDS and lane r sums the slice of the tile
// that belongs to its row.  Lanes never idle on short rows and col/val traffic
// is contiguous whatever the row lengths are.  A row longer than a tile is
// summed across tiles by its lane.  The offsets of the wavefront's next group
// are fetched before the current group is processed.
constexpr int kTileU = 8;            // 64-entry slices per tile
constexpr int kTile = 64 * kTileU;   // products per wavefront tile (4 KiB of LDS)

template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_spmv_stream(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col,
              const double *__restrict__ val, const double *__restrict__ x,
              const double *__restrict__ xdot, double *__restrict__ y,
              double *__restrict__ partials) {
    __shared__ double prod[kWavesPerBlock][kTile];
    __shared__ double red[kWavesPerBlock];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t groups = (n + 63) >> 6;
    // one "chunk" = the 4 row groups handled by the 4 wavefronts of a workgroup
    const int64_t total = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t per_xcd = (total + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7;
    const int64_t slot = blockIdx.x >> 3;
    const int64_t slots = gridDim.x >> 3;
    int64_t c_end = (xcd + 1) * per_xcd;
    if (c_end > total) c_end = total;
    double acc = 0.0;
    double *myprod = prod[wave];

    int64_t c = xcd * per_xcd + slot;
    // offsets of the first group (rows past the end read rp[n]: an empty row)
    RP nxt_lo = 0, nxt_hi = 0;
    if (c < c_end) {
        const int64_t row = ((c * kWavesPerBlock + wave) << 6) + lane;
        nxt_lo = rp[row < n ? row : n];
        nxt_hi = rp[row + 1 < n ? row + 1 : n];
    }
    for (; c < c_end; c += slots) {
        const int64_t row = ((c * kWavesPerBlock + wave) << 6) + lane;
        const RP my_lo = nxt_lo, my_hi = nxt_hi;
        if (c + slots < c_end) {
            const int64_t nrow = (((c + slots) * kWavesPerBlock + wave) << 6) + lane;
            nxt_lo = rp[nrow < n ? nrow : n];
            nxt_hi = rp[nrow + 1 < n ? nrow + 1 : n];
        }
        const double xd = xdot[row < n ? row : n - 1];
        const RP span_lo = __shfl(my_lo, 0);
        const RP span_hi = __shfl(my_hi, 63);
        double sum = 0.0;
        for (RP t0 = span_lo; t0 < span_hi; t0 += kTile) {
            const RP last = span_hi - 1;
            int32_t cj[kTileU];
            double vj[kTileU], xj[kTileU];
#pragma unroll
            for (int u = 0; u < kTileU; ++u) {
                RP j = t0 + u * 64 + lane;
                j = j < last ? j : last;
                cj[u] = col[j];
                vj[u] = val[j];
            }
#pragma unroll
            for (int u = 0; u < kTileU; ++u) xj[u] = x[cj[u]];
#pragma unroll
            for (int u = 0; u < kTileU; ++u) myprod[u * 64 + lane] = vj[u] * xj[u];
            __builtin_amdgcn_wave_barrier();
            // entries past span_hi hold clamped duplicates; no row slice reaches them
            const RP t1 = (span_hi - t0 > (RP)kTile) ? (RP)(t0 + kTile) : span_hi;
            const RP lo = my_lo > t0 ? my_lo : t0;
            const RP hi = my_hi < t1 ? my_hi : t1;
            // every lane calls slice_sum (it uses wavefront-wide operations)
            const int a1 = hi > lo ? (int)(lo - t0) : 0, b1 = hi > lo ? (int)(hi - t0) : 0;
            sum += slice_sum(myprod, a1, b1, lane);
            __builtin_amdgcn_wave_barrier();
        }
        if (row < n) {
            y[row] = sum;
            acc += sum * xd;
        }
    }
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// ---------------------------------------------------------------------------
// SpMV, "pipe" form: the stream form, software pipelined across row groups
// ---------------------------------------------------------------------------
//
// A wavefront owns groups of R = 2^LOG2R consecutive rows (lane r < R owns a
// row; R is chosen on the host so that a typical group's nnz span fits one
// tile).  SpMV has a two-deep dependent load chain (col -> x[col]); to keep
// HBM busy through it, iteration i issues, in this order,
//     the x gathers of group i            (its col/val were loaded during i-1)
//     the row offsets of group i+2
//     the col/val loads of group i+1      (its offsets were loaded during i-1)
// and only then waits for the gathers of group i.  The hardware retires vector
// loads in issue order, so that wait leaves the 18 younger loads in flight.
// The loop body is branch-free for single-tile groups (indices are clamped to
// valid entries instead of being predicated); spans longer than a tile take a
// synchronous tail loop.
template <typename RP, int LOG2R, bool NT>
__global__ void __launch_bounds__(kBlock)
k_spmv_pipe(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col,
            const double *__restrict__ val, const double *__restrict__ x,
            const double *__restrict__ xdot, double *__restrict__ y,
            double *__restrict__ partials, const long long *__restrict__ xcd_split) {
    __shared__ double prod[kWavesPerBlock][kTile];
    __shared__ double red[kWavesPerBlock];
    constexpr int R = 1 << LOG2R;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t xcd = blockIdx.x & 7;
    const int64_t slot = blockIdx.x >> 3;
    const int64_t slots = gridDim.x >> 3;
    // this XCD's range of chunks (4 groups each), cut so that the ranges hold equal numbers of entries
    const int64_t c_end = xcd_split[xcd + 1];
    const int64_t c0 = xcd_split[xcd] + slot;
    const int64_t count = c0 < c_end ? (c_end - c0 + slots - 1) / slots : 0;
    double *myprod = prod[wave];
    double acc = 0.0;

    auto ld_col = [&](RP j) -> int32_t { return NT ? __builtin_nontemporal_load(col + j) : col[j]; };
    auto ld_val = [&](RP j) -> double { return NT ? __builtin_nontemporal_load(val + j) : val[j]; };
    // row owned by this lane in the wavefront's i-th group (n: none, an empty row)
    auto row_of = [&](int64_t i) -> int64_t {
        const int64_t r = (((c0 + i * slots) * kWavesPerBlock + wave) << LOG2R) + lane;
        return (lane < R && r < n) ? r : n;
    };
    auto offsets = [&](int64_t r, RP &lo, RP &hi) {
        lo = rp[r];
        hi = rp[r < n ? r + 1 : n];
    };

    // issue the col loads, then the val loads, of the first tile of a span
    auto issue_colval = [&](RP s_lo, RP s_hi, int32_t (&cj)[kTileU], double (&vj)[kTileU]) {
        // `last` is a valid entry whenever nnz > 0 (the host never launches this kernel on nnz == 0)
        const RP last = s_hi > s_lo ? s_hi - 1 : (s_lo > 0 ? s_lo - 1 : 0);
#pragma unroll
        for (int u = 0; u < kTileU; ++u) {
            RP j = s_lo + u * 64 + lane;
            j = j < last ? j : last;
            cj[u] = ld_col(j);
        }
#pragma unroll
        for (int u = 0; u < kTileU; ++u) {
            RP j = s_lo + u * 64 + lane;
            j = j < last ? j : last;
            vj[u] = ld_val(j);
        }
    };
    // One pipeline step for group i.  (lo_c, hi_c) hold its offsets on entry and the
    // offsets of group i+2 on exit; (lo_n, hi_n) are the offsets of group i+1;
    // (c_c, v_c) hold its first tile, (c_n, v_n) receive the first tile of group i+1.
    // The two register sets alternate between steps so that no loaded value is ever
    // copied (a copy would have to wait for the load and drain the queue).
    auto step = [&](int64_t i, int32_t (&c_c)[kTileU], double (&v_c)[kTileU], int32_t (&c_n)[kTileU],
                    double (&v_n)[kTileU], RP &lo_c, RP &hi_c, const RP &lo_n, const RP &hi_n) {
        const int64_t row = row_of(i);
        const RP my_lo = lo_c, my_hi = hi_c;
        const RP span_lo = __shfl(my_lo, 0), span_hi = __shfl(my_hi, R - 1);
        // (a) gathers of this group
        double xj[kTileU];
#pragma unroll
        for (int u = 0; u < kTileU; ++u) xj[u] = x[c_c[u]];
        const double xd = xdot[row < n ? row : n - 1];
        // (b) offsets of group i+2 (clamped to the last group)
        offsets(row_of(i + 2 < count ? i + 2 : count - 1), lo_c, hi_c);
        // (c) col/val of group i+1 (when there is none: a harmless reload of valid entries)
        issue_colval(__shfl(lo_n, 0), __shfl(hi_n, R - 1), c_n, v_n);
        // (d) products of this group's first tile -> LDS -> per-row sums
#pragma unroll
        for (int u = 0; u < kTileU; ++u) myprod[u * 64 + lane] = v_c[u] * xj[u];
        __builtin_amdgcn_wave_barrier();
        double sum = 0.0;
        {
            const RP t1 = (span_hi - span_lo > (RP)kTile) ? (RP)(span_lo + kTile) : span_hi;
            const RP hi = my_hi < t1 ? my_hi : t1;
            const int a0 = (int)(my_lo - span_lo), b0 = hi > my_lo ? (int)(hi - span_lo) : a0;
            sum = slice_sum(myprod, a0, b0, lane);
        }
        __builtin_amdgcn_wave_barrier();
        // rare: the span does not fit one tile
        for (RP t0 = span_lo + kTile; t0 < span_hi; t0 += kTile) {
            const RP last = span_hi - 1;
#pragma unroll
            for (int u = 0; u < kTileU; ++u) {
                RP j = t0 + u * 64 + lane;
                j = j < last ? j : last;
                myprod[u * 64 + lane] = val[j] * x[col[j]];
            }
            __builtin_amdgcn_wave_barrier();
            const RP t1 = (span_hi - t0 > (RP)kTile) ? (RP)(t0 + kTile) : span_hi;
            const RP lo = my_lo > t0 ? my_lo : t0;
            const RP hi = my_hi < t1 ? my_hi : t1;
            // every lane calls slice_sum (it uses wavefront-wide operations)
            const int a1 = hi > lo ? (int)(lo - t0) : 0, b1 = hi > lo ? (int)(hi - t0) : 0;
            sum += slice_sum(myprod, a1, b1, lane);
            __builtin_amdgcn_wave_barrier();
        }
        if (row < n) {
            y[row] = sum;
            acc += sum * xd;
        }
    };

    if (count > 0) {
        // ---- prologue: offsets of groups 0 and 1, col/val of group 0
        RP loA, hiA, loB, hiB;
        offsets(row_of(0), loA, hiA);
        offsets(row_of(count > 1 ? 1 : 0), loB, hiB);
        int32_t cA[kTileU], cB[kTileU];
        double vA[kTileU], vB[kTileU];
        issue_colval(__shfl(loA, 0), __shfl(hiA, R - 1), cA, vA);
        int64_t i = 0;
        for (; i + 1 < count; i += 2) {
            step(i, cA, vA, cB, vB, loA, hiA, loB, hiB);
            step(i + 1, cB, vB, cA, vA, loB, hiB, loA, hiA);
        }
        if (i < count) step(i, cA, vA, cB, vB, loA, hiA, loB, hiB);
    }
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// ---------------------------------------------------------------------------
// SpMV, "merge" form: equal shares of ENTRIES, not of rows
// ---------------------------------------------------------------------------
//
// For matrices whose row lengths are wildly uneven (power-law graphs: most rows
// hold a handful of entries, some hold tens of thousands) any fixed number of
// rows per wavefront leaves most lanes idle.  Here a wavefront takes tiles of
// kTile consecutive entries, finds the rows they belong to by binary search in
// the row offsets, multiplies with fully used coalesced loads, and sums each
// row's slice of the tile (slice_sum: rows of 64 or more entries are summed by
// the whole wavefront).  A row that lies inside one tile is written at once; the
// at most two rows a tile shares with its neighbours leave their partial sums in
// carry[2 * tile + {0, 1}], which k_merge_fixup adds up in tile order: no atomics,
// the same sums on every run.  y must be zero on entry (empty rows, shared rows).
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_spmv_merge(int64_t n, int64_t nnz, const RP *__restrict__ rp, const int32_t *__restrict__ col,
             const double *__restrict__ val, const double *__restrict__ x, const double *__restrict__ xdot,
             double *__restrict__ y, double *__restrict__ partials, const int *__restrict__ tile_rows,
             int *__restrict__ carry_row, double *__restrict__ carry_val) {
    __shared__ double prod[kWavesPerBlock][kTile];
    __shared__ double red[kWavesPerBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tiles = (nnz + kTile - 1) / kTile;
    const int64_t per_xcd = (tiles + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    int64_t t_end = (xcd + 1) * per_xcd;
    if (t_end > tiles) t_end = tiles;
    double *myprod = prod[wave];
    double acc = 0.0;
    for (int64_t t = xcd * per_xcd + slot * kWavesPerBlock + wave; t < t_end; t += slots * kWavesPerBlock) {
        const int64_t e0 = t * kTile, e1 = (e0 + kTile < nnz) ? e0 + kTile : nnz;
        const int64_t row_lo = tile_rows[2 * t], row_hi = tile_rows[2 * t + 1];  // rows of its first / last entry
#pragma unroll
        for (int u = 0; u < kTileU; ++u) {
            const int64_t j = e0 + u * 64 + lane;
            myprod[u * 64 + lane] = j < e1 ? val[j] * x[col[j]] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        int c_row0 = -1, c_row1 = -1;  // wave-uniform: the rows shared with the previous / next tile
        double c_val0 = 0.0, c_val1 = 0.0;
        for (int64_t base = row_lo; base <= row_hi; base += 64) {
            const int64_t r = base + lane;
            const bool valid = r <= row_hi;
            const int64_t a = valid ? (int64_t)rp[r] : e1, b = valid ? (int64_t)rp[r + 1] : e1;
            const int64_t lo = a > e0 ? a : e0, hi = b < e1 ? b : e1;
            const bool some = valid && hi > lo;
            const double sum = slice_sum(myprod, some ? (int)(lo - e0) : 0, some ? (int)(hi - e0) : 0, lane);
            const bool whole = some && a >= e0 && b <= e1;
            if (whole) {
                y[r] = sum;
                acc += xdot[r] * sum;
            }
            const bool head = some && a < e0;               // continues a row of the previous tile
            const bool tail = some && !head && b > e1;      // continued by the next tile
            const unsigned long long mh = __ballot(head), mt = __ballot(tail);
            if (mh) {
                const int src = __ffsll((long long)mh) - 1;
                c_row0 = (int)__shfl((int)r, src);
                c_val0 = __shfl(sum, src);
            }
            if (mt) {
                const int src = __ffsll((long long)mt) - 1;
                c_row1 = (int)__shfl((int)r, src);
                c_val1 = __shfl(sum, src);
            }
        }
        if (lane == 0) {
            carry_row[2 * t] = c_row0;
            carry_val[2 * t] = c_val0;
            carry_row[2 * t + 1] = c_row1;
            carry_val[2 * t + 1] = c_val1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// tile_rows[2 t], [2 t + 1] = the rows holding the first and the last entry of tile t (the row
// holding entry e is the first r with rp[r + 1] > e); computed once per matrix
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_merge_tile_rows(int64_t n, int64_t nnz, const RP *__restrict__ rp, int *__restrict__ tile_rows) {
    const int64_t tiles = (nnz + kTile - 1) / kTile;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < 2 * tiles; i += step) {
        const int64_t t = i >> 1;
        const int64_t last = (t * kTile + kTile < nnz ? t * kTile + kTile : nnz) - 1;
        const int64_t e = (i & 1) ? last : t * kTile;
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)rp[mid + 1] > e) hi = mid;
            else lo = mid + 1;
        }
        tile_rows[i] = (int)lo;
    }
}

// The rows shared between tiles: the first carry of every run of equal rows adds the run up, in
// tile order, and writes the row.  partials[b] = this workgroup's part of xdot . y over those rows.
__global__ void __launch_bounds__(kBlock)
k_merge_fixup(int64_t count, const int *__restrict__ carry_row, const double *__restrict__ carry_val,
              const double *__restrict__ xdot, double *__restrict__ y, double *__restrict__ partials) {
    __shared__ double red[kWavesPerBlock];
    const int64_t step = (int64_t)gridDim.x * kBlock;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += step) {
        const int row = carry_row[i];
        if (row < 0) continue;
        int64_t j = i - 1;
        while (j >= 0 && carry_row[j] < 0) --j;
        if (j >= 0 && carry_row[j] == row) continue;  // not the first of its run
        double s = 0.0;
        for (int64_t k = i; k < count; ++k) {
            const int rk = carry_row[k];
            if (rk < 0) continue;
            if (rk != row) break;
            s += carry_val[k];
        }
        y[row] = s;
        acc += xdot[row] * s;
    }
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// ---------------------------------------------------------------------------
// SpMV, "rows" form: the pipe form with the gathers issued row-wise
// ---------------------------------------------------------------------------
//
// In the stream/pipe forms lane l gathers x for the l-th entry of the group's
// span, so neighbouring lanes read unrelated addresses and the texture
// addresser spends one cycle per lane on every gather: profiling the pipe form
// on the 7-point Laplacian shows it ~93 % busy, i.e. that kernel is bound by the
// gather issue rate, not by HBM.  Here the col/val tile is still loaded
// coalesced in span order, but it is parked in LDS and lane r then walks ITS OWN
// row: the j-th gather of the wavefront reads x[col(row r, entry j)] for
// r = 0..63.  For stencil-like matrices consecutive rows hold consecutive
// columns, so those 64 addresses are contiguous and the gather costs what a
// coalesced load costs.  Each lane sums its row left to right (the order of
// src/crs.c:172-175) and no product is staged in LDS.
//
// Used when every 64-row group's span, counted from the multiple of four entries below its
// start, fits one tile (checked on the host: span <= kRowsMaxSpan); rows longer than kRowChunk
// entries loop over chunks.
constexpr int kRowChunk = 8;
constexpr int kRowsAlignC = 32, kRowsAlignV = 16;  // entries per 128-byte line of col / val
constexpr int kRowsMaxSpan = kTile - (kRowsAlignC - 1);

template <typename RP, bool MASK>
__global__ void __launch_bounds__(kBlock)
k_spmv_rows(int64_t n, int64_t nnz, const RP *__restrict__ rp, const int32_t *__restrict__ col,
            const double *__restrict__ val, const double *__restrict__ x,
            const double *__restrict__ xdot, double *__restrict__ y,
            double *__restrict__ partials) {
    __shared__ __align__(16) double lds_v[kWavesPerBlock][kTile];
    __shared__ __align__(16) int32_t lds_c[kWavesPerBlock][kTile];
    __shared__ double red[kWavesPerBlock];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t groups = (n + 63) >> 6;
    const int64_t total = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t per_xcd = (total + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7;
    const int64_t slot = blockIdx.x >> 3;
    const int64_t slots = gridDim.x >> 3;
    int64_t c_end = (xcd + 1) * per_xcd;
    if (c_end > total) c_end = total;
    const int64_t c0 = xcd * per_xcd + slot;
    const int64_t count = c0 < c_end ? (c_end - c0 + slots - 1) / slots : 0;
    double *myv = lds_v[wave];
    int32_t *myc = lds_c[wave];
    double acc = 0.0;
    // the last aligned granules that still start inside the arrays (both are padded by the host)
    const RP last4 = (RP)((nnz - 1) & ~(int64_t)3), last2 = (RP)((nnz - 1) & ~(int64_t)1);

    auto row_of = [&](int64_t i) -> int64_t {
        const int64_t r = (((c0 + i * slots) * kWavesPerBlock + wave) << 6) + lane;
        return r < n ? r : n;
    };
    auto offsets = [&](int64_t r, RP &lo, RP &hi) {
        lo = rp[r];
        hi = rp[r < n ? r + 1 : n];
    };
    // The col and val tiles of a group start at the 128-byte lines that hold its first entry, so that
    // every lane loads 16 aligned bytes at a time and no two loads of a wavefront share a line: two loads of four column indices, four of two values
    // (6 memory instructions per tile where loads of single entries take 16), with the non-temporal
    // hint: the entries are read once per product and should not push x out of the caches (3 % on
    // the 256^3 Laplacian).  Granules past the end of the arrays are clamped to the last one; no row
    // reads those tile slots.
    // MASK: the tiles are loaded through buffer descriptors that end at the group's last entry (rounded up to
    // the 16-byte granule), so lanes past it return zeros without a memory request -- those lines belong to
    // the next group's tile, which the neighbouring wavefront of this workgroup requests anyway.  The
    // descriptor is rebuilt per group from wave-uniform values (scalar registers); its base moves with the
    // group, so arrays beyond 4 GiB are addressed as well.
    auto issue_colval = [&](RP lo_lane, RP hi_lane, i4v (&cj)[2], d2v (&vj)[4]) {
        if (MASK) {
            const RP s_lo = uniform_lane<0>(lo_lane), s_hi = uniform_lane<63>(hi_lane);
            const RP ac = s_lo & ~(RP)(kRowsAlignC - 1), av = s_lo & ~(RP)(kRowsAlignV - 1);
            RP ec = (s_hi - ac + 3) & ~(RP)3, ev = (s_hi - av + 1) & ~(RP)1;
            ec = ec < (RP)kTile ? ec : (RP)kTile;
            ev = ev < (RP)kTile ? ev : (RP)kTile;
            const __amdgpu_buffer_rsrc_t rc =
                __builtin_amdgcn_make_buffer_rsrc((void *)(col + ac), 0, (int)ec * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc((void *)(val + av), 0, (int)ev * 8, 0x00020000);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u4v t = __builtin_amdgcn_raw_buffer_load_b128(rc, u * 1024 + 16 * lane, 0, 2);
                cj[u] = i4v{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u4v t = __builtin_amdgcn_raw_buffer_load_b128(rv, u * 1024 + 16 * lane, 0, 2);
                vj[u] = __builtin_bit_cast(d2v, t);
            }
        } else {
            const RP s_lo = __shfl(lo_lane, 0);
            const RP ac = s_lo & ~(RP)(kRowsAlignC - 1), av = s_lo & ~(RP)(kRowsAlignV - 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                RP j = ac + u * 256 + 4 * lane;
                j = j < last4 ? j : last4;
                cj[u] = __builtin_nontemporal_load(reinterpret_cast<const i4v *>(col + j));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                RP j = av + u * 128 + 2 * lane;
                j = j < last2 ? j : last2;
                vj[u] = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(val + j));
            }
        }
    };
    auto step = [&](int64_t i, i4v (&c_c)[2], d2v (&v_c)[4], i4v (&c_n)[2], d2v (&v_n)[4], RP &lo_c,
                    RP &hi_c, const RP &lo_n, const RP &hi_n) {
        const int64_t row = row_of(i);
        const RP my_lo = lo_c, my_hi = hi_c;
        const RP span_lo = __shfl(my_lo, 0);
        // this row's first entry inside the two tiles
        const int off = (int)(my_lo - (span_lo & ~(RP)(kRowsAlignC - 1)));
        const int offv = (int)(my_lo - (span_lo & ~(RP)(kRowsAlignV - 1)));
        const int len = (int)(my_hi - my_lo);
        const int32_t safe_col = (int32_t)(row < n ? row : n - 1);
        // (a) park this group's tile in LDS
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<i4v *>(myc + u * 256 + 4 * lane) = c_c[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<d2v *>(myv + u * 128 + 2 * lane) = v_c[u];
        __builtin_amdgcn_wave_barrier();
        // (b) row-wise gathers of the first chunk
        int32_t cr[kRowChunk];
        double xr[kRowChunk];
#pragma unroll
        for (int j = 0; j < kRowChunk; ++j) cr[j] = j < len ? myc[off + j] : safe_col;
#pragma unroll
        for (int j = 0; j < kRowChunk; ++j) xr[j] = x[cr[j]];
        const double xd = xdot[row < n ? row : n - 1];
        // (c) offsets of group i+2, col/val of group i+1 (behind the gathers in the queue)
        offsets(row_of(i + 2 < count ? i + 2 : count - 1), lo_c, hi_c);
        issue_colval(lo_n, hi_n, c_n, v_n);
        // (d) the row sum, left to right
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < kRowChunk; ++j) {
            const double v = j < len ? myv[offv + j] : 0.0;
            sum += v * xr[j];
        }
        // rows longer than one chunk (wave-uniform trip count)
        int maxlen = len;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(maxlen, o);
            maxlen = other > maxlen ? other : maxlen;
        }
        for (int j0 = kRowChunk; j0 < maxlen; j0 += kRowChunk) {
#pragma unroll
            for (int j = 0; j < kRowChunk; ++j) {
                const bool in = j0 + j < len;
                const double v = in ? myv[offv + j0 + j] : 0.0;
                const int32_t c = in ? myc[off + j0 + j] : safe_col;
                sum += v * x[c];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (row < n) {
            y[row] = sum;
            acc += sum * xd;
        }
    };

    if (count > 0) {
        RP loA, hiA, loB, hiB;
        offsets(row_of(0), loA, hiA);
        offsets(row_of(count > 1 ? 1 : 0), loB, hiB);
        i4v cA[2], cB[2];
        d2v vA[4], vB[4];
        issue_colval(loA, hiA, cA, vA);
        int64_t i = 0;
        for (; i + 1 < count; i += 2) {
            step(i, cA, vA, cB, vB, loA, hiA, loB, hiB);
            step(i + 1, cB, vB, cA, vA, loB, hiB, loA, hiA);
        }
        if (i < count) step(i, cA, vA, cB, vB, loA, hiA, loB, hiB);
    }
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// The rows form with its tiles loaded straight into LDS (`buffer_load_dwordx4 ... lds`): the col/val
// tile of a group never passes through vector registers, and there is no LDS store pass.  Two tile
// buffers per wavefront alternate: while lane r walks its row of group i out of one, the tile of
// group i+1 lands in the other.  One instruction moves 64 x 16 bytes to consecutive LDS addresses,
// which is the layout the rows form parks its tiles in anyway; the same per-group descriptors end
// the tile at the group's last entry (lanes past it write zeros into slots no row reads).  Same sums
// in the same order as k_spmv_rows: the results are bit for bit equal.
typedef __attribute__((address_space(3))) void lds_void;

template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_spmv_rows_dma(int64_t n, int64_t nnz, const RP *__restrict__ rp, const int32_t *__restrict__ col,
                const double *__restrict__ val, const double *__restrict__ x,
                const double *__restrict__ xdot, double *__restrict__ y,
                double *__restrict__ partials) {
    // ONE LDS object: per wavefront and buffer 4 KiB of values then 2 KiB of columns; the four doubles of the
    // final reduction live behind the tiles (a tile load may still be landing when the loop ends)
    constexpr int kBufBytes = kTile * 12;
    __shared__ __align__(16) unsigned char lds[kWavesPerBlock * 2 * kBufBytes + 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t groups = (n + 63) >> 6;
    const int64_t total = (groups + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t per_xcd = (total + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7;
    const int64_t slot = blockIdx.x >> 3;
    const int64_t slots = gridDim.x >> 3;
    int64_t c_end = (xcd + 1) * per_xcd;
    if (c_end > total) c_end = total;
    const int64_t c0 = xcd * per_xcd + slot;
    const int64_t count = c0 < c_end ? (c_end - c0 + slots - 1) / slots : 0;
    unsigned char *mybuf = lds + (size_t)wave * 2 * kBufBytes;
    double acc = 0.0;

    auto row_of = [&](int64_t i) -> int64_t {
        const int64_t r = (((c0 + i * slots) * kWavesPerBlock + wave) << 6) + lane;
        return r < n ? r : n;
    };
    auto offsets = [&](int64_t r, RP &lo, RP &hi) {
        lo = rp[r];
        hi = rp[r < n ? r + 1 : n];
    };
    // tile of the group whose first / last row offsets sit in lanes 0 / 63 -> buffer `b`
    auto issue_tile = [&](RP lo_lane, RP hi_lane, int b) {
        const RP s_lo = uniform_lane<0>(lo_lane), s_hi = uniform_lane<63>(hi_lane);
        const RP ac = s_lo & ~(RP)(kRowsAlignC - 1), av = s_lo & ~(RP)(kRowsAlignV - 1);
        RP ec = (s_hi - ac + 3) & ~(RP)3, ev = (s_hi - av + 1) & ~(RP)1;
        ec = ec < (RP)kTile ? ec : (RP)kTile;
        ev = ev < (RP)kTile ? ev : (RP)kTile;
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(col + ac), 0, (int)ec * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(val + av), 0, (int)ev * 8, 0x00020000);
        lds_void *lv = (lds_void *)(mybuf + b * kBufBytes);
        lds_void *lc = (lds_void *)(mybuf + b * kBufBytes + kTile * 8);
        // the instruction offset moves the memory address and the LDS address together
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, lc, 16, 16 * lane, 0, 0, 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, lc, 16, 16 * lane, 0, 1024, 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, lv, 16, 16 * lane, 0, 0, 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, lv, 16, 16 * lane, 0, 1024, 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, lv, 16, 16 * lane, 0, 2048, 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, lv, 16, 16 * lane, 0, 3072, 2);
    };
    // hipcc does not order an LDS read behind the tile load that fills it: every step opens with a wait of its
    // own for everything in flight (the tile of this group is the youngest of it).  The row sums of a group
    // are stored at the top of the next step, behind that wait, so that the wait never sits on a fresh store.
    double held_sum = 0.0;
    int64_t held_row = n;
    auto step = [&](int64_t i, int b, RP &lo_c, RP &hi_c, const RP &lo_n, const RP &hi_n) {
        const double *myv = reinterpret_cast<const double *>(mybuf + b * kBufBytes);
        const int32_t *myc = reinterpret_cast<const int32_t *>(mybuf + b * kBufBytes + kTile * 8);
        const int64_t row = row_of(i);
        const RP my_lo = lo_c, my_hi = hi_c;
        const RP span_lo = __shfl(my_lo, 0);
        const int off = (int)(my_lo - (span_lo & ~(RP)(kRowsAlignC - 1)));
        const int offv = (int)(my_lo - (span_lo & ~(RP)(kRowsAlignV - 1)));
        const int len = (int)(my_hi - my_lo);
        const int32_t safe_col = (int32_t)(row < n ? row : n - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (held_row < n) y[held_row] = held_sum;
        // (a) row-wise gathers of the first chunk out of this group's tile
        int32_t cr[kRowChunk];
        double xr[kRowChunk];
#pragma unroll
        for (int j = 0; j < kRowChunk; ++j) cr[j] = j < len ? myc[off + j] : safe_col;
#pragma unroll
        for (int j = 0; j < kRowChunk; ++j) xr[j] = x[cr[j]];
        const double xd = xdot[row < n ? row : n - 1];
        // (b) offsets of group i+2, tile of group i+1 into the other buffer (behind the gathers in the queue)
        offsets(row_of(i + 2 < count ? i + 2 : count - 1), lo_c, hi_c);
        issue_tile(lo_n, hi_n, b ^ 1);
        // (c) the row sum, left to right
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < kRowChunk; ++j) {
            const double v = j < len ? myv[offv + j] : 0.0;
            sum += v * xr[j];
        }
        int maxlen = len;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(maxlen, o);
            maxlen = other > maxlen ? other : maxlen;
        }
        for (int j0 = kRowChunk; j0 < maxlen; j0 += kRowChunk) {
#pragma unroll
            for (int j = 0; j < kRowChunk; ++j) {
                const bool in = j0 + j < len;
                const double v = in ? myv[offv + j0 + j] : 0.0;
                const int32_t c = in ? myc[off + j0 + j] : safe_col;
                sum += v * x[c];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (row < n) acc += sum * xd;
        held_sum = sum;
        held_row = row;
    };

    if (count > 0) {
        RP loA, hiA, loB, hiB;
        offsets(row_of(0), loA, hiA);
        offsets(row_of(count > 1 ? 1 : 0), loB, hiB);
        issue_tile(loA, hiA, 0);
        int64_t i = 0;
        for (; i + 1 < count; i += 2) {
            step(i, 0, loA, hiA, loB, hiB);
            step(i + 1, 1, loB, hiB, loA, hiA);
        }
        if (i < count) step(i, 0, loA, hiA, loB, hiB);
    }
    if (held_row < n) y[held_row] = held_sum;
    // every tile load has landed before the LDS words of the reduction are written
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double *red = reinterpret_cast<double *>(lds + kWavesPerBlock * 2 * kBufBytes);
    const double total_acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = total_acc;
}

// ---------------------------------------------------------------------------
// Small finaliser: out[0] = sum(partials[0..count)) in a fixed order
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_finalize(const double *__restrict__ partials, int count, double *__restrict__ out) {
    __shared__ double red[kWavesPerBlock];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) s += partials[i];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = t;
}

// Two sums in one launch: out[0] = sum(a[0..na)), out[1] = sum(b[0..nb)), each in a fixed order
__global__ void __launch_bounds__(kBlock)
k_finalize2(const double *__restrict__ a, int na, const double *__restrict__ b, int nb, double *__restrict__ out) {
    __shared__ double red[kWavesPerBlock];
    double s = 0.0;
    for (int i = threadIdx.x; i < na; i += kBlock) s += a[i];
    const double ta = block_sum(s, red);
    __syncthreads();
    s = 0.0;
    for (int i = threadIdx.x; i < nb; i += kBlock) s += b[i];
    const double tb = block_sum(s, red);
    if (threadIdx.x == 0) {
        out[0] = ta;
        out[1] = tb;
    }
}

// h[l] = sum_b hpart[l * stride + b], one workgroup per l
__global__ void __launch_bounds__(kBlock)
k_hreduce(const double *__restrict__ hpart, int stride, int count, double *__restrict__ h) {
    __shared__ double red[kWavesPerBlock];
    const double *p = hpart + (size_t)blockIdx.x * stride;
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) s += p[i];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) h[blockIdx.x] = t;
}

// Sum of `count` per-workgroup partials in a fixed order: every workgroup that
// calls this on the same buffer gets the bitwise identical value, so consumers
// fold the reduction's last stage into their prologue instead of waiting for a
// separate finaliser launch.  (count == 1: an already reduced scalar.)
__device__ __forceinline__ double sum_partials(const double *__restrict__ parts, int count, double *red) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) s += parts[i];
    return block_sum(s, red);
}

// ---------------------------------------------------------------------------
// Three-term recurrence  w <- (w - beta * v_prev) - alpha * v_k
// (src/lanczos.c:150-154; src/lanczos_cuda.c:671-677) fused with the partial
// of w.w (src/lanczos.c:169-176).  alpha arrives as the partials the SpMV left
// (or as one reduced value).  Vectors are processed as double2; `n2` is the
// number of double2 elements (vector padding is zero and stays zero).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_recur(double2 *__restrict__ w, const double2 *__restrict__ vk, const double2 *__restrict__ vprev,
        int64_t n2, const double *__restrict__ alpha_parts, int alpha_count,
        const double *__restrict__ beta_p, double *__restrict__ partials,
        double *__restrict__ alpha_hist, double *alpha_host) {
    __shared__ double red[kWavesPerBlock];
    const double alpha = sum_partials(alpha_parts, alpha_count, red);
    const double beta = vprev ? beta_p[0] : 0.0;
    // a workgroup owns whole tiles of consecutive elements (a grid-stride walk, 8 MiB between a thread's
    // consecutive accesses over four streams, lost a quarter of its rate on vectors beyond the caches)
    const int64_t tiles = n2 / kTileV2;
    double acc = 0.0;
    if (vprev) {
        for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int64_t base = t * kTileV2 + threadIdx.x;
            double2 a[kEPT2], p[kEPT2], q[kEPT2];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) a[u] = w[base + u * kBlock];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) p[u] = vprev[base + u * kBlock];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) q[u] = vk[base + u * kBlock];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                a[u].x = (a[u].x - beta * p[u].x) - alpha * q[u].x;
                a[u].y = (a[u].y - beta * p[u].y) - alpha * q[u].y;
                w[base + u * kBlock] = a[u];
                acc += a[u].x * a[u].x + a[u].y * a[u].y;
            }
        }
    } else {
        for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int64_t base = t * kTileV2 + threadIdx.x;
            double2 a[kEPT2], q[kEPT2];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) a[u] = w[base + u * kBlock];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) q[u] = vk[base + u * kBlock];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                a[u].x = a[u].x - alpha * q[u].x;
                a[u].y = a[u].y - alpha * q[u].y;
                w[base + u * kBlock] = a[u];
                acc += a[u].x * a[u].x + a[u].y * a[u].y;
            }
        }
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t;
        if (blockIdx.x == 0) {
            alpha_hist[0] = alpha;
            if (alpha_host) alpha_host[0] = alpha;
        }
    }
}

// v_{k+1} = w / beta, beta = sqrt(sum of the w.w partials)   (src/lanczos.c:176-180)
__global__ void __launch_bounds__(kBlock)
k_scale(double2 *__restrict__ w, int64_t n2, const double *__restrict__ norm_parts, int norm_count,
        double *__restrict__ beta_cur, double *__restrict__ beta_hist, double *beta_host) {
    __shared__ double red[kWavesPerBlock];
    const double beta = sqrt(sum_partials(norm_parts, norm_count, red));
    const int64_t tiles = n2 / kTileV2;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * kTileV2 + threadIdx.x;
        double2 a[kEPT2];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) a[u] = w[base + u * kBlock];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) {
            a[u].x /= beta;
            a[u].y /= beta;
            w[base + u * kBlock] = a[u];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        beta_cur[0] = beta;
        beta_hist[0] = beta;
        if (beta_host) beta_host[0] = beta;
    }
}

// ---------------------------------------------------------------------------
// Recurrence with deferred normalisation (three-vector mode, no
// reorthogonalisation).  The basis is kept unnormalised, u_j = beta_{j-1} v_j,
// so the separate scaling pass (16 n bytes per iteration) disappears:
//     y = A u_k                      (the SpMV, unchanged)
//     alpha_k = s_k^2 (u_k . y)      s_k = 1 / beta_{k-1}, s_0 = 1
//     u_{k+1} = (s_k y - beta_{k-1} s_{k-1} u_{k-1}) - alpha_k s_k u_k
//     beta_k  = ||u_{k+1}||
// which is src/lanczos.c:150-180 with v_j = s_j u_j substituted.  beta_{k-1}
// arrives as the w.w partials the previous launch left (so no finaliser runs
// between launches); workgroup 0 records it.  beta_{k-2} is read from the
// history, written by the previous launch.
// ---------------------------------------------------------------------------
// NT_STORE: u_{k+1} is written past the caches (vectors that do not fit them: Engine::nt_stores)
template <bool NT_STORE>
__global__ void __launch_bounds__(kBlock)
k_recur_lazy(double2 *__restrict__ w, const double2 *__restrict__ uk, const double2 *__restrict__ uprev,
             int64_t n2, int k, const double *__restrict__ dot_parts, int dot_count,
             const double *__restrict__ norm_parts, int norm_count, double *__restrict__ partials,
             double *__restrict__ alpha_hist, double *__restrict__ beta_hist, double *host_hist) {
    __shared__ double red[kWavesPerBlock];
    const double dot = sum_partials(dot_parts, dot_count, red);
    double beta_km1 = 0.0, s_k = 1.0, s_km1 = 1.0;
    if (k > 0) {
        beta_km1 = sqrt(sum_partials(norm_parts, norm_count, red));
        s_k = 1.0 / beta_km1;
        if (k > 1) s_km1 = 1.0 / beta_hist[k - 2];
    }
    const double alpha = s_k * s_k * dot;
    const double c_p = beta_km1 * s_km1, c_k = alpha * s_k;
    const int64_t tiles = n2 / kTileV2;
    double acc = 0.0;
    if (k > 0) {
        for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int64_t base = t * kTileV2 + threadIdx.x;
            double2 a[kEPT2], p[kEPT2], q[kEPT2];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) a[u] = ld_stream(w + base + u * kBlock);
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) p[u] = ld_stream(uprev + base + u * kBlock);
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) q[u] = ld_stream(uk + base + u * kBlock);
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                a[u].x = (s_k * a[u].x - c_p * p[u].x) - c_k * q[u].x;
                a[u].y = (s_k * a[u].y - c_p * p[u].y) - c_k * q[u].y;
                if constexpr (NT_STORE) st_stream(w + base + u * kBlock, a[u]);
                else w[base + u * kBlock] = a[u];
                acc += a[u].x * a[u].x + a[u].y * a[u].y;
            }
        }
    } else {
        for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
            const int64_t base = t * kTileV2 + threadIdx.x;
            double2 a[kEPT2], q[kEPT2];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) a[u] = ld_stream(w + base + u * kBlock);
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) q[u] = ld_stream(uk + base + u * kBlock);
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                a[u].x = a[u].x - c_k * q[u].x;
                a[u].y = a[u].y - c_k * q[u].y;
                if constexpr (NT_STORE) st_stream(w + base + u * kBlock, a[u]);
                else w[base + u * kBlock] = a[u];
                acc += a[u].x * a[u].x + a[u].y * a[u].y;
            }
        }
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t;
        if (blockIdx.x == 0) {
            alpha_hist[k] = alpha;
            if (host_hist) host_hist[2 * k] = alpha;
            if (k > 0) {
                beta_hist[k - 1] = beta_km1;
                if (host_hist) host_hist[2 * (k - 1) + 1] = beta_km1;
            }
        }
    }
}

// Coefficients against an unnormalised basis u_l = beta_{l-1} v_l (u_0 = v_0): the component of w
// along u_l is (u_l . w) / ||u_l||^2.  h[l] holds u_l . w on entry.
__global__ void __launch_bounds__(kBlock)
k_lazy_coeff(double *__restrict__ h, int count, const double *__restrict__ beta_hist) {
    const int l = blockIdx.x * kBlock + threadIdx.x;
    if (l >= count || l == 0) return;
    const double b = beta_hist[l - 1];
    h[l] /= b * b;
}

// beta_hist[k] = sqrt(sum of partials): closes the last deferred iteration
__global__ void __launch_bounds__(kBlock)
k_lazy_flush(const double *__restrict__ norm_parts, int norm_count, double *__restrict__ beta_slot,
             double *beta_host) {
    __shared__ double red[kWavesPerBlock];
    const double beta = sqrt(sum_partials(norm_parts, norm_count, red));
    if (threadIdx.x == 0) {
        beta_slot[0] = beta;
        if (beta_host) beta_host[0] = beta;
    }
}

// out[0] = sum of partials, host[0] = its square root: the norm a host-side
// decision waits for while the device result stays where the next kernel reads it
__global__ void __launch_bounds__(kBlock)
k_norm_host(const double *__restrict__ parts, int count, double *__restrict__ out, double *host) {
    __shared__ double red[kWavesPerBlock];
    const double s = sum_partials(parts, count, red);
    if (threadIdx.x == 0) {
        out[0] = s;
        host[0] = sqrt(s);
    }
}

// |w - V h|^2 for an orthonormal V and h = V^T w:  out[0] = ww[0] - sum_l h[l]^2 (clamped at 0).
// h holds the few-ulp corrections of a reorthogonalisation pass, so nothing cancels; it saves a
// row-partitioned run the separate reduction of the new norm.  out may alias h (h is read first).
__global__ void __launch_bounds__(kBlock)
k_pyth_norm(const double *__restrict__ h, int nl, const double *__restrict__ ww, double *__restrict__ out) {
    __shared__ double red[kWavesPerBlock];
    double s = 0.0;
    for (int i = threadIdx.x; i < nl; i += kBlock) s += h[i] * h[i];
    const double w2 = ww[0];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = fmax(w2 - t, 0.0);
}

// v <- scale * v  (start-vector normalisation, src/lanczos.c:95-98)
__global__ void __launch_bounds__(kBlock)
k_scal_mul(double2 *__restrict__ v, int64_t n2, const double *__restrict__ norm2_p, double fixed) {
    const double s = norm2_p ? 1.0 / sqrt(norm2_p[0]) : fixed;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
        double2 a = v[i];
        a.x = s * a.x;
        a.y = s * a.y;
        v[i] = a;
    }
}

// partials[b] = partial of a.b
__global__ void __launch_bounds__(kBlock)
k_dot(const double2 *__restrict__ a, const double2 *__restrict__ b, int64_t n2,
      double *__restrict__ partials) {
    __shared__ double red[kWavesPerBlock];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
        const double2 p = a[i], q = b[i];
        acc += p.x * q.x + p.y * q.y;
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------
// Reorthogonalisation against basis vectors l0..l1-1 as a blocked GEMV pair
//   k_dots   : hpart[(l-l0) * hstride + block] = partial of V_l . w
//   k_update : w -= sum_l h[l-l0] * V_l ; partials[block] = partial of w.w
// (src/lanczos.c:156-167, src/lanczos_cuda.c:679-687).  A workgroup owns tiles
// of kTileV2 consecutive double2 elements; the w tile stays in registers while
// the basis streams past it once per kernel, so each kernel reads V exactly
// once.  Every vector is padded with zeros to a whole number of tiles, so no
// bounds checks are needed and the padding stays zero.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_dots(const double2 *__restrict__ V, int64_t ldv2, int l0, int l1, const double2 *__restrict__ w,
       int64_t n2, double *__restrict__ hpart, int hstride) {
    extern __shared__ double hs[];  // [4][nl]
    const int nl = l1 - l0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < nl * kWavesPerBlock; j += kBlock) hs[j] = 0.0;
    __syncthreads();
    double *myhs = hs + wave * nl;
    const int64_t tiles = (n2 + kTileV2 - 1) / kTileV2;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * kTileV2 + threadIdx.x;
        double2 wt[kEPT2];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) wt[u] = w[base + u * kBlock];
        const double2 *vl = V + (int64_t)l0 * ldv2 + base;
#pragma unroll 2
        for (int l = 0; l < nl; ++l, vl += ldv2) {
            double p = 0.0;
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                const double2 q = ld_stream(vl + u * kBlock);
                p += q.x * wt[u].x + q.y * wt[u].y;
            }
            p = wave_sum(p);
            if (lane == 0) myhs[l] += p;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nl; j += kBlock)
        hpart[(size_t)j * hstride + blockIdx.x] = (hs[j] + hs[nl + j]) + (hs[2 * nl + j] + hs[3 * nl + j]);
}

// k_dots on long vectors, as k_update_big is to k_update: a workgroup holds P / kEPT2 consecutive tiles of w in
// registers while the basis streams past, 128 KiB of every vector at a time; the tiles that do not fill such a group
// are taken as k_dots takes them.  The products are the same; a wavefront's sums run over the group's pieces before
// they are added up across lanes, so the coefficients agree with k_dots' to rounding, not bit for bit.
template <int P>
__global__ void __launch_bounds__(kBlock)
k_dots_big(const double2 *__restrict__ V, int64_t ldv2, int l0, int l1, const double2 *__restrict__ w,
           int64_t n2, double *__restrict__ hpart, int hstride) {
    extern __shared__ double hs[];  // [4][nl]
    const int nl = l1 - l0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < nl * kWavesPerBlock; j += kBlock) hs[j] = 0.0;
    __syncthreads();
    double *myhs = hs + wave * nl;
    const int64_t tiles = (n2 + kTileV2 - 1) / kTileV2;
    constexpr int G = P / kEPT2;
    const int64_t groups = tiles / G;
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const int64_t base = g * (int64_t)(G * kTileV2) + threadIdx.x;
        double2 wt[P];
#pragma unroll
        for (int u = 0; u < P; ++u) wt[u] = w[base + u * kBlock];
        const double2 *vl = V + (int64_t)l0 * ldv2 + base;
        for (int l = 0; l < nl; ++l, vl += ldv2) {
            // eight running sums: one chain of 2 P dependent multiply-adds per vector would leave the loads waiting
            double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            double2 q[P];  // all of the group's loads of this vector before the first product
#pragma unroll
            for (int u = 0; u < P; ++u) q[u] = ld_stream(vl + u * kBlock);
#pragma unroll
            for (int u = 0; u < P; ++u) p[u & 7] += q[u].x * wt[u].x + q[u].y * wt[u].y;
            const double s = wave_sum(((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
            if (lane == 0) myhs[l] += s;
        }
    }
    for (int64_t t = groups * G + blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * kTileV2 + threadIdx.x;
        double2 wt[kEPT2];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) wt[u] = w[base + u * kBlock];
        const double2 *vl = V + (int64_t)l0 * ldv2 + base;
        for (int l = 0; l < nl; ++l, vl += ldv2) {
            double p = 0.0;
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                const double2 q = ld_stream(vl + u * kBlock);
                p += q.x * wt[u].x + q.y * wt[u].y;
            }
            p = wave_sum(p);
            if (lane == 0) myhs[l] += p;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nl; j += kBlock)
        hpart[(size_t)j * hstride + blockIdx.x] = (hs[j] + hs[nl + j]) + (hs[2 * nl + j] + hs[3 * nl + j]);
}

// The three-term recurrence and the first half of the reorthogonalisation pass in one sweep: a workgroup
// forms its tile of w <- (w - beta v_{k-1}) - alpha v_k (the expression of k_recur), writes it, and keeps it in
// registers while V_{l0..l1-1} stream past it as in k_dots.  Same tiles, same grid and the same order of
// every sum as k_recur followed by k_dots, so the bits are theirs; what is saved is k_recur's own pass over
// w, v_k and v_{k-1} (32 n bytes become the 16 n of reading v_k and writing w here) and one launch.
__global__ void __launch_bounds__(kBlock)
k_recur_dots(const double2 *__restrict__ V, int64_t ldv2, int l0, int l1, double2 *__restrict__ w,
             const double2 *__restrict__ vk, const double2 *__restrict__ vprev, int64_t n2,
             const double *__restrict__ alpha_parts, int alpha_count, const double *__restrict__ beta_p,
             double *__restrict__ hpart, int hstride, double *__restrict__ alpha_hist, double *alpha_host) {
    extern __shared__ double hs[];  // [4][nl], then the four doubles of the alpha sum
    const int nl = l1 - l0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double alpha = sum_partials(alpha_parts, alpha_count, hs + nl * kWavesPerBlock);
    const double beta = beta_p[0];
    for (int j = threadIdx.x; j < nl * kWavesPerBlock; j += kBlock) hs[j] = 0.0;
    __syncthreads();
    double *myhs = hs + wave * nl;
    const int64_t tiles = (n2 + kTileV2 - 1) / kTileV2;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * kTileV2 + threadIdx.x;
        double2 wt[kEPT2], p[kEPT2], q[kEPT2];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) wt[u] = w[base + u * kBlock];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) p[u] = vprev[base + u * kBlock];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) q[u] = vk[base + u * kBlock];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) {
            wt[u].x = (wt[u].x - beta * p[u].x) - alpha * q[u].x;
            wt[u].y = (wt[u].y - beta * p[u].y) - alpha * q[u].y;
            w[base + u * kBlock] = wt[u];
        }
        const double2 *vl = V + (int64_t)l0 * ldv2 + base;
#pragma unroll 2
        for (int l = 0; l < nl; ++l, vl += ldv2) {
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                const double2 v = ld_stream(vl + u * kBlock);
                s += v.x * wt[u].x + v.y * wt[u].y;
            }
            s = wave_sum(s);
            if (lane == 0) myhs[l] += s;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nl; j += kBlock)
        hpart[(size_t)j * hstride + blockIdx.x] = (hs[j] + hs[nl + j]) + (hs[2 * nl + j] + hs[3 * nl + j]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        alpha_hist[0] = alpha;
        if (alpha_host) alpha_host[0] = alpha;
    }
}

__global__ void __launch_bounds__(kBlock)
k_update(const double2 *__restrict__ V, int64_t ldv2, int l0, int l1, double2 *__restrict__ w,
         int64_t n2, const double *__restrict__ h, double *__restrict__ partials) {
    __shared__ double red[kWavesPerBlock];
    const int nl = l1 - l0;
    const int64_t tiles = (n2 + kTileV2 - 1) / kTileV2;
    double acc = 0.0;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * kTileV2 + threadIdx.x;
        double2 wt[kEPT2];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) wt[u] = w[base + u * kBlock];
        const double2 *vl = V + (int64_t)l0 * ldv2 + base;
#pragma unroll 2
        for (int l = 0; l < nl; ++l, vl += ldv2) {
            const double c = h[l];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                const double2 q = ld_stream(vl + u * kBlock);
                wt[u].x -= q.x * c;
                wt[u].y -= q.y * c;
            }
        }
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) {
            st_stream(w + base + u * kBlock, wt[u]);
            acc += wt[u].x * wt[u].x + wt[u].y * wt[u].y;
        }
    }
    const double tsum = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tsum;
}

// k_update on long vectors: a workgroup owns P / kEPT2 consecutive tiles at a time (P = 32: 128 KiB of every basis
// vector in one go), the tiles that do not fill such a group are taken as k_update takes them.  Every element of w
// goes through the same subtractions in the same order as in k_update -- only the grouping of the w.w partial sums
// differs.  Longer runs of consecutive addresses per workgroup and vector are what the basis stream wants: at
// k = 150 on 2^24 rows 7.0 TB/s against 6.5 with 16 KiB tiles (32 and 64 KiB tiles reach that only at some grid
// sizes and fall to 4.2-4.8 TB/s at others; 128 KiB is level from 2 to 8 workgroups a CU).
template <int P>
__global__ void __launch_bounds__(kBlock)
k_update_big(const double2 *__restrict__ V, int64_t ldv2, int l0, int l1, double2 *__restrict__ w,
             int64_t n2, const double *__restrict__ h, double *__restrict__ partials) {
    __shared__ double red[kWavesPerBlock];
    const int nl = l1 - l0;
    const int64_t tiles = (n2 + kTileV2 - 1) / kTileV2;
    constexpr int G = P / kEPT2;  // tiles of kTileV2 in a group
    const int64_t groups = tiles / G;
    double acc = 0.0;
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const int64_t base = g * (int64_t)(G * kTileV2) + threadIdx.x;
        double2 wt[P];
#pragma unroll
        for (int u = 0; u < P; ++u) wt[u] = w[base + u * kBlock];
        const double2 *vl = V + (int64_t)l0 * ldv2 + base;
        for (int l = 0; l < nl; ++l, vl += ldv2) {
            const double c = h[l];
#pragma unroll
            for (int u = 0; u < P; ++u) {
                const double2 q = ld_stream(vl + u * kBlock);
                wt[u].x -= q.x * c;
                wt[u].y -= q.y * c;
            }
        }
#pragma unroll
        for (int u = 0; u < P; ++u) {
            st_stream(w + base + u * kBlock, wt[u]);
            acc += wt[u].x * wt[u].x + wt[u].y * wt[u].y;
        }
    }
    for (int64_t t = groups * G + blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * kTileV2 + threadIdx.x;
        double2 wt[kEPT2];
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) wt[u] = w[base + u * kBlock];
        const double2 *vl = V + (int64_t)l0 * ldv2 + base;
        for (int l = 0; l < nl; ++l, vl += ldv2) {
            const double c = h[l];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) {
                const double2 q = ld_stream(vl + u * kBlock);
                wt[u].x -= q.x * c;
                wt[u].y -= q.y * c;
            }
        }
#pragma unroll
        for (int u = 0; u < kEPT2; ++u) {
            st_stream(w + base + u * kBlock, wt[u]);
            acc += wt[u].x * wt[u].x + wt[u].y * wt[u].y;
        }
    }
    const double tsum = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tsum;
}

// out_j = sum_{l < m} coef[l][j] V_l for j < J: Ritz vectors, or the compressed basis of a thick
// restart, J at a time with one pass over the basis (the eigenvector product the reference leaves
// undone behind its `eigenvectors` argument, src/lanczos.h:34-36).  coef is [m][J], zero beyond nout.
template <int J>
__global__ void __launch_bounds__(kBlock)
k_combine(const double2 *__restrict__ V, int64_t ldv2, int m, const double *__restrict__ coef,
          double2 *__restrict__ out, int64_t out_ld2, int nout, int64_t n2) {
    const int64_t tiles = (n2 + kTileV2 - 1) / kTileV2;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t base = t * kTileV2 + threadIdx.x;
        double2 acc[J][kEPT2];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) acc[j][u] = make_double2(0.0, 0.0);
        const double2 *vl = V + base;
        for (int l = 0; l < m; ++l, vl += ldv2) {
            double2 q[kEPT2];
#pragma unroll
            for (int u = 0; u < kEPT2; ++u) q[u] = ld_stream(vl + u * kBlock);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const double c = coef[l * J + j];
#pragma unroll
                for (int u = 0; u < kEPT2; ++u) {
                    acc[j][u].x += c * q[u].x;
                    acc[j][u].y += c * q[u].y;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j)
            if (j < nout) {  // only the first nout outputs own memory
#pragma unroll
                for (int u = 0; u < kEPT2; ++u) out[(int64_t)j * out_ld2 + base + u * kBlock] = acc[j][u];
            }
    }
}

// Matrix statistics for choosing the SpMV form: stats[0] = largest nnz span of a
// 64-row group, stats[1] / stats[2] = over every `stride`-th row r, the number of
// entry positions shared with row r+1 / how many of them hold col(r+1,j) == col(r,j)+1;
// stats[3] / stats[4] = sampled non-empty rows / the sum of their column extents (max - min).
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_row_stats(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col, int64_t stride,
            unsigned long long *__restrict__ stats) {
    const int64_t groups = (n + 63) >> 6;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    unsigned long long span = 0, pairs = 0, alike = 0, rows = 0, width = 0;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += step) {
        const int64_t r1 = (g + 1) * 64 < n ? (g + 1) * 64 : n;
        const unsigned long long s = (unsigned long long)(rp[r1] - rp[g * 64]);
        span = s > span ? s : span;
    }
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t * stride + 1 < n; t += step) {
        const int64_t r = t * stride;
        const RP a0 = rp[r], a1 = rp[r + 1], a2 = rp[r + 2];
        RP len = (a1 - a0) < (a2 - a1) ? (a1 - a0) : (a2 - a1);
        if (len > 4096) len = 4096;          // a sample is enough; hub rows would keep one thread busy for milliseconds
        for (RP j = 0; j < len; ++j) {
            pairs++;
            alike += (col[a1 + j] == col[a0 + j] + 1);
        }
        if (a1 > a0) {  // stats[3], stats[4]: sampled rows and the sum of their column extents
            int lo = col[a0], hi = lo;
            const RP end = (a1 - a0) > 4096 ? a0 + 4096 : a1;
            for (RP j = a0 + 1; j < end; ++j) {
                const int c = col[j];
                lo = c < lo ? c : lo;
                hi = c > hi ? c : hi;
            }
            rows++;
            width += (unsigned long long)(hi - lo);
        }
    }
    if (span) atomicMax(&stats[0], span);
    if (pairs) atomicAdd(&stats[1], pairs);
    if (alike) atomicAdd(&stats[2], alike);
    if (rows) atomicAdd(&stats[3], rows);
    if (width) atomicAdd(&stats[4], width);
}

// 64-bit column indices -> 32-bit (matrix upload)
__global__ void __launch_bounds__(kBlock)
k_narrow_i64(const int64_t *__restrict__ in, int32_t *__restrict__ out, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride)
        out[i] = (int32_t)in[i];
}

}  // namespace lz
