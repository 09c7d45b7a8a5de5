// Feasibility probe, not part of the library: SpMV as two streaming passes.
//   pass 1  entries in (column panel, row block, row) order; the panel's slice of x sits in LDS;
//           each product goes to its slot in (row block, column panel, row) order
//   pass 2  a workgroup sums one piece of a row block's products into the block's y, held in LDS
//   pass 3  the pieces of a block are added up in order
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/probe/libbin_probe.so scripts/probe/bin_spmv_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef KPANEL
#define KPANEL 16384
#endif
#ifndef KBLOCK
#define KBLOCK 16384
#endif
#ifndef KUNROLL
#define KUNROLL 4
#endif
constexpr int kPanel = KPANEL;     // columns per panel
constexpr int kBlock = KBLOCK;     // rows per block
constexpr int kThreads = 1024;

#if KPANEL <= 16384
__global__ __launch_bounds__(kThreads) void k_bin_mul(const int64_t* __restrict__ item_lo, const int64_t* __restrict__ item_hi,
                                                      const int* __restrict__ item_panel, int64_t n,
                                                      const double* __restrict__ val, const uint16_t* __restrict__ col,
                                                      const uint32_t* __restrict__ dst, const double* __restrict__ x,
                                                      double* __restrict__ prod) {
    __shared__ double xs[kPanel];
    const int it = blockIdx.x;
    const int64_t base = (int64_t)item_panel[it] * kPanel;
    for (int t = threadIdx.x; t < kPanel; t += kThreads) xs[t] = base + t < n ? x[base + t] : 0.0;
    __syncthreads();
    const int64_t lo = item_lo[it], hi = item_hi[it];
    int64_t j = lo + threadIdx.x;
    for (; j + (KUNROLL - 1) * kThreads < hi; j += KUNROLL * kThreads) {
        double v[KUNROLL]; uint16_t c[KUNROLL]; uint32_t d[KUNROLL];
#pragma unroll
        for (int u = 0; u < KUNROLL; ++u) {
            v[u] = __builtin_nontemporal_load(val + j + u * kThreads);
            c[u] = __builtin_nontemporal_load(col + j + u * kThreads);
            d[u] = __builtin_nontemporal_load(dst + j + u * kThreads);
        }
#pragma unroll
        for (int u = 0; u < KUNROLL; ++u) prod[d[u]] = v[u] * xs[c[u]];
    }
    for (; j < hi; j += kThreads) prod[dst[j]] = val[j] * xs[col[j]];
}
#endif

// Every piece starts a multiple of kThreads entries after its block's first entry, so a chunk of
// kThreads entries never spans two pieces.  round[j] (heads only): how many earlier heads of the same
// chunk carry the same row; a head is a lane whose row differs from the lane before it in its wavefront.
template <bool kGather>
__global__ __launch_bounds__(kThreads) void k_bin_sum(const int64_t* __restrict__ piece_lo, const int64_t* __restrict__ piece_hi,
                                                      const int* __restrict__ gcol, const double* __restrict__ x,
                                                      const double* __restrict__ prod, const uint16_t* __restrict__ row,
                                                      const uint8_t* __restrict__ round, const uint8_t* __restrict__ chunk_rounds,
                                                      double* __restrict__ part) {
    __shared__ double ys[kBlock];
    const int pc = blockIdx.x;
    for (int t = threadIdx.x; t < kBlock; t += kThreads) ys[t] = 0.0;
    __syncthreads();
    const int64_t lo = piece_lo[pc], hi = piece_hi[pc];      // both multiples of kThreads (the stream is padded per block)
    const int lane = threadIdx.x & 63;
    constexpr int kAhead = 4;
    struct Group { double p[kAhead]; int c[kAhead], r[kAhead], rd[kAhead], rounds[kAhead]; };
    auto fetch = [&](Group& g, int64_t g0) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int64_t j = g0 + u * kThreads + threadIdx.x;
            const bool live = j < hi;
            g.p[u] = live ? __builtin_nontemporal_load(prod + j) : 0.0;
            if constexpr (kGather) g.c[u] = live ? __builtin_nontemporal_load(gcol + j) : 0;
            g.r[u] = live ? (int)__builtin_nontemporal_load(row + j) : 0;
            g.rd[u] = live ? (int)__builtin_nontemporal_load(round + j) : 255;
            g.rounds[u] = live ? (int)chunk_rounds[j >> 10] : -1;
        }
    };
    auto consume = [&](const Group& g) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            if (g.rounds[u] < 0) break;                         // uniform: whole chunks only
            const int rprev = __shfl_up(g.r[u], 1);
            const bool head = lane == 0 || rprev != g.r[u];
            const unsigned long long hm = __ballot(head);
            const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1)) << (lane + 1);
            const int end = above ? __ffsll((long long)above) - 1 : 64;      // first lane of the next run
            double acc = g.p[u];
            if constexpr (kGather) acc *= x[g.c[u]];
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const double q = __shfl_down(acc, s);
                if (lane + s < end) acc += q;
            }
            for (int t = 0; t <= g.rounds[u]; ++t) {
                if (head && g.rd[u] == t) ys[g.r[u]] += acc;
                __syncthreads();
            }
        }
    };
    Group ga, gb;
    fetch(ga, lo);
    for (int64_t g0 = lo; g0 < hi; g0 += 2 * kAhead * kThreads) {
        fetch(gb, g0 + kAhead * kThreads);
        consume(ga);
        fetch(ga, g0 + 2 * kAhead * kThreads);
        consume(gb);
    }
    double* out = part + (int64_t)pc * kBlock;
    for (int t = threadIdx.x; t < kBlock; t += kThreads) out[t] = ys[t];
}

__global__ __launch_bounds__(256) void k_bin_combine(const int* __restrict__ block_piece0, int64_t n,
                                                     const double* __restrict__ part, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / kBlock), t = (int)(i % kBlock);
    double s = 0.0;
    for (int pc = block_piece0[b]; pc < block_piece0[b + 1]; ++pc) s += part[(int64_t)pc * kBlock + t];
    y[i] = s;
}

extern "C" int bin_spmv(int items, const int64_t* item_lo, const int64_t* item_hi, const int* item_panel, int64_t n,
                        const double* val, const uint16_t* col, const uint32_t* dst, const double* x, double* prod,
                        int pieces, const int64_t* piece_lo, const int64_t* piece_hi, const uint16_t* row,
                        const uint8_t* round, const uint8_t* chunk_rounds, double* part, const int* block_piece0, double* y,
                        int which, const double* bval, const int* bcol) {
#if KPANEL <= 16384
    if (which & 1) k_bin_mul<<<items, kThreads, 0, 0>>>(item_lo, item_hi, item_panel, n, val, col, dst, x, prod);
#endif
    if (which & 2) k_bin_sum<false><<<pieces, kThreads, 0, 0>>>(piece_lo, piece_hi, nullptr, x, prod, row, round, chunk_rounds, part);
    if (which & 8) k_bin_sum<true><<<pieces, kThreads, 0, 0>>>(piece_lo, piece_hi, bcol, x, bval, row, round, chunk_rounds, part);
    if (which & 4) k_bin_combine<<<(unsigned)((n + 255) / 256), 256, 0, 0>>>(block_piece0, n, part, y);
    return (int)hipGetLastError();
}
