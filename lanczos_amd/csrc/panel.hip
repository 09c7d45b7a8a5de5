// SpMV by tiles, for matrices whose rows reach all over a vector that is too
// large for the caches (random sparsity, power-law graphs: the row-wise product
// of src/crs.c:155-178 then fetches one memory line per entry, and the fabric's
// request rate, not its bandwidth, sets the time).  The entries are regrouped
// once, on the device, by (row block, column panel): a row block is 2048 to
// 16384 rows, whose part of y a workgroup keeps in LDS; a panel is 2^bits
// columns, whose slice of x stays in cache while it is used.  A block's entries
// form one stream, panel after panel, cut into chunks of as many entries as the
// workgroup has threads; inside a chunk the entries are sorted by row (stable,
// so a row's entries keep their panel order).  A workgroup takes a piece -- a
// whole block, or a share of a block too full for one workgroup: a chunk is
// multiplied lane per entry, runs of equal rows are summed inside each
// wavefront (a segmented shuffle scan) and the first lane of a run adds it to
// the row's LDS slot, collecting the parts of a run that crosses wavefronts;
// chunks are separated by barriers -- no atomics on data, the same sums on
// every run.  A whole block goes straight to y; the pieces of a split block
// (the hub rows of a power-law graph put 2 % of all entries into one block)
// leave their sums in a side buffer, and a second kernel adds them in piece
// order.  Two schedules.  Even blocks (random sparsity): launches of as many
// workgroups as are resident together; they start together, carry equal work
// per panel, sweep the panels in step, and the gathers hit L2; 12 bytes per
// entry (value; row in block and column relative to the chunk's first in one
// word, the rare entry too far from it kept in a list of its own).  Uneven
// blocks (power-law graphs): one launch whose workgroups draw pieces from a
// counter, fullest blocks first, with the stream read two chunks and the
// gathers issued one chunk ahead; 14 bytes per entry (value, column, row).
#include <hipcub/hipcub.hpp>

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "engine.h"

namespace lz {

static constexpr int kBlock = 256;              // threads of the small kernels (keys, bounds, combine, far, fold)
static constexpr int kPieceEntries = 131072;    // entries in one piece at most (four times as many with 16384-row blocks)
// Four shapes of a workgroup's row block, RB = log2(rows), T = threads = entries per chunk; the y block is
// 8 << RB bytes of LDS, and kGroups(RB) workgroups (16 wavefronts per CU) are resident together:
//   RB 11, T  128   16 KiB, 8 per CU   even blocks, x up to 128 MiB
//   RB 12, T  256   32 KiB, 4 per CU   even blocks
//   RB 13, T  512   64 KiB, 2 per CU   uneven blocks
//   RB 14, T 1024  128 KiB, 1 per CU   uneven blocks, x beyond the memory-side cache (256 MiB)
// Taller blocks put more of a block's entries into each panel, so more gathers share a sector: on R-MAT
// graphs 8192 rows are worth 10 % at 2^24 rows and 16384 rows 20 % at 2^26, where the tiled SpMV is bound
// by the sectors its misses move; on even matrices, swept in step, they cost 6-30 %.
static constexpr int kGroups(int rb) { return rb >= 12 ? 1024 >> (rb - 12) : 1024 << (12 - rb); }

// slot < 0: the piece is its whole block and writes y; else it writes sums to side-buffer slot `slot`
struct TilePiece { int32_t block; uint32_t chunk0; int32_t chunks; int32_t slot; uint32_t gchunk0; uint32_t unused; };
// packed form: 32 - RB bits hold the column relative to the chunk's first; all ones there: no entry here (it is in the far list)
static constexpr int col_field(int rb) { return 32 - rb; }
static constexpr uint32_t far_column(int rb) { return (1u << (32 - rb)) - 1u; }
static constexpr uint32_t kFarCap = 1u << 22;   // entries the far list holds at most
struct TileSplit { int32_t block; int32_t slot0; int32_t count; int32_t unused; };

// keys[e] = (row block << jbits) | panel, idx[e] = e, rowid[e] = row, for every entry e.  A thread per
// row; rows of more than kKeyRowMax entries are left to k_tile_keys_long (their numbers go to `longs`,
// *nlong counts them): the hub rows of a power-law graph hold 10^5 to 10^6 entries.
static constexpr int kKeyRowMax = 256;
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_tile_keys(int64_t n, const RP *__restrict__ rp, const int32_t *__restrict__ col, int rowbits, int bits, int jbits,
            uint32_t *__restrict__ keys, uint32_t *__restrict__ idx, int32_t *__restrict__ rowid,
            int32_t *__restrict__ longs, int *__restrict__ nlong) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += step) {
        const int64_t e0 = (int64_t)rp[r], e1 = (int64_t)rp[r + 1];
        if (e1 - e0 > kKeyRowMax) {
            longs[atomicAdd(nlong, 1)] = (int32_t)r;
            continue;
        }
        for (int64_t e = e0; e < e1; ++e) {
            keys[e] = ((uint32_t)(r >> rowbits) << jbits) | ((uint32_t)col[e] >> bits);
            idx[e] = (uint32_t)e;
            rowid[e] = (int32_t)r;
        }
    }
}

// one workgroup per long row
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_tile_keys_long(const int32_t *__restrict__ longs, const RP *__restrict__ rp, const int32_t *__restrict__ col,
                 int rowbits, int bits, int jbits, uint32_t *__restrict__ keys, uint32_t *__restrict__ idx,
                 int32_t *__restrict__ rowid) {
    const int32_t r = longs[blockIdx.x];
    const int64_t e1 = (int64_t)rp[r + 1];
    const uint32_t high = (uint32_t)(r >> rowbits) << jbits;
    for (int64_t e = (int64_t)rp[r] + threadIdx.x; e < e1; e += kBlock) {
        keys[e] = high | ((uint32_t)col[e] >> bits);
        idx[e] = (uint32_t)e;
        rowid[e] = r;
    }
}

// block_ptr[K] = first position of the sorted keys that belongs to row block K or a later one, K = 0..nblocks
__global__ void __launch_bounds__(kBlock)
k_block_bounds(int64_t nnz, const uint32_t *__restrict__ keys, int64_t nblocks, int jbits,
               uint32_t *__restrict__ block_ptr) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t <= nblocks; t += step) {
        const uint64_t want = (uint64_t)t << jbits;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((uint64_t)keys[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        block_ptr[t] = (uint32_t)lo;
    }
}

// One workgroup per chunk: the chunk's entries (positions of the (block, panel) order) sorted by
// row in block, stable, and written out as the stream the product reads.  chunk_first[K] = number
// of chunks of the blocks before K.  kPacked: one 32-bit word per entry, row in block << 20 | column -
// base[chunk] with base the chunk's smallest column.  An entry 2^(32 - RB) - 1 or more columns away from it
// does not fit: its number goes to the far list (*far_count counts them, also beyond far_cap), and
// its place in the stream, after the chunk's other entries, holds far_column(RB), which the product skips.
template <bool kPacked, int RB, int T>
__global__ void __launch_bounds__(T)
k_tile_place(uint32_t chunks, int64_t nblocks, const uint32_t *__restrict__ chunk_first,
             const uint32_t *__restrict__ block_ptr, const uint32_t *__restrict__ idx,
             const int32_t *__restrict__ rowid, const int32_t *__restrict__ col, const double *__restrict__ val,
             uint16_t *__restrict__ trow, uint32_t *__restrict__ tcol, double *__restrict__ tval,
             uint32_t *__restrict__ base, int *__restrict__ far_count, uint32_t far_cap,
             uint32_t *__restrict__ far_entry) {
    using Sort = hipcub::BlockRadixSort<uint32_t, T, 1, uint32_t>;
    using Reduce = hipcub::BlockReduce<uint32_t, T>;
    __shared__ typename Sort::TempStorage tmp;
    __shared__ typename Reduce::TempStorage rtmp;
    __shared__ uint32_t lowest;
    for (uint32_t g = blockIdx.x; g < chunks; g += gridDim.x) {
        int64_t lo = 0, hi = nblocks;              // the last K with chunk_first[K] <= g
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (chunk_first[mid] <= g) lo = mid;
            else hi = mid;
        }
        const uint32_t q = block_ptr[lo] + (g - chunk_first[lo]) * (uint32_t)T + threadIdx.x;
        const bool inside = q < block_ptr[lo + 1];     // false in the empty lanes of a block's last chunk
        uint32_t key[1] = {(uint32_t)(1 << RB)}, e[1] = {0xffffffffu};
        uint32_t c = 0;
        if (inside) {
            e[0] = idx[q];
            key[0] = (uint32_t)rowid[e[0]] & (uint32_t)((1 << RB) - 1);
            c = (uint32_t)col[e[0]];
        }
        if (kPacked) {
            const uint32_t cmin = Reduce(rtmp).Reduce(inside ? c : 0xffffffffu, hipcub::Min());
            if (threadIdx.x == 0) {
                lowest = cmin;
                base[g] = cmin;
            }
            __syncthreads();
            if (inside && c - lowest >= far_column(RB)) {
                const uint32_t slot = (uint32_t)atomicAdd(far_count, 1);
                if (slot < far_cap) far_entry[slot] = e[0];
                key[0] = (uint32_t)(1 << RB);      // sorts behind the entries that stay
                e[0] = 0xffffffffu;
            }
        }
        Sort(tmp).Sort(key, e, 0, RB + 1);
        if (kPacked) {
            if (e[0] != 0xffffffffu) {
                tcol[q] = (key[0] << col_field(RB)) | ((uint32_t)col[e[0]] - lowest);
                tval[q] = val[e[0]];
            } else if (inside) {                       // the stream positions are the same set before and after the sort
                tcol[q] = far_column(RB);
                tval[q] = 0.0;
            }
        } else if (e[0] != 0xffffffffu) {
            trow[q] = (uint16_t)key[0];
            tcol[q] = (uint32_t)col[e[0]];
            tval[q] = val[e[0]];
        }
        __syncthreads();
    }
}

// The entries the packed stream left out, in CRS order (far_entry sorted): a thread at the first
// entry of a row adds the row's entries to y, which the tiled kernels have written before;
// dots[workgroup] = xdot . (what it added).
__global__ void __launch_bounds__(kBlock)
k_tiles_far(int nfar, const uint32_t *__restrict__ far_entry, const int32_t *__restrict__ far_row,
            const int32_t *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
            const double *__restrict__ xdot, double *__restrict__ y, double *__restrict__ dots) {
    __shared__ double red[kBlock / 64];
    const int t = (int)blockIdx.x * kBlock + (int)threadIdx.x;
    double acc = 0.0;
    if (t < nfar && (t == 0 || far_row[t - 1] != far_row[t])) {
        const int r = far_row[t];
        double sum = 0.0;
        for (int j = t; j < nfar && far_row[j] == r; ++j) sum += val[far_entry[j]] * x[col[far_entry[j]]];
        y[r] += sum;
        acc = xdot[r] * sum;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dots[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// far_row[t] = row of entry far_entry[t]
__global__ void __launch_bounds__(kBlock)
k_tiles_far_rows(int nfar, const uint32_t *__restrict__ far_entry, const int32_t *__restrict__ rowid,
                 int32_t *__restrict__ far_row) {
    const int t = (int)blockIdx.x * kBlock + (int)threadIdx.x;
    if (t < nfar) far_row[t] = rowid[far_entry[t]];
}

// A workgroup takes piece piece0 + its own number and then, with draw set, draws further ones from
// a counter (zero at launch) until none is left.  A piece that is a whole block writes
// its rows of y and dots[piece] = xdot . y over them; a share of a block writes its sums to the side
// buffer and dots[piece] = 0.
template <bool kAhead, bool kPacked, int RB, int T>
__global__ void __launch_bounds__(T)
k_spmv_tiles(int64_t n, int piece0, int npieces, int draw, const TilePiece *__restrict__ pieces,
             const uint32_t *__restrict__ block_ptr, const uint16_t *__restrict__ trow,
             const uint32_t *__restrict__ tcol, const uint32_t *__restrict__ tbase,
             const double *__restrict__ tval, const double *__restrict__ x,
             const double *__restrict__ xdot, double *__restrict__ y, double *__restrict__ side,
             double *__restrict__ dots, int *__restrict__ counter) {
    __shared__ double ylds[(1 << RB)];
    __shared__ double wsum[T / 64];     // per wavefront: sum of the run that starts at its lane 0,
    __shared__ int wrow[T / 64];        //   that run's row,
    __shared__ int wfull[T / 64];       //   whether the run fills the wavefront,
    __shared__ int wlast[T / 64];       //   and the row of its lane 63
    __shared__ double red[T / 64];
    __shared__ int next;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the first piece is the workgroup's own number, the later ones are drawn; pi is kept in a scalar
    // register (readfirstlane) so that the loop, with its barriers, stays uniform for the compiler too
    int pi = piece0 + (int)blockIdx.x;
    while (pi < npieces) {
        double acc = 0.0;
        const TilePiece pc = pieces[pi];
        for (int i = tid; i < (1 << RB); i += T) ylds[i] = 0.0;
        __syncthreads();
        const uint32_t a = block_ptr[pc.block] + pc.chunk0 * (uint32_t)T;
        const uint64_t stop = (uint64_t)a + (uint64_t)pc.chunks * T;
        const uint32_t b = stop < (uint64_t)block_ptr[pc.block + 1] ? (uint32_t)stop : block_ptr[pc.block + 1];
        // kAhead: a two-stage pipeline -- while a chunk is summed, the next chunk's gathers and the stream
        // of the one after it are in flight.  That pays where the workgroups wander through the panels on
        // their own (+8 % and +4-13 % on R-MAT graphs) and costs a quarter where they sweep in step and the
        // L2s' request rate is the limit (random sparsity).
        uint32_t cn = 0;
        int rn = -1, rc = -1;
        double vn = 0.0, vc = 0.0, xg = 0.0;
        if (kAhead) {                          // chunk 0: stream and gather; chunk 1: stream
            if (a + tid < b) {
                cn = __builtin_nontemporal_load(tcol + a + tid);
                rc = (int)__builtin_nontemporal_load(trow + a + tid);
                vc = __builtin_nontemporal_load(tval + a + tid);
                xg = x[cn];
            }
            if ((uint64_t)a + T + tid < b) {
                cn = __builtin_nontemporal_load(tcol + a + T + tid);
                rn = (int)__builtin_nontemporal_load(trow + a + T + tid);
                vn = __builtin_nontemporal_load(tval + a + T + tid);
            }
        }
        uint32_t cbn = kPacked && a < b ? tbase[pc.gchunk0] : 0u;
        for (uint32_t q0 = a; q0 < b; q0 += T) {
            int r = -1;
            double sum = 0.0;
            if (kAhead) {
                // this chunk's products; then the next chunk's gathers and the stream of the one after it go
                // out, to be in flight while this chunk is summed
                r = rc;
                sum = vc * xg;
                rc = rn;
                vc = vn;
                xg = rc >= 0 ? x[cn] : 0.0;
                const uint64_t qn = (uint64_t)q0 + 2 * T + tid;
                rn = -1;
                if (qn < b) {
                    cn = __builtin_nontemporal_load(tcol + qn);
                    rn = (int)__builtin_nontemporal_load(trow + qn);
                    vn = __builtin_nontemporal_load(tval + qn);
                }
            } else if (kPacked) {
                uint32_t pk = far_column(RB);
                double v = 0.0;
                if (q0 + tid < b) {
                    pk = __builtin_nontemporal_load(tcol + q0 + tid);
                    v = __builtin_nontemporal_load(tval + q0 + tid);
                }
                const uint32_t cb = cbn;           // uniform: one base per chunk, read one chunk ahead
                if (q0 + T < b) cbn = tbase[pc.gchunk0 + (q0 - a) / T + 1];
                const uint32_t rel = pk & far_column(RB);
                const bool live = rel != far_column(RB);       // else: no entry, or one that k_tiles_far adds afterwards
                const double xv = live ? x[(uint64_t)cb + rel] : 0.0;
                r = live ? (int)(pk >> col_field(RB)) : -1;
                sum = v * xv;                              // v on both sides of the test: its load stays up here too
            } else if (q0 + tid < b) {
                const uint32_t c = __builtin_nontemporal_load(tcol + q0 + tid);
                r = (int)__builtin_nontemporal_load(trow + q0 + tid);
                sum = __builtin_nontemporal_load(tval + q0 + tid) * x[c];
            }
            // runs of equal rows inside the wavefront: after the scan a run's first lane holds its sum
            const int rprev = __shfl_up(r, 1);
            const bool head = lane == 0 || rprev != r;
            const unsigned long long heads = __ballot(head);
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
            const int end = above ? __ffsll((long long)above) - 1 : 64;      // first lane of the next run
            // offset s is needed only when some run is longer than s, i.e. s lanes in a row continue a run
            unsigned long long cont = ~heads;
#pragma unroll
            for (int s = 1; s < 64 && cont; s <<= 1) {
                const double t = __shfl_down(sum, s);
                if (lane + s < end) sum += t;
                cont &= cont >> s;
            }
            if (lane == 0) {
                wsum[w] = sum;
                wrow[w] = r;
                wfull[w] = end == 64;
            }
            if (lane == 63) wlast[w] = r;
            __syncthreads();
            // a run that crosses wavefronts is added by its first part, which collects the others in order
            if (head && r >= 0 && !(lane == 0 && w > 0 && wlast[w - 1] == r)) {
                if (end == 64)
                    for (int v = w + 1; v < T / 64 && wrow[v] == r; ++v) {
                        sum += wsum[v];
                        if (!wfull[v]) break;
                    }
                ylds[r] += sum;
            }
            __syncthreads();
        }
        if (pc.slot < 0) {
            const int64_t row0 = (int64_t)pc.block << RB;
            for (int i = tid; i < (1 << RB); i += T)
                if (row0 + i < n) {
                    const double v = ylds[i];
                    y[row0 + i] = v;
                    acc += xdot[row0 + i] * v;
                }
        } else {
            double *out = side + (int64_t)pc.slot * (1 << RB);
            for (int i = tid; i < (1 << RB); i += T) out[i] = ylds[i];
        }
        // workgroup sum in a fixed order
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) red[w] = acc;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int i = 0; i < T / 64; ++i) t += red[i];
            dots[pi] = t;
            next = draw ? (int)gridDim.x + atomicAdd(counter, 1) : npieces;
        }
        __syncthreads();
        pi = __builtin_amdgcn_readfirstlane(next);
    }
}

// y of the split blocks: their pieces' sums added in piece order.  A workgroup takes 256 rows of one
// split block, a thread one row (loads four pieces ahead, adds in order); dots[workgroup] = its xdot . y
template <int RB>
__global__ void __launch_bounds__(kBlock)
k_tiles_combine(int64_t n, const TileSplit *__restrict__ splits, const double *__restrict__ side,
                const double *__restrict__ xdot, double *__restrict__ y, double *__restrict__ dots) {
    __shared__ double red[kBlock / 64];
    const int tid = threadIdx.x;
    const TileSplit sp = splits[blockIdx.x / ((1 << RB) / kBlock)];
    const int i = (int)(blockIdx.x % ((1 << RB) / kBlock)) * kBlock + tid;
    const int64_t row = ((int64_t)sp.block << RB) + i;
    const double *src = side + (int64_t)sp.slot0 * (1 << RB) + i;
    double v = 0.0;
    int p = 0;
    for (; p + 4 <= sp.count; p += 4) {
        const double a0 = src[(int64_t)p * (1 << RB)], a1 = src[(int64_t)(p + 1) * (1 << RB)];
        const double a2 = src[(int64_t)(p + 2) * (1 << RB)], a3 = src[(int64_t)(p + 3) * (1 << RB)];
        v = (((v + a0) + a1) + a2) + a3;
    }
    for (; p < sp.count; ++p) v += src[(int64_t)p * (1 << RB)];
    double acc = 0.0;
    if (row < n) {
        y[row] = v;
        acc = xdot[row] * v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) dots[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// partials[t] = sum of dots[t], dots[t + 256], ... : the per-piece dots folded to 256 partial sums
__global__ void __launch_bounds__(kBlock)
k_tiles_fold(int count, const double *__restrict__ dots, double *__restrict__ partials) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 8 * kBlock) {      // eight loads in flight, added in index order
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = i + u * kBlock < count ? dots[i + u * kBlock] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += a[u];
    }
    partials[threadIdx.x] = s;
}

void Engine::free_panels() {
    (void)hipFree(d_trow);
    (void)hipFree(d_tcol);
    (void)hipFree(d_pval);
    (void)hipFree(d_tblock_ptr);
    (void)hipFree(d_tpieces);
    (void)hipFree(d_tsplits);
    (void)hipFree(d_tside);
    (void)hipFree(d_tdots);
    (void)hipFree(d_tcounter);
    (void)hipFree(d_tbase);
    (void)hipFree(d_tfar_entry);
    (void)hipFree(d_tfar_row);
    d_tbase = nullptr;
    d_tfar_entry = nullptr;
    d_tfar_row = nullptr;
    tile_nfar = 0;
    tile_packed = false;
    d_tdots = nullptr;
    d_tcounter = nullptr;
    d_trow = nullptr;
    d_tcol = nullptr;
    d_pval = nullptr;
    d_tblock_ptr = nullptr;
    d_tpieces = nullptr;
    d_tsplits = nullptr;
    d_tside = nullptr;
    tile_panels = 0;
    tile_npieces = tile_nsplit = tile_nslots = 0;
}

// Regroup the device CRS by (row block of 1 << RB rows, column panel), chunks of T entries.  `even`: the
// blocks stay whole and are launched in step (RB 12 only); else pieces are cut and drawn.
template <int RB, int T>
int Engine::build_panels_shape(int bits, bool even) {
    constexpr int kRows = 1 << RB;
    const int64_t nblocks = (n + kRows - 1) >> RB;
    int jbits = 1, kbits = 1, panels = 0;
    while (((int64_t)1 << kbits) < nblocks) kbits++;
    for (;; ++bits) {                  // wider panels when (row block, panel) does not fit the 32-bit sort key
        panels = (int)(((n_ext - 1) >> bits) + 1);
        jbits = 1;
        while ((1 << jbits) < panels) jbits++;
        if (jbits + kbits <= 32 || bits >= 24) break;
    }
    if (jbits + kbits > 32) return 0;
    uint32_t *d_keys[2] = {nullptr, nullptr}, *d_idx[2] = {nullptr, nullptr}, *d_chunk_first = nullptr;
    int32_t *d_rowid = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_keys[0]);
        (void)hipFree(d_keys[1]);
        (void)hipFree(d_idx[0]);
        (void)hipFree(d_idx[1]);
        (void)hipFree(d_rowid);
        (void)hipFree(d_tmp);
        (void)hipFree(d_chunk_first);
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
    // d_keys[1] is free until the sort: its head holds the list of long rows (at most nnz / 257 of them) and
    // their count behind it
    int32_t *d_longs = (int32_t *)d_keys[1];
    const int64_t long_cap = nnz / (kKeyRowMax + 1) + 1;
    int nlong = 0;
    if ((long_cap + 1) * (int64_t)sizeof(int32_t) > (int64_t)sizeof(uint32_t) * nnz) {   // a handful of entries: no room, no need
        (void)hipFree(d_keys[1]);
        d_keys[1] = nullptr;
        PN_TRY(hipMalloc(&d_keys[1], sizeof(uint32_t) * (size_t)std::max<int64_t>(nnz, long_cap + 1)));
        d_longs = (int32_t *)d_keys[1];
    }
    int *d_nlong = (int *)(d_longs + long_cap);
    PN_TRY(hipMemsetAsync(d_nlong, 0, sizeof(int), stream));
    const int rblocks = (int)std::min<int64_t>(max_blocks, (n + kBlock - 1) / kBlock);
    if (rp64)
        hipLaunchKernelGGL((k_tile_keys<int64_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int64_t *)d_rp,
                           (const int32_t *)d_col, RB, bits, jbits, d_keys[0], d_idx[0], d_rowid, d_longs, d_nlong);
    else
        hipLaunchKernelGGL((k_tile_keys<int32_t>), dim3(rblocks), dim3(kBlock), 0, stream, n, (const int32_t *)d_rp,
                           (const int32_t *)d_col, RB, bits, jbits, d_keys[0], d_idx[0], d_rowid, d_longs, d_nlong);
    PN_TRY(hipGetLastError());
    PN_TRY(hipMemcpyAsync(&nlong, d_nlong, sizeof(int), hipMemcpyDeviceToHost, stream));
    PN_TRY(hipStreamSynchronize(stream));
    if (nlong > 0) {
        if (rp64)
            hipLaunchKernelGGL((k_tile_keys_long<int64_t>), dim3(nlong), dim3(kBlock), 0, stream, (const int32_t *)d_longs,
                               (const int64_t *)d_rp, (const int32_t *)d_col, RB, bits, jbits, d_keys[0], d_idx[0], d_rowid);
        else
            hipLaunchKernelGGL((k_tile_keys_long<int32_t>), dim3(nlong), dim3(kBlock), 0, stream, (const int32_t *)d_longs,
                               (const int32_t *)d_rp, (const int32_t *)d_col, RB, bits, jbits, d_keys[0], d_idx[0], d_rowid);
        PN_TRY(hipGetLastError());
    }
    hipcub::DoubleBuffer<uint32_t> keys(d_keys[0], d_keys[1]);
    hipcub::DoubleBuffer<uint32_t> vals(d_idx[0], d_idx[1]);
    size_t tmp_bytes = 0;
    PN_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, vals, (unsigned int)nnz, 0, jbits + kbits,
                                              stream));
    PN_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    PN_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, keys, vals, (unsigned int)nnz, 0, jbits + kbits,
                                              stream));
    PN_TRY(hipMalloc(&d_tblock_ptr, sizeof(uint32_t) * (nblocks + 1)));
    const int bblocks = (int)std::min<int64_t>(max_blocks, (nblocks + 1 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_block_bounds, dim3(bblocks), dim3(kBlock), 0, stream, nnz, (const uint32_t *)keys.Current(),
                       nblocks, jbits, d_tblock_ptr);
    PN_TRY(hipGetLastError());
    // the pieces: whole blocks where they fit, shares of at most kPieceEntries entries otherwise
    std::vector<uint32_t> bp, cf;
    std::vector<TilePiece> pieces;
    std::vector<TileSplit> splits;
    int slots = 0;
    try {
        bp.resize(nblocks + 1);
        cf.resize(nblocks + 1);
        PN_TRY(hipMemcpyAsync(bp.data(), d_tblock_ptr, sizeof(uint32_t) * (nblocks + 1), hipMemcpyDeviceToHost, stream));
        PN_TRY(hipStreamSynchronize(stream));
        uint64_t chunks = 0;
        uint32_t most = 0;
        for (int64_t K = 0; K < nblocks; ++K) {
            cf[K] = (uint32_t)chunks;
            const uint32_t c = (bp[K + 1] - bp[K] + T - 1) / T;
            chunks += c;
            most = std::max(most, c);
        }
        // Even blocks stay whole.  Otherwise a piece holds at most `limit` chunks: half of a workgroup's
        // share of the matrix, between 16384 and kPieceEntries entries, so that no piece is a long pole.
        tile_even = even;
        // measured on R-MAT graphs: 2^17 entries with 8192-row blocks (2^16: -2 %, 2^18: -3 % at 2^24 rows),
        // 2^19 with 16384-row blocks (2^17: -4 %, 2^16: -10 % at 2^26 rows)
        const uint64_t piece_entries = RB == 14 ? 4 * (uint64_t)kPieceEntries : (uint64_t)kPieceEntries;
        const uint32_t limit = even ? most
                                    : (uint32_t)std::min<uint64_t>(piece_entries / T,
                                                                   std::max<uint64_t>(16384 / T, chunks / (2 * kGroups(RB))));
        for (int64_t K = 0; K < nblocks; ++K) {
            const uint32_t c = (bp[K + 1] - bp[K] + T - 1) / T;
            if (c <= limit) {
                pieces.push_back({(int32_t)K, 0u, (int32_t)c, -1, cf[K], 0u});   // empty blocks too: their rows of y are zeros
            } else {
                splits.push_back({(int32_t)K, slots, 0, 0});
                for (uint32_t c0 = 0; c0 < c; c0 += limit, ++slots, ++splits.back().count)
                    pieces.push_back({(int32_t)K, c0, (int32_t)std::min<uint32_t>(limit, c - c0), slots, cf[K] + c0, 0u});
            }
        }
        cf[nblocks] = (uint32_t)chunks;
    } catch (const std::bad_alloc &) {
        cleanup();
        free_panels();
        return fail("lanczos_hip_create: out of host memory while planning the tiled SpMV");
    }
    const uint32_t chunks = cf[nblocks];
    PN_TRY(hipMalloc(&d_chunk_first, sizeof(uint32_t) * (nblocks + 1)));
    PN_TRY(hipMemcpyAsync(d_chunk_first, cf.data(), sizeof(uint32_t) * (nblocks + 1), hipMemcpyHostToDevice, stream));
    PN_TRY(hipMalloc(&d_tcol, sizeof(uint32_t) * nnz));
    PN_TRY(hipMalloc(&d_pval, sizeof(double) * nnz));
    const unsigned pgrid = (unsigned)std::min<uint32_t>(std::max<uint32_t>(chunks, 1u), 65536u * 16u);
    // Even blocks: try the 12-byte stream (row and chunk-relative column in one word) first; it holds
    // when no chunk's columns span 2^20 or more.  Otherwise, and for uneven blocks whose chunks cross
    // many small tiles, 14 bytes: value, column, row.
    tile_packed = false;
    if (even && chunks > 0) {
        int far = 0;
        uint32_t *far_raw = nullptr;                        // unsorted, as the workgroups append them
        PN_TRY(hipMalloc(&d_tbase, sizeof(uint32_t) * chunks));
        PN_TRY(hipMalloc(&d_tcounter, sizeof(int)));
        PN_TRY(hipMemsetAsync(d_tcounter, 0, sizeof(int), stream));
        hipError_t st = hipMalloc(&far_raw, sizeof(uint32_t) * kFarCap);
        if (st == hipSuccess) {
            hipLaunchKernelGGL((k_tile_place<true, RB, T>), dim3(pgrid), dim3(T), 0, stream, chunks, nblocks,
                               (const uint32_t *)d_chunk_first, (const uint32_t *)d_tblock_ptr,
                               (const uint32_t *)vals.Current(), (const int32_t *)d_rowid, (const int32_t *)d_col,
                               (const double *)d_val, (uint16_t *)nullptr, d_tcol, d_pval, d_tbase, d_tcounter, kFarCap,
                               far_raw);
            st = hipGetLastError();
        }
        if (st == hipSuccess) st = hipMemcpyAsync(&far, d_tcounter, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (st == hipSuccess) st = hipStreamSynchronize(stream);
        tile_packed = st == hipSuccess && far >= 0 && (uint32_t)far <= kFarCap;
        if (tile_packed && far > 0) {                      // the far list in CRS order, with each entry's row
            size_t sort_bytes = 0;
            void *sort_tmp = nullptr;
            st = hipMalloc(&d_tfar_entry, sizeof(uint32_t) * far);
            if (st == hipSuccess) st = hipMalloc(&d_tfar_row, sizeof(int32_t) * far);
            if (st == hipSuccess)
                st = hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, far_raw, d_tfar_entry, far, 0, 32, stream);
            if (st == hipSuccess) st = hipMalloc(&sort_tmp, sort_bytes ? sort_bytes : 1);
            if (st == hipSuccess)
                st = hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, far_raw, d_tfar_entry, far, 0, 32, stream);
            if (st == hipSuccess) {
                hipLaunchKernelGGL(k_tiles_far_rows, dim3((far + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, far,
                                   (const uint32_t *)d_tfar_entry, (const int32_t *)d_rowid, d_tfar_row);
                st = hipGetLastError();
            }
            if (st == hipSuccess) st = hipStreamSynchronize(stream);
            (void)hipFree(sort_tmp);
        }
        (void)hipFree(far_raw);
        PN_TRY(st);
        tile_nfar = tile_packed ? far : 0;
        if (getenv("LANCZOS_HIP_VERBOSE"))
            fprintf(stderr, "lanczos_hip: tiled SpMV, %u chunks, %d entries kept out of the packed stream%s\n", chunks,
                    far, tile_packed ? "" : " (too many: unpacked form)");
        if (!tile_packed) {
            (void)hipFree(d_tbase);
            d_tbase = nullptr;
        }
    }
    if (!tile_packed) {
        PN_TRY(hipMalloc(&d_trow, sizeof(uint16_t) * nnz));
        if (chunks > 0)
            hipLaunchKernelGGL((k_tile_place<false, RB, T>), dim3(pgrid), dim3(T), 0, stream, chunks, nblocks,
                               (const uint32_t *)d_chunk_first, (const uint32_t *)d_tblock_ptr,
                               (const uint32_t *)vals.Current(), (const int32_t *)d_rowid, (const int32_t *)d_col,
                               (const double *)d_val, d_trow, d_tcol, d_pval, (uint32_t *)nullptr, (int *)nullptr, 0u,
                               (uint32_t *)nullptr);
        PN_TRY(hipGetLastError());
    }
    PN_TRY(hipMalloc(&d_tpieces, sizeof(TilePiece) * std::max<size_t>(pieces.size(), 1)));
    PN_TRY(hipMemcpyAsync(d_tpieces, pieces.data(), sizeof(TilePiece) * pieces.size(), hipMemcpyHostToDevice, stream));
    if (!splits.empty()) {
        PN_TRY(hipMalloc(&d_tsplits, sizeof(TileSplit) * splits.size()));
        PN_TRY(hipMemcpyAsync(d_tsplits, splits.data(), sizeof(TileSplit) * splits.size(), hipMemcpyHostToDevice, stream));
        PN_TRY(hipMalloc(&d_tside, sizeof(double) * (size_t)slots * kRows));
    }
    PN_TRY(hipMalloc(&d_tdots, sizeof(double) * (pieces.size() + splits.size() * (kRows / kBlock) + kFarCap / kBlock)));
    if (!d_tcounter) PN_TRY(hipMalloc(&d_tcounter, sizeof(int)));
    PN_TRY(hipStreamSynchronize(stream));
    cleanup();
#undef PN_TRY
    if (getenv("LANCZOS_HIP_VERBOSE"))
        fprintf(stderr, "lanczos_hip: tiled SpMV, %s schedule, %d-row blocks, %d threads, panels of 2^%d columns, %zu pieces, "
                        "%zu split blocks (%d side slots), %s stream\n",
                even ? "even" : "uneven", kRows, T, bits, pieces.size(), splits.size(), slots,
                tile_packed ? "12-byte" : "14-byte");
    tile_panels = panels;
    tile_bits = bits;
    tile_rowbits = RB;
    tile_npieces = (int)pieces.size();
    tile_nsplit = (int)splits.size();
    tile_nslots = slots;
    return 0;
}

// Regroup the device CRS for the tiled SpMV.  tile_panels > 0 afterwards when the form was built; 0
// when the matrix does not qualify (too many entries for 32-bit positions, or a key wider than 32
// bits).  Non-zero return on errors.
int Engine::build_panels(int bits) {
    free_panels();
    if (bits > 24) bits = 24;
    if (nnz == 0 || nnz >= ((int64_t)1 << 32) - 1024) return 0;
    // The weights of the 4096-row blocks decide the schedule: about equal (the fullest within 1.25 of
    // the mean) and enough of them to go round -> whole blocks in step; else pieces drawn from a counter.
    const int64_t nb12 = (n + 4095) >> 12;
    bool even = false;
    try {
        std::vector<int64_t> head(nb12 + 1);
        const size_t esz = rp64 ? 8 : 4;
        std::vector<unsigned char> raw((size_t)(nb12 + 1) * esz);
        hipError_t st = hipSuccess;
        if (nb12 > 0)
            st = hipMemcpy2DAsync(raw.data(), esz, d_rp, 4096 * esz, esz, (size_t)nb12, hipMemcpyDeviceToHost, stream);
        if (st == hipSuccess)
            st = hipMemcpyAsync(raw.data() + (size_t)nb12 * esz, (const unsigned char *)d_rp + (size_t)n * esz, esz,
                                hipMemcpyDeviceToHost, stream);
        if (st == hipSuccess) st = hipStreamSynchronize(stream);
        if (st != hipSuccess) return fail("lanczos_hip: reading the row offsets failed: %s", hipGetErrorString(st));
        for (int64_t K = 0; K <= nb12; ++K)
            head[K] = rp64 ? ((const int64_t *)raw.data())[K] : (int64_t)((const int32_t *)raw.data())[K];
        int64_t most = 0;
        for (int64_t K = 0; K < nb12; ++K) most = std::max(most, head[K + 1] - head[K]);
        even = nb12 >= kGroups(12) / 4 && 4 * most <= 5 * (nnz / nb12) + 8 * 256;
    } catch (const std::bad_alloc &) {
        return fail("lanczos_hip_create: out of host memory while planning the tiled SpMV");
    }
    if (const char *force = getenv("LANCZOS_HIP_TILE_EVEN")) even = atoi(force) != 0;   // for tests
    // even blocks: 4096 rows, 2048 while x is small against the memory-side cache (random 21 per row: 1.65
    // against 1.74 ms at 10M rows, 9.86 against 9.33 at 50M; 8192 and 16384 rows: 1.84 / 10.0 and 2.32 / 11.1)
    const double x_mib = 8.0 * (double)n_ext / 1048576.0;
    int rb = even ? (x_mib <= 128.0 ? 11 : 12) : x_mib > 256.0 ? 14 : 13;
    if (const char *s = getenv("LANCZOS_HIP_TILE_ROWBITS"))
        if (atoi(s) >= 11 && atoi(s) <= 14) rb = atoi(s);
    if (rb == 11) return build_panels_shape<11, 128>(bits, even);
    if (rb == 12) return build_panels_shape<12, 256>(bits, even);
    if (rb == 13) return build_panels_shape<13, 512>(bits, even);
    return build_panels_shape<14, 1024>(bits, even);
}

int Engine::launch_spmv_panels(const double *x, const double *xdot, double *y, int *nparts) {
#define TILES_ARGS(p0, draw)                                                                                         \
    n, p0, tile_npieces, draw, (const TilePiece *)d_tpieces, (const uint32_t *)d_tblock_ptr, (const uint16_t *)d_trow,  \
        (const uint32_t *)d_tcol, (const uint32_t *)d_tbase, (const double *)d_pval, x, xdot, y, d_tside, d_tdots,      \
        d_tcounter
    const int groups = kGroups(tile_rowbits);
    if (tile_even) {
        // equal work in every piece: launches of `groups` workgroups that start together and sweep the
        // panels in step, so that an XCD's L2 holds one panel's slice of x at a time (one launch whose
        // workgroups draw the pieces as they finish loses that: 15.4 against 9.3 ms at 50M rows)
        for (int p0 = 0; p0 < tile_npieces; p0 += groups) {
            const dim3 grid(std::min(groups, tile_npieces - p0));
#define EVEN_LAUNCH(RB, T)                                                                                              \
    if (tile_packed)                                                                                                     \
        hipLaunchKernelGGL((k_spmv_tiles<false, true, RB, T>), grid, dim3(T), 0, stream, TILES_ARGS(p0, 0));             \
    else                                                                                                                 \
        hipLaunchKernelGGL((k_spmv_tiles<false, false, RB, T>), grid, dim3(T), 0, stream, TILES_ARGS(p0, 0))
            if (tile_rowbits == 11) {
                EVEN_LAUNCH(11, 128);
            } else if (tile_rowbits == 12) {
                EVEN_LAUNCH(12, 256);
            } else if (tile_rowbits == 13) {
                EVEN_LAUNCH(13, 512);
            } else {
                EVEN_LAUNCH(14, 1024);
            }
#undef EVEN_LAUNCH
        }
    } else {
        // uneven pieces, fullest blocks first: one launch whose workgroups draw pieces as they finish
        if (hipMemsetAsync(d_tcounter, 0, sizeof(int), stream) != hipSuccess)
            return fail("lanczos_hip: clearing the tiled SpMV's piece counter failed");
        const dim3 grid(std::min(groups, tile_npieces));
        if (tile_rowbits == 11)
            hipLaunchKernelGGL((k_spmv_tiles<true, false, 11, 128>), grid, dim3(128), 0, stream, TILES_ARGS(0, 1));
        else if (tile_rowbits == 12)
            hipLaunchKernelGGL((k_spmv_tiles<true, false, 12, 256>), grid, dim3(256), 0, stream, TILES_ARGS(0, 1));
        else if (tile_rowbits == 13)
            hipLaunchKernelGGL((k_spmv_tiles<true, false, 13, 512>), grid, dim3(512), 0, stream, TILES_ARGS(0, 1));
        else
            hipLaunchKernelGGL((k_spmv_tiles<true, false, 14, 1024>), grid, dim3(1024), 0, stream, TILES_ARGS(0, 1));
    }
#undef TILES_ARGS
    int count = tile_npieces;
    if (tile_nsplit > 0) {
        const int parts = (1 << tile_rowbits) / kBlock;
#define COMBINE_ARGS n, (const TileSplit *)d_tsplits, (const double *)d_tside, xdot, y, d_tdots + tile_npieces
        if (tile_rowbits == 11)
            hipLaunchKernelGGL(k_tiles_combine<11>, dim3(tile_nsplit * parts), dim3(kBlock), 0, stream, COMBINE_ARGS);
        else if (tile_rowbits == 12)
            hipLaunchKernelGGL(k_tiles_combine<12>, dim3(tile_nsplit * parts), dim3(kBlock), 0, stream, COMBINE_ARGS);
        else if (tile_rowbits == 13)
            hipLaunchKernelGGL(k_tiles_combine<13>, dim3(tile_nsplit * parts), dim3(kBlock), 0, stream, COMBINE_ARGS);
        else
            hipLaunchKernelGGL(k_tiles_combine<14>, dim3(tile_nsplit * parts), dim3(kBlock), 0, stream, COMBINE_ARGS);
#undef COMBINE_ARGS
        count += tile_nsplit * parts;
    }
    if (tile_nfar > 0) {
        const int groups = (tile_nfar + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_tiles_far, dim3(groups), dim3(kBlock), 0, stream, tile_nfar, (const uint32_t *)d_tfar_entry,
                           (const int32_t *)d_tfar_row, (const int32_t *)d_col, (const double *)d_val, x, xdot, y,
                           d_tdots + count);
        count += groups;
    }
    hipLaunchKernelGGL(k_tiles_fold, dim3(1), dim3(kBlock), 0, stream, count, (const double *)d_tdots, d_partials);
    *nparts = kBlock;
    const hipError_t st = hipGetLastError();
    if (st != hipSuccess) return fail("lanczos_hip: %s", hipGetErrorString(st));
    return 0;
}

}  // namespace lz
