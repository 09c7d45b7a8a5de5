// Row ranges of equal weight for the 8 XCDs.  The SpMV / SpMM kernels give workgroup b the
// (b % 8)-th range of row units (workgroups b and b + 8 share an XCD and its L2) and deal the units
// of a range round-robin over the XCD's workgroups.  Ranges of equal row count starve seven XCDs
// on a power-law graph, whose heavy rows sit together; these hold equal numbers of entries instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lz {

// out[x] = first unit of range x (x = 0..8), a unit being `unit_rows` consecutive rows; range x
// starts at the first unit boundary whose row offset reaches x/8 of the entries plus x/8 of the
// rows (rows weigh one entry each, so empty stretches are shared out too).
template <typename RP>
__global__ void k_balance_split(int64_t n, const RP *__restrict__ rp, int64_t unit_rows, int64_t units,
                                long long *__restrict__ out) {
    const int x = threadIdx.x;
    if (x > 8) return;
    if (x == 0 || x == 8) {
        out[x] = x == 0 ? 0 : units;
        return;
    }
    // (rp may point into the middle of a matrix -- the rows of a slab that read nothing remote -- so offsets
    // are taken relative to the first one)
    const double first = (double)rp[0];
    const double total = (double)rp[n] - first + (double)n;
    const double want = total * x / 8.0;
    int64_t lo = 0, hi = units;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t r = mid * unit_rows < n ? mid * unit_rows : n;
        if ((double)rp[r] - first + (double)r < want) lo = mid + 1;
        else hi = mid;
    }
    out[x] = lo;
}

}  // namespace lz
