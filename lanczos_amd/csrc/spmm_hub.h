// Sparse product of the block variant for the rows of a power-law graph that hold most of its entries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lz {
struct Engine;
}

namespace lzb {

constexpr int kHubClasses = 16;  // 8 column classes x {columns meant to stay in L2, the others}

// One entry of the regrouped stream (16 bytes, one load per lane)
struct HubEntry {
    int32_t col;   // column of A = row of X
    int32_t h;     // index of the entry's row in the list of hub rows
    double val;
};

struct HubLayout {
    uint32_t e_lo[kHubClasses + 1];  // first stream position of every class
    uint32_t s_lo[kHubClasses + 1];  // first slice of every class
};

struct HubStream {
    int64_t H = 0;            // hub rows
    int64_t entries = 0;      // entries in them
    int64_t slices = 0;       // slices over all classes
    int slice_log2 = 8;       // log2 of the entries eight lanes walk before they move on
    int64_t hot_entries = 0;  // entries whose column belongs to the set meant to stay in L2
    int classes = 8;          // 8 column classes; 16 when the columns are also split by popularity
    HubLayout lay;
    int *d_rows = nullptr;           // [H] the hub rows, ascending
    HubEntry *d_ent = nullptr;       // [entries] sorted by (class, hub row), CRS order inside
    uint32_t *d_vptr = nullptr;      // [classes H + 1] first stream position of (class, hub row)
    double *d_part = nullptr;        // [classes H][16] partial rows
    int *d_carry_h = nullptr;        // [2 slices] hub row of a slice's first / last run when it goes on in a neighbour, or -1
    double *d_carry_val = nullptr;   // [2 slices][16]
    bool ready() const { return d_ent != nullptr; }
};

// rows[0..count): the rows with more than kMedRow entries (any order, device memory).  degree_floor: a column
// belongs to the hot set when it holds at least that many entries (0: every column).  On failure (no memory)
// the stream stays empty and the caller keeps its row-wise kernels.
int hub_build(lz::Engine *base, const int *d_rows_a, int count_a, const int *d_rows_b, int count_b, int64_t degree_floor,
              HubStream *out);
// Y[hub rows] = A[hub rows, :] X, the other rows of Y untouched
int hub_spmm(const HubStream &hs, hipStream_t stream, int cus, const double *X, double *Y);
void hub_free(HubStream *hs);
// d_out[0..8] (device): the rows of the engine's matrix cut into eight ranges of whole units of unit_rows rows, one
// per XCD, of equal cost to the row-wise kernel (a row of at most long_row entries: 2 + its entries; a longer one: 1)
int row_cost_split(lz::Engine *base, int long_row, int64_t unit_rows, int64_t units, long long *d_out);

}  // namespace lzb
