// COO -> CRS on the device, bit-exact with the reference's host build
// (src/crs.c:8-153): per-row counts, exclusive prefix sum, and a STABLE
// distribution of the entries so that a row keeps its entries in input order and
// duplicates stay separate.  The host build is two serial passes over nnz and
// then 16 B/nnz cross PCIe again as CRS; here the triplets are uploaded once and
// the CRS arrays are produced where the SpMV reads them.
//
// Stability comes from a stable LSD radix sort of (row key, entry index) pairs
// (rocPRIM through hipCUB; setup path, not the iteration).
#include <hipcub/hipcub.hpp>

#include <stdio.h>
#include <string.h>

#include <vector>

#include "engine.h"

namespace lz {

static constexpr int kBlock = 256;  // threads per workgroup, as everywhere in the engine

#define HIP_TRY_CB(call)                                                                              \
    do {                                                                                              \
        hipError_t st__ = (call);                                                                     \
        if (st__ != hipSuccess)                                                                       \
            return fail("HIP API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,           \
                        hipGetErrorString(st__), (int)st__);                                          \
    } while (0)

// keys[i] = row of entry i (after the optional exchange), idx[i] = i; range check
// as src/crs.c:68-77 (the first offending entry is reported through `bad`).
__global__ void __launch_bounds__(kBlock)
k_coo_keys(const Coo *__restrict__ coo, int64_t len, int64_t dim, int swap, uint32_t *__restrict__ keys,
           uint32_t *__restrict__ idx, unsigned long long *__restrict__ bad) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += step) {
        const Coo e = coo[i];
        const int r = swap ? e.index_column : e.index_row;
        const int c = swap ? e.index_row : e.index_column;
        if (r < 0 || c < 0 || r >= dim || c >= dim) {
            atomicMin(bad, (unsigned long long)i);
            keys[i] = 0;
        } else {
            keys[i] = (uint32_t)r;
        }
        idx[i] = (uint32_t)i;
    }
}

// after the sort: position p holds entry idx[p]
__global__ void __launch_bounds__(kBlock)
k_coo_place(const Coo *__restrict__ coo, int64_t len, int swap, const uint32_t *__restrict__ idx,
            double *__restrict__ val, int32_t *__restrict__ col) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < len; p += step) {
        const Coo e = coo[idx[p]];
        val[p] = e.value;
        col[p] = swap ? e.index_row : e.index_column;
    }
}

// row offsets from the sorted keys: rp[r] = number of keys < r (binary search per row)
template <typename RP>
__global__ void __launch_bounds__(kBlock)
k_rp_from_keys(const uint32_t *__restrict__ keys, int64_t len, int64_t dim, RP *__restrict__ rp) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r <= dim; r += step) {
        int64_t lo = 0, hi = len;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        rp[r] = (RP)lo;
    }
}

int Engine::build_from_coo(const Coo *coo, int64_t length, int swap) {
    if (length >= ((int64_t)1 << 32))
        return fail("lanczos_hip_create_from_coo: %lld entries; the device build indexes entries with 32 bits",
                    (long long)length);
    if (length == 0) {
        HIP_TRY_CB(hipMemsetAsync(d_rp, 0, (rp64 ? 8 : 4) * (n + 1), stream));
        HIP_TRY_CB(hipStreamSynchronize(stream));
        return 0;
    }
    Coo *d_coo = nullptr;
    uint32_t *d_keys[2] = {nullptr, nullptr}, *d_idx[2] = {nullptr, nullptr};
    unsigned long long *d_bad = nullptr;
    void *d_tmp = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        (void)hipFree(d_coo);
        (void)hipFree(d_keys[0]);
        (void)hipFree(d_keys[1]);
        (void)hipFree(d_idx[0]);
        (void)hipFree(d_idx[1]);
        (void)hipFree(d_bad);
        (void)hipFree(d_tmp);
    };
#define CB_TRY(call)                                                                                   \
    do {                                                                                               \
        hipError_t st__ = (call);                                                                      \
        if (st__ != hipSuccess) {                                                                      \
            cleanup();                                                                                 \
            return fail("HIP API failed at %s:%d with error: %s (%d)", __FILE__, __LINE__,            \
                        hipGetErrorString(st__), (int)st__);                                           \
        }                                                                                              \
    } while (0)
    CB_TRY(hipMalloc(&d_coo, sizeof(Coo) * length));
    for (int b = 0; b < 2; ++b) {
        CB_TRY(hipMalloc(&d_keys[b], sizeof(uint32_t) * length));
        CB_TRY(hipMalloc(&d_idx[b], sizeof(uint32_t) * length));
    }
    CB_TRY(hipMalloc(&d_bad, sizeof(unsigned long long)));
    CB_TRY(hipMemsetAsync(d_bad, 0xFF, sizeof(unsigned long long), stream));
    CB_TRY(hipMemcpyAsync(d_coo, coo, sizeof(Coo) * length, hipMemcpyHostToDevice, stream));
    const int blocks = (int)std::min<int64_t>(max_blocks, (length + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_coo_keys, dim3(blocks), dim3(kBlock), 0, stream, d_coo, length, n, swap, d_keys[0],
                       d_idx[0], d_bad);
    CB_TRY(hipGetLastError());
    unsigned long long bad = ~0ull;
    CB_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream));
    CB_TRY(hipStreamSynchronize(stream));
    if (bad != ~0ull) {
        const Coo &e = coo[bad];
        const int r = swap ? e.index_column : e.index_row, c = swap ? e.index_row : e.index_column;
        cleanup();
        // the message of src/crs.c:69-71
        return fail("convert_from_coo: index out of range at entry %llu (row=%d, col=%d, dim=%lld)", bad, r, c,
                    (long long)n);
    }
    int end_bit = 1;
    while (end_bit < 32 && ((int64_t)1 << end_bit) < n) end_bit++;
    hipcub::DoubleBuffer<uint32_t> keys(d_keys[0], d_keys[1]);
    hipcub::DoubleBuffer<uint32_t> vals(d_idx[0], d_idx[1]);
    size_t tmp_bytes = 0;
    CB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, vals, (unsigned int)length, 0, end_bit, stream));
    CB_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    CB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, keys, vals, (unsigned int)length, 0, end_bit, stream));
    hipLaunchKernelGGL(k_coo_place, dim3(blocks), dim3(kBlock), 0, stream, d_coo, length, swap, vals.Current(), d_val,
                       d_col);
    const int rblocks = (int)std::min<int64_t>(max_blocks, (n + 1 + kBlock - 1) / kBlock);
    if (rp64)
        hipLaunchKernelGGL((k_rp_from_keys<int64_t>), dim3(rblocks), dim3(kBlock), 0, stream, keys.Current(), length, n,
                           (int64_t *)d_rp);
    else
        hipLaunchKernelGGL((k_rp_from_keys<int32_t>), dim3(rblocks), dim3(kBlock), 0, stream, keys.Current(), length, n,
                           (int32_t *)d_rp);
    CB_TRY(hipGetLastError());
    CB_TRY(hipStreamSynchronize(stream));
    cleanup();
    (void)rc;
    return 0;
#undef CB_TRY
}

int Engine::create_from_coo(Engine **out, const Mat_Coo *coo, int swap, int max_iter,
                            const lanczos_hip_options *opt_in) {
    *out = nullptr;
    lanczos_hip_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.device = -1;
    if (opt_in) memcpy(&opt, opt_in, std::min(sizeof(opt), opt_in->struct_size));
    if (opt.reorth < 0 || opt.reorth > LANCZOS_HIP_REORTH_SELECTIVE)
        return fail("lanczos_hip_create_from_coo: bad reorth mode %d", opt.reorth);
    if (opt.spmv_kernel < 0 || opt.spmv_kernel > LANCZOS_HIP_SPMV_MERGE)
        return fail("lanczos_hip_create_from_coo: bad spmv_kernel %d", opt.spmv_kernel);
    if (!coo || (coo->length > 0 && !coo->data)) return fail("convert_from_coo: mat_coo is NULL");
    if (coo->dimension == 0) return fail("convert_from_coo: dimension is 0");
    if (coo->dimension > (size_t)INT32_MAX)
        return fail("Matrix dimension too large for HIP CRS (dim=%zu)", coo->dimension);
    if (max_iter < 2) return fail("max_iter must be >= 2");
    if (opt.x_ext || opt.basis) return fail("lanczos_hip_create_from_coo: row-partitioned engines take CRS slabs");
    Engine *e = new Engine();
    int rc = e->setup_device(opt, (int64_t)coo->dimension, (int64_t)coo->dimension, (int64_t)coo->length, max_iter);
    if (rc == 0) rc = e->build_from_coo(coo->data, (int64_t)coo->length, swap);
    if (rc == 0) rc = e->choose_spmv(opt.spmv_kernel);
    if (rc == 0) rc = e->alloc_state();
    if (rc != 0) {
        delete e;
        return rc;
    }
    *out = e;
    return 0;
}

}  // namespace lz

extern "C" int lanczos_hip_create_from_coo(lanczos_hip_engine **out, const Mat_Coo *mat_coo, int swap_row_column,
                                           int max_iter, const lanczos_hip_options *opt) {
    return lz::guarded("lanczos_hip_create_from_coo", [&]() -> int {
    if (!out) return lz::fail("lanczos_hip_create_from_coo: NULL out");
    lz::Engine *e = nullptr;
    int rc = lz::Engine::create_from_coo(&e, mat_coo, swap_row_column, max_iter, opt);
    *out = reinterpret_cast<lanczos_hip_engine *>(e);
    return rc;
    });
}

// Test hook: the device CRS widened back to the layout of Mat_Crs.
extern "C" int lanczos_hip_download_crs(lanczos_hip_engine *eh, int64_t *row_head_indexes, int64_t *column_index,
                                        double *values) {
    return lz::guarded("lanczos_hip_download_crs", [&]() -> int {
    lz::Engine *e = reinterpret_cast<lz::Engine *>(eh);
    if (!e || !row_head_indexes) return lz::fail("lanczos_hip_download_crs: NULL argument");
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        return lz::fail("lanczos_hip_download_crs: device not reachable");
    hipError_t st = hipSuccess;
    if (e->rp64) {
        st = hipMemcpy(row_head_indexes, e->d_rp, 8 * (e->n + 1), hipMemcpyDeviceToHost);
    } else {
        std::vector<int32_t> t((size_t)e->n + 1);
        st = hipMemcpy(t.data(), e->d_rp, 4 * (e->n + 1), hipMemcpyDeviceToHost);
        for (int64_t i = 0; i <= e->n; ++i) row_head_indexes[i] = t[(size_t)i];
    }
    if (st == hipSuccess && e->nnz > 0 && column_index) {
        std::vector<int32_t> t((size_t)e->nnz);
        st = hipMemcpy(t.data(), e->d_col, 4 * e->nnz, hipMemcpyDeviceToHost);
        for (int64_t i = 0; i < e->nnz; ++i) column_index[i] = t[(size_t)i];
    }
    if (st == hipSuccess && e->nnz > 0 && values)
        st = hipMemcpy(values, e->d_val, 8 * e->nnz, hipMemcpyDeviceToHost);
    if (st != hipSuccess) return lz::fail("lanczos_hip_download_crs: %s", hipGetErrorString(st));
    return 0;
    });
}
