// lanczos_hip_spmv_noalloc: the SpMV of an engine on device vectors, without any allocation.
//
// lanczos_hip_spmv stages its host vectors through two hipMalloc'd buffers per call and the SpMV
// kernels leave their partial sums (and, in the merge form, the rows shared between tiles; in the
// tiled form, the sums of the pieces of row blocks too full for one workgroup) in buffers the
// engine allocated.  Callers that must not allocate -- a stream under HIP graph
// capture, a serving loop, a harness that forbids hipMalloc -- get the same kernels here with that
// scratch memory in a fixed array that lives in the library's device image: kScratchBytes per GPU
// the library runs on, held until the process ends.
//
// For the duration of the call the engine's own scratch pointers are swapped for the fixed array, so
// the engine must not be used from another thread meanwhile; and every call on a GPU shares that
// GPU's array, so calls that overlap on one GPU (two engines, two streams) race.
#include <mutex>

#include "engine.h"

namespace {

constexpr int kMaxParts = 4096;              // per-workgroup partials of x . y (two per workgroup slot)
constexpr int64_t kMaxMergeTiles = 131072;   // merge form: tiles of 512 entries, i.e. nnz <= 67 108 864
constexpr int kMergeTile = 512;

struct Scratch {
    double parts[kMaxParts];
    double carry_val[2 * kMaxMergeTiles];
    int carry_row[2 * kMaxMergeTiles];
};
constexpr size_t kScratchBytes = sizeof(Scratch);  // 3 178 496 bytes

// In host code this name only stands for the symbol; the memory is reached through the address
// hipGetSymbolAddress returns for the current device (every GPU holds its own copy).
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_mu;
Scratch *g_ptr[kMaxDevices] = {nullptr};

// The device's copy of the array; looked up once per device, with that device current.
int scratch_for_device(int device, Scratch **out) {
    if (device < 0 || device >= kMaxDevices) return lz::fail("lanczos_hip_spmv_noalloc: device %d out of range", device);
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ptr[device]) {
        void *p = nullptr;
        const hipError_t st = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (st != hipSuccess || !p)
            return lz::fail("lanczos_hip_spmv_noalloc: hipGetSymbolAddress failed on device %d: %s", device,
                            hipGetErrorString(st));
        g_ptr[device] = static_cast<Scratch *>(p);
    }
    *out = g_ptr[device];
    return 0;
}

// out[0] = sum(parts[0..count)) in a fixed order (one workgroup)
__global__ void __launch_bounds__(256) k_noalloc_sum(const double *__restrict__ parts, int count, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += parts[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

}  // namespace

namespace lz {

int Engine::spmv_noalloc(const double *x_dev, double *y_dev, double *dot_dev) {
    if (hipSetDevice(device) != hipSuccess) return fail("lanczos_hip_spmv_noalloc: hipSetDevice(%d) failed", device);
    if (ext_matvec)  // the engine holds an empty matrix then: the product is the caller's host function
        return fail("lanczos_hip_spmv_noalloc: this engine's operator is a host function (lanczos_hip_set_operator); "
                    "there is no device product to launch");
    if (2 * max_blocks > kMaxParts)
        return fail("lanczos_hip_spmv_noalloc: %d workgroups leave more partial sums than the fixed scratch holds (%d)",
                    2 * max_blocks, kMaxParts);
    const int64_t tiles = (nnz + kMergeTile - 1) / kMergeTile;
    if (spmv_kind == LANCZOS_HIP_SPMV_MERGE && tiles > kMaxMergeTiles)
        return fail("lanczos_hip_spmv_noalloc: the matrix has %lld entries; on the merge SpMV this entry point takes "
                    "at most %lld (fixed scratch of %zu bytes)",
                    (long long)nnz, (long long)(kMaxMergeTiles * kMergeTile), kScratchBytes);
    if (spmv_kind == LANCZOS_HIP_SPMV_PANEL && ((int64_t)tile_nslots << tile_rowbits) > 2 * kMaxMergeTiles)
        return fail("lanczos_hip_spmv_noalloc: the tiled SpMV of this matrix cuts its fullest row blocks into %d pieces; "
                    "this entry point takes at most %lld (fixed scratch of %zu bytes)",
                    tile_nslots, (long long)((2 * kMaxMergeTiles) >> tile_rowbits), kScratchBytes);
    Scratch *s = nullptr;
    if (int rc = scratch_for_device(device, &s)) return rc;
    // The kernels take their scratch as arguments; hand them the fixed array for this one launch.
    // Every partial and every carry slot the reductions read is written by the launch itself, so
    // what an earlier call left in the array is never seen.
    double *const keep_parts = d_partials, *const keep_cval = d_carry_val;
    int *const keep_crow = d_carry_row;
    const bool keep_profile = profile;
    d_partials = s->parts;
    if (spmv_kind == LANCZOS_HIP_SPMV_MERGE) {
        d_carry_val = s->carry_val;
        d_carry_row = s->carry_row;
    }
    double *const keep_side = d_tside;
    if (spmv_kind == LANCZOS_HIP_SPMV_PANEL && tile_nslots > 0) d_tside = s->carry_val;
    profile = false;  // the event pool may have to create events
    int nparts = 0;
    int rc = launch_spmv(x_dev, x_dev, y_dev, &nparts);
    d_partials = keep_parts;
    d_carry_val = keep_cval;
    d_carry_row = keep_crow;
    d_tside = keep_side;
    profile = keep_profile;
    if (rc) return rc;
    if (dot_dev) {
        hipLaunchKernelGGL(k_noalloc_sum, dim3(1), dim3(256), 0, stream, (const double *)s->parts, nparts, dot_dev);
        if (hipGetLastError() != hipSuccess) return fail("lanczos_hip_spmv_noalloc: launch failed");
    }
    return 0;
}

}  // namespace lz

extern "C" int lanczos_hip_spmv_noalloc(lanczos_hip_engine *e, const double *x_dev, double *y_dev, double *dot_dev) {
    if (!e || !x_dev || !y_dev) return lz::fail("lanczos_hip_spmv_noalloc: NULL argument");
    return reinterpret_cast<lz::Engine *>(e)->spmv_noalloc(x_dev, y_dev, dot_dev);
}
