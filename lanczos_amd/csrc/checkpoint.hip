// Checkpoint / resume of the iteration state.  The reference keeps V and T only
// in memory and frees them at exit (src/lanczos.c:195-207); long runs on large
// matrices want to survive that.  A checkpoint holds alpha[0..k), beta[0..k),
// every resident basis vector and the partial sums the next launch would read,
// so a resumed run continues bit for bit.
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "engine.h"

namespace {

struct Header {
    char magic[8];        // "LZCKPT4"
    int64_t n, ldv, nnz;
    uint64_t matrix_print;  // fingerprint of the device CRS arrays (row offsets, columns, values)
    int32_t reorth, max_iter, nvec, k, lazy, lazy_flushed_k, alpha_count, norm_count, max_blocks, hist_cap;
    int32_t reorth_steps, tracker_doubles, spmv_kind, host_k;
    char spmv_name[32];   // the SpMV form in full (kind, pattern count, plane walk): the saved partial sums are its
};
const char kMagic[8] = "LZCKPT4";

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// out[0] += sum over the words of `data` of mix64(word ^ mix64(index + salt)): a sum, so the order the
// threads arrive in does not matter, and position dependent, so permuted arrays differ.
template <typename T>
__global__ void __launch_bounds__(256) k_fingerprint(const T *__restrict__ data, int64_t count, uint64_t salt,
                                                     unsigned long long *__restrict__ out) {
    uint64_t acc = 0;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += step) {
        uint64_t w;
        if constexpr (sizeof(T) == 8) {
            w = *reinterpret_cast<const uint64_t *>(data + i);
        } else {
            w = (uint64_t)(uint32_t)data[i];
        }
        acc += mix64(w ^ mix64((uint64_t)i + salt));
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor((unsigned long long)acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}

// One pass over the matrix as it sits on the device (a fraction of a millisecond per GB).
int matrix_fingerprint(lz::Engine *e, uint64_t *out) {
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipError_t st = hipMemsetAsync(d, 0, sizeof(h), e->stream);
    const int blocks = e->max_blocks;
    if (st == hipSuccess) {
        if (e->rp64)
            hipLaunchKernelGGL((k_fingerprint<int64_t>), dim3(blocks), dim3(256), 0, e->stream, (const int64_t *)e->d_rp,
                               e->n + 1, (uint64_t)1 << 60, d);
        else
            hipLaunchKernelGGL((k_fingerprint<int32_t>), dim3(blocks), dim3(256), 0, e->stream, (const int32_t *)e->d_rp,
                               e->n + 1, (uint64_t)1 << 60, d);
        if (e->nnz > 0) {
            hipLaunchKernelGGL((k_fingerprint<int32_t>), dim3(blocks), dim3(256), 0, e->stream, (const int32_t *)e->d_col,
                               e->nnz, (uint64_t)2 << 60, d);
            hipLaunchKernelGGL((k_fingerprint<double>), dim3(blocks), dim3(256), 0, e->stream, (const double *)e->d_val,
                               e->nnz, (uint64_t)3 << 60, d);
        }
        st = hipGetLastError();
    }
    if (st == hipSuccess) st = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    (void)hipFree(d);
    *out = (uint64_t)h;
    return st != hipSuccess;
}

constexpr size_t kChunk = (size_t)8 << 20;  // doubles per staging piece (64 MiB)

int copy_out(FILE *fp, const double *dev, size_t count, std::vector<double> &stage) {
    for (size_t off = 0; off < count; off += kChunk) {
        const size_t c = count - off < kChunk ? count - off : kChunk;
        if (hipMemcpy(stage.data(), dev + off, c * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (fwrite(stage.data(), sizeof(double), c, fp) != c) return 1;
    }
    return 0;
}

int copy_in(FILE *fp, double *dev, size_t count, std::vector<double> &stage) {
    for (size_t off = 0; off < count; off += kChunk) {
        const size_t c = count - off < kChunk ? count - off : kChunk;
        if (fread(stage.data(), sizeof(double), c, fp) != c) return 1;
        if (hipMemcpy(dev + off, stage.data(), c * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    return 0;
}

}  // namespace

extern "C" int lanczos_hip_checkpoint_save(lanczos_hip_engine *eh, const char *path) {
    return lz::guarded("lanczos_hip_checkpoint_save", [&]() -> int {
    lz::Engine *e = reinterpret_cast<lz::Engine *>(eh);
    if (!e || !path) return lz::fail("lanczos_hip_checkpoint_save: NULL argument");
    if (!e->started) return lz::fail("lanczos_hip_checkpoint_save: no run to save");
    if (e->slab) return lz::fail("lanczos_hip_checkpoint_save: not available on a row-partitioned engine");
    if (e->ext_matvec)  // the matrix fingerprint is what ties a file to its operator; a host function has none
        return lz::fail("lanczos_hip_checkpoint_save: not available for an operator supplied as a host function");
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        return lz::fail("lanczos_hip_checkpoint_save: device not reachable");
    Header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    if (matrix_fingerprint(e, &h.matrix_print)) return lz::fail("lanczos_hip_checkpoint_save: fingerprint of the matrix failed");
    FILE *fp = fopen(path, "wb");
    if (!fp) return lz::fail("lanczos_hip_checkpoint_save: cannot open %s", path);
    h.n = e->n;
    h.ldv = e->ldv;
    h.nnz = e->nnz;
    h.spmv_kind = e->spmv_kind;
    snprintf(h.spmv_name, sizeof h.spmv_name, "%s", e->spmv_name);
    h.host_k = e->ckpt_host_k >= 0 ? std::min(e->ckpt_host_k, e->k) : e->k;
    h.reorth = e->reorth;
    h.max_iter = e->max_iter;
    h.nvec = e->nvec;
    h.k = e->k;
    h.lazy = e->lazy;
    h.lazy_flushed_k = e->lazy_flushed_k;
    h.alpha_count = e->alpha_count;
    h.norm_count = e->norm_count;
    h.max_blocks = e->max_blocks;
    h.hist_cap = e->hist_cap;
    h.reorth_steps = e->reorth_steps;
    // resident vectors: slots 0..k+... of a full basis, or all 3 ring slots
    const int slots = (e->reorth == LANCZOS_HIP_REORTH_NONE) ? 3 : std::min(e->k + 1, e->nvec);
    std::vector<double> tracker, stage;
    try {
        tracker.resize(lanczos_hip_omega_state_doubles(e->omega));
        stage.resize(std::min(kChunk, (size_t)e->ldv * (size_t)std::max(slots, 1)));
    } catch (const std::bad_alloc &) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_save: out of host memory");
    }
    if (e->omega) lanczos_hip_omega_state_save(e->omega, tracker.data());
    h.tracker_doubles = (int32_t)tracker.size();
    int bad = fwrite(&h, sizeof h, 1, fp) != 1;
    bad = bad || fwrite(tracker.data(), sizeof(double), tracker.size(), fp) != tracker.size();
    bad = bad || copy_out(fp, e->d_alpha, (size_t)e->k, stage) || copy_out(fp, e->d_beta, (size_t)e->k, stage);
    bad = bad || copy_out(fp, e->d_scal, 8, stage);
    bad = bad || copy_out(fp, e->d_partials, 2 * (size_t)e->max_blocks, stage);
    bad = bad || copy_out(fp, e->d_norm_parts, 2 * (size_t)e->max_blocks, stage);
    bad = bad || copy_out(fp, e->d_V, (size_t)e->ldv * (size_t)slots, stage);
    bad = (fclose(fp) != 0) || bad;
    if (bad) return lz::fail("lanczos_hip_checkpoint_save: write to %s failed", path);
    return 0;
    });
}

extern "C" int lanczos_hip_checkpoint_load(lanczos_hip_engine *eh, const char *path) {
    return lz::guarded("lanczos_hip_checkpoint_load", [&]() -> int {
    lz::Engine *e = reinterpret_cast<lz::Engine *>(eh);
    if (!e || !path) return lz::fail("lanczos_hip_checkpoint_load: NULL argument");
    if (e->slab) return lz::fail("lanczos_hip_checkpoint_load: not available on a row-partitioned engine");
    if (e->ext_matvec)
        return lz::fail("lanczos_hip_checkpoint_load: not available for an operator supplied as a host function");
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        return lz::fail("lanczos_hip_checkpoint_load: device not reachable");
    uint64_t print = 0;
    if (matrix_fingerprint(e, &print)) return lz::fail("lanczos_hip_checkpoint_load: fingerprint of the matrix failed");
    FILE *fp = fopen(path, "rb");
    if (!fp) return lz::fail("lanczos_hip_checkpoint_load: cannot open %s", path);
    Header h;
    if (fread(&h, sizeof h, 1, fp) != 1 || memcmp(h.magic, kMagic, 8) != 0) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_load: %s is not a checkpoint", path);
    }
    // the engine must have been created for the same matrix and mode
    if (h.n != e->n || h.ldv != e->ldv || h.reorth != e->reorth || h.max_blocks != e->max_blocks ||
        h.lazy != (int)e->lazy || h.k < 0 || h.k > e->hist_cap ||
        (h.tracker_doubles != 0) != (e->omega != nullptr) || (e->reorth != LANCZOS_HIP_REORTH_NONE && h.k + 1 > e->nvec)) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_load: checkpoint (n=%lld, reorth=%d, k=%d) does not fit this engine "
                        "(n=%lld, reorth=%d, max_iter=%d)", (long long)h.n, h.reorth, h.k, (long long)e->n, e->reorth,
                        e->max_iter);
    }
    h.spmv_name[sizeof h.spmv_name - 1] = 0;
    if (h.nnz != e->nnz || h.matrix_print != print || h.spmv_kind != e->spmv_kind || strcmp(h.spmv_name, e->spmv_name) != 0) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_load: the checkpoint was written for another matrix or SpMV form (nnz=%lld, "
                        "fingerprint %016llx, form %s; this engine: nnz=%lld, %016llx, form %s)", (long long)h.nnz,
                        (unsigned long long)h.matrix_print, h.spmv_name, (long long)e->nnz, (unsigned long long)print,
                        e->spmv_name);
    }
    // counts that become loop bounds and buffer sizes: inside what this engine allocated
    const int64_t tracker_max = 64 + 8 * (int64_t)std::max(h.max_iter, 2);
    if (h.alpha_count < 0 || h.alpha_count > 2 * e->max_blocks || h.norm_count < 0 || h.norm_count > e->max_blocks ||
        h.lazy_flushed_k < 0 || h.lazy_flushed_k > h.k || h.host_k < 0 || h.host_k > h.k || h.reorth_steps < 0 ||
        h.max_iter < 2 || h.tracker_doubles < 0 || h.tracker_doubles > tracker_max) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_load: %s has a damaged header", path);
    }
    const int slots = (e->reorth == LANCZOS_HIP_REORTH_NONE) ? 3 : std::min(h.k + 1, e->nvec);
    std::vector<double> stage, tracker, fitted;
    try {
        stage.resize(std::min(kChunk, (size_t)e->ldv * (size_t)std::max(slots, 1)));
        tracker.resize((size_t)h.tracker_doubles);
        fitted.resize(lanczos_hip_omega_state_doubles(e->omega));
    } catch (const std::bad_alloc &) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_load: out of host memory");
    }
    int bad = fread(tracker.data(), sizeof(double), tracker.size(), fp) != tracker.size();
    if (!bad && e->omega)  // the saving engine may have been created for another max_iter
        bad = lanczos_hip_omega_state_fit(e->omega, tracker.data(), tracker.size(), fitted.data()) != 0 ||
              lanczos_hip_omega_state_load(e->omega, fitted.data()) != 0;
    bad = bad || hipMemset(e->d_V, 0, sizeof(double) * (size_t)e->ldv * (size_t)e->nvec) != hipSuccess;
    bad = bad || copy_in(fp, e->d_alpha, (size_t)h.k, stage) || copy_in(fp, e->d_beta, (size_t)h.k, stage);
    bad = bad || copy_in(fp, e->d_scal, 8, stage);
    bad = bad || copy_in(fp, e->d_partials, 2 * (size_t)e->max_blocks, stage);
    bad = bad || copy_in(fp, e->d_norm_parts, 2 * (size_t)e->max_blocks, stage);
    bad = bad || copy_in(fp, e->d_V, (size_t)e->ldv * (size_t)slots, stage);
    fclose(fp);
    if (bad) return lz::fail("lanczos_hip_checkpoint_load: read from %s failed", path);
    e->k = h.k;
    e->lazy_flushed_k = h.lazy_flushed_k;
    e->spec_spmv_k = -1;
    e->alpha_count = h.alpha_count;
    e->norm_count = h.norm_count;
    e->reorth_steps = h.reorth_steps;
    e->ckpt_host_k = h.host_k;
    if (e->h_hist) {  // the host mirror the selective rule reads beta_{k-1} from
        std::vector<double> a, b;
        try {
            a.resize((size_t)h.k);
            b.resize((size_t)h.k);
        } catch (const std::bad_alloc &) {
            return lz::fail("lanczos_hip_checkpoint_load: out of host memory");
        }
        if (h.k > 0 && (hipMemcpy(a.data(), e->d_alpha, sizeof(double) * h.k, hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(b.data(), e->d_beta, sizeof(double) * h.k, hipMemcpyDeviceToHost) != hipSuccess))
            return lz::fail("lanczos_hip_checkpoint_load: history read-back failed");
        for (int i = 0; i < h.k; ++i) {
            e->h_hist[2 * (size_t)i] = a[(size_t)i];
            e->h_hist[2 * (size_t)i + 1] = b[(size_t)i];
        }
    }
    e->started = true;
    return 0;
    });
}
