// Checkpoint / resume of the iteration state.  The reference keeps V and T only
// in memory and frees them at exit (src/lanczos.c:195-207); long runs on large
// matrices want to survive that.  A checkpoint holds alpha[0..k), beta[0..k),
// every resident basis vector and the partial sums the next launch would read,
// so a resumed run continues bit for bit.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "engine.h"

namespace {

struct Header {
    char magic[8];        // "LZCKPT2"
    int64_t n, ldv;
    int32_t reorth, max_iter, nvec, k, lazy, lazy_flushed_k, alpha_count, norm_count, max_blocks, hist_cap;
    int32_t reorth_steps, tracker_doubles;
};

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
    lz::Engine *e = reinterpret_cast<lz::Engine *>(eh);
    if (!e || !path) return lz::fail("lanczos_hip_checkpoint_save: NULL argument");
    if (!e->started) return lz::fail("lanczos_hip_checkpoint_save: no run to save");
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        return lz::fail("lanczos_hip_checkpoint_save: device not reachable");
    FILE *fp = fopen(path, "wb");
    if (!fp) return lz::fail("lanczos_hip_checkpoint_save: cannot open %s", path);
    Header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "LZCKPT2", 8);
    h.n = e->n;
    h.ldv = e->ldv;
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
    std::vector<double> tracker(lanczos_hip_omega_state_doubles(e->omega));
    if (e->omega) lanczos_hip_omega_state_save(e->omega, tracker.data());
    h.tracker_doubles = (int32_t)tracker.size();
    // resident vectors: slots 0..k+... of a full basis, or all 3 ring slots
    const int slots = (e->reorth == LANCZOS_HIP_REORTH_NONE) ? 3 : std::min(e->k + 1, e->nvec);
    std::vector<double> stage(std::min(kChunk, (size_t)e->ldv * (size_t)std::max(slots, 1)));
    int bad = fwrite(&h, sizeof h, 1, fp) != 1;
    bad = bad || fwrite(tracker.data(), sizeof(double), tracker.size(), fp) != tracker.size();
    bad = bad || copy_out(fp, e->d_alpha, (size_t)e->k, stage) || copy_out(fp, e->d_beta, (size_t)e->k, stage);
    bad = bad || copy_out(fp, e->d_scal, 8, stage);
    bad = bad || copy_out(fp, e->d_partials, (size_t)e->max_blocks, stage);
    bad = bad || copy_out(fp, e->d_norm_parts, 2 * (size_t)e->max_blocks, stage);
    bad = bad || copy_out(fp, e->d_V, (size_t)e->ldv * (size_t)slots, stage);
    bad = (fclose(fp) != 0) || bad;
    if (bad) return lz::fail("lanczos_hip_checkpoint_save: write to %s failed", path);
    return 0;
}

extern "C" int lanczos_hip_checkpoint_load(lanczos_hip_engine *eh, const char *path) {
    lz::Engine *e = reinterpret_cast<lz::Engine *>(eh);
    if (!e || !path) return lz::fail("lanczos_hip_checkpoint_load: NULL argument");
    if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        return lz::fail("lanczos_hip_checkpoint_load: device not reachable");
    FILE *fp = fopen(path, "rb");
    if (!fp) return lz::fail("lanczos_hip_checkpoint_load: cannot open %s", path);
    Header h;
    if (fread(&h, sizeof h, 1, fp) != 1 || memcmp(h.magic, "LZCKPT2", 8) != 0) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_load: %s is not a checkpoint", path);
    }
    // the engine must have been created for the same matrix shape and mode
    if (h.n != e->n || h.ldv != e->ldv || h.reorth != e->reorth || h.max_blocks != e->max_blocks ||
        h.lazy != (int)e->lazy || h.k > e->hist_cap ||
        (h.tracker_doubles != 0) != (e->omega != nullptr) || (e->reorth != LANCZOS_HIP_REORTH_NONE && h.k + 1 > e->nvec)) {
        fclose(fp);
        return lz::fail("lanczos_hip_checkpoint_load: checkpoint (n=%lld, reorth=%d, k=%d) does not fit this engine "
                        "(n=%lld, reorth=%d, max_iter=%d)", (long long)h.n, h.reorth, h.k, (long long)e->n, e->reorth,
                        e->max_iter);
    }
    const int slots = (e->reorth == LANCZOS_HIP_REORTH_NONE) ? 3 : std::min(h.k + 1, e->nvec);
    std::vector<double> stage(std::min(kChunk, (size_t)e->ldv * (size_t)std::max(slots, 1)));
    std::vector<double> tracker((size_t)h.tracker_doubles);
    int bad = fread(tracker.data(), sizeof(double), tracker.size(), fp) != tracker.size();
    if (!bad && e->omega) {  // the saving engine may have been created for another max_iter
        std::vector<double> fitted(lanczos_hip_omega_state_doubles(e->omega));
        bad = lanczos_hip_omega_state_fit(e->omega, tracker.data(), tracker.size(), fitted.data()) != 0 ||
              lanczos_hip_omega_state_load(e->omega, fitted.data()) != 0;
    }
    bad = bad || hipMemset(e->d_V, 0, sizeof(double) * (size_t)e->ldv * (size_t)e->nvec) != hipSuccess;
    bad = bad || copy_in(fp, e->d_alpha, (size_t)h.k, stage) || copy_in(fp, e->d_beta, (size_t)h.k, stage);
    bad = bad || copy_in(fp, e->d_scal, 8, stage);
    bad = bad || copy_in(fp, e->d_partials, (size_t)e->max_blocks, stage);
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
    if (e->h_hist) {  // the host mirror the selective rule reads beta_{k-1} from
        std::vector<double> a((size_t)h.k), b((size_t)h.k);
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
}
