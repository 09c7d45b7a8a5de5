// lanczos_hip_crs(): the drop-in for lanczos_cuda_crs (reference
// src/lanczos_cuda.c:281-788) with the termination rules of the CPU solver
// (src/lanczos.c:104-191), which is the parity target.
//
// The GPU never waits for the host: iterations are enqueued a few ahead of the
// one whose alpha/beta the host is looking at, the two scalars of every
// iteration land in pinned host memory straight from the kernels that produce
// them, and the host solves the small tridiagonal eigenproblem while the next
// iterations run.  Stopping early only discards the iterations in flight.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "engine.h"

namespace {

double now_sec() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

int env_int(const char *name, int dflt) {
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

int env_reorth() {
    const char *s = getenv("LANCZOS_HIP_REORTH");
    if (!s || !*s || !strcmp(s, "full")) return LANCZOS_HIP_REORTH_FULL;
    if (!strcmp(s, "full2")) return LANCZOS_HIP_REORTH_FULL2;
    if (!strcmp(s, "mgs")) return LANCZOS_HIP_REORTH_MGS;
    if (!strcmp(s, "cuda_compat")) return LANCZOS_HIP_REORTH_CUDA_COMPAT;
    if (!strcmp(s, "none")) return LANCZOS_HIP_REORTH_NONE;
    if (!strcmp(s, "selective")) return LANCZOS_HIP_REORTH_SELECTIVE;
    return -1;
}

// The solver behind both entry points: `mat` is the matrix to upload, or NULL when the operator is
// the host function `op` (the engine then holds an empty matrix of that dimension).
int solve(const Mat_Crs *mat, lanczos_hip_matvec_fn *op, const void *op_mat, size_t dimension, double eigenvalues[],
          double *eigenvectors[], int nth_eig, int max_iter, double threshold) {
    const size_t length = mat ? mat->length : 0;
    // Neither reference solver fills `eigenvectors` (src/lanczos.c:20, src/lanczos_cuda.c:284), and the
    // reference's driver passes an uninitialised pointer array (src/main.c:20): it is only written to
    // when LANCZOS_HIP_EIGENVECTORS=1 says that the caller's pointers are real.
    const bool want_vectors = eigenvectors != NULL && env_int("LANCZOS_HIP_EIGENVECTORS", 0) != 0;
    if ((mat == NULL && op == NULL) || eigenvalues == NULL || dimension == 0) return EXIT_FAILURE;  // src/lanczos_cuda.c:316-318
    if (dimension > (size_t)INT_MAX) {                            // src/lanczos_cuda.c:319-322
        printf("Matrix dimension too large for HIP CRS (dim=%zu)\n", dimension);
        return EXIT_FAILURE;
    }
    if (max_iter < 2) {  // src/lanczos_cuda.c:324-327
        printf("max_iter must be >= 2\n");
        return EXIT_FAILURE;
    }
    if (nth_eig < 1) {  // src/lanczos.c:32-35
        printf("nth_eig must be >= 1\n");
        return EXIT_FAILURE;
    }
    memset(eigenvalues, 0, (size_t)nth_eig * sizeof(double));  // src/lanczos_cuda.c:359

    const double t_start = now_sec();
    double time_init = 0.0, time_diag = 0.0;
    int status = EXIT_FAILURE;
    const bool quiet = env_int("LANCZOS_HIP_QUIET", 0) != 0;
    const int window = std::max(1, env_int("LANCZOS_HIP_WINDOW", 4));

    lanczos_hip_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
    opt.device = env_int("LANCZOS_HIP_DEVICE", -1);
    opt.reorth = env_reorth();
    opt.profile = env_int("LANCZOS_HIP_PROFILE", 1);
    if (opt.reorth < 0) {
        printf("LANCZOS_HIP_REORTH must be one of full|full2|mgs|cuda_compat|none|selective\n");
        return EXIT_FAILURE;
    }
    unsigned long long seed = 123456789ULL;  // the constant of src/lanczos_cuda.c:395
    if (const char *s = getenv("LANCZOS_HIP_SEED"))
        if (*s) seed = strtoull(s, NULL, 10);

    lz::Engine *e = nullptr;
    lanczos_hip_multi *mg = nullptr;  // several GPUs: e is then the engine of slab 0, owned by mg
    const int ngpu = env_int("LANCZOS_HIP_NGPU", 1);
    const bool loopback = env_int("LANCZOS_HIP_MULTI_LOOPBACK", 0) != 0;
    const bool force_slabs = env_int("LANCZOS_HIP_MULTI_FORCE", 0) != 0;  // the row-slab path with one slab (rehearsal)
    std::vector<hipEvent_t> done((size_t)window, nullptr);
    std::vector<double> a((size_t)max_iter, 0.0), b((size_t)max_iter, 0.0), last((size_t)nth_eig, 0.0);
    const int total = max_iter - 1;
    int enq = 0;
    int steps_used = 0;
    int replayed = 0;  // iterations restored from a checkpoint: their scalars are there, nothing to wait for
    int printed = 0;   // of those, the ones whose lines the interrupted run had already printed
    bool reached_max = true;
    int exit_kind = 0;  // 0 max-iter (src/lanczos.c:193), 1 Ritz values stood still (:127-133), 2 beta (:185-188)
    const char *ckpt_path = getenv("LANCZOS_HIP_CHECKPOINT");
    const int ckpt_every = env_int("LANCZOS_HIP_CHECKPOINT_EVERY", 0);
    const char *resume_path = getenv("LANCZOS_HIP_RESUME");

    const double t_create = now_sec();
    if (mat && (ngpu > 1 || loopback || force_slabs)) {
        if ((resume_path && *resume_path) || (ckpt_path && *ckpt_path && ckpt_every > 0) || want_vectors) {
            lz::fail("LANCZOS_HIP_NGPU: checkpoints and eigenvectors need the whole basis on one GPU");
            goto cleanup;
        }
        if (lanczos_hip_multi_create(&mg, std::max(ngpu, 1), nullptr, (int64_t)dimension, (int64_t)length,
                                     mat->row_head_indexes, mat->column_index, mat->values, max_iter, &opt,
                                     loopback ? LANCZOS_HIP_MULTI_LOOPBACK : 0) != 0)
            goto cleanup;
        e = reinterpret_cast<lz::Engine *>(lanczos_hip_multi_engine(mg, 0));
    } else if (mat) {
        if (lz::Engine::create(&e, (int64_t)dimension, (int64_t)length, mat->row_head_indexes, mat->column_index,
                               mat->values, max_iter, &opt) != 0)
            goto cleanup;
    } else {
        const std::vector<int64_t> no_rows(dimension + 1, 0);
        if (lz::Engine::create(&e, (int64_t)dimension, 0, no_rows.data(), nullptr, nullptr, max_iter, &opt) != 0)
            goto cleanup;
        if (e->set_operator(op, op_mat) != 0) goto cleanup;
    }
    // From here on this function touches e's stream itself (events, pinned history).  With several GPUs e
    // is slab 0's engine and the handle's calls leave the caller's device current, not slab 0's: HIP refuses
    // an event created or recorded under another device than its stream's.
    if (e->make_current() != 0) goto cleanup;
    if (e->enable_host_history() != 0) goto cleanup;
    // What the [TOTAL] lines report is what src/lanczos_cuda.c:672-688 puts between its events: the product, and the
    // recurrence with the reorthogonalisation; only those kernel classes are bracketed by events (the scaling pass
    // and the small reductions are not: two events a launch are not free on short kernels).  With several slabs the
    // lines describe slab 0, whose engine this is; the other slabs' kernels are not timed at all.
    {
        const unsigned timed = (1u << LANCZOS_HIP_K_SPMV) | (1u << LANCZOS_HIP_K_RECUR) | (1u << LANCZOS_HIP_K_DOTS) |
                               (1u << LANCZOS_HIP_K_UPDATE);
        e->prof_mask = timed;
        if (mg) {
            int slabs = 1;
            lanczos_hip_multi_info(mg, &slabs, NULL, NULL, NULL, NULL);
            for (int p = 1; p < slabs; ++p) reinterpret_cast<lz::Engine *>(lanczos_hip_multi_engine(mg, p))->prof_mask = 0;
        }
    }
    if (mg) {
        int halo = 0, over = 0;
        if (lanczos_hip_multi_set_start_seed(mg, seed) != 0) goto cleanup;
        lanczos_hip_multi_info(mg, NULL, &halo, &over, NULL, NULL);
        double slab_sec = 0.0;
        int slab_threads = 1;
        lanczos_hip_multi_build_stats(mg, &slab_sec, &slab_threads);
        fprintf(stderr, "[INIT] name=row_slabs count=%d transport=%s exchange=%s overlapped=%d sec=%.6f host_threads=%d\n",
                std::max(ngpu, 1), loopback ? "loopback" : "rccl", halo ? "halo" : "allgather", over, slab_sec,
                slab_threads);
    } else {
        std::vector<double> v0(dimension);
        lanczos_hip_gaussian_fill(seed, 0, (int64_t)dimension, v0.data());
        if (e->set_start(v0.data(), true, 1.0) != 0) goto cleanup;
    }
    if (resume_path && *resume_path) {
        // continue a run saved by LANCZOS_HIP_CHECKPOINT: the engine takes up where it was (bit for bit),
        // the host replays its checks on the restored alpha/beta without printing them again
        if (lanczos_hip_checkpoint_load(reinterpret_cast<lanczos_hip_engine *>(e), resume_path) != 0) goto cleanup;
        enq = replayed = std::min(e->k, total);
        printed = std::min(std::max(e->ckpt_host_k, 0), replayed);
        fprintf(stderr, "[INIT] name=resumed iterations=%d printed=%d\n", replayed, printed);
    }
    if (e->make_current() != 0) goto cleanup;
    for (auto &ev : done)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            lz::fail("hipEventCreate failed");
            goto cleanup;
        }
    time_init = now_sec() - t_start;
    fprintf(stderr, "[INIT] name=upload sec=%.6f bytes=%.0f GBps=%.2f pinned=%d detail_ref=%s\n", e->upload_sec, e->upload_bytes,
            e->upload_sec > 0 ? e->upload_bytes / e->upload_sec / 1e9 : 0.0, (int)e->upload_pinned, e->upload_pin_note);
    if (env_int("LANCZOS_HIP_INIT_TRACE", 0))  // where the set-up went: the engine's four stages, then the rest (start vector, events)
        fprintf(stderr, "[INIT] name=stages device=%.6f upload=%.6f spmv_form=%.6f state=%.6f before_create=%.6f\n", e->create_sec[0],
                e->create_sec[1], e->create_sec[2], e->create_sec[3], t_create - t_start);
    fprintf(stderr, "[INIT] name=total_wall sec=%.6f\n", time_init);
    fprintf(stderr, "[INIT] name=spmv_kernel detail_ref=%s\n", e->spmv_name);

    for (int k = 0; k < total; k++) {
        while (enq < total && enq < k + window) {
            if ((mg ? lanczos_hip_multi_step(mg) : e->enqueue_iteration()) != 0) goto cleanup;
            if (e->make_current() != 0 || e->record_event(done[(size_t)(enq % window)]) != 0) goto cleanup;
            enq++;
        }
        if (k >= replayed && hipEventSynchronize(done[(size_t)(k % window)]) != hipSuccess) {
            lz::fail("hipEventSynchronize failed");
            goto cleanup;
        }
        if (k >= replayed) e->harvest_completed();  // keeps the list of live profiling events short
        const double alpha = e->h_hist[2 * (size_t)k];
        const double beta = e->h_hist[2 * (size_t)k + 1];
        a[(size_t)k] = alpha;  // T(k,k), src/lanczos.c:113

        int eval_count = k + 1;  // src/lanczos.c:105-108
        if (eval_count > nth_eig) eval_count = nth_eig;
        for (int i = 0; i < nth_eig; i++) last[(size_t)i] = eigenvalues[i];  // src/lanczos.c:114-116
        const double td = now_sec();
        if (lanczos_hip_tridiag_eigs(a.data(), b.data(), k + 1, nth_eig, eigenvalues) != 0) {
            lz::fail("tridiagonal eigen-solve failed at step %d", k + 1);
            goto cleanup;
        }
        time_diag += now_sec() - td;

        bool all = true;  // src/lanczos.c:120-126
        for (int i = 0; i < eval_count; i++) {
            const double diff = eigenvalues[i] - last[(size_t)i];
            if (diff * diff > threshold * threshold) all = false;
        }
        if (all) {  // src/lanczos.c:127-133
            printf("converged ");
            for (int i = 0; i < nth_eig; i++) printf("%.1E ", last[(size_t)i] - eigenvalues[i]);
            printf("\n");
            status = EXIT_SUCCESS;
            reached_max = false;
            exit_kind = 1;
            steps_used = k + 1;
            break;
        }
        if (!quiet && k >= printed) {  // src/lanczos.c:134-138
            printf("%d\t", k + 1);
            for (int i = 0; i < nth_eig; i++) printf("%.7f\t", eigenvalues[i]);
            printf("\n");
        }
        if (beta * beta < threshold * threshold * threshold * threshold) {  // src/lanczos.c:185-188
            printf("%.7f beta converged\n", beta);
            status = EXIT_SUCCESS;
            reached_max = false;
            exit_kind = 2;
            steps_used = k + 1;
            break;
        }
        b[(size_t)k] = beta;  // T(k,k+1), src/lanczos.c:189-190
        if (ckpt_path && *ckpt_path && ckpt_every > 0 && k >= replayed && (k + 1) % ckpt_every == 0) {
            // written beside the target and renamed: a crash while writing leaves the previous file whole
            const std::string tmp = std::string(ckpt_path) + ".tmp";
            e->ckpt_host_k = k + 1;  // the engine is up to window - 1 iterations ahead of the lines printed so far
            if (lanczos_hip_checkpoint_save(reinterpret_cast<lanczos_hip_engine *>(e), tmp.c_str()) != 0 ||
                rename(tmp.c_str(), ckpt_path) != 0) {
                lz::fail("checkpoint to %s failed", ckpt_path);
                goto cleanup;
            }
        }
    }
    if (reached_max) {
        status = EXIT_SUCCESS;
        steps_used = total;
        printf("Reached max iterations (%d)\n", max_iter);  // src/lanczos.c:193
    }
    if (want_vectors && steps_used > 0) {
        // eigenvectors[i][0..n) = Ritz vector of eigenvalues[i], i < min(nth_eig, steps): one pass over
        // the resident basis per eight vectors
        const int m = std::min(nth_eig, steps_used);
        const size_t n = dimension;
        std::vector<double> y((size_t)m * n), ev((size_t)m);
        if (lanczos_hip_ritz_vectors(reinterpret_cast<lanczos_hip_engine *>(e), steps_used, m, ev.data(), y.data(),
                                     NULL) != 0) {
            printf("%s\n", lanczos_hip_last_error());
            status = EXIT_FAILURE;
        } else {
            for (int i = 0; i < m; i++)
                if (eigenvectors[i]) memcpy(eigenvectors[i], y.data() + (size_t)i * n, sizeof(double) * n);
        }
    }

cleanup:
    double t_spmv = 0.0, t_reorth = 0.0;
    int iterations = 0, reorth_steps = 0;
    char kernel_name[32] = "";
    if (e) {
        iterations = e->k;
        reorth_steps = e->reorth_steps;
        if (mg) lanczos_hip_multi_info(mg, NULL, NULL, NULL, &iterations, &reorth_steps);
        snprintf(kernel_name, sizeof kernel_name, "%s", e->spmv_name);
        (void)e->make_current();
        (void)hipStreamSynchronize(e->stream);
        e->harvest_profile();
        t_spmv = e->prof_ms[LANCZOS_HIP_K_SPMV] * 1e-3;
        t_reorth = (e->prof_ms[LANCZOS_HIP_K_RECUR] + e->prof_ms[LANCZOS_HIP_K_DOTS] + e->prof_ms[LANCZOS_HIP_K_UPDATE]) * 1e-3;
    }
    for (auto ev : done)
        if (ev) (void)hipEventDestroy(ev);
    if (mg)
        lanczos_hip_multi_destroy(mg);
    else
        delete e;
    // the grammar of src/lanczos_cuda.c:782-786 (no T upload exists here, so no cudaMemcpy line)
    fprintf(stderr, "[TOTAL] name=init detail_ref=INIT sec=%.6f\n", time_init);
    fprintf(stderr, "[TOTAL] name=spmv sec=%.6f\n", t_spmv);
    fprintf(stderr, "[TOTAL] name=diagonalization sec=%.6f\n", time_diag);
    fprintf(stderr, "[TOTAL] name=reorthogonalization sec=%.6f\n", t_reorth);
    const double wall = now_sec() - t_start;
    fprintf(stderr, "[TOTAL] name=wall sec=%.6f\n", wall);
    if (env_int("LANCZOS_HIP_JSON", 0))  // the same split, one machine-readable line
        fprintf(stderr,
                "{\"lanczos_hip\": {\"status\": %d, \"dimension\": %zu, \"nnz\": %zu, \"iterations\": %d, "
                "\"exit_kind\": %d, \"exit_step\": %d, \"reorth_steps\": %d, \"spmv_kernel\": \"%s\", \"sec\": {\"init\": %.6f, \"spmv\": %.6f, "
                "\"diagonalization\": %.6f, \"reorthogonalization\": %.6f, \"wall\": %.6f}}}\n",
                status, dimension, length, iterations, exit_kind, steps_used, reorth_steps, kernel_name, time_init, t_spmv,
                time_diag, t_reorth, wall);
    return status;
}

}  // namespace

extern "C" int lanczos_hip_crs(const Mat_Crs *mat, double eigenvalues[], double *eigenvectors[], int nth_eig,
                               int max_iter, double threshold) {
    return lz::guarded("lanczos_hip_crs", [&]() -> int {
    if (mat == NULL) return EXIT_FAILURE;  // src/lanczos_cuda.c:316-318
    return solve(mat, nullptr, nullptr, mat->dimension, eigenvalues, eigenvectors, nth_eig, max_iter, threshold);
    });
}

extern "C" int lanczos_hip_operator(lanczos_hip_matvec_fn *matvec, const void *mat, size_t dimension,
                                    double eigenvalues[], double *eigenvectors[], int nth_eig, int max_iter,
                                    double threshold) {
    return lz::guarded("lanczos_hip_operator", [&]() -> int {
    if (matvec == NULL) return EXIT_FAILURE;
    return solve(nullptr, matvec, mat, dimension, eigenvalues, eigenvectors, nth_eig, max_iter, threshold);
    });
}
