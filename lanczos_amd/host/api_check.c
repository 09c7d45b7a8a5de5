/*
 * api_check -- walks the C ABI of include/lanczos_hip.h beyond what the command-line driver reaches: the engine
 * handle (every SpMV form on matrices that suit it, Ritz vectors, thick restart, checkpoints, the device build
 * from triplets), the block variant (also with the rows of a power-law graph regrouped by column class), several
 * slabs on one GPU.  Built with the address sanitizer on the host code (make -C lanczos_amd/csrc asan) it is
 * the sanitizer's view of those set-up paths; results are checked against each other so that it is worth
 * running without it.  Needs a GPU.
 *
 *   api_check <scratch directory>
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lanczos_host.h"

static int failures = 0;
#define CHECK(cond, ...)                                                   \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "api_check: ");                                \
            fprintf(stderr, __VA_ARGS__);                                  \
            fprintf(stderr, " [last error: %s]\n", lanczos_hip_last_error()); \
            failures += 1;                                                 \
        }                                                                  \
    } while (0)

static lanczos_hip_options options(int reorth, int spmv) {
    lanczos_hip_options o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.device = -1;
    o.reorth = reorth;
    o.spmv_kernel = spmv;
    o.profile = 1;
    return o;
}

/* a symmetric matrix with a few very long rows: vertex 0..hubs-1 joined to `deg` others each, a ring, a heavy diagonal */
static Mat_Crs power_law(int64_t n, int hubs, int64_t deg) {
    const size_t cap = (size_t)(2 * hubs * deg + 3 * n);
    Coo *t = malloc(cap * sizeof(Coo));
    size_t m = 0;
    uint64_t s = 12345;
    for (int h = 0; h < hubs; ++h)
        for (int64_t j = 0; j < deg; ++j) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const int64_t c = hubs + (int64_t)((s >> 33) % (uint64_t)(n - hubs));
            t[m++] = (Coo){.index_row = (int)h, .index_column = (int)c, .value = -0.001};
            t[m++] = (Coo){.index_row = (int)c, .index_column = (int)h, .value = -0.001};
        }
    for (int64_t i = 0; i < n; ++i) {
        t[m++] = (Coo){.index_row = (int)i, .index_column = (int)i, .value = 4.0 + 1e-3 * (double)(i % 97)};
        t[m++] = (Coo){.index_row = (int)i, .index_column = (int)((i + 1) % n), .value = -1.0};
        t[m++] = (Coo){.index_row = (int)((i + 1) % n), .index_column = (int)i, .value = -1.0};
    }
    Mat_Coo coo = {.data = t, .length = m, .dimension = (size_t)n};
    Mat_Crs crs = convert_from_coo(&coo, 0);
    free(t);
    return crs;
}

static double rel(double a, double b) { return fabs(a - b) / fmax(fabs(b), 1e-300); }

/* every SpMV form the matrix admits: the same alpha/beta to 1e-11, the same product to 1e-12 */
static void check_forms(const Mat_Crs *a, const char *what, const int *forms, int nforms, int steps) {
    const int64_t n = (int64_t)a->dimension;
    double *ref_a = NULL, *ref_b = NULL, *ref_y = NULL;
    double *x = malloc(sizeof(double) * (size_t)n);
    lanczos_hip_gaussian_fill(5, 0, n, x);
    for (int f = 0; f < nforms; ++f) {
        lanczos_hip_options o = options(LANCZOS_HIP_REORTH_FULL, forms[f]);
        lanczos_hip_engine *e = NULL;
        if (lanczos_hip_create(&e, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, steps + 1, &o) != 0) {
            /* a form may refuse a matrix (the row-wise form wants short 64-row groups): said, not counted */
            fprintf(stderr, "api_check: %s: form %d refused: %s\n", what, forms[f], lanczos_hip_last_error());
            continue;
        }
        double *al = calloc((size_t)steps, sizeof(double)), *be = calloc((size_t)steps, sizeof(double));
        double *y = calloc((size_t)n, sizeof(double));
        CHECK(lanczos_hip_set_start_seed(e, 7) == 0, "%s form %d: set_start_seed", what, forms[f]);
        CHECK(lanczos_hip_run(e, steps, al, be) == 0, "%s form %d: run", what, forms[f]);
        CHECK(lanczos_hip_iterations(e) == steps, "%s form %d: iterations", what, forms[f]);
        CHECK(lanczos_hip_spmv(e, x, y) == 0, "%s form %d: spmv", what, forms[f]);
        if (!ref_a) {
            ref_a = al;
            ref_b = be;
            ref_y = y;
            printf("api_check: %s: reference form %s\n", what, lanczos_hip_spmv_name(e));
        } else {
            double worst = 0.0, worst_y = 0.0, scale = 0.0;
            for (int k = 0; k < steps; ++k) worst = fmax(worst, fmax(rel(al[k], ref_a[k]), rel(be[k], ref_b[k])));
            for (int64_t i = 0; i < n; ++i) {
                worst_y = fmax(worst_y, fabs(y[i] - ref_y[i]));
                scale = fmax(scale, fabs(ref_y[i]));
            }
            CHECK(worst < 1e-9, "%s form %s: alpha/beta differ by %g", what, lanczos_hip_spmv_name(e), worst);
            CHECK(worst_y <= 1e-12 * scale, "%s form %s: product differs by %g of %g", what, lanczos_hip_spmv_name(e), worst_y, scale);
            printf("api_check: %s: form %s agrees (alpha/beta %.1e, product %.1e)\n", what, lanczos_hip_spmv_name(e), worst, worst_y / fmax(scale, 1e-300));
            free(al);
            free(be);
            free(y);
        }
        lanczos_hip_destroy(e);
    }
    free(ref_a);
    free(ref_b);
    free(ref_y);
    free(x);
}

static void check_handle_extras(const Mat_Crs *a, const char *dir) {
    const int64_t n = (int64_t)a->dimension;
    const int steps = 40, m = 3;
    lanczos_hip_options o = options(LANCZOS_HIP_REORTH_FULL, LANCZOS_HIP_SPMV_AUTO);
    lanczos_hip_engine *e = NULL, *e2 = NULL;
    double al[64], be[64], al2[64], be2[64], eig[3], res[3];
    double *vecs = malloc(sizeof(double) * (size_t)(m * n));
    char path[4096];
    snprintf(path, sizeof path, "%s/api_check.ckpt", dir);
    CHECK(lanczos_hip_create(&e, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, steps + 1, &o) == 0, "create");
    CHECK(lanczos_hip_set_start_seed(e, 11) == 0 && lanczos_hip_run(e, 25, al, be) == 0, "first 25 steps");
    CHECK(lanczos_hip_checkpoint_save(e, path) == 0, "checkpoint_save");
    CHECK(lanczos_hip_run(e, 15, al + 25, be + 25) == 0, "15 more steps");
    CHECK(lanczos_hip_ritz_vectors(e, steps, m, eig, vecs, res) == 0, "ritz_vectors");
    for (int i = 0; i < m; ++i) {   /* a unit vector whose Rayleigh quotient is the Ritz value */
        double nn = 0.0;
        for (int64_t r = 0; r < n; ++r) nn += vecs[i * n + r] * vecs[i * n + r];
        CHECK(fabs(nn - 1.0) < 1e-10, "Ritz vector %d has norm^2 %g", i, nn);
    }
    CHECK(lanczos_hip_create(&e2, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, steps + 1, &o) == 0, "second engine");
    CHECK(lanczos_hip_checkpoint_load(e2, path) == 0, "checkpoint_load");
    CHECK(lanczos_hip_run(e2, 15, al2 + 25, be2 + 25) == 0, "resumed steps");
    CHECK(memcmp(al + 25, al2 + 25, 15 * sizeof(double)) == 0 && memcmp(be + 25, be2 + 25, 15 * sizeof(double)) == 0,
          "the resumed run differs from the uninterrupted one");
    lanczos_hip_destroy(e2);
    /* thick restart: the lowest value again, from a basis of 24 vectors */
    double teig[2], tres[2];
    int cycles = 0, products = 0;
    lanczos_hip_destroy(e);
    CHECK(lanczos_hip_create(&e, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, 24, &o) == 0, "restart engine");
    CHECK(lanczos_hip_set_start_seed(e, 11) == 0, "set_start_seed");
    CHECK(lanczos_hip_thick_restart(e, 2, 6, 40, 1e-9, teig, tres, NULL, &cycles, &products) == 0, "thick_restart");
    printf("api_check: thick restart: %d cycles, %d products, lowest %.12g (residual bound %.1e)\n", cycles, products, teig[0], tres[0]);
    lanczos_hip_destroy(e);
    /* the device build from triplets against the host build */
    Coo *t = malloc(sizeof(Coo) * a->length);
    size_t q = 0;
    for (size_t r = 0; r < a->dimension; ++r)
        for (int64_t j = a->row_head_indexes[r]; j < a->row_head_indexes[r + 1]; ++j)
            t[q++] = (Coo){.index_row = (int)r, .index_column = (int)a->column_index[j], .value = a->values[j]};
    Mat_Coo coo = {.data = t, .length = a->length, .dimension = a->dimension};
    CHECK(lanczos_hip_create_from_coo(&e, &coo, 0, 8, &o) == 0, "create_from_coo");
    int64_t *rp = malloc(sizeof(int64_t) * (size_t)(n + 1)), *ci = malloc(sizeof(int64_t) * a->length);
    double *va = malloc(sizeof(double) * a->length);
    CHECK(lanczos_hip_download_crs(e, rp, ci, va) == 0, "download_crs");
    CHECK(memcmp(rp, a->row_head_indexes, sizeof(int64_t) * (size_t)(n + 1)) == 0 && memcmp(ci, a->column_index, sizeof(int64_t) * a->length) == 0 &&
              memcmp(va, a->values, sizeof(double) * a->length) == 0,
          "the device build differs from the host build");
    lanczos_hip_destroy(e);
    free(rp);
    free(ci);
    free(va);
    free(t);
    free(vecs);
}

static void check_block(const Mat_Crs *a, const char *what, int steps) {
    const int64_t n = (int64_t)a->dimension;
    lanczos_hip_options o = options(LANCZOS_HIP_REORTH_FULL, LANCZOS_HIP_SPMV_AUTO);
    lanczos_hip_block *b = NULL;
    double eig[4];
    CHECK(lanczos_hip_block_create(&b, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, steps, &o) == 0, "%s: block_create", what);
    if (!b) return;
    CHECK(lanczos_hip_block_set_start_seed(b, 3) == 0, "%s: block_set_start_seed", what);
    CHECK(lanczos_hip_block_run(b, steps) == 0 && lanczos_hip_block_steps(b) == steps, "%s: block_run", what);
    CHECK(lanczos_hip_block_ritz(b, steps, 4, eig) == 0, "%s: block_ritz", what);
    double *ta = malloc(sizeof(double) * 256 * (size_t)steps), *tb = malloc(sizeof(double) * 256 * (size_t)steps);
    CHECK(lanczos_hip_block_tmatrix(b, steps, ta, tb) == 0, "%s: block_tmatrix", what);
    /* A_0 is symmetric, B_0 upper triangular */
    double asym = 0.0, low = 0.0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < i; ++j) {
            asym = fmax(asym, fabs(ta[i * 16 + j] - ta[j * 16 + i]));
            low = fmax(low, fabs(tb[i * 16 + j]));
        }
    CHECK(asym < 1e-12 && low == 0.0, "%s: A_0 asymmetry %g, B_0 below the diagonal %g", what, asym, low);
    /* the product of slot 0 through the test hook, against the host product column by column */
    double *q0 = malloc(sizeof(double) * 16 * (size_t)n), *q1 = malloc(sizeof(double) * 16 * (size_t)n);
    double *x = malloc(sizeof(double) * (size_t)n), *y = malloc(sizeof(double) * (size_t)n);
    CHECK(lanczos_hip_block_get_block(b, 0, q0) == 0, "%s: get_block", what);
    CHECK(lanczos_hip_block_spmm(b, 0, steps) == 0 && lanczos_hip_block_get_block(b, steps, q1) == 0, "%s: block_spmm", what);
    double worst = 0.0, scale = 0.0;
    for (int c = 0; c < 16; c += 5) {
        for (int64_t r = 0; r < n; ++r) x[r] = q0[r * 16 + c];
        matvec_crs(a, x, y);
        for (int64_t r = 0; r < n; ++r) {
            worst = fmax(worst, fabs(y[r] - q1[r * 16 + c]));
            scale = fmax(scale, fabs(y[r]));
        }
    }
    CHECK(worst <= 1e-12 * scale, "%s: block product differs from the host product by %g of %g", what, worst, scale);
    printf("api_check: %s: block variant, lowest Ritz value %.12g after %d block steps; product error %.1e\n", what, eig[0], steps, worst / fmax(scale, 1e-300));
    /* the same run with the rows cut into three slabs on this GPU */
    lanczos_hip_block_multi *bm = NULL;
    double eig3[4];
    CHECK(lanczos_hip_block_multi_create(&bm, 3, NULL, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, steps, &o,
                                         LANCZOS_HIP_MULTI_LOOPBACK) == 0, "%s: block_multi_create", what);
    if (bm) {
        CHECK(lanczos_hip_block_multi_set_start_seed(bm, 3) == 0 && lanczos_hip_block_multi_run(bm, steps) == 0, "%s: block_multi run", what);
        CHECK(lanczos_hip_block_multi_ritz(bm, steps, 4, eig3) == 0, "%s: block_multi_ritz", what);
        double w3 = 0.0;
        for (int i = 0; i < 4; ++i) w3 = fmax(w3, rel(eig3[i], eig[i]));
        CHECK(w3 < 1e-10, "%s: three slabs differ from one engine by %g", what, w3);
        printf("api_check: %s: block variant on 3 slabs within %.1e of one engine\n", what, w3);
        lanczos_hip_block_multi_destroy(bm);
    }
    free(q0);
    free(q1);
    free(x);
    free(y);
    free(ta);
    free(tb);
    lanczos_hip_block_destroy(b);
}

static void check_slabs(const Mat_Crs *a, const char *what, int reorth) {
    const int64_t n = (int64_t)a->dimension;
    const int steps = 30;
    double a1[32], b1[32], aP[32], bP[32];
    lanczos_hip_options o = options(reorth, LANCZOS_HIP_SPMV_AUTO);
    lanczos_hip_engine *e = NULL;
    double *v0 = malloc(sizeof(double) * (size_t)n);
    lanczos_hip_gaussian_fill(21, 0, n, v0);
    CHECK(lanczos_hip_create(&e, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, steps + 1, &o) == 0, "%s: create", what);
    CHECK(lanczos_hip_set_start(e, v0) == 0 && lanczos_hip_run(e, steps, a1, b1) == 0, "%s: single run", what);
    lanczos_hip_destroy(e);
    for (int P = 2; P <= 5; P += 3) {
        lanczos_hip_multi *m = NULL;
        CHECK(lanczos_hip_multi_create(&m, P, NULL, n, (int64_t)a->length, a->row_head_indexes, a->column_index, a->values, steps + 1, &o,
                                       LANCZOS_HIP_MULTI_LOOPBACK) == 0, "%s: multi_create(%d)", what, P);
        if (!m) continue;
        int slabs = 0, halo = 0, over = 0, threads = 0;
        double sec = 0.0;
        CHECK(lanczos_hip_multi_set_start(m, v0) == 0 && lanczos_hip_multi_run(m, steps, aP, bP) == 0, "%s: %d slabs: run", what, P);
        lanczos_hip_multi_info(m, &slabs, &halo, &over, NULL, NULL);
        lanczos_hip_multi_build_stats(m, &sec, &threads);
        double worst = 0.0;
        for (int k = 0; k < steps; ++k) worst = fmax(worst, fmax(rel(aP[k], a1[k]), rel(bP[k], b1[k])));
        CHECK(worst < 1e-9, "%s: %d slabs differ from one engine by %g", what, P, worst);
        printf("api_check: %s: %d slabs (%s exchange%s, %d build threads): alpha/beta within %.1e of one engine\n", what, slabs,
               halo ? "halo" : "all-gather", over ? ", overlapped" : "", threads, worst);
        lanczos_hip_multi_destroy(m);
    }
    free(v0);
}

int main(int argc, char **argv) {
    if (argc != 2) {
        fprintf(stderr, "usage: api_check <scratch directory>\n");
        return EXIT_FAILURE;
    }
    Mat_Crs lap3, lap2, rnd;
    CHECK(lanczos_synth_laplacian(3, 24, &lap3) == 0 && lanczos_synth_laplacian(2, 300, &lap2) == 0, "synthetic Laplacians");
    CHECK(lanczos_synth_random_symmetric(40000, 9, 17, &rnd) == 0, "random symmetric");
    Mat_Crs pl = power_law(30000, 6, 5000);
    const int regular[] = {LANCZOS_HIP_SPMV_DICT, LANCZOS_HIP_SPMV_ROWS, LANCZOS_HIP_SPMV_PIPE, LANCZOS_HIP_SPMV_PIPE_NT, LANCZOS_HIP_SPMV_STREAM,
                           LANCZOS_HIP_SPMV_VECTOR, LANCZOS_HIP_SPMV_MERGE, LANCZOS_HIP_SPMV_PANEL};
    const int irregular[] = {LANCZOS_HIP_SPMV_MERGE, LANCZOS_HIP_SPMV_PANEL, LANCZOS_HIP_SPMV_PIPE, LANCZOS_HIP_SPMV_VECTOR, LANCZOS_HIP_SPMV_STREAM};
    check_forms(&lap3, "laplacian 24^3", regular, 8, 30);
    check_forms(&lap2, "laplacian 300^2", regular, 8, 30);
    check_forms(&rnd, "random 40000 x 19", irregular, 5, 30);
    check_forms(&pl, "power law 30000", irregular, 5, 30);
    check_handle_extras(&lap3, argv[1]);
    check_block(&lap3, "laplacian 24^3", 5);
    check_block(&pl, "power law 30000", 4);
    check_slabs(&lap3, "laplacian 24^3", LANCZOS_HIP_REORTH_FULL);
    check_slabs(&lap3, "laplacian 24^3, no reorthogonalisation", LANCZOS_HIP_REORTH_NONE);
    check_slabs(&rnd, "random 40000 x 19", LANCZOS_HIP_REORTH_FULL);
    lanczos_free_crs(&lap3);
    lanczos_free_crs(&lap2);
    lanczos_free_crs(&rnd);
    lanczos_free_crs(&pl);
    if (failures) {
        fprintf(stderr, "api_check: %d check(s) failed\n", failures);
        return EXIT_FAILURE;
    }
    printf("api_check: ok\n");
    return EXIT_SUCCESS;
}
