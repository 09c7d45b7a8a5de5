/*
 * host_check -- walks the host-side C code (triplet and Matrix Market readers, the COO->CRS build, the
 * host matvecs, the synthetic generators, the tridiagonal and dense eigen-solvers, the orthogonality
 * tracker, the start-vector generator) over small inputs with awkward shapes.  Meant to be built with
 * -fsanitize=address,undefined (make -C lanczos_amd/csrc host_check): a clean exit with "host_check: ok"
 * means neither sanitizer saw anything; it also checks a few results against closed forms so that it is
 * worth running without them.  Uses no GPU and nothing of liblanczos_hip.so's device side.
 *
 *   host_check <scratch directory>
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/lanczos_host.h"

static int failures = 0;
#define CHECK(cond, ...)                         \
    do {                                         \
        if (!(cond)) {                           \
            fprintf(stderr, "host_check: ");     \
            fprintf(stderr, __VA_ARGS__);        \
            fprintf(stderr, "\n");               \
            failures += 1;                       \
        }                                        \
    } while (0)

static const double kPi = 3.14159265358979323846;

static void write_file(const char *path, const char *text) {
    FILE *f = fopen(path, "w");
    if (!f) {
        perror(path);
        exit(EXIT_FAILURE);
    }
    fputs(text, f);
    fclose(f);
}

/* the triplet reader, the build with and without the exchange of rows and columns, both host products */
static void check_triplets(const char *dir) {
    char path[4096];
    snprintf(path, sizeof path, "%s/triplets.txt", dir);
    /* an empty line, an empty row (row 4), a duplicate entry, no newline at the end */
    write_file(path, "1 1 2.0\n1 2 -1.0\n\n2 1 -1.0\n2 2 2.0\n2 3 -1.0\n3 2 -1.0\n3 3 2.0\n5 5 1.5\n5 1 0.25\n1 5 0.25\n1 5 0.5");
    Mat_Coo coo = read_mat_coo_incremental(path);
    CHECK(coo.dimension == 5 && coo.length == 11, "triplets: dimension %zu length %zu", coo.dimension, coo.length);
    for (int swap = 0; swap < 2; ++swap) {
        Mat_Crs crs = convert_from_coo(&coo, swap);
        CHECK(crs.dimension == 5 && crs.length == 11 && crs.row_head_indexes[5] == 11, "convert_from_coo(swap=%d) shape", swap);
        CHECK(crs.row_head_indexes[3] == crs.row_head_indexes[4], "row 4 should be empty");
        double x[5] = {1.0, -2.0, 3.0, 0.5, -0.75}, y1[5], y2[5];
        matvec_crs(&crs, x, y1);
        if (!swap) {   /* the matrix is symmetric up to the duplicate: the plain build must agree with the COO product */
            matvec_coo(&coo, x, y2);
            for (int i = 0; i < 5; ++i) CHECK(fabs(y1[i] - y2[i]) < 1e-15, "matvec_crs vs matvec_coo at %d: %g %g", i, y1[i], y2[i]);
        }
        lanczos_free_crs(&crs);
    }
    free(coo.data);
}

/* Malformed triplet files: the reader either takes them or ends the process with EXIT_FAILURE and a message, as the
 * reference's does (src/coo.c:19-40); each case runs in a child process so that the exit can be looked at.  A
 * sanitizer report ends the child with the code ASAN_OPTIONS / UBSAN_OPTIONS name (the test sets exitcode=99)
 * or a signal, neither of which is accepted. */
static void check_malformed_triplets(const char *dir) {
    char *longline = malloc(200001 + 16);
    memset(longline, '7', 200000);
    strcpy(longline + 200000, " 1 1.0\n");
    const char *cases[] = {
        "abc 1 2.0\n", "1 1\n", "1\n", "0 1 1.0\n", "1 0 1.0\n", "-3 2 1.0\n", "99999999999999999999999 1 1.0\n",
        "1 1 notanumber\n", "   \n \n", "1 1 1.0\r\n2 2 2.0\r\n", "1 1 1.0 trailing words\n", "\n\n\n", "", "1 1 1e999\n",
        "1\t1\t1.0\n", longline,
    };
    char path[4096];
    snprintf(path, sizeof path, "%s/malformed.txt", dir);
    for (size_t i = 0; i < sizeof cases / sizeof cases[0]; ++i) {
        write_file(path, cases[i]);
        fflush(NULL);
        const pid_t pid = fork();
        if (pid == 0) {
            if (!freopen("/dev/null", "w", stderr)) _exit(98);
            Mat_Coo coo = read_mat_coo_incremental(path);
            if (coo.length > 0) {
                Mat_Crs crs = convert_from_coo(&coo, 0);
                lanczos_free_crs(&crs);
            }
            free(coo.data);
            _exit(EXIT_SUCCESS);
        }
        int status = 0;
        CHECK(pid > 0 && waitpid(pid, &status, 0) == pid, "fork / waitpid");
        CHECK(WIFEXITED(status) && (WEXITSTATUS(status) == EXIT_SUCCESS || WEXITSTATUS(status) == EXIT_FAILURE),
              "malformed case %zu: the reader ended with status 0x%x", i, status);
    }
    free(longline);
}

static void check_matrix_market(const char *dir) {
    char path[4096];
    snprintf(path, sizeof path, "%s/sym.mtx", dir);
    write_file(path, "%%MatrixMarket matrix coordinate real symmetric\n% a comment\n\n3 3 4\n1 1 2.0\n2 1 -1.0\n2 2 2.0\n3 2 -1.0\n");
    CHECK(is_matrix_market(path) == 1, "banner not recognised");
    Mat_Coo coo;
    CHECK(read_matrix_market(path, &coo) == EXIT_SUCCESS, "symmetric file refused");
    CHECK(coo.dimension == 3 && coo.length == 6, "symmetric: dimension %zu length %zu", coo.dimension, coo.length);
    free(coo.data);
    snprintf(path, sizeof path, "%s/pat.mtx", dir);
    write_file(path, "%%MatrixMarket matrix coordinate pattern general\n2 2 3\n1 1\n1 2\n2 2\n");
    CHECK(read_matrix_market(path, &coo) == EXIT_SUCCESS && coo.length == 3, "pattern file");
    free(coo.data);
    snprintf(path, sizeof path, "%s/skew.mtx", dir);
    write_file(path, "%%MatrixMarket matrix coordinate integer skew-symmetric\n3 3 2\n2 1 4\n3 1 -2\n");
    CHECK(read_matrix_market(path, &coo) == EXIT_SUCCESS && coo.length == 4, "skew-symmetric file");
    free(coo.data);
    /* fewer entries than the size line promises, an index beyond the size, a truncated banner: refused, nothing leaked */
    snprintf(path, sizeof path, "%s/short.mtx", dir);
    write_file(path, "%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n");
    CHECK(read_matrix_market(path, &coo) != EXIT_SUCCESS, "short file accepted");
    snprintf(path, sizeof path, "%s/range.mtx", dir);
    write_file(path, "%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n");
    CHECK(read_matrix_market(path, &coo) != EXIT_SUCCESS, "index beyond the size accepted");
    snprintf(path, sizeof path, "%s/banner.mtx", dir);
    write_file(path, "%%MatrixMarket matrix");
    CHECK(read_matrix_market(path, &coo) != EXIT_SUCCESS, "truncated banner accepted");
    snprintf(path, sizeof path, "%s/triplets.txt", dir);
    CHECK(is_matrix_market(path) == 0, "a triplet file taken for Matrix Market");
}

/* generators: whole matrices against their row slabs, row sums of the Laplacians, symmetry of the random matrix */
static void check_generators(void) {
    for (int d = 1; d <= 3; ++d) {
        const int64_t N = d == 1 ? 37 : d == 2 ? 9 : 5;
        Mat_Crs a, s;
        CHECK(lanczos_synth_laplacian(d, N, &a) == 0, "laplacian d=%d", d);
        const int64_t n = (int64_t)a.dimension;
        const int64_t r0 = n / 3, r1 = n - n / 4;
        CHECK(lanczos_synth_laplacian_rows(d, N, r0, r1, &s) == 0, "laplacian rows d=%d", d);
        CHECK((int64_t)s.dimension == r1 - r0, "slab rows");
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t la = a.row_head_indexes[r + 1] - a.row_head_indexes[r];
            const int64_t ls = s.row_head_indexes[r - r0 + 1] - s.row_head_indexes[r - r0];
            CHECK(la == ls, "d=%d row %lld: %lld entries in the whole matrix, %lld in the slab", d, (long long)r, (long long)la, (long long)ls);
            for (int64_t j = 0; j < la && j < ls; ++j) {
                const int64_t ea = a.row_head_indexes[r] + j, es = s.row_head_indexes[r - r0] + j;
                CHECK(a.column_index[ea] == s.column_index[es] && a.values[ea] == s.values[es], "d=%d row %lld entry %lld", d, (long long)r, (long long)j);
            }
        }
        lanczos_free_crs(&s);
        lanczos_free_crs(&a);
    }
    Mat_Crs r, rs;
    CHECK(lanczos_synth_random_symmetric(500, 6, 7, &r) == 0, "random symmetric");
    /* symmetric: the entry (i, j) has its mirror with the same value */
    for (int64_t i = 0; i < 500; ++i)
        for (int64_t e = r.row_head_indexes[i]; e < r.row_head_indexes[i + 1]; ++e) {
            const int64_t j = r.column_index[e];
            int found = 0;
            for (int64_t f = r.row_head_indexes[j]; f < r.row_head_indexes[j + 1]; ++f)
                if (r.column_index[f] == i && r.values[f] == r.values[e]) found = 1;
            CHECK(found, "random symmetric: (%lld, %lld) has no mirror", (long long)i, (long long)j);
        }
    CHECK(lanczos_synth_random_symmetric_rows(500, 6, 7, 123, 377, &rs) == 0, "random symmetric rows");
    for (int64_t i = 123; i < 377; ++i) {
        const int64_t la = r.row_head_indexes[i + 1] - r.row_head_indexes[i];
        const int64_t ls = rs.row_head_indexes[i - 123 + 1] - rs.row_head_indexes[i - 123];
        CHECK(la == ls, "random rows: row %lld", (long long)i);
        for (int64_t j = 0; j < la && j < ls; ++j)
            CHECK(r.column_index[r.row_head_indexes[i] + j] == rs.column_index[rs.row_head_indexes[i - 123] + j] &&
                      r.values[r.row_head_indexes[i] + j] == rs.values[rs.row_head_indexes[i - 123] + j],
                  "random rows: row %lld entry %lld", (long long)i, (long long)j);
    }
    lanczos_free_crs(&rs);
    lanczos_free_crs(&r);
}

/* eigen-solvers against the closed form of the 1-D Laplacian, 2 - 2 cos(k pi / (n + 1)) */
static void check_eigensolvers(void) {
    enum { n = 61, m = 5 };
    double alpha[n], beta[n], eig[m], vec[m * n];
    for (int i = 0; i < n; ++i) {
        alpha[i] = 2.0;
        beta[i] = -1.0;
    }
    CHECK(lanczos_hip_tridiag_eigs(alpha, beta, n, m, eig) == 0, "tridiag_eigs");
    for (int k = 0; k < m; ++k) {
        const double want = 2.0 - 2.0 * cos((k + 1) * kPi / (n + 1));
        CHECK(fabs(eig[k] - want) < 1e-13, "tridiag eigenvalue %d: %.17g, closed form %.17g", k, eig[k], want);
    }
    CHECK(lanczos_hip_tridiag_eigvecs(alpha, beta, n, m, eig, vec) == 0, "tridiag_eigvecs");
    for (int k = 0; k < m; ++k) {   /* |T v - lambda v| */
        double worst = 0.0;
        for (int i = 0; i < n; ++i) {
            double t = alpha[i] * vec[k * n + i] - eig[k] * vec[k * n + i];
            if (i > 0) t += beta[i - 1] * vec[k * n + i - 1];
            if (i + 1 < n) t += beta[i] * vec[k * n + i + 1];
            worst = fmax(worst, fabs(t));
        }
        CHECK(worst < 1e-12, "tridiag eigenvector %d: residual %g", k, worst);
    }
    /* sizes 1 and 2, and more values asked for than there are */
    double one = 3.5, e1[2];
    CHECK(lanczos_hip_tridiag_eigs(&one, beta, 1, 1, e1) == 0 && fabs(e1[0] - 3.5) < 1e-14, "1 x 1 tridiagonal: %.17g", e1[0]);
    CHECK(lanczos_hip_tridiag_eigs(alpha, beta, 2, 2, e1) == 0 && fabs(e1[0] - 1.0) < 1e-15 && fabs(e1[1] - 3.0) < 1e-15, "2 x 2 tridiagonal");
    /* dense symmetric: the same matrix, full storage */
    enum { d = 24 };
    double *a = calloc((size_t)d * d, sizeof(double)), *a2 = calloc((size_t)d * d, sizeof(double));
    double ev[d], *evec = calloc((size_t)d * d, sizeof(double));
    for (int i = 0; i < d; ++i) {
        a[i * d + i] = a2[i * d + i] = 2.0;
        if (i + 1 < d) a[i * d + i + 1] = a[(i + 1) * d + i] = a2[i * d + i + 1] = a2[(i + 1) * d + i] = -1.0;
    }
    CHECK(lanczos_hip_sym_lowest_eigs(a, d, 4, ev) == 0, "sym_lowest_eigs");
    for (int k = 0; k < 4; ++k)
        CHECK(fabs(ev[k] - (2.0 - 2.0 * cos((k + 1) * kPi / (d + 1)))) < 1e-13, "dense eigenvalue %d: %.17g", k, ev[k]);
    CHECK(lanczos_hip_sym_eig_jacobi(a2, d, ev, evec) == 0, "sym_eig_jacobi");
    double lowest = ev[0];
    for (int k = 1; k < d; ++k) lowest = fmin(lowest, ev[k]);
    CHECK(fabs(lowest - (2.0 - 2.0 * cos(kPi / (d + 1)))) < 1e-13, "jacobi lowest eigenvalue %.17g", lowest);
    free(a);
    free(a2);
    free(evec);
}

/* the orthogonality tracker: a run to its last step, state saved and loaded, a longer tracker fitted from a shorter one */
static void check_tracker(void) {
    enum { steps = 40 };
    lanczos_hip_omega *o = lanczos_hip_omega_create(steps);
    CHECK(o != NULL, "omega_create");
    double est = 0.0;
    int asked = 0;
    for (int j = 0; j < steps; ++j) {
        if (j > 0) lanczos_hip_omega_set_beta(o, j - 1, 1.0 + 0.01 * j);
        const int need = lanczos_hip_omega_step(o, j, 2.0 + 0.1 * sin(j), 1.0 + 0.01 * (j + 1), &est);
        CHECK(need == 0 || need == 1, "omega_step returned %d", need);
        asked += need;
    }
    CHECK(lanczos_hip_omega_reorths(o) == asked, "omega_reorths %d, counted %d", lanczos_hip_omega_reorths(o), asked);
    const size_t len = lanczos_hip_omega_state_doubles(o);
    double *state = calloc(len, sizeof(double));
    lanczos_hip_omega_state_save(o, state);
    lanczos_hip_omega *p = lanczos_hip_omega_create(steps);
    CHECK(lanczos_hip_omega_state_load(p, state) == 0, "omega_state_load");
    CHECK(lanczos_hip_omega_reorths(p) == asked, "loaded tracker counts %d", lanczos_hip_omega_reorths(p));
    lanczos_hip_omega_reset(p);
    CHECK(lanczos_hip_omega_reorths(p) == 0, "reset tracker counts %d", lanczos_hip_omega_reorths(p));
    lanczos_hip_omega_destroy(p);
    /* the same state laid out for a tracker with room for more steps, which then goes on */
    lanczos_hip_omega *q = lanczos_hip_omega_create(2 * steps);
    double *wide = calloc(lanczos_hip_omega_state_doubles(q), sizeof(double));
    CHECK(lanczos_hip_omega_state_fit(q, state, len, wide) == 0, "omega_state_fit");
    CHECK(lanczos_hip_omega_state_load(q, wide) == 0, "omega_state_load of the fitted state");
    lanczos_hip_omega_set_beta(q, steps - 1, 1.0 + 0.01 * steps);
    const int more = lanczos_hip_omega_step(q, steps, 2.0, 1.4, &est);
    CHECK(more == 0 || more == 1, "fitted tracker: step %d returned %d", steps, more);
    CHECK(lanczos_hip_omega_step(q, steps + 5, 2.0, 1.4, &est) == -1, "a call out of order was accepted");
    lanczos_hip_omega_destroy(q);
    free(wide);
    lanczos_hip_omega_destroy(o);
    lanczos_hip_omega_destroy(NULL);
    free(state);
}

/* the start vector is a function of (seed, global row): any cut gives the same numbers */
static void check_start_vector(void) {
    enum { n = 1000 };
    double whole[n], parts[n];
    lanczos_hip_gaussian_fill(99, 0, n, whole);
    lanczos_hip_gaussian_fill(99, 0, 1, parts);
    lanczos_hip_gaussian_fill(99, 1, 332, parts + 1);
    lanczos_hip_gaussian_fill(99, 333, n - 333, parts + 333);
    lanczos_hip_gaussian_fill(99, 17, 0, parts);   /* nothing asked for: nothing written */
    CHECK(memcmp(whole, parts, sizeof whole) == 0, "gaussian_fill depends on the cut");
    double mean = 0.0, var = 0.0;
    for (int i = 0; i < n; ++i) mean += whole[i] / n;
    for (int i = 0; i < n; ++i) var += (whole[i] - mean) * (whole[i] - mean) / n;
    CHECK(fabs(mean) < 0.15 && fabs(var - 1.0) < 0.2, "gaussian_fill: mean %g variance %g", mean, var);
}

int main(int argc, char **argv) {
    if (argc != 2) {
        fprintf(stderr, "usage: host_check <scratch directory>\n");
        return EXIT_FAILURE;
    }
    check_triplets(argv[1]);
    check_malformed_triplets(argv[1]);
    check_matrix_market(argv[1]);
    check_generators();
    check_eigensolvers();
    check_tracker();
    check_start_vector();
    if (failures) {
        fprintf(stderr, "host_check: %d check(s) failed\n", failures);
        return EXIT_FAILURE;
    }
    printf("host_check: ok\n");
    return EXIT_SUCCESS;
}
