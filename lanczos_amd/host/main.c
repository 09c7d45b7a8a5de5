/*
 * Command-line driver with the reference's argv contract (src/main.c:14-82):
 *   <input_file> <coo|crs|crs_hip> <number_of_eigenvalues> <max_iter>
 * "crs_hip" sits where the reference has "crs_cuda" (src/main.c:34) and calls
 * lanczos_hip_crs(); "crs" and "coo" keep their lanczos(MAKE_MAT_MATVEC(..))
 * call sites (src/main.c:59, 73), which this build also runs on the GPU.
 * stdout/stderr lines keep the reference's grammar.
 */
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/lanczos_host.h"

#define buf_size 1000

static double wall(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
#define MEASURE(label, code)                                          \
    do {                                                              \
        double st__ = wall();                                         \
        code double en__ = wall();                                    \
        fprintf(stderr, "[TIME] %s: %lf sec\n", #label, en__ - st__); \
    } while (0)

typedef enum { COO, CRS, CRS_HIP } Mat_Type;

int main(int argc, const char *argv[]) {
    static double eigenvalues[buf_size];
    static double *eigenvectors[buf_size];
    Mat_Coo mat;
    Mat_Crs mat_crs;
    Mat_Type mat_type;
    printf("Using %d threads\n", omp_get_max_threads()); /* src/main.c:23 */
    if (argc != 5 || (strcmp(argv[2], "coo") && strcmp(argv[2], "crs") && strcmp(argv[2], "crs_hip"))) {
        fprintf(stderr, "Usage: %s <input_file> <coo|crs|crs_hip> <number_of_eigenvalues> <max_iter>\n", argv[0]);
        return EXIT_FAILURE;
    }
    mat_type = !strcmp(argv[2], "coo") ? COO : (!strcmp(argv[2], "crs") ? CRS : CRS_HIP);
    const int number_of_eigenvalues = atoi(argv[3]);
    const int max_iter = atoi(argv[4]);
    if (number_of_eigenvalues > buf_size) { /* the reference overflows here (src/main.c:12, 19) */
        fprintf(stderr, "number_of_eigenvalues must be <= %d\n", buf_size);
        return EXIT_FAILURE;
    }
    /* a file with the Matrix Market banner goes through that reader (the reference would exit on it,
     * src/coo.c:19-22); anything else is the reference's headerless triplet format */
    if (is_matrix_market(argv[1])) {
        int rc = 0;
        MEASURE(read_mat, rc = read_matrix_market(argv[1], &mat););
        if (rc != EXIT_SUCCESS) return EXIT_FAILURE;
    } else {
        MEASURE(read_mat, mat = read_mat_coo_incremental(argv[1]););
    }
    printf("mat dim: %ld\n", (long)mat.dimension);
    printf("mat num-of-non-zero: %ld\n", (long)mat.length);
    printf("sparsity: %f\n", ((double)mat.length / ((double)mat.dimension * (double)mat.dimension)));
    if (mat_type == CRS) {
        printf("[MODE] CRS selected\n");
        MEASURE(convert_from_coo, mat_crs = convert_from_coo(&mat, 1););
        free(mat.data);
        MEASURE(lanczos, lanczos(MAKE_MAT_MATVEC(&mat_crs), eigenvalues, eigenvectors, number_of_eigenvalues,
                                 max_iter, 10e-5););
    } else if (mat_type == CRS_HIP) {
        printf("[MODE] CRS_HIP selected\n");
        MEASURE(convert_from_coo, mat_crs = convert_from_coo(&mat, 1););
        free(mat.data);
        int rc = 0;
        MEASURE(lanczos_hip_crs, rc = lanczos_hip_crs(&mat_crs, eigenvalues, eigenvectors, number_of_eigenvalues,
                                                      max_iter, 10e-5););
        if (rc != EXIT_SUCCESS) return EXIT_FAILURE;
    } else {
        printf("[MODE] COO selected\n");
        MEASURE(lanczos, lanczos(MAKE_MAT_MATVEC(&mat), eigenvalues, eigenvectors, number_of_eigenvalues,
                                 max_iter, 10e-5););
    }
    for (int i = 0; i < number_of_eigenvalues; i++) printf("eigenvalue\t\t%d\t%.12f\n", i + 1, eigenvalues[i]);
    return EXIT_SUCCESS;
}
