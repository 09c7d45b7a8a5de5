/*
 * lanczos(): the reference's CPU entry point (src/lanczos.h:34-36) re-pointed
 * at the HIP engine.  Argument checks, messages and exit behaviour follow
 * src/lanczos.c:28-43; the iteration itself is lanczos_hip_crs(), or lanczos_hip_operator() for a
 * plugin that is neither of the reference's two.
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/lanczos_host.h"

void lanczos(const Mat_Matvec mat_matvec, double eigenvalues[], double *eigenvectors[], int nth_eig,
             int max_iter, double threshold) {
    if (mat_matvec.mat == NULL || mat_matvec.matvec == NULL || eigenvalues == NULL) {
        fprintf(stderr, "lanczos: NULL argument\n");
        exit(EXIT_FAILURE);
    }
    if (max_iter < 2 || nth_eig < 1) {
        fprintf(stderr, "lanczos: invalid parameters (nth_eig=%d, max_iter=%d)\n", nth_eig, max_iter);
        exit(EXIT_FAILURE);
    }
    if (mat_matvec.dimension == 0) {
        fprintf(stderr, "lanczos: mat_dim is 0\n");
        exit(EXIT_FAILURE);
    }
    if (mat_matvec.dimension > (size_t)INT_MAX) {
        fprintf(stderr, "lanczos: mat_dim=%zu exceeds INT_MAX (util routines use int)\n", mat_matvec.dimension);
        exit(EXIT_FAILURE);
    }
    int rc;
    if (mat_matvec.matvec == (Matvec_General *)matvec_crs) {
        rc = lanczos_hip_crs((const Mat_Crs *)mat_matvec.mat, eigenvalues, eigenvectors, nth_eig, max_iter,
                             threshold);
    } else if (mat_matvec.matvec == (Matvec_General *)matvec_coo) {
        /* matvec_coo applies the triplets as written (no exchange), so the
         * equivalent CRS operator is convert_from_coo(mat, 0) */
        Mat_Crs crs = convert_from_coo((const Mat_Coo *)mat_matvec.mat, 0);
        rc = lanczos_hip_crs(&crs, eigenvalues, eigenvectors, nth_eig, max_iter, threshold);
        lanczos_free_crs(&crs);
    } else {
        /* any other plugin (src/lanczos.h:5): the engine calls it on host memory once per iteration,
         * as src/lanczos.c:110 does, and keeps the vectors and their arithmetic on the GPU */
        rc = lanczos_hip_operator(mat_matvec.matvec, mat_matvec.mat, mat_matvec.dimension, eigenvalues, eigenvectors,
                                  nth_eig, max_iter, threshold);
    }
    if (rc != EXIT_SUCCESS) {
        fprintf(stderr, "lanczos: HIP engine failed: %s\n", lanczos_hip_last_error());
        exit(EXIT_FAILURE);
    }
}
