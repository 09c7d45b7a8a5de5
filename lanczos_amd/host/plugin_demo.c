/*
 * A host program that uses the reference's plugin contract (src/lanczos.h:5-36) with an operator
 * of its own: the 2-D 5-point Laplacian on an N x N grid applied matrix-free, handed to lanczos()
 * as a Mat_Matvec.  Nothing here knows about the GPU; liblanczos_host runs the iteration on it
 * and calls grid_matvec on host memory once per step, as src/lanczos.c:110 does.
 *   usage: plugin_demo <N> <number_of_eigenvalues> <max_iter>
 */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/lanczos_host.h"

typedef struct {
    int n;        /* grid points per side */
    long calls;   /* how often the solver asked for a product */
} Grid;

static void grid_matvec(const void *mat, const double *vec, double *dist) {
    Grid *g = (Grid *)mat;
    const int n = g->n;
    g->calls++;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            const size_t i = (size_t)y * n + x;
            double s = 4.0 * vec[i];
            if (x > 0) s -= vec[i - 1];
            if (x < n - 1) s -= vec[i + 1];
            if (y > 0) s -= vec[i - n];
            if (y < n - 1) s -= vec[i + n];
            dist[i] = s;
        }
}

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "Usage: %s <N> <number_of_eigenvalues> <max_iter>\n", argv[0]);
        return EXIT_FAILURE;
    }
    Grid g = {atoi(argv[1]), 0};
    const int nth = atoi(argv[2]), max_iter = atoi(argv[3]);
    if (g.n < 1 || nth < 1 || nth > 1000) return EXIT_FAILURE;
    static double eigenvalues[1000];
    Mat_Matvec op = {.mat = &g, .matvec = grid_matvec, .dimension = (size_t)g.n * (size_t)g.n};
    lanczos(op, eigenvalues, NULL, nth, max_iter, 0.0);
    printf("operator calls\t%ld\n", g.calls);
    for (int i = 0; i < nth; i++) printf("eigenvalue\t\t%d\t%.12f\n", i + 1, eigenvalues[i]);
    return EXIT_SUCCESS;
}
