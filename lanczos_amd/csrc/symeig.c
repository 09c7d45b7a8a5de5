/* Lowest eigenvalues of a small dense symmetric matrix: Householder reduction
 * to tridiagonal form, then the bisection of lanczos_hip_tridiag_eigs.  Used
 * for the block-tridiagonal T of the block variant (order 16 x steps), where
 * the reference's single-vector solver would call LAPACKE_dsyevr
 * (src/util.c:118-272) on its tridiagonal T. */
#include <math.h>
#include <stdlib.h>

#include "../../include/lanczos_hip.h"

/* a: n x n, row-major, symmetric; destroyed.  eig: the lowest nth values, ascending. */
int lanczos_hip_sym_lowest_eigs(double *a, int n, int nth, double *eig) {
    if (!a || !eig || n < 1 || nth < 1 || nth > n) return 1;
    double *d = (double *)malloc(sizeof(double) * (size_t)n);
    double *e = (double *)malloc(sizeof(double) * (size_t)n);
    if (!d || !e) {
        free(d);
        free(e);
        return 1;
    }
#define A(i, j) a[(size_t)(i) * (size_t)n + (size_t)(j)]
    /* rows n-1 .. 1: annihilate A(i, 0..i-2) with a reflector built from row i's leading part;
     * only the lower triangle is referenced and updated */
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += fabs(A(i, k));
            if (scale == 0.0) {
                e[i] = A(i, l);
            } else {
                for (int k = 0; k <= l; ++k) {
                    A(i, k) /= scale;
                    h += A(i, k) * A(i, k);
                }
                double f = A(i, l);
                double g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                A(i, l) = f - g;
                f = 0.0;
#pragma omp parallel for reduction(+ : f) schedule(static) if (l > 256)
                for (int j = 0; j <= l; ++j) {
                    double s = 0.0;
                    for (int k = 0; k <= j; ++k) s += A(j, k) * A(i, k);
                    for (int k = j + 1; k <= l; ++k) s += A(k, j) * A(i, k);
                    e[j] = s / h;
                    f += e[j] * A(i, j);
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) e[j] -= hh * A(i, j);
#pragma omp parallel for schedule(static) if (l > 256)
                for (int j = 0; j <= l; ++j) {
                    const double fj = A(i, j), gj = e[j];
                    for (int k = 0; k <= j; ++k) A(j, k) -= fj * e[k] + gj * A(i, k);
                }
            }
        } else {
            e[i] = A(i, l);
        }
    }
    e[0] = 0.0;
    for (int i = 0; i < n; ++i) d[i] = A(i, i);
#undef A
    /* e[i] couples rows i-1 and i */
    int rc = lanczos_hip_tridiag_eigs(d, e + 1, n, nth, eig);
    free(d);
    free(e);
    return rc;
}
