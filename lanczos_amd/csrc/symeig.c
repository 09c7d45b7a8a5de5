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

/* All eigenpairs of a small dense symmetric matrix by cyclic Jacobi rotations
 * (the projected matrix of a thick-restarted run is an arrowhead joined to a
 * tridiagonal, of order at most the basis size).  a: n x n row-major,
 * destroyed; eig ascending; vec row i = unit eigenvector of eig[i]. */
int lanczos_hip_sym_eig_jacobi(double *a, int n, double *eig, double *vec) {
    if (!a || !eig || !vec || n < 1) return 1;
#define A(i, j) a[(size_t)(i) * (size_t)n + (size_t)(j)]
#define V(i, j) vec[(size_t)(i) * (size_t)n + (size_t)(j)]
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V(i, j) = (i == j) ? 1.0 : 0.0; /* columns of V^T: row i tracks e_i */
    double total = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) total += A(i, j) * A(i, j);
    const double floor_ = total * 1e-34;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
        if (off <= floor_) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A(p, q);
                if (apq == 0.0 || apq * apq < 1e-40 * fabs(A(p, p) * A(q, q)) + 1e-300) continue;
                const double theta = (A(q, q) - A(p, p)) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) { /* columns p, q */
                    const double akp = A(k, p), akq = A(k, q);
                    A(k, p) = c * akp - s * akq;
                    A(k, q) = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) { /* rows p, q */
                    const double apk = A(p, k), aqk = A(q, k);
                    A(p, k) = c * apk - s * aqk;
                    A(q, k) = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) { /* accumulated rotations, stored transposed */
                    const double vpk = V(p, k), vqk = V(q, k);
                    V(p, k) = c * vpk - s * vqk;
                    V(q, k) = s * vpk + c * vqk;
                }
            }
    }
    /* sort ascending, rows of vec along */
    for (int i = 0; i < n; ++i) eig[i] = A(i, i);
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j)
            if (eig[j] < eig[m]) m = j;
        if (m != i) {
            const double t = eig[i];
            eig[i] = eig[m];
            eig[m] = t;
            for (int k = 0; k < n; ++k) {
                const double u = V(i, k);
                V(i, k) = V(m, k);
                V(m, k) = u;
            }
        }
    }
#undef A
#undef V
    return 0;
}
