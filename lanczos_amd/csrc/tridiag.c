/*
 * Lowest eigenvalues of a symmetric tridiagonal matrix by Sturm-sequence
 * bisection (host, plain C).
 *
 * The reference hands the dense ld x ld matrix T to MKL's dsyevr
 * (src/util.c:186-259) or cuSOLVER's Dsyevdx (src/lanczos_cuda.c:608-625)
 * every iteration with range 'I', il = 1, iu = min(nth_eig, n), values only.
 * T is tridiagonal by construction (src/lanczos.c:113, 189-190), so this
 * routine takes its diagonal and off-diagonal directly; for that range and
 * job LAPACK's driver also ends in a bisection on the tridiagonal form.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/lanczos_hip.h"

/* number of eigenvalues strictly below x (inertia of T - x I by the LDL^T
 * pivot recurrence, pivots kept away from zero) */
static int sturm_count(const double *a, const double *b2, int n, double x, double pivmin) {
    int cnt = 0;
    double d = a[0] - x;
    if (fabs(d) < pivmin) d = -pivmin;
    if (d < 0.0) cnt++;
    for (int i = 1; i < n; i++) {
        d = (a[i] - x) - b2[i - 1] / d;
        if (fabs(d) < pivmin) d = -pivmin;
        if (d < 0.0) cnt++;
    }
    return cnt;
}

int lanczos_hip_tridiag_eigs(const double *alpha, const double *beta, int n, int m, double *eig) {
    if (!alpha || !eig || n < 1 || m < 1 || (n > 1 && !beta)) return 1;
    const int iu = m < n ? m : n;
    double *b2 = (double *)malloc(sizeof(double) * (size_t)(n > 1 ? n - 1 : 1));
    if (!b2) return 1;
    double bmax2 = 0.0;
    for (int i = 0; i + 1 < n; i++) {
        b2[i] = beta[i] * beta[i];
        if (b2[i] > bmax2) bmax2 = b2[i];
    }
    /* Gershgorin interval */
    double lo = alpha[0], hi = alpha[0];
    for (int i = 0; i < n; i++) {
        double r = 0.0;
        if (i > 0) r += fabs(beta[i - 1]);
        if (i + 1 < n) r += fabs(beta[i]);
        if (alpha[i] - r < lo) lo = alpha[i] - r;
        if (alpha[i] + r > hi) hi = alpha[i] + r;
    }
    const double tnorm = fmax(fabs(lo), fabs(hi));
    const double pivmin = DBL_MIN * fmax(1.0, bmax2);
    const double pad = 2.0 * tnorm * DBL_EPSILON * n + 2.0 * pivmin;
    lo -= pad;
    hi += pad;
    const double floor_tol = DBL_EPSILON * tnorm / 1024.0;
    double prev = lo;
    for (int j = 0; j < iu; j++) {
        /* smallest x with count(x) >= j + 1; eigenvalues come out ascending,
         * so the previous one is a valid lower end */
        double l = prev, h = hi;
        for (int it = 0; it < 2200; it++) {
            const double mid = l + (h - l) * 0.5;
            if (!(mid > l) || !(mid < h)) break;
            if (sturm_count(alpha, b2, n, mid, pivmin) >= j + 1)
                h = mid;
            else
                l = mid;
            const double tol = fmax(2.0 * DBL_EPSILON * fmax(fabs(l), fabs(h)), floor_tol);
            if (h - l <= tol) break;
        }
        eig[j] = l + (h - l) * 0.5;
        prev = l;
    }
    for (int j = iu; j < m; j++) eig[j] = 0.0;
    free(b2);
    return 0;
}
