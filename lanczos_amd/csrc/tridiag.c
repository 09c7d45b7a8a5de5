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

/*
 * Eigenvectors of the same tridiagonal matrix for given eigenvalues, by inverse
 * iteration: (T - theta I) z = r solved with Gaussian elimination with partial
 * pivoting on the three diagonals, three sweeps, vectors of eigenvalues closer
 * than 1e-3 ||T|| re-orthogonalised against each other.  S is m x n row-major;
 * each row has unit 2-norm and a non-negative first component (the sign rule of
 * the reference's diagonalize_double, src/util.c:100-107).
 */
int lanczos_hip_tridiag_eigvecs(const double *alpha, const double *beta, int n, int m, const double *eig, double *S) {
    if (!alpha || !eig || !S || n < 1 || m < 1 || m > n || (n > 1 && !beta)) return 1;
    double tnorm = 0.0;
    for (int i = 0; i < n; i++) {
        double r = fabs(alpha[i]);
        if (i > 0) r += fabs(beta[i - 1]);
        if (i + 1 < n) r += fabs(beta[i]);
        if (r > tnorm) tnorm = r;
    }
    if (tnorm == 0.0) tnorm = 1.0;
    const double eps = DBL_EPSILON * tnorm;
    double *work = (double *)malloc(sizeof(double) * (size_t)n * 5);
    int *piv = (int *)malloc(sizeof(int) * (size_t)n);
    if (!work || !piv) { free(work); free(piv); return 1; }
    double *d = work, *u = work + n, *u2 = work + 2 * n, *l = work + 3 * n, *z = work + 4 * n;
    for (int k = 0; k < m; k++) {
        /* separate a repeated shift slightly so the factorisation differs */
        double theta = eig[k];
        for (int j = 0; j < k; j++)
            if (fabs(eig[j] - theta) < 10 * eps) theta = eig[j] + 10 * eps * (k - j);
        /* LU of T - theta I with row interchanges: rows i and i+1 */
        for (int i = 0; i < n; i++) {
            d[i] = alpha[i] - theta;
            u[i] = (i + 1 < n) ? beta[i] : 0.0;
            u2[i] = 0.0;
        }
        for (int i = 0; i + 1 < n; i++) {
            const double sub = beta[i]; /* entry (i+1, i) */
            if (fabs(sub) > fabs(d[i])) { /* swap rows i and i+1 */
                piv[i] = 1;
                const double m_ = d[i] / sub;
                const double di1 = d[i + 1], ui1 = u[i + 1];
                l[i] = m_;
                /* new row i = old row i+1: (sub, di1, ui1); new row i+1 = old row i - m_ * that */
                const double old_u = u[i];
                d[i] = sub;
                u[i] = di1;
                u2[i] = ui1;
                d[i + 1] = old_u - m_ * di1;
                u[i + 1] = -m_ * ui1;
            } else {
                piv[i] = 0;
                const double piv_d = (d[i] == 0.0) ? eps : d[i];
                d[i] = piv_d;
                const double m_ = sub / piv_d;
                l[i] = m_;
                d[i + 1] -= m_ * u[i];
                /* u[i+1] unchanged, u2[i] stays 0 */
            }
        }
        if (d[n - 1] == 0.0) d[n - 1] = eps;
        double *s = S + (size_t)k * n;
        for (int i = 0; i < n; i++) z[i] = 1.0 / sqrt((double)n) * ((i * 2654435761u >> 7 & 1) ? 1.0 : 0.7);
        for (int sweep = 0; sweep < 3; sweep++) {
            /* forward: apply the row operations to z */
            for (int i = 0; i + 1 < n; i++) {
                if (piv[i]) {
                    const double t = z[i];
                    z[i] = z[i + 1];
                    z[i + 1] = t - l[i] * z[i];
                } else {
                    z[i + 1] -= l[i] * z[i];
                }
            }
            /* back substitution with the upper factor (d, u, u2) */
            for (int i = n - 1; i >= 0; i--) {
                double t = z[i];
                if (i + 1 < n) t -= u[i] * z[i + 1];
                if (i + 2 < n) t -= u2[i] * z[i + 2];
                z[i] = t / d[i];
            }
            /* keep clustered vectors apart */
            for (int j = 0; j < k; j++) {
                if (fabs(eig[j] - eig[k]) < 1e-3 * tnorm) {
                    const double *sj = S + (size_t)j * n;
                    double dot = 0.0;
                    for (int i = 0; i < n; i++) dot += sj[i] * z[i];
                    for (int i = 0; i < n; i++) z[i] -= dot * sj[i];
                }
            }
            double nrm = 0.0;
            for (int i = 0; i < n; i++) nrm += z[i] * z[i];
            nrm = sqrt(nrm);
            if (!(nrm > 0.0) || !isfinite(nrm)) { free(work); free(piv); return 2; }
            for (int i = 0; i < n; i++) z[i] /= nrm;
        }
        const double sign = (z[0] < 0.0) ? -1.0 : 1.0;
        for (int i = 0; i < n; i++) s[i] = sign * z[i];
    }
    free(work);
    free(piv);
    return 0;
}
