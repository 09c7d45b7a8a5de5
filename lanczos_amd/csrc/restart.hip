// Thick-restarted Lanczos on top of the engine: when the basis is full, the wanted Ritz vectors
// replace it and the run continues, so memory stays at max_iter vectors however many products the
// spectrum needs.  The reference keeps every vector of a run, V[max_iter][n]
// (src/lanczos.c:58-61, src/lanczos_cuda.c:424-427), and simply stops at max_iter
// (src/lanczos.c:193).  The projected matrix after a restart is diag(theta) joined by one coupling
// row to the tridiagonal of the new steps; it is small and solved on the host.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "engine.h"

extern "C" int lanczos_hip_thick_restart(lanczos_hip_engine *eh, int nth, int keep, int max_cycles, double tol,
                                         double *eig_out, double *resid_out, double *vectors_host,
                                         int *cycles_out, int *products_out) {
    return lz::guarded("lanczos_hip_thick_restart", [&]() -> int {
    lz::Engine *e = reinterpret_cast<lz::Engine *>(eh);
    if (!e || !eig_out || nth < 1 || max_cycles < 1) return lz::fail("lanczos_hip_thick_restart: bad argument");
    if (e->slab) return lz::fail("lanczos_hip_thick_restart: not available on a row-partitioned engine");
    const int m = e->max_iter - 1;  // steps that fill the basis
    if (keep < nth || keep + 2 > m)
        return lz::fail("lanczos_hip_thick_restart: need nth <= keep <= max_iter - 3 (nth=%d, keep=%d, max_iter=%d)",
                        nth, keep, e->max_iter);
    if (!e->started || e->k != 0) return lz::fail("lanczos_hip_thick_restart: set a start vector first");
    if (e->reorth != LANCZOS_HIP_REORTH_FULL && e->reorth != LANCZOS_HIP_REORTH_FULL2 &&
        e->reorth != LANCZOS_HIP_REORTH_MGS)
        return lz::fail("lanczos_hip_thick_restart: needs reorthogonalisation against the whole basis");
    const size_t mm = (size_t)m * (size_t)m;
    std::vector<double> T(mm), work(mm), S(mm), all(m), theta((size_t)keep), coup((size_t)keep, 0.0);
    std::vector<double> alpha((size_t)m), beta((size_t)m), resid((size_t)keep);
    int p = 0, products = 0, cycle = 0;
    for (;; ++cycle) {
        const int steps = m - p;
        if (int rc = e->run(steps, nullptr, nullptr)) return rc;
        products += steps;
        if (hipMemcpy(alpha.data(), e->d_alpha, sizeof(double) * m, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(beta.data(), e->d_beta, sizeof(double) * m, hipMemcpyDeviceToHost) != hipSuccess)
            return lz::fail("lanczos_hip_thick_restart: history read-back failed");
        std::fill(T.begin(), T.end(), 0.0);
        for (int i = 0; i < p; ++i) {
            T[(size_t)i * m + i] = theta[(size_t)i];
            T[(size_t)p * m + i] = T[(size_t)i * m + p] = coup[(size_t)i];
        }
        for (int j = p; j < m; ++j) {
            T[(size_t)j * m + j] = alpha[(size_t)j];
            if (j + 1 < m) T[(size_t)j * m + j + 1] = T[(size_t)(j + 1) * m + j] = beta[(size_t)j];
        }
        const double beta_last = beta[(size_t)m - 1];
        work = T;
        if (lanczos_hip_sym_lowest_eigs(work.data(), m, keep, theta.data()) != 0)
            return lz::fail("lanczos_hip_thick_restart: eigenvalue solve failed");
        work = T;
        if (lanczos_hip_sym_eig_jacobi(work.data(), m, all.data(), S.data()) != 0)
            return lz::fail("lanczos_hip_thick_restart: eigenvector solve failed");
        const double anorm = std::max(fabs(all[0]), fabs(all[(size_t)m - 1]));
        bool converged = true;
        for (int i = 0; i < keep; ++i) {
            coup[(size_t)i] = beta_last * S[(size_t)i * m + m - 1];
            resid[(size_t)i] = fabs(coup[(size_t)i]);  // the Lanczos bound on ||A y - theta y||
            if (i < nth && !(resid[(size_t)i] <= tol * anorm)) converged = false;
        }
        if (converged || cycle + 1 >= max_cycles) break;
        if (int rc = e->restart(keep, S.data(), coup[(size_t)keep - 1])) return rc;
        p = keep;
    }
    for (int i = 0; i < nth; ++i) eig_out[i] = theta[(size_t)i];
    if (resid_out)
        for (int i = 0; i < nth; ++i) resid_out[i] = resid[(size_t)i];
    if (cycles_out) *cycles_out = cycle + 1;
    if (products_out) *products_out = products;
    if (vectors_host) {
        double *d_y = nullptr;
        if (hipMalloc(&d_y, sizeof(double) * (size_t)e->ldv * (size_t)nth) != hipSuccess)
            return lz::fail("lanczos_hip_thick_restart: out of device memory");
        int rc = e->combine(m, nth, S.data(), d_y, e->ldv);
        if (rc == 0 && hipMemcpy2D(vectors_host, sizeof(double) * e->n, d_y, sizeof(double) * e->ldv,
                                   sizeof(double) * e->n, (size_t)nth, hipMemcpyDeviceToHost) != hipSuccess)
            rc = lz::fail("lanczos_hip_thick_restart: copy to the host failed");
        (void)hipFree(d_y);
        if (rc) return rc;
    }
    return 0;
    });
}
