/* Orthogonality tracker for selective reorthogonalisation.
 *
 * The reference either reorthogonalises every new vector against the whole
 * basis (src/lanczos.c:156-167) or skips the two newest ones by a fixed rule
 * (src/lanczos_cuda.c:679-687).  This tracker decides per step from the
 * tridiagonal entries alone: it carries estimates w(j,i) of v_j . v_i through
 *
 *   beta_j w(j+1,i) = beta_i w(j,i+1) + (alpha_i - alpha_j) w(j,i)
 *                     + beta_{i-1} w(j,i-1) - beta_{j-1} w(j-1,i) + noise
 *
 * (which is v_i . A v_j = v_j . A v_i written out with the three-term
 * recurrence), w(j,j) = 1, w(j+1,j) = local rounding level, and asks for a
 * reorthogonalisation of the vector under construction when an estimate
 * passes sqrt(eps).  That vector and the next one are then both
 * reorthogonalised against everything before them, which returns both rows
 * of estimates to rounding level.  Semi-orthogonality at sqrt(eps) keeps the
 * computed tridiagonal equal to the projected matrix up to rounding, so the
 * Ritz values agree with those of the full scheme.  Host arithmetic only:
 * O(j) flops per step.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lanczos_hip.h"

struct lanczos_hip_omega {
    int cap;        /* largest number of steps */
    int next;       /* the step the next call must carry */
    int force;      /* the next vector is reorthogonalised regardless */
    int reorths;    /* reorthogonalised steps so far */
    double anorm;   /* running bound on the spectral radius from the tridiagonal */
    double *alpha, *beta;
    double *w_old, *w_cur, *w_new; /* rows j-1, j, j+1; cap + 2 entries each */
};

static const double kNoise = 0.2;  /* multiples of eps (beta_i + beta_j) per term */
static const double kLocal = 0.5;  /* w(j+1,j) in multiples of eps anorm / beta_j */

lanczos_hip_omega *lanczos_hip_omega_create(int max_steps) {
    if (max_steps < 1) return NULL;
    lanczos_hip_omega *o = (lanczos_hip_omega *)calloc(1, sizeof *o);
    if (!o) return NULL;
    o->cap = max_steps;
    const size_t row = (size_t)max_steps + 2;
    o->alpha = (double *)calloc((size_t)max_steps, sizeof(double));
    o->beta = (double *)calloc((size_t)max_steps, sizeof(double));
    o->w_old = (double *)calloc(3 * row, sizeof(double));
    if (!o->alpha || !o->beta || !o->w_old) {
        lanczos_hip_omega_destroy(o);
        return NULL;
    }
    o->w_cur = o->w_old + row;
    o->w_new = o->w_cur + row;
    lanczos_hip_omega_reset(o);
    return o;
}

void lanczos_hip_omega_destroy(lanczos_hip_omega *o) {
    if (!o) return;
    free(o->alpha);
    free(o->beta);
    /* the three rows are one allocation whose start is the lowest of the rotating pointers */
    double *base = o->w_old;
    if (o->w_cur && o->w_cur < base) base = o->w_cur;
    if (o->w_new && o->w_new < base) base = o->w_new;
    free(base);
    free(o);
}

void lanczos_hip_omega_reset(lanczos_hip_omega *o) {
    if (!o) return;
    const size_t row = (size_t)o->cap + 2;
    double *base = o->w_old;
    if (o->w_cur < base) base = o->w_cur;
    if (o->w_new < base) base = o->w_new;
    memset(base, 0, 3 * row * sizeof(double));
    o->w_old = base;
    o->w_cur = base + row;
    o->w_new = base + 2 * row;
    o->w_cur[0] = 1.0; /* w(0,0) */
    o->next = 0;
    o->force = 0;
    o->reorths = 0;
    o->anorm = 0.0;
}

int lanczos_hip_omega_step(lanczos_hip_omega *o, int j, double alpha_j, double beta_j, double *estimate) {
    if (!o || j != o->next || j >= o->cap) return -1;
    o->alpha[j] = alpha_j;
    o->beta[j] = beta_j;
    const double reach = fabs(alpha_j) + beta_j + (j > 0 ? o->beta[j - 1] : 0.0);
    if (reach > o->anorm) o->anorm = reach;
    const double *a = o->alpha, *b = o->beta, *wc = o->w_cur, *wo = o->w_old;
    double *wn = o->w_new;
    double worst = 0.0;
    for (int i = 0; i < j; ++i) {
        double t = b[i] * wc[i + 1] + (a[i] - alpha_j) * wc[i];
        if (i > 0) t += b[i - 1] * wc[i - 1];
        t -= b[j - 1] * wo[i];
        t += copysign(kNoise * DBL_EPSILON * (b[i] + beta_j), t);
        wn[i] = t / beta_j;
        if (fabs(wn[i]) > worst) worst = fabs(wn[i]);
    }
    wn[j] = kLocal * DBL_EPSILON * o->anorm / beta_j;
    if (wn[j] > worst) worst = wn[j];
    wn[j + 1] = 1.0;
    if (estimate) *estimate = worst;
    int need = o->force || !(worst <= sqrt(DBL_EPSILON)); /* NaN (beta = 0) also asks for it */
    if (need) {
        for (int i = 0; i <= j; ++i) wn[i] = DBL_EPSILON;
        o->force = !o->force; /* a triggered step forces the next; a forced step ends the pair */
        o->reorths += 1;
    }
    o->w_old = o->w_cur;
    o->w_cur = wn;
    o->w_new = (double *)wo;
    o->next = j + 1;
    return need;
}

void lanczos_hip_omega_set_beta(lanczos_hip_omega *o, int j, double beta_j) {
    if (o && j >= 0 && j < o->next) o->beta[j] = beta_j;
}

int lanczos_hip_omega_reorths(const lanczos_hip_omega *o) { return o ? o->reorths : 0; }

/* Flat image of the state for checkpoints: 4 header values, alpha, beta, rows j-1 and j. */
size_t lanczos_hip_omega_state_doubles(const lanczos_hip_omega *o) {
    return o ? 4 + 2 * (size_t)o->cap + 2 * ((size_t)o->cap + 2) : 0;
}

void lanczos_hip_omega_state_save(const lanczos_hip_omega *o, double *out) {
    const size_t row = (size_t)o->cap + 2;
    out[0] = o->next;
    out[1] = o->force;
    out[2] = o->reorths;
    out[3] = o->anorm;
    memcpy(out + 4, o->alpha, (size_t)o->cap * sizeof(double));
    memcpy(out + 4 + o->cap, o->beta, (size_t)o->cap * sizeof(double));
    memcpy(out + 4 + 2 * (size_t)o->cap, o->w_old, row * sizeof(double));
    memcpy(out + 4 + 2 * (size_t)o->cap + row, o->w_cur, row * sizeof(double));
}

int lanczos_hip_omega_state_load(lanczos_hip_omega *o, const double *in) {
    const size_t row = (size_t)o->cap + 2;
    if (in[0] < 0 || in[0] > o->cap) return -1;
    o->next = (int)in[0];
    o->force = in[1] != 0.0;
    o->reorths = (int)in[2];
    o->anorm = in[3];
    memcpy(o->alpha, in + 4, (size_t)o->cap * sizeof(double));
    memcpy(o->beta, in + 4 + o->cap, (size_t)o->cap * sizeof(double));
    memcpy(o->w_old, in + 4 + 2 * (size_t)o->cap, row * sizeof(double));
    memcpy(o->w_cur, in + 4 + 2 * (size_t)o->cap + row, row * sizeof(double));
    return 0;
}

/* A state saved by a tracker of another capacity, laid out for this one (zeros beyond what was saved).
 * -1 when the saved state holds more steps than this tracker can. */
int lanczos_hip_omega_state_fit(const lanczos_hip_omega *o, const double *in, size_t in_doubles, double *out) {
    if (!o || !in || !out || in_doubles < 8 || (in_doubles - 8) % 4 != 0) return -1;
    const size_t cap_in = (in_doubles - 8) / 4, cap = (size_t)o->cap;
    const size_t row_in = cap_in + 2, row = cap + 2;
    if (in[0] < 0 || (size_t)in[0] > cap) return -1;
    const size_t keep = cap_in < cap ? cap_in : cap, keep_row = row_in < row ? row_in : row;
    memset(out, 0, lanczos_hip_omega_state_doubles(o) * sizeof(double));
    memcpy(out, in, 4 * sizeof(double));
    memcpy(out + 4, in + 4, keep * sizeof(double));
    memcpy(out + 4 + cap, in + 4 + cap_in, keep * sizeof(double));
    memcpy(out + 4 + 2 * cap, in + 4 + 2 * cap_in, keep_row * sizeof(double));
    memcpy(out + 4 + 2 * cap + row, in + 4 + 2 * cap_in + row_in, keep_row * sizeof(double));
    return 0;
}
