/*
 * Synthetic matrices for tests and benchmarks, generated straight into CRS
 * (the reference ships no inputs: its .gitignore:6 keeps *.txt out of the tree).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lanczos_host.h"

void lanczos_free_crs(Mat_Crs *m) {
    if (!m) return;
    free(m->values);
    free(m->column_index);
    free(m->row_head_indexes);
    memset(m, 0, sizeof *m);
}

static int alloc_crs(Mat_Crs *out, size_t n, size_t nnz) {
    out->values = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    out->column_index = (int64_t *)malloc(sizeof(int64_t) * (nnz ? nnz : 1));
    out->dimension = n;
    out->length = nnz;
    if (!out->values || !out->column_index) {
        lanczos_free_crs(out);
        return 1;
    }
    return 0;
}

int lanczos_synth_laplacian_rows(int d, int64_t N, int64_t row_begin, int64_t row_end, Mat_Crs *out) {
    if (!out || d < 1 || d > 3 || N < 1) return 1;
    memset(out, 0, sizeof *out);
    int64_t n = 1;
    for (int i = 0; i < d; i++) {
        if (n > INT32_MAX / N) return 1;
        n *= N;
    }
    if (row_begin < 0 || row_end > n || row_begin >= row_end) return 1;
    const int64_t rows = row_end - row_begin;
    const int64_t stride[3] = {1, N, N * N};
    int64_t *head = (int64_t *)malloc(sizeof(int64_t) * (size_t)(rows + 1));
    if (!head) return 1;
    out->row_head_indexes = head;
    /* row lengths are known in closed form: 1 + number of in-range neighbours */
    head[0] = 0;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        int cnt = 1;
        int64_t rem = row_begin + r;
        for (int a = 0; a < d; a++) {
            const int64_t c = rem % N;
            rem /= N;
            cnt += (c > 0) + (c < N - 1);
        }
        head[r + 1] = cnt;
    }
    for (int64_t r = 0; r < rows; r++) head[r + 1] += head[r];
    if (alloc_crs(out, (size_t)rows, (size_t)head[rows])) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        const int64_t i = row_begin + r;
        int64_t coord[3] = {0, 0, 0};
        int64_t rem = i;
        for (int a = 0; a < d; a++) {
            coord[a] = rem % N;
            rem /= N;
        }
        int64_t p = head[r];
        for (int a = d - 1; a >= 0; a--) /* lower neighbours, ascending column */
            if (coord[a] > 0) {
                out->column_index[p] = i - stride[a];
                out->values[p++] = -1.0;
            }
        out->column_index[p] = i;
        out->values[p++] = 2.0 * d;
        for (int a = 0; a < d; a++)
            if (coord[a] < N - 1) {
                out->column_index[p] = i + stride[a];
                out->values[p++] = -1.0;
            }
    }
    return 0;
}

int lanczos_synth_laplacian(int d, int64_t N, Mat_Crs *out) {
    if (d < 1 || d > 3 || N < 1) return 1;
    int64_t n = 1;
    for (int i = 0; i < d; i++) {
        if (n > INT32_MAX / N) return 1;
        n *= N;
    }
    return lanczos_synth_laplacian_rows(d, N, 0, n, out);
}

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef struct {
    int32_t r, c;
    double v;
} ent_t;

/* Rows [row_begin, row_end) of lanczos_synth_random_symmetric(n, per_row, seed): the same entries,
 * column indices global, so that every rank of a row-partitioned run can build its own slab.  Every
 * draw of the whole matrix is replayed (counter-based generator, nothing stored) and the ones that
 * land in the slab, in either orientation, are kept. */
int lanczos_synth_random_symmetric_rows(int64_t n, int per_row, uint64_t seed, int64_t row_begin, int64_t row_end,
                                        Mat_Crs *out) {
    if (!out || n < 1 || n > INT32_MAX || per_row < 0) return 1;
    if (row_begin < 0 || row_end > n || row_begin >= row_end) return 1;
    memset(out, 0, sizeof *out);
    const uint64_t key = mix64(seed);
    const int64_t rows = row_end - row_begin;
    /* pass 1: count both orientations of every off-diagonal draw */
    int64_t *cnt = (int64_t *)calloc((size_t)rows + 1, sizeof(int64_t));
    if (!cnt) return 1;
    for (int64_t i = 0; i < n; i++)
        for (int t = 0; t < per_row; t++) {
            const int64_t j = (int64_t)(mix64(key ^ (uint64_t)(2 * (i * per_row + t))) % (uint64_t)n);
            if (j == i) continue;
            if (i >= row_begin && i < row_end) cnt[i - row_begin + 1]++;
            if (j >= row_begin && j < row_end) cnt[j - row_begin + 1]++;
        }
    for (int64_t i = 0; i < rows; i++) cnt[i + 1] += cnt[i];
    const size_t tot = (size_t)cnt[rows];
    ent_t *buf = (ent_t *)malloc(sizeof(ent_t) * (tot ? tot : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)rows);
    if (!buf || !cur) {
        free(cnt); free(buf); free(cur);
        return 1;
    }
    memcpy(cur, cnt, sizeof(int64_t) * (size_t)rows);
    for (int64_t i = 0; i < n; i++)
        for (int t = 0; t < per_row; t++) {
            const uint64_t id = (uint64_t)(i * per_row + t);
            const int64_t j = (int64_t)(mix64(key ^ (2 * id)) % (uint64_t)n);
            if (j == i) continue;
            const int in_i = i >= row_begin && i < row_end, in_j = j >= row_begin && j < row_end;
            if (!in_i && !in_j) continue;
            const double v = (double)(mix64(key ^ (2 * id + 1)) >> 11) / 4503599627370496.0 - 1.0;
            ent_t a = {(int32_t)i, (int32_t)j, v}, b = {(int32_t)j, (int32_t)i, v};
            if (in_i) buf[cur[i - row_begin]++] = a;
            if (in_j) buf[cur[j - row_begin]++] = b;
        }
    /* per row: sort by column (insertion sort, rows are short), drop repeats of a
     * column keeping the first (both orientations see the same winner only if
     * the order is orientation independent, so order ties by value bits) */
    int64_t *head = (int64_t *)malloc(sizeof(int64_t) * (size_t)(rows + 1));
    if (!head) { free(cnt); free(buf); free(cur); return 1; }
    head[0] = 0;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < rows; i++) {
        ent_t *row = buf + cnt[i];
        const int64_t m = cnt[i + 1] - cnt[i];
        for (int64_t a = 1; a < m; a++) {
            ent_t x = row[a];
            int64_t b = a - 1;
            while (b >= 0 && (row[b].c > x.c || (row[b].c == x.c && row[b].v > x.v))) {
                row[b + 1] = row[b];
                b--;
            }
            row[b + 1] = x;
        }
        int64_t uniq = 0;
        for (int64_t a = 0; a < m; a++)
            if (a == 0 || row[a].c != row[a - 1].c) uniq++;
        head[i + 1] = uniq + 1; /* + diagonal */
    }
    for (int64_t i = 0; i < rows; i++) head[i + 1] += head[i];
    out->row_head_indexes = head;
    if (alloc_crs(out, (size_t)rows, (size_t)head[rows])) { free(cnt); free(buf); free(cur); return 1; }
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t r = 0; r < rows; r++) {
        const int64_t i = row_begin + r;
        const ent_t *row = buf + cnt[r];
        const int64_t m = cnt[r + 1] - cnt[r];
        double absum = 0.0;
        for (int64_t a = 0; a < m; a++)
            if (a == 0 || row[a].c != row[a - 1].c) absum += fabs(row[a].v);
        int64_t p = head[r];
        int placed = 0;
        for (int64_t a = 0; a < m; a++) {
            if (a > 0 && row[a].c == row[a - 1].c) continue;
            if (!placed && row[a].c > i) {
                out->column_index[p] = i;
                out->values[p++] = 1.0 + absum;
                placed = 1;
            }
            out->column_index[p] = row[a].c;
            out->values[p++] = row[a].v;
        }
        if (!placed) {
            out->column_index[p] = i;
            out->values[p++] = 1.0 + absum;
        }
    }
    free(cnt);
    free(buf);
    free(cur);
    return 0;
}

int lanczos_synth_random_symmetric(int64_t n, int per_row, uint64_t seed, Mat_Crs *out) {
    return lanczos_synth_random_symmetric_rows(n, per_row, seed, 0, n, out);
}
