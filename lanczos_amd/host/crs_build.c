/*
 * COO -> CRS build and host CRS matvec: the behaviour of the reference's crs.c
 * kept bit for bit (same arrays for the same input, same messages, same exit).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lanczos_host.h"

static void die(const char *what, void *a, void *b, void *c, void *d) {
    fputs(what, stderr);
    free(a);
    free(b);
    free(c);
    free(d);
    exit(EXIT_FAILURE);
}

/* src/crs.c:8-153.  Stable counting sort: row counts, exclusive prefix sum,
 * scatter in input order through a per-row cursor. */
Mat_Crs convert_from_coo(const Mat_Coo *mat_coo, int swap_row_column) {
    char msg[256];
    if (!mat_coo || !mat_coo->data) die("convert_from_coo: mat_coo is NULL\n", NULL, NULL, NULL, NULL);
    const size_t dim = mat_coo->dimension, len = mat_coo->length;
    if (dim == 0) die("convert_from_coo: dimension is 0\n", NULL, NULL, NULL, NULL);
    if (len > SIZE_MAX / sizeof(double)) {
        snprintf(msg, sizeof msg, "convert_from_coo: allocation size overflow (length=%zu)\n", len);
        die(msg, NULL, NULL, NULL, NULL);
    }
    if (dim + 1 < dim || dim + 1 > SIZE_MAX / sizeof(size_t)) {
        snprintf(msg, sizeof msg, "convert_from_coo: allocation size overflow (dimension=%zu)\n", dim);
        die(msg, NULL, NULL, NULL, NULL);
    }
    double *values = len ? (double *)malloc(sizeof(double) * len) : NULL;
    int64_t *cols = len ? (int64_t *)malloc(sizeof(int64_t) * len) : NULL;
    if (len && (!values || !cols)) {
        snprintf(msg, sizeof msg, "convert_from_coo: malloc failed (length=%zu)\n", len);
        die(msg, values, cols, NULL, NULL);
    }
    /* head[r + 1] first holds the count of row r, then the prefix sum */
    int64_t *head = (int64_t *)calloc(dim + 1, sizeof(int64_t));
    int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * dim);
    if (!head || !cursor) {
        snprintf(msg, sizeof msg, "convert_from_coo: allocation failed (dimension=%zu)\n", dim);
        die(msg, values, cols, head, cursor);
    }
    const Coo *e = mat_coo->data;
    for (size_t i = 0; i < len; i++) {
        const int r = swap_row_column ? e[i].index_column : e[i].index_row;
        const int c = swap_row_column ? e[i].index_row : e[i].index_column;
        if (r < 0 || c < 0 || (size_t)r >= dim || (size_t)c >= dim) {
            snprintf(msg, sizeof msg,
                     "convert_from_coo: index out of range at entry %zu (row=%d, col=%d, dim=%zu)\n", i, r, c, dim);
            die(msg, values, cols, head, cursor);
        }
        head[(size_t)r + 1]++;
    }
    for (size_t r = 0; r < dim; r++) {
        cursor[r] = head[r];
        head[r + 1] += head[r];
    }
    if ((size_t)head[dim] != len) {
        snprintf(msg, sizeof msg, "convert_from_coo: internal error (row_ptr[dim]=%zu != length=%zu)\n",
                 (size_t)head[dim], len);
        die(msg, values, cols, head, cursor);
    }
    for (size_t i = 0; i < len; i++) {
        const int r = swap_row_column ? e[i].index_column : e[i].index_row;
        const int c = swap_row_column ? e[i].index_row : e[i].index_column;
        const int64_t pos = cursor[r]++;
        values[pos] = e[i].value;
        cols[pos] = (int64_t)c;
    }
    free(cursor);
    Mat_Crs out = {values, cols, head, len, dim};
    return out;
}

/* src/crs.c:155-178 */
void matvec_crs(const Mat_Crs *mat, const double vec[], double *dist) {
    if (!mat || !vec || !dist) {
        fprintf(stderr, "matvec_crs: NULL argument\n");
        exit(EXIT_FAILURE);
    }
    const int64_t n = (int64_t)mat->dimension;
#pragma omp parallel for schedule(guided) if (n >= 1024)
    for (int64_t i = 0; i < n; i++) {
        double s = 0.0;
        for (int64_t j = mat->row_head_indexes[i]; j < mat->row_head_indexes[i + 1]; j++)
            s += mat->values[j] * vec[mat->column_index[j]];
        dist[i] = s;
    }
}
