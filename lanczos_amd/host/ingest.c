/*
 * Triplet text ingest: the host-side behaviour of the reference's coo.c kept
 * for its callers (src/main.c:45).  Input is headerless, one "row col value"
 * per line, 1-based.
 */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lanczos_host.h"

static const char *skip_blank(const char *p) {
    while (isspace((unsigned char)*p)) p++;
    return p;
}

static int take_index(const char **pp) {
    const char *p = skip_blank(*pp);
    char *stop = NULL;
    long v = strtol(p, &stop, 10);
    if (stop == p) { /* message and exit as src/coo.c:19-22, 27-30 */
        fprintf(stderr, "Error: Invalid integer %s\n", p);
        exit(EXIT_FAILURE);
    }
    *pp = stop;
    return (int)v;
}

/* src/coo.c:11-40 */
void parse_to_coo(char *src_line, Coo *dist) {
    const char *p = src_line;
    const int r = take_index(&p);
    const int c = take_index(&p);
    double v; /* left unset when the field is missing, as the reference does */
    p = skip_blank(p);
    if (sscanf(p, "%lg", &v) != 1) v = dist->value;
    dist->index_row = r - 1;
    dist->index_column = c - 1;
    dist->value = v;
}

/* Growable line buffer fed by fgets pieces so that lines of any length work. */
typedef struct {
    char *s;
    size_t len, cap;
} line_t;

static void line_push(line_t *l, const char *piece, size_t n) {
    if (l->len + n + 1 > l->cap) {
        size_t cap = l->cap ? l->cap : 4096;
        while (cap < l->len + n + 1) cap *= 2;
        char *t = (char *)realloc(l->s, cap);
        if (!t) {
            fprintf(stderr, "realloc failed\n");
            exit(EXIT_FAILURE);
        }
        l->s = t;
        l->cap = cap;
    }
    memcpy(l->s + l->len, piece, n);
    l->len += n;
    l->s[l->len] = '\0';
}

/* Calls fn(line, ctx) for every non-empty line of the file (src/coo.c:120-169). */
static void for_each_line(const char *path, void (*fn)(char *, void *), void *ctx) {
    FILE *fp = fopen(path, "r");
    if (!fp) {
        perror("fopen");
        exit(EXIT_FAILURE);
    }
    char piece[8192];
    line_t l = {NULL, 0, 0};
    while (fgets(piece, sizeof piece, fp)) {
        size_t n = strlen(piece);
        const int eol = n > 0 && piece[n - 1] == '\n';
        line_push(&l, piece, eol ? n - 1 : n);
        if (eol) {
            if (l.len > 0) fn(l.s, ctx); /* only zero-length lines are skipped */
            l.len = 0;
        }
    }
    if (ferror(fp)) {
        perror("fgets");
        exit(EXIT_FAILURE);
    }
    if (l.len > 0) fn(l.s, ctx); /* last line without a newline */
    free(l.s);
    fclose(fp);
}

typedef struct {
    Coo *data;
    size_t count;
    size_t dim;
} fill_t;

static void count_line(char *line, void *ctx) {
    (void)line;
    ((fill_t *)ctx)->count++;
}

static void fill_line(char *line, void *ctx) {
    fill_t *f = (fill_t *)ctx;
    Coo *e = &f->data[f->count++];
    e->value = 0.0;
    parse_to_coo(line, e);
    /* running maximum in size_t, so a negative index wraps exactly as the
     * reference's MAX(size_t, int) does (src/coo.c:231-232, src/util.h:4) */
    const size_t r = (size_t)e->index_row, c = (size_t)e->index_column;
    if (!(f->dim >= r)) f->dim = r;
    if (!(f->dim >= c)) f->dim = c;
}

/* src/coo.c:101-269 */
Mat_Coo read_mat_coo_incremental(const char *filepath) {
    fill_t f = {NULL, 0, 0};
    for_each_line(filepath, count_line, &f);
    const size_t total = f.count;
    if (total > SIZE_MAX / sizeof(Coo)) {
        fprintf(stderr, "Error: allocation size overflow\n");
        exit(EXIT_FAILURE);
    }
    if (total > 0) {
        f.data = (Coo *)malloc(sizeof(Coo) * total);
        if (!f.data) {
            fprintf(stderr, "malloc failed\n");
            exit(EXIT_FAILURE);
        }
    }
    f.count = 0;
    for_each_line(filepath, fill_line, &f);
    Mat_Coo m = {f.data, f.count, f.dim + 1};
    return m;
}

/* src/coo.c:271-283 */
void matvec_coo(const Mat_Coo *mat, const double vec[], double *dist) {
    if (!mat || !vec || !dist) {
        fprintf(stderr, "matvec_coo: NULL argument\n");
        exit(EXIT_FAILURE);
    }
    memset(dist, 0, sizeof(double) * mat->dimension);
    for (size_t i = 0; i < mat->length; i++) {
        const Coo *e = &mat->data[i];
        dist[e->index_row] += e->value * vec[e->index_column];
    }
}
