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

/*
 * Matrix Market coordinate files.  No counterpart in the reference, whose reader takes a
 * "%%MatrixMarket" banner for a malformed index and exits (src/coo.c:19-22) and would read the size
 * line as an entry.  Banner "%%MatrixMarket matrix coordinate <real|integer|pattern>
 * <general|symmetric|skew-symmetric>", '%' comment lines, the size line "M N NNZ", then NNZ lines
 * "row col [value]" (1-based).  Symmetric and skew-symmetric files hold one triangle: the mirror
 * entries are appended after the stored ones.  Pattern entries get the value 1.
 */
typedef struct {
    int phase;           /* 0 banner, 1 size line, 2 entries */
    int pattern, mirror; /* mirror: 0 none, 1 symmetric, -1 skew-symmetric */
    long rows, cols, declared, seen;
    Coo *data;
    size_t count;
    int error;
} mm_t;

static void mm_lower(char *s) {
    for (; *s; ++s) *s = (char)tolower((unsigned char)*s);
}

static void mm_line(char *line, void *ctx) {
    mm_t *m = (mm_t *)ctx;
    if (m->error) return;
    if (m->phase == 0) {
        char banner[64] = "", object[64] = "", format[64] = "", field[64] = "", symmetry[64] = "";
        if (sscanf(line, "%63s %63s %63s %63s %63s", banner, object, format, field, symmetry) != 5 ||
            strcmp(banner, "%%MatrixMarket") != 0) {
            fprintf(stderr, "read_matrix_market: no %%%%MatrixMarket banner\n");
            m->error = 1;
            return;
        }
        mm_lower(object), mm_lower(format), mm_lower(field), mm_lower(symmetry);
        if (strcmp(object, "matrix") || strcmp(format, "coordinate")) {
            fprintf(stderr, "read_matrix_market: only 'matrix coordinate' files are supported (got %s %s)\n", object, format);
            m->error = 1;
            return;
        }
        if (!strcmp(field, "pattern")) m->pattern = 1;
        else if (strcmp(field, "real") && strcmp(field, "integer")) {
            fprintf(stderr, "read_matrix_market: field '%s' is not supported (real, integer, pattern)\n", field);
            m->error = 1;
            return;
        }
        if (!strcmp(symmetry, "symmetric")) m->mirror = 1;
        else if (!strcmp(symmetry, "skew-symmetric")) m->mirror = -1;
        else if (strcmp(symmetry, "general")) {
            fprintf(stderr, "read_matrix_market: symmetry '%s' is not supported\n", symmetry);
            m->error = 1;
            return;
        }
        m->phase = 1;
        return;
    }
    const char *p = skip_blank(line);
    if (*p == '%' || *p == '\0') return; /* comment or blank line */
    if (m->phase == 1) {
        if (sscanf(p, "%ld %ld %ld", &m->rows, &m->cols, &m->declared) != 3 || m->rows < 1 || m->cols < 1 ||
            m->declared < 0 || m->rows > INT32_MAX || m->cols > INT32_MAX) {
            fprintf(stderr, "read_matrix_market: bad size line '%s'\n", p);
            m->error = 1;
            return;
        }
        const size_t cap = (size_t)m->declared * (m->mirror ? 2 : 1);
        m->data = (Coo *)malloc(sizeof(Coo) * (cap ? cap : 1));
        if (!m->data) {
            fprintf(stderr, "read_matrix_market: malloc failed\n");
            m->error = 1;
            return;
        }
        m->phase = 2;
        return;
    }
    long r = 0, c = 0;
    double v = 1.0;
    const int got = m->pattern ? sscanf(p, "%ld %ld", &r, &c) : sscanf(p, "%ld %ld %lg", &r, &c, &v);
    if (got != (m->pattern ? 2 : 3) || r < 1 || c < 1 || r > m->rows || c > m->cols) {
        fprintf(stderr, "read_matrix_market: bad entry '%s'\n", p);
        m->error = 1;
        return;
    }
    if (m->seen >= m->declared) {
        fprintf(stderr, "read_matrix_market: more entries than the size line declares (%ld)\n", m->declared);
        m->error = 1;
        return;
    }
    m->seen++;
    Coo *e = &m->data[m->count++];
    e->index_row = (int)(r - 1);
    e->index_column = (int)(c - 1);
    e->value = v;
}

int read_matrix_market(const char *filepath, Mat_Coo *out) {
    if (!filepath || !out) {
        fprintf(stderr, "read_matrix_market: NULL argument\n");
        return EXIT_FAILURE;
    }
    FILE *probe = fopen(filepath, "r");
    if (!probe) {
        perror("fopen");
        return EXIT_FAILURE;
    }
    fclose(probe);
    mm_t m;
    memset(&m, 0, sizeof m);
    for_each_line(filepath, mm_line, &m);
    if (!m.error && m.phase != 2) {
        fprintf(stderr, "read_matrix_market: file ends before the size line\n");
        m.error = 1;
    }
    if (!m.error && m.seen != m.declared) {
        fprintf(stderr, "read_matrix_market: %ld entries declared, %ld found\n", m.declared, m.seen);
        m.error = 1;
    }
    if (m.error) {
        free(m.data);
        return EXIT_FAILURE;
    }
    if (m.mirror) { /* the other triangle, in file order, behind the stored entries */
        const size_t stored = m.count;
        for (size_t i = 0; i < stored; i++) {
            const Coo e = m.data[i];
            if (e.index_row == e.index_column) continue;
            Coo *t = &m.data[m.count++];
            t->index_row = e.index_column;
            t->index_column = e.index_row;
            t->value = m.mirror > 0 ? e.value : -e.value;
        }
    }
    out->data = m.data;
    out->length = m.count;
    out->dimension = (size_t)(m.rows > m.cols ? m.rows : m.cols);
    return EXIT_SUCCESS;
}

/* 1 when the file starts with the Matrix Market banner, 0 when not (or unreadable). */
int is_matrix_market(const char *filepath) {
    FILE *fp = filepath ? fopen(filepath, "r") : NULL;
    if (!fp) return 0;
    char head[16] = "";
    const size_t n = fread(head, 1, 14, fp);
    fclose(fp);
    return n == 14 && memcmp(head, "%%MatrixMarket", 14) == 0;
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
