/*
 * lanczos_host.h -- the host-side C API kept from the reference so that its
 * call sites compile unchanged: triplet ingest (reference src/coo.h:15-20),
 * COO->CRS build and CRS matvec (src/crs.h:23-24), the Matvec_General plugin
 * types and lanczos() (src/lanczos.h:5-36).  Matrix types come from
 * lanczos_hip.h.  lanczos() here runs on the HIP engine.
 */
#ifndef LANCZOS_HOST_H
#define LANCZOS_HOST_H

#include "lanczos_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ingest (src/coo.h:15-18) ------------------------------------------ */
/* One headerless "row col value" line, 1-based -> 0-based; a malformed index
 * prints "Error: Invalid integer ..." to stderr and exits (src/coo.c:11-40). */
void parse_to_coo(char *src_line, Coo *dist);
/* Two-pass reader: '\n'-separated lines, zero-length lines skipped,
 * dimension = largest index + 1 (src/coo.c:101-269). */
Mat_Coo read_mat_coo_incremental(const char *filepath);
/*
 * Matrix Market coordinate files (real / integer / pattern; general / symmetric / skew-symmetric,
 * the stored triangle mirrored).  An extension: the reference's reader takes the banner for a
 * malformed index and exits (src/coo.c:19-22).  Returns EXIT_SUCCESS and a malloc'd Mat_Coo, or
 * EXIT_FAILURE with a message on stderr.  is_matrix_market() tells whether a file starts with the
 * banner.
 */
int read_matrix_market(const char *filepath, Mat_Coo *out);
int is_matrix_market(const char *filepath);
/* dist = A vec by scatter-add in entry order (src/coo.c:271-283). */
void matvec_coo(const Mat_Coo *mat, const double vec[], double *dist);

/* ---- format build (src/crs.h:23-24) ------------------------------------ */
/* Stable counting sort into CRS, optional row/column exchange; bit-exact with
 * src/crs.c:8-153, same stderr messages and exit(EXIT_FAILURE) on bad input. */
Mat_Crs convert_from_coo(const Mat_Coo *mat_coo, int swap_row_column);
/* dist = A vec on the host (src/crs.c:155-178); kept so MAKE_MAT_MATVEC has
 * its tag function -- lanczos() does not call it. */
void matvec_crs(const Mat_Crs *mat, const double vec[], double *dist);

/* ---- solver API (src/lanczos.h:5-36) ----------------------------------- */
typedef void Matvec_General(const void *mat, const double *vec, double *dist);

typedef struct mat_matvec {
    void *mat;
    Matvec_General *matvec;
    size_t dimension;
} Mat_Matvec;

#ifndef __cplusplus
#define MAKE_MAT_MATVEC(mat_ptr)                                   \
    _Generic((mat_ptr),                                            \
        Mat_Coo *: ((Mat_Matvec){ .mat = (void *)(mat_ptr),        \
                                  .matvec = (Matvec_General *)matvec_coo, \
                                  .dimension = (mat_ptr)->dimension }),   \
        Mat_Crs *: ((Mat_Matvec){ .mat = (void *)(mat_ptr),        \
                                  .matvec = (Matvec_General *)matvec_crs, \
                                  .dimension = (mat_ptr)->dimension }))
#endif

/*
 * Same signature and error behaviour as src/lanczos.h:34-36 / src/lanczos.c:10-56
 * (message on stderr + exit(EXIT_FAILURE) for NULL / out-of-range arguments),
 * but the iteration runs on the GPU: a Mat_Crs operator (matvec == matvec_crs)
 * goes to lanczos_hip_crs() as it is, a Mat_Coo operator (matvec == matvec_coo)
 * is first converted with convert_from_coo(mat, 0).  Any other plugin goes to
 * lanczos_hip_operator(): the function is called on host memory once per iteration, as
 * src/lanczos.c:110 calls it, and everything else of the iteration runs on the GPU.
 */
void lanczos(const Mat_Matvec mat_matvec, double eigenvalues[], double *eigenvectors[],
             int nth_eig, int max_iter, double threshold);

/* ---- synthetic matrices for tests and benchmarks ------------------------ */
/*
 * d-dimensional (d = 1, 2, 3) Dirichlet Laplacian on an N^d grid: diagonal 2d,
 * off-diagonals -1, natural ordering i = x + N (y + N z), entries of a row in
 * ascending column order.  Arrays are malloc'd into *out; free with
 * lanczos_free_crs.  Returns 0 on success.
 */
int lanczos_synth_laplacian(int d, int64_t N, Mat_Crs *out);
/* Rows [row_begin, row_end) of the same matrix (column indices stay global), for
 * row-partitioned runs that never hold the whole matrix on one host. */
int lanczos_synth_laplacian_rows(int d, int64_t N, int64_t row_begin, int64_t row_end, Mat_Crs *out);
/*
 * Random sparse symmetric matrix: for every row `per_row` column draws uniform
 * in [0, n) from a counter-based generator, values uniform in (-1, 1), pattern
 * symmetrised (an entry and its mirror share one value, duplicates merged by
 * keeping the first), diagonal = 1 + sum of |off-diagonals| of the row.
 * Rows sorted by column.
 */
int lanczos_synth_random_symmetric(int64_t n, int per_row, uint64_t seed, Mat_Crs *out);
/* Rows [row_begin, row_end) of the same matrix (column indices stay global): every draw of the whole
 * matrix is replayed and the entries that land in the slab are kept, so a rank of a row-partitioned
 * run builds its slab without ever holding the whole matrix. */
int lanczos_synth_random_symmetric_rows(int64_t n, int per_row, uint64_t seed, int64_t row_begin, int64_t row_end,
                                        Mat_Crs *out);
void lanczos_free_crs(Mat_Crs *m);

#ifdef __cplusplus
}
#endif
#endif /* LANCZOS_HOST_H */
