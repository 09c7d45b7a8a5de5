/*
 * TEST INFRASTRUCTURE ONLY -- command-line shim around the reference's own
 * coo.c and crs.c, compiled from /root/reference/src where they lie (see
 * oracle/Makefile, target _ref).  It calls read_mat_coo_incremental
 * (src/coo.c:101), convert_from_coo (src/crs.c:8), matvec_crs (src/crs.c:155)
 * and matvec_coo (src/coo.c:271) and dumps their outputs as raw bytes so that
 * the CPU restatement and the product's host code can be compared with them
 * byte for byte.  No reference source is copied: this file only calls it.
 *
 * The reference's util.c (which also holds the legacy whole-file reader's
 * helper read_from_file) cannot be built here because it includes MKL's
 * mkl_lapacke.h, which the image lacks; nothing this driver calls reaches it,
 * and the link leaves that one symbol unresolved on purpose.
 *
 * usage: ref_driver <triplets.txt> <swap 0|1> <out.bin>
 * out.bin layout (little endian): u64 dim, u64 len,
 *   Coo[len] as read (i32 row, i32 col, f64 value),
 *   f64 values[len], i64 column_index[len], i64 row_head[dim+1],
 *   f64 x[dim], f64 y_crs[dim], f64 y_coo[dim]
 * x[i] is the xorshift64 sequence seeded 0x9E3779B97F4A7C15 mapped to
 * [-0.5, 0.5).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "crs.h"

int main(int argc, char **argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s <triplets.txt> <swap> <out.bin>\n", argv[0]);
        return 2;
    }
    Mat_Coo coo = read_mat_coo_incremental(argv[1]);
    Mat_Crs crs = convert_from_coo(&coo, atoi(argv[2]));
    uint64_t dim = coo.dimension, len = coo.length;
    double *x = malloc(sizeof(double) * dim), *y1 = malloc(sizeof(double) * dim),
           *y2 = malloc(sizeof(double) * dim);
    uint64_t s = 0x9E3779B97F4A7C15ULL;
    for (uint64_t i = 0; i < dim; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    matvec_crs(&crs, x, y1);
    matvec_coo(&coo, x, y2);
    FILE *fp = fopen(argv[3], "wb");
    if (!fp) { perror("fopen"); return 1; }
    fwrite(&dim, 8, 1, fp);
    fwrite(&len, 8, 1, fp);
    fwrite(coo.data, sizeof(Coo), len, fp);
    fwrite(crs.values, 8, len, fp);
    fwrite(crs.column_index, 8, len, fp);
    fwrite(crs.row_head_indexes, 8, dim + 1, fp);
    fwrite(x, 8, dim, fp);
    fwrite(y1, 8, dim, fp);
    fwrite(y2, 8, dim, fp);
    fclose(fp);
    return 0;
}
