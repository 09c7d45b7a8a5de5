/*
 * lanczos_hip.h -- C ABI of the MI355X (gfx950) Lanczos engine.
 *
 * This is the whole boundary between a C host program and the HIP engine:
 * plain pointers and sizes, no C++ types, no exceptions.  A host that today
 * calls lanczos_cuda_crs() (reference src/lanczos_cuda.h:139-141) links
 * liblanczos_hip.so and calls lanczos_hip_crs() with the same arguments.
 *
 * Return convention (all int-returning functions): 0 (== EXIT_SUCCESS) on
 * success, non-zero on failure.  On failure a one-line message has been
 * printed to stdout, the way the reference's CHECK_*_GOTO macros do
 * (src/lanczos_cuda.h:76-124), and lanczos_hip_last_error() returns it.
 */
#ifndef LANCZOS_HIP_H
#define LANCZOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Host matrix types.  Field-for-field the reference's Coo / Mat_Coo
 * (src/coo.h:3-13) and Mat_Crs (src/crs.h:4-21), so a Mat_Crs built by the
 * reference's convert_from_coo() can be handed over unchanged.  A translation
 * unit that already included the reference's coo.h / crs.h defines
 * LANCZOS_HIP_HAVE_REFERENCE_TYPES before including this header.
 */
#ifndef LANCZOS_HIP_HAVE_REFERENCE_TYPES
typedef struct coo {
    int index_row;
    int index_column;
    double value;
} Coo;

typedef struct mat_coo {
    Coo *data;
    size_t length;
    size_t dimension;
} Mat_Coo;

typedef struct crs {
    double *values;
    int64_t *column_index;
    int64_t *row_head_indexes; /* length dimension + 1, last entry == length */
    size_t length;
    size_t dimension;
} Mat_Crs;
#endif

/* ------------------------------------------------------------------------ */
/* Drop-in solver entry                                                      */
/* ------------------------------------------------------------------------ */

/*
 * Replaces lanczos_cuda_crs (src/lanczos_cuda.h:139-141, body
 * src/lanczos_cuda.c:281-788); same argument meaning:
 *   mat           CRS matrix on the host, owned by the caller, not modified
 *   eigenvalues   caller buffer of >= nth_eig doubles; zero-filled on entry
 *                 (src/lanczos_cuda.c:359), on return the lowest nth_eig Ritz
 *                 values of the last tridiagonal matrix, ascending, entries
 *                 beyond its dimension zero (src/util.c:269-271)
 *   eigenvectors  accepted and ignored, as in both reference solvers
 *                 (src/lanczos.c:20, src/lanczos_cuda.c:284), unless
 *                 LANCZOS_HIP_EIGENVECTORS=1: then eigenvectors[i], i < nth_eig, when not NULL,
 *                 receives the Ritz vector of eigenvalues[i] (dimension doubles, unit norm)
 *   max_iter      basis size; the loop runs max_iter - 1 iterations
 *                 (src/lanczos.c:104)
 *   threshold     convergence threshold of src/lanczos.c:123, 185
 * Termination, iteration-line printing and the reorthogonalisation range
 * follow the CPU solver src/lanczos.c:104-191 (all l < k), which is the parity
 * target; stdout carries the same lines ("%d\t%.7f\t...", "converged ...",
 * "%.7f beta converged", "Reached max iterations (%d)"), stderr the
 * "[TOTAL] name=... sec=..." lines of src/lanczos_cuda.c:782-786.
 * Returns EXIT_SUCCESS on max-iter and on either convergence exit,
 * EXIT_FAILURE on bad arguments or a HIP error (src/lanczos_cuda.c:316-327).
 *
 * Environment knobs (the argv contract of src/main.c:25-42 is unchanged):
 *   LANCZOS_HIP_REORTH  full | full2 | mgs | cuda_compat | none | selective   (default full)
 *   LANCZOS_HIP_SEED    start-vector seed (default 123456789, the constant of
 *                       src/lanczos_cuda.c:395)
 *   LANCZOS_HIP_DEVICE  HIP device ordinal (default: current device)
 *   LANCZOS_HIP_NGPU    number of GPUs: above 1 the matrix is row-partitioned over devices 0..N-1 of
 *                       this node (lanczos_hip_multi_create below; no checkpoints or eigenvectors there);
 *                       LANCZOS_HIP_MULTI_LOOPBACK=1 keeps the slabs on one GPU, LANCZOS_HIP_MULTI_FORCE=1 takes
 *                       the row-slab path even with one slab (both for rehearsal on a one-GPU machine)
 *   LANCZOS_HIP_QUIET   1 suppresses the per-iteration stdout lines
 *   LANCZOS_HIP_WINDOW  iterations enqueued ahead of the one the host is looking at (default 4)
 *   LANCZOS_HIP_PROFILE 0 drops the per-kernel events behind the [TOTAL] lines (default 1)
 *   LANCZOS_HIP_JSON    1 adds one JSON line with the same split to stderr
 *   LANCZOS_HIP_CHECKPOINT  path of a checkpoint file, written every LANCZOS_HIP_CHECKPOINT_EVERY
 *                       iterations (0 = never, the default) beside the target and renamed over it
 *   LANCZOS_HIP_RESUME  path of such a file: the run continues from it (same matrix and
 *                       reorthogonalisation mode; max_iter may be larger) and ends bit for bit
 *                       as the uninterrupted run would have
 *   LANCZOS_HIP_EIGENVECTORS 1 fills the eigenvectors argument (off by default: the reference's
 *                       driver hands in an uninitialised pointer array, src/main.c:20)
 * and, read by every engine at creation:
 *   LANCZOS_HIP_NO_DICT    1 keeps the CRS kernels on matrices whose rows repeat
 *   LANCZOS_HIP_PIN_UPLOAD 0 uploads from pageable memory (default: the caller's arrays are page-locked
 *                          around the copies, as src/lanczos_cuda.c:144, 168 do)
 *   LANCZOS_HIP_PANEL_BITS log2 of the column-panel width of the tiled SpMV (default 16; 10 on matrices
 *                          with wildly uneven rows)
 *   LANCZOS_HIP_TILE_EVEN  1 / 0 forces the tiled SpMV's schedule for even / uneven row blocks (tests; by
 *                          default the row blocks' weights decide)
 *   LANCZOS_HIP_TILE_ROWBITS 11 to 14: rows per block of the tiled SpMV as a power of two (default: even
 *                          schedule 12, 11 for vectors up to 128 MiB; uneven 13, 14 beyond 256 MiB)
 *   LANCZOS_HIP_ROWS_PLANES 0: the row-wise CRS SpMV stays on the CRS arrays as they are; by default, when the rows'
 *                          column offsets repeat as patterns and the matrix is a grid in three dimensions (rows of
 *                          up to eight entries, fewer than 2^29 entries), it walks the grid's planes over a dictionary
 *                          of column offsets and reads the values from the CRS array (same bits)
 *   LANCZOS_HIP_ROWS_MASK  0: the row-wise CRS SpMV loads whole tiles instead of ending each group's loads at
 *                          its last entry (for comparison)
 *   LANCZOS_HIP_DICT_PLANES / LANCZOS_HIP_DICT_AHEAD / LANCZOS_HIP_DICT_WIDE  0 switches off, in turn, the
 *                          dictionary SpMV's plane-walking form (3-D grids), its forms that load one step
 *                          ahead, and their 16-byte loads (for comparison; the results are the same bits);
 *                          LANCZOS_HIP_DICT_BLOCK 256 runs the ahead form with 256-thread workgroups
 *   LANCZOS_HIP_DICT_SPAN  1024 / 2048: rows of a plane per step of the plane-walking dictionary SpMV (default: 2048 for grid
 *                          lines of 384 rows or more, else 1024); LANCZOS_HIP_DICT_CACHE 0 / 1 / 2: it keeps nothing / the
 *                          values / the values and window offsets of a wavefront's pattern in registers when all its rows
 *                          share one (default 0).  The same bits in every form; read when an engine first multiplies
 *   LANCZOS_HIP_NT_STORES  0 / 1: the recurrence with deferred normalisation writes its vector cacheably / past the caches
 *                          (default: past them from 32 MiB per vector on); read at every launch
 *   LANCZOS_HIP_VERBOSE    1 prints what the tiled SpMV's set-up chose (schedule, block shape, pieces,
 *                          entries kept out of the packed stream) on stderr
 */
int lanczos_hip_crs(const Mat_Crs *mat, double eigenvalues[], double *eigenvectors[],
                    int nth_eig, int max_iter, double threshold);

/*
 * The same solver for an operator that exists only as a host function: the reference's plugin
 * contract (src/lanczos.h:5, call site src/lanczos.c:110), where any
 *     void matvec(const void *mat, const double *vec, double *dist)
 * that overwrites dist[0..dimension) with A vec can be iterated on.  The Lanczos vectors, the
 * recurrence, the reorthogonalisation against the basis and the norms stay on the GPU; per
 * iteration the current vector goes to the host, `matvec` is called once on host memory exactly
 * as src/lanczos.c:110 calls it, and the product goes back.  Arguments, termination rules, output
 * lines, return codes and environment knobs are those of lanczos_hip_crs.
 */
typedef void lanczos_hip_matvec_fn(const void *mat, const double *vec, double *dist);
int lanczos_hip_operator(lanczos_hip_matvec_fn *matvec, const void *mat, size_t dimension, double eigenvalues[],
                         double *eigenvectors[], int nth_eig, int max_iter, double threshold);

/* ------------------------------------------------------------------------ */
/* Engine handle API (what lanczos_hip_crs is built from; used by tests and  */
/* benchmarks, and by multi-GPU hosts that interleave collectives)           */
/* ------------------------------------------------------------------------ */

typedef struct lanczos_hip_engine lanczos_hip_engine;

/* Reorthogonalisation range/order of one iteration k. */
enum {
    LANCZOS_HIP_REORTH_FULL = 0,        /* all l < k (src/lanczos.c:156), one blocked
                                           classical Gram-Schmidt pass after the
                                           three-term recurrence */
    LANCZOS_HIP_REORTH_CUDA_COMPAT = 1, /* l < k-2 (src/lanczos_cuda.c:679-687) */
    LANCZOS_HIP_REORTH_NONE = 2,        /* three-term recurrence only; keeps 3 vectors */
    LANCZOS_HIP_REORTH_MGS = 3,         /* all l < k, one basis vector at a time in
                                           ascending l: the exact order of
                                           src/lanczos.c:156-167 */
    LANCZOS_HIP_REORTH_FULL2 = 4,       /* all l < k, two blocked passes */
    LANCZOS_HIP_REORTH_SELECTIVE = 5    /* all l <= k in one blocked pass, but only in the
                                           steps where the tracked loss of orthogonality
                                           passes sqrt(eps), and in the step after each
                                           (the reference has a fixed rule instead:
                                           src/lanczos_cuda.c:679-687) */
};

enum {
    LANCZOS_HIP_SPMV_AUTO = 0,
    LANCZOS_HIP_SPMV_VECTOR = 1, /* 2^g lanes per row, g chosen from mean row length */
    LANCZOS_HIP_SPMV_STREAM = 2, /* one wavefront per group of 64 rows, nnz streamed through LDS */
    LANCZOS_HIP_SPMV_PIPE = 3,   /* stream form, software pipelined across row groups */
    LANCZOS_HIP_SPMV_PIPE_NT = 4, /* pipe form with non-temporal col/val loads */
    LANCZOS_HIP_SPMV_ROWS = 5,   /* pipe form whose gathers run row-wise (coalesced for stencil-like
                                    matrices); needs every 64-row group to hold <= 481 entries */
    LANCZOS_HIP_SPMV_PANEL = 6,  /* for matrices whose rows reach all over a vector too large for the
                                    caches (random sparsity, large power-law graphs): entries regrouped on
                                    the device by (block of 2048 to 16384 rows, column panel of 2^16 or 2^10 columns);
                                    a workgroup keeps its rows of y in LDS and sweeps the panels, whose slice
                                    of x stays in cache; row blocks too full for one workgroup are cut into
                                    pieces whose sums a second kernel adds in order.  Needs nnz < 2^32.
                                    LANCZOS_HIP_PANEL_BITS overrides the panel width. */
    LANCZOS_HIP_SPMV_DICT = 7,   /* dictionary of row patterns ((col - row, value) lists): one 16-bit pattern
                                    number per row replaces the row's entries; for matrices whose rows
                                    repeat (stencils on grids), at most 65535 patterns of at most 32
                                    entries, found and verified entry by entry on the device */
    LANCZOS_HIP_SPMV_MERGE = 8   /* equal shares of entries per wavefront, rows found by binary search: for
                                    wildly uneven row lengths (power-law graphs) */
};

/* Kernel classes for lanczos_hip_profile(). */
enum {
    LANCZOS_HIP_K_SPMV = 0,
    LANCZOS_HIP_K_RECUR = 1,
    LANCZOS_HIP_K_SCALE = 2,
    LANCZOS_HIP_K_DOTS = 3,
    LANCZOS_HIP_K_UPDATE = 4,
    LANCZOS_HIP_K_SMALL = 5, /* partial-sum finalisers */
    LANCZOS_HIP_K_COUNT = 6
};

typedef struct lanczos_hip_options {
    size_t struct_size;  /* sizeof(lanczos_hip_options), for forward compatibility */
    int device;          /* HIP device ordinal, -1 = the current device */
    void *stream;        /* hipStream_t to launch on; NULL = a private non-blocking
                            stream, as src/lanczos_cuda.c:364 */
    int reorth;          /* LANCZOS_HIP_REORTH_* */
    int spmv_kernel;     /* LANCZOS_HIP_SPMV_* */
    int profile;         /* non-zero: bracket every kernel with HIP events and
                            accumulate per-class time (lanczos_hip_profile) */
    /*
     * Row-partitioned (multi-GPU) use.  The engine then holds rows
     * [row_begin, row_begin + n) of an n_global-row matrix; column_index
     * values passed to lanczos_hip_create index an external gather buffer
     * x_ext of n_ext doubles that the host fills (all-gather or halo exchange)
     * before each SpMV.  Leave all of them 0 / NULL for single-GPU use.
     */
    int64_t n_global;
    int64_t row_begin;
    int64_t n_ext;
    double *x_ext; /* device pointer, n_ext doubles, owned by the caller */
    /*
     * Row-partitioned use without the gather buffer: the caller owns the Krylov basis.
     * basis points to nvec * basis_ld doubles of device memory (nvec = 3 with
     * LANCZOS_HIP_REORTH_NONE, else max_iter), vector j at basis + slot(j) * basis_ld with
     * slot(j) = j % 3 resp. j.  The SpMV of iteration k reads x straight from vector k:
     * entries [0, n) are the slab itself, and column indices >= lanczos_hip_gather_base(n)
     * address the tail of the same vector, which the host fills with the remote entries
     * (halo exchange) before lanczos_hip_phase_spmv -- no copy of the vector per iteration.
     * n_ext = gather base + number of remote entries <= basis_ld; basis_ld even, basis
     * 16-byte aligned.  Entries [n, gather base) must stay zero; the engine clears the whole
     * basis at lanczos_hip_set_start and its vector kernels never touch the tails.
     * x_ext must be NULL.
     */
    double *basis;
    int64_t basis_ld;
} lanczos_hip_options;

/* First x index of the remote entries for an engine created with options.basis:
 * n rounded up to the engine's vector tile (2048 doubles). */
int64_t lanczos_hip_gather_base(int64_t n);

/*
 * Upload a CRS matrix (the three arrays of Mat_Crs, src/crs.h:4-21) and
 * allocate the Krylov basis: max_iter vectors of n doubles as
 * src/lanczos_cuda.c:424-427 does, or 3 vectors with LANCZOS_HIP_REORTH_NONE.
 * Column indices are narrowed to 32 bits on the device (the reference caps the
 * dimension at INT_MAX, src/lanczos_cuda.c:319-322) and row offsets to 32 bits
 * when nnz < 2^31.  Replaces create_cusparse_matrix (src/lanczos_cuda.c:32-260)
 * plus the allocations of src/lanczos_cuda.c:424-427.
 */
int lanczos_hip_create(lanczos_hip_engine **out, int64_t n, int64_t nnz,
                       const int64_t *row_head_indexes, const int64_t *column_index,
                       const double *values, int max_iter, const lanczos_hip_options *opt);

/*
 * Same engine, built from triplets: the COO -> CRS conversion of convert_from_coo
 * (src/crs.c:8-153; swap_row_column as there) runs on the device -- the triplets are
 * uploaded once (16 B each) and the CRS arrays appear where the SpMV reads them,
 * with rows in input order and duplicates kept, bit for bit what the host build
 * gives.  An out-of-range index fails with the message of src/crs.c:69-71 (the
 * host build exits the process there).  Fewer than 2^32 entries.
 */
int lanczos_hip_create_from_coo(lanczos_hip_engine **out, const Mat_Coo *mat_coo, int swap_row_column,
                                int max_iter, const lanczos_hip_options *opt);
/* Test hook: copy the device CRS back in the layout of Mat_Crs (caller-allocated
 * row_head_indexes[n+1], column_index[nnz], values[nnz]; the last two may be NULL). */
int lanczos_hip_download_crs(lanczos_hip_engine *e, int64_t *row_head_indexes, int64_t *column_index,
                             double *values);

/* Replaces destroy_cusparse_matrix and the cleanup block src/lanczos_cuda.c:720-779. */
void lanczos_hip_destroy(lanczos_hip_engine *e);

/*
 * Start vector.  set_start copies n host doubles into basis slot 0 and
 * normalises them by multiplying with 1/sqrt(v.v) (src/lanczos.c:95-98,
 * src/lanczos_cuda.c:436-442); set_start_seed draws N(0,1) entries on the host
 * from a counter-based generator (element i depends only on seed and the
 * global row index, so every row partition sees the same vector) -- the
 * deterministic stand-in for src/util.c:48-56 / src/lanczos_cuda.c:394-432.
 * Both reset the iteration counter to 0.  In row-partitioned use the norm must
 * be global: pass it through lanczos_hip_set_start_scaled.
 */
int lanczos_hip_set_start(lanczos_hip_engine *e, const double *v0_host);
int lanczos_hip_set_start_seed(lanczos_hip_engine *e, uint64_t seed);
/* Fill out[0..n) with the seed's N(0,1) sequence for global rows
 * [row_begin, row_begin+n) (host only, no GPU needed). */
void lanczos_hip_gaussian_fill(uint64_t seed, int64_t row_begin, int64_t n, double *out);

/*
 * Run `steps` more iterations of src/lanczos.c:104-191 / src/lanczos_cuda.c:557-714
 * without host round trips, then return alpha_k and beta_k of those
 * iterations (host arrays of `steps` doubles; either may be NULL).  Blocks
 * until the stream has drained.  At most max_iter - 1 iterations in total
 * (3-vector mode: max(max_iter, 65536) per run, the length of the alpha/beta history).
 */
int lanczos_hip_run(lanczos_hip_engine *e, int steps, double *alpha_out, double *beta_out);

/* From now on the engine's product is the caller's host function (see lanczos_hip_operator) instead
 * of the uploaded matrix, which may be empty (nnz = 0).  Single-GPU engines. */
int lanczos_hip_set_operator(lanczos_hip_engine *e, lanczos_hip_matvec_fn *matvec, const void *mat);

/* Number of iterations completed since the last set_start. */
int lanczos_hip_iterations(const lanczos_hip_engine *e);

/* Test hook: y = A x with host vectors of n_ext (x) and n (y) doubles, through
 * the same SpMV kernel the iteration uses (src/crs.c:155-178,
 * src/lanczos_cuda.c:562-577). */
int lanczos_hip_spmv(lanczos_hip_engine *e, const double *x_host, double *y_host);

/*
 * The same product on DEVICE vectors, without any allocation: y_dev[0..n) = A x_dev[0..n_ext) and,
 * when dot_dev is not NULL, dot_dev[0] = x_dev[0..n) . y_dev, enqueued on the engine's stream (the
 * call does not wait for it).  No counterpart in the reference.  It is for callers that cannot
 * allocate: a stream under HIP graph capture, a tight serving loop, a harness that forbids
 * hipMalloc -- lanczos_hip_spmv stages its host vectors through two hipMalloc'd buffers per call.
 * The kernels are the ones lanczos_hip_spmv runs (y is bit for bit the same); their scratch memory
 * (per-workgroup partial sums; on the merge SpMV the rows shared between tiles; on the tiled SpMV
 * the sums of the pieces of split row blocks) is a fixed array of 3 178 496 bytes in the library's
 * device image, one copy per GPU, held until the process ends.  Limits: on the merge SpMV at most
 * 67 108 864 entries, on the tiled SpMV at most 262 144 rows of split-block pieces (32 pieces of 8192 rows), otherwise none beyond
 * the engine's own; beyond them the call fails with a message and launches nothing.  Calls on the same
 * GPU share the array: two that overlap in time (different streams or threads) race on it, and the
 * engine itself must not be used from another thread while the call runs (its scratch pointers are
 * swapped for the fixed array until the launch is enqueued).  Refused on an engine whose operator is
 * a host function (lanczos_hip_set_operator): there is no device product to launch.
 */
int lanczos_hip_spmv_noalloc(lanczos_hip_engine *e, const double *x_dev, double *y_dev, double *dot_dev);

/* Copy basis vector k (n doubles) to the host. */
int lanczos_hip_get_vector(lanczos_hip_engine *e, int k, double *out_host);

/*
 * Per-class kernel statistics gathered when options.profile != 0:
 * accumulated HIP-event milliseconds, launch count and algorithmic bytes
 * (each operand read once, each result written once).
 */
int lanczos_hip_profile(const lanczos_hip_engine *e, int kernel_class, double *ms,
                        int64_t *launches, double *bytes);
void lanczos_hip_profile_reset(lanczos_hip_engine *e);
/* Which kernel classes are timed (bit LANCZOS_HIP_K_* set: bracketed by two events per launch; default all).  The
 * events cost a run of short kernels several per cent (256^3 without reorthogonalisation: 6940 it/s with none, 6300
 * with all), so a timed run selects the classes it reports; launches and bytes are counted for every class. */
int lanczos_hip_profile_select(lanczos_hip_engine *e, unsigned class_mask);

/* Name of the SpMV kernel variant the engine selected ("dict27", "rows64", "pipe16", "panel16",
 * "merge", "vector8", ...).  LANCZOS_HIP_SPMV_AUTO decides from the matrix as it sits on the device:
 * when, on a sample of rows, most neighbouring rows hold neighbouring columns, "dict" if the rows
 * repeat as at most 65535 patterns, else "rows" if every 64-row group fits one tile; otherwise
 * "panel" when the sampled rows reach across more than 16 MiB of the vector; when the fullest
 * 64-row group holds 16 times the average, "panel" (narrow panels, split row blocks) if the vector
 * exceeds 6 MiB and the sampled rows reach across 2 MiB of it, else "merge"; else "pipe". */
const char *lanczos_hip_spmv_name(const lanczos_hip_engine *e);

/*
 * Lowest m eigenvalues, ascending, of the symmetric tridiagonal matrix with
 * diagonal alpha[0..n) and off-diagonal beta[0..n-1), by Sturm-sequence
 * bisection on the host.  Replaces the dense MKL / cuSOLVER solve of
 * src/util.c:118-272 and src/lanczos_cuda.c:598-640, which are handed the same
 * tridiagonal matrix in dense form every step.  Entries eig[min(m,n)..m) are
 * zeroed (src/util.c:269-271).
 */
int lanczos_hip_tridiag_eigs(const double *alpha, const double *beta, int n, int m, double *eig);

/* Eigenvectors of the same tridiagonal matrix for eigenvalues eig[0..m) (as returned
 * by lanczos_hip_tridiag_eigs), by inverse iteration on the host; S is m x n
 * row-major, rows of unit norm with a non-negative first component (the sign rule
 * of the reference's unused diagonalize_double, src/util.c:100-107). */
int lanczos_hip_tridiag_eigvecs(const double *alpha, const double *beta, int n, int m, const double *eig,
                                double *S);

/*
 * Ritz pairs after `steps` iterations: eig[0..m) = lowest m Ritz values, row i of
 * vectors_host (m x n row-major, may be NULL) = sum_j s_i[j] v_j with s_i the
 * eigenvector of the steps x steps tridiagonal matrix, formed on the device from
 * the resident basis; residual_est[i] (may be NULL) = |beta_{steps-1} s_i[steps-1]|,
 * the Lanczos bound on ||A y_i - eig_i y_i||.  Both reference solvers accept an
 * `eigenvectors` argument and never fill it (src/lanczos.c:20,
 * src/lanczos_cuda.c:284); this is the missing piece, as a separate call because
 * the reference's driver passes an uninitialised pointer array (src/main.c:20).
 * Needs the whole basis on one GPU (any reorth mode but none).
 */
int lanczos_hip_ritz_vectors(lanczos_hip_engine *e, int steps, int m, double *eig, double *vectors_host,
                             double *residual_est);

/*
 * Checkpoint / resume (the reference keeps V and T only in memory and frees them at
 * exit, src/lanczos.c:195-207).  save writes alpha/beta of the iterations done,
 * every resident basis vector and the pending partial sums to `path`; load
 * restores them into an engine created for the same matrix and reorth mode (max_iter may be
 * larger), after which lanczos_hip_run continues bit for bit.  Single-GPU engines.
 */
int lanczos_hip_checkpoint_save(lanczos_hip_engine *e, const char *path);
int lanczos_hip_checkpoint_load(lanczos_hip_engine *e, const char *path);

/* Message of the most recent failure on the calling thread ("" if none). */
const char *lanczos_hip_last_error(void);

/* ------------------------------------------------------------------------ */
/* Phase-level entry points for row-partitioned runs                         */
/* ------------------------------------------------------------------------ */
/*
 * One iteration k, split where a multi-GPU host must reduce across ranks.
 * Each call only enqueues kernels on the engine's stream.  `red` is a device
 * buffer of doubles owned by the caller; a phase leaves this rank's partial
 * sums there and the host all-reduces them in place (sum) before the next
 * phase, which reads the reduced values.  The reference has no multi-GPU code;
 * these have no counterpart in it beyond the loop body they split
 * (src/lanczos.c:109-190).
 *
 *   lanczos_hip_phase_spmv    w = A x_ext ; red[0] = local v_k . w
 *   lanczos_hip_phase_recur   alpha = red[0] ; w -= beta_prev v_{k-1} + alpha v_k ;
 *                             red[0..nl) = local V_l . w for the nl basis vectors
 *                             of the reorthogonalisation range and red[nl] = local w.w
 *                             (*n_red = nl + 1 doubles to reduce, *is_norm = 0);
 *                             with nl == 0: red[0] = local w.w, *n_red = 1, *is_norm = 1
 *   lanczos_hip_phase_update  w -= sum_l red[l] V_l.  After phase_recur: red[0] = the
 *                             GLOBAL norm^2 of the new w, red[nl] - sum_l red[l]^2 (the
 *                             basis is orthonormal and the red[l] are rounding-level
 *                             corrections, so nothing cancels): no further reduction.
 *                             After phase_dots: red[0] = local w.w, to be reduced
 *   lanczos_hip_phase_scale   beta = sqrt(red[0]) ; v_{k+1} = w / beta ; k += 1
 *
 * lanczos_hip_local_vector returns the device address of basis vector k's
 * local slab so the host can gather it into x_ext.
 */
int lanczos_hip_phase_spmv(lanczos_hip_engine *e, double *red);
int lanczos_hip_phase_recur(lanczos_hip_engine *e, double *red, int *n_red, int *is_norm);
/* second Gram-Schmidt pass (LANCZOS_HIP_REORTH_FULL2): red[0..nl) = local V_l . w again */
int lanczos_hip_phase_dots(lanczos_hip_engine *e, double *red, int *n_red);
int lanczos_hip_phase_update(lanczos_hip_engine *e, double *red);
int lanczos_hip_phase_scale(lanczos_hip_engine *e, double *red);
/*
 * LANCZOS_HIP_REORTH_NONE on a row-partitioned engine keeps the basis unnormalised,
 * u_k = beta_{k-1} v_k (the recurrence of src/lanczos.c:150-180 with that substitution), so an
 * iteration is two phases and ONE reduction of two doubles:
 *   lanczos_hip_phase_spmv        w = A x_ext ; red[0] = local u_k . w ; red[1] = local |u_k|^2
 *   lanczos_hip_phase_recur_lazy  alpha_k, beta_{k-1} from the reduced red[0..2) ;
 *                                 u_{k+1} = w / beta_{k-1} - ... ; k += 1
 * (lanczos_hip_phase_recur and lanczos_hip_phase_scale are refused there.)  The vector the host
 * gathers into x_ext is u_k as it stands.  beta of the last iteration is still pending when the
 * run stops: lanczos_hip_phase_lazy_norm leaves the local |u_k|^2 in red[0]; after the reduction
 * lanczos_hip_phase_lazy_close records beta_{k-1} = sqrt(red[0]) for lanczos_hip_history.
 */
int lanczos_hip_phase_recur_lazy(lanczos_hip_engine *e, double *red);
/*
 * lanczos_hip_phase_spmv in two parts, so that the exchange of the remote entries runs while the
 * product does (engines with options.basis whose matrix took the dictionary, rows, stream or
 * vector SpMV: lanczos_hip_phase_spmv_can_overlap).  Rows [first_interior_row, end_interior_row) are the ones
 * that read no remote entry.
 *   lanczos_hip_phase_spmv_begin  may be enqueued while the tail of vector k is still being
 *                                 received: w = A x and u_k . w for the interior rows (the
 *                                 dictionary form computes every row, the others not counting)
 *   lanczos_hip_phase_spmv_end    once the receives are ordered before it on the stream: the
 *                                 other rows again (plain row-wise kernel, src/crs.c:167-177),
 *                                 then red[] as lanczos_hip_phase_spmv leaves it
 */
int lanczos_hip_phase_spmv_begin(lanczos_hip_engine *e, int64_t first_interior_row, int64_t end_interior_row);
int lanczos_hip_phase_spmv_end(lanczos_hip_engine *e, double *red);
int lanczos_hip_phase_spmv_can_overlap(const lanczos_hip_engine *e);
int lanczos_hip_phase_lazy_norm(lanczos_hip_engine *e, double *red);
int lanczos_hip_phase_lazy_close(lanczos_hip_engine *e, double *red);
double *lanczos_hip_local_vector(lanczos_hip_engine *e, int k);
/* Enqueue a device-to-device copy of basis vector k's local slab (n doubles) to dst. */
int lanczos_hip_export_vector(lanczos_hip_engine *e, int k, double *dst_device);
/* v_0 = scale * v0_host, for a caller that has reduced the norm across ranks. */
int lanczos_hip_set_start_scaled(lanczos_hip_engine *e, const double *v0_host, double scale);
/* Copy alpha/beta history [0, iterations) from the device (row-partitioned
 * runs record the reduced values as the phases consume them). */
int lanczos_hip_history(lanczos_hip_engine *e, double *alpha_out, double *beta_out, int count);

/* ---- several GPUs of one node from one host process ------------------------
 * No counterpart in the reference (one device, one stream: src/lanczos_cuda.c:364).  The matrix is cut
 * into ngpu contiguous row slabs, one engine per slab on devices[p] (NULL: devices 0..ngpu-1); per
 * iteration the slabs exchange the current vector (grouped ncclSend/ncclRecv into the tails of the
 * neighbours' vectors when every remote column lives in an adjacent slab, beside the interior rows'
 * product; ncclAllGather otherwise) and all-reduce alpha, then the reorthogonalisation coefficients
 * together with w.w (one reduction of two doubles with LANCZOS_HIP_REORTH_NONE), all through RCCL on
 * the engines' streams.  librccl.so.1 is opened when the first such handle is created.  options:
 * reorth (any but LANCZOS_HIP_REORTH_MGS), spmv_kernel, profile; the rest is set per slab.
 * LANCZOS_HIP_MULTI_LOOPBACK places every slab on ONE device (devices[0], else options.device, else
 * 0) and replaces the collectives by copies and a summing kernel on one stream: the same partition,
 * column remap and iteration order where only one GPU exists.
 * lanczos_hip_crs() takes this path when LANCZOS_HIP_NGPU > 1 (LANCZOS_HIP_MULTI_LOOPBACK=1 for the
 * loopback transport).  alpha/beta agree with the single-GPU run to rounding (sums are taken in
 * another order), not bit for bit. */
typedef struct lanczos_hip_multi lanczos_hip_multi;
enum { LANCZOS_HIP_MULTI_LOOPBACK = 1 };
int lanczos_hip_multi_create(lanczos_hip_multi **out, int ngpu, const int *devices, int64_t n, int64_t nnz,
                             const int64_t *row_head_indexes, const int64_t *column_index, const double *values,
                             int max_iter, const lanczos_hip_options *opt, int flags);
void lanczos_hip_multi_destroy(lanczos_hip_multi *m);
/* v0_host: the whole start vector (n doubles), normalised over all slabs; _seed draws it as
 * lanczos_hip_set_start_seed does, so the run equals the single-GPU run of the same seed */
int lanczos_hip_multi_set_start(lanczos_hip_multi *m, const double *v0_host);
int lanczos_hip_multi_set_start_seed(lanczos_hip_multi *m, uint64_t seed);
/* `steps` more iterations, then alpha/beta of those iterations (as lanczos_hip_run) */
int lanczos_hip_multi_run(lanczos_hip_multi *m, int steps, double *alpha_out, double *beta_out);
/* enqueue one iteration and return without waiting (what lanczos_hip_crs's host loop calls) */
int lanczos_hip_multi_step(lanczos_hip_multi *m);
/* the engine that holds slab `slab` (history, profile, kernel name; owned by the handle) */
lanczos_hip_engine *lanczos_hip_multi_engine(lanczos_hip_multi *m, int slab);
int lanczos_hip_multi_info(const lanczos_hip_multi *m, int *slabs, int *halo_exchange, int *overlapped, int *iterations,
                           int *reorth_steps);
/* what the set-up of the slabs took (column remap on the host, upload, engine) and how many host threads shared it:
 * one per slab on several devices (LANCZOS_HIP_MULTI_BUILD_THREADS=0: one slab after the other) */
int lanczos_hip_multi_build_stats(const lanczos_hip_multi *m, double *seconds, int *host_threads);

/* ---- thick restart -------------------------------------------------------
 * No counterpart in the reference, which keeps every vector of a run and stops at max_iter
 * (src/lanczos.c:58-61, 193).  lanczos_hip_restart replaces the basis v_0..v_k (k iterations
 * done) by y_i = sum_l coef_host[i * k + l] v_l, i < keep, followed by v_k, and sets the
 * iteration count to keep; the next step subtracts last_coupling * y_{keep-1} in place of
 * beta * v_{k-1}.  Needs a mode that reorthogonalises against the whole basis.
 * lanczos_hip_thick_restart drives whole cycles from a freshly set start vector: fill the basis,
 * solve the projected problem, stop when the Lanczos residual bounds of the lowest nth pairs are
 * below tol * ||A|| or after max_cycles, else keep `keep` Ritz vectors and go on.  eig_out[nth];
 * resid_out[nth], vectors_host[nth][n], cycles_out, products_out may be NULL. */
int lanczos_hip_restart(lanczos_hip_engine *e, int keep, const double *coef_host, double last_coupling);
int lanczos_hip_thick_restart(lanczos_hip_engine *e, int nth, int keep, int max_cycles, double tol,
                              double *eig_out, double *resid_out, double *vectors_host, int *cycles_out,
                              int *products_out);

/* ---- selective reorthogonalisation: the decision rule on its own --------
 * No counterpart in the reference, whose solvers reorthogonalise by fixed
 * rules (src/lanczos.c:156-167, src/lanczos_cuda.c:679-687).  The tracker
 * follows estimates of v_j . v_i through the recurrence the tridiagonal
 * entries imply.  Host arithmetic only.  lanczos_hip_run drives one itself
 * under LANCZOS_HIP_REORTH_SELECTIVE; row-partitioned runs, whose ranks all
 * hold the same alpha and beta, drive one per rank and reach the same
 * decisions.  `step` is called once per iteration j = 0, 1, ... with alpha_j
 * and the norm beta_j of the vector under construction after the three-term
 * recurrence; it returns 1 when that vector must now be reorthogonalised
 * against v_0..v_j (the caller then reports the new norm with `set_beta`),
 * 0 when not, and -1 on a call out of order. */
typedef struct lanczos_hip_omega lanczos_hip_omega;
lanczos_hip_omega *lanczos_hip_omega_create(int max_steps);
void lanczos_hip_omega_destroy(lanczos_hip_omega *o);
void lanczos_hip_omega_reset(lanczos_hip_omega *o);
int lanczos_hip_omega_step(lanczos_hip_omega *o, int j, double alpha_j, double beta_j, double *estimate);
void lanczos_hip_omega_set_beta(lanczos_hip_omega *o, int j, double beta_j);
int lanczos_hip_omega_reorths(const lanczos_hip_omega *o);
size_t lanczos_hip_omega_state_doubles(const lanczos_hip_omega *o);
void lanczos_hip_omega_state_save(const lanczos_hip_omega *o, double *out);
int lanczos_hip_omega_state_load(lanczos_hip_omega *o, const double *in);
/* a state saved by a tracker created for another number of steps, laid out for this one */
int lanczos_hip_omega_state_fit(const lanczos_hip_omega *o, const double *in, size_t in_doubles, double *out);
/* steps of this engine's current run that reorthogonalised (every step but the
 * first under the fixed rules) */
int lanczos_hip_reorth_steps(const lanczos_hip_engine *e);

/* ---- block variant (block width 16, full reorthogonalisation) -------------
 * No counterpart in the reference, whose solvers iterate on one vector
 * (src/lanczos.c:104-191).  Sixteen vectors per step turn the
 * reorthogonalisation against the Krylov basis (src/lanczos.c:156-167) into a
 * tall-skinny dense contraction for the fp64 matrix cores and let the sparse
 * product gather whole 128-byte rows.  Blocks are n x 16, row-major.  T is
 * block tridiagonal: A_j (16 x 16, symmetric) on the diagonal, B_j (upper
 * triangular) below it, both row-major. */
typedef struct lanczos_hip_block lanczos_hip_block;
enum {
    LANCZOS_HIP_BK_SPMM = 0,
    LANCZOS_HIP_BK_GRAM = 1,
    LANCZOS_HIP_BK_UPDATE = 2,
    LANCZOS_HIP_BK_SCALE = 3,
    LANCZOS_HIP_BK_SMALL = 4,
    LANCZOS_HIP_BK_COUNT = 5
};
/* matrix arguments as lanczos_hip_create; room for max_steps block steps */
int lanczos_hip_block_create(lanczos_hip_block **out, int64_t n, int64_t nnz, const int64_t *row_head_indexes,
                             const int64_t *column_index, const double *values, int max_steps,
                             const lanczos_hip_options *opt);
void lanczos_hip_block_destroy(lanczos_hip_block *b);
/* start block: n x 16 row-major host array (orthonormalised on the device), or N(0,1) entries
 * from lanczos_hip_gaussian_fill(seed, 0, 16 n, .) */
int lanczos_hip_block_set_start(lanczos_hip_block *b, const double *x0_host);
int lanczos_hip_block_set_start_seed(lanczos_hip_block *b, uint64_t seed);
int lanczos_hip_block_run(lanczos_hip_block *b, int steps);
int lanczos_hip_block_steps(const lanczos_hip_block *b);
/* a_out, b_out: [steps][16][16] */
int lanczos_hip_block_tmatrix(lanczos_hip_block *b, int steps, double *a_out, double *b_out);
/* lowest nth eigenvalues of the leading 16 steps x 16 steps part of T */
int lanczos_hip_block_ritz(lanczos_hip_block *b, int steps, int nth, double *eig_out);
/* test hooks on the resident basis (slot t = block Q_t; slots above the current step are free) */
int lanczos_hip_block_get_block(lanczos_hip_block *b, int t, double *out_host);
int lanczos_hip_block_set_block(lanczos_hip_block *b, int t, const double *in_host);
int lanczos_hip_block_spmm(lanczos_hip_block *b, int t_in, int t_out);            /* Q_out = A Q_in */
int lanczos_hip_block_project(lanczos_hip_block *b, int t0, int t1, int t_w,      /* H = Q^T W; W -= Q H */
                              double *h_out_host);
int lanczos_hip_block_orthonormalise(lanczos_hip_block *b, int t, double *factor_out_host);
int lanczos_hip_block_profile(const lanczos_hip_block *b, int cls, double *ms, int64_t *launches, double *bytes);
void lanczos_hip_block_profile_reset(lanczos_hip_block *b);
/*
 * The block variant on row slabs, one host process over `ngpu` GPUs (no counterpart in the reference, which has
 * neither; it must agree with lanczos_hip_block_* on one GPU, to rounding: sums over the rows are taken slab by
 * slab).  The whole matrix is passed; slab p takes rows [p n / ngpu, (p + 1) n / ngpu) with its columns renumbered
 * into a table of X that an all-gather fills every block step (n x 16 doubles), and the 16 x 16 matrices the one-GPU
 * engine sums over its rows are all-reduced over the slabs (four reductions per block step).  `flags`:
 * LANCZOS_HIP_MULTI_LOOPBACK puts every slab on one device, copies and a summing kernel in place of RCCL.
 */
typedef struct lanczos_hip_block_multi lanczos_hip_block_multi;
int lanczos_hip_block_multi_create(lanczos_hip_block_multi **out, int ngpu, const int *devices, int64_t n, int64_t nnz,
                                   const int64_t *row_head_indexes, const int64_t *column_index, const double *values,
                                   int max_steps, const lanczos_hip_options *opt, int flags);
void lanczos_hip_block_multi_destroy(lanczos_hip_block_multi *m);
int lanczos_hip_block_multi_set_start(lanczos_hip_block_multi *m, const double *x0_host);   /* n x 16, row-major */
int lanczos_hip_block_multi_set_start_seed(lanczos_hip_block_multi *m, uint64_t seed);      /* as lanczos_hip_block_set_start_seed */
int lanczos_hip_block_multi_run(lanczos_hip_block_multi *m, int steps);
int lanczos_hip_block_multi_steps(const lanczos_hip_block_multi *m);
int lanczos_hip_block_multi_slabs(const lanczos_hip_block_multi *m);
int lanczos_hip_block_multi_tmatrix(lanczos_hip_block_multi *m, int steps, double *a_out, double *b_out);
int lanczos_hip_block_multi_ritz(lanczos_hip_block_multi *m, int steps, int nth, double *eig_out);
int lanczos_hip_block_multi_get_block(lanczos_hip_block_multi *m, int t, double *out_host); /* block Q_t, n x 16 */
/* lowest nth eigenvalues of a dense symmetric n x n matrix (row-major, destroyed) */
int lanczos_hip_sym_lowest_eigs(double *a, int n, int nth, double *eig);
/* all eigenpairs of a small dense symmetric matrix (cyclic Jacobi): eig ascending, row i of vec
 * (n x n) the unit eigenvector of eig[i]; a is destroyed */
int lanczos_hip_sym_eig_jacobi(double *a, int n, double *eig, double *vec);

#ifdef __cplusplus
}
#endif
#endif /* LANCZOS_HIP_H */
