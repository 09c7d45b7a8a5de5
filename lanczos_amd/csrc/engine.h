// Internal C++ view of the engine behind include/lanczos_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <exception>
#include <new>
#include <vector>

#include "../../include/lanczos_hip.h"

namespace lz {

constexpr int kMaxDotsSegment = 1024;  // basis vectors per k_dots launch (32 KiB of LDS)

int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *last_error();

struct Engine;
// Which row slab of a several-GPU handle the calling thread has made current (-1: none).  HIP wants
// the device of a stream current when an event is created for it or recorded on it; with every slab
// on ONE device (the loopback transport) HIP cannot tell a call made under the wrong slab from a
// right one, this number can.  LANCZOS_HIP_CHECK_AFFINITY=1 turns the comparison on: an engine of a
// several-GPU handle then refuses calls made while another slab is current.
extern thread_local int tl_current_slab;
int check_affinity(const Engine *e, const char *where);

// The body of an extern "C" entry point: no C++ exception (a failed host allocation above all) may
// cross the C boundary; it becomes the usual message + EXIT_FAILURE.
template <class F>
int guarded(const char *what, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail("%s: out of host memory", what);
    } catch (const std::exception &ex) {
        return fail("%s: %s", what, ex.what());
    } catch (...) {
        return fail("%s: unknown C++ exception", what);
    }
}

// What the plane walk needs to apply the deferred-normalisation recurrence (kernels.hip.h: k_recur_lazy) in its
// epilogue: the vector before x, the iteration, the partial sums that hold x . A x and the previous ||u||^2, and where
// alpha_k and beta_{k-1} are recorded.
struct DictRecur {
    const double *uprev;
    int k;
    const double *dot_parts;
    int dot_count;
    const double *norm_parts;
    int norm_count;
    double *alpha_hist, *beta_hist, *host_hist;
};

struct Engine {
    // geometry
    int64_t n = 0;         // local rows
    int64_t n_ext = 0;     // length of the SpMV input vector
    int64_t n_global = 0;
    int64_t row_begin = 0;
    int64_t nnz = 0;
    int64_t ldv = 0;       // doubles between consecutive basis vectors (zero padded)
    int64_t npad = 0;      // n rounded up to whole vector tiles: what the vector kernels sweep
    bool own_V = true;     // false: the caller's memory (options.basis), with room for remote entries behind npad
    bool slab = false;     // row-partitioned engine (gather buffer or caller-owned basis)
    int max_iter = 0;
    int nvec = 0;          // resident basis vectors (max_iter, or a ring of 3)
    int reorth = 0;
    int device = -1;
    int affinity_id = -1;  // slab number inside a several-GPU handle (-1: an engine of its own)
    int cus = 0;
    int max_blocks = 0;    // 8 workgroups per CU, multiple of 8
    // what the matrix upload took (host arrays page-locked around the copies unless LANCZOS_HIP_PIN_UPLOAD=0)
    double upload_sec = 0.0, upload_bytes = 0.0;
    double create_sec[4] = {0.0, 0.0, 0.0, 0.0};  // device set-up, upload, choice of the SpMV form (with its tables), state
    bool upload_pinned = false;
    char upload_pin_note[48] = "";  // "ok", "off", or the error the registration was refused with
    // device memory
    void *d_rp = nullptr;  // int32 or int64 row offsets
    bool rp64 = false;
    int32_t *d_col = nullptr;
    double *d_val = nullptr;
    double *d_V = nullptr;
    double *x_ext = nullptr;  // caller-owned gather buffer (row-partitioned use)
    double *d_partials = nullptr;    // per-workgroup partials of v_k . w left by the SpMV
    double *d_norm_parts = nullptr;  // per-workgroup partials of w.w, two halves used alternately
    int alpha_count = 0, norm_count = 0;
    bool lazy = false;               // deferred normalisation (three-vector mode on one GPU)
    int lazy_flushed_k = 0;
    bool lazy_slab = false;          // deferred normalisation on a row slab (three-vector mode, phase calls)
    int spec_spmv_k = -1;            // iteration whose product was already launched ahead (selective mode)
    double *d_hpart = nullptr;
    int hstride = 0;
    double *d_red = nullptr;   // [max_iter] reduction results (single-GPU runs)
    double *d_scal = nullptr;  // [1] = beta of the previous iteration
    double *d_alpha = nullptr, *d_beta = nullptr;
    int hist_cap = 0;
    double *h_hist = nullptr;  // pinned mirror {alpha_k, beta_k} for the pipelined host loop
    // selective reorthogonalisation: the host decides per step from alpha and the
    // norm after the recurrence, which one small kernel mirrors into pinned memory
    lanczos_hip_omega *omega = nullptr;
    double *h_sel = nullptr;
    hipEvent_t sel_event = nullptr;
    int reorth_steps = 0;
    // stream
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // SpMV variant
    int spmv_kind = 0;
    int group = 8;   // lanes per row of the vector kernel
    int log2r = 6;   // log2(rows per wavefront group) of the pipe kernel
    char spmv_name[32] = "";
    int *d_tile_rows = nullptr;        // merge SpMV: first and last row of every tile of 512 entries
    int *d_carry_row = nullptr;        // merge SpMV: rows shared between tiles, two slots per tile
    double *d_carry_val = nullptr;
    long long *d_xcd_split = nullptr;  // [9] chunk ranges of the pipe kernel, equal entries per XCD
    long long *d_xcd_split_in = nullptr;  // the same for the rows [split_in_lo, split_in_hi) of a split product
    int64_t split_in_lo = -1, split_in_hi = -1;
    // tiled form of the matrix (LANCZOS_HIP_SPMV_PANEL): entries grouped by (row block, column panel)
    uint16_t *d_trow = nullptr;      // row in its 4096-row block (unpacked form)
    uint32_t *d_tcol = nullptr;      // column; packed form: row in block << 20 | column - d_tbase[chunk]
    uint32_t *d_tbase = nullptr;     // packed form: smallest column of each chunk
    uint32_t *d_tfar_entry = nullptr;  // packed form: CRS positions of the entries whose column is too far from
    int32_t *d_tfar_row = nullptr;     //   their chunk's first (sorted), and their rows; added by a kernel of their own
    int tile_nfar = 0;
    bool tile_packed = false;
    double *d_pval = nullptr;
    uint32_t *d_tblock_ptr = nullptr;  // [row blocks + 1] first stream position of each block
    void *d_tpieces = nullptr, *d_tsplits = nullptr;  // work items; blocks cut into several of them
    double *d_tside = nullptr;       // [tile_nslots][4096] sums of the pieces of split blocks
    double *d_tdots = nullptr;       // [pieces + 256] xdot . y of each piece, of the combining workgroups
    int *d_tcounter = nullptr;       // next piece to draw
    int tile_panels = 0, tile_bits = 16, panel_bits = 16;
    int tile_npieces = 0, tile_nsplit = 0, tile_nslots = 0;
    int tile_rowbits = 12;           // log2 of the rows of a block: 12, 13 or 14
    bool tile_even = false;          // every piece carries about the same work
    // dictionary form of the matrix (LANCZOS_HIP_SPMV_DICT): one pattern number per row
    uint16_t *d_pattern = nullptr;
    int32_t *d_dlen = nullptr, *d_doff = nullptr;  // [patterns], [patterns][dict_stride]
    double *d_dval = nullptr;
    int dict_patterns = 0, dict_stride = 0, dict_halo = 0;
    bool dict_values = true;   // false: the dictionary holds column offsets only (values read from the CRS array)
    bool pdict = false;        // the row-wise CRS SpMV runs as a plane walk over such a dictionary
    int dict_blocks = 0;       // grid of the ahead-of-time form: one round of resident workgroups
    int dict_segs = 0, dict_seg_len = 0;   // plane-walking form: segments of planes per column of steps
    int dict_span = 0, dict_cache = 0;     // plane-walking form: rows per step (0: not chosen yet) and its register form
    int32_t dict_plane_f = 0;  // plane-walking form: rows per plane (0: not a grid of whole-step planes)
    int32_t dict_plane_a = 0, dict_plane_b = 0;  // a slab's offsets to the remote planes (INT32_MIN: none)
    int dict_far_count = -1;   // distinct column offsets beyond the LDS window (-1: more than two)
    int32_t dict_far[2] = {0, 0};
    // operator supplied by the caller as a host function (the reference's Matvec_General plugin,
    // src/lanczos.h:5): the product goes through two pinned host vectors, everything else stays here
    lanczos_hip_matvec_fn *ext_matvec = nullptr;
    const void *ext_mat = nullptr;
    double *h_ext_x = nullptr, *h_ext_w = nullptr;
    int set_operator(lanczos_hip_matvec_fn *fn, const void *mat);
    int launch_product(int kk);  // w = A v_kk into slot kk + 1 and the partials of v_kk . w
    // iteration state
    int k = 0;
    bool started = false;
    int ckpt_host_k = -1;  // iterations whose scalars a pipelined host loop has consumed (-1: all of k); travels
                           // with a checkpoint so a resumed loop knows which lines were already printed
    // profiling
    bool profile = false;
    unsigned prof_mask = ~0u;  // kernel classes whose launches are bracketed by events (bytes and launches are counted for all)
    double prof_ms[LANCZOS_HIP_K_COUNT] = {0};
    int64_t prof_launches[LANCZOS_HIP_K_COUNT] = {0};
    double prof_bytes[LANCZOS_HIP_K_COUNT] = {0};
    struct Pending { int cls; hipEvent_t start, stop; };
    std::vector<Pending> prof_pending;
    std::vector<hipEvent_t> event_pool;

    static int create(Engine **out, int64_t n, int64_t nnz, const int64_t *rp, const int64_t *ci,
                      const double *va, int max_iter, const lanczos_hip_options *opt);
    ~Engine();
    static int create_from_coo(Engine **out, const Mat_Coo *coo, int swap_row_column, int max_iter,
                               const lanczos_hip_options *opt);
    int setup_device(const lanczos_hip_options &opt, int64_t n, int64_t n_ext, int64_t nnz, int max_iter);
    int upload_crs(const int64_t *rp, const int64_t *ci, const double *va);
    int build_from_coo(const Coo *coo, int64_t length, int swap_row_column);
    int choose_spmv(int requested);
    int build_dict(bool with_values = true);
    bool pdict_usable() const;
    int launch_spmv_pdict(const double *x, const double *xdot, double *y, int *nparts);
    int dict_form() const;
    void free_dict();
    int launch_spmv_dict(const double *x, const double *xdot, double *y, int *nparts, int64_t dot_lo = 0, int64_t dot_hi = -1,
                         int mode = 0, const void *recur = nullptr, double *parts_out = nullptr);
    int launch_walk1(int kk);        // an iteration with deferred normalisation as two plane walks (dict.hip)
    int launch_walk2();
    bool spec_walk = false;          // the product launched ahead was a first walk (writes nothing)
    bool two_pass_ok();
    bool walk_pending = false;       // a row slab's iteration in progress runs as two plane walks
    int build_panels(int bits);
    template <int RB, int T>
    int build_panels_shape(int bits, bool even);
    void free_panels();
    int launch_spmv_panels(const double *x, const double *xdot, double *y, int *nparts);
    int alloc_state();

    // basis slot of Lanczos vector j (a ring of 3 when nothing older is needed)
    double *vec(int j) const { return d_V + (size_t)(reorth == LANCZOS_HIP_REORTH_NONE ? j % 3 : j) * (size_t)ldv; }
    int vec_blocks() const;
    int recur_blocks() const;
    int tile_blocks() const;
    double spmv_bytes() const;

    int set_start(const double *v0, bool normalise, double scale);
    int check_can_step();
    void reorth_range(int *l0, int *l1) const;
    int launch_spmv(const double *x, const double *xdot, double *y, int *nparts);
    double *norm_parts(int j) const { return d_norm_parts + (size_t)(j & 1) * (size_t)max_blocks; }
    int launch_finalize(const double *parts, int nparts, double *out);
    int launch_recur(const double *alpha_parts, int alpha_cnt);
    int launch_scale(const double *parts, int cnt);
    int flush_lazy();
    int launch_dots(int l0, int l1, double *red_out);
    int launch_recur_dots(const double *alpha_parts, int alpha_cnt, int l0, int l1, double *red_out);
    int launch_update(int l0, int l1, const double *h, int *nparts);
    bool red_has_norm = false;  // phase_recur left w.w behind the coefficients in the caller's buffer
    int phase_spmv(double *red);
    int phase_recur(double *red, int *n_red, int *is_norm);
    int phase_dots(double *red, int *n_red);
    int phase_update(double *red);
    int phase_scale(double *red);
    int phase_spmv_begin(int64_t b_lo, int64_t b_hi);
    int phase_spmv_end(double *red);
    bool can_split_product() const;
    int finalize_spmv(double *red);
    int launch_spmv_edges(int64_t lo, int64_t hi, double *parts_out, int *nparts);
    int64_t pending_lo = -1, pending_hi = -1;  // rows [0, lo) and [hi, n) wait for phase_spmv_end
    int phase_recur_lazy(double *red);
    int phase_lazy_norm(double *red);
    int phase_lazy_close(double *red);
    int fuse_limit() const;
    bool nt_stores() const;             // vectors too long for the caches: the vector kernels write past them
    void (*recur_lazy_kernel() const)(double2 *, const double2 *, const double2 *, int64_t, int, const double *, int, const double *, int,
                                      double *, double *, double *, double *);
    int enqueue_iteration();
    int combine(int m, int count, const double *coef_host, double *out_dev, int64_t out_ld);
    int restart(int p, const double *coef_host, double last_coupling);
    int selective_tail();
    int selective_lazy_tail(bool walks);
    int run(int steps, double *alpha_out, double *beta_out);
    int spmv_host(const double *x_host, double *y_host);
    int spmv_noalloc(const double *x_dev, double *y_dev, double *dot_dev);
    int enable_host_history();
    int make_current();                 // this engine's device becomes the thread's current one
    int record_event(hipEvent_t ev);    // on this engine's stream; its device must be current

    hipEvent_t take_event();
    void harvest_profile();
    void harvest_completed();
};

// Collectives over the row slabs that ONE host process drives (multi.hip): RCCL communicators over the slabs' devices
// (ncclCommInitAll, calls grouped over the devices), or, with every slab on one device and one stream, copies and a
// summing kernel in their place.  Shared by the single-vector and the block form of the row-partitioned iteration.
struct SlabComm;
int slab_comm_create(SlabComm **out, int slabs, const int *devices, bool loopback);
void slab_comm_destroy(SlabComm *c);
// bufs[p][0..count) <- sum over the slabs, in place, ordered behind what streams[p] holds; slabs added in ascending order
// by the loopback form
int slab_all_reduce(SlabComm *c, double *const *bufs, size_t count, const hipStream_t *streams);
// tables[p] + q * stride is the place of slab q's piece (count <= stride doubles) in slab p's table; slab p's own piece is
// there already, the others arrive
int slab_all_gather(SlabComm *c, double *const *tables, size_t stride, size_t count, const hipStream_t *streams);

}  // namespace lz
