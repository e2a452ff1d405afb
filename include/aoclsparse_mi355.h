/*
 * aoclsparse_mi355.h -- MI355X-specific additions next to the aoclsparse_* drop-in ABI.
 *
 * Nothing here exists in the reference (a CPU library has no device boundary).  Two groups:
 *   1. aoclsparse_mi355_*: runtime control and introspection of a handle's device plan.
 *   2. mi355_*: the thin HIP C-ABI the host layer itself calls -- plain device pointers,
 *      sizes and a hipStream_t (passed as void*), no handles, asynchronous.  These are the
 *      entry points a benchmark or a solver that keeps x/y resident in HBM binds directly.
 */
#ifndef AOCLSPARSE_MI355_H_
#define AOCLSPARSE_MI355_H_

#include "aoclsparse.h"

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime control ------------------------------------------------------------------ */
typedef enum aoclsparse_mi355_pointer_mode_
{
    aoclsparse_mi355_pointer_auto   = 0, /* classify x/y/B/C per call (hipPointerGetAttributes) */
    aoclsparse_mi355_pointer_host   = 1, /* reference semantics: host pointers, staged, synchronous */
    aoclsparse_mi355_pointer_device = 2 /* device pointers, stream-ordered, no query */
} aoclsparse_mi355_pointer_mode;

DLL_PUBLIC aoclsparse_status aoclsparse_mi355_set_pointer_mode(aoclsparse_mi355_pointer_mode mode);
/* Plan options: ONE hook for the tests and measurements that must reach a kernel the automatic choice would not pick for
 * their (small) input.  Process-wide, read when a handle's plan is built (so: set before aoclsparse_optimize / the first
 * product, reset afterwards).  Everything else the library chooses by itself; the selection switches of rounds 1-3
 * (environment variables) are gone -- their measurements are under profiles/.
 *   spmv_kernel  0 automatic (default: CSR-Adaptive, merge-path once the longest row spans 16 LDS tiles = 8,192 entries for
 *                matrices below 4 M non-zeros), 1 CSR-Adaptive, 2 merge-path whenever it can serve the request.  Since round 5 both
 *                CSR kernels sum rows of >= spmv_info.tree_min (32) entries with a wavefront tree in their automatic mode: such
 *                rows are within the stated bound, no longer the reference's bits -- spmv_strict = 1 or a pinned kid restores them
 *   sell         -1 automatic (default: SELL-64 copy for an mv hint when its padding is <= 1.35 x), 0 never, 1 always
 *   spmv_strict  0 (default): without a pinned kid, CSR-Adaptive sums a row of >= spmv_info.tree_min entries with a wavefront tree
 *                (componentwise bound (2 ceil(log2 n) + 4) eps sum|a||x|; shorter rows are the reference's chain, bit for bit);
 *                1: every row of every product in the reference's order, as a pinned kid does -- bit-exact everywhere, at the
 *                price of one lane's serial chain per long row.  Read at every product (not a plan option).
 *   alternate_sweeps  1 (default): consecutive products of a handle (SpMV on the SELL-64 copy or on the row blocks, row-major
 *                csrmm) walk the matrix in alternating directions, so the end of one sweep -- what the 256 MB Infinity Cache still
 *                holds -- is where the next one starts; 0: always ascending.  Same bits either way (a row's chain does not depend
 *                on when the row is visited); a measurement of HBM throughput sets 0 or flushes the cache between products.
 *                Read at every product.
 *   trsv_chunks  -1 (default): the two-level TRSV schedule (chunks of consecutive blocks, hand-offs inside a chunk through LDS;
 *                schedule 5) is built when the plan-time model of the triangle's DAG predicts a gain over the lane-per-block
 *                schedule (deep, narrow DAGs: a mesh numbered line by line); 0 never; 1 whenever the triangle has the shape the
 *                kernel serves.  Read when a TRSV plan is built.  Same bits on every schedule. */
typedef enum aoclsparse_mi355_option_
{
    aoclsparse_mi355_option_spmv_kernel = 0,
    aoclsparse_mi355_option_sell        = 1,
    aoclsparse_mi355_option_spmv_strict = 2,
    aoclsparse_mi355_option_alternate_sweeps = 3,
    aoclsparse_mi355_option_trsv_chunks = 4,
    aoclsparse_mi355_option_count       = 5
} aoclsparse_mi355_option;
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_set_option(aoclsparse_mi355_option option, aoclsparse_int value);
/* aoclsparse_?csrmm with beta == 0.  Default (0): C is read and multiplied by zero, exactly as every kernel of the reference
 * does (level3/aoclsparse_csrmm.hpp:83,129; aoclsparse_csrmm_kt.cpp:176-191,246), so a NaN / Inf already in C propagates.
 * 1: C is overwritten without being read (BLAS convention; identical results for every finite C, including the sign of
 * exact zeros; saves the read of C: ~15 % at 256 columns).  Also AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=1. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_set_csrmm_beta0_overwrite(int overwrite);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_set_stream(void *hip_stream);
DLL_PUBLIC void             *aoclsparse_mi355_get_stream(void);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_synchronize(void);
/* device ordinal in use, CU count and name; status internal_error when no HIP device exists */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_device_info(aoclsparse_int *device,
                                                          aoclsparse_int *compute_units,
                                                          char            name[256]);
/* Path of the HIP runtime (libamdhip64) this library's calls are bound to, as the dynamic loader resolved them.  A process can
 * hold two copies (a framework that ships its own next to /opt/rocm's): streams and events are objects of ONE runtime, so a
 * stream handed to aoclsparse_mi355_set_stream must come from the copy named here.  Writes at most `capacity` bytes including
 * the terminating zero; invalid_pointer / invalid_size for a null or empty buffer, internal_error when the loader cannot tell. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_hip_runtime_path(char *path, size_t capacity);
/* hipEvent pair on the library's stream: start, run work, stop -> elapsed milliseconds */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_timer_start(void);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_timer_stop(float *elapsed_ms);
/* per-iteration timing (the reference harness reports min / quartiles / max of its iterations,
 * tests/include/aoclsparse_stats.hpp:41-129): mark() records one event on the library's stream (at most 65536 between
 * two laps() calls); laps() waits for the last mark, writes min(*count, capacity) elapsed times between consecutive
 * marks (milliseconds) and resets the ring. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_timer_mark(void);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_timer_laps(float *laps_ms, aoclsparse_int capacity, aoclsparse_int *count);

/* ---- column shards of csrmm: one process per GPU ------------------------------------------ */
/* The reference splits B's columns over its worker threads inside the library
 * (library/src/level3/aoclsparse_csrmm_kt.cpp:68-82: start = n*t/T rounded up to a multiple of 4, capped at n).
 * column_shard applies that rule with (world, rank) in place of (threads, thread id); ?csrmm_shard computes the slab
 * C[:, j0:j1) of rank `rank` from the FULL B / C arrays (same arguments as aoclsparse_?csrmm otherwise).  There is no
 * communication on the data path: a rank needs A and its own columns of B only. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_column_shard(aoclsparse_int n, aoclsparse_int world, aoclsparse_int rank,
                                                           aoclsparse_int *j0, aoclsparse_int *j1);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_dcsrmm_shard(aoclsparse_operation op, const double alpha,
                                                           const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                           aoclsparse_order order, const double *B, aoclsparse_int n,
                                                           aoclsparse_int ldb, const double beta, double *C,
                                                           aoclsparse_int ldc, aoclsparse_int world, aoclsparse_int rank);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_scsrmm_shard(aoclsparse_operation op, const float alpha,
                                                           const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                           aoclsparse_order order, const float *B, aoclsparse_int n,
                                                           aoclsparse_int ldb, const float beta, float *C,
                                                           aoclsparse_int ldc, aoclsparse_int world, aoclsparse_int rank);

/* ---- one process, several GPUs (round 3) ----------------------------------------------------
 * C = alpha * op(A) * B + beta * C with the columns of B and C split over `ndev` devices by the rule above; the GPU takes
 * the place of the reference's worker thread (aoclsparse_csrmm_kt.cpp:68-82).  devices[0] must be the library's own device
 * (AOCLSPARSE_MI355_DEVICE, else the caller's current device); devices == NULL means that device and the next ndev - 1
 * ordinals.  The first call builds a replica of the handle on every other device (same host arrays, hints copied,
 * aoclsparse_optimize there), later calls reuse it; ?set_value / ?update_values / aoclsparse_mi355_invalidate drop the
 * replicas.  No collective on the data path.  Returns when every device has finished.
 *   ?csrmm_multi        B, C = the FULL operands in HOST memory (the reference's calling convention): each device stages
 *                       and returns its own slab over its own PCIe link.  AOCLSPARSE_MI355_DEVICES=N makes plain
 *                       aoclsparse_?csrmm take this path for host operands.
 *   dcsrmm_multi_slabs  B_slabs[i] / C_slabs[i] = device i's slab, resident in ITS memory (column-major: columns
 *                       [j0_i, j1_i) with leading dimension ldb / ldc; row-major: rows of j1_i - j0_i columns); n is the
 *                       TOTAL column count.  The path for operands that already live on the GPUs. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_dcsrmm_multi(aoclsparse_operation op, const double alpha,
                                                           const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                           aoclsparse_order order, const double *B, aoclsparse_int n,
                                                           aoclsparse_int ldb, const double beta, double *C,
                                                           aoclsparse_int ldc, aoclsparse_int ndev,
                                                           const aoclsparse_int *devices);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_scsrmm_multi(aoclsparse_operation op, const float alpha,
                                                           const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                           aoclsparse_order order, const float *B, aoclsparse_int n,
                                                           aoclsparse_int ldb, const float beta, float *C,
                                                           aoclsparse_int ldc, aoclsparse_int ndev,
                                                           const aoclsparse_int *devices);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_dcsrmm_multi_slabs(aoclsparse_operation op, const double alpha,
                                                                 const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                                 aoclsparse_order order, const double *const *B_slabs,
                                                                 aoclsparse_int n, aoclsparse_int ldb, const double beta,
                                                                 double *const *C_slabs, aoclsparse_int ldc,
                                                                 aoclsparse_int ndev, const aoclsparse_int *devices);

/* Operands of the two calls above must be COMPLETE on every device before the call (work in flight on a device's null stream
 * is ordered before the product: the slot streams are blocking streams; work on other non-blocking streams is not).
 * multi_last_ms: wall time each device spent on its share in the last multi-device call of the process (its launches and the
 * wait for its stream), ms[i] for i < min(returned count, capacity); returns the device count of that call (0: none yet). */
DLL_PUBLIC aoclsparse_int aoclsparse_mi355_multi_last_ms(float *ms, aoclsparse_int capacity);

/* replicas of the handle currently alive on other runtime slots (0 before the first multi-device call); -1 for NULL */
DLL_PUBLIC aoclsparse_int aoclsparse_mi355_replica_count(const aoclsparse_matrix A);
/* ... of which built by copying the primary handle's device format device to device (peer copy over xGMI) instead of analysing
 * the host arrays again: the case when the handle was optimized (or used) before its first multi-device call; -1 for NULL */
DLL_PUBLIC aoclsparse_int aoclsparse_mi355_replicas_cloned(const aoclsparse_matrix A);

/* ---- one process per GPU: shipping the ANALYSED device state of a handle (round 4) ----------------------------------------
 * The column shards of csrmm need A on every rank.  What travels is the device format: the CSR arrays in HBM and every csrmm
 * plan (row blocks, row groups, row runs, column windows, row pairs, the blocked-ELL copy), so that the receiving ranks do no
 * analysis (the reference has nothing to ship: its column split is a thread split inside one call,
 * library/src/level3/aoclsparse_csrmm_kt.cpp:68-82, and A is shared memory).
 *   export  builds every csrmm plan of A that is still missing, then fills `state` (sizes + scalars, a POD that can be sent as
 *           bytes) and buffers[i] = device pointer of buffer i (owned by A, valid until A is modified or destroyed; NULL where
 *           state->bytes[i] == 0).
 *   adopt   creates a NEW handle (*R, to be destroyed by the caller) from `state` and device copies of the buffers that the caller
 *           has placed in THIS process's device memory (received over whatever wire it uses): the buffers are copied device to
 *           device, the CSR arrays are copied back once to give the handle its host view, no analysis runs.  Every
 *           aoclsparse_* call works on the new handle; it carries an optimized mm hint. */
#define AOCLSPARSE_MI355_MM_STATE_BUFFERS 14
#define AOCLSPARSE_MI355_MM_STATE_SCALARS 40
typedef struct aoclsparse_mi355_mm_state_
{
    long long scalars[AOCLSPARSE_MI355_MM_STATE_SCALARS];
    long long bytes[AOCLSPARSE_MI355_MM_STATE_BUFFERS];
} aoclsparse_mi355_mm_state;
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_mm_state_export(aoclsparse_matrix A, aoclsparse_mi355_mm_state *state,
                                                              const void *buffers[AOCLSPARSE_MI355_MM_STATE_BUFFERS]);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_mm_state_adopt(aoclsparse_matrix *R, const aoclsparse_mi355_mm_state *state,
                                                             const void *const buffers[AOCLSPARSE_MI355_MM_STATE_BUFFERS]);

/* ---- the same over a communicator the LIBRARY owns: RCCL (librccl.so, loaded with dlopen at the first call) ---------------
 * One communicator per process, on the library's device.  Rank 0 calls comm_unique_id and hands the 128 bytes to the other
 * ranks by any means (MPI, a file, torch.distributed's store); every rank then calls comm_init -- a collective, like all calls
 * below.  Collectives are enqueued on the library's stream (aoclsparse_mi355_get_stream) and must be issued by one thread at a
 * time, in the same order on every rank.
 *   comm_broadcast_matrix  rank `root` passes its handle, every other rank passes *A == NULL and receives a new handle (to be
 *                          destroyed by the caller) holding root's device CSR and csrmm plans: ncclBroadcast of a header, then
 *                          one grouped ncclBroadcast per buffer straight into the new handle's device buffers.
 *   comm_allgather         bytes_per_rank bytes from `send` of every rank, concatenated in rank order in `recv` (column-major C
 *                          slabs of equal width: the whole C, in place when send == recv + rank * bytes_per_rank).
 *   comm_broadcast         a device buffer from `root` to everyone.
 * status not_implemented: librccl.so could not be loaded; invalid_operation: no communicator (or a second comm_init). */
typedef struct aoclsparse_mi355_comm_id_
{
    char internal[128];
} aoclsparse_mi355_comm_id;
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_comm_unique_id(aoclsparse_mi355_comm_id *id);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_comm_init(aoclsparse_int world, aoclsparse_int rank, const aoclsparse_mi355_comm_id *id);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_comm_destroy(void);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_comm_info(aoclsparse_int *world, aoclsparse_int *rank, aoclsparse_int *rccl_version);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_comm_broadcast(void *device_buffer, size_t bytes, aoclsparse_int root);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_comm_allgather(const void *send, void *recv, size_t bytes_per_rank);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_comm_broadcast_matrix(aoclsparse_matrix *A, aoclsparse_int root);

/* ---- introspection of a handle --------------------------------------------------------- */
/* idiag / iurow of the clean CSR (host arrays owned by the handle, length m, matrix base);
 * the counterpart of the reference-internal csr::idiag / csr::iurow the unit tests inspect
 * (tests/unit_tests/hint_tests.cpp:172-192).  Requires a prior optimize/trsv. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_export_diag(const aoclsparse_matrix A,
                                                          aoclsparse_int        **idiag,
                                                          aoclsparse_int        **iurow,
                                                          aoclsparse_int         *is_internal);
typedef struct aoclsparse_mi355_spmv_info_
{
    aoclsparse_int kernel; /* 0 none yet, 1 csr-adaptive stream, 2 merge-path (scalar order, no pinned kid; else 1), 3 SELL-64 (mv hint + optimize), 4 SELL-64 with one column list per run of rows that share it */
    aoclsparse_int order; /* 0 scalar chain (kid 0), 1 4-lane (kid 1/2), 2 8-lane (kid 3) */
    aoclsparse_int row_blocks; /* workgroups per launch */
    aoclsparse_int tile; /* non-zeros staged in LDS per workgroup */
    aoclsparse_int long_rows; /* rows longer than one LDS tile */
    aoclsparse_int max_row_nnz;
    aoclsparse_int device_resident; /* 1 once the CSR arrays are in HBM */
    aoclsparse_int sell_slices; /* SELL-64: 64-row slices (0 otherwise) */
    long long      stored_cells; /* SELL-64: value cells stored, padding included (kernel 4 stores fewer column cells) */
    aoclsparse_int mm_groups; /* row-major csrmm: row groups (runs of rows with one column pattern) in use, else 0 */
    aoclsparse_int mm_window_rows; /* column-major csrmm: rows per workgroup of the LDS-window kernel (banded matrices), else 0 */
    aoclsparse_int mm_bell_width; /* csrmm: 16 x 16 block slots per block row of the blocked-ELL copy (MFMA kernel), else 0 */
    aoclsparse_int mm_bell_fill_permille; /* ... and 1000 * nnz / (256 * stored blocks) */
    aoclsparse_int tree_min; /* CSR-Adaptive, scalar order, no pinned kid, spmv_strict 0: rows with at least this many entries are summed
                                by a wavefront tree (stated bound) instead of the reference's chain; 0: every row is the reference's order */
    aoclsparse_int mm_bell_xcd_chunk; /* blocked-ELL csrmm, which XCD works through which block rows: chunks of this many consecutive block
                                         rows dealt to the XCDs in turn (1 = launch order); 0: no copy, or the lattice sweep below */
    aoclsparse_int mm_bell_model_fetches_permille; /* ... 1000 * modelled fabric fetches per B block row in that order ... */
    aoclsparse_int mm_bell_model_fetches_launch_order_permille; /* ... and in launch order */
    aoclsparse_int mm_bell_lattice_line, mm_bell_lattice_lines; /* lattice sweep (block columns at offsets 1, n1, n1 n2): n1, n2; else 0 */
    aoclsparse_int mm_bell_region_a, mm_bell_region_b; /* ... the a x b block rows of the cross-section an XCD follows through the planes */
} aoclsparse_mi355_spmv_info;
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_get_spmv_info(const aoclsparse_matrix     A,
                                                            aoclsparse_operation        op,
                                                            aoclsparse_mi355_spmv_info *info);
/* The plan behind mm_bell_xcd_chunk / the lattice sweep, computed from host arrays (no device involved; what aoclsparse_optimize runs on the
 * blocked-ELL copy's block columns): bcol = nbr x width block columns, ascending per block row, empty slots (-1) last; nbc = block columns of
 * the matrix.  forced: -2 automatic, -1 the lattice sweep whenever a lattice is found, 0 launch order, c >= 1 chunks of c block rows.
 * order (room for order_capacity entries; 8 * nbr always suffices) receives order[8 p + x] = the p-th block row of XCD x, -1 past the end of
 * its list; *order_len = list positions per XCD (0: launch order, nothing written).  info = {xcd chunk (0: lattice sweep), block rows per
 * line, lines per plane, planes, region a, region b, 1000 * modelled fetches per B block row, the same in launch order}. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_plan_block_row_order(aoclsparse_int nbr, aoclsparse_int width, aoclsparse_int nbc,
                                                                   const aoclsparse_int *bcol, aoclsparse_int forced, aoclsparse_int *order,
                                                                   aoclsparse_int order_capacity, aoclsparse_int *order_len,
                                                                   aoclsparse_int info[8]);
/* number of dependency levels of the triangle a trsv with (fill, op) walks; -1 before analysis */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_get_trsv_levels(const aoclsparse_matrix A,
                                                              aoclsparse_fill_mode    fill,
                                                              aoclsparse_operation    op,
                                                              aoclsparse_int         *levels);
/* the TRSV plan of (fill, op) in numbers (zeros before analysis): what the automatic schedule will run and why */
typedef struct aoclsparse_mi355_trsv_info_
{
    aoclsparse_int levels; /* dependency levels of the rows */
    aoclsparse_int blocks, block_levels; /* blocks of chained rows (0: no block plan) and their dependency levels */
    aoclsparse_int chunks, steps, lds_slots; /* two-level schedule (0: not built): chunks of consecutive blocks, steps, LDS words of the largest chunk */
    aoclsparse_int model_chunk_us, model_block_us; /* plan-time estimates of the two-level / the lane-per-block schedule */
    aoclsparse_int schedule; /* the schedule a solve with the reference chain runs now (set_trsv_schedule included) */
    aoclsparse_int slices, slice_fan_in_permille; /* lane-per-block schedule: wavefronts (slices of <= 64 blocks of one level; 32 where
                                                     1000 * the producer slices a slice of 64 waits for, on average, exceeds 8000) */
} aoclsparse_mi355_trsv_info;
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_get_trsv_info(const aoclsparse_matrix     A,
                                                            aoclsparse_fill_mode        fill,
                                                            aoclsparse_operation        op,
                                                            aoclsparse_mi355_trsv_info *info);
/* Asynchronous (device-pointer) aoclsparse_?trsv / ?trsm calls return before the solve has run.  Should one of the
 * sync-free kernels ever give up a wait (5 s of wall time: a lost dependency, never seen in testing), it leaves x untouched
 * and sets a word owned by the handle.  Call this AFTER synchronising your stream: internal_error if a solve of THIS handle
 * expired since the last query (the word is cleared), success otherwise.  Host-pointer solves report it themselves. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_trsv_status(aoclsparse_matrix A);
/* TRSV / TRSM schedule.  -1 (default): chosen from the plan -- one launch per level for <= 32 levels, else a single
 * sync-free launch (a lane per block of chained rows where the triangle has them, else a level slice per wavefront or a
 * lane per position).  0: one launch per level, 1: hybrid (narrow level runs inside one workgroup), 2: sync-free, lane per
 * position, 3: sync-free, level slice per wavefront, 4: sync-free, lane per block.  Every schedule returns the same bits;
 * the `kid` of aoclsparse_?trsv_kid selects the arithmetic (0: ref_trsv_*; 1/2: 256-bit KT kernels; 3: 512-bit KT kernels), as
 * in the reference (library/src/level2/aoclsparse_trsv.cpp:321-353).  Process-wide; for tests and measurements. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_set_trsv_schedule(aoclsparse_int schedule);
/* drop every device-side copy/plan of the handle (call after mutating the aliased arrays) */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_invalidate(aoclsparse_matrix A);
/* free the library's staging buffers in HBM (process-wide: the primary device's and those of every device a multi-device call has
 * used; each device's stream is synchronised first): the grow-only scratch that host-pointer calls (?csrmv, ?mv, ?trsv, the ELL
 * family) and sp2m (operands that are not a handle's own arrays, bin lists, the global slabs of very long rows) keep between calls
 * so that the next call allocates nothing.  The next call that needs one allocates it again.  Returns the number of bytes freed
 * through *bytes_freed (may be NULL).  No reference counterpart (the reference has no device). */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_release_staging(size_t *bytes_freed);

/* trsv with strides AND a kernel id in one call: what the reference offers only through its C++ template
 * aoclsparse::trsv<T> (library/include/aoclsparse.hpp); include/aoclsparse.hpp forwards to these. */
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_strsv_full(aoclsparse_operation trans, float alpha, aoclsparse_matrix A,
                                                         const aoclsparse_mat_descr descr, const float *b,
                                                         aoclsparse_int incb, float *x, aoclsparse_int incx,
                                                         aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_dtrsv_full(aoclsparse_operation trans, double alpha, aoclsparse_matrix A,
                                                         const aoclsparse_mat_descr descr, const double *b,
                                                         aoclsparse_int incb, double *x, aoclsparse_int incx,
                                                         aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_ctrsv_full(aoclsparse_operation trans, aoclsparse_float_complex alpha,
                                                         aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                         const aoclsparse_float_complex *b, aoclsparse_int incb,
                                                         aoclsparse_float_complex *x, aoclsparse_int incx,
                                                         aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_mi355_ztrsv_full(aoclsparse_operation trans, aoclsparse_double_complex alpha,
                                                         aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                         const aoclsparse_double_complex *b, aoclsparse_int incb,
                                                         aoclsparse_double_complex *x, aoclsparse_int incx,
                                                         aoclsparse_int kid);

/* ---- thin HIP C-ABI: device pointers, explicit stream ---------------------------------- */
/* Row-block table for mi355_?csrmv, built on the HOST from a host row_ptr: nblocks+1 entries
 * {first row, first non-zero (0-based)}, i.e. 2*(nblocks+1) ints; blocks_host must hold
 * mi355_csrmv_plan_bound(m, nnz) ints.  tile = non-zeros staged in LDS per workgroup: 512, 1024 or 2048.
 * Returns the number of blocks, <0 on error. */
DLL_PUBLIC aoclsparse_int mi355_csrmv_plan_bound(aoclsparse_int m, aoclsparse_int nnz);
DLL_PUBLIC aoclsparse_int mi355_csrmv_plan_host(aoclsparse_int        m,
                                                aoclsparse_int        base,
                                                aoclsparse_int        tile,
                                                const aoclsparse_int *row_ptr_host,
                                                aoclsparse_int       *blocks_host);
/* y = alpha*A*x + beta*y; order: 0 scalar chain, 1 4-lane, 2 8-lane (reference kid 0 / 1,2 / 3);
 * strict != 0 keeps the reference order for rows longer than one LDS tile too; tile must be the
 * value the plan was built with; blocks is the DEVICE copy of the plan. */
DLL_PUBLIC aoclsparse_status mi355_dcsrmv(void                 *stream,
                                          aoclsparse_int        order,
                                          aoclsparse_int        strict,
                                          aoclsparse_int        tile,
                                          aoclsparse_int        base,
                                          double                alpha,
                                          aoclsparse_int        m,
                                          const double         *val,
                                          const aoclsparse_int *col,
                                          const aoclsparse_int *row_ptr,
                                          const aoclsparse_int *blocks,
                                          aoclsparse_int        nblocks,
                                          const double         *x,
                                          double                beta,
                                          double               *y);
DLL_PUBLIC aoclsparse_status mi355_scsrmv(void                 *stream,
                                          aoclsparse_int        order,
                                          aoclsparse_int        strict,
                                          aoclsparse_int        tile,
                                          aoclsparse_int        base,
                                          float                 alpha,
                                          aoclsparse_int        m,
                                          const float          *val,
                                          const aoclsparse_int *col,
                                          const aoclsparse_int *row_ptr,
                                          const aoclsparse_int *blocks,
                                          aoclsparse_int        nblocks,
                                          const float          *x,
                                          float                 beta,
                                          float                *y);
/* C = alpha*A*B + beta*C (op none), dense B/C device pointers; order as aoclsparse_order. */
DLL_PUBLIC aoclsparse_status mi355_dcsrmm(void                 *stream,
                                          aoclsparse_int        order,
                                          aoclsparse_int        base,
                                          double                alpha,
                                          aoclsparse_int        m,
                                          aoclsparse_int        k,
                                          const double         *val,
                                          const aoclsparse_int *col,
                                          const aoclsparse_int *row_ptr,
                                          const double         *B,
                                          aoclsparse_int        n,
                                          aoclsparse_int        ldb,
                                          double                beta,
                                          double               *C,
                                          aoclsparse_int        ldc);

#ifdef __cplusplus
}
#endif
#endif /* AOCLSPARSE_MI355_H_ */
