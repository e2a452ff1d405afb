// internal.hpp -- object model of the MI355X engine behind the aoclsparse_* C ABI.
//
// Host side keeps the reference's semantics (handles alias the user's CSR arrays, clean CSR
// with idiag/iurow, hint list, status codes); everything a kernel touches lives in HBM in the
// DeviceCsr mirrors below.  Layout rationale is in DESIGN.md.
#pragma once

#include "aoclsparse.h"
#include "aoclsparse_mi355.h"

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <algorithm>
#include <thread>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <system_error>
#include <vector>

// ---- descriptor (library/src/include/aoclsparse_descr.h:37-47) --------------------------
struct _aoclsparse_mat_descr
{
    aoclsparse_matrix_type type      = aoclsparse_matrix_type_general;
    aoclsparse_fill_mode   fill_mode = aoclsparse_fill_mode_lower;
    aoclsparse_diag_type   diag_type = aoclsparse_diag_type_non_unit;
    aoclsparse_index_base  base      = aoclsparse_index_base_zero;
};

namespace mi355
{

// ---- status helpers ----------------------------------------------------------------------
#define MI355_HIP_TRY(expr)                                   \
    do                                                        \
    {                                                         \
        hipError_t e__ = (expr);                              \
        if(e__ != hipSuccess)                                 \
            return ::mi355::map_hip_error(e__);               \
    } while(0)

aoclsparse_status map_hip_error(hipError_t e);

// ---- complex values (aoclsparse_float_complex / aoclsparse_double_complex layout: {real, imag}) ---------
template <typename R>
struct cplx
{
    R re, im;
    __host__ __device__ cplx() = default;
    __host__ __device__ constexpr cplx(R r, R i = R(0))
        : re(r)
        , im(i)
    {
    }
};
using cfloat  = cplx<float>;
using cdouble = cplx<double>;
inline float  conj_of(float v) { return v; }
inline double conj_of(double v) { return v; }
template <typename R>
inline cplx<R> conj_of(cplx<R> v)
{
    return cplx<R>(v.re, -v.im);
}
// f(T{}) with T = float / double / cfloat / cdouble for the handle's value type
template <typename F>
auto dispatch_value_type(aoclsparse_matrix_data_type t, F &&f)
{
    switch(t)
    {
    case aoclsparse_smat:
        return f(float{});
    case aoclsparse_cmat:
        return f(cfloat{});
    case aoclsparse_zmat:
        return f(cdouble{});
    default:
        return f(double{});
    }
}
inline bool is_complex_type(aoclsparse_matrix_data_type t)
{
    return t == aoclsparse_cmat || t == aoclsparse_zmat;
}

// ---- hinted actions (library/src/include/aoclsparse_mat_structures.hpp:36-68) -------------
enum hinted_action
{
    action_none = 0,
    action_mv,
    action_sv,
    action_mm,
    action_2m,
    action_ilu0,
    action_sm_row,
    action_sm_col,
    action_dotmv,
    action_symgs,
    action_sorv_forward,
    action_sorv_backward,
    action_sorv_symm,
    action_max
};

// descriptor+operation id; same enumeration idea as include/aoclsparse_mtx_dispatcher.hpp:41-74
// restricted to what real types can produce (conjugate == plain for real data).
enum class doid : int
{
    gn = 0, // general, no-trans
    gt, // general, transpose
    sl, // symmetric lower / upper (transpose is the same operation)
    su,
    tln, // triangular lower/upper x none/transpose
    tlt,
    tun,
    tut,
    len
};

doid get_doid(const _aoclsparse_mat_descr *d, aoclsparse_operation op);

struct Hint
{
    hinted_action          act;
    doid                   id;
    aoclsparse_operation   trans;
    aoclsparse_matrix_type type;
    aoclsparse_fill_mode   fill;
    aoclsparse_int         nop;
    aoclsparse_int         kid;
    bool                   optimized = false;
};


// Host-side analysis loops over rows / blocks that are independent of each other: fixed chunking (results do not depend on
// the thread count), at most 16 threads, inline below `grain` items.  fn(begin, end).
template <typename F>
inline void parallel_for(long long n, long long grain, F fn)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int      nt = (int)std::min<long long>(std::min<unsigned>(16u, hw ? hw : 1u), (n + grain - 1) / std::max<long long>(grain, 1));
    if(nt <= 1)
    {
        fn(0LL, n);
        return;
    }
    // An exception inside a chunk (bad_alloc) is carried to the caller, and a thread that cannot be started costs nothing
    // but its parallelism (the chunk then runs here): the C ABI above this never sees std::terminate.
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    std::exception_ptr       err;
    std::mutex               err_lock;
    auto                     chunk = [&](int t) {
        try
        {
            fn(n * t / nt, n * (t + 1) / nt);
        }
        catch(...)
        {
            std::lock_guard<std::mutex> g(err_lock);
            if(!err)
                err = std::current_exception();
        }
    };
    for(int t = 0; t < nt; t++)
    {
        try
        {
            th.emplace_back(chunk, t);
        }
        catch(const std::system_error &)
        {
            chunk(t);
        }
    }
    for(auto &t : th)
        t.join();
    if(err)
        std::rethrow_exception(err);
}

// Diagnostic: AOCLSPARSE_MI355_TIMING=1 prints the wall time of the analysis phases it brackets to stderr.
struct PhaseTimer
{
    const char                           *name;
    std::chrono::steady_clock::time_point t0;
    static bool                           on()
    {
        static const bool v = [] { const char *e = std::getenv("AOCLSPARSE_MI355_TIMING"); return e && std::atoi(e) != 0; }();
        return v;
    }
    explicit PhaseTimer(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~PhaseTimer()
    {
        if(on())
            std::fprintf(stderr, "[mi355 timing] %-34s %8.1f ms\n", name,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

struct LapTimer // same switch: lap("what") prints the time since the previous lap
{
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void                                  lap(const char *what)
    {
        if(!PhaseTimer::on())
            return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mi355 timing]     %-30s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// ---- device memory -------------------------------------------------------------------------
struct DeviceBuffer
{
    void  *ptr   = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &)            = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer();
    aoclsparse_status alloc(size_t nbytes);
    aoclsparse_status upload(const void *host, size_t nbytes, hipStream_t s); // alloc + H2D
    // alloc on the CURRENT device + device-to-device copy of src (which may live on another device: peer copy over xGMI when
    // peer access is enabled, staged by the runtime otherwise); enqueued on s, not waited for
    aoclsparse_status clone_from(const DeviceBuffer &src, hipStream_t s);
    void              release();
    // take over other's allocation (this one's own is released first)
    void adopt(DeviceBuffer &other)
    {
        if(this == &other)
            return;
        release();
        ptr = other.ptr, bytes = other.bytes;
        other.ptr = nullptr, other.bytes = 0;
    }
    template <typename T>
    T *as() const
    {
        return static_cast<T *>(ptr);
    }
};

// CSR arrays resident in HBM, always 4/8-byte element arrays in the SAME base as the host
// copy they mirror (kernels subtract base at load time).
struct DeviceCsr
{
    aoclsparse_int m = 0, n = 0, nnz = 0, base = 0;
    DeviceBuffer   ptr, ind, val;
    bool           valid = false;
};

// SELL-64 twin of a device CSR (sell_kernels.hip): built by aoclsparse_optimize for an mv hint when the
// padding stays small; the handle's ?mv then runs on it instead of the CSR-Adaptive kernel.
struct SellPlan
{
    aoclsparse_int nslices = 0;
    int            pack    = 1; // 1: cell (p, lane) at 64 p + lane; 4: at 256 (p/4) + 4 lane + p%4
    long long      cells   = 0; // stored cells = sum over slices of 64 * (longest row of the slice)
    DeviceBuffer   slice_ptr; // nslices+1 cell offsets (long long)
    DeviceBuffer   val, col; // cells values / 0-based columns (-1 = padding)
    DeviceBuffer   rowlen; // m row lengths (read by the 4- and 8-lane orders only)
    // shared column lists (sell_kernels.hip): col holds one list per leader lane of a slice, ccells entries in all;
    // cptr = nslices+1 offsets into it, lead = m x 16 bits (leader index of a row inside its slice | column shift << 8)
    bool           shared = false;
    long long      ccells = 0;
    DeviceBuffer   cptr, lead;
    bool           valid = false, tried = false;
    bool           wanted = false; // optimize chose SELL: rebuilt lazily after the values change
    // products served by this copy: odd ones walk the slices in descending order, so that what one product leaves in the
    // Infinity Cache (the END of its sweep) is where the next one starts (sell_kernels.hip); the bits do not depend on it
    mutable std::atomic<unsigned> products{0};
    int next_direction() const;
};

// csrmm row groups (csrmm_kernels.hip: csrmm_rowgroup_kernel): runs of consecutive rows with one column pattern
constexpr int CSRMM_GROUP = 8; // rows per group at most
struct MmGroups
{
    // row-major, n >= 128, no groups: most rows repeat the previous row's column list shifted by one (a stencil) and have
    // <= 8 entries -> csrmm_row_run_kernel (a wave walks 8 rows and keeps the previous row's B rows in registers)
    bool           runs_tried = false, row_runs = false;
    // ... and, when the matrix is BANDED (most rows end `band` columns right of the diagonal, band >= 256, a 2-D / 3-D
    // stencil), the order in which the kernel's 8-row blocks are walked: strips of <= 256 rows of the band, every strip
    // from the top of the matrix to the bottom (run_order[b] = first row of the b-th block).  In row order the three
    // touches of a B row (from the band above, the row itself, the band below) are `band` rows of traffic apart --
    // 2 and 4 MB at band 1000 and 128 columns, more than an XCD's 4 MB L2 keeps: measured 2.5 x B fetched; in strip
    // order they are one strip apart.
    DeviceBuffer   run_order;
    aoclsparse_int band = 0;
    // ... and row blocks for the narrow (n < 128) row-major kernel that follow the lines of the band: every line of `band` rows in 8
    // blocks, {first row, first non-zero} like SpmvPlan::rowblocks (+ a terminal entry).  In launch order block b runs on XCD b % 8, so
    // rows i and i +- band meet in one L2 and all eight XCDs work inside one line (csrmm_tile_kernel; 0 blocks: not built)
    DeviceBuffer   slab_blocks;
    aoclsparse_int slab_nblocks = 0;
    aoclsparse_int ngroups = 0;
    int            max_rows = 0; // rows of the largest group
    DeviceBuffer   first; // ngroups + 1 row indices
    bool           valid = false, tried = false;
    // column-major csrmm: row pairs (2r, 2r+1) where row 2r+1 carries row 2r's pattern shifted by one column (scalar
    // stencils, banded matrices) and both fit the register cache -- csrmm_colpair_kernel serves them with 16-byte loads
    bool           pairs_tried = false, pairs = false;
    aoclsparse_int npairs = 0, nsingles = 0;
    DeviceBuffer   pair_first, single_rows; // first row of every pair; rows without a partner
    // column-major csrmm, round 4: every block of win_rows consecutive rows touches one short stretch of columns (banded
    // matrices, stencils in natural order) -- csrmm_colwin_kernel stages that stretch of each B column in LDS.
    // windows[2b], windows[2b+1] = first column (0-based, 16-byte granule) and 16-byte pieces of block b's stretch
    bool           win_tried = false, win = false;
    int            win_rows  = 0;
    DeviceBuffer   windows;
};

// merge-path tiling of a device CSR (mergepath_kernels.hip): tile w starts at {row ends, non-zeros} =
// starts[2w], starts[2w+1]
constexpr int MP_ITEMS = 1024; // rows + non-zeros per workgroup
constexpr int SELL_PROMOTE_CALLS = 8; // an un-hinted handle gets its SELL-64 copy at this many products
struct MergePlan
{
    aoclsparse_int ntiles = 0;
    DeviceBuffer   starts; // (ntiles + 1) x {i, j}
    // 2 * ntiles: first[w] = first tile that holds a head piece of the row whose END lies in tile w, or -1; then endt[v] = the
    // tile in which the row of tile v's head piece ends, or -1
    DeviceBuffer   first;
    bool           valid = false, tried = false;
    // Pieces of cut rows meet through a piece set: per tile a head piece, a tail piece (8 bytes each) and an arrival counter
    // (mergepath_kernels.hip).  A launch leaves every counter at 0, so a set carries nothing from one launch to the next; two
    // launches in flight at once must not share one, and launches on one stream never are: every stream that has run this plan
    // owns a set.  No launch ever waits for another stream.  Streams are told apart by {address, hipStreamGetId}: an address
    // the runtime hands out again after the caller destroyed a stream is a NEW stream, and work of the old one may still be
    // running -- the set is taken over after a device synchronisation.  (mutable: products run on a const plan under the
    // handle's shared lock; `launch_lock` makes {find the set, enqueue} one step.)
    struct GranuleSet
    {
        void              *stream = nullptr;
        unsigned long long uid    = 0;
        DeviceBuffer       granules; // 3 * ntiles x 8 bytes: head pieces, tail pieces, counters; zeroed at allocation
    };
    static constexpr size_t                           MAX_SETS = 16; // more streams than this: the oldest sets are recycled after a device sync
    mutable std::mutex                               launch_lock;
    mutable std::vector<std::unique_ptr<GranuleSet>> sets;
};

// Blocked-ELL copy of a block-dense matrix for the MFMA csrmm (csrmm_bell_kernels.hip; round 4): 16 x 16 blocks, `width` block
// slots per block row (bcol = -1: empty slot), values dense per block in the v_mfma_f64_16x16x4 A-operand order (element
// (i, k) of a block: fragment t = k / 4, lane = 16 * (k % 4) + i, stored at 128 * (t / 2) + 2 * lane + t % 2).  Built for double matrices whose 16 x 16 tiles are at least half full
// (the reference picks its blocked CSR by the same kind of fill threshold: conversion/aoclsparse_convert.cpp:36-147).
constexpr int BELL_BS = 16;
struct BellPlan
{
    bool           tried = false, valid = false;
    aoclsparse_int nbr = 0, width = 0; // block rows, block slots per block row
    long long      nblocks = 0; // stored (non-empty) blocks
    double         fill = 0.0; // nnz / (256 * nblocks)
    DeviceBuffer   val, bcol; // nbr * width * 256 values; nbr * width block columns
    // Which XCD works through which block rows, in which order (build_bell's model of the eight L2s, csrmm_api.cpp): order[8 p + x] = the
    // p-th block row of XCD x (-1 past the end of its list), order_len lists entries per XCD; order_len == 0: launch order.
    DeviceBuffer   order;
    aoclsparse_int order_len = 0;
    int            xcd_chunk = 1; // chunked deal: consecutive block rows per XCD turn (1: launch order; 0: a lattice sweep)
    aoclsparse_int lattice[3] = {0, 0, 0}; // lattice sweep: block rows per line, lines per plane, planes (else 0)
    int            region[2]  = {0, 0}; // ... and the cross-section of an XCD's region (block rows along the line x lines)
    int            region_cut[2] = {0, 0}; // ... pieces per line, segments of the plane range
    double         model_fetches = 0.0, model_fetches_launch_order = 0.0; // modelled L2 misses per distinct B block row, chosen / launch order
};

// SpMV execution plan (CSR-Adaptive row blocks), see spmv_kernels.hip
struct SpmvPlan
{
    aoclsparse_int nblocks     = 0;
    aoclsparse_int long_rows   = 0;
    aoclsparse_int max_row_nnz = 0;
    aoclsparse_int tile        = 0; // LDS tile (non-zeros per row block): 1024 or 2048
    DeviceBuffer   rowblocks; // nblocks+1 entries {first row, first non-zero (0-based)}
    // the same blocks as {first row, first non-zero, rows, non-zeros}, blocks holding a long row FIRST (a row is one
    // lane's serial chain: a block with a 300-entry row runs twice as long as the others, and started last it is the
    // kernel's tail); empty when no block is heavy.  Only csr_adaptive_kernel reads it.
    DeviceBuffer   rowblocks4;
    bool           heavy_first = false;
    bool           valid = false;
    SellPlan       sell;
    MergePlan      merge;
    MmGroups       mm;
    BellPlan       bell;
    std::atomic<int> mv_calls{0}; // products served from this plan without a SELL copy (promotion counter;
                                  // concurrent ?mv calls on one handle are allowed, as in the reference)
    // products of the row-block kernel: odd ones walk the blocks in descending order (see SellPlan::products)
    mutable std::atomic<unsigned> sweeps{0};
    // the same for csrmm: every second product of a handle runs its blocks in descending order (csrmm_kernels.hip: mm_block_index)
    mutable std::atomic<unsigned> mm_products{0};
};

// Direction of the NEXT csrmm launches of the calling thread (set by csrmm_api.cpp right before it dispatches; the launchers put it
// into bit 30 of the kernels' XCD-chunk word).  0: ascending.
constexpr int MM_DESCENDING = 0x40000000;
constexpr int MM_DEAL       = 0x20000000; // the chunk of the word counts blocks per XCD TURN (chunks dealt round-robin), not per XCD
int  &mm_direction_word();

// TRSV plan of one (triangle, op) pair (trsv_api.cpp / trsv_kernels.hip): the strict triangle
// re-laid out in LEVEL ORDER.  Position k holds row rowmap[k]; its entries sit at
// [pptr[k], pptr[k+1]) of pind/pval already in the order the reference's chain consumes them, so
// a level is one contiguous slab of HBM and every kernel walks forward.
struct TrsvSegment
{
    aoclsparse_int l0, l1; // levels [l0, l1)
    bool           narrow; // true: one single-workgroup launch loops over the levels
};
// Supernodal variant of the plan (trsv_block_kernel; all four (fill, op) triangles of real types): rows consecutive in
// SOLVE order where each depends on exactly its predecessor's dependencies plus the predecessor (the dofs of a mesh node
// in an ILU(0) factor) form a BLOCK that ONE lane solves back to back -- the block's external dependencies are waited for
// once, the rows inside it hand their results on in registers -- so the dependency DAG is levelled per block (shell-like
// factor: 1,101 block levels instead of 5,505 row levels).  Own level-ordered copy of the triangle (positions = rows in
// block-level order, in solve order inside a block; entries in chain order).
constexpr int TRSV_BLK_ROWS = 8; // rows per block at most
constexpr int TRSV_BLK_EXT  = 24; // external dependencies of a multi-row block at most
// entries of a multi-row block at most: 5 rows on 20 external dependencies (a mesh node of 5 unknowns below 4 neighbours: 5 x 20 + 10).
// (96 until round 6: such a node was cut into 4 rows + 1 single row of 24 entries -- an extra block level per node, and the single rows
// put the whole solve on the (5, 24) shape of trsv_block_kernel, whose values live in LDS: 2.5 instead of 1.7 us per block level.)
constexpr int TRSV_BLK_NV   = 110;
// spare elements behind the m x nrhs position-ordered solution buffer: [0, 64) parked tag stores, [64, 128) parked x stores
// (one slot per lane), the last one (191) the slot that always holds 0 -- apart from the parked ones: a parked NaN must not reach it
constexpr int TRSV_XP_PAD = 192;
// Two-level schedule on top of the block plan (trsv_chunk_kernel; round 6; L, L^T, U^T of real types): the blocks in their
// natural (solve) order are cut into CHUNKS of consecutive blocks, one workgroup per chunk.  A chunk walks its own blocks in
// block-level order, a STEP (<= 64 / TRSV_CHUNK_LANES blocks of one level, TRSV_CHUNK_LANES lanes per block: a lane per row) per
// wavefront, steps dealt round-robin to the workgroup's wavefronts.  The chunk's x lives in LDS (NaN-tagged words, as in HBM):
// a dependency inside the chunk is polled there (~0.1 us per hand-off).  Rows of EARLIER chunks a chunk depends on are its HALO:
// one wavefront of the workgroup polls them in xp (HBM, ~1.5-2 us) in the order of their first use and copies them into LDS
// slots behind the chunk's own rows -- the wavefronts that solve only ever poll LDS.  The chunk plan shares the block plan's
// level-ordered copy of the triangle; its own data:
//   steps  nsteps x 8 words {first block (index in block-level order), position of its first row, LDS slot of that row, rows of
//          the step's blocks as nibbles; rows in front of block j as bytes (2 words), rows of the step, blocks}
//   cptr   nchunks + 1 first step of every chunk, then nchunks: rows of the chunk, nchunks: first halo entry (a multiple of 4),
//          nchunks: halo entries
//   eptr   nblocks + 1 offsets into cind (indexed like bfirst)
//   cind   the external dependencies of every block as LDS slots of its chunk (own rows first, halo behind them)
//   hind   the halos: positions in xp, per chunk in the order of first use
constexpr int TRSV_CHUNK_LANES = 8; // lanes per block = TRSV_BLK_ROWS
// wavefronts per workgroup: all but the last take steps, the last one fetches the halo.  (4 + 1: 1.144 ms on the shell-like factor, 5 + 1:
// 1.103, 7 + 1: 1.102 -- the solve is paced by the dependency chain; two staging areas fewer leave LDS for 12 % more rows per chunk,
// i.e. fewer chunk boundaries on the critical path)
constexpr int TRSV_CHUNK_WAVES = 6;
// a solving wavefront's staging area holds the values of one step: 8 blocks of bs rows with ext external + (bs - 1) / 2 internal entries
constexpr int trsv_chunk_pcap(int bs, int ext)
{
    return 8 * (bs * ext + bs * (bs - 1) / 2) + ext + 16; // (+ what a read past the last block's entries can overshoot)
}
// LDS slots (8 bytes as double) of a chunk -- own rows + halo -- at most: what the CU's 160 KB leave next to the staging areas
constexpr int trsv_chunk_slots(int bs, int ext)
{
    return ((160 * 1024 - 16 - (TRSV_CHUNK_WAVES - 1) * (trsv_chunk_pcap(bs, ext) * 8 + 9 * ext * 4) - 65 * 8) / 8) / 256 * 256;
}
constexpr int TRSV_CHUNK_ROWS = trsv_chunk_slots(5, 16); // the largest of the four shapes
struct TrsvChunkPlan
{
    bool           tried = false, valid = false;
    aoclsparse_int nchunks = 0, nsteps = 0, max_rows = 0; // max_rows: LDS slots (rows + halo) of the largest chunk
    double         model_us = 0.0, model_block_us = 0.0; // plan-time estimates: this schedule / the lane-per-block one
    DeviceBuffer   steps, cptr, eptr, cind, hind;
};
struct TrsvBlockPlan
{
    double         slice_fan_in = 0.0; // producer slices a slice of 64 blocks waits for, on average (trsv_api.cpp: the slice width)
    TrsvChunkPlan  chunk;
    bool           tried = false, valid = false;
    bool           front = false; // a row's chain starts with the rows of its own block (U), instead of ending with them
    aoclsparse_int nblocks = 0, nslices = 0, nlevels = 0;
    int            max_rows = 1, max_ext = 0; // largest block / largest external list of a multi-row block
    DeviceBuffer   rowmap, pptr, pind, pval; // as TrsvPlan, in block-level order
    DeviceBuffer   bfirst; // nblocks+1: first position of every block
    // nslices+1: first BLOCK of every slice (<= 64 blocks of one block level); then nslices: the slice's block level;
    // then nlevels+1: first slice of every level (the kernel's gate counts finished slices per level)
    DeviceBuffer   slices;
};
struct TrsvPlan
{
    TrsvBlockPlan               blk;
    aoclsparse_int              nlevels   = -1;
    aoclsparse_int              max_width = 0; // widest level
    aoclsparse_int              launches  = 0; // kernel launches of the hybrid schedule
    aoclsparse_int              nnz_tri   = 0; // entries of the strict triangle
    std::vector<aoclsparse_int> level_ptr; // host, nlevels+1
    std::vector<TrsvSegment>    segments; // hybrid schedule
    DeviceBuffer                rowmap, levels; // device: m rows in level order; level_ptr copy
    DeviceBuffer                pptr, pind, pval; // device: level-ordered strict triangle; pind = positions
    // level slices for trsv_slice_kernel: slice w = positions [slices[w], slices[w+1]), <= 64 of them, never across a
    // level boundary (so the lanes of a wavefront are independent of each other)
    aoclsparse_int              nslices = 0;
    DeviceBuffer                slices;
    bool                        valid = false;
    // the level-ordered layout above (rowmap .. slices) is on the device.  Round 3: when the triangle has a block plan the
    // automatic schedule never touches it, so it is built only when something asks for it (a forced row-level schedule, the
    // hybrid schedule, complex types, sorv): levels / nlevels / nnz_tri are always valid once `valid` is.
    bool rows_valid = false;
};

// host CSR view; owned==true when the library allocated the arrays (clean copy / transpose)
struct HostCsr
{
    aoclsparse_int        m = 0, n = 0, nnz = 0;
    aoclsparse_index_base base  = aoclsparse_index_base_zero;
    aoclsparse_int       *ptr   = nullptr;
    aoclsparse_int       *ind   = nullptr;
    void                 *val   = nullptr;
    aoclsparse_int       *idiag = nullptr; // owned whenever non-null
    aoclsparse_int       *iurow = nullptr;
    bool                  owned        = false;
    bool                  is_optimized = false;
    // owned ind / val came from host_result_alloc (sp2m results: 2 MB-aligned, huge pages asked for) and are freed with free()
    bool                  result_arrays = false;
    ~HostCsr();
};
// Host arrays a product hands back (sp2m): a copy out of HBM first-touches them, and on a fresh new[] that costs more than the
// copy -- 156 MB: 11.6-27 ms in 4 KB pages whatever the number of touching threads, 1.1 ms as 2 MB pages touched by 16 threads
// (tools/history/pagefault_probe.cpp on the GPU box).  2 MB-aligned + MADV_HUGEPAGE from 4 MB up, plain malloc below; released with free().
void *host_result_alloc(size_t bytes);
void  host_result_touch(void *p, size_t bytes); // first touch by several threads (no-op for small arrays)

// A general CSR derived from the clean CSR so that the general kernels can serve a symmetric or
// triangular descriptor: key = (type, fill, diag, transposed).
struct Derived
{
    int       type = 0, fill = 0, diag = 0, trans = 0;
    HostCsr   host; // owned, 0-based, sorted rows
    DeviceCsr dev;
    SpmvPlan  plan;
};

} // namespace mi355

// ---- matrix handle (library/src/include/aoclsparse_mat_structures.hpp:774-859) -------------
struct _aoclsparse_matrix
{
    aoclsparse_int                m = 0, n = 0, nnz = 0;
    aoclsparse_index_base         base         = aoclsparse_index_base_zero;
    aoclsparse_matrix_data_type   val_type     = aoclsparse_dmat;
    aoclsparse_matrix_format_type input_format = aoclsparse_csr_mat;
    int                           sort         = 0; // aoclsparse_matrix_sort
    bool                          fulldiag     = false;
    aoclsparse_memory_usage       mem_policy   = aoclsparse_memory_usage_unrestricted;
    bool                          optimized    = false;
    bool                          opt_csr_full_diag = false;

    mi355::HostCsr                  user; // aliases the caller's arrays
    std::unique_ptr<mi355::HostCsr> opt_copy; // clean copy when the user arrays are not clean
    mi355::HostCsr                 *opt = nullptr; // clean CSR: &user or opt_copy.get()
    std::unique_ptr<mi355::HostCsr> trans; // A^T of the user CSR (built for gt hints / first use)

    std::vector<mi355::Hint> hints; // newest first (csr_util.cpp:47-100 prepends)

    // device side; guarded by `guard` (executors take it shared, builders exclusive)
    mi355::DeviceCsr dev_user, dev_trans;
    mi355::SpmvPlan  plan_user, plan_trans;
    mi355::TrsvPlan  trsv_plan[6]; // index: (upper?2:0) + (transpose?1:0); complex op = H: 4 + (upper?1:0)
    mi355::DeviceBuffer dev_diag; // diagonal values of the clean CSR (length min(m,n))
    mi355::DeviceBuffer trsv_scratch; // ticket (one per right-hand side) + timeout words of the sync-free solve
    mi355::DeviceBuffer trsv_xp; // solution(s) in level order, m x nrhs (stream-ordered reuse)
    // the stream of the last call that used the handle's workspaces (the two above, `work` below): they are reused in stream
    // order, so a call on ANOTHER stream first waits for the device -- not for that stream, which the caller may have destroyed
    // since (workspace_stream_guard; written under the runtime's stage lock)
    void              *ws_last_stream = nullptr;
    unsigned long long ws_last_uid    = 0; // hipStreamGetId: a recycled stream address is another stream
    bool               ws_ran         = false;
    // one word of pinned, device-mapped host memory THIS handle's sync-free solves set when a wait expires (round 3,
    // the process-wide word of round 2 could not say which handle had failed, and a failure surfaced on an
    // unrelated solve).  Allocated at the handle's first sync-free solve; read without a device round trip.
    volatile unsigned int *trsv_timeout_host = nullptr;
    unsigned int          *trsv_timeout_dev  = nullptr;

    // matrices derived from the clean CSR for non-general descriptors (symmetric expansion,
    // triangular slices), built on first use: see derived.cpp
    std::vector<std::unique_ptr<mi355::Derived>> derived;

    // sp2m stage-1 state (C handles own their arrays)
    bool owns_user_arrays = false;

    // other input formats (formats_api.cpp): the caller's CSC arrays of a handle created from CSC (its CSR
    // lives in `user`, owned), or the caller's COO arrays (input_format == aoclsparse_coo_mat, no CSR)
    aoclsparse_int *csc_ptr = nullptr, *csc_ind = nullptr;
    void           *csc_val = nullptr;
    aoclsparse_int *coo_row = nullptr, *coo_col = nullptr;
    void           *coo_val = nullptr;

    // composite solvers (solvers_api.cpp): vector workspaces in HBM, and the ILU(0) factors
    // (solvers/aoclsparse_ilu.hpp:94-104, analysis.cpp:390-425): values on the user's pattern, kept
    // on the host for *precond_csr_val and mirrored by a factor handle whose TRSV plans do the solves
    mi355::DeviceBuffer work[6];
    bool                ilu_ready = false, ilu_factorized = false;
    void               *ilu_val = nullptr; // nnz values, library-owned (malloc)
    aoclsparse_matrix   ilu_factor = nullptr; // aliases user.ptr / user.ind / ilu_val

    // multi-device calls (aoclsparse_mi355_?csrmm_multi): replica[i] = a handle over the SAME host arrays whose device copy
    // and plans live on the device of runtime slot i (slot 0 is this handle itself); built on first use, dropped by
    // aoclsparse_mi355_invalidate / ?set_value / ?update_values and by aoclsparse_destroy
    std::vector<aoclsparse_matrix> replicas;
    int                            replicas_cloned = 0; // of them: device state copied device to device instead of re-analysed

    mutable std::shared_mutex guard;
};

namespace mi355
{
// identity of a stream beyond its address (the runtime reuses the address of a destroyed stream for a new one): hipStreamGetId
// where the loaded HIP runtime has it (ROCm >= 7.1; looked up at run time -- the process may run on the older runtime a
// framework bundles), else 0 for every stream (the address alone decides, as in rounds 1-5)
unsigned long long stream_uid(hipStream_t s);
// call with the runtime's stage lock held, before a handle's workspaces are touched on stream s
inline aoclsparse_status workspace_stream_guard(_aoclsparse_matrix *A, hipStream_t s)
{
    const unsigned long long uid = stream_uid(s);
    if(A->ws_ran && (A->ws_last_stream != (void *)s || A->ws_last_uid != uid))
        MI355_HIP_TRY(hipDeviceSynchronize());
    A->ws_last_stream = (void *)s, A->ws_last_uid = uid, A->ws_ran = true;
    return aoclsparse_status_success;
}
} // namespace mi355

namespace mi355
{

// ---- runtime (one process per GPU; uses the current HIP device) ----------------------------
class Runtime
{
public:
    // The calling thread's runtime: the primary one (the process's GPU), unless a RuntimeScope put a secondary slot in
    // charge (in-library multi-device calls, csrmm_api.cpp: one worker thread per device).
    static Runtime &get();
    static Runtime &primary();
    // secondary slot idx >= 1 bound to HIP device `dev` (created on first use; own stream, staging buffers and pinned words).
    // Two slots may name the same device: that is how the multi-device control flow is tested on a one-GPU box.
    static Runtime *slot(int idx, int dev);
    static std::vector<Runtime *> secondary(); // the slots created so far
    aoclsparse_status init(); // lazy; internal_error when no device; binds the calling thread to `device`
    void              bind_thread();
    hipStream_t       stream() const
    {
        return stream_;
    }
    void set_stream(hipStream_t s)
    {
        stream_ = s;
    }
    aoclsparse_mi355_pointer_mode pointer_mode = aoclsparse_mi355_pointer_auto;
    // TRSV schedule: -1 = chosen from the plan (default); 0..5 force one (aoclsparse_mi355_set_trsv_schedule; trsv_api.cpp)
    int trsv_schedule = -1;
    // true when p is memory the device can dereference (device or managed allocation)
    bool is_device_pointer(const void *p);
    int  device = -1, cus = 0;
    char name[256] = {0};
    // scratch staging buffers for host-pointer calls (grown on demand, reused)
    aoclsparse_status staging(int slot, size_t bytes, void **out);
    size_t            release_staging(); // frees every slot (after the stream has drained); bytes freed
    // Host <-> device copies of PAGEABLE caller memory on the library's stream (plain stream-ordered copies: the runtime's own
    // staging runs at 56 GB/s on these boxes; h2d returns once enqueued, d2h likewise -- callers synchronise the stream)
    aoclsparse_status h2d(void *dev, const void *host, size_t bytes);
    aoclsparse_status d2h(void *host, const void *dev, size_t bytes);
    hipEvent_t        ev0 = nullptr, ev1 = nullptr;
    // per-iteration timing: a ring of events recorded by aoclsparse_mi355_timer_mark (runtime.cpp)
    std::vector<hipEvent_t> marks;
    size_t                  marks_used = 0;
    // sync-free TRSV: one word of pinned, device-mapped host memory a kernel sets when a wait expires.  The host reads
    // it without a device round trip: right after the stream sync of a host-pointer solve, and at the START of the next
    // solve for device-pointer callers (which therefore stay asynchronous).
    volatile unsigned int *trsv_timeout_host = nullptr;
    unsigned int          *trsv_timeout_dev  = nullptr;
    // second word of the same pinned line: "cached raw-csrmv plan does not match this row_ptr" (spmv_api.cpp)
    volatile unsigned int *plan_stale_host = nullptr;
    unsigned int          *plan_stale_dev  = nullptr;
    std::mutex        lock;
    std::recursive_mutex stage_lock; // serialises calls that stage host buffers

    int forced_device = -1; // secondary slots: the device to bind to (the primary honours AOCLSPARSE_MI355_DEVICE / the current device)

private:
    std::atomic<bool> inited_{false}; // set (release) after init_status_ / device / events are written
    aoclsparse_status init_status_ = aoclsparse_status_success;
    hipStream_t  stream_ = nullptr;
    DeviceBuffer stage_[48]; // 0-7: csrmv / mv / trsv / dotmv, 8-15: ELL family and BLKCSR (ell_api.cpp, blk_api.cpp), 16-39: sp2m (sp2m_api.cpp),
                             // 40-45: the temporaries of the device transpose (transpose_kernels.hip; sp2m transposes operands while its own slots are in use)
};

// While one of these is alive on a thread, Runtime::get() on that thread is the given slot: its device is current, its
// stream / staging buffers are used and handles built under it live on its device.  Settings of the primary runtime
// (pointer mode, csrmm beta = 0 policy) are inherited at entry.
struct RuntimeScope
{
    explicit RuntimeScope(Runtime *r);
    ~RuntimeScope();
    RuntimeScope(const RuntimeScope &)            = delete;
    RuntimeScope &operator=(const RuntimeScope &) = delete;
    Runtime          *prev;
    aoclsparse_status status;
};

// While one of these is alive on a thread, every pointer handed to the executors by that thread is
// taken as device memory: composite routines (symgs, ilu smoother, iterative solvers) chain the
// executors on their own HBM workspaces whatever pointer mode the caller selected.
struct DeviceScope
{
    DeviceScope();
    ~DeviceScope();
    DeviceScope(const DeviceScope &)            = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// A host or device array made addressable by the GPU for one call of a raw-array entry point.
struct StagedArg
{
    void  *dev = nullptr, *host = nullptr;
    size_t bytes  = 0;
    bool   staged = false;
    aoclsparse_status in(Runtime &rt, int slot, const void *p, size_t nbytes, bool copy)
    {
        bytes = nbytes;
        if(rt.is_device_pointer(p))
        {
            dev = const_cast<void *>(p);
            return aoclsparse_status_success;
        }
        staged = true;
        host   = const_cast<void *>(p);
        aoclsparse_status st = rt.staging(slot, nbytes ? nbytes : 1, &dev);
        if(st != aoclsparse_status_success)
            return st;
        if(copy && nbytes)
            return rt.h2d(dev, p, nbytes); // pipelined through pinned memory when large
        return aoclsparse_status_success;
    }
    aoclsparse_status out(Runtime &rt)
    {
        if(staged && bytes)
            return rt.d2h(host, dev, bytes);
        return aoclsparse_status_success;
    }
};

// ---- host analysis (matrix.cpp) --------------------------------------------------------------
aoclsparse_status mat_check(aoclsparse_int maj, aoclsparse_int mind, aoclsparse_int nnz,
                            const aoclsparse_int *ptr, const aoclsparse_int *ind, const void *val,
                            int shape, aoclsparse_index_base base, int &sort, bool &fulldiag);
aoclsparse_status check_sort_diag(aoclsparse_int m, aoclsparse_int n, aoclsparse_index_base base,
                                  const aoclsparse_int *ptr, const aoclsparse_int *ind, bool &sorted,
                                  bool &fulldiag);
aoclsparse_status csr_indices(aoclsparse_int m, aoclsparse_index_base base,
                              const aoclsparse_int *ptr, const aoclsparse_int *ind,
                              aoclsparse_int **idiag, aoclsparse_int **iurow);
// builds A->opt (clean CSR + idiag/iurow) if absent; thread-safe (double-checked)
aoclsparse_status csr_optimize(aoclsparse_matrix A);
// forget every copy derived from the user's arrays (clean CSR, transposes, device mirrors, SELL, TRSV plans)
void drop_derived_state(aoclsparse_matrix A);
// value mutation on handles created from CSC / COO arrays (formats_api.cpp): the caller's arrays are the first
// representation there; csc_refresh_csr rebuilds the handle's CSR values from the CSC arrays
aoclsparse_status coo_set_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx, const void *val);
void              csc_set_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx, const void *val);
aoclsparse_status csc_refresh_csr(aoclsparse_matrix A);
// allocates the ILU(0) value array as a copy of A's values (solvers_api.cpp; analysis.cpp:390-425)
aoclsparse_status ilu_prepare(aoclsparse_matrix A);
// builds A->trans (host transpose of the user CSR, 0-based) if absent
aoclsparse_status build_transpose(aoclsparse_matrix A);
// B = A^T of a device CSR in the reference's counting-sort order (transpose_kernels.hip), into buffers it allocates once the matrix
// is accepted; aoclsparse_status_not_implemented when a column is too long for the device sort (the caller sorts on the host; the
// three buffers are then empty)
aoclsparse_status device_transpose(hipStream_t s, aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz, int base,
                                   const aoclsparse_int *d_ptr, const aoclsparse_int *d_ind, const void *d_val, size_t vsize,
                                   DeviceBuffer &tptr, DeviceBuffer &tind, DeviceBuffer &tval);

// ---- device mirrors / plans ---------------------------------------------------------------------
size_t            val_size(aoclsparse_matrix_data_type t);
aoclsparse_status upload_csr(const HostCsr &h, size_t vsize, DeviceCsr &d);
aoclsparse_status ensure_spmv(aoclsparse_matrix A, bool transposed, DeviceCsr *&dcsr,
                              SpmvPlan *&plan);
// symmetric expansion / triangular slice of the clean CSR as a general device CSR (derived.cpp)
aoclsparse_status ensure_derived(aoclsparse_matrix A, aoclsparse_matrix_type type, aoclsparse_fill_mode fill,
                                 aoclsparse_diag_type diag, bool transposed, Derived *&out);
// clean CSR on the device + level sets of one triangle (trsv_api.cpp)
// need_rows: also the level-ordered row layout (TrsvPlan::rows_valid); false = only what the automatic schedule needs
aoclsparse_status ensure_trsv(aoclsparse_matrix A, bool upper, bool transposed, bool conj = false, bool need_rows = true);
// SELL-64 copy of d (row_ptr_host = the host row pointer d mirrors); leaves plan.sell.valid false when the
// padding would exceed the budget (aoclsparse_mi355_set_option(aoclsparse_mi355_option_sell, 0 / 1): never / always)
aoclsparse_status build_sell(const aoclsparse_int *row_ptr_host, const DeviceCsr &d, size_t vsize, SpmvPlan &plan, bool complex_values = false);
// merge-path tiling (host binary searches over row_ptr_host); built when the longest row spans >= 32 LDS tiles, or when
// aoclsparse_mi355_set_option(aoclsparse_mi355_option_spmv_kernel, ...) asks for it
aoclsparse_status build_merge_plan(aoclsparse_int m, aoclsparse_int nnz, aoclsparse_index_base base,
                                   const aoclsparse_int *row_ptr_host, size_t vsize, SpmvPlan &plan);
aoclsparse_status build_spmv_plan(aoclsparse_int m, aoclsparse_int nnz, aoclsparse_index_base base,
                                  const aoclsparse_int *row_ptr_host, SpmvPlan &plan, size_t vsize = 8);

// ---- kernel launchers (HIP translation units) --------------------------------------------------
// order: 0 scalar, 1 lane4, 2 lane8
template <typename T>
aoclsparse_status launch_csrmv(hipStream_t s, int order, bool strict, int tile, int base, T alpha,
                               aoclsparse_int m, const T *val, const aoclsparse_int *col,
                               const aoclsparse_int *row_ptr, const aoclsparse_int *blocks,
                               aoclsparse_int nblocks, const T *x, T beta, T *y, const aoclsparse_int *blocks4 = nullptr,
                               aoclsparse_int max_row_nnz = 1 << 30, unsigned int *stale = nullptr);
// (stale != nullptr: the block table is a cached plan of a raw-array call -- every workgroup validates its own entry against
// the live row_ptr, computes its rows from the live arrays on a mismatch and sets *stale, a pinned host word)
template <typename T>
aoclsparse_status launch_sell_fill(hipStream_t s, int pack, aoclsparse_int m, int base, const aoclsparse_int *row_ptr,
                                   const aoclsparse_int *col, const T *val, aoclsparse_int nslices,
                                   const long long *slice_ptr, T *sval, aoclsparse_int *scol, aoclsparse_int *rowlen,
                                   const long long *cptr = nullptr, const unsigned short *lead = nullptr);
template <typename R>
aoclsparse_status launch_sellmv_complex(hipStream_t s, bool conj, cplx<R> alpha, aoclsparse_int m, aoclsparse_int nslices,
                                        const long long *slice_ptr, const cplx<R> *sval, const aoclsparse_int *scol,
                                        const aoclsparse_int *rowlen, const cplx<R> *x, cplx<R> beta, cplx<R> *y,
                                        const long long *cptr, const unsigned short *lead, aoclsparse_int max_width, int rev = 0);
aoclsparse_status launch_sell_leaders(hipStream_t s, aoclsparse_int m, int base, const aoclsparse_int *row_ptr, const aoclsparse_int *col,
                                      aoclsparse_int nslices, unsigned short *lead, aoclsparse_int *nl);
template <typename T>
aoclsparse_status launch_sellmv(hipStream_t s, int order, int pack, T alpha, aoclsparse_int m, aoclsparse_int nslices,
                                const long long *slice_ptr, const T *sval, const aoclsparse_int *scol,
                                const aoclsparse_int *rowlen, const T *x, T beta, T *y, const long long *cptr = nullptr,
                                const unsigned short *lead = nullptr,
                                aoclsparse_int max_width = 0, int rev = 0);
// BLKCSR (blk_kernels.hip): value offset of every block (three small launches: per-chunk popcount scan, scan of
// the chunk totals in part[], add), then the product
constexpr int     BLK_PART_SHIFT = 10;
aoclsparse_status launch_blk_valoff(hipStream_t s, aoclsparse_int nblk, int rows, const uint8_t *masks,
                                    aoclsparse_int *valoff, aoclsparse_int *part);
aoclsparse_status launch_blkcsrmv(hipStream_t s, int base, double alpha, aoclsparse_int m, int rows, const uint8_t *masks,
                                  const double *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                                  const aoclsparse_int *valoff, const double *x, double beta, double *y);
template <typename T>
aoclsparse_status launch_mergepath(hipStream_t s, int base, T alpha, aoclsparse_int ntiles, const aoclsparse_int *starts,
                                   const aoclsparse_int *first, const T *val, const aoclsparse_int *col,
                                   const aoclsparse_int *row_ptr, const T *x, T beta, T *y, void *pieces);
template <typename T>
aoclsparse_status launch_scale(hipStream_t s, T *y, aoclsparse_int n, T beta);
// w = a*x + b*y elementwise (w may alias x or y); a == 1, b == -1 is an exact subtraction
template <typename T>
aoclsparse_status launch_waxpby(hipStream_t s, aoclsparse_int n, T a, const T *x, T b, const T *y, T *w);
// d = x . y; partial must hold 1024 elements (spmv_kernels.hip)
template <typename T>
aoclsparse_status launch_dot(hipStream_t s, aoclsparse_int n, const T *x, const T *y, T *partial, T *d);
template <typename T>
aoclsparse_status launch_strided_gather(hipStream_t s, const T *src, aoclsparse_int inc,
                                        aoclsparse_int n, T *dst);
template <typename T>
aoclsparse_status launch_strided_scatter(hipStream_t s, const T *src, aoclsparse_int n, T *dst,
                                         aoclsparse_int inc);

// complex SpMV on a device CSR (complex_kernels.hip); conj multiplies by the conjugated matrix values
template <typename R>
aoclsparse_status launch_cspmv(hipStream_t s, int base, bool conj, cplx<R> alpha, aoclsparse_int m, aoclsparse_int nnz,
                               const cplx<R> *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                               const cplx<R> *x, cplx<R> beta, cplx<R> *y);
template <typename R>
aoclsparse_status launch_cscale(hipStream_t s, cplx<R> *y, aoclsparse_int n, cplx<R> beta);
template <typename R>
aoclsparse_status launch_ccsrmm(hipStream_t s, aoclsparse_order order, int base, bool conj, cplx<R> alpha,
                                aoclsparse_int m, const cplx<R> *val, const aoclsparse_int *col,
                                const aoclsparse_int *row_ptr, const cplx<R> *B, aoclsparse_int n, aoclsparse_int ldb,
                                cplx<R> beta, cplx<R> *C, aoclsparse_int ldc);
template <typename R>
aoclsparse_status launch_cscale_dense(hipStream_t s, aoclsparse_order order, cplx<R> *C, aoclsparse_int m,
                                      aoclsparse_int n, aoclsparse_int ld, cplx<R> beta);

// ELL family (ell_kernels.hip).  ellmv: row-major ELL, padding = column -1, double in the reference's
// 4-lane order (ellmv.hpp:90-208), float in its scalar order (:34-85).  elltmv: column-major ELL, one
// FMA chain per row (:316-444).  csr_rows: the CSR part of ELLT-HYB, rows listed in `map`, 4-lane
// order, beta term taken from ysrc[i] (:660-757).
template <typename T>
aoclsparse_status launch_ellmv(hipStream_t s, int base, T alpha, aoclsparse_int m, const T *val,
                               const aoclsparse_int *col, aoclsparse_int width, const T *x, T beta, T *y);
template <typename T>
aoclsparse_status launch_elltmv(hipStream_t s, int base, T alpha, aoclsparse_int m, const T *val,
                                const aoclsparse_int *col, aoclsparse_int width, const T *x, T beta, T *y);
template <typename T>
aoclsparse_status launch_csr_rows(hipStream_t s, int base, T alpha, aoclsparse_int nrows,
                                  const aoclsparse_int *map, const T *val, const aoclsparse_int *col,
                                  const aoclsparse_int *row_ptr, const T *x, T beta, const T *ysrc, T *y);
template <typename T>
aoclsparse_status launch_gather_rows(hipStream_t s, aoclsparse_int n, const aoclsparse_int *map, const T *src,
                                     T *dst);

// one dependency level of the in-place ILU(0) factorisation (ilu_kernels.hip): rows[0..nrows) of the level,
// diag[] = position of each finished row's diagonal, *error set on a bad pivot / missing diagonal
template <typename T>
aoclsparse_status launch_ilu0_level(hipStream_t s, int base, aoclsparse_int nrows, const aoclsparse_int *rows,
                                    const aoclsparse_int *row_ptr, const aoclsparse_int *col, T *val,
                                    aoclsparse_int *diag, int maxlen, int *error);
template <typename T>
aoclsparse_status launch_ilu0_syncfree(hipStream_t s, int base, aoclsparse_int n, const aoclsparse_int *rows,
                                       const aoclsparse_int *row_ptr, const aoclsparse_int *col, T *val, aoclsparse_int *diag,
                                       int maxlen, int *error, unsigned int *ticket);

// dense-vector steps of the iterative solvers (itsol_kernels.hip); `partial` holds
// vec_reduce_scratch_elems(k) elements, reduction results land in device memory
int vec_reduce_scratch_elems(int k);
template <typename T>
aoclsparse_status launch_cg_init(hipStream_t s, aoclsparse_int n, const T *b, const T *x, T *r, T *p);
template <typename T>
aoclsparse_status launch_vec_copy(hipStream_t s, aoclsparse_int n, const T *src, T *dst);
template <typename T>
aoclsparse_status launch_vec_add(hipStream_t s, aoclsparse_int n, const T *src, T *dst);
template <typename T>
aoclsparse_status launch_vec_mul(hipStream_t s, aoclsparse_int n, const T *src, T *dst);
template <typename T>
aoclsparse_status launch_vec_fill(hipStream_t s, aoclsparse_int n, T *dst, T value);
template <typename T>
aoclsparse_status launch_cg_direction(hipStream_t s, aoclsparse_int n, T beta, T *p, const T *z);
template <typename T>
aoclsparse_status launch_cg_step(hipStream_t s, aoclsparse_int n, T alpha, const T *p, const T *q, T *x, T *r,
                                 T *partial, T *rr);
template <typename T>
aoclsparse_status launch_cg_step_dev(hipStream_t s, aoclsparse_int n, T rz, T tiny, const T *p, const T *q, T *x, T *r,
                                     T *partial, T *out2);
template <typename T>
aoclsparse_status launch_multidot(hipStream_t s, aoclsparse_int n, int k, const T *V, long long ld, const T *w,
                                  T *partial, T *out);
template <typename T>
aoclsparse_status launch_lincomb(hipStream_t s, int sign, aoclsparse_int n, int k, const T *c, const T *V,
                                 long long ld, T *w);

// TRSV on the level-ordered layout (trsv_kernels.hip).
// schedule 0: one launch per level; 1: hybrid (narrow level runs inside one workgroup); 2: sync-free, a lane per
// position; 3: sync-free, a level slice per wavefront (single right-hand side; falls back to 2 otherwise);
// 4: sync-free, a lane per BLOCK of chained rows (plan.blk, real types, one right-hand side; falls back to 3 / 2).
// csrmm with beta == 0: false (default) = C is read and multiplied by zero as in every reference kernel (NaN / Inf in C
// propagate); true = C is overwritten without being read (BLAS semantics, a third less traffic at 256 columns).  One word for the
// process (aoclsparse_mi355_set_csrmm_beta0_overwrite; AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE seeds it once, on first touch).
std::atomic<bool> &csrmm_beta0_overwrite_flag();
// process-wide plan options (aoclsparse_mi355_set_option; read when a plan is built): tests and measurements only
int plan_option(aoclsparse_mi355_option option);
inline int SellPlan::next_direction() const
{
    if(plan_option(aoclsparse_mi355_option_alternate_sweeps) == 0)
        return 0;
    return (int)(products.fetch_add(1u, std::memory_order_relaxed) & 1u);
}
// does a csrmm kernel read C?  always for beta != 0; for beta == 0 unless the overwrite mode is on
bool csrmm_reads_c(bool beta_nonzero);
// timeout_word: where a sync-free kernel reports an expired wait (pinned host memory, Runtime::trsv_timeout_dev).
constexpr int TRSV_NARROW = 1024; // a level this narrow is solved by one workgroup (one row per lane)
template <typename T>
aoclsparse_status launch_trsv(hipStream_t s, int schedule, bool unit, T alpha, aoclsparse_int m,
                              const TrsvPlan &plan, const T *diag, const T *b, T *x, T *xp,
                              unsigned int *scratch, aoclsparse_int nrhs, long long b_off, aoclsparse_int incb,
                              long long x_off, aoclsparse_int incx, unsigned int *timeout_word = nullptr, int kt_bits = 0);

template <typename R>
aoclsparse_status launch_cvec_mul(hipStream_t s, aoclsparse_int n, const cplx<R> *d, cplx<R> *y);
template <typename R>
aoclsparse_status launch_cdiff(hipStream_t s, aoclsparse_int n, const cplx<R> *x, const cplx<R> *y, cplx<R> *w);
// d = sum conj(x_i) y_i; partial holds 1024 elements
template <typename R>
aoclsparse_status launch_cdot(hipStream_t s, aoclsparse_int n, const cplx<R> *x, const cplx<R> *y, cplx<R> *partial,
                              cplx<R> *d, bool conj_x = true);
template <typename R>
aoclsparse_status launch_caxpby(hipStream_t s, aoclsparse_int n, cplx<R> a, const cplx<R> *x, cplx<R> b, const cplx<R> *y,
                                cplx<R> *w);
// row bins of spgemm_hash_kernel (spgemm_kernels.hip): list capacity 32 / 256 / 2048 / 8192 in LDS; the last bin keeps its tables
// in a global slab.  The fill pass has no 8192 bin (values and slots would not fit the LDS): rows with more than 2048 entries of
// C go to the last bin there.
constexpr int SPGEMM_BINS = 5;
int           spgemm_bin_of(long long entries, bool fill);
// one row of the last bin: its hash table of 2^logh slots starts at h_off (g_key / g_pos), its list at c_off (g_list / g_acc)
struct SpgHeavy
{
    int       row, logh;
    long long h_off, c_off;
};
// the analysis around the two passes, on the device (spgemm_kernels.hip): upper bounds, bin histogram (+ a validity word), the
// rows of every bin, the prefix sum of the counts
aoclsparse_status launch_spg_bounds(hipStream_t s, aoclsparse_int m, aoclsparse_int n, int base_a, const aoclsparse_int *ptr_a,
                                    const aoclsparse_int *ind_a, const aoclsparse_int *ptr_b, int *cap);
aoclsparse_status launch_spg_diff(hipStream_t s, aoclsparse_int m, const aoclsparse_int *ptr, int *key);
aoclsparse_status launch_spg_hist(hipStream_t s, aoclsparse_int m, const int *key, const int *limit, bool fill, unsigned int *hist);
aoclsparse_status launch_spg_order(hipStream_t s, aoclsparse_int m, const int *key, bool fill, const aoclsparse_int *bounds,
                                   unsigned int *cursor, aoclsparse_int *order);
aoclsparse_status launch_spg_gather(hipStream_t s, aoclsparse_int count, const aoclsparse_int *ids, const int *key, int *out);
size_t            spg_scan_scratch_bytes(aoclsparse_int m);
aoclsparse_status launch_spg_scan(hipStream_t s, aoclsparse_int m, const int *cnt, aoclsparse_int *ptr, long long *scratch,
                                  long long **total_dev);
// complex triangular solve (complex_kernels.hip): the hybrid schedule of the plan (runs of narrow levels inside one
// workgroup, one launch per wide level); conj_diag for op = H (the plan's values are stored conjugated)
template <typename R>
aoclsparse_status launch_ctrsv(hipStream_t s, bool unit, bool conj_diag, cplx<R> alpha, aoclsparse_int m,
                               const TrsvPlan &plan, const cplx<R> *diag, const cplx<R> *b, cplx<R> *x, cplx<R> *xp,
                               aoclsparse_int nrhs, long long b_off, aoclsparse_int incb, long long x_off,
                               aoclsparse_int incx);

template <typename T>
aoclsparse_status launch_csrmm(hipStream_t s, aoclsparse_order order, int base, T alpha,
                               aoclsparse_int m, aoclsparse_int k, const T *val,
                               const aoclsparse_int *col, const aoclsparse_int *row_ptr, const T *B,
                               aoclsparse_int n, aoclsparse_int ldb, T beta, T *C,
                               aoclsparse_int ldc, const aoclsparse_int *grp = nullptr, aoclsparse_int ngroups = 0,
                               int group_rows = 0, bool row_runs = false,
                               const aoclsparse_int *run_order = nullptr, int kt_lanes = 0, aoclsparse_int band = 0);
// column-major detour, handles with row groups: row-major B scratch in, column-major C written directly (no C copies)
template <typename T>
bool csrmm_groups_ccol_applies(aoclsparse_int n, aoclsparse_int ldb, const T *B);
template <typename T>
aoclsparse_status launch_csrmm_groups_ccol(hipStream_t s, int base, T alpha, const T *val, const aoclsparse_int *col,
                                           const aoclsparse_int *row_ptr, const T *B, aoclsparse_int n, aoclsparse_int ldb,
                                           T beta, T *C, aoclsparse_int ldc, const aoclsparse_int *grp, aoclsparse_int ngroups,
                                           int group_rows, aoclsparse_int m = 0, aoclsparse_int band = 0);
// row-major, n < 128: workgroup per row block of the handle's SpMV plan, A staged in LDS (csrmm_tile_kernel)
template <typename T>
bool csrmm_tiled_applies(aoclsparse_int n, aoclsparse_int ldb, aoclsparse_int ldc, const T *B, const T *C);
template <typename T>
aoclsparse_status launch_csrmm_tiled(hipStream_t s, int base, T alpha, const T *val, const aoclsparse_int *col,
                                     const aoclsparse_int *row_ptr, const aoclsparse_int *blocks, aoclsparse_int nblocks,
                                     int tile, aoclsparse_int max_row_nnz, const T *B, aoclsparse_int n, aoclsparse_int ldb,
                                     T beta, T *C, aoclsparse_int ldc, int kt_lanes = 0, bool launch_order = false);
// column-major: a lane owns a row PAIR; 16-byte loads where the second row is the first shifted by one column
template <typename T>
aoclsparse_status launch_csrmm_colpair(hipStream_t s, int base, T alpha, aoclsparse_int npairs,
                                       const aoclsparse_int *pair_first, aoclsparse_int nsingles,
                                       const aoclsparse_int *single_rows, const T *val, const aoclsparse_int *col,
                                       const aoclsparse_int *row_ptr, aoclsparse_int max_row_nnz, const T *B, aoclsparse_int n,
                                       aoclsparse_int ldb, T beta, T *C, aoclsparse_int ldc);
// all csrmm plans of the untransposed matrix, built from the host arrays now (csrmm_api.cpp)
aoclsparse_status prepare_mm_plans(aoclsparse_matrix A);
// row-major, block-dense matrices: blocked-ELL copy + v_mfma_f64_16x16x4_f64 (csrmm_bell_kernels.hip)
aoclsparse_status build_bell(const HostCsr &h, const DeviceCsr &d, SpmvPlan &plan, aoclsparse_matrix_data_type vt);
aoclsparse_status launch_bell_fill(hipStream_t s, aoclsparse_int m, int base, const aoclsparse_int *ptr, const aoclsparse_int *ind,
                                   const double *val, aoclsparse_int nbr, aoclsparse_int width, const aoclsparse_int *bcol, double *out);
// (base / rp / ci / cv: the device CSR the copy was built from -- an element whose tile sum is not finite is recomputed from it)
aoclsparse_status launch_csrmm_bell(hipStream_t s, double alpha, aoclsparse_int m, aoclsparse_int k, const BellPlan &bell,
                                    int base, const aoclsparse_int *rp, const aoclsparse_int *ci, const double *cv,
                                    const double *B, aoclsparse_int n, aoclsparse_int ldb, double beta, double *C,
                                    aoclsparse_int ldc, bool column_major = false);
// column-major, banded: a workgroup stages the stretch of a B column its rows can touch in LDS (csrmm_window_kernels.hip)
int csrmm_window_rows(aoclsparse_int max_row_nnz, size_t elem); // rows per workgroup the kernel will use
int csrmm_window_max_pieces(); // 16-byte pieces a window may hold
template <typename T>
bool csrmm_window_applies(aoclsparse_int n, aoclsparse_int ldb, const T *B);
template <typename T>
aoclsparse_status launch_csrmm_window(hipStream_t s, int base, T alpha, aoclsparse_int m, const T *val, const aoclsparse_int *col,
                                      const aoclsparse_int *row_ptr, const aoclsparse_int *win, int win_rows,
                                      aoclsparse_int max_row_nnz, const T *B, aoclsparse_int n, aoclsparse_int ldb, T beta, T *C,
                                      aoclsparse_int ldc);
// aoclsparse_?csrmm_kid with kid 1/2/3: the reference's KT kernels' arithmetic (csrmm_kt.cpp:31-363), lanes = 4 / 8 (double), 8 / 16 (float)
template <typename T>
aoclsparse_status launch_csrmm_kt(hipStream_t s, aoclsparse_order order, int lanes, int base, T alpha, aoclsparse_int m,
                                  const T *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr, const T *B,
                                  aoclsparse_int n, aoclsparse_int ldb, T beta, T *C, aoclsparse_int ldc);
template <typename T>
aoclsparse_status launch_relayout(hipStream_t s, bool to_row_major, const T *src, T *dst, aoclsparse_int R, aoclsparse_int N,
                                  aoclsparse_int ld);
template <typename T>
aoclsparse_status launch_scale_dense(hipStream_t s, aoclsparse_order order, T *C, aoclsparse_int m,
                                     aoclsparse_int n, aoclsparse_int ld, T beta);

// sp2md_kernels.hip: dense-result product, CSR -> dense, sparse sum
template <typename T>
aoclsparse_status launch_dense_scale(hipStream_t s, T *C, aoclsparse_int inner, aoclsparse_int outer, long long ld,
                                     T beta, bool zero);
template <typename T>
aoclsparse_status launch_sp2md(hipStream_t s, aoclsparse_int m, int base_a, const aoclsparse_int *ptr_a,
                               const aoclsparse_int *ind_a, const T *val_a, bool conj_a, int base_b,
                               const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b, const T *val_b, bool conj_b,
                               T alpha, T *C, long long rs, long long cs);
template <typename T>
aoclsparse_status launch_csr2dense(hipStream_t s, aoclsparse_int m, int base, const aoclsparse_int *ptr,
                                   const aoclsparse_int *ind, const T *val, T *A, long long rs, long long cs, int mode,
                                   int fill, int diag);
template <typename T>
aoclsparse_status launch_csradd(hipStream_t s, bool fill, aoclsparse_int m, int base_a, const aoclsparse_int *ptr_a,
                                const aoclsparse_int *ind_a, const T *val_a, bool conj_a, T alpha, int base_b,
                                const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b, const T *val_b, int base_c,
                                const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c);
// a new handle that owns host CSR arrays (sp2m_api.cpp); row_ptr copied when given, else filled with `base`
aoclsparse_status new_csr_result(aoclsparse_matrix *C, aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                 aoclsparse_matrix_data_type vt, const aoclsparse_int *row_ptr,
                                 aoclsparse_index_base base = aoclsparse_index_base_zero);

// level1_kernels.hip: sparse-vector operations
constexpr int L1_DOT_PARTIALS = 1024;
template <typename T>
aoclsparse_status launch_axpyi(hipStream_t s, aoclsparse_int nnz, T a, const T *x, const aoclsparse_int *indx, T *y);
template <typename T>
aoclsparse_status launch_gather_scatter(hipStream_t s, aoclsparse_int nnz, T *x, const aoclsparse_int *indx,
                                        long long stride, T *y, int mode);
template <typename T>
aoclsparse_status launch_roti(hipStream_t s, aoclsparse_int nnz, T *x, const aoclsparse_int *indx, T *y, T c, T sn);
template <typename T>
aoclsparse_status launch_doti(hipStream_t s, aoclsparse_int nnz, const T *x, const aoclsparse_int *indx, const T *y,
                              bool conj, T *partial, T *out);

// sorv_kernels.hip: one level of the forward SOR sweep
template <typename T>
aoclsparse_status launch_sorv_level(hipStream_t s, const aoclsparse_int *rows, aoclsparse_int count, int base,
                                    const aoclsparse_int *ptr, const aoclsparse_int *ind, const T *val, T omega, T *x,
                                    const T *xold, const T *b);

// SpMV plan constants shared by host planner and kernels
// LDS tile = non-zeros staged per workgroup: 512 (128 threads), 1024 or 2048 (256 threads);
// rows per stream block (their row_ptr slice is kept in LDS)
// csr_adaptive_kernel, auto mode (scalar order, no pinned kid): rows of an LDS tile with at least this many entries are summed by a
// whole wavefront (64 strided FMA chains + a fixed-order tree) instead of one lane's chain
constexpr int SPMV_TREE_MIN = 32;
constexpr int spmv_maxrows(int tile)
{
    return tile / 2 < 512 ? tile / 2 : 512;
}

} // namespace mi355
