// comm_api.cpp -- one process per GPU: shipping a handle's ANALYSED device state to the other ranks, and the RCCL communicator
// the library itself can own (round 4; BASELINE north_star: "host code stays C++ ... B's columns sharded across the 8 GPUs of one
// node via RCCL broadcast/all-gather over xGMI").
//
// The reference has no counterpart: its column split of csrmm is a thread split inside one call
// (library/src/level3/aoclsparse_csrmm_kt.cpp:68-82) and A is simply shared memory.  Across processes A has to travel once, and
// what should travel is the device format -- CSR arrays in HBM plus every csrmm plan (row blocks, row groups, row runs, column
// windows, row pairs, the blocked-ELL copy) -- so that the receiving ranks do no analysis at all (SURVEY.md section 8e:
// "A replicated: broadcast from rank 0 once at optimize time, in its device format").
//
//   aoclsparse_mi355_mm_state_export / _adopt   the state as a fixed table of device buffers + a POD of scalars; the wire is the
//                                               caller's (torch.distributed in aocl-sparse_amd/sharded.py: RCCL with "nccl",
//                                               CPU tensors with "gloo", so the path is testable with two ranks on one GPU)
//   aoclsparse_mi355_comm_*                     the same over a communicator the LIBRARY owns: ncclGetUniqueId /
//                                               ncclCommInitRank / ncclBroadcast / ncclAllGather, librccl.so loaded with dlopen
//                                               on first use (a single-GPU user never pays for it and the library keeps
//                                               libamdhip64 as its only link-time dependency)
#include "internal.hpp"

#include <dlfcn.h>

#include <cstring>

using namespace mi355;

namespace
{
    // ---- the state table -------------------------------------------------------------------------------------------------
    // buffer i of the table, for a handle's untransposed device CSR + plan
    DeviceBuffer *state_buffer(_aoclsparse_matrix &A, int i)
    {
        SpmvPlan &p = A.plan_user;
        switch(i)
        {
        case 0: return &A.dev_user.ptr;
        case 1: return &A.dev_user.ind;
        case 2: return &A.dev_user.val;
        case 3: return &p.rowblocks;
        case 4: return &p.rowblocks4;
        case 5: return &p.mm.run_order;
        case 6: return &p.mm.first;
        case 7: return &p.mm.pair_first;
        case 8: return &p.mm.single_rows;
        case 9: return &p.mm.windows;
        case 10: return &p.bell.val;
        case 11: return &p.bell.bcol;
        case 12: return &p.bell.order;
        case 13: return &p.mm.slab_blocks;
        default: return nullptr;
        }
    }
    static_assert(AOCLSPARSE_MI355_MM_STATE_BUFFERS == 14, "state table and header disagree");

    enum
    {
        S_MAGIC = 0,
        S_M,
        S_N,
        S_NNZ,
        S_BASE,
        S_VTYPE,
        S_SORT,
        S_FULLDIAG,
        S_NBLOCKS,
        S_LONG_ROWS,
        S_MAX_ROW,
        S_TILE,
        S_HEAVY_FIRST,
        S_RUNS,
        S_BAND,
        S_NGROUPS,
        S_MAX_ROWS,
        S_GROUPS_VALID,
        S_PAIRS,
        S_NPAIRS,
        S_NSINGLES,
        S_WIN,
        S_WIN_ROWS,
        S_BELL,
        S_BELL_NBR,
        S_BELL_WIDTH,
        S_BELL_NBLOCKS,
        S_BELL_FILL_BITS,
        S_BELL_ORDER_LEN,
        S_BELL_XCD_CHUNK,
        S_SLAB_NBLOCKS,
        S_COUNT
    };
    static_assert(S_COUNT <= AOCLSPARSE_MI355_MM_STATE_SCALARS, "scalar table too small");
    constexpr long long STATE_MAGIC = 0x6d69333535723664LL; // "mi355r6d"

    void fill_scalars(const _aoclsparse_matrix &A, aoclsparse_mi355_mm_state &st)
    {
        std::memset(&st, 0, sizeof(st));
        const SpmvPlan &p = A.plan_user;
        long long      *s = st.scalars;
        s[S_MAGIC] = STATE_MAGIC, s[S_M] = A.m, s[S_N] = A.n, s[S_NNZ] = A.nnz, s[S_BASE] = A.base, s[S_VTYPE] = A.val_type;
        s[S_SORT] = A.sort, s[S_FULLDIAG] = A.fulldiag;
        s[S_NBLOCKS] = p.nblocks, s[S_LONG_ROWS] = p.long_rows, s[S_MAX_ROW] = p.max_row_nnz, s[S_TILE] = p.tile;
        s[S_HEAVY_FIRST] = p.heavy_first;
        s[S_RUNS] = p.mm.row_runs, s[S_BAND] = p.mm.band, s[S_NGROUPS] = p.mm.ngroups, s[S_MAX_ROWS] = p.mm.max_rows;
        s[S_GROUPS_VALID] = p.mm.valid, s[S_PAIRS] = p.mm.pairs, s[S_NPAIRS] = p.mm.npairs, s[S_NSINGLES] = p.mm.nsingles;
        s[S_WIN] = p.mm.win, s[S_WIN_ROWS] = p.mm.win_rows;
        s[S_BELL] = p.bell.valid, s[S_BELL_NBR] = p.bell.nbr, s[S_BELL_WIDTH] = p.bell.width, s[S_BELL_NBLOCKS] = p.bell.nblocks;
        std::memcpy(&s[S_BELL_FILL_BITS], &p.bell.fill, sizeof(double));
        s[S_BELL_ORDER_LEN] = p.bell.order_len, s[S_BELL_XCD_CHUNK] = p.bell.xcd_chunk;
        s[S_SLAB_NBLOCKS] = p.mm.slab_nblocks;
    }

    bool state_ok(const aoclsparse_mi355_mm_state &st)
    {
        const long long *s = st.scalars;
        if(s[S_MAGIC] != STATE_MAGIC || s[S_M] < 0 || s[S_N] < 0 || s[S_NNZ] < 0 || s[S_M] > 2147483647LL || s[S_N] > 2147483647LL
           || s[S_NNZ] > 2147483647LL || (s[S_BASE] != 0 && s[S_BASE] != 1))
            return false;
        if(s[S_VTYPE] != aoclsparse_dmat && s[S_VTYPE] != aoclsparse_smat)
            return false;
        const long long vs = (long long)val_size((aoclsparse_matrix_data_type)s[S_VTYPE]), I = (long long)sizeof(aoclsparse_int);
        if(st.bytes[0] != I * (s[S_M] + 1) || st.bytes[1] != I * s[S_NNZ] || st.bytes[2] != vs * s[S_NNZ])
            return false;
        // every plan the scalars announce must come with a buffer of at least the size the kernels will index (a truncated or
        // mismatched transfer is refused here, not found by a kernel); the CONTENTS are the sender's -- a state is only ever
        // produced by _export of this library (the magic word names the layout version)
        const long long *b = st.bytes;
        const long long  m = s[S_M], nb = s[S_NBLOCKS];
        if(nb < 0 || nb > m + 1 || (nb > 0 && b[3] < 2 * I * (nb + 1)))
            return false;
        if(nb > 0 && (s[S_TILE] & ~1LL) != 512 && (s[S_TILE] & ~1LL) != 1024 && (s[S_TILE] & ~1LL) != 2048)
            return false;
        if(s[S_HEAVY_FIRST] && b[4] < 4 * I * nb)
            return false;
        if(s[S_RUNS] && s[S_BAND] > 0 && b[5] < I * ((m + 7) / 8))
            return false;
        if(s[S_SLAB_NBLOCKS] < 0 || s[S_SLAB_NBLOCKS] > m || (s[S_SLAB_NBLOCKS] > 0 && (!s[S_RUNS] || s[S_BAND] <= 0 || b[13] < 2 * I * (s[S_SLAB_NBLOCKS] + 1))))
            return false;
        if(s[S_BAND] < 0 || s[S_NGROUPS] < 0 || s[S_NGROUPS] > m || s[S_MAX_ROWS] < 0 || s[S_MAX_ROWS] > 64)
            return false;
        if(s[S_GROUPS_VALID] && b[6] < I * (s[S_NGROUPS] + 1))
            return false;
        if(s[S_NPAIRS] < 0 || s[S_NSINGLES] < 0 || 2 * s[S_NPAIRS] + s[S_NSINGLES] > m)
            return false;
        if(s[S_PAIRS] && (b[7] < I * s[S_NPAIRS] || b[8] < I * s[S_NSINGLES]))
            return false;
        if(s[S_WIN])
        {
            const long long wr = s[S_WIN_ROWS];
            if((wr != 2048 && wr != 4096) || b[9] < 2 * I * ((m + wr - 1) / wr))
                return false;
        }
        if(s[S_BELL])
        {
            const long long nbr = s[S_BELL_NBR], w = s[S_BELL_WIDTH];
            if(s[S_VTYPE] != aoclsparse_dmat || nbr != (m + BELL_BS - 1) / BELL_BS || w < 1 || w > (1LL << 30) / (nbr > 0 ? nbr : 1)
               || b[10] < nbr * w * BELL_BS * BELL_BS * vs || b[11] < nbr * w * I || s[S_BELL_NBLOCKS] < 0 || s[S_BELL_NBLOCKS] > nbr * w)
                return false;
            // the order list: at most one entry per block row and XCD position (entries are block rows or -1: checked by the kernel's
            // bounds on nothing -- the list is the sender's, like every other plan)
            if(s[S_BELL_ORDER_LEN] < 0 || s[S_BELL_ORDER_LEN] > nbr || (s[S_BELL_ORDER_LEN] > 0 && b[12] < 8 * I * s[S_BELL_ORDER_LEN]))
                return false;
        }
        return true;
    }

    // a fresh handle for an adopted state: owns its host arrays (filled by the caller), no hints yet
    aoclsparse_status new_adopted(aoclsparse_matrix *R, const aoclsparse_mi355_mm_state &st)
    {
        const long long *s = st.scalars;
        aoclsparse_status rc = new_csr_result(R, (aoclsparse_int)s[S_M], (aoclsparse_int)s[S_N], (aoclsparse_int)s[S_NNZ],
                                              (aoclsparse_matrix_data_type)s[S_VTYPE], nullptr, (aoclsparse_index_base)s[S_BASE]);
        if(rc != aoclsparse_status_success)
            return rc;
        (*R)->sort = (int)s[S_SORT], (*R)->fulldiag = s[S_FULLDIAG] != 0;
        return aoclsparse_status_success;
    }

    // after the device buffers of R hold the state: scalars, flags, the host copy of the CSR arrays, an optimized mm hint
    aoclsparse_status finish_adopted(_aoclsparse_matrix &R, const aoclsparse_mi355_mm_state &st, hipStream_t stream)
    {
        const long long *s = st.scalars;
        DeviceCsr       &d = R.dev_user;
        SpmvPlan        &p = R.plan_user;
        d.m = R.m, d.n = R.n, d.nnz = R.nnz, d.base = R.base;
        p.nblocks = (aoclsparse_int)s[S_NBLOCKS], p.long_rows = (aoclsparse_int)s[S_LONG_ROWS];
        p.max_row_nnz = (aoclsparse_int)s[S_MAX_ROW], p.tile = (aoclsparse_int)s[S_TILE], p.heavy_first = s[S_HEAVY_FIRST] != 0;
        MmGroups &g = p.mm;
        g.runs_tried = g.tried = g.pairs_tried = g.win_tried = true; // the sender did the analysis: this handle never starts one
        g.row_runs = s[S_RUNS] != 0, g.band = (aoclsparse_int)s[S_BAND], g.ngroups = (aoclsparse_int)s[S_NGROUPS];
        g.max_rows = (int)s[S_MAX_ROWS], g.valid = s[S_GROUPS_VALID] != 0, g.pairs = s[S_PAIRS] != 0;
        g.npairs = (aoclsparse_int)s[S_NPAIRS], g.nsingles = (aoclsparse_int)s[S_NSINGLES];
        g.win = s[S_WIN] != 0, g.win_rows = (int)s[S_WIN_ROWS];
        g.slab_nblocks = (aoclsparse_int)s[S_SLAB_NBLOCKS];
        BellPlan &b = p.bell;
        b.tried = true, b.valid = s[S_BELL] != 0, b.nbr = (aoclsparse_int)s[S_BELL_NBR], b.width = (aoclsparse_int)s[S_BELL_WIDTH];
        b.nblocks = s[S_BELL_NBLOCKS];
        std::memcpy(&b.fill, &s[S_BELL_FILL_BITS], sizeof(double));
        b.order_len = (aoclsparse_int)s[S_BELL_ORDER_LEN], b.xcd_chunk = (int)s[S_BELL_XCD_CHUNK];
        // the host view of the matrix (everything outside csrmm -- export, ?mv's SELL copy, TRSV analysis -- works on it)
        const size_t vs = val_size(R.val_type);
        MI355_HIP_TRY(hipMemcpyAsync(R.user.ptr, d.ptr.ptr, sizeof(aoclsparse_int) * ((size_t)R.m + 1), hipMemcpyDeviceToHost, stream));
        if(R.nnz > 0)
        {
            host_result_touch(R.user.ind, sizeof(aoclsparse_int) * (size_t)R.nnz); // (fresh arrays: see host_result_alloc)
            host_result_touch(R.user.val, vs * (size_t)R.nnz);
            MI355_HIP_TRY(hipMemcpyAsync(R.user.ind, d.ind.ptr, sizeof(aoclsparse_int) * (size_t)R.nnz, hipMemcpyDeviceToHost, stream));
            MI355_HIP_TRY(hipMemcpyAsync(R.user.val, d.val.ptr, vs * (size_t)R.nnz, hipMemcpyDeviceToHost, stream));
        }
        MI355_HIP_TRY(hipStreamSynchronize(stream));
        d.valid = p.valid = true;
        try
        {
            Hint h{};
            h.act = action_mm, h.trans = aoclsparse_operation_none, h.type = aoclsparse_matrix_type_general;
            h.fill = aoclsparse_fill_mode_lower, h.nop = 1, h.kid = -1, h.optimized = true;
            R.hints.push_back(h);
        }
        catch(const std::bad_alloc &)
        {
            return aoclsparse_status_memory_error;
        }
        return aoclsparse_status_success;
    }

    // ---- RCCL, loaded on first use -------------------------------------------------------------------------------------------
    struct Rccl
    {
        void *lib = nullptr;
        int (*GetVersion)(int *)                                                    = nullptr;
        int (*GetUniqueId)(void *)                                                  = nullptr;
        int (*CommInitRank)(void **, int, aoclsparse_mi355_comm_id, int)            = nullptr;
        int (*CommDestroy)(void *)                                                  = nullptr;
        int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
        int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t)      = nullptr;
        int (*GroupStart)()                                                         = nullptr;
        int (*GroupEnd)()                                                           = nullptr;
        const char *(*GetErrorString)(int)                                          = nullptr;
        void *comm  = nullptr;
        int   world = 0, rank = -1, version = 0;
    };
    Rccl       g_rccl;
    std::mutex g_rccl_lock;
    constexpr int NCCL_UINT8 = 1; // rccl.h:460

    bool rccl_load()
    {
        if(g_rccl.lib)
            return true;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if(!h)
            h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if(!h)
        {
            std::fprintf(stderr, "aoclsparse(mi355): librccl.so not found (%s)\n", dlerror());
            return false;
        }
#define MI355_SYM(field, name)                                          \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
    if(!g_rccl.field)                                                   \
    {                                                                   \
        std::fprintf(stderr, "aoclsparse(mi355): %s missing in librccl\n", name); \
        dlclose(h);                                                     \
        return false;                                                   \
    }
        MI355_SYM(GetVersion, "ncclGetVersion")
        MI355_SYM(GetUniqueId, "ncclGetUniqueId")
        MI355_SYM(CommInitRank, "ncclCommInitRank")
        MI355_SYM(CommDestroy, "ncclCommDestroy")
        MI355_SYM(Broadcast, "ncclBroadcast")
        MI355_SYM(AllGather, "ncclAllGather")
        MI355_SYM(GroupStart, "ncclGroupStart")
        MI355_SYM(GroupEnd, "ncclGroupEnd")
        MI355_SYM(GetErrorString, "ncclGetErrorString")
#undef MI355_SYM
        g_rccl.lib = h;
        (void)g_rccl.GetVersion(&g_rccl.version);
        return true;
    }

    aoclsparse_status rccl_status(int rc, const char *what)
    {
        if(rc == 0)
            return aoclsparse_status_success;
        std::fprintf(stderr, "aoclsparse(mi355): %s failed: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return aoclsparse_status_internal_error;
    }
} // namespace

extern "C" {

aoclsparse_status aoclsparse_mi355_mm_state_export(aoclsparse_matrix A, aoclsparse_mi355_mm_state *state,
                                                   const void *buffers[AOCLSPARSE_MI355_MM_STATE_BUFFERS])
{
    if(!A || !state || !buffers)
        return aoclsparse_status_invalid_pointer;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(A->val_type != aoclsparse_dmat && A->val_type != aoclsparse_smat)
        return aoclsparse_status_wrong_type;
    Runtime          &rt = Runtime::get();
    aoclsparse_status rc = rt.init();
    if(rc != aoclsparse_status_success)
        return rc;
    if((rc = prepare_mm_plans(A)) != aoclsparse_status_success)
        return rc;
    MI355_HIP_TRY(hipStreamSynchronize(rt.stream())); // the plan uploads: the caller reads the buffers from its own streams
    std::shared_lock<std::shared_mutex> r(A->guard);
    fill_scalars(*A, *state);
    for(int i = 0; i < AOCLSPARSE_MI355_MM_STATE_BUFFERS; i++)
    {
        const DeviceBuffer *b = state_buffer(*A, i);
        buffers[i]            = b->ptr;
        state->bytes[i]       = b->ptr ? (long long)b->bytes : 0;
    }
    // a DeviceBuffer may be larger than its content: the CSR arrays are what the receiver sizes its host copy by
    state->bytes[0] = (long long)sizeof(aoclsparse_int) * ((long long)A->m + 1);
    state->bytes[1] = (long long)sizeof(aoclsparse_int) * A->nnz;
    state->bytes[2] = (long long)val_size(A->val_type) * A->nnz;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_mm_state_adopt(aoclsparse_matrix *R, const aoclsparse_mi355_mm_state *state,
                                                  const void *const buffers[AOCLSPARSE_MI355_MM_STATE_BUFFERS])
{
    if(!R || !state || !buffers)
        return aoclsparse_status_invalid_pointer;
    *R = nullptr;
    if(!state_ok(*state))
        return aoclsparse_status_invalid_value;
    for(int i = 0; i < AOCLSPARSE_MI355_MM_STATE_BUFFERS; i++)
        if(state->bytes[i] < 0 || (state->bytes[i] > 0 && !buffers[i]))
            return aoclsparse_status_invalid_pointer;
    Runtime          &rt = Runtime::get();
    aoclsparse_status rc = rt.init();
    if(rc != aoclsparse_status_success)
        return rc;
    aoclsparse_matrix H = nullptr;
    if((rc = new_adopted(&H, *state)) != aoclsparse_status_success)
        return rc;
    for(int i = 0; i < AOCLSPARSE_MI355_MM_STATE_BUFFERS && rc == aoclsparse_status_success; i++)
    {
        DeviceBuffer *b = state_buffer(*H, i);
        if(state->bytes[i] == 0)
            continue;
        rc = b->alloc((size_t)state->bytes[i]);
        if(rc == aoclsparse_status_success
           && hipMemcpyAsync(b->ptr, buffers[i], (size_t)state->bytes[i], hipMemcpyDeviceToDevice, rt.stream()) != hipSuccess)
            rc = aoclsparse_status_internal_error;
    }
    if(rc == aoclsparse_status_success)
        rc = finish_adopted(*H, *state, rt.stream());
    if(rc != aoclsparse_status_success)
    {
        (void)hipGetLastError();
        aoclsparse_destroy(&H);
        return rc;
    }
    *R = H;
    return aoclsparse_status_success;
}

// ---- the library's own communicator ----------------------------------------------------------------------------------------
aoclsparse_status aoclsparse_mi355_comm_unique_id(aoclsparse_mi355_comm_id *id)
{
    if(!id)
        return aoclsparse_status_invalid_pointer;
    std::lock_guard<std::mutex> l(g_rccl_lock);
    if(!rccl_load())
        return aoclsparse_status_not_implemented;
    return rccl_status(g_rccl.GetUniqueId(id), "ncclGetUniqueId");
}

aoclsparse_status aoclsparse_mi355_comm_init(aoclsparse_int world, aoclsparse_int rank, const aoclsparse_mi355_comm_id *id)
{
    if(!id)
        return aoclsparse_status_invalid_pointer;
    if(world < 1 || rank < 0 || rank >= world)
        return aoclsparse_status_invalid_value;
    Runtime          &rt = Runtime::get();
    aoclsparse_status rc = rt.init(); // binds this thread to the library's device: the communicator lives on it
    if(rc != aoclsparse_status_success)
        return rc;
    std::lock_guard<std::mutex> l(g_rccl_lock);
    if(!rccl_load())
        return aoclsparse_status_not_implemented;
    if(g_rccl.comm)
        return aoclsparse_status_invalid_operation; // one communicator per process: destroy it first
    void *comm = nullptr;
    rc         = rccl_status(g_rccl.CommInitRank(&comm, (int)world, *id, (int)rank), "ncclCommInitRank");
    if(rc != aoclsparse_status_success)
        return rc;
    g_rccl.comm = comm, g_rccl.world = (int)world, g_rccl.rank = (int)rank;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_comm_destroy(void)
{
    std::lock_guard<std::mutex> l(g_rccl_lock);
    if(!g_rccl.comm)
        return aoclsparse_status_success;
    const int rc = g_rccl.CommDestroy(g_rccl.comm);
    g_rccl.comm = nullptr, g_rccl.world = 0, g_rccl.rank = -1;
    return rccl_status(rc, "ncclCommDestroy");
}

aoclsparse_status aoclsparse_mi355_comm_info(aoclsparse_int *world, aoclsparse_int *rank, aoclsparse_int *rccl_version)
{
    std::lock_guard<std::mutex> l(g_rccl_lock);
    if(world)
        *world = g_rccl.world;
    if(rank)
        *rank = g_rccl.rank;
    if(rccl_version)
        *rccl_version = g_rccl.version;
    return g_rccl.comm ? aoclsparse_status_success : aoclsparse_status_invalid_operation;
}

aoclsparse_status aoclsparse_mi355_comm_broadcast(void *device_buffer, size_t bytes, aoclsparse_int root)
{
    std::lock_guard<std::mutex> l(g_rccl_lock);
    if(!g_rccl.comm)
        return aoclsparse_status_invalid_operation;
    if(root < 0 || root >= g_rccl.world || (bytes && !device_buffer))
        return aoclsparse_status_invalid_value;
    if(!bytes)
        return aoclsparse_status_success;
    Runtime &rt = Runtime::get();
    return rccl_status(g_rccl.Broadcast(device_buffer, device_buffer, bytes, NCCL_UINT8, (int)root, g_rccl.comm, rt.stream()), "ncclBroadcast");
}

aoclsparse_status aoclsparse_mi355_comm_allgather(const void *send, void *recv, size_t bytes_per_rank)
{
    std::lock_guard<std::mutex> l(g_rccl_lock);
    if(!g_rccl.comm)
        return aoclsparse_status_invalid_operation;
    if(bytes_per_rank && (!send || !recv))
        return aoclsparse_status_invalid_pointer;
    if(!bytes_per_rank)
        return aoclsparse_status_success;
    Runtime &rt = Runtime::get();
    return rccl_status(g_rccl.AllGather(send, recv, bytes_per_rank, NCCL_UINT8, g_rccl.comm, rt.stream()), "ncclAllGather");
}

// rank `root` passes its handle; every other rank passes *A == NULL and receives a new handle (to be destroyed by the caller)
// whose device CSR arrays and csrmm plans are root's, byte for byte, and which has done no analysis.
aoclsparse_status aoclsparse_mi355_comm_broadcast_matrix(aoclsparse_matrix *A, aoclsparse_int root)
{
    if(!A)
        return aoclsparse_status_invalid_pointer;
    std::unique_lock<std::mutex> l(g_rccl_lock);
    if(!g_rccl.comm)
        return aoclsparse_status_invalid_operation;
    if(root < 0 || root >= g_rccl.world)
        return aoclsparse_status_invalid_value;
    const bool sender = g_rccl.rank == root;
    if(sender != (*A != nullptr))
        return aoclsparse_status_invalid_value;
    Rccl &r = g_rccl;
    l.unlock(); // (export takes the handle's locks; the communicator is only used by this call: one collective at a time is the caller's contract)
    Runtime          &rt = Runtime::get();
    aoclsparse_status rc = rt.init();
    if(rc != aoclsparse_status_success)
        return rc;
    hipStream_t               s = rt.stream();
    aoclsparse_mi355_mm_state st;
    std::memset(&st, 0, sizeof(st));
    const void *bufs[AOCLSPARSE_MI355_MM_STATE_BUFFERS] = {};
    // a failure on the sender must not leave the receivers waiting in a collective: the header always travels, with a status word
    aoclsparse_status send_rc = aoclsparse_status_success;
    if(sender)
    {
        send_rc = aoclsparse_mi355_mm_state_export(*A, &st, bufs);
        if(send_rc != aoclsparse_status_success)
            std::memset(&st, 0, sizeof(st));
        st.scalars[AOCLSPARSE_MI355_MM_STATE_SCALARS - 1] = (long long)send_rc;
    }
    DeviceBuffer hdr;
    if((rc = hdr.alloc(sizeof(st))) != aoclsparse_status_success)
        return rc;
    if(sender)
        MI355_HIP_TRY(hipMemcpyAsync(hdr.ptr, &st, sizeof(st), hipMemcpyHostToDevice, s));
    if((rc = rccl_status(r.Broadcast(hdr.ptr, hdr.ptr, sizeof(st), NCCL_UINT8, (int)root, r.comm, s), "ncclBroadcast(header)"))
       != aoclsparse_status_success)
        return rc;
    MI355_HIP_TRY(hipMemcpyAsync(&st, hdr.ptr, sizeof(st), hipMemcpyDeviceToHost, s));
    MI355_HIP_TRY(hipStreamSynchronize(s));
    if(st.scalars[AOCLSPARSE_MI355_MM_STATE_SCALARS - 1] != 0)
        return sender ? send_rc : aoclsparse_status_internal_error; // the sender could not export: nobody enters the data phase
    if(!state_ok(st))
        return aoclsparse_status_internal_error;
    aoclsparse_matrix H = sender ? *A : nullptr;
    if(!sender && (rc = new_adopted(&H, st)) != aoclsparse_status_success)
        return rc; // (the other ranks will block in the data phase: an allocation failure here is fatal for the job anyway)
    // data phase: one grouped launch of all broadcasts, straight into the receivers' own DeviceBuffers
    for(int i = 0; i < AOCLSPARSE_MI355_MM_STATE_BUFFERS && !sender && rc == aoclsparse_status_success; i++)
        if(st.bytes[i] > 0)
            rc = state_buffer(*H, i)->alloc((size_t)st.bytes[i]);
    if(rc == aoclsparse_status_success)
    {
        int nrc = r.GroupStart();
        for(int i = 0; i < AOCLSPARSE_MI355_MM_STATE_BUFFERS && nrc == 0; i++)
            if(st.bytes[i] > 0)
            {
                void *p = state_buffer(*H, i)->ptr;
                nrc     = r.Broadcast(p, p, (size_t)st.bytes[i], NCCL_UINT8, (int)root, r.comm, s);
            }
        const int erc = r.GroupEnd();
        rc            = rccl_status(nrc ? nrc : erc, "ncclBroadcast(matrix)");
    }
    if(rc == aoclsparse_status_success && !sender)
        rc = finish_adopted(*H, st, s);
    if(rc == aoclsparse_status_success && sender && hipStreamSynchronize(s) != hipSuccess)
        rc = aoclsparse_status_internal_error;
    if(rc != aoclsparse_status_success)
    {
        (void)hipGetLastError();
        if(!sender)
            aoclsparse_destroy(&H);
        return rc;
    }
    if(!sender)
        *A = H;
    return aoclsparse_status_success;
}

} // extern "C"
