// trsv_kernels.hip -- level-scheduled sparse triangular solve for gfx950.
//
// The reference solves row by row on one core (level2/aoclsparse_trsv_kr.hpp:38-222); the
// dependency DAG of the triangle is the only thing that orders rows, so rows of one LEVEL (all
// dependencies in earlier levels) are solved concurrently here.  Every row is still reduced by ONE
// lane as the reference's chain  xi = alpha*b_i; xi = fma(-a_ij, x_j, xi) in the reference's order;
// xi /= d, so x is bit-identical to ref_trsv_l / _u / _lth / _uth (kid 0) whatever the schedule.
//
// Data layout (built once per (fill, op) by trsv_api.cpp): the strict triangle re-laid out in level
// order.  Position k holds row rowmap[k]; its entries are [pptr[k], pptr[k+1]) of pind (POSITION of the
// row depended on) / pval, stored in chain order (L: left to right; U: left to right; L^T: descending source row, as the column sweep of
// ref_trsv_lth applies them; U^T: ascending).  A level is a contiguous slab, so per-level traffic is
// streaming and the next level can be prefetched before the current one has been published.
//
// Schedules:
//   0  one launch per level (rows of a level spread over the whole chip).
//   1  hybrid (default): runs of NARROW levels (<= 1024 rows each) execute inside ONE 1024-lane
//      workgroup that walks the levels with a workgroup barrier between them, exchanging x through an
//      LDS ring, instead of paying a ~3.5 us kernel boundary per level; wide levels get a chip-wide
//      launch each.
//   2  sync-free: ONE launch; rows are taken in level order and a lane polls x[col] until it is no
//      longer the NOT-READY tag.  x doubles as the flag (one naturally aligned 8-byte agent-scope store
//      per row): the data-tagged hand-off of MI355X_MICROARCH.md ("handoff-1to1"); relaxed agent-scope
//      atomics = sc1 loads/stores that bypass the non-coherent per-CU L1.  Logical block ids come from
//      an atomic ticket, so a block only waits on rows of blocks that already started.
//
// Algorithmic bytes per solve: 12 B per stored entry of the triangle + (4+4+8+8+8) B per row.  Bound:
// the dependency chain (levels x per-level latency), not HBM, unless levels are tens of thousands wide.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

template <typename T>
struct tag;
template <>
struct tag<double>
{
    using bits = unsigned long long;
    static constexpr bits value = 0x7FF8DEADBEEF0355ull; // quiet NaN with a payload no FP op produces
};
template <>
struct tag<float>
{
    using bits = unsigned int;
    static constexpr bits value = 0x7FC0D355u;
};

// what a result whose bits EQUAL the tag is published as (a NaN with that exact payload can only come from b or A)
template <typename T>
struct qnan_bits;
template <>
struct qnan_bits<double>
{
    static constexpr unsigned long long value = 0x7FF8000000000000ull;
};
template <>
struct qnan_bits<float>
{
    static constexpr unsigned int value = 0x7FC00000u;
};

__device__ __forceinline__ double neg_fma(double a, double b, double c)
{
    return fma(-a, b, c);
}
__device__ __forceinline__ float neg_fma(float a, float b, float c)
{
    return fmaf(-a, b, c);
}

template <typename T>
__global__ void trsv_fill_tag_kernel(T *x, long long m)
{
    using B           = typename tag<T>::bits;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < m)
        reinterpret_cast<B *>(x)[i] = tag<T>::value;
}

// All kernels work in LEVEL-ORDER ("position") space: xp[k] = x[rowmap[k]], and pind[] holds the
// POSITION of the row an entry depends on, always smaller than the position of the row it belongs to.
// Results are written twice: xp[k] (what later rows read) and x[rowmap[k]] (what the caller gets).

// ---- schedule 0 / wide levels: positions [first, first+count) -------------------------------------------
// Right-hand-side geometry shared by the kernels: column c of a multi-RHS solve (aoclsparse_?trsm loops
// trsv over the columns, level3/aoclsparse_trsm.hpp:150-158) lives at b + c*b_off with element stride
// incb, its solution at x + c*x_off with stride incx; blockIdx.y selects the column.  trsv is nrhs = 1.
struct RhsGeom
{
    long long b_off, x_off;
    int       incb, incx;
    int       cols_fast; // sync-free kernel only: blockIdx.x = column
};

template <typename T>
__global__ void trsv_level_kernel(aoclsparse_int first, aoclsparse_int count, aoclsparse_int m,
                                  const aoclsparse_int *__restrict__ rowmap,
                                  const aoclsparse_int *__restrict__ pptr,
                                  const aoclsparse_int *__restrict__ pind, const T *__restrict__ pval,
                                  const T *__restrict__ diag, const T *__restrict__ b, T *xp, T *x, T alpha,
                                  int unit, RhsGeom g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= count)
        return;
    const int c = blockIdx.y;
    b += c * g.b_off;
    x += c * g.x_off;
    xp += (size_t)c * m;
    const int k  = first + t;
    const int i  = rowmap[k];
    T         xi = alpha * b[(size_t)i * g.incb];
    const int s = pptr[k], e = pptr[k + 1];
    for(int p = s; p < e; p++)
        xi = neg_fma(pval[p], xp[pind[p]], xi);
    if(!unit)
        xi /= diag[i];
    xp[k]                  = xi;
    x[(size_t)i * g.incx] = xi;
}

// ---- schedule 1, narrow runs: one workgroup walks levels [l0, l1), one lane per row ----------------------
// The run's recent solutions live in an LDS ring of TRSV_RING positions, so a dependency on a recent
// level costs an LDS read (~100 ns) instead of a global round trip, and the level barrier is a bare
// s_barrier behind an LDS-only wait: global loads of the NEXT levels' matrix data (three-stage software
// pipeline: level bounds -> row id / entry range -> rhs, diagonal, first TRSV_PF entries) stay in flight
// across barriers.  Dependencies older than the ring are read from xp in global memory; this workgroup
// wrote them itself (same CU, so its L1 is coherent for them) at least TRSV_RING-TRSV_NARROW positions
// ago, and a full vmcnt(0) drain every TRSV_DRAIN levels bounds how long such a store can be pending.
constexpr unsigned long long TRSV_WAIT_TICKS = 500000000ull; // 5 s of s_memrealtime (100 MHz): sync-free wait budget
constexpr int TRSV_PF    = 8;
constexpr int TRSV_RING  = 8192; // positions kept in LDS (64 KiB fp64)
constexpr int TRSV_DRAIN = 4; // levels between full memory drains (< (RING-NARROW)/NARROW)

template <typename T>
__global__ __launch_bounds__(TRSV_NARROW) void trsv_multilevel_kernel(
    aoclsparse_int l0, aoclsparse_int l1, const aoclsparse_int *__restrict__ levels,
    const aoclsparse_int *__restrict__ rowmap, const aoclsparse_int *__restrict__ pptr,
    const aoclsparse_int *__restrict__ pind, const T *__restrict__ pval, const T *__restrict__ diag,
    const T *__restrict__ b, T *xp, T *x, T alpha, int unit)
{
    __shared__ T ring[TRSV_RING];
    const int    tid = threadIdx.x;
    auto stage_a = [&](int l) -> int {
        if(l >= l1)
            return -1;
        const int first = levels[l], cnt = levels[l + 1] - first;
        return tid < cnt ? first + tid : -1;
    };
    const int run_first = levels[l0]; // positions before this were solved by earlier launches
    int       kA = stage_a(l0 + 2);
    int       kB = stage_a(l0 + 1);
    int       kC = stage_a(l0);
    int       iB = -1, sB = 0, eB = 0;
    if(kB >= 0)
    {
        iB = rowmap[kB];
        sB = pptr[kB];
        eB = pptr[kB + 1];
    }
    int iC = -1, sC = 0, eC = 0;
    T   rhs = T(0), dg = T(1);
    T   pv[TRSV_PF];
    int pc[TRSV_PF];
    if(kC >= 0)
    {
        iC  = rowmap[kC];
        sC  = pptr[kC];
        eC  = pptr[kC + 1];
        rhs = alpha * b[iC];
        if(!unit)
            dg = diag[iC];
#pragma unroll
        for(int j = 0; j < TRSV_PF; j++)
            if(sC + j < eC)
            {
                pv[j] = pval[sC + j];
                pc[j] = pind[sC + j];
            }
    }
    for(int l = l0; l < l1; l++)
    {
        const int lfirst = levels[l];
        // positions >= lo are valid in the ring while this level is being written
        const int lo = max(run_first, lfirst - TRSV_RING + TRSV_NARROW);
        const int ci = iC, ck = kC, cs = sC, ce = eC;
        T         xi = rhs;
        const T   cd = dg;
        T         cv[TRSV_PF];
        int       cc[TRSV_PF];
#pragma unroll
        for(int j = 0; j < TRSV_PF; j++)
        {
            cv[j] = pv[j];
            cc[j] = pc[j];
        }
        // ---- prefetch batch for the next levels (independent global loads) ----
        const int kA2 = stage_a(l + 3);
        int       iB2 = -1, sB2 = 0, eB2 = 0;
        if(kA >= 0)
        {
            iB2 = rowmap[kA];
            sB2 = pptr[kA];
            eB2 = pptr[kA + 1];
        }
        kC = kB, iC = iB, sC = sB, eC = eB;
        if(iC >= 0)
        {
            rhs = alpha * b[iC];
            if(!unit)
                dg = diag[iC];
#pragma unroll
            for(int j = 0; j < TRSV_PF; j++)
                if(sC + j < eC)
                {
                    pv[j] = pval[sC + j];
                    pc[j] = pind[sC + j];
                }
        }
        kB = kA, iB = iB2, sB = sB2, eB = eB2;
        kA = kA2;
        // ---- solve my row of level l ----
        if(ci >= 0)
        {
#pragma unroll
            for(int j = 0; j < TRSV_PF; j++)
                if(cs + j < ce)
                {
                    const int q  = cc[j];
                    const T   xv = q >= lo ? ring[q & (TRSV_RING - 1)] : xp[q];
                    xi           = neg_fma(cv[j], xv, xi);
                }
            for(int p = cs + TRSV_PF; p < ce; p++)
            {
                const int q  = pind[p];
                const T   xv = q >= lo ? ring[q & (TRSV_RING - 1)] : xp[q];
                xi           = neg_fma(pval[p], xv, xi);
            }
            if(!unit)
                xi /= cd;
            ring[ck & (TRSV_RING - 1)] = xi;
            xp[ck]                     = xi;
            x[ci]                      = xi;
        }
        if(((l - l0) % TRSV_DRAIN) == TRSV_DRAIN - 1)
            __syncthreads(); // full drain: every xp store older than this is visible to the workgroup
        else
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); // LDS writes only
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    }
}

// ---- schedule 2: sync-free, two tiers ----------------------------------------------------------------------
// A workgroup owns TRSV_SF_BLOCK consecutive positions.  A dependency inside the workgroup's own range is
// polled in LDS (a tagged copy of the workgroup's slice of xp: ~100 ns per hop), anything older in global
// memory (sc1 / agent-scope, ~1.5-3 us per hop).  Rows are in level order, so most dependencies of a
// narrow-level DAG are the workgroup's own recent rows.
// Two shapes, chosen from the triangle's mean row length: (1024 lanes, 12 staged entries) for short rows
// and wide levels (more dependencies stay inside the workgroup), (512, 20) when rows carry more entries
// than 12 (ILU(0) of the shell-like matrix: 7.2 ms vs 10.5 ms; of the 2-D Laplacian: 2.1 ms vs 2.5 ms).
template <typename T, int TRSV_SF_BLOCK, int TRSV_SF_PF>
__global__ __launch_bounds__(TRSV_SF_BLOCK) void trsv_syncfree_kernel(
    aoclsparse_int m, const aoclsparse_int *__restrict__ rowmap, const aoclsparse_int *__restrict__ pptr,
    const aoclsparse_int *__restrict__ pind, const T *__restrict__ pval, const T *__restrict__ diag,
    const T *__restrict__ b, T *xp, T *x, T alpha, int unit, unsigned int *ticket, unsigned int *timeout_flag,
    RhsGeom g)
{
    using B = typename tag<T>::bits;
    __shared__ unsigned int s_bid;
    __shared__ B            s_x[TRSV_SF_BLOCK];
    // right-hand side: every column has its own ticket counter and xp slab.  Columns are the FAST grid dimension
    // (when the block count fits gridDim.y): the dispatcher then hands out the k-th workgroup of every column
    // together, so the independent chains of a multi-RHS solve advance side by side instead of one column's
    // waiting workgroups filling every CU before the next column starts.
    const int c = g.cols_fast ? blockIdx.x : blockIdx.y;
    b += c * g.b_off;
    x += c * g.x_off;
    xp += (size_t)c * m;
    ticket += c;
    // the first TRSV_SF_PF entries of each row, [entry][lane] so that a wavefront reads one bank row;
    // without this every entry of a row is a dependent global load on the critical path of its level
    __shared__ T   s_ev[TRSV_SF_PF][TRSV_SF_BLOCK];
    __shared__ int s_ec[TRSV_SF_PF][TRSV_SF_BLOCK];
    const int      tid = threadIdx.x;
    if(tid == 0)
        s_bid = atomicAdd(ticket, 1u);
    s_x[tid] = tag<T>::value;
    __syncthreads();
    const long long k0 = (long long)s_bid * TRSV_SF_BLOCK;
    const long long k  = k0 + tid;
    if(k >= m)
        return;
    const int i  = rowmap[k];
    const int p0 = pptr[k];
    const int pe = pptr[k + 1];
#pragma unroll
    for(int j = 0; j < TRSV_SF_PF; j++)
        if(p0 + j < pe)
        {
            s_ev[j][tid] = pval[p0 + j];
            s_ec[j][tid] = pind[p0 + j];
        }
    T    xi = alpha * b[(size_t)i * g.incb];
    T    dg = T(1);
    if(!unit)
        dg = diag[i];
    int  p    = p0;
    B   *xb   = reinterpret_cast<B *>(xp);
    bool done = false;
    // every lane keeps iterating until ITS row is published: a lane may wait on a row owned by
    // another lane of the same wavefront, so the store must happen inside the loop
    unsigned int       spins  = 0;
    unsigned long long t_wait = 0;
    while(!done)
    {
        if(p != pe)
        {
            const int  e = p - p0;
            const bool staged = e < TRSV_SF_PF;
            const int  q = staged ? s_ec[e][tid] : pind[p];
            B          bits;
            if(q >= k0)
                bits = __hip_atomic_load(&s_x[q - k0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                bits = __hip_atomic_load(&xb[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(bits != tag<T>::value)
            {
                T xv;
                __builtin_memcpy(&xv, &bits, sizeof(T));
                xi = neg_fma(staged ? s_ev[e][tid] : pval[p], xv, xi);
                p++;
                spins = 0, t_wait = 0;
            }
            else
            {
                if((++spins & 4095u) == 0)
                {
                    // bounded by WALL time (s_memrealtime, 100 MHz), not by a spin count: a long serial chain, a shared
                    // GPU or a profiler must not trip it
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if(t_wait == 0)
                        t_wait = now;
                    else if(now - t_wait > TRSV_WAIT_TICKS)
                    {
                        // never expected: report (host returns internal_error), publish nothing, keep the caller's x
                        __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return;
                    }
                }
                if(q < k0)
                    __builtin_amdgcn_s_sleep(1);
            }
        }
        if(p == pe)
        {
            if(!unit)
                xi /= dg;
            B out;
            __builtin_memcpy(&out, &xi, sizeof(T));
            if(out == tag<T>::value) // a NaN carrying the NOT-READY payload: publish the canonical quiet NaN instead
                out = qnan_bits<T>::value;
            __hip_atomic_store(&s_x[tid], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&xb[k], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x[(size_t)i * g.incx] = xi;
            done = true;
        }
    }
}

// ---- schedule 3: sync-free, one LEVEL SLICE per wavefront ------------------------------------------------------
// The plan cuts every level into slices of <= 64 consecutive positions (TrsvPlan::slices), so the 64 lanes of a
// wavefront never depend on each other and may advance in lockstep.  That buys what the lane-per-position kernel
// above cannot have: a lane issues the loads of ALL its staged dependencies at once (one memory round trip for
// everything that is already solved, instead of one per entry) and keeps entries and values in registers; only the
// entries still tagged NOT-READY are polled again, in chain order.  A workgroup takes TRSV_WV consecutive slices
// through an atomic ticket (about three levels of a 300-row-wide level structure): dependencies inside the workgroup
// are exchanged through LDS, the others with agent-scope (sc1) loads of the position-ordered xp[].
// Waits are bounded by WALL time (s_memrealtime, 100 MHz): a lane that times out raises the flag and leaves x untouched.
// A result whose bits equal the NOT-READY tag (a NaN carrying exactly that payload: only possible when b or A holds
// it) is published as the canonical quiet NaN, so a consumer can never mistake it for "not solved yet".

template <typename T, int WV, int PF>
__global__ __launch_bounds__(64 * WV) void trsv_slice_kernel(
    aoclsparse_int m, aoclsparse_int nslices, const aoclsparse_int *__restrict__ slices,
    const aoclsparse_int *__restrict__ rowmap, const aoclsparse_int *__restrict__ pptr,
    const aoclsparse_int *__restrict__ pind, const T *__restrict__ pval, const T *__restrict__ diag,
    const T *__restrict__ b, T *xp, T *x, T alpha, int unit, unsigned int *ticket, unsigned int *timeout_flag, int incb,
    int incx)
{
    using B = typename tag<T>::bits;
    __shared__ unsigned int s_bid;
    __shared__ B            s_x[64 * WV];
    const int tid = threadIdx.x;
    if(tid == 0)
        s_bid = atomicAdd(ticket, 1u);
    s_x[tid] = tag<T>::value;
    __syncthreads();
    const int w    = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sl   = (int)s_bid * WV + w;
    if(sl >= nslices)
        return;
    const int k0 = slices[(int)s_bid * WV]; // first position of this workgroup
    const int kf = slices[sl], kl = slices[sl + 1];
    const int k  = kf + (tid & 63);
    if(k >= kl)
        return;
    const int i = rowmap[k], p0 = pptr[k], pe = pptr[k + 1];
    const int n = pe - p0;
    T         v[PF];
    int       q[PF];
    B         bits[PF];
    B        *xb = reinterpret_cast<B *>(xp);
#pragma unroll
    for(int e = 0; e < PF; e++)
    {
        v[e] = T(0), q[e] = 0;
        if(e < n)
            v[e] = pval[p0 + e], q[e] = pind[p0 + e];
    }
    T xi = alpha * b[(size_t)i * incb];
    T dg = T(1);
    if(!unit)
        dg = diag[i];
    auto peek = [&](int qq) -> B {
        return qq >= k0 ? __hip_atomic_load(&s_x[qq - k0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                        : __hip_atomic_load(&xb[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // everything that is already solved arrives in ONE round trip
#pragma unroll
    for(int e = 0; e < PF; e++)
        bits[e] = e < n ? peek(q[e]) : B(0);
    unsigned long long t0   = 0;
    bool               dead = false;
    auto               wait = [&](int qq, B got) -> B {
        unsigned int spins = 0;
        while(got == tag<T>::value && !dead)
        {
            if(qq < k0)
                __builtin_amdgcn_s_sleep(1);
            got = peek(qq);
            if((++spins & 1023u) == 0)
            {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if(t0 == 0)
                    t0 = now;
                else if(now - t0 > TRSV_WAIT_TICKS)
                    dead = true;
            }
        }
        return got;
    };
    // staged entries in chain order.  (Re-reading all pending entries in one batch per pass with a back-off sleep was
    // tried and lost: 4.77 vs 1.69 ms on the Laplacian factor, 11.8 vs 13.2 ms on the shell-like one -- the sleeps put
    // their own latency on the critical path; profiles/r2/trsv_schedules.txt.)
#pragma unroll
    for(int e = 0; e < PF; e++)
        if(e < n)
        {
            const B got = wait(q[e], bits[e]);
            T       xv;
            __builtin_memcpy(&xv, &got, sizeof(T));
            xi = neg_fma(v[e], xv, xi);
        }
    for(int p = p0 + PF; p < pe && !dead; p++)
    {
        const int qq  = pind[p];
        const B   got = wait(qq, peek(qq));
        T         xv;
        __builtin_memcpy(&xv, &got, sizeof(T));
        xi = neg_fma(pval[p], xv, xi);
    }
    if(dead)
    {
        // never expected: report, publish nothing (dependants time out the same way), keep the caller's x
        __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if(!unit)
        xi /= dg;
    B out;
    __builtin_memcpy(&out, &xi, sizeof(T));
    if(out == tag<T>::value)
        out = qnan_bits<T>::value;
    __hip_atomic_store(&s_x[k - k0], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&xb[k], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x[(size_t)i * incx] = xi;
}

// scratch: nrhs ticket words followed by one timeout word (zeroed here for the sync-free schedule)
template <typename T>
aoclsparse_status launch_trsv(hipStream_t s, int schedule, bool unit, T alpha, aoclsparse_int m,
                              const TrsvPlan &plan, const T *diag, const T *b, T *x, T *xp, unsigned int *scratch,
                              aoclsparse_int nrhs, long long b_off, aoclsparse_int incb, long long x_off,
                              aoclsparse_int incx, unsigned int *timeout_word)
{
    if(m <= 0 || nrhs <= 0)
        return aoclsparse_status_success;
    const aoclsparse_int *rowmap = plan.rowmap.as<aoclsparse_int>();
    const aoclsparse_int *pptr   = plan.pptr.as<aoclsparse_int>();
    const aoclsparse_int *pind   = plan.pind.as<aoclsparse_int>();
    const T              *pval   = plan.pval.as<T>();
    RhsGeom               g{b_off, x_off, incb, incx, 0};
    if(schedule == 1 && (nrhs != 1 || incb != 1 || incx != 1))
        schedule = 2; // the single-workgroup runs of the hybrid schedule are single-RHS, unit stride
    if(schedule == 3 && (nrhs != 1 || plan.nslices <= 0))
        schedule = 2; // the slice kernel is single-RHS; trsm keeps the lane-per-position kernel
    auto level_launch = [&](aoclsparse_int l) {
        const aoclsparse_int first = plan.level_ptr[l], count = plan.level_ptr[l + 1] - first;
        const int            bs = count >= 256 ? 256 : 64;
        for(aoclsparse_int c0 = 0; c0 < nrhs; c0 += 65535)
        {
            const int nc = nrhs - c0 < 65535 ? nrhs - c0 : 65535;
            hipLaunchKernelGGL((trsv_level_kernel<T>), dim3((count + bs - 1) / bs, nc), dim3(bs), 0, s, first, count,
                               m, rowmap, pptr, pind, pval, diag, b + c0 * b_off, xp + (size_t)c0 * m, x + c0 * x_off,
                               alpha, (int)unit, g);
        }
    };
    if(schedule == 0)
    {
        for(aoclsparse_int l = 0; l < plan.nlevels; l++)
            level_launch(l);
    }
    else if(schedule == 1)
    {
        for(const TrsvSegment &sg : plan.segments)
        {
            if(sg.narrow)
                hipLaunchKernelGGL((trsv_multilevel_kernel<T>), dim3(1), dim3(TRSV_NARROW), 0, s, sg.l0, sg.l1,
                                   plan.levels.as<aoclsparse_int>(), rowmap, pptr, pind, pval, diag, b, xp, x, alpha,
                                   (int)unit);
            else
                for(aoclsparse_int l = sg.l0; l < sg.l1; l++)
                    level_launch(l);
        }
    }
    else if(schedule == 3)
    {
        // sync-free, one level slice per wavefront (single right-hand side)
        MI355_HIP_TRY(hipMemsetAsync(scratch, 0, 2 * sizeof(unsigned int), s));
        hipLaunchKernelGGL((trsv_fill_tag_kernel<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, xp, (long long)m);
        static const int wv_env = [] {
            const char *e = getenv("AOCLSPARSE_MI355_TRSV_WAVES");
            return e ? atoi(e) : 0;
        }();
        const bool  wide = (long long)plan.nnz_tri > 10LL * m;
        const int   wv   = wv_env == 4 || wv_env == 8 || wv_env == 16 ? wv_env : 16;
        const aoclsparse_int *sl = plan.slices.as<aoclsparse_int>();
        auto go = [&](auto wv_tag, auto pf_tag) {
            constexpr int WV = decltype(wv_tag)::value, PF = decltype(pf_tag)::value;
            const unsigned nblk = (unsigned)((plan.nslices + WV - 1) / WV);
            hipLaunchKernelGGL((trsv_slice_kernel<T, WV, PF>), dim3(nblk), dim3(64 * WV), 0, s, m, plan.nslices, sl, rowmap,
                               pptr, pind, pval, diag, b, xp, x, alpha, (int)unit, scratch, timeout_word ? timeout_word : scratch + 1,
                               (int)incb, (int)incx);
        };
        using I4 = std::integral_constant<int, 4>;
        using I8 = std::integral_constant<int, 8>;
        using I16 = std::integral_constant<int, 16>;
        using PW = std::integral_constant<int, 20>;
        using PN = std::integral_constant<int, 8>;
        if(wide)
            wv == 4 ? go(I4{}, PW{}) : wv == 8 ? go(I8{}, PW{}) : go(I16{}, PW{});
        else
            wv == 4 ? go(I4{}, PN{}) : wv == 8 ? go(I8{}, PN{}) : go(I16{}, PN{});
    }
    else
    {
        // sync-free: tag xp, reset tickets + timeout word, one launch over all right-hand sides
        MI355_HIP_TRY(hipMemsetAsync(scratch, 0, ((size_t)nrhs + 1) * sizeof(unsigned int), s));
        const long long total = (long long)m * nrhs;
        hipLaunchKernelGGL((trsv_fill_tag_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xp,
                           total);
        unsigned int *tmo = timeout_word ? timeout_word : scratch + nrhs;
        for(aoclsparse_int c0 = 0; c0 < nrhs; c0 += 65535)
        {
            const int nc = nrhs - c0 < 65535 ? nrhs - c0 : 65535;
            const bool     wide = (long long)plan.nnz_tri > 10LL * m;
            const unsigned nblk = (unsigned)((m + (wide ? 511 : 1023)) / (wide ? 512 : 1024));
            g.cols_fast         = nc > 1 && nblk <= 65535u;
            const dim3 grid     = g.cols_fast ? dim3(nc, nblk) : dim3(nblk, nc);
            if(wide)
                hipLaunchKernelGGL((trsv_syncfree_kernel<T, 512, 20>), grid, dim3(512), 0, s, m, rowmap, pptr, pind, pval,
                                   diag, b + c0 * b_off, xp + (size_t)c0 * m, x + c0 * x_off, alpha, (int)unit, scratch + c0,
                                   tmo, g);
            else
                hipLaunchKernelGGL((trsv_syncfree_kernel<T, 1024, 12>), grid, dim3(1024), 0, s, m, rowmap, pptr, pind,
                                   pval, diag, b + c0 * b_off, xp + (size_t)c0 * m, x + c0 * x_off, alpha, (int)unit,
                                   scratch + c0, tmo, g);
        }
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_trsv<double>(hipStream_t, int, bool, double, aoclsparse_int, const TrsvPlan &,
                                               const double *, const double *, double *, double *, unsigned int *,
                                               aoclsparse_int, long long, aoclsparse_int, long long, aoclsparse_int,
                                               unsigned int *);
template aoclsparse_status launch_trsv<float>(hipStream_t, int, bool, float, aoclsparse_int, const TrsvPlan &,
                                              const float *, const float *, float *, float *, unsigned int *,
                                              aoclsparse_int, long long, aoclsparse_int, long long, aoclsparse_int,
                                              unsigned int *);

} // namespace mi355
