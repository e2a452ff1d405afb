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
#include "kt_order.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

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
__device__ __forceinline__ double kt_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float kt_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}
__device__ __forceinline__ float neg_fma(float a, float b, float c)
{
    return fmaf(-a, b, c);
}

template <typename T>
__global__ void trsv_fill_tag_kernel(T *x, long long m, long long zero_at = -1)
{
    using B           = typename tag<T>::bits;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < m)
        reinterpret_cast<B *>(x)[i] = tag<T>::value;
    if(i == 0 && zero_at >= 0)
        x[zero_at] = T(0); // the block kernel's "no dependency here" slot
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

// TSZ = 0: the chain of ref_trsv_l / ref_trsv_u (kid 0).  TSZ = 4 / 8 (double), 8 / 16 (float): the row sequence of
// kt_trsv_l / kt_trsv_u for 256- / 512-bit vectors (level2/aoclsparse_trsv_kt.cpp:92-137, :324-371; kid 1/2 / kid 3):
// lane l of the vector accumulates entries l, l + TSZ, ... of the full groups, xi -= hsum; a remainder of TSZ - 1 entries
// is one zero-padded vector product (mul, hsum), any other remainder the scalar chain.
template <typename T, int TSZ>
__device__ __forceinline__ T trsv_row_chain(T xi, int s, int e, const aoclsparse_int *__restrict__ pind,
                                            const T *__restrict__ pval, const T *xp)
{
    if constexpr(TSZ == 0)
    {
        for(int p = s; p < e; p++)
            xi = neg_fma(pval[p], xp[pind[p]], xi);
        return xi;
    }
    else
    {
        const int cnt = e - s, rem = cnt % TSZ;
        T         acc[TSZ];
#pragma unroll
        for(int l = 0; l < TSZ; l++)
            acc[l] = T(0);
        int p = s;
        for(; p < e - rem; p += TSZ)
#pragma unroll
            for(int l = 0; l < TSZ; l++)
                acc[l] = kt_fma(pval[p + l], xp[pind[p + l]], acc[l]);
        if(cnt >= TSZ)
            xi -= kt_hsum<T, TSZ>(acc);
        if(rem == TSZ - 1)
        {
#pragma unroll
            for(int l = 0; l < TSZ - 1; l++)
                acc[l] = pval[p + l] * xp[pind[p + l]];
            acc[TSZ - 1] = T(0);
            xi -= kt_hsum<T, TSZ>(acc);
        }
        else
            for(; p < e; p++)
                xi = neg_fma(pval[p], xp[pind[p]], xi);
        return xi;
    }
}

template <typename T, int TSZ>
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
    xi = trsv_row_chain<T, TSZ>(xi, pptr[k], pptr[k + 1], pind, pval, xp);
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
template <typename T, int TRSV_SF_BLOCK, int TRSV_SF_PF, int TSZ = 0>
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
    // TSZ > 0: the KT order (trsv_kt.cpp:92-137) -- entries are still consumed one by one as their x arrives, but entry e of
    // the full groups goes to vector lane e % TSZ, the lanes are reduced with the reference's tree when the last full group
    // is in, a remainder of TSZ - 1 entries is a zero-padded product vector reduced the same way, any other remainder the
    // scalar chain.  (acc[] is updated through selects: a run-time register index would go to scratch memory.)
    constexpr int AN = TSZ > 0 ? TSZ : 1;
    T             acc[AN];
#pragma unroll
    for(int l = 0; l < AN; l++)
        acc[l] = T(0);
    const int cnt = pe - p0, full = TSZ > 0 ? cnt - cnt % AN : 0;
    const bool masked = TSZ > 0 && cnt % AN == AN - 1;
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
                const T av = staged ? s_ev[e][tid] : pval[p];
                if constexpr(TSZ == 0)
                    xi = neg_fma(av, xv, xi);
                else
                {
                    if(e < full)
                    {
                        const int l = e % TSZ;
#pragma unroll
                        for(int q2 = 0; q2 < TSZ; q2++)
                            acc[q2] = q2 == l ? kt_fma(av, xv, acc[q2]) : acc[q2];
                        if(e == full - 1)
                            xi -= kt_hsum<T, AN>(acc);
                    }
                    else if(masked)
                    {
                        const int l = e - full;
#pragma unroll
                        for(int q2 = 0; q2 < TSZ; q2++)
                            acc[q2] = q2 == l ? av * xv : acc[q2];
                        if(e == cnt - 1)
                        {
                            acc[TSZ - 1] = T(0);
                            xi -= kt_hsum<T, AN>(acc);
                        }
                    }
                    else
                        xi = neg_fma(av, xv, xi);
                }
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
    // ... but whenever a wait of this wavefront actually had to spin, the entries still pending behind it are re-read
    // in ONE batch: when the chain STARTS with the row solved last (U: ref_trsv_u walks a row left to right, nearest
    // first) every later entry has long been solved by then, and polling them one after the other cost a round trip
    // each -- 45 ms instead of 7.7 for the shell-like factor's U against its L.
#pragma unroll
    for(int e = 0; e < PF; e++)
    {
        const bool spun = e < n && bits[e] == tag<T>::value;
        if(e < n)
        {
            const B got = wait(q[e], bits[e]);
            T       xv;
            __builtin_memcpy(&xv, &got, sizeof(T));
            xi = neg_fma(v[e], xv, xi);
        }
        if(PF > 8 && __builtin_amdgcn_ballot_w64(spun) != 0) // (short rows: nothing to batch, and the Laplacian factor lost 40 %)
        {
#pragma unroll
            for(int e2 = e + 1; e2 < PF; e2++)
                if(e2 < n && bits[e2] == tag<T>::value)
                    bits[e2] = peek(q[e2]);
        }
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

// ---- schedule 4: sync-free, a lane per BLOCK of chained rows (supernodal; lower non-transposed) ------------------
// plan.blk groups consecutive rows r, r+1, ... where each row depends on exactly its predecessor's dependencies plus the
// predecessor (the dofs of one mesh node in an ILU(0) factor).  One lane solves a block: it waits ONCE for the block's
// external dependencies (those of its first row), then runs the rows back to back -- row a's chain is [the external
// entries in order, then rows 0..a-1 of the block], which is exactly its CSR order, so every x is bit-identical to the
// serial reference chain.  The dependency DAG is levelled per BLOCK: the shell-like factor has 1,101 block levels
// instead of 5,505 row levels, i.e. a fifth of the cross-CU hand-offs, and the 15 neighbour values of a node are
// fetched once instead of five times.
// One wavefront per workgroup = one slice of <= 64 blocks of ONE block level (lanes never depend on each other).
// Blocks are taken through an atomic ticket, so a wavefront only ever waits on blocks of wavefronts that have started.
//
// What the time of a block level is made of was measured with the per-slice trace below (tools/trsv_trace.py,
// profiles/r2/trsv_block_trace.txt) -- a single wavefront issues one instruction every ~2.6 cycles, so everything
// between "the last dependency is in" and "my values are published" is priced by its INSTRUCTION COUNT:
//   * everything that does not need the dependencies happens before the wait: the block's values are laid out in LDS
//     at compile-time slots ([lane][row a][entry e], zero where the block has no entry), so that after the wait the
//     rows are 53 ds_read_b128 at constant offsets + 90 FMAs, no address arithmetic, no selects (as first written, with
//     clamped indices and selects, that phase was ~1,500 instructions = 1.84 us per block level);
//   * pending dependencies are polled together, one round of loads per look;
//   * a wavefront does not look at all before the level two below its own is complete (the gate).
// Paired LDS reads of the row-by-row path (larger shapes).  Ordinary vector loads: round 2 issued them from inline asm
// without a wait in the same statement, which left their ordering against the C++ stores that fill the slots and against
// register copies to the register allocator; the compiler now sees the loads and places the waits itself.
template <typename T>
__device__ __forceinline__ void lds_read_pair(T (&d)[2], const T *slot)
{
    typedef T T2 __attribute__((ext_vector_type(2)));
    const T2  v = *reinterpret_cast<const T2 *>(slot);
    d[0] = v.x, d[1] = v.y;
}

// (trace builds) keeps the stamp that follows behind the arithmetic that produced v
template <typename T>
__device__ __forceinline__ void lds_read_landed(T &v)
{
    asm volatile("" : "+v"(v));
}

// slots per lane of the fixed layout: BS x EXT external + BS x BS internal values, padded so that a lane's stride is an
// odd number of (2 values): the paired reads of the 64 lanes then fall on distinct LDS banks
// Shapes up to (5 rows, 20 external entries) keep a block's values in registers across the wait.  (5, 20) is 130 values: 256 VGPRs + 96
// AGPRs as their overflow, one wavefront per SIMD -- still faster than the same blocks out of LDS (unstructured shell-like factor, same box:
// 3.42 against 3.72 ms; profiles/r6/trsv_chunk_experiments.txt).
constexpr int TRSV_BLK_REG_SLOTS = 130;
constexpr int trsv_blk_slots(int bs, int ext)
{
    const int sl = bs * ext + bs * (bs + (bs & 1)); // ext and the padded internal stride are even
    return (sl / 2) % 2 ? sl : sl + 2;
}

template <typename T, int BS, int EXT, bool FRONT, bool TRACE = false>
__global__ __launch_bounds__(64) void trsv_block_kernel(
    aoclsparse_int m, aoclsparse_int nslices, const aoclsparse_int *__restrict__ slices,
    const aoclsparse_int *__restrict__ bfirst, const aoclsparse_int *__restrict__ rowmap,
    const aoclsparse_int *__restrict__ pptr, const aoclsparse_int *__restrict__ pind, const T *__restrict__ pval,
    const T *__restrict__ diag, const T *__restrict__ b, T *xp, T *x, T alpha, int unit, unsigned int *ticket,
    unsigned int *timeout_flag, int incb, int incx, unsigned long long *trace, unsigned int *level_done, int gate,
    int nrhs, long long b_off, long long x_off, int nlevels)
{
    using B = typename tag<T>::bits;
    static_assert(EXT % 2 == 0, "paired LDS reads");
    // several right-hand sides (trsm): the column is the FAST grid dimension, so the resident workgroups are the next
    // slices of EVERY column rather than all of one column's; each column has its own ticket, level counters and xp slab.
    // `spare` = elements from this column's xp slab to the spare slots behind the m x nrhs solution buffer.
    const int col   = nrhs > 1 ? (int)blockIdx.x : 0;
    const int spare = m * (nrhs - col);
    ticket += col, level_done += (size_t)col * nlevels;
    b += col * b_off, x += col * x_off, xp += (size_t)col * m;
    const bool tracing = TRACE && col == 0; // (a template parameter: run-time tests around the stamps cost the SpMV kernel 6 %)
    constexpr int BSP = BS + (BS & 1), SLOTS = trsv_blk_slots(BS, EXT), INT0 = BS * EXT; // internal values from slot INT0
    extern __shared__ unsigned char s_raw[];
    T            *s_mine = reinterpret_cast<T *>(s_raw) + (size_t)threadIdx.x * SLOTS;
    const int     tid    = threadIdx.x;
    unsigned int  tk     = 0;
    if(tid == 0)
        tk = atomicAdd(ticket, 1u);
    const int sl = __builtin_amdgcn_readfirstlane((int)tk);
    if(sl >= nslices)
        return;
    // (diagnostic stamps are kept in registers and stored at the very end: a store right after the wait would put its own
    // round trip into the phase it is timing)
    const unsigned long long t_start = TRACE ? __builtin_amdgcn_s_memrealtime() : 0;
    const int lev = slices[nslices + 1 + sl]; // block level of this slice
    const int bl  = slices[sl] + tid;
    // lanes beyond the slice own an empty block (c = 0): they run the same straight-line code and publish nothing
    const bool live = bl < slices[sl + 1];
    const int  k0   = live ? bfirst[bl] : 0;
    const int  c    = live ? bfirst[bl + 1] - k0 : 0; // positions [k0, k0 + c), c <= BS
    const int  p0   = live ? pptr[k0] : 0;
    const int  n0   = live ? pptr[k0 + 1] - p0 : 0; // external dependencies = the first row's entries
    const int  nl   = n0 < EXT ? n0 : EXT; // a single row may have more: the tail loop below
    B         *xb   = reinterpret_cast<B *>(xp);
    int        q[EXT];
#pragma unroll
    for(int e = 0; e < EXT; e++)
        q[e] = e < nl ? pind[p0 + e] : spare + TRSV_XP_PAD - 1; // beyond the row: a slot that always holds 0
    // larger shapes: the block's values -> their LDS slots (zero elsewhere)
    if constexpr(SLOTS > TRSV_BLK_REG_SLOTS)
    {
        for(int j = 0; j < SLOTS; j++)
            s_mine[j] = T(0);
        int p = p0;
        for(int a = 0; a < c; a++)
        {
            for(int e = 0; e < nl; e++)
                s_mine[a * EXT + e] = pval[p + (FRONT ? a : 0) + e];
            for(int tt = 0; tt < a; tt++)
                s_mine[INT0 + a * BSP + tt] = pval[p + (FRONT ? a - 1 - tt : n0 + tt)];
            p += n0 + a;
        }
    }
    // right-hand sides, diagonals, destinations
    T   rhs[BS], dg[BS];
    T  *xdst[BS];
    B  *bdst[BS]; // rows this lane does not own are parked behind the m positions (one slot per lane, never read)
#pragma unroll
    for(int a = 0; a < BS; a++)
    {
        const int row = a < c ? rowmap[k0 + a] : 0;
        rhs[a]        = a < c ? alpha * b[(size_t)row * incb] : T(0);
        dg[a]         = (a < c && !unit) ? diag[row] : T(1);
        xdst[a]       = a < c ? x + (size_t)row * incx : xp + (size_t)spare + 64 + tid;
        bdst[a]       = a < c ? xb + k0 + a : xb + (size_t)spare + tid;
    }
    unsigned long long t0   = 0;
    bool               dead = false;
    unsigned int       spins = 0;
    auto               tick  = [&](unsigned int every) {
        if((++spins & every) == 0)
        {
            // bounded by WALL time (s_memrealtime, 100 MHz), not by a spin count
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if(t0 == 0)
                t0 = now;
            else if(now - t0 > TRSV_WAIT_TICKS)
                dead = true;
        }
    };
    // Small shapes keep every value of the block in REGISTERS across the wait (read straight from the plan; the LDS copy
    // is then not needed at all): after the wait there is nothing left but FMAs and stores.
    constexpr bool IN_REGS = SLOTS <= TRSV_BLK_REG_SLOTS;
    T              ve[IN_REGS ? BS : 1][EXT], vn[IN_REGS ? BS : 1][BSP];
    if constexpr(IN_REGS)
    {
#pragma unroll
        for(int a = 0; a < BS; a++)
        {
            // row a = its n0 external entries and a internal ones: [external, rows 0..a-1], or, FRONT, [rows a-1..0,
            // external]; vn[a][tt] is the coefficient of block row tt either way
            const int off = p0 + a * n0 + (a * (a - 1)) / 2; // first entry of row a
#pragma unroll
            for(int e = 0; e < EXT; e++)
                ve[a][e] = (a < c && e < nl) ? pval[off + (FRONT ? a : 0) + e] : T(0);
#pragma unroll
            for(int tt = 0; tt < BSP; tt++)
                vn[a][tt] = (a < c && tt < a) ? pval[off + (FRONT ? a - 1 - tt : n0 + tt)] : T(0);
        }
    }
    // The gate: a wavefront does not look at its dependencies before block level (mine - gate) is complete -- until then
    // it polls ONE word (level_done, bumped by every finished slice).  Wavefronts are resident hundreds of levels ahead
    // of the front; with all of them polling 15 x 64 scattered lines per look, the looks of the few wavefronts that
    // matter queued behind ~800 k requests per round (measured: 4.07 ms without the gate, 3.64 ms with, same kernel).
    if(gate > 0 && lev >= gate)
    {
        const aoclsparse_int *lsl    = slices + 2 * (size_t)nslices + 1;
        const unsigned int    target = (unsigned int)(lsl[lev - gate + 1] - lsl[lev - gate]);
        while(__hip_atomic_load(&level_done[lev - gate], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && !dead)
        {
            __builtin_amdgcn_s_sleep(4);
            tick(255u);
        }
    }
    // All pending dependencies are polled TOGETHER: one round of loads per look, whatever the number outstanding (a
    // node's 10-15 external values are published by its neighbours at about the same time; polling them one after the
    // other put 15 trips on every block level, "wait for the last one, then re-read the rest" still two; two looks in
    // flight half a trip apart doubled the polling traffic and lost 0.9 ms).
    B bits[EXT];
#pragma unroll
    for(int e = 0; e < EXT; e++)
        bits[e] = __hip_atomic_load(&xb[q[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (entry e is re-read by the whole wavefront while ANY of its lanes still misses it: a uniform branch around an
    // unpredicated load.  Predicated per lane, the 16 exec-masked loads and the register copies hipcc wrapped around them
    // were ~200 instructions = 0.4 us per look on top of the trip itself; a value that has arrived never changes, so
    // reading it again is harmless.)
    for(;;)
    {
        unsigned long long miss[EXT], any = 0;
#pragma unroll
        for(int e = 0; e < EXT; e++)
        {
            miss[e] = __builtin_amdgcn_ballot_w64(bits[e] == tag<T>::value);
            any |= miss[e];
        }
        if(any == 0 || __builtin_amdgcn_ballot_w64(dead) != 0)
            break;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for(int e = 0; e < EXT; e++)
            if(miss[e] != 0)
                bits[e] = __hip_atomic_load(&xb[q[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tick(127u);
    }
    const unsigned long long t_ready = TRACE ? __builtin_amdgcn_s_memrealtime() : 0;
    // ---- from here on every instruction is on the critical path of the solve ----
    T xe[EXT];
#pragma unroll
    for(int e = 0; e < EXT; e++)
        __builtin_memcpy(&xe[e], &bits[e], sizeof(T)); // 0 beyond the row (bits = 0), times a 0 value below
    auto read_row = [&](int a, T(&ve)[EXT], T(&vn)[BSP]) {
#pragma unroll
        for(int e = 0; e < EXT; e += 2)
        {
            T pr[2];
            lds_read_pair(pr, &s_mine[a * EXT + e]);
            ve[e] = pr[0], ve[e + 1] = pr[1];
        }
#pragma unroll
        for(int tt = 0; tt < BSP; tt += 2)
            if(tt < a)
            {
                T pr[2];
                lds_read_pair(pr, &s_mine[INT0 + a * BSP + tt]);
                vn[tt] = pr[0], vn[tt + 1] = pr[1];
            }
    };
    // a single row with more than EXT dependencies (never inside a multi-row block): the rest one by one
    auto long_row_tail = [&](T &x0) {
        if(n0 > EXT)
            for(int p = p0 + EXT; p < p0 + n0 && !dead; p++)
            {
                const int qq  = pind[p];
                B         got = __hip_atomic_load(&xb[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while(got == tag<T>::value && !dead)
                {
                    __builtin_amdgcn_s_sleep(1);
                    got = __hip_atomic_load(&xb[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tick(1023u);
                }
                T xv;
                __builtin_memcpy(&xv, &got, sizeof(T));
                x0 = neg_fma(pval[p], xv, x0);
            }
    };
    auto publish = [&](int a, T xa) {
        if(a < c && !dead)
        {
            B out;
            __builtin_memcpy(&out, &xa, sizeof(T));
            if(out == tag<T>::value)
                out = qnan_bits<T>::value;
            __hip_atomic_store(bdst[a], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *xdst[a] = xa;
        }
    };
    T                  xi[BS];
    unsigned long long t_lds = 0, t_ext = 0;
    if constexpr(IN_REGS)
    {
        t_lds = TRACE ? __builtin_amdgcn_s_memrealtime() : 0;
        auto put = [&](int a) {
            B out;
            __builtin_memcpy(&out, &xi[a], sizeof(T));
            out = out == tag<T>::value ? qnan_bits<T>::value : out;
            __hip_atomic_store(bdst[a], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if(dead)
        {
#pragma unroll
            for(int a = 0; a < BS; a++)
                bdst[a] = xb + (size_t)spare + tid, xdst[a] = xp + (size_t)spare + 64 + tid;
        }
        if constexpr(FRONT)
        {
            // U: a row's chain STARTS with the rows of its own block (nearest first), then the external entries: one
            // chain per row, each starting when the row before is final
#pragma unroll
            for(int a = 0; a < BS; a++)
            {
                T xa = rhs[a];
#pragma unroll
                for(int tt = BS - 1; tt >= 0; tt--)
                    if(tt < a)
                        xa = neg_fma(vn[a][tt], xi[tt], xa);
#pragma unroll
                for(int e = 0; e < EXT; e++)
                    xa = neg_fma(ve[a][e], xe[e], xa);
                if(a == 0)
                    long_row_tail(xa);
                if(!unit)
                    xa /= dg[a];
                xi[a] = xa;
                put(a);
            }
        }
        else
        {
            // the external parts of the rows are BS independent chains, interleaved (x - 0 * 0 = x beyond a row's entries)
#pragma unroll
            for(int a = 0; a < BS; a++)
                xi[a] = rhs[a];
#pragma unroll
            for(int e = 0; e < EXT; e++)
#pragma unroll
                for(int a = 0; a < BS; a++)
                    xi[a] = neg_fma(ve[a][e], xe[e], xi[a]);
            long_row_tail(xi[0]);
            if constexpr(TRACE)
            {
                lds_read_landed(xi[BS - 1]);
                t_ext = __builtin_amdgcn_s_memrealtime();
            }
            // internal part, column by column: as soon as row tt is known it is taken out of every later row, so the
            // chain from the first row to the last is BS - 1 FMAs long, not BS (BS - 1) / 2 (each row still receives its
            // terms in CSR order).  Every row is published the moment it is final -- without a branch: rows (and lanes)
            // that own nothing store to a parked slot.  (Published together at the end, the last row queued behind the
            // other nine stores: +0.3 us per block level; with a predicated store after every row the compiler kept the
            // rows in separate basic blocks: 0.56 us for 10 FMAs.)
            if(unit)
            {
#pragma unroll
                for(int tt = 0; tt < BS; tt++)
                {
                    put(tt);
#pragma unroll
                    for(int a = tt + 1; a < BS; a++)
                        xi[a] = neg_fma(vn[a][tt], xi[tt], xi[a]);
                }
            }
            else
            {
#pragma unroll
                for(int tt = 0; tt < BS; tt++)
                {
                    xi[tt] /= dg[tt];
                    put(tt);
#pragma unroll
                    for(int a = tt + 1; a < BS; a++)
                        xi[a] = neg_fma(vn[a][tt], xi[tt], xi[a]);
                }
            }
        }
        // the caller's x: nobody waits for these
#pragma unroll
        for(int a = 0; a < BS; a++)
            *xdst[a] = xi[a];
    }
    else
    {
        // larger shapes: row by row (all rows at once would need BS x (EXT + BS) values in registers)
#pragma unroll
        for(int a = 0; a < BS; a++)
        {
            T ve[EXT], vn[BSP];
            read_row(a, ve, vn);
            xi[a] = rhs[a];
            if constexpr(FRONT)
            {
#pragma unroll
                for(int tt = BS - 1; tt >= 0; tt--)
                    if(tt < a)
                        xi[a] = neg_fma(vn[tt], xi[tt], xi[a]);
            }
#pragma unroll
            for(int e = 0; e < EXT; e++)
                xi[a] = neg_fma(ve[e], xe[e], xi[a]);
            if(a == 0)
                long_row_tail(xi[0]);
            if constexpr(!FRONT)
            {
#pragma unroll
                for(int tt = 0; tt < BS; tt++)
                    if(tt < a)
                        xi[a] = neg_fma(vn[tt], xi[tt], xi[a]);
            }
            if(!unit)
                xi[a] /= dg[a];
            publish(a, xi[a]);
        }
    }
    if(tracing && tid == 0)
    {
        unsigned long long *tr = trace + 6 * (size_t)sl;
        tr[0] = t_start, tr[1] = t_ready, tr[2] = __builtin_amdgcn_s_memrealtime(), tr[3] = (unsigned long long)lev;
        tr[4] = t_lds, tr[5] = t_ext;
    }
    if(tid == 0 && !dead)
        __hip_atomic_fetch_add(&level_done[lev], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(dead)
        __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- schedule 4 with the KT arithmetic (round 3) -----------------------------------------------------------------
// kid 1 / 2 / 3 ask for the summation order of the reference's KT kernels (kt_trsv_l / kt_trsv_u, trsv_kt.cpp:64-150, :297-383):
// full groups of TSZ entries into TSZ lane sums, the reference's horizontal tree, then a masked product vector for a remainder
// of TSZ - 1 or the scalar chain.  The lane-per-position kernel serves that order but consumes a row's dependencies strictly one
// after the other -- 15 dependent agent-scope loads per row even when all of them are published: 9-10 ms on the shell-like factor.
// This kernel keeps the block kernel's protocol (one lane per block of chained rows, ticket, gate, ALL external dependencies
// polled together) and replaces its fixed-shape FMA section by run-time loops over the block's entries, which sit in LDS in
// chain order ([entry][lane]: conflict-free whatever entry each lane is at) next to the x values they multiply.
template <typename T, int TSZ, bool FRONT>
__global__ __launch_bounds__(64) void trsv_block_kt_kernel(
    aoclsparse_int m, aoclsparse_int nslices, const aoclsparse_int *__restrict__ slices,
    const aoclsparse_int *__restrict__ bfirst, const aoclsparse_int *__restrict__ rowmap,
    const aoclsparse_int *__restrict__ pptr, const aoclsparse_int *__restrict__ pind, const T *__restrict__ pval,
    const T *__restrict__ diag, const T *__restrict__ b, T *xp, T *x, T alpha, int unit, unsigned int *ticket,
    unsigned int *timeout_flag, int incb, int incx, unsigned int *level_done, int gate, int nrhs, long long b_off,
    long long x_off, int nlevels, unsigned long long *trace)
{
    using B = typename tag<T>::bits;
    constexpr int BS = TRSV_BLK_ROWS, EXT = TRSV_BLK_EXT, NV = TRSV_BLK_NV;
    // (dynamic LDS: 64 x (NV + EXT + BS) values = 71 KB for double, past the 64 KB a kernel gets without asking)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_kt_raw[];
    T(*s_val)[64] = reinterpret_cast<T(*)[64]>(s_kt_raw); // the block's entries, rows back to back, each row in chain (= CSR) order
    // the x values a row multiplies, laid out so that row a's p-th entry meets s_xv[x0_a + p]: L (a row = [external in order,
    // rows 0..a-1]): external e at e, row a's result at n0 + a, x0_a = 0; U (FRONT, a row = [rows a-1..0, external]): row a's
    // result at BS - 1 - a, external e at BS + e, x0_a = BS - a.  Values and x are then walked with one induction variable.
    T(*s_xv)[64] = reinterpret_cast<T(*)[64]>(s_kt_raw + sizeof(T) * 64 * NV);
    const int col   = nrhs > 1 ? (int)blockIdx.x : 0;
    const int spare = m * (nrhs - col);
    ticket += col, level_done += (size_t)col * nlevels;
    b += col * b_off, x += col * x_off, xp += (size_t)col * m;
    const int    tid = threadIdx.x;
    unsigned int tk  = 0;
    if(tid == 0)
        tk = atomicAdd(ticket, 1u);
    const int sl = __builtin_amdgcn_readfirstlane((int)tk);
    if(sl >= nslices)
        return;
    const bool               tracing = trace != nullptr && col == 0; // (diagnostic: tools/trsv_trace.py)
    const unsigned long long t_start = tracing ? __builtin_amdgcn_s_memrealtime() : 0;
    const int  lev  = slices[nslices + 1 + sl];
    const int  bl   = slices[sl] + tid;
    const bool live = bl < slices[sl + 1];
    const int  k0   = live ? bfirst[bl] : 0;
    const int  c    = live ? bfirst[bl + 1] - k0 : 0;
    const int  p0   = live ? pptr[k0] : 0;
    const int  n0   = live ? pptr[k0 + 1] - p0 : 0; // external dependencies = the first row's entries
    const int  nl   = n0 < EXT ? n0 : EXT;
    const bool slow = n0 > EXT; // a single long row (never inside a multi-row block): its tail is polled entry by entry
    B         *xb   = reinterpret_cast<B *>(xp);
    int        q[EXT];
#pragma unroll
    for(int e = 0; e < EXT; e++)
        q[e] = e < nl ? pind[p0 + e] : spare + TRSV_XP_PAD - 1; // beyond the row: a slot that always holds 0
    // everything that does not need the dependencies: the block's values -> LDS (c n0 + c (c - 1) / 2 <= NV entries)
    const int tot = slow ? 0 : c * n0 + (c * (c - 1)) / 2;
    for(int j = 0; j < tot; j++)
        s_val[j][tid] = pval[p0 + j];
    T  rhs[BS], dg[BS];
    T *xdst[BS];
    B *bdst[BS];
#pragma unroll
    for(int a = 0; a < BS; a++)
    {
        const int row = a < c ? rowmap[k0 + a] : 0;
        rhs[a]        = a < c ? alpha * b[(size_t)row * incb] : T(0);
        dg[a]         = (a < c && !unit) ? diag[row] : T(1);
        xdst[a]       = a < c ? x + (size_t)row * incx : xp + (size_t)spare + 64 + tid;
        bdst[a]       = a < c ? xb + k0 + a : xb + (size_t)spare + tid;
    }
    unsigned long long t0    = 0;
    bool               dead  = false;
    unsigned int       spins = 0;
    auto               tick  = [&](unsigned int every) {
        if((++spins & every) == 0)
        {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if(t0 == 0)
                t0 = now;
            else if(now - t0 > TRSV_WAIT_TICKS)
                dead = true;
        }
    };
    if(gate > 0 && lev >= gate)
    {
        const aoclsparse_int *lsl    = slices + 2 * (size_t)nslices + 1;
        const unsigned int    target = (unsigned int)(lsl[lev - gate + 1] - lsl[lev - gate]);
        while(__hip_atomic_load(&level_done[lev - gate], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && !dead)
        {
            __builtin_amdgcn_s_sleep(4);
            tick(255u);
        }
    }
    B bits[EXT];
#pragma unroll
    for(int e = 0; e < EXT; e++)
        bits[e] = __hip_atomic_load(&xb[q[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for(;;)
    {
        unsigned long long miss[EXT], any = 0;
#pragma unroll
        for(int e = 0; e < EXT; e++)
        {
            miss[e] = __builtin_amdgcn_ballot_w64(bits[e] == tag<T>::value);
            any |= miss[e];
        }
        if(any == 0 || __builtin_amdgcn_ballot_w64(dead) != 0)
            break;
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for(int e = 0; e < EXT; e++)
            if(miss[e] != 0)
                bits[e] = __hip_atomic_load(&xb[q[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tick(127u);
    }
    const unsigned long long t_ready = tracing ? __builtin_amdgcn_s_memrealtime() : 0;
    // Everything loaded so far has long arrived; saying so here (vmcnt(0) lgkmcnt(0)) keeps the compiler from putting a
    // vmcnt(0) into the row loops below for values it can no longer track across their back edges -- on gfx9 that wait would
    // also cover the PUBLISH stores of the row before (stores count in vmcnt): ~0.7 us per row, measured 4.0 us per slice.
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for(int e = 0; e < EXT; e++)
    {
        T xv;
        __builtin_memcpy(&xv, &bits[e], sizeof(T));
        s_xv[FRONT ? BS + e : e][tid] = xv;
    }
    // A published value keeps its register until the kernel ends (okeep[], pinned below): on gfx9 a store's data register may
    // not be overwritten before the store has completed, and the only way the compiler can make sure is s_waitcnt vmcnt(0) --
    // it put one into every row's loop where the allocator had recycled the register of the row before: ~0.7 us per row.
    // The caller's x is written at the very end (nobody waits for it).
    B okeep[BS];
    T xkeep[BS];
#pragma unroll
    for(int a = 0; a < BS; a++)
        okeep[a] = 0, xkeep[a] = T(0);
    auto publish = [&](int a, T xa) {
        xkeep[a] = xa;
        __builtin_memcpy(&okeep[a], &xa, sizeof(T));
        if(okeep[a] == tag<T>::value)
            okeep[a] = qnan_bits<T>::value;
        if(!dead)
            __hip_atomic_store(bdst[a], okeep[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
#pragma unroll
    for(int a = 0; a < BS; a++)
        if(a < c)
        {
            const int cnt = n0 + a; // row a: n0 external entries and a internal ones
            T         xa  = rhs[a];
            T         pv[TSZ];
#pragma unroll
            for(int l = 0; l < TSZ; l++)
                pv[l] = T(0);
            if(!slow)
            {
                const T *vp = &s_val[a * n0 + (a * (a - 1)) / 2][tid]; // row a's p-th value at vp[64 p]
                const T *xq = &s_xv[FRONT ? BS - a : 0][tid]; // ... and the x it multiplies at xq[64 p]
                int      g  = 0;
                // (a single wavefront per SIMD cannot hide an LDS round trip: two groups of 4 per trip where the width is 4)
                if constexpr(TSZ <= 4)
                    for(; g + 2 * TSZ <= cnt; g += 2 * TSZ)
                    {
                        T va[2 * TSZ], xa2[2 * TSZ];
#pragma unroll
                        for(int l = 0; l < 2 * TSZ; l++)
                            va[l] = vp[64 * (g + l)], xa2[l] = xq[64 * (g + l)];
#pragma unroll
                        for(int l = 0; l < 2 * TSZ; l++)
                            pv[l % TSZ] = kt_fma(va[l], xa2[l], pv[l % TSZ]);
                    }
                for(; g + TSZ <= cnt; g += TSZ)
                {
#pragma unroll
                    for(int l = 0; l < TSZ; l++)
                        pv[l] = kt_fma(vp[64 * (g + l)], xq[64 * (g + l)], pv[l]);
                }
                if(cnt >= TSZ)
                    xa -= kt_hsum<T, TSZ>(pv);
                if(cnt - g == TSZ - 1)
                {
#pragma unroll
                    for(int l = 0; l < TSZ - 1; l++)
                        pv[l] = vp[64 * (g + l)] * xq[64 * (g + l)];
                    pv[TSZ - 1] = T(0);
                    xa -= kt_hsum<T, TSZ>(pv);
                }
                else
                    for(int p = g; p < cnt; p++)
                        xa = neg_fma(vp[64 * p], xq[64 * p], xa);
            }
            else
            {
                // the long single row: the first EXT values are in, the rest arrive one by one; lane sums through selects
                const int  full   = cnt - cnt % TSZ;
                const bool masked = cnt % TSZ == TSZ - 1;
                for(int e = 0; e < cnt && !dead; e++)
                {
                    T xv;
                    if(e < EXT)
                        xv = s_xv[FRONT ? BS + e : e][tid];
                    else
                    {
                        const int qq  = pind[p0 + e];
                        B         got = __hip_atomic_load(&xb[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        while(got == tag<T>::value && !dead)
                        {
                            __builtin_amdgcn_s_sleep(1);
                            got = __hip_atomic_load(&xb[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            tick(1023u);
                        }
                        __builtin_memcpy(&xv, &got, sizeof(T));
                    }
                    const T av = pval[p0 + e];
                    if(e < full)
                    {
                        const int l = e % TSZ;
#pragma unroll
                        for(int q2 = 0; q2 < TSZ; q2++)
                            pv[q2] = q2 == l ? kt_fma(av, xv, pv[q2]) : pv[q2];
                        if(e == full - 1)
                            xa -= kt_hsum<T, TSZ>(pv);
                    }
                    else if(masked)
                    {
                        const int l = e - full;
#pragma unroll
                        for(int q2 = 0; q2 < TSZ; q2++)
                            pv[q2] = q2 == l ? av * xv : pv[q2];
                        if(e == cnt - 1)
                        {
                            pv[TSZ - 1] = T(0);
                            xa -= kt_hsum<T, TSZ>(pv);
                        }
                    }
                    else
                        xa = neg_fma(av, xv, xa);
                }
                // (nothing of this branch is pending where the branches join: otherwise the rows that follow get a
                // vmcnt(0) for registers these loads might still be writing -- and that wait covers the publish stores)
                __builtin_amdgcn_s_waitcnt(0);
            }
            if(!unit)
                xa /= dg[a];
            if(!slow) // (the long single row has no successor inside its block, and n0 + a may lie beyond the array)
                s_xv[FRONT ? BS - 1 - a : n0 + a][tid] = xa;
            publish(a, xa);
        }
#pragma unroll
    for(int a = 0; a < BS; a++)
    {
        asm volatile("" ::"v"(okeep[a]), "v"(bdst[a])); // (see publish: data AND address registers of the pending stores)
        if(a < c && !dead)
            *xdst[a] = xkeep[a];
    }
    if(tracing && tid == 0)
    {
        unsigned long long *tr = trace + 6 * (size_t)sl;
        tr[0] = t_start, tr[1] = t_ready, tr[2] = __builtin_amdgcn_s_memrealtime(), tr[3] = (unsigned long long)lev;
        tr[4] = 0, tr[5] = 0;
    }
    if(tid == 0 && !dead)
        __hip_atomic_fetch_add(&level_done[lev], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(dead)
        __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- schedule 5: two levels -- chunks of consecutive blocks, hand-offs inside a chunk through LDS (round 6) ----------------
// The lane-per-block kernel above pays one hand-off through L2 / HBM per block level (1.2-1.7 us against 0.4 us of work: profiles/r5/
// trsv_experiments.txt), because consecutive levels always run in different wavefronts on different CUs.  Here the blocks keep their
// NATURAL order at the top level: a workgroup owns a CHUNK of consecutive blocks (internal.hpp: TrsvChunkPlan) and walks them in
// block-level order, one STEP (<= 8 blocks of one level) per wavefront, steps dealt round-robin to all but the last of its wavefronts (5 of 6).
// The chunk's x lives in LDS as NaN-tagged words -- exactly the protocol of the sync-free kernels, one level closer: a dependency
// on a row of the same chunk is polled in LDS (a hand-off costs an LDS write + read).  Rows of earlier chunks the chunk depends on
// (its HALO) are polled in xp by the LAST wavefront, in the order of their first use, and copied into LDS slots behind the chunk's
// own rows: the solving wavefronts never wait for an HBM round trip, and one wavefront per chunk polls HBM instead of all of them.
// A mesh numbered line by line puts most of a block's dependencies a few hundred blocks back, inside its chunk: the remote
// hand-offs that remain are one per chunk boundary along the critical path, and they are pipelined (chunk c + 1 runs one remote
// latency behind chunk c).  Chunks are taken through a ticket in natural order and depend on earlier chunks only.
// Inside a step a block owns 8 lanes, lane a = row a of the block: the external parts of the rows (the block's 16-24 dependencies
// times its rows) are independent chains and run in parallel lanes; their sums are gathered into the block's first lane (DPP
// row_shl), which eliminates the block's rows column by column in registers, exactly the order of the lane-per-block kernel -- row
// a's chain is [external entries in CSR order, then rows 0..a-1 of the block] -- so x is bit-identical to ref_trsv_* whatever the
// schedule.  Metadata of a wavefront's NEXT step is requested while it waits for the current one.
template <int R>
__device__ __forceinline__ int trsv_dpp_shl(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x100 + R, 0xf, 0xf, true); // lane l <- lane l + R of its row of 16 (0 beyond it)
}
template <int R>
__device__ __forceinline__ double trsv_dpp_shl(double v)
{
    return __hiloint2double(trsv_dpp_shl<R>(__double2hiint(v)), trsv_dpp_shl<R>(__double2loint(v)));
}
template <int R>
__device__ __forceinline__ float trsv_dpp_shl(float v)
{
    return __int_as_float(trsv_dpp_shl<R>(__float_as_int(v)));
}

// FRONT (U, not transposed): a row's chain STARTS with the rows of its own block (nearest first) and ends with the external entries,
// so row a cannot begin before row a - 1 is final: the rows of a block are BS phases one after the other.  Lane a keeps its own
// row's coefficients; in phase r every lane runs row r's chain shape, lane r's result is the block's x_r, broadcast to the block's
// eight lanes (ds_swizzle) for the phases that follow.
template <int R>
__device__ __forceinline__ double trsv_bcast8(double v)
{
    constexpr int pat = (R << 5) | 0x18; // swizzle(BROADCAST, 8, R): lane R of every group of 8
    return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), pat), __builtin_amdgcn_ds_swizzle(__double2loint(v), pat));
}
template <int R>
__device__ __forceinline__ float trsv_bcast8(float v)
{
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (R << 5) | 0x18));
}

template <typename T, int EXT, int BS, bool UNIT, bool FRONT>
__global__ __launch_bounds__(64 * TRSV_CHUNK_WAVES) void trsv_chunk_kernel(
    aoclsparse_int m, aoclsparse_int nnz, aoclsparse_int nchunks, const int4 *__restrict__ steps, const aoclsparse_int *__restrict__ cptr,
    const aoclsparse_int *__restrict__ rowmap, const aoclsparse_int *__restrict__ pptr, const T *__restrict__ pval,
    const aoclsparse_int *__restrict__ eptr, const aoclsparse_int *__restrict__ cind, const aoclsparse_int *__restrict__ hind,
    const T *__restrict__ diag, const T *__restrict__ b, T *xp, T *x, T alpha, unsigned int *ticket,
    unsigned int *timeout_flag, int incb, int incx, int nrhs, long long b_off, long long x_off, unsigned long long *trace, int dbg)
{
    using B          = typename tag<T>::bits;
    constexpr int BL = TRSV_CHUNK_LANES; // lanes per block
    constexpr int NW = TRSV_CHUNK_WAVES - 1; // wavefronts that take steps
    // (16-byte aligned: behind a 4-byte static variable the dynamic array started at offset 4, and every 8-byte word of the chunk
    // was a misaligned LDS access -- slow, and not the single access the tagged-word protocol needs)
    // NO static LDS variable next to it: behind a 4-byte one the dynamic array started at offset 4 whatever its declared alignment)
    extern __shared__ __align__(16) unsigned char s_raw[];
    int &s_chunk = *reinterpret_cast<int *>(s_raw); // the first 16 bytes: the chunk this workgroup drew
    // then one staging area per solving wavefront (the values and the dependency lists of ONE step, see level2 below), then the
    // chunk's words
    constexpr int PCAP = trsv_chunk_pcap(BS, EXT), ECAP = 8 * EXT + EXT, SCR = PCAP * (int)sizeof(T) + ECAP * 4;
    constexpr int PK = (PCAP + 63) / 64, EK = (ECAP + 63) / 64;
    B            *lx = reinterpret_cast<B *>(s_raw + 16 + (TRSV_CHUNK_WAVES - 1) * SCR); // [rows of the chunk][its halo]: tags / x; then one slot that holds 0; then 64 parked slots
    const int col = nrhs > 1 ? (int)blockIdx.x : 0;
    ticket += col;
    b += col * b_off, x += col * x_off, xp += (size_t)col * m;
    B        *xb  = reinterpret_cast<B *>(xp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane / BL, a = lane % BL; // block of the step, row of the block
    if(tid == 0)
        s_chunk = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int ch = s_chunk;
    if(ch >= nchunks)
        return;
    const int s0 = cptr[ch], s1 = cptr[ch + 1], nrows = cptr[nchunks + 1 + ch];
    const int hb = cptr[2 * nchunks + 1 + ch], nh = cptr[3 * nchunks + 1 + ch]; // this chunk's halo: hind[hb .. hb + nh)
    for(int i = tid; i < nrows + nh; i += 64 * TRSV_CHUNK_WAVES)
        lx[i] = tag<T>::value;
    if(tid == 0)
        lx[nrows + nh] = 0; // "no dependency here"
    __syncthreads();
    // (from here on the chunk's words are polled while other wavefronts write them: relaxed workgroup-scope atomics, so that the
    // compiler re-reads them; one 4- / 8-byte LDS access each)
    auto lds_get = [&](int slot) { return __hip_atomic_load(&lx[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_put = [&](int slot, B v) { __hip_atomic_store(&lx[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    const int          zslot = nrows + nh, park = zslot + 1 + lane; // the slot that holds 0; this lane's parked stores
    unsigned long long t0    = 0;
    bool               dead  = false;
    unsigned int       spins = 0;
    auto               tick  = [&](unsigned int every) {
        if((++spins & every) == 0)
        {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if(t0 == 0)
                t0 = now;
            else if(now - t0 > TRSV_WAIT_TICKS)
                dead = true;
        }
    };
    if(wave == NW)
    {
        // the fetching wavefront: the halo in ROUNDS of 256 entries (sorted by first use), lane l takes entries 4 l .. 4 l + 3 of
        // the round: their positions are one 16-byte load (requested a round ahead), their words four loads per look; what has
        // arrived goes to its LDS slot at once, the round ends when all of it has.  A look costs one HBM round trip whatever the
        // number of words still missing.  (First version: two entries per lane, each followed by the load of the next position --
        // two dependent round trips per entry, 30-40 entries per us; an unstructured mesh needs ~3,000 per chunk.)
        const int4 *h4    = reinterpret_cast<const int4 *>(hind); // (hb is a multiple of 4: the plan pads every chunk's list)
        const int   nr    = (nh + 255) / 256;
        int4        qn    = nr > 0 ? h4[hb / 4 + lane] : int4{0, 0, 0, 0};
        int         idle  = 0;
        for(int r = 0; r < nr && __builtin_amdgcn_ballot_w64(dead) == 0; r++)
        {
            const int4 q  = qn;
            const int  i0 = 256 * r + 4 * lane;
            if(r + 1 < nr)
                qn = h4[hb / 4 + 64 * (r + 1) + lane];
            unsigned need = (i0 < nh ? 1u : 0u) | (i0 + 1 < nh ? 2u : 0u) | (i0 + 2 < nh ? 4u : 0u) | (i0 + 3 < nh ? 8u : 0u);
            while(__builtin_amdgcn_ballot_w64(need != 0) != 0 && __builtin_amdgcn_ballot_w64(dead) == 0)
            {
                // (all four unconditionally: entries beyond the list hold position 0, a word that exists)
                const B v0 = __hip_atomic_load(&xb[q.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const B v1 = __hip_atomic_load(&xb[q.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const B v2 = __hip_atomic_load(&xb[q.z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const B v3 = __hip_atomic_load(&xb[q.w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned got = 0;
                if((need & 1u) && v0 != tag<T>::value)
                    lds_put(nrows + i0, v0), got |= 1u;
                if((need & 2u) && v1 != tag<T>::value)
                    lds_put(nrows + i0 + 1, v1), got |= 2u;
                if((need & 4u) && v2 != tag<T>::value)
                    lds_put(nrows + i0 + 2, v2), got |= 4u;
                if((need & 8u) && v3 != tag<T>::value)
                    lds_put(nrows + i0 + 3, v3), got |= 8u;
                need &= ~got;
                if(__builtin_amdgcn_ballot_w64(got != 0) != 0)
                    idle = 0;
                else if(++idle > 8) // nothing new for a while: this chunk is far ahead of its predecessors
                    __builtin_amdgcn_s_sleep(16);
                tick(255u);
            }
        }
        if(__builtin_amdgcn_ballot_w64(dead) != 0 && lane == 0)
            __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // per-lane metadata of a step, everything ONE load away from the step's header:
    struct Meta
    {
        int  mypos, myslot, c, row, pa, p0, p1, e0; // position / LDS slot of my row, rows of my block, row index, first entry of my
        int  pbeg, pend, ebeg, eend; // row, of the block's first row and of its second, offset of the block's dependency list;
        bool live; // (uniform) the step's range of pval and of cind
    };
    auto header = [&](int s, int4 &h0, int4 &h1) {
        const int sv = __builtin_amdgcn_readfirstlane(s); // (one step per wavefront: scalar loads; as vector loads 1 % slower)
        h0 = steps[2 * (size_t)sv], h1 = steps[2 * (size_t)sv + 1];
    };
    auto level1 = [&](const int4 &h0, const int4 &h1, Meta &mt) {
        const int c   = (int)(((unsigned)h0.w >> (4 * j)) & 15u);
        const int pre = (int)(((unsigned)(j < 4 ? h1.x : h1.y) >> (8 * (j & 3))) & 255u);
        mt.c          = c;
        mt.live       = a < c;
        const int k0  = h0.y + pre; // position of the block's first row
        mt.mypos      = mt.live ? k0 + a : 0;
        mt.myslot     = mt.live ? h0.z + pre + a : zslot;
        mt.row        = mt.live ? rowmap[k0 + a] : 0;
        mt.pa         = mt.live ? pptr[k0 + a] : 0;
        mt.p0         = c > 0 ? pptr[k0] : 0;
        mt.p1         = c > 0 ? pptr[k0 + 1] : 0;
        mt.e0         = c > 0 ? eptr[h0.x + j] : 0;
        mt.pbeg = pptr[h0.y], mt.pend = pptr[h0.y + h1.z]; // (h1.z: rows of the step, h1.w: its blocks)
        mt.ebeg = eptr[h0.x], mt.eend = eptr[h0.x + h1.w];
    };
    // Two steps of this wavefront are in flight: the CURRENT one, and the NEXT one, whose metadata is requested when the current
    // step starts and whose data when it ends (its header came a step earlier).  (A third step in flight -- data of the next one
    // requested at the START of the current one, metadata two steps ahead -- measured 8 % slower, 1.20 vs 1.10 ms on the shell-like
    // factor: 20 more registers and instructions in front of every first look; profiles/r6/trsv_chunk_experiments.txt.)
    int4 h0 = {0, 0, 0, 0}, h1 = h0, g0 = h0, g1 = h0;
    Meta cur = {0, zslot, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, false}, nxt;
    if(s0 + wave < s1)
    {
        header(s0 + wave, h0, h1);
        level1(h0, h1, cur);
    }
    nxt = cur;
    if(s0 + wave + NW < s1)
        header(s0 + wave + NW, g0, g1);
    // level 2 of a step: my row's external coefficients, the block's dependency list (LDS slots), right-hand side, diagonal, and --
    // for the block's first lane -- the coefficients of the rows inside the block, vn[r][tt] = row r on block row tt.  Every load
    // unconditional, from an address that exists, and masked when it is used (predicated loads become one basic block each, and
    // the compiler's waits at their joins serialised the batches).  Requested for the NEXT step of this wavefront as soon as the
    // current one has left its registers, IN FRONT of the current step's stores to HBM: vmcnt retires in order, and behind the
    // write-through stores of x the loads waited for the stores' acknowledgements too (2 us more before every first look).
    // Level 2 of a step: the values of its rows, the dependency lists of its blocks, right-hand sides, diagonals.
    // STAGED (every step whose data fits the wavefront's staging area: all but steps with a very long single row): a step's rows
    // are consecutive positions, so their values are ONE contiguous range of pval and their dependency lists one of cind -- the
    // wavefront fetches the ranges with coalesced loads (lane l takes elements l, l + 64, ...: a few cache lines per instruction)
    // and deals them out through its staging area when the step starts.  Fetched lane by lane -- lane (block, row) its own row's
    // entries, one entry per instruction -- every instruction touched ~40 different lines, ~950 per step: the solving wavefronts kept
    // the CU's address path busy for ~1 us per step, which was the rate of the whole solve (profiles/r6/trsv_chunk_trace.txt).
    T         bb = T(0), dd = T(1), rv[PK];
    int       re[EK];
    bool      staged = false, staged_cur = false;
    const int last   = nnz - 1;
    T        *scr_p  = reinterpret_cast<T *>(s_raw + 16 + wave * SCR);
    int      *scr_e  = reinterpret_cast<int *>(scr_p + PCAP);
    auto      level2 = [&](const Meta &mt) {
        staged = __builtin_amdgcn_readfirstlane((mt.pend - mt.pbeg <= PCAP - EXT - 16 && mt.eend - mt.ebeg <= 8 * EXT) ? 1 : 0) != 0;
        bb = b[(size_t)mt.row * incb];
        if constexpr(!UNIT)
            dd = diag[mt.row];
        if(staged)
        {
#pragma unroll
            for(int k = 0; k < PK; k++)
                rv[k] = pval[min(mt.pbeg + lane + 64 * k, last)];
#pragma unroll
            for(int k = 0; k < EK; k++)
                re[k] = cind[mt.ebeg + lane + 64 * k]; // (cind is padded by 256 words)
        }
    };
    if(s0 + wave < s1)
        level2(cur);
    for(int s = s0 + wave; s < s1; s += NW)
    {
        if(__builtin_amdgcn_ballot_w64(dead) != 0)
            break;
        staged_cur = staged;
        const bool live = cur.live, base = a == 0 && cur.c > 0;
        const int  c = cur.c, n0 = cur.p1 - cur.p0, nl = n0 < EXT ? n0 : EXT;
        const unsigned long long t_begin = trace ? __builtin_amdgcn_s_memrealtime() : 0;
        const unsigned long long c_begin = (trace && (dbg & 8)) ? __builtin_amdgcn_s_memtime() : 0;
        int la[EXT];
        T   ve[EXT], vn[BS][BS - 1];
        if(!staged_cur) // (a step with a very long single row: lane by lane, when the step starts -- nothing of it is carried over)
        {
#pragma unroll
            for(int e = 0; e < EXT; e++)
            {
                la[e] = cind[cur.e0 + e];
                ve[e] = pval[min(cur.pa + (FRONT ? a : 0) + e, last)];
            }
            if constexpr(FRONT)
            {
#pragma unroll
                for(int tt = 0; tt < BS - 1; tt++)
                    vn[1][tt] = pval[min(max(cur.pa + a - 1 - tt, 0), last)]; // (vn[1][.]: this lane's own row on block row tt)
            }
            else
            {
#pragma unroll
                for(int r = 1; r < BS; r++)
#pragma unroll
                    for(int tt = 0; tt < r; tt++)
                        vn[r][tt] = pval[min(cur.p0 + r * n0 + (r * (r - 1)) / 2 + n0 + tt, last)];
            }
        }
        else // deal the step's data out: staging area -> this lane's row, this block's list, the first lane's block rows
        {
#pragma unroll
            for(int k = 0; k < PK; k++)
                if(64 * k < PCAP)
                    scr_p[min(lane + 64 * k, PCAP - 1)] = rv[k];
#pragma unroll
            for(int k = 0; k < EK; k++)
                scr_e[min(lane + 64 * k, ECAP - 1)] = re[k];
            // (offsets of lanes that own nothing are 0; a live lane's reads stay inside the area: it is padded by EXT + 16 entries,
            // what a read past the end of the step's last row can overshoot -- no clamps, constant offsets in the instructions)
            const int po = live ? cur.pa - cur.pbeg : 0, eo = c > 0 ? cur.e0 - cur.ebeg : 0, bo = base ? cur.p0 - cur.pbeg : 0;
#pragma unroll
            for(int e = 0; e < EXT; e++)
            {
                la[e] = scr_e[eo + e];
                ve[e] = scr_p[po + (FRONT && live ? a : 0) + e];
            }
            if constexpr(FRONT)
            {
#pragma unroll
                for(int tt = 0; tt < BS - 1; tt++)
                    vn[1][tt] = scr_p[max(po + (live ? a : 0) - 1 - tt, 0)];
            }
            else
            {
#pragma unroll
                for(int r = 1; r < BS; r++)
#pragma unroll
                    for(int tt = 0; tt < r; tt++)
                        vn[r][tt] = scr_p[bo + min(r, c > 0 ? c - 1 : 0) * n0 + (r * (r - 1)) / 2 + n0 + tt];
            }
        }
        // (masked NOW, before the wait: left to the compiler, the selects moved behind it)
#pragma unroll
        for(int e = 0; e < EXT; e++)
        {
            la[e] = (c > 0 && e < nl) ? la[e] : zslot;
            ve[e] = (live && e < nl) ? ve[e] : T(0);
            asm volatile("" : "+v"(ve[e]), "+v"(la[e]));
        }
        T rhs = live ? alpha * bb : T(0);
        T dg  = (live && !UNIT) ? dd : T(1);
        asm volatile("" : "+v"(rhs), "+v"(dg));
        if constexpr(FRONT)
        {
#pragma unroll
            for(int tt = 0; tt < BS - 1; tt++)
            {
                vn[1][tt] = (live && tt < a) ? vn[1][tt] : T(0);
                asm volatile("" : "+v"(vn[1][tt]));
            }
        }
        else
        {
#pragma unroll
            for(int r = 1; r < BS; r++)
#pragma unroll
                for(int tt = 0; tt < r; tt++)
                {
                    vn[r][tt] = (base && r < c) ? vn[r][tt] : T(0);
                    asm volatile("" : "+v"(vn[r][tt]));
                }
        }
        // the next step of this wavefront: level-1 metadata now (its header came during the previous step), header of the one after
        if(s + NW < s1)
        {
            level1(g0, g1, nxt);
            if(s + 2 * NW < s1)
                header(s + 2 * NW, g0, g1);
        }
        // One round of LDS reads for all dependencies; then, while some lane misses something (`pend`, uniform), the wavefront
        // spins on ONE missing word and looks at the other missing ones again when that one has arrived.  Six wavefronts that
        // re-read all their 15-24 words every ~0.1 us saturate the CU's LDS pipe (traced: a hand-off through LDS took 2.5-3.7 us,
        // the writes of the one wavefront that works queued behind the polls); one word per look is ~100 reads per us.
        B bits[EXT];
#pragma unroll
        for(int e = 0; e < EXT; e++)
            bits[e] = lds_get(la[e]);
        unsigned long long t_first = 0;
        if(trace)
        {
            asm volatile("" : "+v"(bits[EXT - 1]));
            t_first = __builtin_amdgcn_s_memrealtime(); // the first look has landed: values and dependency list were in
        }
        unsigned int pend = 0;
#pragma unroll
        for(int e = 0; e < EXT; e++)
            if(__builtin_amdgcn_ballot_w64(bits[e] == tag<T>::value) != 0)
                pend |= 1u << e;
        while(pend != 0 && __builtin_amdgcn_ballot_w64(dead) == 0)
        {
            int ad = zslot; // the LAST missing entry: the block solved last is the nearest in the chain order of L
#pragma unroll
            for(int e = 0; e < EXT; e++)
                if(pend & (1u << e))
                    ad = la[e];
            for(;;)
            {
                const B v = lds_get(ad);
                if(__builtin_amdgcn_ballot_w64(v == tag<T>::value) == 0 || __builtin_amdgcn_ballot_w64(dead) != 0)
                    break;
                __builtin_amdgcn_s_sleep(2); // (no sleep / s_sleep 8: the same hand-off time, profiles/r6/trsv_chunk_experiments.txt)
                tick(4095u);
            }
            if(trace && (dbg & 4))
                t_first = __builtin_amdgcn_s_memrealtime(); // (diagnostics, dbg 4: when the spin word was seen)
            // then ALL entries again, without a branch (guarded re-reads of the missing ones only became one basic block each
            // with a wait at every join: 15 LDS round trips one after the other, ~1 us on the critical path)
#pragma unroll
            for(int e = 0; e < EXT; e++)
                bits[e] = lds_get(la[e]);
            pend = 0;
#pragma unroll
            for(int e = 0; e < EXT; e++)
                pend |= __builtin_amdgcn_ballot_w64(bits[e] == tag<T>::value) != 0 ? (1u << e) : 0u;
        }
        const unsigned long long t_ready = trace ? __builtin_amdgcn_s_memrealtime() : 0;
        // ---- from here on every instruction is on the critical path of the solve ----
        T xe[EXT];
#pragma unroll
        for(int e = 0; e < EXT; e++)
            __builtin_memcpy(&xe[e], &bits[e], sizeof(T));
        unsigned long long t_ext = 0, t_elim = 0;
        if constexpr(FRONT)
        {
            // phase r: row r of every block -- [its block rows r-1 .. 0, then the external entries], the order ref_trsv_u applies them
            T          xs[BS];
            const bool okl = live && !__builtin_amdgcn_ballot_w64(dead);
            auto       phase = [&](auto rtag) {
                constexpr int r   = decltype(rtag)::value;
                T             acc = rhs;
#pragma unroll
                for(int tt = BS - 2; tt >= 0; tt--)
                    if(tt < r)
                        acc = neg_fma(vn[1][tt], xs[tt], acc);
#pragma unroll
                for(int e = 0; e < EXT; e++)
                    acc = neg_fma(ve[e], xe[e], acc);
                if constexpr(r == 0)
                {
                    // a single row with more than EXT dependencies (never inside a multi-row block): the rest one by one
                    if(__builtin_amdgcn_ballot_w64(base && n0 > EXT) != 0)
                    {
                        if(base && n0 > EXT)
                            for(int p = EXT; p < n0 && !dead; p++)
                            {
                                const int q = cind[cur.e0 + p];
                                B         got;
                                for(;;)
                                {
                                    got = lds_get(q);
                                    if(got != tag<T>::value || dead)
                                        break;
                                    __builtin_amdgcn_s_sleep(1);
                                    tick(1023u);
                                }
                                T xv;
                                __builtin_memcpy(&xv, &got, sizeof(T));
                                acc = neg_fma(pval[cur.pa + p], xv, acc);
                            }
                    }
                }
                if constexpr(!UNIT)
                    acc /= dg;
                B out;
                __builtin_memcpy(&out, &acc, sizeof(T));
                out = out == tag<T>::value ? qnan_bits<T>::value : out;
                lds_put((okl && a == r) ? cur.myslot : park, out);
                xs[r] = trsv_bcast8<r>(acc);
            };
            phase(std::integral_constant<int, 0>{});
            if(trace)
            {
                asm volatile("" : "+v"(xs[0]));
                t_ext = __builtin_amdgcn_s_memrealtime();
            }
            phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
            phase(std::integral_constant<int, 4>{});
            if constexpr(BS > 5)
            {
                phase(std::integral_constant<int, 5>{});
                phase(std::integral_constant<int, 6>{});
                phase(std::integral_constant<int, 7>{});
            }
            if(trace)
            {
                asm volatile("" : "+v"(xs[BS - 1]));
                t_elim = __builtin_amdgcn_s_memrealtime();
            }
        }
        else
        {
        T sa = rhs;
#pragma unroll
        for(int e = 0; e < EXT; e++)
            sa = neg_fma(ve[e], xe[e], sa);
        // a single row with more than EXT dependencies (never inside a multi-row block): the rest one by one
        if(__builtin_amdgcn_ballot_w64(base && n0 > EXT) != 0)
        {
            if(base && n0 > EXT)
                for(int p = EXT; p < n0 && !dead; p++)
                {
                    const int q = cind[cur.e0 + p];
                    B         got;
                    for(;;)
                    {
                        got = lds_get(q);
                        if(got != tag<T>::value || dead)
                            break;
                        __builtin_amdgcn_s_sleep(1);
                        tick(1023u);
                    }
                    T xv;
                    __builtin_memcpy(&xv, &got, sizeof(T));
                    sa = neg_fma(pval[cur.pa + p], xv, sa);
                }
        }
        if(trace)
        {
            asm volatile("" : "+v"(sa));
            t_ext = __builtin_amdgcn_s_memrealtime();
        }
        // the sums and the diagonals of the block's rows -> its first lane
        T sg[8], dgs[8];
        sg[0] = sa, dgs[0] = dg;
        sg[1] = trsv_dpp_shl<1>(sa), sg[2] = trsv_dpp_shl<2>(sa), sg[3] = trsv_dpp_shl<3>(sa), sg[4] = trsv_dpp_shl<4>(sa);
        if constexpr(BS > 5)
            sg[5] = trsv_dpp_shl<5>(sa), sg[6] = trsv_dpp_shl<6>(sa), sg[7] = trsv_dpp_shl<7>(sa);
        if constexpr(!UNIT)
        {
            dgs[1] = trsv_dpp_shl<1>(dg), dgs[2] = trsv_dpp_shl<2>(dg), dgs[3] = trsv_dpp_shl<3>(dg), dgs[4] = trsv_dpp_shl<4>(dg);
            if constexpr(BS > 5)
                dgs[5] = trsv_dpp_shl<5>(dg), dgs[6] = trsv_dpp_shl<6>(dg), dgs[7] = trsv_dpp_shl<7>(dg);
        }
        // column by column: as soon as block row tt is final it goes to LDS (what the chunk's other wavefronts poll) and is taken
        // out of every later row; lanes and rows that own nothing store to their parked slot
        const int  slot0 = cur.myslot; // (first lane: slot of block row 0)
        const bool ok    = base && !__builtin_amdgcn_ballot_w64(dead);
#pragma unroll
        for(int tt = 0; tt < BS; tt++)
        {
            T xt = sg[tt];
            if constexpr(!UNIT) // (a template parameter: as a run-time test the compiler computed the quotient anyway and selected)
                xt /= dgs[tt];
            B out;
            __builtin_memcpy(&out, &xt, sizeof(T));
            out = out == tag<T>::value ? qnan_bits<T>::value : out;
            lds_put((ok && tt < c) ? slot0 + tt : park, out);
#pragma unroll
            for(int r = tt + 1; r < BS; r++)
                sg[r] = neg_fma(vn[r][tt], xt, sg[r]);
        }
        if(trace)
        {
            asm volatile("" : "+v"(sg[BS - 1]));
            t_elim = __builtin_amdgcn_s_memrealtime();
        }
        }
        // every row lane takes its own x back out of LDS: the tagged word for the other chunks, and the caller's x -- stored BEHIND
        // the requests for the next step's data
        const bool alive  = __builtin_amdgcn_ballot_w64(dead) == 0;
        B          mine   = lds_get(cur.myslot);
        const int  st_pos = cur.mypos, st_row = cur.row;
        asm volatile("" : "+v"(mine));
        if(s + NW < s1)
        {
            cur = nxt;
            level2(cur);
        }
        if(alive && live)
        {
            __hip_atomic_store(&xb[st_pos], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            T xv;
            __builtin_memcpy(&xv, &mine, sizeof(T));
            x[(size_t)st_row * incx] = xv;
        }
        if(trace && lane == 0) // (diagnostics: AOCLSPARSE_MI355_TRSV_TRACE with schedule 5 -- 8 words per step)
        {
            unsigned long long *tr = trace + 8 * (size_t)s;
            tr[0] = t_begin, tr[1] = t_first, tr[2] = t_ready, tr[3] = __builtin_amdgcn_s_memrealtime();
            tr[4] = t_ext, tr[5] = t_elim, tr[6] = (unsigned long long)ch, tr[7] = (unsigned long long)(s - s0);
            if(dbg & 8) // (diagnostics: the shader clock next to the 100 MHz one -> the clock the CU really runs at)
                tr[4] = __builtin_amdgcn_s_memtime(), tr[5] = c_begin;
        }
    }
    if(__builtin_amdgcn_ballot_w64(dead) != 0 && lane == 0)
        __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// scratch: nrhs ticket words followed by one timeout word (zeroed here for the sync-free schedule)
template <typename T>
aoclsparse_status launch_trsv(hipStream_t s, int schedule, bool unit, T alpha, aoclsparse_int m,
                              const TrsvPlan &plan, const T *diag, const T *b, T *x, T *xp, unsigned int *scratch,
                              aoclsparse_int nrhs, long long b_off, aoclsparse_int incb, long long x_off,
                              aoclsparse_int incx, unsigned int *timeout_word, int kt_bits)
{
    if(m <= 0 || nrhs <= 0)
        return aoclsparse_status_success;
    // kt_bits: 0 = the reference chain (ref_trsv_*); 256 / 512 = the KT kernels' order for that vector width (kid 1/2 / 3).
    // Served by trsv_block_kt_kernel when the triangle has a block plan (schedule 4), else by the per-level launches and the
    // lane-per-position sync-free kernel.
    constexpr int T256 = std::is_same<T, double>::value ? 4 : 8;
    if(kt_bits != 0 && kt_bits != 256 && kt_bits != 512)
        return aoclsparse_status_internal_error;
    // (the block plan serves the KT orders through trsv_block_kt_kernel; without it: per-level launches or lane per position)
    if(kt_bits != 0 && schedule == 5)
        schedule = 4; // (the two-level kernel serves the reference chain only)
    if(kt_bits != 0 && schedule != 0 && !(schedule == 4 && plan.blk.valid))
        schedule = 2;
    // the level-ordered row layout, or -- when only the block plan was built (TrsvPlan::rows_valid == false) -- the block
    // plan's layout: also a topological order of the rows, which is all the lane-per-position kernel (schedule 2) needs
    const bool            rows   = plan.rows_valid;
    const aoclsparse_int *rowmap = rows ? plan.rowmap.as<aoclsparse_int>() : plan.blk.rowmap.as<aoclsparse_int>();
    const aoclsparse_int *pptr   = rows ? plan.pptr.as<aoclsparse_int>() : plan.blk.pptr.as<aoclsparse_int>();
    const aoclsparse_int *pind   = rows ? plan.pind.as<aoclsparse_int>() : plan.blk.pind.as<aoclsparse_int>();
    const T              *pval   = rows ? plan.pval.as<T>() : plan.blk.pval.as<T>();
    if(!rows && !plan.blk.valid)
        return aoclsparse_status_internal_error;
    RhsGeom               g{b_off, x_off, incb, incx, 0};
    if(schedule == 1 && (nrhs != 1 || incb != 1 || incx != 1))
        schedule = 2; // the single-workgroup runs of the hybrid schedule are single-RHS, unit stride
    // (several right-hand sides: the column is grid dimension x, the slice y <= 65,535; index arithmetic in int)
    if(schedule == 5
       && (!plan.blk.valid || !plan.blk.chunk.valid || kt_bits != 0
           || (nrhs > 1 && (plan.blk.chunk.nchunks > 65535 || (long long)m * nrhs + TRSV_XP_PAD >= (1LL << 31)))))
        schedule = 4;
    if(schedule == 4
       && (!plan.blk.valid || (nrhs > 1 && (plan.blk.nslices > 65535 || (long long)m * nrhs + TRSV_XP_PAD >= (1LL << 31)))))
        schedule = 3;
    if(schedule == 3 && (nrhs != 1 || plan.nslices <= 0 || !rows))
        schedule = 2; // the slice kernel is single-RHS; trsm keeps the lane-per-position kernel
    if((schedule == 0 || schedule == 1) && !rows)
        return aoclsparse_status_internal_error; // the caller asks for the row layout before choosing these (trsv_api.cpp)
    auto level_launch = [&](aoclsparse_int l) {
        const aoclsparse_int first = plan.level_ptr[l], count = plan.level_ptr[l + 1] - first;
        const int            bs = count >= 256 ? 256 : 64;
        for(aoclsparse_int c0 = 0; c0 < nrhs; c0 += 65535)
        {
            const int nc = nrhs - c0 < 65535 ? nrhs - c0 : 65535;
#define MI355_LEVEL_ARGS first, count, m, rowmap, pptr, pind, pval, diag, b + c0 * b_off, xp + (size_t)c0 * m, x + c0 * x_off, alpha, (int)unit, g
            if(kt_bits == 0)
                hipLaunchKernelGGL((trsv_level_kernel<T, 0>), dim3((count + bs - 1) / bs, nc), dim3(bs), 0, s, MI355_LEVEL_ARGS);
            else if(kt_bits == 256)
                hipLaunchKernelGGL((trsv_level_kernel<T, T256>), dim3((count + bs - 1) / bs, nc), dim3(bs), 0, s, MI355_LEVEL_ARGS);
            else
                hipLaunchKernelGGL((trsv_level_kernel<T, 2 * T256>), dim3((count + bs - 1) / bs, nc), dim3(bs), 0, s, MI355_LEVEL_ARGS);
#undef MI355_LEVEL_ARGS
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
    else if(schedule == 4)
    {
        // sync-free, one lane per block of chained rows (plan.blk has its own level-ordered copy of the triangle)
        const TrsvBlockPlan &bp = plan.blk;
        // scratch: one ticket per column, the timeout word, then one finished-slices counter per (column, block level)
        const long long total = (long long)m * nrhs;
        MI355_HIP_TRY(hipMemsetAsync(scratch, 0, ((size_t)nrhs + 1 + (size_t)nrhs * bp.nlevels) * sizeof(unsigned int), s));
        hipLaunchKernelGGL((trsv_fill_tag_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xp, total,
                           total + TRSV_XP_PAD - 1);
        constexpr int gate = 2; // a wavefront starts looking at its dependencies when the block level two below is complete
        // diagnostic: AOCLSPARSE_MI355_TRSV_TRACE=<file> dumps, per slice, the 100 MHz clock after the ticket, when the
        // dependencies were all in, at the end, the slice's block level, after the LDS reads, after the external FMAs (6 x u64 per slice; tools/trsv_trace.py)
        static const char  *trace_path = getenv("AOCLSPARSE_MI355_TRSV_TRACE");
        unsigned long long *trace      = nullptr;
        if(trace_path && std::is_same<T, double>::value
           && hipMalloc(&trace, sizeof(unsigned long long) * 6 * (size_t)bp.nslices) != hipSuccess)
            trace = nullptr;
        aoclsparse_status lst = aoclsparse_status_success;
        if(kt_bits != 0)
        {
            if(bp.max_rows > TRSV_BLK_ROWS)
                return aoclsparse_status_internal_error; // (the plan never builds larger blocks)
            const dim3 grid = nrhs > 1 ? dim3((unsigned)nrhs, (unsigned)bp.nslices) : dim3((unsigned)bp.nslices);
#define MI355_BLKKT_ARGS                                                                                                      \
    m, bp.nslices, bp.slices.as<aoclsparse_int>(), bp.bfirst.as<aoclsparse_int>(), bp.rowmap.as<aoclsparse_int>(),              \
        bp.pptr.as<aoclsparse_int>(), bp.pind.as<aoclsparse_int>(), bp.pval.as<T>(), diag, b, xp, x, alpha, (int)unit, scratch,   \
        timeout_word ? timeout_word : scratch + nrhs, (int)incb, (int)incx, scratch + nrhs + 1, gate, (int)nrhs, b_off, x_off,     \
        (int)bp.nlevels, trace
            constexpr size_t kt_lds = sizeof(T) * 64 * (size_t)(TRSV_BLK_NV + TRSV_BLK_EXT + TRSV_BLK_ROWS);
            auto go_kt = [&](auto tsz_tag, auto front_tag) {
                constexpr int  TSZ = decltype(tsz_tag)::value;
                constexpr bool FR  = decltype(front_tag)::value;
                static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_block_kt_kernel<T, TSZ, FR>),
                                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kt_lds);
                if(raised != hipSuccess)
                    return aoclsparse_status_internal_error;
                hipLaunchKernelGGL((trsv_block_kt_kernel<T, TSZ, FR>), grid, dim3(64), kt_lds, s, MI355_BLKKT_ARGS);
                return aoclsparse_status_success;
            };
            if(kt_bits == 256)
                lst = bp.front ? go_kt(std::integral_constant<int, T256>{}, std::true_type{})
                               : go_kt(std::integral_constant<int, T256>{}, std::false_type{});
            else
                lst = bp.front ? go_kt(std::integral_constant<int, 2 * T256>{}, std::true_type{})
                               : go_kt(std::integral_constant<int, 2 * T256>{}, std::false_type{});
            if(lst != aoclsparse_status_success)
                return lst;
#undef MI355_BLKKT_ARGS
            MI355_HIP_TRY(hipGetLastError());
        }
        else
        {
        auto go_form = [&](auto bs_tag, auto ext_tag, auto front_tag) {
            constexpr int    BS = decltype(bs_tag)::value, EXT = decltype(ext_tag)::value;
            constexpr bool   FRONT = decltype(front_tag)::value;
            // (the small shapes hold the block in registers and use no LDS)
            constexpr size_t need = trsv_blk_slots(BS, EXT) <= TRSV_BLK_REG_SLOTS ? 0 : sizeof(T) * 64 * (size_t)trsv_blk_slots(BS, EXT);
            static_assert(need <= 160 * 1024, "LDS of one CU");
            const size_t     lds  = std::min<size_t>(need, 160 * 1024);
            if(lds > 64 * 1024)
            {
                static const hipError_t raised = hipFuncSetAttribute(
                    reinterpret_cast<const void *>(&trsv_block_kernel<T, BS, EXT, FRONT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if(raised != hipSuccess)
                    return aoclsparse_status_internal_error;
            }
            const dim3 grid = nrhs > 1 ? dim3((unsigned)nrhs, (unsigned)bp.nslices) : dim3((unsigned)bp.nslices);
#define MI355_BLK_ARGS                                                                                                       \
    m, bp.nslices, bp.slices.as<aoclsparse_int>(), bp.bfirst.as<aoclsparse_int>(), bp.rowmap.as<aoclsparse_int>(),              \
        bp.pptr.as<aoclsparse_int>(), bp.pind.as<aoclsparse_int>(), bp.pval.as<T>(), diag, b, xp, x, alpha, (int)unit, scratch,   \
        timeout_word ? timeout_word : scratch + nrhs, (int)incb, (int)incx, trace, scratch + nrhs + 1, gate, (int)nrhs, b_off,     \
        x_off, (int)bp.nlevels
            if constexpr(std::is_same<T, double>::value) // the traced build exists for double only
            {
                if(trace)
                {
                    if(lds > 64 * 1024)
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_block_kernel<T, BS, EXT, FRONT, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    hipLaunchKernelGGL((trsv_block_kernel<T, BS, EXT, FRONT, true>), grid, dim3(64), lds, s, MI355_BLK_ARGS);
                    return aoclsparse_status_success;
                }
            }
            hipLaunchKernelGGL((trsv_block_kernel<T, BS, EXT, FRONT>), grid, dim3(64), lds, s, MI355_BLK_ARGS);
#undef MI355_BLK_ARGS
            return aoclsparse_status_success;
        };
        auto go = [&](auto bs_tag, auto ext_tag) {
            return bp.front ? go_form(bs_tag, ext_tag, std::true_type{}) : go_form(bs_tag, ext_tag, std::false_type{});
        };
        using std::integral_constant;
        // shapes by the plan's largest block / external list (the loops over rows and external entries are unrolled)
        const bool small_ext = bp.max_ext <= 16, small_bs = bp.max_rows <= 5;
        lst = small_ext && small_bs ? go(integral_constant<int, 5>{}, integral_constant<int, 16>{})
              : bp.max_ext <= 20 && small_bs ? go(integral_constant<int, 5>{}, integral_constant<int, 20>{})
              : small_ext           ? go(integral_constant<int, TRSV_BLK_ROWS>{}, integral_constant<int, 16>{})
              : small_bs            ? go(integral_constant<int, 5>{}, integral_constant<int, TRSV_BLK_EXT>{})
                                    : go(integral_constant<int, TRSV_BLK_ROWS>{}, integral_constant<int, TRSV_BLK_EXT>{});
        }
        if(trace)
        {
            std::vector<unsigned long long> host(6 * (size_t)bp.nslices);
            if(hipStreamSynchronize(s) == hipSuccess
               && hipMemcpy(host.data(), trace, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) == hipSuccess)
                if(FILE *f = fopen(trace_path, "wb"))
                {
                    fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
                    fclose(f);
                }
            fprintf(stderr, "[trsv trace] blocks %d slices %d levels %d max_rows %d max_ext %d\n", (int)bp.nblocks, (int)bp.nslices,
                    (int)bp.nlevels, bp.max_rows, bp.max_ext);
            (void)hipFree(trace);
        }
        if(lst != aoclsparse_status_success)
            return lst;
    }
    else if(schedule == 5)
    {
        // two levels: a workgroup per chunk of consecutive blocks, hand-offs inside a chunk through LDS (trsv_chunk_kernel)
        const TrsvBlockPlan &bp = plan.blk;
        const TrsvChunkPlan &cp = bp.chunk;
        const long long      total = (long long)m * nrhs;
        MI355_HIP_TRY(hipMemsetAsync(scratch, 0, ((size_t)nrhs + 1) * sizeof(unsigned int), s));
        hipLaunchKernelGGL((trsv_fill_tag_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xp, total);
        const bool   small_ext5 = bp.max_ext <= 16, small_bs5 = bp.max_rows <= 5;
        const size_t stage_bytes = (size_t)(TRSV_CHUNK_WAVES - 1)
                               * ((size_t)trsv_chunk_pcap(small_bs5 ? 5 : TRSV_CHUNK_LANES, small_ext5 ? 16 : TRSV_BLK_EXT) * sizeof(T)
                                  + 9 * (size_t)(small_ext5 ? 16 : TRSV_BLK_EXT) * 4);
        const size_t lds = 16 + stage_bytes + sizeof(typename tag<T>::bits) * ((size_t)cp.max_rows + 1 + 64);
        if(lds > 160 * 1024)
            return aoclsparse_status_internal_error; // (the plan caps a chunk's rows)
        // diagnostic: AOCLSPARSE_MI355_TRSV_TRACE=<file> dumps, per step, the 100 MHz clock when its wavefront took it, when its first
        // look at the dependencies had landed, when all of them were in, at the end; then per step its chunk and block level
        // (tools/trsv_chunk_trace.py)
        static const int    dbg5        = getenv("AOCLSPARSE_MI355_TRSV_DBG") ? atoi(getenv("AOCLSPARSE_MI355_TRSV_DBG")) : 0;
        static const char  *trace_path5 = getenv("AOCLSPARSE_MI355_TRSV_TRACE");
        unsigned long long *trace5      = nullptr;
        if(trace_path5 && nrhs == 1 && hipMalloc(&trace5, sizeof(unsigned long long) * 8 * (size_t)cp.nsteps) != hipSuccess)
            trace5 = nullptr;
        auto go_form = [&](auto ext_tag, auto bs_tag, auto unit_tag, auto front_tag) {
            constexpr int  EXT = decltype(ext_tag)::value, BS = decltype(bs_tag)::value;
            constexpr bool UNIT = decltype(unit_tag)::value, FRONT = decltype(front_tag)::value;
            static const hipError_t raised = hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_chunk_kernel<T, EXT, BS, UNIT, FRONT>),
                                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if(raised != hipSuccess)
                return aoclsparse_status_internal_error;
            const dim3 grid = nrhs > 1 ? dim3((unsigned)nrhs, (unsigned)cp.nchunks) : dim3((unsigned)cp.nchunks);
            hipLaunchKernelGGL((trsv_chunk_kernel<T, EXT, BS, UNIT, FRONT>), grid, dim3(64 * TRSV_CHUNK_WAVES), lds, s, m,
                               std::max<aoclsparse_int>(plan.nnz_tri, 1), cp.nchunks,
                               reinterpret_cast<const int4 *>(cp.steps.as<aoclsparse_int>()), cp.cptr.as<aoclsparse_int>(),
                               bp.rowmap.as<aoclsparse_int>(), bp.pptr.as<aoclsparse_int>(), bp.pval.as<T>(),
                               cp.eptr.as<aoclsparse_int>(), cp.cind.as<aoclsparse_int>(), cp.hind.as<aoclsparse_int>(), diag, b, xp, x,
                               alpha, scratch, timeout_word ? timeout_word : scratch + nrhs, (int)incb, (int)incx, (int)nrhs, b_off,
                               x_off, trace5, dbg5);
            return aoclsparse_status_success;
        };
        auto go = [&](auto ext_tag, auto bs_tag) {
            if(bp.front)
                return unit ? go_form(ext_tag, bs_tag, std::true_type{}, std::true_type{})
                            : go_form(ext_tag, bs_tag, std::false_type{}, std::true_type{});
            return unit ? go_form(ext_tag, bs_tag, std::true_type{}, std::false_type{})
                        : go_form(ext_tag, bs_tag, std::false_type{}, std::false_type{});
        };
        using std::integral_constant;
        const bool              small_ext = bp.max_ext <= 16, small_bs = bp.max_rows <= 5;
        const aoclsparse_status lst
            = small_ext && small_bs ? go(integral_constant<int, 16>{}, integral_constant<int, 5>{})
              : small_ext           ? go(integral_constant<int, 16>{}, integral_constant<int, TRSV_CHUNK_LANES>{})
              : small_bs            ? go(integral_constant<int, TRSV_BLK_EXT>{}, integral_constant<int, 5>{})
                                    : go(integral_constant<int, TRSV_BLK_EXT>{}, integral_constant<int, TRSV_CHUNK_LANES>{});
        if(trace5)
        {
            std::vector<unsigned long long> host(8 * (size_t)cp.nsteps);
            if(hipStreamSynchronize(s) == hipSuccess
               && hipMemcpy(host.data(), trace5, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) == hipSuccess)
                if(FILE *f = fopen(trace_path5, "wb"))
                {
                    fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
                    fclose(f);
                }
            fprintf(stderr, "[trsv chunk trace] chunks %d steps %d lds slots %d model %.1f us (block schedule %.1f us)\n", (int)cp.nchunks,
                    (int)cp.nsteps, (int)cp.max_rows, cp.model_us, cp.model_block_us);
            (void)hipFree(trace5);
        }
        if(lst != aoclsparse_status_success)
            return lst;
    }
    else if(schedule == 3)
    {
        // sync-free, one level slice per wavefront (single right-hand side)
        MI355_HIP_TRY(hipMemsetAsync(scratch, 0, 2 * sizeof(unsigned int), s));
        hipLaunchKernelGGL((trsv_fill_tag_kernel<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, xp, (long long)m);
        const bool  wide = (long long)plan.nnz_tri > 10LL * m;
        constexpr int wv = 16; // slices per workgroup (4 / 8 measured slower: 2.02 / 1.90 vs 1.69 ms, round 2)
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
#define MI355_SF_ARGS m, rowmap, pptr, pind, pval, diag, b + c0 * b_off, xp + (size_t)c0 * m, x + c0 * x_off, alpha, (int)unit, scratch + c0, tmo, g
            if(kt_bits == 256)
            {
                if(wide)
                    hipLaunchKernelGGL((trsv_syncfree_kernel<T, 512, 20, T256>), grid, dim3(512), 0, s, MI355_SF_ARGS);
                else
                    hipLaunchKernelGGL((trsv_syncfree_kernel<T, 1024, 12, T256>), grid, dim3(1024), 0, s, MI355_SF_ARGS);
            }
            else if(kt_bits == 512)
            {
                if(wide)
                    hipLaunchKernelGGL((trsv_syncfree_kernel<T, 512, 20, 2 * T256>), grid, dim3(512), 0, s, MI355_SF_ARGS);
                else
                    hipLaunchKernelGGL((trsv_syncfree_kernel<T, 1024, 12, 2 * T256>), grid, dim3(1024), 0, s, MI355_SF_ARGS);
            }
            else if(wide)
                hipLaunchKernelGGL((trsv_syncfree_kernel<T, 512, 20>), grid, dim3(512), 0, s, MI355_SF_ARGS);
            else
                hipLaunchKernelGGL((trsv_syncfree_kernel<T, 1024, 12>), grid, dim3(1024), 0, s, MI355_SF_ARGS);
#undef MI355_SF_ARGS
        }
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_trsv<double>(hipStream_t, int, bool, double, aoclsparse_int, const TrsvPlan &,
                                               const double *, const double *, double *, double *, unsigned int *,
                                               aoclsparse_int, long long, aoclsparse_int, long long, aoclsparse_int,
                                               unsigned int *, int);
template aoclsparse_status launch_trsv<float>(hipStream_t, int, bool, float, aoclsparse_int, const TrsvPlan &,
                                              const float *, const float *, float *, float *, unsigned int *,
                                              aoclsparse_int, long long, aoclsparse_int, long long, aoclsparse_int,
                                              unsigned int *, int);

} // namespace mi355
