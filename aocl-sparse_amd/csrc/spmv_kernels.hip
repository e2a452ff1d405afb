// spmv_kernels.hip -- CSR-Adaptive SpMV for gfx950 (wave64, 256 CUs / 8 XCDs, 160 KiB LDS per CU).
//
// One workgroup (256 threads = 4 wavefronts) owns one ROW BLOCK: consecutive whole rows whose
// non-zeros fit one LDS tile (TILE = 1024 or 2048, chosen per matrix by the planner).  The plan
// entry of a block is {first row, first non-zero}, so a workgroup knows its row range AND its
// non-zero range after one small cached load.
//   phase 1: the block's val[] / col_ind[] are streamed with coalesced 16-byte / 8-byte loads (two
//            non-zeros per lane), x[col] is gathered (L2 / Infinity-Cache hits) and {val, x} are
//            parked in LDS; the block's row_ptr slice is loaded in the same breath into LDS.
//   phase 2: every row is reduced out of LDS in EXACTLY the summation order of the CPU kernel the
//            reference would dispatch:
//     order 0: one lane per row, left-to-right FMA chain      = ref_csrmv_gn         (csrmv_kr.hpp:448-513)
//     order 1: 4 lanes per row, j mod 4, (l0+l1)+(l2+l3), tail = ..._vectorized_avx2  (csrmv_kr.hpp:949-1040)
//     order 2: 8 lanes per row, j mod 8, AVX-512 tree, tail    = ..._vectorized_avx512 (csrmv_avx512.cpp:36-134)
//              (float: the AVX2 8-lane tree of csrmv_kr.hpp:734-831)
// so y is bit-identical to that CPU kernel (GPU fma == x86 vfmadd).  Two exceptions, both only without a pinned kid
// (auto mode, VAR 0) and both with a stated componentwise bound (tests): a scalar-order row of >= SPMV_TREE_MIN (32)
// entries inside a tile is summed by its wavefront (64 strided chains + wave_sum; round 5), and a row longer than a tile gets a
// workgroup of its own with a workgroup-wide tree.  `strict` (a pinned kid, or aoclsparse_mi355_set_option(spmv_strict, 1); VAR 1)
// keeps the reference order for every row: tiles through LDS, owner lanes chain.
//
// HBM traffic per launch = the algorithmic bytes: val 8 B + col 4 B per nnz, row_ptr 4 B + y 8 B per
// row, x 8 B per column once (re-reads are cache hits), + 8 B per row block of plan.
// Roofline: HBM (AI ~ 0.16 flop/B).
#include "internal.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include <algorithm>

namespace mi355
{

__device__ __forceinline__ double dev_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float dev_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}

// alpha/beta epilogue of every reference csrmv kernel (csrmv_kr.hpp:497-509)
template <typename T>
__device__ __forceinline__ T finish(T r, T alpha, T beta, const T *yi)
{
    if(alpha != T(1))
        r = alpha * r;
    if(beta != T(0))
        r = dev_fma(beta, *yi, r);
    return r;
}

// y is written once and not re-read by this launch: a non-temporal store (flags bit3, set for vectors
// larger than the aggregate L2) keeps it out of the L2 the x gathers live in (measured +2 %)
template <typename T>
__device__ __forceinline__ void store_y(T *p, T v, int flags)
{
    if(flags & 8)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// lane-group reduction in the reference's horizontal-add order; valid in lane 0 of the group
template <typename T, int ORDER>
__device__ __forceinline__ T group_reduce(T acc)
{
    if constexpr(ORDER == 1)
    {
        T t = acc + __shfl_down(acc, 1, 4); // l0+l1 | l2+l3
        return t + __shfl_down(t, 2, 4);
    }
    else if constexpr(sizeof(T) == 8)
    {
        T v = acc + __shfl_down(acc, 4, 8); // lo4 + hi4
        T t = v + __shfl_down(v, 1, 8); // v0+v1 | v2+v3
        return t + __shfl_down(t, 2, 8);
    }
    else
    {
        T q = acc + __shfl_down(acc, 4, 8); // x0+x4 .. x3+x7
        T d = q + __shfl_down(q, 2, 8); // q0+q2 | q1+q3
        return d + __shfl_down(d, 1, 8);
    }
}

template <int ORDER>
struct lanes_of
{
    static constexpr int value = ORDER == 0 ? 1 : (ORDER == 1 ? 4 : 8);
};

template <typename T>
struct pair_of;
template <>
struct pair_of<double>
{
    using type = double2;
};
template <>
struct pair_of<float>
{
    using type = float2;
};

// workgroup-wide sum (any order) for the non-strict long-row path
template <typename T, int BLOCK>
__device__ __forceinline__ T block_sum(T v, T *scratch)
{
    for(int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if((threadIdx.x & 63) == 0)
        scratch[wave] = v;
    __syncthreads();
    T r = T(0);
    if(threadIdx.x == 0)
        for(int w = 0; w < BLOCK / 64; w++)
            r += scratch[w];
    return r;
}

// Lane exchange inside a row of 16 lanes by DPP (no LDS crossbar): ctrl 0xB1 = quad_perm [1,0,3,2], 0x4E = quad_perm [2,3,0,1],
// 0x141 = row_half_mirror, 0x140 = row_mirror.  Every lane is a valid source for these four, so no bound_ctrl fill is seen.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo     = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi     = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double read_lane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float read_lane(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// sum over the 64 lanes of a wavefront in a FIXED order (pairs, quads, halves of a row, rows of 16, then (r0 + r1) + (r2 + r3));
// every lane returns the total
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// rows of an LDS tile with at least SPMV_TREE_MIN entries (internal.hpp) are summed by a whole wavefront (auto mode, VAR 0) instead
// of one lane's chain: lane l takes entries l, l + 64, ... as an FMA chain, wave_sum adds the 64 partial sums
constexpr int TREE_MIN = SPMV_TREE_MIN;

// flags: bit0 strict long rows, bit1 16-byte-aligned val and col (quad loads allowed),
//        bit2 XCD-contiguous block order, bit3 non-temporal y stores, bit4 every row <= 8 entries,
//        bit5 VALIDATE: the block table may be STALE (a cached plan of a raw-array call, spmv_api.cpp: the cache key is the
//        row_ptr address, m, nnz and the base -- not the contents).  The kernel relies on exactly one thing from a block's
//        entry: that its rows [r0, r0 + nrows) hold the entries [p0, p0 + cnt).  The live row_ptr values of those rows are
//        loaded anyway (s_row), so each workgroup checks its own two boundaries for free; on a mismatch it computes its rows
//        straight from the live arrays (same chains, no LDS tile) and raises *stale for the host's next call.  No check
//        kernel, no stream round trip per call (round 3 paid ~70 us for one: 0.334 vs 0.262 ms on the 4096^2 Laplacian).
//        bit6 the blocks in DESCENDING order: every second product of a plan whose blocks are in row order (spmv_api.cpp) -- the end
//        of one sweep over the matrix is still in the 256 MB Infinity Cache when the next one starts there (round 5; see
//        sell_kernels.hip, where the gain is larger); the bits do not depend on the order.
// VAR: 0 general (auto mode: wavefront tree for scalar-order rows of >= 32 entries); 1 strict: every row in the reference's order (the tile-by-tile chain of the longest rows keeps the next tile in registers); 2 every row of the
// plan has <= 8 entries (stencils: no batched-read code, no long-row code).  Template parameters, not run-time flags: the register
// allocation of a kernel is the maximum over ALL its paths, and the strict path's prefetch registers and the batched reads of
// the general path had cost the 4096^2 Laplacian -- which uses neither -- a fifth of its speed (0.250 ms in round 1, 0.266 with
// the batched reads, 0.330 with the strict prefetch: 96 VGPRs = 5 waves per SIMD; tools/history/exp_bisect_adaptive.py, profiles/r4).
template <typename T, int ORDER, int TILE, int BLOCK, bool TRACE = false, int VAR = 0>
__global__ __launch_bounds__(BLOCK) void csr_adaptive_kernel(const int2 *__restrict__ blocks,
                                                                  const aoclsparse_int *__restrict__ row_ptr,
                                                                  const aoclsparse_int *__restrict__ col,
                                                                  const T *__restrict__ val,
                                                                  const T *__restrict__ x,
                                                                  T *__restrict__ y,
                                                                  T   alpha,
                                                                  T   beta,
                                                                  int base,
                                                                  int nblocks,
                                                                  int chunk,
                                                                  int flags,
                                                                  unsigned long long *trace,
                                                                  const int4 *__restrict__ blocks4,
                                                                  unsigned int *stale)
{
    constexpr int MAXROWS = spmv_maxrows(TILE); // planner guarantees rows <= MAXROWS
    // diagnostic (AOCLSPARSE_MI355_SPMV_TRACE, tools/spmv_trace.py): 100 MHz stamps per workgroup, kept in registers and
    // stored by thread 0 at the very end -- start / block table read / tile in LDS / after the barrier / rows reduced
    // (a template parameter, not a run-time test: with "if(trace)" around the stamps the product kernel was 6 % slower on
    // the 4096^2 Laplacian, 0.288 vs 0.272 ms, although no stamp was ever taken)
    unsigned long long t_st[4] = {0, 0, 0, 0};
    if constexpr(TRACE)
        t_st[0] = __builtin_amdgcn_s_memrealtime();
    __shared__ __attribute__((aligned(16))) T s_val[TILE + 4];
    __shared__ __attribute__((aligned(16))) T s_x[TILE + 4];
    __shared__ aoclsparse_int s_row[MAXROWS + 1];
    using P2          = typename pair_of<T>::type;
    constexpr int L   = lanes_of<ORDER>::value;
    const int     tid = threadIdx.x;
    const T      *xb  = x - base; // gathers take the raw column values (the base is folded into the pointer once)
    // XCD-aware order: workgroups with equal blockIdx%8 share an XCD (one L2); give each XCD a
    // contiguous eighth of the row blocks so its x windows stay in its own L2.
    const int b = (flags & 4) ? (blockIdx.x & 7) * chunk + (blockIdx.x >> 3) : (flags & 64) ? nblocks - 1 - (int)blockIdx.x : (int)blockIdx.x;
    if(b >= nblocks)
        return;
    int r0, p0, nrows, cnt;
    if(blocks4) // heavy-first order (SpmvPlan::rowblocks4): one 16-byte entry per block
    {
        const int4 e = blocks4[b];
        r0 = e.x, p0 = e.y, nrows = e.z, cnt = e.w;
    }
    else
    {
        const int2 e0 = blocks[b], e1 = blocks[b + 1];
        r0 = e0.x, p0 = e0.y, nrows = e1.x - r0, cnt = e1.y - p0;
    }
    if constexpr(TRACE)
        t_st[1] = __builtin_amdgcn_readfirstlane(cnt) >= 0 ? __builtin_amdgcn_s_memrealtime() : 0;
    if((flags & 32) && cnt > TILE)
    {
        // VALIDATE, single-row block: the live extent of the row replaces the planned one (the long-row code below works for
        // any length)
        const int ls = row_ptr[r0] - base, le = row_ptr[r0 + 1] - base;
        if(ls != p0 || le - ls != cnt)
        {
            if(tid == 0)
                __hip_atomic_store(stale, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            p0 = ls, cnt = le - ls;
        }
    }

    if(cnt <= TILE)
    {
        // ---- phase 1: coalesced stream of the block into LDS ----------------------------------------
        // window start rounded down to 4 non-zeros when the arrays allow 16-byte loads (flags bit1):
        // the <= 3 extra leading entries belong to the previous row and are never reduced
        const int w0   = (flags & 2) ? (p0 & ~3) : p0;
        const int cntw = cnt + (p0 - w0);
        for(int i = tid; i <= nrows; i += BLOCK)
            s_row[i] = row_ptr[r0 + i] - base - w0;
        if(flags & 2)
        {
#pragma unroll
            for(int k = 0; k < TILE / (4 * BLOCK); k++)
            {
                const int i = 4 * (tid + k * BLOCK);
                if(i + 3 < cntw)
                {
                    const int4 c = *reinterpret_cast<const int4 *>(col + w0 + i);
                    if constexpr(sizeof(T) == 8)
                    {
                        const P2 va = *reinterpret_cast<const P2 *>(val + w0 + i);
                        const P2 vb = *reinterpret_cast<const P2 *>(val + w0 + i + 2);
                        s_val[i]     = va.x;
                        s_val[i + 1] = va.y;
                        s_val[i + 2] = vb.x;
                        s_val[i + 3] = vb.y;
                    }
                    else
                    {
                        const float4 vq = *reinterpret_cast<const float4 *>(val + w0 + i);
                        s_val[i]     = vq.x;
                        s_val[i + 1] = vq.y;
                        s_val[i + 2] = vq.z;
                        s_val[i + 3] = vq.w;
                    }
                    s_x[i]     = xb[c.x];
                    s_x[i + 1] = xb[c.y];
                    s_x[i + 2] = xb[c.z];
                    s_x[i + 3] = xb[c.w];
                }
                else
                {
                    for(int q = i; q < cntw && q < i + 4; q++)
                    {
                        s_val[q] = val[w0 + q];
                        s_x[q]   = xb[col[w0 + q]];
                    }
                }
            }
            // the window shift can push a full tile up to 3 entries past index TILE-1
            if(tid < 3 && TILE + tid < cntw)
            {
                s_val[TILE + tid] = val[w0 + TILE + tid];
                s_x[TILE + tid]   = xb[col[w0 + TILE + tid]];
            }
        }
        else
        {
#pragma unroll
            for(int k = 0; k < TILE / BLOCK; k++)
            {
                const int i = tid + k * BLOCK;
                if(i < cntw)
                {
                    s_val[i] = val[w0 + i];
                    s_x[i]   = xb[col[w0 + i]];
                }
            }
        }
        if constexpr(TRACE)
            t_st[2] = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if constexpr(TRACE)
            t_st[3] = __builtin_amdgcn_s_memrealtime();
        const int grp  = tid / L;
        const int lane = tid % L;
        if((flags & 32) && (s_row[0] + w0 != p0 || s_row[nrows] + w0 != p0 + cnt))
        {
            // VALIDATE: this block's entry does not describe the live matrix.  Its rows, straight from the live arrays, in the
            // same order as below (scalar chain, or L strided chains + the reference's reduction + scalar tail).
            if(tid == 0)
                __hip_atomic_store(stale, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for(int rr = grp; rr < nrows; rr += BLOCK / L)
            {
                const int s = s_row[rr] + w0, e = s_row[rr + 1] + w0, r = r0 + rr;
                T         acc = T(0);
                if constexpr(L == 1)
                {
                    for(int j = s; j < e; j++)
                        acc = dev_fma(val[j], xb[col[j]], acc);
                    y[r] = finish(acc, alpha, beta, &y[r]);
                }
                else
                {
                    const int nfull = (e - s) & ~(L - 1);
                    for(int j = s + lane; j < s + nfull; j += L)
                        acc = dev_fma(val[j], xb[col[j]], acc);
                    T res = group_reduce<T, ORDER>(acc);
                    if(lane == 0)
                    {
                        if(nfull == 0)
                            res = T(0);
                        for(int j = s + nfull; j < e; j++)
                            res = dev_fma(val[j], xb[col[j]], res);
                        y[r] = finish(res, alpha, beta, &y[r]);
                    }
                }
            }
            return;
        }
        // ---- phase 2: per-row reduction ------------------------------------------------------------------
        if constexpr(L == 1 && VAR == 0)
        {
            // Auto mode, scalar order: a row of fewer than TREE_MIN entries is one lane's left-to-right FMA chain (the bits of
            // ref_csrmv_gn, csrmv_kr.hpp:448-513); a longer one is summed by its WAVEFRONT -- 64 strided chains + wave_sum --
            // because as one lane's chain a 300-entry row outlasted everything else its workgroup did (traced: 6-7.5 us of a
            // 9.7 us kernel on the circuit-like stand-in).  Componentwise bound of such a row: (ceil(n / 64) + 7) eps sum|a||x|,
            // inside the (2 ceil(log2 n) + 4) eps sum|a||x| the tests state; a pinned kid (VAR 1) keeps every row a chain.
            const int lane64 = tid & 63;
            for(int rb = 0; rb < nrows; rb += BLOCK)
            {
                const int  rr  = rb + tid;
                const bool act = rr < nrows;
                int        s = 0, e = 0;
                if(act)
                    s = s_row[rr], e = s_row[rr + 1];
                const bool lng = e - s >= TREE_MIN;
                if(act && !lng)
                {
                    T   acc = T(0);
                    int j   = s;
                    while(e - j >= 8)
                    {
                        T a8[8], b8[8];
#pragma unroll
                        for(int q = 0; q < 8; q++)
                        {
                            a8[q] = s_val[j + q];
                            b8[q] = s_x[j + q];
                        }
#pragma unroll
                        for(int q = 0; q < 8; q++)
                            acc = dev_fma(a8[q], b8[q], acc);
                        j += 8;
                    }
                    if(j < e) // the last 1..7 entries: one clamped batch
                    {
                        T a[7], b[7];
#pragma unroll
                        for(int q = 0; q < 7; q++)
                        {
                            const int jj = min(j + q, e - 1);
                            a[q]         = s_val[jj];
                            b[q]         = s_x[jj];
                        }
#pragma unroll
                        for(int q = 0; q < 7; q++)
                            acc = j + q < e ? dev_fma(a[q], b[q], acc) : acc;
                    }
                    store_y(&y[r0 + rr], finish(acc, alpha, beta, &y[r0 + rr]), flags);
                }
                unsigned long long mask = __ballot(lng);
                while(mask)
                {
                    const int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int ls = __builtin_amdgcn_readlane(s, l), le = __builtin_amdgcn_readlane(e, l);
                    T         acc = T(0);
                    for(int j = ls + lane64; j < le; j += 64)
                        acc = dev_fma(s_val[j], s_x[j], acc);
                    const T tot = wave_sum(acc);
                    if(lane64 == 0)
                    {
                        const int r = r0 + rb + (tid & ~63) + l;
                        store_y(&y[r], finish(tot, alpha, beta, &y[r]), flags);
                    }
                }
            }
        }
        else
        for(int rr = grp; rr < nrows; rr += BLOCK / L)
        {
            const int s   = s_row[rr];
            const int e   = s_row[rr + 1];
            const int r   = r0 + rr;
            T         acc = T(0);
            if constexpr(L == 1)
            {
                // same left-to-right chain, but the LDS reads of 8 entries are issued together (indices clamped to the
                // row, FMAs beyond it predicated off), so that a row pays the LDS latency once per 8 entries instead of
                // once per entry: traced on the circuit-like stand-in (tools/spmv_trace.py), the entry-by-entry loop
                // was 1.8 us of a workgroup's 5.4 us (rows of 1-15 entries: ~100 cycles of LDS latency per FMA)
                int j = s;
                if constexpr(VAR == 2)
                {
                    // every row is short: the plain loop (right for any length; the batched reads below read 14 LDS words for a
                    // 5-entry row instead of 10)
                    for(; j < e; j++)
                        acc = dev_fma(s_val[j], s_x[j], acc);
                }
                else
                {
                if(e - j >= 9) // full batches: no clamps, no predicates on the chain.  Values and x's are read in PAIRS
                {              // (one LDS instruction per entry instead of two: a lone wavefront issues an instruction
                               // every ~3.5 ns, so a 330-entry row was 6-7 us of reads + FMAs) and the reads of batch
                               // k + 1 are issued before the FMAs of batch k
                    if(j & 1) // pairs start at even LDS indices
                    {
                        acc = dev_fma(s_val[j], s_x[j], acc);
                        j++;
                    }
                    // two register sets, used alternately (a rotating copy doubled the loop's instruction count)
                    P2   a[4], b[4], an[4], bn[4];
                    auto rd = [&](P2(&va)[4], P2(&vb)[4], int at) {
#pragma unroll
                        for(int q = 0; q < 4; q++)
                        {
                            va[q] = *reinterpret_cast<const P2 *>(&s_val[at + 2 * q]);
                            vb[q] = *reinterpret_cast<const P2 *>(&s_x[at + 2 * q]);
                        }
                    };
                    auto mac = [&](const P2(&va)[4], const P2(&vb)[4]) {
#pragma unroll
                        for(int q = 0; q < 4; q++)
                        {
                            acc = dev_fma(va[q].x, vb[q].x, acc);
                            acc = dev_fma(va[q].y, vb[q].y, acc);
                        }
                    };
                    rd(a, b, j);
                    j += 8;
                    while(j + 16 <= e)
                    {
                        // (scheduling barriers: hipcc otherwise rotates the loop so that a batch's reads are consumed in the trip
                        // that issues them -- two to three exposed LDS round trips per 16 entries of a lone wavefront)
                        rd(an, bn, j);
                        __builtin_amdgcn_sched_barrier(0);
                        mac(a, b);
                        __builtin_amdgcn_sched_barrier(0);
                        rd(a, b, j + 8);
                        __builtin_amdgcn_sched_barrier(0);
                        mac(an, bn);
                        __builtin_amdgcn_sched_barrier(0);
                        j += 16;
                    }
                    if(j + 8 <= e)
                    {
                        rd(an, bn, j);
                        mac(a, b);
                        mac(an, bn);
                        j += 8;
                    }
                    else
                        mac(a, b);
                }
                if(e - j >= 8) // (a row of exactly 8, or 8 left after an odd start)
                {
                    T a8[8], b8[8];
#pragma unroll
                    for(int q = 0; q < 8; q++)
                    {
                        a8[q] = s_val[j + q];
                        b8[q] = s_x[j + q];
                    }
#pragma unroll
                    for(int q = 0; q < 8; q++)
                        acc = dev_fma(a8[q], b8[q], acc);
                    j += 8;
                }
                if(j < e) // the last 1..7 entries: one clamped batch
                {
                    T a[7], b[7];
#pragma unroll
                    for(int q = 0; q < 7; q++)
                    {
                        const int jj = min(j + q, e - 1);
                        a[q]         = s_val[jj];
                        b[q]         = s_x[jj];
                    }
#pragma unroll
                    for(int q = 0; q < 7; q++)
                        acc = j + q < e ? dev_fma(a[q], b[q], acc) : acc;
                }
                } // VAR != 2
                store_y(&y[r], finish(acc, alpha, beta, &y[r]), flags);
            }
            else
            {
                const int nfull = (e - s) & ~(L - 1);
                for(int j = s + lane; j < s + nfull; j += L)
                    acc = dev_fma(s_val[j], s_x[j], acc);
                T res = group_reduce<T, ORDER>(acc);
                if(lane == 0)
                {
                    if(nfull == 0)
                        res = T(0);
                    for(int j = s + nfull; j < e; j++)
                        res = dev_fma(s_val[j], s_x[j], res);
                    store_y(&y[r], finish(res, alpha, beta, &y[r]), flags);
                }
            }
        }
    }
    else if constexpr(VAR != 2) // (a plan whose rows all have <= 8 entries holds no block of more than TILE entries)
    {
        // ---- long row: this workgroup owns the single row r0 ---------------------------------------------
        const int n = cnt;
        if constexpr(VAR == 1)
        {
            // Tiles through LDS, the owner lanes chain them in the reference's order.  Round 3: the NEXT tile's values and x
            // are requested (into registers) before the owner lanes start on the current one, and the chain reads LDS in
            // batches of 8 -- a row then costs its FMA chain (~8 cycles per entry), not a memory round trip + an LDS round trip
            // per entry: web-like (91 rows of up to 2,908 entries) 59 -> 44.5 us with AOCLSPARSE_MI355_STRICT_LONG=8192 (28.5 us without:
            // the longest chain, ~15 ns per entry, now IS the kernel; 16-entry steps with paired reads and two alternating register
            // sets measured 50 us).
            const int nfull = n & ~(L - 1);
            T         acc   = T(0);
            constexpr int EPT = TILE / BLOCK;
            T             nv[EPT], nx[EPT];
            auto          fetch = [&](int t0) {
                const int tn = min(TILE, nfull - t0);
#pragma unroll
                for(int k = 0; k < EPT; k++)
                {
                    const int i = tid + k * BLOCK;
                    nv[k] = T(0), nx[k] = T(0);
                    if(i < tn)
                    {
                        nv[k] = val[p0 + t0 + i];
                        nx[k] = xb[col[p0 + t0 + i]];
                    }
                }
            };
            if(nfull > 0)
                fetch(0);
            for(int t0 = 0; t0 < nfull; t0 += TILE)
            {
                const int tn = min(TILE, nfull - t0);
                __syncthreads(); // the owner lanes are done with the previous tile
#pragma unroll
                for(int k = 0; k < EPT; k++)
                {
                    const int i = tid + k * BLOCK;
                    if(i < tn)
                        s_val[i] = nv[k], s_x[i] = nx[k];
                }
                __syncthreads();
                if(t0 + TILE < nfull)
                    fetch(t0 + TILE); // in flight while the chain below runs
                if(tid < L)
                {
                    int j = tid;
                    for(; j + 7 * L < tn; j += 8 * L)
                    {
                        T a[8], b[8];
#pragma unroll
                        for(int q = 0; q < 8; q++)
                            a[q] = s_val[j + q * L], b[q] = s_x[j + q * L];
#pragma unroll
                        for(int q = 0; q < 8; q++)
                            acc = dev_fma(a[q], b[q], acc);
                    }
                    for(; j < tn; j += L)
                        acc = dev_fma(s_val[j], s_x[j], acc);
                }
            }
            if(tid < 64) // first wavefront: owner lanes 0..L-1 hold the chains
            {
                T res = acc;
                if constexpr(L > 1)
                    res = group_reduce<T, ORDER>(acc);
                if(tid == 0)
                {
                    if(nfull == 0)
                        res = T(0);
                    for(int j = nfull; j < n; j++)
                        res = dev_fma(val[p0 + j], xb[col[p0 + j]], res);
                    y[r0] = finish(res, alpha, beta, &y[r0]);
                }
            }
        }
        else
        {
            // wavefront tree: same flops, order differs (bound stated in DESIGN.md / tests).  Eight strided
            // entries per lane are fetched together (values + columns, then the x gathers): a 3,000-entry row
            // costs 3 memory round trips per lane instead of 23
            T   acc = T(0);
            int j   = tid;
            for(; j + 7 * BLOCK < n; j += 8 * BLOCK)
            {
                T   a[8], xv[8];
                int c[8];
#pragma unroll
                for(int q = 0; q < 8; q++)
                    a[q] = val[p0 + j + q * BLOCK], c[q] = col[p0 + j + q * BLOCK];
#pragma unroll
                for(int q = 0; q < 8; q++)
                    xv[q] = xb[c[q]];
#pragma unroll
                for(int q = 0; q < 8; q++)
                    acc = dev_fma(a[q], xv[q], acc);
            }
            for(; j < n; j += BLOCK)
                acc = dev_fma(val[p0 + j], xb[col[p0 + j]], acc);
            T res = block_sum<T, BLOCK>(acc, s_val);
            if(tid == 0)
                y[r0] = finish(res, alpha, beta, &y[r0]);
        }
    }
    if(TRACE && tid == 0)
    {
        unsigned long long *tr = trace + 8 * (size_t)b;
        tr[0] = t_st[0], tr[1] = t_st[1], tr[2] = t_st[2], tr[3] = t_st[3], tr[4] = __builtin_amdgcn_s_memrealtime();
        tr[5] = ((unsigned long long)(unsigned)nrows << 32) | (unsigned)cnt;
        tr[6] = (unsigned long long)r0, tr[7] = 0;
    }
}

template <typename T>
__global__ void scale_kernel(T *y, aoclsparse_int n, T beta)
{
    // level2/aoclsparse_mv_helpers.hpp:31-51 (vscale): beta == 0 writes zeros without reading y
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        y[i] = beta != T(0) ? beta * y[i] : T(0);
}

template <typename T>
__global__ void gather_strided_kernel(const T *src, aoclsparse_int inc, aoclsparse_int n, T *dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        dst[i] = src[(size_t)i * inc];
}

template <typename T>
__global__ void scatter_strided_kernel(const T *src, aoclsparse_int n, T *dst, aoclsparse_int inc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        dst[(size_t)i * inc] = src[i];
}

// d = sum_i x[i]*y[i] for ?dotmv (level2/aoclsparse_dotmv.hpp:47-70): fixed two-stage tree, so the result
// is deterministic (the reference's KT kernel sums SIMD lanes; parity is a tolerance)
constexpr int DOT_BLOCKS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void dot_partial_kernel(const T *__restrict__ x, const T *__restrict__ y,
                                                          aoclsparse_int n, T *partial)
{
    __shared__ T sh[4];
    T            acc = T(0);
    for(long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        acc = dev_fma(x[i], y[i], acc);
    for(int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    if((threadIdx.x & 63) == 0)
        sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if(threadIdx.x == 0)
        partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
template <typename T>
__global__ __launch_bounds__(256) void dot_final_kernel(const T *__restrict__ partial, int count, T *d)
{
    __shared__ T sh[4];
    T            acc = T(0);
    for(int i = threadIdx.x; i < count; i += 256)
        acc += partial[i];
    for(int off = 32; off > 0; off >>= 1)
        acc += __shfl_down(acc, off, 64);
    if((threadIdx.x & 63) == 0)
        sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if(threadIdx.x == 0)
        *d = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename T>
aoclsparse_status launch_dot(hipStream_t s, aoclsparse_int n, const T *x, const T *y, T *partial, T *d)
{
    const int blocks = n <= 0 ? 1 : (int)std::min<long long>(DOT_BLOCKS, ((long long)n + 255) / 256);
    hipLaunchKernelGGL((dot_partial_kernel<T>), dim3(blocks), dim3(256), 0, s, x, y, n, partial);
    hipLaunchKernelGGL((dot_final_kernel<T>), dim3(1), dim3(256), 0, s, partial, blocks, d);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T, int ORDER, int TILE, int BLOCK, int VAR>
static void launch_inst(hipStream_t s, int flags, int base, T alpha, const T *val, const aoclsparse_int *col,
                        const aoclsparse_int *row_ptr, const aoclsparse_int *blocks, aoclsparse_int nblocks,
                        const T *x, T beta, T *y, const aoclsparse_int *blocks4, unsigned int *stale)
{
    const int chunk = (nblocks + 7) / 8;
    const int grid  = (flags & 4) ? chunk * 8 : (int)nblocks;
    // diagnostic: AOCLSPARSE_MI355_SPMV_TRACE=<file> dumps 8 x u64 per row block of the LAST launch (synchronises)
    static const char  *trace_path = getenv("AOCLSPARSE_MI355_SPMV_TRACE");
    unsigned long long *trace      = nullptr;
    if(trace_path && hipMalloc(&trace, sizeof(unsigned long long) * 8 * (size_t)nblocks) != hipSuccess)
        trace = nullptr;
    if constexpr(ORDER == 0) // the traced build exists for the scalar order only
    {
        if(trace)
            hipLaunchKernelGGL((csr_adaptive_kernel<T, ORDER, TILE, BLOCK, true, VAR>), dim3(grid), dim3(BLOCK), 0, s,
                               reinterpret_cast<const int2 *>(blocks), row_ptr, col, val, x, y, alpha, beta, base,
                               (int)nblocks, chunk, flags, trace, reinterpret_cast<const int4 *>(blocks4), stale);
    }
    else if(trace)
    {
        (void)hipFree(trace);
        trace = nullptr;
    }
    if(!(ORDER == 0 && trace))
        hipLaunchKernelGGL((csr_adaptive_kernel<T, ORDER, TILE, BLOCK, false, VAR>), dim3(grid), dim3(BLOCK), 0, s,
                           reinterpret_cast<const int2 *>(blocks), row_ptr, col, val, x, y, alpha, beta, base,
                           (int)nblocks, chunk, flags, (unsigned long long *)nullptr, reinterpret_cast<const int4 *>(blocks4), stale);
    if(trace)
    {
        std::vector<unsigned long long> host(8 * (size_t)nblocks);
        if(hipStreamSynchronize(s) == hipSuccess
           && hipMemcpy(host.data(), trace, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) == hipSuccess)
            if(FILE *f = fopen(trace_path, "wb"))
            {
                fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
                fclose(f);
            }
        (void)hipFree(trace);
    }
}

// tile: 512 (128-thread workgroups), 1024 or 2048 (256 threads); bit0 set = XCD-contiguous block order
template <typename T>
aoclsparse_status launch_csrmv(hipStream_t s, int order, bool strict, int tile, int base, T alpha,
                               aoclsparse_int m, const T *val, const aoclsparse_int *col,
                               const aoclsparse_int *row_ptr, const aoclsparse_int *blocks,
                               aoclsparse_int nblocks, const T *x, T beta, T *y, const aoclsparse_int *blocks4,
                               aoclsparse_int max_row_nnz, unsigned int *stale)
{
    if(m <= 0 || nblocks <= 0)
        return aoclsparse_status_success;
    const bool xcd = (tile & 1) != 0, rev = (tile & 2) != 0;
    tile &= ~3;
    int flags = strict ? 1 : 0;
    if(reinterpret_cast<uintptr_t>(val) % 16 == 0 && reinterpret_cast<uintptr_t>(col) % 16 == 0)
        flags |= 2;
    if(xcd)
        flags |= 4;
    if((size_t)m * sizeof(T) > (size_t)32 << 20)
        flags |= 8;
    // kernel variant (template parameter VAR): strict long rows / every row short (stencils) / general
    const int var = strict ? 1 : (max_row_nnz <= 8 ? 2 : 0);
    if(stale)
        flags |= 32; // the block table is a cached plan of a raw-array call: every workgroup validates its own entry
    if(rev && !xcd)
        flags |= 64; // blocks in descending order (the caller alternates: the end of one sweep is still in the Infinity Cache)
    const int tsel = tile == 512 ? 0 : (tile == 1024 ? 1 : (tile == 2048 ? 2 : -1));
    if(tsel < 0 || order < 0 || order > 2)
        return aoclsparse_status_invalid_kid;
#define MI355_CASE(O, TL, BL)                                                                                                \
    if(var == 1)                                                                                                             \
        launch_inst<T, O, TL, BL, 1>(s, flags, base, alpha, val, col, row_ptr, blocks, nblocks, x, beta, y, blocks4, stale); \
    else if(var == 2)                                                                                                        \
        launch_inst<T, O, TL, BL, 2>(s, flags, base, alpha, val, col, row_ptr, blocks, nblocks, x, beta, y, blocks4, stale); \
    else                                                                                                                     \
        launch_inst<T, O, TL, BL, 0>(s, flags, base, alpha, val, col, row_ptr, blocks, nblocks, x, beta, y, blocks4, stale); \
    break
    switch(order * 3 + tsel)
    {
    case 0:
        MI355_CASE(0, 512, 128);
    case 1:
        MI355_CASE(0, 1024, 256);
    case 2:
        MI355_CASE(0, 2048, 256);
    case 3:
        MI355_CASE(1, 512, 128);
    case 4:
        MI355_CASE(1, 1024, 256);
    case 5:
        MI355_CASE(1, 2048, 256);
    case 6:
        MI355_CASE(2, 512, 128);
    case 7:
        MI355_CASE(2, 1024, 256);
    case 8:
        MI355_CASE(2, 2048, 256);
    }
#undef MI355_CASE
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}


template <typename T>
__global__ void waxpby_kernel(aoclsparse_int n, T a, const T *x, T b, const T *y, T *w)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        w[i] = a * x[i] + b * y[i]; // built with -ffp-contract=off: two products, one sum
}

template <typename T>
aoclsparse_status launch_waxpby(hipStream_t s, aoclsparse_int n, T a, const T *x, T b, const T *y, T *w)
{
    if(n <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((waxpby_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, n, a, x, b, y, w);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_scale(hipStream_t s, T *y, aoclsparse_int n, T beta)
{
    if(n <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((scale_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, y, n, beta);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_strided_gather(hipStream_t s, const T *src, aoclsparse_int inc,
                                        aoclsparse_int n, T *dst)
{
    if(n <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((gather_strided_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, src, inc, n, dst);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_strided_scatter(hipStream_t s, const T *src, aoclsparse_int n, T *dst,
                                         aoclsparse_int inc)
{
    if(n <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((scatter_strided_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, src, n, dst, inc);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_INSTANTIATE(T)                                                                                   \
    template aoclsparse_status launch_csrmv<T>(hipStream_t, int, bool, int, int, T, aoclsparse_int, const T *, \
                                               const aoclsparse_int *, const aoclsparse_int *,                  \
                                               const aoclsparse_int *, aoclsparse_int, const T *, T, T *,       \
                                               const aoclsparse_int *, aoclsparse_int, unsigned int *);         \
    template aoclsparse_status launch_scale<T>(hipStream_t, T *, aoclsparse_int, T);                           \
    template aoclsparse_status launch_waxpby<T>(hipStream_t, aoclsparse_int, T, const T *, T, const T *, T *); \
    template aoclsparse_status launch_dot<T>(hipStream_t, aoclsparse_int, const T *, const T *, T *, T *);     \
    template aoclsparse_status launch_strided_gather<T>(hipStream_t, const T *, aoclsparse_int,                 \
                                                        aoclsparse_int, T *);                                   \
    template aoclsparse_status launch_strided_scatter<T>(hipStream_t, const T *, aoclsparse_int, T *,           \
                                                         aoclsparse_int);
MI355_INSTANTIATE(double)
MI355_INSTANTIATE(float)

} // namespace mi355

using namespace mi355;

extern "C" {

aoclsparse_status mi355_dcsrmv(void *stream, aoclsparse_int order, aoclsparse_int strict, aoclsparse_int tile,
                               aoclsparse_int base, double alpha, aoclsparse_int m, const double *val,
                               const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                               const aoclsparse_int *blocks, aoclsparse_int nblocks, const double *x,
                               double beta, double *y)
{
    if(!val || !col || !row_ptr || !blocks || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(m < 0 || nblocks < 0)
        return aoclsparse_status_invalid_size;
    if((base != 0 && base != 1) || (tile != 512 && tile != 1024 && tile != 2048))
        return aoclsparse_status_invalid_value;
    return launch_csrmv<double>((hipStream_t)stream, order, strict != 0, tile, base, alpha, m, val, col, row_ptr,
                                blocks, nblocks, x, beta, y);
}

aoclsparse_status mi355_scsrmv(void *stream, aoclsparse_int order, aoclsparse_int strict, aoclsparse_int tile,
                               aoclsparse_int base, float alpha, aoclsparse_int m, const float *val,
                               const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                               const aoclsparse_int *blocks, aoclsparse_int nblocks, const float *x, float beta,
                               float *y)
{
    if(!val || !col || !row_ptr || !blocks || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(m < 0 || nblocks < 0)
        return aoclsparse_status_invalid_size;
    if((base != 0 && base != 1) || (tile != 512 && tile != 1024 && tile != 2048))
        return aoclsparse_status_invalid_value;
    return launch_csrmv<float>((hipStream_t)stream, order, strict != 0, tile, base, alpha, m, val, col, row_ptr,
                               blocks, nblocks, x, beta, y);
}

} // extern "C"
