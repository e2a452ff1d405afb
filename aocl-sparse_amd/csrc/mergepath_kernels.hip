// mergepath_kernels.hip -- merge-path CSR SpMV: the balanced-work companion of the CSR-Adaptive kernel.  ONE launch (round 5).
//
// The work list of a CSR SpMV is the merge of the m row ends with the nnz non-zeros (Merrill & Garland's
// merge path).  It is cut into tiles of MP_ITEMS consecutive items -- whatever the row lengths, every
// workgroup gets the same number of rows + non-zeros, and a row longer than a tile is spread over several
// workgroups.  The tile coordinates {row ends consumed, non-zeros consumed} are found on the HOST at plan
// time (one binary search per tile, integer work: matrix.cpp build_merge_plan), so the kernel does no searching:
//   phase 1  the tile's values and gathered x are parked in LDS together with the tile's slice of row_ptr;
//   phase 2  a row (or the piece of a row that lies in this tile) of fewer than SPMV_TREE_MIN entries is one lane's
//            left-to-right FMA chain -- the reference's scalar order (csrmv_kr.hpp:448-513), so a short row inside one
//            tile is bit-identical; a longer one is summed by its wavefront (64 strided chains + a fixed-order tree),
//            as in csr_adaptive_kernel's auto mode.  Rounds 2-4 gave every piece to ONE lane: a tile in the middle of a
//            170 k-entry row was a 1,024-entry serial chain, ~13 us, and that chain WAS the kernel (26 us);
//   carries  a tile whose entries after its last row end belong to a row that ends later stores that HEAD PIECE in its slot
//            of the stream's piece set and adds 1 to the counter of the tile that holds the row's END; so does the end tile
//            with its TAIL piece.  Whoever adds LAST sums the head pieces of the tiles [first[w], w) (lane v % 64 takes tile
//            v, then the wavefront tree), adds the tail piece, resets the counter and writes y.  first[] / endt[] are plan
//            data (the host knows which tiles a row crosses).  Deterministic: the order of additions depends on the tiling
//            only.  Nobody waits for anybody (round 6; round 5's end tile polled epoch-tagged granules: correct only while
//            workgroups start in index order, and a captured launch replayed its epoch).  Rounds 2-4: a second kernel.
// Rows cut by a tile boundary and rows of >= SPMV_TREE_MIN entries carry the forward-error bound instead of bit-exactness.
// Served: the scalar order only (nnz <= 10 m -- where irregular rows live) without a pinned kid.
// HBM bytes as CSR-Adaptive + 20 B per tile of pieces / counters + 8 B per tile of first[] / endt[].
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

namespace
{

constexpr int MP_BLOCK = 256;

__device__ __forceinline__ double mp_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float mp_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}
template <typename T>
__device__ __forceinline__ T mp_finish(T r, T alpha, T beta, const T *yi)
{
    if(alpha != T(1))
        r = alpha * r;
    if(beta != T(0))
        r = mp_fma(beta, *yi, r);
    return r;
}

// 64-lane sum in a fixed order (DPP inside rows of 16 lanes, then (r0 + r1) + (r2 + r3)); every lane gets the total
template <int CTRL>
__device__ __forceinline__ double mp_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo     = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi     = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float mp_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double mp_lane(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ float mp_lane(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <typename T>
__device__ __forceinline__ T mp_wave_sum(T v)
{
    v += mp_dpp<0xB1>(v); // quad_perm [1,0,3,2]
    v += mp_dpp<0x4E>(v); // quad_perm [2,3,0,1]
    v += mp_dpp<0x141>(v); // row_half_mirror
    v += mp_dpp<0x140>(v); // row_mirror
    return (mp_lane(v, 0) + mp_lane(v, 16)) + (mp_lane(v, 32) + mp_lane(v, 48));
}

// Pieces of a row that is cut by tile boundaries meet through three per-tile arrays of the stream's piece set:
//   head[v]  the head piece of tile v (the part of the row that ENDS LATER, in tile endt[v]),
//   tail[w]  the tail piece of the row that ends in tile w and started in tile first[w] < w,
//   cnt[w]   arrivals for that row: every tile of [first[w], w] adds 1 after its piece has been written.
// Nobody waits: the tile whose add comes LAST (it reads w - first[w] back) sums the pieces in the fixed order below, resets the
// counter and writes y.  So the kernel needs no assumption about the order in which workgroups are dispatched, carries no state
// from one launch to the next (a captured launch replays correctly whatever x is), and the sum does not depend on who finishes.
// Hand-off form (MI355X_MICROARCH.md, hand-offs with sc1 loads, first row): sc1 stores of the pieces by ONE lane -> that lane's
// s_waitcnt vmcnt(0) -> its agent-scope atomic add; the lane whose add came last loads the pieces with sc1 loads after the add
// has returned.
template <typename T>
__device__ __forceinline__ void mp_store(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T mp_load(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tile w owns the row ends [i0, i1) and the non-zeros [j0, j1) (0-based).  Slot t of the tile, t = 0 .. nr (nr = i1 - i0):
//   t < nr   the entries of row i0 + t that lie in this tile; slot 0 is a TAIL piece when row i0 started in an earlier tile
//            (first[w] >= 0: its head pieces come from the tiles [first[w], w))
//   t == nr  the HEAD piece: what follows the tile's last row end belongs to row i1, which ends in a later tile
template <typename T>
__global__ __launch_bounds__(MP_BLOCK) void mp_kernel(const int2 *__restrict__ starts, const aoclsparse_int *__restrict__ first,
                                                      const aoclsparse_int *__restrict__ row_ptr,
                                                      const aoclsparse_int *__restrict__ col,
                                                      const T *__restrict__ val, const T *__restrict__ x,
                                                      T *__restrict__ y, T alpha, T beta, int base,
                                                      T *__restrict__ pieces, unsigned *__restrict__ cnt, int ntiles)
{
    __shared__ T              s_val[MP_ITEMS];
    __shared__ T              s_x[MP_ITEMS];
    __shared__ aoclsparse_int s_row[MP_ITEMS + 2];
    __shared__ T              s_tail, s_head;
    const int  w   = blockIdx.x;
    const int  tid = threadIdx.x, lane64 = tid & 63;
    const int2 a = starts[w], b = starts[w + 1];
    const int  f0 = first[w];
    const int  i0 = a.x, j0 = a.y, i1 = b.x, j1 = b.y;
    const int  nr = i1 - i0, nz = j1 - j0;
    const T   *xb = x - base;
    for(int t = tid; t <= nr; t += MP_BLOCK)
        s_row[t] = row_ptr[i0 + t] - base; // start of row i0 .. start of row i1
    for(int t = tid; t < nz; t += MP_BLOCK)
    {
        s_val[t] = val[j0 + t];
        s_x[t]   = xb[col[j0 + t]];
    }
    __syncthreads();
    auto emit = [&](int t, int rs, bool nonempty, T r) {
        if(t < nr)
        {
            if(t == 0 && rs < j0)
                s_tail = r; // completed by the last arriver below
            else
                y[i0 + t] = mp_finish(r, alpha, beta, y + i0 + t);
        }
        else if(nonempty)
            s_head = r;
    };
    for(int rb = 0; rb <= nr; rb += MP_BLOCK)
    {
        const int  t   = rb + tid;
        const bool act = t <= nr;
        int        rs = 0, s = 0, e = 0;
        if(act)
        {
            rs = s_row[t];
            s  = max(rs, j0) - j0;
            e  = (t < nr ? s_row[t + 1] : j1) - j0;
        }
        const bool lng = act && e - s >= SPMV_TREE_MIN;
        if(act && !lng)
        {
            T   r = T(0);
            int p = s;
            for(; p + 8 <= e; p += 8) // one chain; the LDS reads of 8 entries are issued together
            {
                T av[8], xv[8];
#pragma unroll
                for(int q = 0; q < 8; q++)
                    av[q] = s_val[p + q], xv[q] = s_x[p + q];
#pragma unroll
                for(int q = 0; q < 8; q++)
                    r = mp_fma(av[q], xv[q], r);
            }
            for(; p < e; p++)
                r = mp_fma(s_val[p], s_x[p], r);
            emit(t, rs, e > s, r);
        }
        unsigned long long mask = __ballot(lng);
        while(mask)
        {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int ls = __builtin_amdgcn_readlane(s, l), le = __builtin_amdgcn_readlane(e, l);
            const int lrs = __builtin_amdgcn_readlane(rs, l);
            T         acc = T(0);
            for(int p = ls + lane64; p < le; p += 64)
                acc = mp_fma(s_val[p], s_x[p], acc);
            const T tot = mp_wave_sum(acc);
            if(lane64 == 0)
                emit(rb + (tid & ~63) + l, lrs, true, tot);
        }
    }
    // (uniform) a head piece exists when the tile ends past the first entry of row i1; a tail piece when row i0 ends here and
    // started in tile f0 < w
    const bool has_head = max(s_row[nr], j0) < j1, has_tail = f0 >= 0;
    if(!has_head && !has_tail)
        return;
    __syncthreads();
    if(tid >= 64)
        return;
    T  *head = pieces, *tail = pieces + ntiles;
    int fin_head = -1, fin_tail = -1; // end tiles whose row this workgroup completes
    if(lane64 == 0)
    {
        const int we = has_head ? first[ntiles + w] : -1; // the tile in which row i1 ends
        if(has_head)
            mp_store(head + w, s_head);
        if(has_tail)
            mp_store(tail + w, s_tail);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(has_head && __hip_atomic_fetch_add(cnt + we, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(we - first[we]))
            fin_head = we;
        if(has_tail && __hip_atomic_fetch_add(cnt + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(w - f0))
            fin_tail = w;
    }
    asm volatile("" ::: "memory"); // the piece loads below stay behind the adds
    fin_head = __builtin_amdgcn_readfirstlane(fin_head);
    fin_tail = __builtin_amdgcn_readfirstlane(fin_tail);
#pragma unroll
    for(int k = 0; k < 2; k++)
    {
        const int we = k == 0 ? fin_tail : fin_head; // (uniform)
        if(we < 0)
            continue;
        const int fe  = first[we];
        T         acc = T(0);
        for(int v = fe + lane64; v < we; v += 64)
            acc += mp_load(head + v);
        const T total = mp_wave_sum(acc) + mp_load(tail + we);
        if(lane64 == 0)
        {
            __hip_atomic_store(cnt + we, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch on this stream
            const int ie = starts[we].x;
            y[ie]        = mp_finish(total, alpha, beta, y + ie);
        }
    }
}

} // namespace

template <typename T>
aoclsparse_status launch_mergepath(hipStream_t s, int base, T alpha, aoclsparse_int ntiles, const aoclsparse_int *starts,
                                   const aoclsparse_int *first, const T *val, const aoclsparse_int *col,
                                   const aoclsparse_int *row_ptr, const T *x, T beta, T *y, void *pieces)
{
    if(ntiles <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((mp_kernel<T>), dim3(ntiles), dim3(MP_BLOCK), 0, s, reinterpret_cast<const int2 *>(starts), first, row_ptr,
                       col, val, x, y, alpha, beta, base, static_cast<T *>(pieces),
                       reinterpret_cast<unsigned *>(static_cast<unsigned long long *>(pieces) + 2 * (size_t)ntiles), (int)ntiles);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_mergepath<double>(hipStream_t, int, double, aoclsparse_int, const aoclsparse_int *,
                                                    const aoclsparse_int *, const double *, const aoclsparse_int *,
                                                    const aoclsparse_int *, const double *, double, double *, void *);
template aoclsparse_status launch_mergepath<float>(hipStream_t, int, float, aoclsparse_int, const aoclsparse_int *,
                                                   const aoclsparse_int *, const float *, const aoclsparse_int *,
                                                   const aoclsparse_int *, const float *, float, float *, void *);

} // namespace mi355
