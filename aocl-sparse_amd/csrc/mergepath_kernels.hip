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
//   carries  a tile whose entries after its last row end belong to a row that ends later publishes that HEAD PIECE
//            as two self-validating 8-byte granules {launch epoch, half of the value} (relaxed agent-scope atomics: sc1
//            stores / loads, no fence); the tile holding the row's END waits for the head pieces of the tiles
//            [first[w], w) -- a decoupled look-back: they have smaller block indices, so they are resident or done --
//            adds them (lane v % 64 takes tile v, then the wavefront tree), adds its own tail piece and writes y.
//            first[] is plan data (the host knows which tiles a row crosses).  Deterministic: the order of additions
//            depends on the tiling only.  Rounds 2-4 ran a second kernel (mp_fixup_kernel) for this.
// Rows cut by a tile boundary and rows of >= SPMV_TREE_MIN entries carry the forward-error bound instead of bit-exactness.
// Served: the scalar order only (nnz <= 10 m -- where irregular rows live) without a pinned kid.
// HBM bytes as CSR-Adaptive + 16 B per tile of granules + 4 B per tile of first[].
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

// head piece of tile w: granule(s) {epoch << 32 | 32 bits of the value}; each is ONE 8-byte store, so a reader that sees the
// epoch sees the bits that came with it (no fence, no flag: MI355X_MICROARCH.md, hand-off forms)
__device__ __forceinline__ void mp_publish(unsigned long long *gran, int w, unsigned epoch, double r)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(r), tag = (unsigned long long)epoch << 32;
    __hip_atomic_store(gran + 2 * (size_t)w, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(gran + 2 * (size_t)w + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mp_publish(unsigned long long *gran, int w, unsigned epoch, float r)
{
    __hip_atomic_store(gran + 2 * (size_t)w, ((unsigned long long)epoch << 32) | (unsigned)__float_as_int(r), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int MP_SPIN_LIMIT = 1 << 21; // polls before a look-back gives up (seconds; never seen: the tiles waited for are older)
__device__ __forceinline__ bool mp_fetch(const unsigned long long *gran, int v, unsigned epoch, double &out)
{
    for(int spin = 0; spin < MP_SPIN_LIMIT; spin++)
    {
        const unsigned long long g0 = __hip_atomic_load(gran + 2 * (size_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long g1 = __hip_atomic_load(gran + 2 * (size_t)v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if((unsigned)(g0 >> 32) == epoch && (unsigned)(g1 >> 32) == epoch)
        {
            out = __longlong_as_double((long long)((g1 << 32) | (g0 & 0xffffffffull)));
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}
__device__ __forceinline__ bool mp_fetch(const unsigned long long *gran, int v, unsigned epoch, float &out)
{
    for(int spin = 0; spin < MP_SPIN_LIMIT; spin++)
    {
        const unsigned long long g0 = __hip_atomic_load(gran + 2 * (size_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if((unsigned)(g0 >> 32) == epoch)
        {
            out = __int_as_float((int)(unsigned)g0);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
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
                                                      unsigned long long *__restrict__ gran, unsigned epoch)
{
    __shared__ T              s_val[MP_ITEMS];
    __shared__ T              s_x[MP_ITEMS];
    __shared__ aoclsparse_int s_row[MP_ITEMS + 2];
    __shared__ T              s_tail;
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
                s_tail = r; // completed by the look-back below
            else
                y[i0 + t] = mp_finish(r, alpha, beta, y + i0 + t);
        }
        else if(nonempty)
            mp_publish(gran, w, epoch, r);
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
    if(f0 >= 0) // (uniform) row i0 ends here and started in tile f0: head pieces of the tiles [f0, w), in a fixed order, + the tail
    {
        __syncthreads();
        if(tid < 64)
        {
            T    acc = T(0);
            bool ok  = true;
            for(int v = f0 + lane64; v < w; v += 64)
            {
                T piece = T(0);
                ok      = mp_fetch(gran, v, epoch, piece) && ok;
                acc += piece;
            }
            const T total = mp_wave_sum(acc) + s_tail;
            if(__ballot(!ok) != 0ull) // a look-back expired (never seen): say so in the result rather than hang or guess
            {
                if(lane64 == 0)
                    y[i0] = total * T(0) + (T)__builtin_nanf("");
            }
            else if(lane64 == 0)
                y[i0] = mp_finish(total, alpha, beta, y + i0);
        }
    }
}

} // namespace

template <typename T>
aoclsparse_status launch_mergepath(hipStream_t s, int base, T alpha, aoclsparse_int ntiles, const aoclsparse_int *starts,
                                   const aoclsparse_int *first, const T *val, const aoclsparse_int *col,
                                   const aoclsparse_int *row_ptr, const T *x, T beta, T *y, unsigned long long *granules,
                                   unsigned epoch)
{
    if(ntiles <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((mp_kernel<T>), dim3(ntiles), dim3(MP_BLOCK), 0, s, reinterpret_cast<const int2 *>(starts), first, row_ptr,
                       col, val, x, y, alpha, beta, base, granules, epoch);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_mergepath<double>(hipStream_t, int, double, aoclsparse_int, const aoclsparse_int *,
                                                    const aoclsparse_int *, const double *, const aoclsparse_int *,
                                                    const aoclsparse_int *, const double *, double, double *,
                                                    unsigned long long *, unsigned);
template aoclsparse_status launch_mergepath<float>(hipStream_t, int, float, aoclsparse_int, const aoclsparse_int *,
                                                   const aoclsparse_int *, const float *, const aoclsparse_int *,
                                                   const aoclsparse_int *, const float *, float, float *, unsigned long long *,
                                                   unsigned);

} // namespace mi355
