// sell_kernels.hip -- SpMV on SELL-64, the layout aoclsparse_optimize builds for an mv hint on MI355X.
//
// The reference's optimize step re-stores a hinted matrix in the format its CPU kernels like best
// (br4 / ELLT-HYB / blocked CSR, analysis.cpp:146-382).  The GPU analogue is sliced ELL with one slice
// per 64-wide wavefront: slice s holds rows [64 s, 64 s + 64) column-major, cell (p, lane) at
// slice_ptr[s] + 64 p + lane, padded to the slice's longest row (column -1, value 0); matrices with >= 16
// non-zeros per row keep four consecutive cells of a row adjacent instead (PACK 4: cell at
// slice_ptr[s] + 256 (p/4) + 4 lane + p%4, width rounded up to a multiple of 4), so that a wavefront's load is
// one contiguous 2 KB piece -- the in-flight slices of a long-row matrix are otherwise 512-byte accesses
// strided by the slice size, which costs HBM page locality.  Lane i of a wave owns row i:
//   * every val / col access is one coalesced line per wavefront instruction, no row_ptr, no LDS;
//   * a lane walks its row front to back, so the reference's summation orders are reproduced exactly:
//     order 0 is the scalar FMA chain (csrmv_kr.hpp:448-513); orders 1 / 2 keep 4 / 8 partial sums per
//     lane for the full groups, reduce them as the AVX2 / AVX-512 kernels do, then run the scalar tail
//     (csrmv_kr.hpp:949-1040, csrmv_avx512.cpp:36-134, csrmv_kr.hpp:734-831 for float).
// Bytes per launch: cells*(8+4) + 8 m (y) (+ 8 m if beta != 0) (+ 4 m row lengths for orders 1 / 2) + x
// gathers; cells <= 1.15 nnz or the handle stays on the CSR-Adaptive kernel (matrix.cpp: build_sell).
#include "internal.hpp"

#include <type_traits>

#include <hip/hip_runtime.h>

namespace mi355
{

namespace
{

__device__ __forceinline__ double s_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float s_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}

template <typename T>
__device__ __forceinline__ T s_finish(T r, T alpha, T beta, const T *yi)
{
    if(alpha != T(1))
        r = alpha * r;
    if(beta != T(0))
        r = s_fma(beta, *yi, r);
    return r;
}

// complex handles (aoclsparse_{c,z}mv through the same SELL-64 copy, round 4): the component-wise multiply-add of
// complex_kernels.hip (c_mac), alpha == 1 and beta == 0 skipped exactly as there; CONJ conjugates the stored value at load
// (op = H on a general matrix, op = T on a hermitian one: complex_api.cpp)
template <typename R>
__device__ __forceinline__ cplx<R> s_fma(cplx<R> a, cplx<R> b, cplx<R> c)
{
    c.re = s_fma(a.re, b.re, c.re);
    c.re = s_fma(-a.im, b.im, c.re);
    c.im = s_fma(a.re, b.im, c.im);
    c.im = s_fma(a.im, b.re, c.im);
    return c;
}
template <typename R>
__device__ __forceinline__ cplx<R> s_finish(cplx<R> r, cplx<R> alpha, cplx<R> beta, const cplx<R> *yi)
{
    if(!(alpha.re == R(1) && alpha.im == R(0)))
        r = s_fma(alpha, r, cplx<R>(R(0), R(0)));
    if(!(beta.re == R(0) && beta.im == R(0))) // beta == 0 never reads y
        r = s_fma(beta, *yi, r);
    return r;
}
template <bool CONJ, typename T>
__device__ __forceinline__ T s_cj(T v)
{
    return v;
}
template <bool CONJ, typename R>
__device__ __forceinline__ cplx<R> s_cj(cplx<R> v)
{
    if constexpr(CONJ)
        v.im = -v.im;
    return v;
}

template <typename T>
__device__ __forceinline__ void s_store(T *p, T v, bool nt)
{
    if(nt)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
template <typename R>
__device__ __forceinline__ void s_store(cplx<R> *p, cplx<R> v, bool nt)
{
    if(nt)
    {
        __builtin_nontemporal_store(v.re, &p->re);
        __builtin_nontemporal_store(v.im, &p->im);
    }
    else
        *p = v;
}

// horizontal sums of the reference's vector kernels, on a lane's private partial sums
template <typename T, int G>
__device__ __forceinline__ T lanes_sum(const T (&l)[G])
{
    if constexpr(G == 4)
        return (l[0] + l[1]) + (l[2] + l[3]); // hadd, then lo + hi (csrmv_kr.hpp:1000-1016)
    else if constexpr(sizeof(T) == 8)
        return ((l[0] + l[4]) + (l[1] + l[5])) + ((l[2] + l[6]) + (l[3] + l[7])); // csrmv_avx512.cpp:86-100
    else
        return ((l[0] + l[4]) + (l[2] + l[6])) + ((l[1] + l[5]) + (l[3] + l[7])); // csrmv_kr.hpp:788-806
}

// cell (p, lane) of a slice that starts at o0: PACK 1 -> o0 + 64 p + lane; PACK 4 -> four consecutive cells of a
// row are adjacent: o0 + 256 (p / 4) + 4 lane + p % 4 (slice width is a multiple of 4 there)
template <int PACK>
__device__ __forceinline__ long long cell_of(int p, int lane)
{
    return PACK == 1 ? (long long)p * 64 + lane : (long long)(p >> 2) * 256 + lane * 4 + (p & 3);
}

template <typename T, int PACK>
__global__ __launch_bounds__(256) void sell_fill_kernel(aoclsparse_int m, int base,
                                                        const aoclsparse_int *__restrict__ row_ptr,
                                                        const aoclsparse_int *__restrict__ col,
                                                        const T *__restrict__ val, aoclsparse_int nslices,
                                                        const long long *__restrict__ slice_ptr,
                                                        T *__restrict__ sval, aoclsparse_int *__restrict__ scol,
                                                        aoclsparse_int *__restrict__ rowlen)
{
    const int s    = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if(s >= nslices)
        return;
    const int       i  = s * 64 + lane;
    const long long o0 = slice_ptr[s];
    const int       w  = (int)((slice_ptr[s + 1] - o0) >> 6);
    int             b = 0, len = 0;
    if(i < m)
    {
        b   = row_ptr[i] - base;
        len = row_ptr[i + 1] - base - b;
        rowlen[i] = len;
    }
    for(int p = 0; p < w; p++)
    {
        const long long o  = o0 + cell_of<PACK>(p, lane);
        const bool      in = p < len;
        sval[o]            = in ? val[b + p] : T(0);
        scol[o]            = in ? col[b + p] - base : -1;
    }
}

// ---- shared column lists (SELL-64 with one column list per run of rows that repeat it) --------------------------------
// Two kinds of repetition, found the same way: the rows of a mesh node (several dofs) carry the SAME column list, and the
// rows of a stencil carry the list of the row before SHIFTED BY ONE (row i of a 5-point Laplacian: i-g, i-1, i, i+1, i+g).
// A slice stores its columns once per "leader" (lane 0, and every lane whose list is neither the previous lane's nor the
// previous lane's plus one): cell (p, leader k) of slice s at cptr[s] + nl_s p + k (PACK 4: cptr[s] + 4 nl_s (p / 4) + 4 k
// + p % 4).  follow[i] (16 bits per row) = leader index inside the slice | shift << 8, where shift = how many of the rows
// between the leader and row i were "plus one" steps: a lane's column is its leader's + shift.  Values stay where they are.
// The column stream shrinks from 4 B per cell to 4 B / (rows per list): 12 -> 8.8 B per cell for 5-dof nodes, 12 -> ~8.1 B for
// the Laplacian (one list per 64 rows, broken at the grid edges).
// One wavefront per slice: each lane compares its row with its predecessor, indices and shifts by ballot + popcount.
constexpr int       SELL_CPTR_MODE_SHIFT = 56;
constexpr long long SELL_CPTR_MASK       = (1LL << SELL_CPTR_MODE_SHIFT) - 1;

__global__ __launch_bounds__(256) void sell_leaders_kernel(aoclsparse_int m, int base, const aoclsparse_int *__restrict__ row_ptr,
                                                           const aoclsparse_int *__restrict__ col, aoclsparse_int nslices,
                                                           unsigned short *__restrict__ follow, aoclsparse_int *__restrict__ nl)
{
    const int s    = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if(s >= nslices)
        return;
    const int i      = s * 64 + lane;
    bool      leader = false, plus1 = false;
    if(i < m)
    {
        leader = lane == 0;
        if(!leader)
        {
            const int b = row_ptr[i] - base, len = row_ptr[i + 1] - base - b, bp = row_ptr[i - 1] - base;
            bool      same = len == b - bp, shifted = same && len > 0; // (column VALUES: only differences are used)
            for(int k = 0; k < len && (same || shifted); k++)
            {
                const int dcol = col[b + k] - col[bp + k];
                same           = same && dcol == 0;
                shifted        = shifted && dcol == 1;
            }
            leader = !same && !shifted;
            plus1  = shifted;
        }
    }
    const unsigned long long upto = (2ull << lane) - 1ull; // lanes 0 .. lane
    const unsigned long long bal  = __builtin_amdgcn_ballot_w64(leader);
    const unsigned long long p1   = __builtin_amdgcn_ballot_w64(plus1);
    if(i < m)
    {
        const unsigned long long mine = bal & upto; // never 0: lane 0 is a leader
        const int                ll   = 63 - __builtin_clzll(mine); // my leader's lane
        const unsigned long long span = upto & ~((2ull << ll) - 1ull); // lanes ll + 1 .. lane
        follow[i] = (unsigned short)((__builtin_popcountll(mine) - 1) | (__builtin_popcountll(p1 & span) << 8));
    }
    // a FULL slice with one leader whose followers are all "plus one" (the interior of a stencil) or all "same" needs no
    // follow[] at run time: mode 1 -> shift = lane, mode 2 -> shift = 0 (bits 8.. of nl[s]; the host moves them into cptr)
    if(lane == 0)
    {
        const bool full = s * 64 + 63 < m;
        const int  mode = (full && bal == 1ull) ? (p1 == ~1ull ? 1 : (p1 == 0ull ? 2 : 0)) : 0;
        nl[s]           = (aoclsparse_int)__builtin_popcountll(bal) | (mode << 8);
    }
}

template <typename T, int PACK>
__global__ __launch_bounds__(256) void sell_fill_shared_kernel(aoclsparse_int m, int base,
                                                               const aoclsparse_int *__restrict__ row_ptr,
                                                               const aoclsparse_int *__restrict__ col,
                                                               const T *__restrict__ val, aoclsparse_int nslices,
                                                               const long long *__restrict__ slice_ptr,
                                                               const long long *__restrict__ cptr,
                                                               const unsigned short *__restrict__ follow, T *__restrict__ sval,
                                                               aoclsparse_int *__restrict__ scol,
                                                               aoclsparse_int *__restrict__ rowlen)
{
    const int s    = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if(s >= nslices)
        return;
    const int       i  = s * 64 + lane;
    const long long o0 = slice_ptr[s], c0 = cptr[s] & SELL_CPTR_MASK;
    const int       w  = (int)((slice_ptr[s + 1] - o0) >> 6);
    const int       nl = w > 0 ? (int)(((cptr[s + 1] & SELL_CPTR_MASK) - c0) / w) : 0;
    int             b = 0, len = 0, k = 0;
    bool            leader = false;
    if(i < m)
    {
        b   = row_ptr[i] - base;
        len = row_ptr[i + 1] - base - b;
        rowlen[i] = len;
        k         = follow[i] & 0xff;
        leader    = lane == 0 || (follow[i - 1] & 0xff) != k;
    }
    for(int p = 0; p < w; p++)
    {
        const bool in = p < len;
        sval[o0 + cell_of<PACK>(p, lane)] = in ? val[b + p] : T(0);
        if(leader)
            scol[c0 + (PACK == 1 ? (long long)p * nl + k : (long long)(p >> 2) * 4 * nl + 4 * k + (p & 3))] = in ? col[b + p] - base : -1;
    }
}

// four adjacent cells of one lane as vector loads (PACK 4): 32 B of values (16 B for float), 16 B of columns
__device__ __forceinline__ void load4(const double *p, double (&o)[4])
{
    const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
    o[0] = a.x, o[1] = a.y, o[2] = b.x, o[3] = b.y;
}
__device__ __forceinline__ void load4(const float *p, float (&o)[4])
{
    const float4 a = *reinterpret_cast<const float4 *>(p);
    o[0] = a.x, o[1] = a.y, o[2] = a.z, o[3] = a.w;
}
__device__ __forceinline__ void load4(const aoclsparse_int *p, int (&o)[4])
{
    const int4 a = *reinterpret_cast<const int4 *>(p);
    o[0] = a.x, o[1] = a.y, o[2] = a.z, o[3] = a.w;
}

// loads the G cells p0 .. p0+G-1 of this lane (wave-uniform guards against the slice width w)
// cs = lanes per column row: 64, or the slice's number of leaders when the column lists are shared
template <typename T, int PACK, int G>
__device__ __forceinline__ void load_step(const T *v, const aoclsparse_int *c, int p0, int w, T (&vv)[G], int (&cc)[G],
                                          int cs = 64)
{
    if constexpr(PACK == 1)
    {
#pragma unroll
        for(int q = 0; q < G; q++)
        {
            const bool ok = p0 + q < w;
            vv[q]         = ok ? v[(p0 + q) * 64] : T(0);
            cc[q]         = ok ? c[(p0 + q) * cs] : -1;
        }
    }
    else
    {
#pragma unroll
        for(int k = 0; k < G / 4; k++)
        {
            T   tv[4] = {T(0), T(0), T(0), T(0)};
            int tc[4] = {-1, -1, -1, -1};
            if(p0 + 4 * k < w) // w is a multiple of 4: the pack is whole or absent
            {
                const long long o = (long long)((p0 >> 2) + k) * 256;
                load4(v + o, tv);
                load4(c + (long long)((p0 >> 2) + k) * (4 * cs), tc);
            }
#pragma unroll
            for(int q = 0; q < 4; q++)
                vv[4 * k + q] = tv[q], cc[4 * k + q] = tc[q];
        }
    }
}

// WAVES slices per workgroup (1 for small matrices so that every slice gets its own CU)
template <typename T, int ORDER, int WAVES, int PACK, bool SHARED = false, bool CONJ = false>
__global__ __launch_bounds__(64 * WAVES) void sell_mv_kernel(aoclsparse_int m, aoclsparse_int nslices,
                                                             const long long *__restrict__ slice_ptr,
                                                             const T *__restrict__ sval,
                                                             const aoclsparse_int *__restrict__ scol,
                                                             const aoclsparse_int *__restrict__ rowlen, T alpha,
                                                             const T *__restrict__ x, T beta, T *__restrict__ y,
                                                             bool nt, const long long *__restrict__ cptr = nullptr,
                                                             const unsigned short *__restrict__ follow = nullptr, int rev = 0)
{
    // rev: the slices in descending order.  Consecutive products of a handle ALTERNATE the direction (SellPlan::products): the
    // end of the matrix, which the previous product left in the 256 MB Infinity Cache, is where this one starts -- round 5,
    // profiles/r5/sell_placement.txt: shell-like 90-101 -> 81-87 us, the headline 0.178 -> 0.165 ms.  Same slices, same bits.
    const unsigned bx = rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    const int s    = __builtin_amdgcn_readfirstlane((int)(bx * WAVES + (threadIdx.x >> 6)));
    const int lane = threadIdx.x & 63;
    if(s >= nslices)
        return;
    const long long       o0 = slice_ptr[s];
    const int             w  = (int)((slice_ptr[s + 1] - o0) >> 6);
    const T              *v  = sval + o0 + lane * PACK;
    const int             i  = s * 64 + lane;
    const aoclsparse_int *c  = scol + o0 + lane * PACK;
    int                   cs = 64, dl = 0;
    if constexpr(SHARED)
    {
        // one column list per leader: cs leaders in this slice; this lane reads its leader's and adds its shift
        // (cptr's top bits carry the slice's mode: 1 = one leader, shift = lane; 2 = one leader, no shift; else follow[])
        const long long cw   = cptr[s];
        const long long c0   = cw & SELL_CPTR_MASK;
        const int       mode = (int)(cw >> SELL_CPTR_MODE_SHIFT);
        int             f    = 0;
        if(mode == 0)
            f = i < m ? follow[i] : 0;
        else if(mode == 1)
            f = lane << 8;
        cs = w > 0 ? (int)(((cptr[s + 1] & SELL_CPTR_MASK) - c0) / w) : 1;
        c  = scol + c0 + (f & 0xff) * PACK;
        dl = f >> 8;
    }
    T r = T(0);
    if constexpr(ORDER == 0 && PACK == 1)
    {
        // short rows (this is the layout of matrices with < 16 non-zeros per row): four independent line loads
        // per step, then the gathers, then the chain.  Measured on the 4096^2 Laplacian (w = 5): 0.218 ms;
        // 8-wide, guard-predicated or software-pipelined steps 0.223-0.26 ms (profiles/r1)
        // Widths up to 8 (round 3): ONE batch of exactly w line loads, then the w gathers, then the chain -- the wave's life is
        // three dependent round trips (slice words -> value / column lines -> x) whatever w is.  The 4-step loop below spent
        // five on a 5-wide slice (4 + 1 entries: lines, gathers, lines, gathers), and the kernel is bound by wave lifetime x
        // occupancy, not by bytes in flight (same box: 0.1845-0.186 -> 0.1786-0.1793 ms, profiles/r3/sell_width_switch.txt).
        int p = 0;
        auto batch = [&](auto wtag) {
            constexpr int W = decltype(wtag)::value;
            T             vv[W], xx[W];
            int           cc[W];
#pragma unroll
            for(int q = 0; q < W; q++)
                vv[q] = s_cj<CONJ>(v[q * 64]), cc[q] = c[q * cs];
#pragma unroll
            for(int q = 0; q < W; q++)
                xx[q] = x[cc[q] >= 0 ? cc[q] + dl : 0];
#pragma unroll
            for(int q = 0; q < W; q++)
                r = cc[q] >= 0 ? s_fma(vv[q], xx[q], r) : r;
            p = W;
        };
        switch(w) // wave-uniform
        {
        case 5: batch(std::integral_constant<int, 5>{}); break;
        case 6: batch(std::integral_constant<int, 6>{}); break;
        case 7: batch(std::integral_constant<int, 7>{}); break;
        case 8: batch(std::integral_constant<int, 8>{}); break;
        default: break;
        }
        for(; p + 4 <= w; p += 4)
        {
            const T   v0 = s_cj<CONJ>(v[(p + 0) * 64]), v1 = s_cj<CONJ>(v[(p + 1) * 64]), v2 = s_cj<CONJ>(v[(p + 2) * 64]),
                      v3 = s_cj<CONJ>(v[(p + 3) * 64]);
            const int c0 = c[(p + 0) * cs], c1 = c[(p + 1) * cs], c2 = c[(p + 2) * cs], c3 = c[(p + 3) * cs];
            // (a padding cell, -1, is never used, but its gather must stay inside x: index 0)
            const T   x0 = x[c0 >= 0 ? c0 + dl : 0], x1 = x[c1 >= 0 ? c1 + dl : 0], x2 = x[c2 >= 0 ? c2 + dl : 0],
                      x3 = x[c3 >= 0 ? c3 + dl : 0];
            r = c0 >= 0 ? s_fma(v0, x0, r) : r;
            r = c1 >= 0 ? s_fma(v1, x1, r) : r;
            r = c2 >= 0 ? s_fma(v2, x2, r) : r;
            r = c3 >= 0 ? s_fma(v3, x3, r) : r;
        }
        for(; p < w; p++)
        {
            const T   v0 = s_cj<CONJ>(v[p * 64]);
            const int c0 = c[p * cs];
            const T   x0 = x[c0 >= 0 ? c0 + dl : 0];
            r = c0 >= 0 ? s_fma(v0, x0, r) : r;
        }
    }
    else
    {
        // Steps of G cells, software-pipelined: the lines of step k+1 are issued BEFORE the x gathers of step k,
        // so a step costs one memory round trip instead of two (vmcnt retires in order: the gathers wait for
        // the younger-issued lines too, but everything is in flight together).
        constexpr int G    = ORDER == 2 ? 8 : 4;
        const int     len  = (ORDER != 0 && i < m) ? rowlen[i] : 0;
        const int     full = len & ~(G - 1);
        T             l[G];
#pragma unroll
        for(int q = 0; q < G; q++)
            l[q] = T(0);
        bool reduced = false;
        T    vn[G];
        int  cn[G];
        load_step<T, PACK, G>(v, c, 0, w, vn, cn, cs);
        for(int p0 = 0; p0 < w; p0 += G)
        {
            T   vv[G], xx[G];
            int cc[G];
#pragma unroll
            for(int q = 0; q < G; q++)
                vv[q] = vn[q], cc[q] = cn[q];
            if(p0 + G < w)
                load_step<T, PACK, G>(v, c, p0 + G, w, vn, cn, cs);
#pragma unroll
            for(int q = 0; q < G; q++)
                xx[q] = x[cc[q] >= 0 ? cc[q] + dl : 0]; // (a padding cell, -1, is never used; its gather stays inside x)
            if constexpr(ORDER == 0)
            {
#pragma unroll
                for(int q = 0; q < G; q++)
                    r = cc[q] >= 0 ? s_fma(vv[q], xx[q], r) : r;
            }
            else if(p0 < full)
            {
#pragma unroll
                for(int q = 0; q < G; q++)
                    l[q] = s_fma(vv[q], xx[q], l[q]);
            }
            else if(p0 < len)
            {
                if(!reduced)
                    r = lanes_sum<T, G>(l), reduced = true;
#pragma unroll
                for(int q = 0; q < G; q++)
                    if(p0 + q < len)
                        r = s_fma(vv[q], xx[q], r);
            }
        }
        if(ORDER != 0 && !reduced)
            r = lanes_sum<T, G>(l);
    }
    if(i < m)
        s_store(y + i, s_finish(r, alpha, beta, y + i), nt);
}

// Short rows (round 3): matrices whose widest slice has WMAX <= 8 cells, scalar summation order, PACK 1, large launches.  ONE
// batch of WMAX value / column line loads (index clamped to the slice's own width: no guard inside the batch; a narrower
// boundary slice re-reads its last line and skips the FMA), then the WMAX gathers, then the chain: three dependent round trips
// per slice, four slices per workgroup.  Same-box sweep on the headline workload (tools/history/exp_r3_short.sh,
// profiles/r3/sell_width_switch.txt): general kernel before the width switch 0.1845-0.186 ms, with it 0.1786-0.1793, this
// kernel with 1 / 2 / 4 slices per workgroup 0.180-0.181 / 0.180-0.181 / 0.1773-0.1779; TWO or more slices per WAVEFRONT
// (walked together, twice the bytes in flight per wave) 0.183-0.236 ms -- more registers, fewer waves, no gain.
template <typename T, int WMAX, int WAVES, bool SHARED, int SPW = 1, bool CONJ = false>
__global__ __launch_bounds__(64 * WAVES) void sell_mv_short_kernel(aoclsparse_int m, aoclsparse_int nslices,
                                                                   const long long *__restrict__ slice_ptr,
                                                                   const T *__restrict__ sval,
                                                                   const aoclsparse_int *__restrict__ scol, T alpha,
                                                                   const T *__restrict__ x, T beta, T *__restrict__ y, bool nt,
                                                                   const long long *__restrict__ cptr,
                                                                   const unsigned short *__restrict__ follow, int rev = 0)
{
    // SPW slices per wavefront, walked TOGETHER (all value / column loads of the SPW slices, then all gathers, then the chains):
    // SPW times the bytes in flight per wavefront.  Lost for double (two: 0.183-0.236 vs 0.177 ms, round 3); float moves half the
    // bytes per load instruction, and large float launches run four (round 4: sell_launch_short).
    const unsigned bx = rev ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    const int sb   = __builtin_amdgcn_readfirstlane((int)(bx * WAVES + (threadIdx.x >> 6)) * SPW);
    const int lane = threadIdx.x & 63;
    if(sb >= nslices)
        return;
    T   vv[SPW][WMAX], xx[SPW][WMAX];
    int cc[SPW][WMAX], w[SPW], dl[SPW];
#pragma unroll
    for(int u = 0; u < SPW; u++)
    {
        const int             s  = min(sb + u, (int)nslices - 1); // (beyond the last slice: it is walked again, nothing is stored)
        const long long       o0 = slice_ptr[s];
        const int             i  = s * 64 + lane;
        const T              *v  = sval + o0 + lane;
        const aoclsparse_int *c  = scol + o0 + lane;
        int                   cs = 64;
        w[u]                     = (int)((slice_ptr[s + 1] - o0) >> 6);
        dl[u]                    = 0;
        if constexpr(SHARED)
        {
            const long long cw   = cptr[s];
            const long long c0   = cw & SELL_CPTR_MASK;
            const int       mode = (int)(cw >> SELL_CPTR_MODE_SHIFT);
            int             f    = 0;
            if(mode == 0)
                f = i < m ? follow[i] : 0;
            else if(mode == 1)
                f = lane << 8;
            cs    = w[u] > 0 ? (int)(((cptr[s + 1] & SELL_CPTR_MASK) - c0) / w[u]) : 1;
            c     = scol + c0 + (f & 0xff);
            dl[u] = f >> 8;
        }
        if(w[u] > 0) // (an empty slice has no cell to read)
        {
#pragma unroll
            for(int q = 0; q < WMAX; q++)
            {
                const int qq = min(q, w[u] - 1); // wave-uniform
                vv[u][q]     = s_cj<CONJ>(v[qq * 64]);
                cc[u][q]     = c[qq * cs];
            }
        }
        else
        {
#pragma unroll
            for(int q = 0; q < WMAX; q++)
                vv[u][q] = T(0), cc[u][q] = -1;
        }
    }
#pragma unroll
    for(int u = 0; u < SPW; u++)
#pragma unroll
        for(int q = 0; q < WMAX; q++)
            xx[u][q] = x[cc[u][q] >= 0 ? cc[u][q] + dl[u] : 0];
#pragma unroll
    for(int u = 0; u < SPW; u++)
    {
        T r = T(0);
#pragma unroll
        for(int q = 0; q < WMAX; q++)
            r = (q < w[u] && cc[u][q] >= 0) ? s_fma(vv[u][q], xx[u][q], r) : r;
        const int i = (sb + u) * 64 + lane;
        if(sb + u < nslices && i < m)
            s_store(y + i, s_finish(r, alpha, beta, y + i), nt);
    }
}

constexpr aoclsparse_int SELL_SHORT_SPW4_SLICES = 100000;

template <typename T, bool SHARED>
bool sell_launch_short(hipStream_t s, int wmax, aoclsparse_int m, aoclsparse_int nslices, const long long *slice_ptr, const T *sval,
                       const aoclsparse_int *scol, T alpha, const T *x, T beta, T *y, bool nt, const long long *cptr,
                       const unsigned short *lead, int rev = 0)
{
    constexpr int WAVES = 4;
    // float, >= 100,000 slices: four slices per wavefront (a float load instruction moves half the bytes of a double one; same box,
    // tools/history/exp_float_headline.py, 1 / 4 / 8 slices per wavefront: 4096^2 0.1013 / 0.0949 / 0.1218 ms, 3000^2 0.0512 / 0.0479 /
    // 0.0543, 2000^2 0.0229 / 0.0235 / 0.0267 -- and no change for double, which stays at one: profiles/r4/float_headline.txt)
    const bool      four   = sizeof(T) == 4 && nslices >= SELL_SHORT_SPW4_SLICES;
    const long long per_wg = (long long)WAVES * (four ? 4 : 1);
    const dim3      grid((unsigned)((nslices + per_wg - 1) / per_wg)), block(64 * WAVES);
#define MI355_SHORT(W)                                                                                                    \
    case W:                                                                                                               \
        if(four)                                                                                                          \
            hipLaunchKernelGGL((sell_mv_short_kernel<T, W, WAVES, SHARED, 4>), grid, block, 0, s, m, nslices, slice_ptr, sval, \
                               scol, alpha, x, beta, y, nt, cptr, lead, rev);                                             \
        else                                                                                                              \
            hipLaunchKernelGGL((sell_mv_short_kernel<T, W, WAVES, SHARED>), grid, block, 0, s, m, nslices, slice_ptr, sval, \
                               scol, alpha, x, beta, y, nt, cptr, lead, rev);                                             \
        return true
    switch(wmax)
    {
        MI355_SHORT(1);
        MI355_SHORT(2);
        MI355_SHORT(3);
        MI355_SHORT(4);
        MI355_SHORT(5);
        MI355_SHORT(6);
        MI355_SHORT(7);
        MI355_SHORT(8);
    default: return false;
    }
#undef MI355_SHORT
}

template <typename T, int ORDER, int PACK>
void sell_launch(hipStream_t s, aoclsparse_int m, aoclsparse_int nslices, const long long *slice_ptr, const T *sval,
                 const aoclsparse_int *scol, const aoclsparse_int *rowlen, T alpha, const T *x, T beta, T *y,
                 const long long *cptr, const unsigned short *lead, int rev = 0)
{
    // one slice per workgroup while the launch is small (every slice its own CU), two otherwise
    // (swept on the headline workload: 1 / 2 / 4 / 8 slices per workgroup = 0.221 / 0.218 / 0.221 / 0.222 ms)
    const bool nt = (size_t)m * sizeof(T) > ((size_t)32 << 20);
    if(cptr)
    {
        if(nslices < 2048)
            hipLaunchKernelGGL((sell_mv_kernel<T, ORDER, 1, PACK, true>), dim3(nslices), dim3(64), 0, s, m, nslices, slice_ptr,
                               sval, scol, rowlen, alpha, x, beta, y, nt, cptr, lead, rev);
        else
            hipLaunchKernelGGL((sell_mv_kernel<T, ORDER, 2, PACK, true>), dim3((nslices + 1) / 2), dim3(128), 0, s, m, nslices,
                               slice_ptr, sval, scol, rowlen, alpha, x, beta, y, nt, cptr, lead, rev);
    }
    else if(nslices < 2048)
        hipLaunchKernelGGL((sell_mv_kernel<T, ORDER, 1, PACK>), dim3(nslices), dim3(64), 0, s, m, nslices, slice_ptr,
                           sval, scol, rowlen, alpha, x, beta, y, nt, (const long long *)nullptr, (const unsigned short *)nullptr, rev);
    else
        hipLaunchKernelGGL((sell_mv_kernel<T, ORDER, 2, PACK>), dim3((nslices + 1) / 2), dim3(128), 0, s, m, nslices,
                           slice_ptr, sval, scol, rowlen, alpha, x, beta, y, nt, (const long long *)nullptr,
                           (const unsigned short *)nullptr, rev);
}

} // namespace

template <typename T>
aoclsparse_status launch_sell_fill(hipStream_t s, int pack, aoclsparse_int m, int base, const aoclsparse_int *row_ptr,
                                   const aoclsparse_int *col, const T *val, aoclsparse_int nslices,
                                   const long long *slice_ptr, T *sval, aoclsparse_int *scol, aoclsparse_int *rowlen,
                                   const long long *cptr, const unsigned short *lead)
{
    if(nslices <= 0)
        return aoclsparse_status_success;
    if(cptr)
    {
        if(pack == 4)
            hipLaunchKernelGGL((sell_fill_shared_kernel<T, 4>), dim3((nslices + 3) / 4), dim3(256), 0, s, m, base, row_ptr,
                               col, val, nslices, slice_ptr, cptr, lead, sval, scol, rowlen);
        else
            hipLaunchKernelGGL((sell_fill_shared_kernel<T, 1>), dim3((nslices + 3) / 4), dim3(256), 0, s, m, base, row_ptr,
                               col, val, nslices, slice_ptr, cptr, lead, sval, scol, rowlen);
    }
    else if(pack == 4)
        hipLaunchKernelGGL((sell_fill_kernel<T, 4>), dim3((nslices + 3) / 4), dim3(256), 0, s, m, base, row_ptr, col,
                           val, nslices, slice_ptr, sval, scol, rowlen);
    else
        hipLaunchKernelGGL((sell_fill_kernel<T, 1>), dim3((nslices + 3) / 4), dim3(256), 0, s, m, base, row_ptr, col,
                           val, nslices, slice_ptr, sval, scol, rowlen);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_sellmv(hipStream_t s, int order, int pack, T alpha, aoclsparse_int m, aoclsparse_int nslices,
                                const long long *slice_ptr, const T *sval, const aoclsparse_int *scol,
                                const aoclsparse_int *rowlen, const T *x, T beta, T *y, const long long *cptr,
                                const unsigned short *lead, aoclsparse_int max_width, int rev)
{
    if(m <= 0 || nslices <= 0)
        return aoclsparse_status_success;
    if(order < 0 || order > 2 || (pack != 1 && pack != 4))
        return aoclsparse_status_invalid_kid;
    // widest slice <= 8 cells, scalar order, a launch large enough that four slices per workgroup still spread over every CU
    if(order == 0 && pack == 1 && max_width >= 1 && max_width <= 8 && nslices >= 4096)
    {
        const bool nt   = (size_t)m * sizeof(T) > ((size_t)32 << 20);
        const bool done = cptr ? sell_launch_short<T, true>(s, (int)max_width, m, nslices, slice_ptr, sval, scol, alpha, x, beta, y, nt, cptr, lead, rev)
                               : sell_launch_short<T, false>(s, (int)max_width, m, nslices, slice_ptr, sval, scol, alpha, x, beta, y, nt,
                                                             nullptr, nullptr, rev);
        if(done)
        {
            MI355_HIP_TRY(hipGetLastError());
            return aoclsparse_status_success;
        }
    }
#define SELL_CASE(O, P)                                                                        \
    sell_launch<T, O, P>(s, m, nslices, slice_ptr, sval, scol, rowlen, alpha, x, beta, y, cptr, lead, rev); \
    break
    switch(order * 2 + (pack == 4 ? 1 : 0))
    {
    case 0:
        SELL_CASE(0, 1);
    case 1:
        SELL_CASE(0, 4);
    case 2:
        SELL_CASE(1, 1);
    case 3:
        SELL_CASE(1, 4);
    case 4:
        SELL_CASE(2, 1);
    case 5:
        SELL_CASE(2, 4);
    }
#undef SELL_CASE
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// aoclsparse_{c,z}mv on the SELL-64 copy (PACK 1, the scalar chain per row): the short-row kernel for large launches whose
// widest slice has <= 8 cells, the general kernel otherwise
template <typename R>
aoclsparse_status launch_sellmv_complex(hipStream_t s, bool conj, cplx<R> alpha, aoclsparse_int m, aoclsparse_int nslices,
                                        const long long *slice_ptr, const cplx<R> *sval, const aoclsparse_int *scol,
                                        const aoclsparse_int *rowlen, const cplx<R> *x, cplx<R> beta, cplx<R> *y,
                                        const long long *cptr, const unsigned short *lead, aoclsparse_int max_width, int rev)
{
    using C = cplx<R>;
    if(m <= 0 || nslices <= 0)
        return aoclsparse_status_success;
    const bool nt = (size_t)m * sizeof(C) > ((size_t)32 << 20);
    auto       go = [&](auto shared_tag, auto conj_tag) {
        constexpr bool SH = decltype(shared_tag)::value, CJ = decltype(conj_tag)::value;
        const long long      *cp = SH ? cptr : nullptr;
        const unsigned short *ld = SH ? lead : nullptr;
        if(max_width >= 1 && max_width <= 8 && nslices >= 4096)
        {
            constexpr int WAVES = 4;
            const dim3    grid((unsigned)((nslices + WAVES - 1) / WAVES)), block(64 * WAVES);
#define MI355_CSHORT(W)                                                                                                          \
    case W:                                                                                                                      \
        hipLaunchKernelGGL((sell_mv_short_kernel<C, W, WAVES, SH, 1, CJ>), grid, block, 0, s, m, nslices, slice_ptr, sval, scol, \
                           alpha, x, beta, y, nt, cp, ld, rev);                                                                  \
        return
            switch((int)max_width)
            {
                MI355_CSHORT(1);
                MI355_CSHORT(2);
                MI355_CSHORT(3);
                MI355_CSHORT(4);
                MI355_CSHORT(5);
                MI355_CSHORT(6);
                MI355_CSHORT(7);
                MI355_CSHORT(8);
            }
#undef MI355_CSHORT
        }
        if(nslices < 2048)
            hipLaunchKernelGGL((sell_mv_kernel<C, 0, 1, 1, SH, CJ>), dim3(nslices), dim3(64), 0, s, m, nslices, slice_ptr, sval, scol,
                               rowlen, alpha, x, beta, y, nt, cp, ld, rev);
        else
            hipLaunchKernelGGL((sell_mv_kernel<C, 0, 2, 1, SH, CJ>), dim3((nslices + 1) / 2), dim3(128), 0, s, m, nslices, slice_ptr,
                               sval, scol, rowlen, alpha, x, beta, y, nt, cp, ld, rev);
    };
    if(cptr)
    {
        if(conj)
            go(std::true_type{}, std::true_type{});
        else
            go(std::true_type{}, std::false_type{});
    }
    else if(conj)
        go(std::false_type{}, std::true_type{});
    else
        go(std::false_type{}, std::false_type{});
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template aoclsparse_status launch_sellmv_complex<double>(hipStream_t, bool, cdouble, aoclsparse_int, aoclsparse_int, const long long *,
                                                         const cdouble *, const aoclsparse_int *, const aoclsparse_int *,
                                                         const cdouble *, cdouble, cdouble *, const long long *,
                                                         const unsigned short *, aoclsparse_int, int);
template aoclsparse_status launch_sellmv_complex<float>(hipStream_t, bool, cfloat, aoclsparse_int, aoclsparse_int, const long long *,
                                                        const cfloat *, const aoclsparse_int *, const aoclsparse_int *, const cfloat *,
                                                        cfloat, cfloat *, const long long *, const unsigned short *, aoclsparse_int, int);
// (the fill kernels only move values: cfloat cells are filled as 8-byte doubles, cdouble cells need their own instantiation)
template aoclsparse_status launch_sell_fill<cdouble>(hipStream_t, int, aoclsparse_int, int, const aoclsparse_int *, const aoclsparse_int *,
                                                     const cdouble *, aoclsparse_int, const long long *, cdouble *, aoclsparse_int *,
                                                     aoclsparse_int *, const long long *, const unsigned short *);

aoclsparse_status launch_sell_leaders(hipStream_t s, aoclsparse_int m, int base, const aoclsparse_int *row_ptr, const aoclsparse_int *col,
                                      aoclsparse_int nslices, unsigned short *lead, aoclsparse_int *nl)
{
    if(nslices <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL(sell_leaders_kernel, dim3((nslices + 3) / 4), dim3(256), 0, s, m, base, row_ptr, col, nslices, lead, nl);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_SELL_INSTANTIATE(T)                                                                                     \
    template aoclsparse_status launch_sell_fill<T>(hipStream_t, int, aoclsparse_int, int, const aoclsparse_int *,     \
                                                   const aoclsparse_int *, const T *, aoclsparse_int,                 \
                                                   const long long *, T *, aoclsparse_int *, aoclsparse_int *,        \
                                                   const long long *, const unsigned short *);                         \
    template aoclsparse_status launch_sellmv<T>(hipStream_t, int, int, T, aoclsparse_int, aoclsparse_int,             \
                                                const long long *, const T *, const aoclsparse_int *,                 \
                                                const aoclsparse_int *, const T *, T, T *, const long long *,          \
                                                const unsigned short *, aoclsparse_int, int);
MI355_SELL_INSTANTIATE(double)
MI355_SELL_INSTANTIATE(float)

} // namespace mi355
