// sell_kernels.hip -- SpMV on SELL-64, the layout aoclsparse_optimize builds for an mv hint on MI355X.
//
// The reference's optimize step re-stores a hinted matrix in the format its CPU kernels like best
// (br4 / ELLT-HYB / blocked CSR, analysis.cpp:146-382).  The GPU analogue is sliced ELL with one slice
// per 64-wide wavefront: slice s holds rows [64 s, 64 s + 64) column-major, cell (p, lane) at
// slice_ptr[s] + 64 p + lane, padded to the slice's longest row (column -1, value 0).  Lane i of a wave
// owns row i:
//   * every val / col access is one coalesced line per wavefront instruction, no row_ptr, no LDS;
//   * a lane walks its row front to back, so the reference's summation orders are reproduced exactly:
//     order 0 is the scalar FMA chain (csrmv_kr.hpp:448-513); orders 1 / 2 keep 4 / 8 partial sums per
//     lane for the full groups, reduce them as the AVX2 / AVX-512 kernels do, then run the scalar tail
//     (csrmv_kr.hpp:949-1040, csrmv_avx512.cpp:36-134, csrmv_kr.hpp:734-831 for float).
// Bytes per launch: cells*(8+4) + 8 m (y) (+ 8 m if beta != 0) (+ 4 m row lengths for orders 1 / 2) + x
// gathers; cells <= 1.15 nnz or the handle stays on the CSR-Adaptive kernel (matrix.cpp: build_sell).
#include "internal.hpp"

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

template <typename T>
__device__ __forceinline__ void s_store(T *p, T v, bool nt)
{
    if(nt)
        __builtin_nontemporal_store(v, p);
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

template <typename T>
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
        const long long o = o0 + (long long)p * 64 + lane;
        const bool      in = p < len;
        sval[o]            = in ? val[b + p] : T(0);
        scol[o]            = in ? col[b + p] - base : -1;
    }
}

// WAVES slices per workgroup (1 for small matrices so that every slice gets its own CU)
template <typename T, int ORDER, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void sell_mv_kernel(aoclsparse_int m, aoclsparse_int nslices,
                                                             const long long *__restrict__ slice_ptr,
                                                             const T *__restrict__ sval,
                                                             const aoclsparse_int *__restrict__ scol,
                                                             const aoclsparse_int *__restrict__ rowlen, T alpha,
                                                             const T *__restrict__ x, T beta, T *__restrict__ y,
                                                             bool nt)
{
    const int s    = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * WAVES + (threadIdx.x >> 6)));
    const int lane = threadIdx.x & 63;
    if(s >= nslices)
        return;
    const long long       o0 = slice_ptr[s];
    const int             w  = (int)((slice_ptr[s + 1] - o0) >> 6);
    const T              *v  = sval + o0 + lane;
    const aoclsparse_int *c  = scol + o0 + lane;
    const int             i  = s * 64 + lane;
    T                     r  = T(0);
    if constexpr(ORDER == 0)
    {
        // four independent line loads per step, then the gathers, then the chain.  Measured on the 4096^2
        // Laplacian (w = 5): 0.218 ms; 8-wide or wave-uniform guarded steps 0.223-0.26 ms (profiles/r1)
        int p = 0;
        for(; p + 4 <= w; p += 4)
        {
            const T   v0 = v[(p + 0) * 64], v1 = v[(p + 1) * 64], v2 = v[(p + 2) * 64], v3 = v[(p + 3) * 64];
            const int c0 = c[(p + 0) * 64], c1 = c[(p + 1) * 64], c2 = c[(p + 2) * 64], c3 = c[(p + 3) * 64];
            const T   x0 = x[max(c0, 0)], x1 = x[max(c1, 0)], x2 = x[max(c2, 0)], x3 = x[max(c3, 0)];
            r = c0 >= 0 ? s_fma(v0, x0, r) : r;
            r = c1 >= 0 ? s_fma(v1, x1, r) : r;
            r = c2 >= 0 ? s_fma(v2, x2, r) : r;
            r = c3 >= 0 ? s_fma(v3, x3, r) : r;
        }
        for(; p < w; p++)
        {
            const T   v0 = v[p * 64];
            const int c0 = c[p * 64];
            const T   x0 = x[max(c0, 0)];
            r = c0 >= 0 ? s_fma(v0, x0, r) : r;
        }
    }
    else
    {
        constexpr int G    = ORDER == 1 ? 4 : 8;
        const int     len  = i < m ? rowlen[i] : 0;
        const int     full = len & ~(G - 1);
        T             l[G];
#pragma unroll
        for(int q = 0; q < G; q++)
            l[q] = T(0);
        bool reduced = false;
        for(int p0 = 0; p0 < w; p0 += G)
        {
            T   vv[G], xx[G];
            int cc[G];
#pragma unroll
            for(int q = 0; q < G; q++)
            {
                const bool ok = p0 + q < w; // wave-uniform
                vv[q]         = ok ? v[(p0 + q) * 64] : T(0);
                cc[q]         = ok ? c[(p0 + q) * 64] : -1;
            }
#pragma unroll
            for(int q = 0; q < G; q++)
                xx[q] = x[max(cc[q], 0)];
            if(p0 < full)
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
        if(!reduced)
            r = lanes_sum<T, G>(l);
    }
    if(i < m)
        s_store(y + i, s_finish(r, alpha, beta, y + i), nt);
}

template <typename T, int ORDER>
void sell_launch(hipStream_t s, aoclsparse_int m, aoclsparse_int nslices, const long long *slice_ptr, const T *sval,
                 const aoclsparse_int *scol, const aoclsparse_int *rowlen, T alpha, const T *x, T beta, T *y)
{
    // one slice per workgroup while the launch is small (every slice its own CU), two otherwise
    // (swept on the headline workload: 1 / 2 / 4 / 8 slices per workgroup = 0.221 / 0.218 / 0.221 / 0.222 ms)
    const bool nt = (size_t)m * sizeof(T) > ((size_t)32 << 20);
    if(nslices < 2048)
        hipLaunchKernelGGL((sell_mv_kernel<T, ORDER, 1>), dim3(nslices), dim3(64), 0, s, m, nslices, slice_ptr, sval,
                           scol, rowlen, alpha, x, beta, y, nt);
    else
        hipLaunchKernelGGL((sell_mv_kernel<T, ORDER, 2>), dim3((nslices + 1) / 2), dim3(128), 0, s, m, nslices,
                           slice_ptr, sval, scol, rowlen, alpha, x, beta, y, nt);
}

} // namespace

template <typename T>
aoclsparse_status launch_sell_fill(hipStream_t s, aoclsparse_int m, int base, const aoclsparse_int *row_ptr,
                                   const aoclsparse_int *col, const T *val, aoclsparse_int nslices,
                                   const long long *slice_ptr, T *sval, aoclsparse_int *scol, aoclsparse_int *rowlen)
{
    if(nslices <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((sell_fill_kernel<T>), dim3((nslices + 3) / 4), dim3(256), 0, s, m, base, row_ptr, col, val,
                       nslices, slice_ptr, sval, scol, rowlen);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_sellmv(hipStream_t s, int order, T alpha, aoclsparse_int m, aoclsparse_int nslices,
                                const long long *slice_ptr, const T *sval, const aoclsparse_int *scol,
                                const aoclsparse_int *rowlen, const T *x, T beta, T *y)
{
    if(m <= 0 || nslices <= 0)
        return aoclsparse_status_success;
    switch(order)
    {
    case 0:
        sell_launch<T, 0>(s, m, nslices, slice_ptr, sval, scol, rowlen, alpha, x, beta, y);
        break;
    case 1:
        sell_launch<T, 1>(s, m, nslices, slice_ptr, sval, scol, rowlen, alpha, x, beta, y);
        break;
    case 2:
        sell_launch<T, 2>(s, m, nslices, slice_ptr, sval, scol, rowlen, alpha, x, beta, y);
        break;
    default:
        return aoclsparse_status_invalid_kid;
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_SELL_INSTANTIATE(T)                                                                                     \
    template aoclsparse_status launch_sell_fill<T>(hipStream_t, aoclsparse_int, int, const aoclsparse_int *,          \
                                                   const aoclsparse_int *, const T *, aoclsparse_int,                 \
                                                   const long long *, T *, aoclsparse_int *, aoclsparse_int *);       \
    template aoclsparse_status launch_sellmv<T>(hipStream_t, int, T, aoclsparse_int, aoclsparse_int,                  \
                                                const long long *, const T *, const aoclsparse_int *,                 \
                                                const aoclsparse_int *, const T *, T, T *);
MI355_SELL_INSTANTIATE(double)
MI355_SELL_INSTANTIATE(float)

} // namespace mi355
