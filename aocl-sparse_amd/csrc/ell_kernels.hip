// ell_kernels.hip -- SpMV on the ELL family (level2/aoclsparse_ellmv.hpp of the reference).
//
//  * ELL, row-major (ell[i*width + p], padding marked by column -1): the rows of a workgroup are
//    contiguous in memory, so a 4-lane group per row reads 32-byte pieces that neighbour each other
//    across the wavefront.  The four lanes ARE the reference's four AVX2 lanes: lane l accumulates
//    entries 4k+l of the full groups, the group is reduced (l0+l1)+(l2+l3), the tail is a scalar
//    chain -- bit-identical to aoclsparse_dellmv_avx2 (:90-208).  float follows the reference's scalar
//    kernel (:34-85), which is what aoclsparse_sellmv runs.
//  * ELLT, column-major (ell[p*m + i]): the GPU-native layout -- lane i of a wavefront owns row i, every
//    load is a coalesced 512-byte (8-byte values) / 256-byte (indices) line, one FMA chain per row
//    exactly as the reference's vector-over-rows kernel (:316-444).
//  * CSR rows by map: the "long rows" of ELLT-HYB (:660-757), a 4-lane group per listed row.
//
// HBM traffic per row: width*(8+4) B streamed + 8 B y (+8 B if beta != 0) + the x gathers.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

__device__ __forceinline__ double ell_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float ell_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}

template <typename T>
__device__ __forceinline__ T finish(T r, T alpha, T beta, const T *ysrc)
{
    if(alpha != T(1))
        r = alpha * r;
    if(beta != T(0))
        r = ell_fma(beta, *ysrc, r); // "result += beta * y" contracts to one FMA in the reference build
    return r;
}

// (l0 + l1) + (l2 + l3) over the 4 lanes of a group; valid on the group's first lane
template <typename T>
__device__ __forceinline__ T group4_sum(T v)
{
    v = v + __shfl_down(v, 1, 4);
    return v + __shfl_down(v, 2, 4);
}

// double: 4 lanes per row (64 rows per 256-thread workgroup)
__global__ __launch_bounds__(256) void ell4_kernel(int base, double alpha, aoclsparse_int m,
                                                    const double *__restrict__ val,
                                                    const aoclsparse_int *__restrict__ col, aoclsparse_int width,
                                                    const double *__restrict__ x, double beta,
                                                    double *__restrict__ y)
{
    const int l = threadIdx.x & 3;
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2);
    if(i >= m)
        return;
    const size_t          off = (size_t)i * (size_t)width;
    const double         *v   = val + off;
    const aoclsparse_int *c   = col + off;
    const int             k_iter = width / 4;
    int                   k_rem  = width % 4, p = 0;
    double                acc = 0.0;
    for(int it = 0; it < k_iter; it++)
    {
        if(c[p + 3] - base < 0) // the group holds padding: the scalar tail walks it (:139-145)
        {
            k_rem = 4;
            break;
        }
        acc = fma(v[p + l], x[c[p + l] - base], acc);
        p += 4;
    }
    double r = k_iter ? group4_sum(acc) : 0.0;
    if(l == 0)
    {
        for(int q = 0; q < k_rem; q++)
        {
            const int cc = c[p + q] - base;
            if(cc < 0)
                break;
            r = fma(v[p + q], x[cc], r);
        }
        y[i] = finish(r, alpha, beta, y + i);
    }
}

// float: the reference's scalar kernel, one lane per row
__global__ __launch_bounds__(256) void ell1_kernel(int base, float alpha, aoclsparse_int m,
                                                    const float *__restrict__ val,
                                                    const aoclsparse_int *__restrict__ col, aoclsparse_int width,
                                                    const float *__restrict__ x, float beta, float *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= m)
        return;
    const size_t off = (size_t)i * (size_t)width;
    float        r   = 0.0f;
    for(int p = 0; p < width; p++)
    {
        const int cc = col[off + p] - base;
        if(cc < 0)
            break;
        r = fmaf(val[off + p], x[cc], r);
    }
    y[i] = finish(r, alpha, beta, y + i);
}

template <typename T>
__global__ __launch_bounds__(256) void ellt_kernel(int base, T alpha, aoclsparse_int m, const T *__restrict__ val,
                                                   const aoclsparse_int *__restrict__ col, aoclsparse_int width,
                                                   const T *__restrict__ x, T beta, T *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= m)
        return;
    T      r = T(0);
    size_t o = (size_t)i;
    int    p = 0;
    for(; p + 4 <= width; p += 4) // four independent coalesced loads in flight per lane
    {
        const T   v0 = val[o], v1 = val[o + m], v2 = val[o + 2 * (size_t)m], v3 = val[o + 3 * (size_t)m];
        const int c0 = col[o] - base, c1 = col[o + m] - base, c2 = col[o + 2 * (size_t)m] - base,
                  c3 = col[o + 3 * (size_t)m] - base;
        const T x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
        r = ell_fma(v0, x0, r), r = ell_fma(v1, x1, r), r = ell_fma(v2, x2, r), r = ell_fma(v3, x3, r);
        o += 4 * (size_t)m;
    }
    for(; p < width; p++, o += m)
        r = ell_fma(val[o], x[col[o] - base], r);
    y[i] = finish(r, alpha, beta, y + i);
}

template <typename T>
__global__ __launch_bounds__(256) void csr_rows4_kernel(int base, T alpha, aoclsparse_int nrows,
                                                        const aoclsparse_int *__restrict__ map,
                                                        const T *__restrict__ val,
                                                        const aoclsparse_int *__restrict__ col,
                                                        const aoclsparse_int *__restrict__ row_ptr,
                                                        const T *__restrict__ x, T beta, const T *__restrict__ ysrc,
                                                        T *__restrict__ y)
{
    const int l = threadIdx.x & 3;
    const int g = blockIdx.x * 64 + (threadIdx.x >> 2);
    if(g >= nrows)
        return;
    const int row = map[g];
    const int s = row_ptr[row] - base, e = row_ptr[row + 1] - base;
    const int full = (e - s) & ~3;
    T         acc  = T(0);
    for(int p = s + l; p < s + full; p += 4)
        acc = ell_fma(val[p], x[col[p] - base], acc);
    T r = full ? group4_sum(acc) : T(0);
    if(l == 0)
    {
        for(int p = s + full; p < e; p++)
            r = ell_fma(val[p], x[col[p] - base], r);
        y[row] = finish(r, alpha, beta, ysrc + g);
    }
}

template <typename T>
__global__ void gather_rows_kernel(aoclsparse_int n, const aoclsparse_int *__restrict__ map, const T *__restrict__ src,
                                   T *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        dst[i] = src[map[i]];
}

template <>
aoclsparse_status launch_ellmv<double>(hipStream_t s, int base, double alpha, aoclsparse_int m, const double *val,
                                       const aoclsparse_int *col, aoclsparse_int width, const double *x, double beta,
                                       double *y)
{
    if(m <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL(ell4_kernel, dim3((m + 63) / 64), dim3(256), 0, s, base, alpha, m, val, col, width, x, beta, y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template <>
aoclsparse_status launch_ellmv<float>(hipStream_t s, int base, float alpha, aoclsparse_int m, const float *val,
                                      const aoclsparse_int *col, aoclsparse_int width, const float *x, float beta,
                                      float *y)
{
    if(m <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL(ell1_kernel, dim3((m + 255) / 256), dim3(256), 0, s, base, alpha, m, val, col, width, x, beta,
                       y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_elltmv(hipStream_t s, int base, T alpha, aoclsparse_int m, const T *val,
                                const aoclsparse_int *col, aoclsparse_int width, const T *x, T beta, T *y)
{
    if(m <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((ellt_kernel<T>), dim3((m + 255) / 256), dim3(256), 0, s, base, alpha, m, val, col, width, x,
                       beta, y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_csr_rows(hipStream_t s, int base, T alpha, aoclsparse_int nrows, const aoclsparse_int *map,
                                  const T *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr, const T *x,
                                  T beta, const T *ysrc, T *y)
{
    if(nrows <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((csr_rows4_kernel<T>), dim3((nrows + 63) / 64), dim3(256), 0, s, base, alpha, nrows, map, val,
                       col, row_ptr, x, beta, ysrc, y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_gather_rows(hipStream_t s, aoclsparse_int n, const aoclsparse_int *map, const T *src, T *dst)
{
    if(n <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((gather_rows_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, s, n, map, src, dst);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_ELL_INSTANTIATE(T)                                                                                     \
    template aoclsparse_status launch_elltmv<T>(hipStream_t, int, T, aoclsparse_int, const T *,                      \
                                                const aoclsparse_int *, aoclsparse_int, const T *, T, T *);          \
    template aoclsparse_status launch_csr_rows<T>(hipStream_t, int, T, aoclsparse_int, const aoclsparse_int *,       \
                                                  const T *, const aoclsparse_int *, const aoclsparse_int *,         \
                                                  const T *, T, const T *, T *);                                     \
    template aoclsparse_status launch_gather_rows<T>(hipStream_t, aoclsparse_int, const aoclsparse_int *, const T *, \
                                                     T *);
MI355_ELL_INSTANTIATE(double)
MI355_ELL_INSTANTIATE(float)

} // namespace mi355
