// level1_kernels.hip -- sparse-vector (level 1) operations on a compressed vector (x, indx) and a dense vector y, gfx950.
//
// Reference: level1/aoclsparse_axpyi.hpp:35-50, aoclsparse_dot.hpp:33-61, aoclsparse_gthr.hpp:33-62,
// aoclsparse_sctr.hpp:34-53, aoclsparse_roti.hpp:36-55.  Every entry is independent (indx holds distinct positions),
// so one lane per entry reproduces the reference's element arithmetic exactly (axpyi: one contracted multiply-add);
// only the dot products are reductions: fixed grid, per-lane strided partial sums, LDS tree, then one block over the
// partials -- deterministic, within the usual n*eps of the reference's serial / AVX orders.
// All of them are gather/scatter streams: 12 B (index + value) per entry plus one 32 B sector of y touched per entry.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{
namespace
{
constexpr int L1_BLOCK = 256, L1_DOT_BLOCKS = 1024;

__device__ __forceinline__ double l_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float  l_fma(float a, float b, float c) { return fmaf(a, b, c); }
template <typename R>
__device__ __forceinline__ cplx<R> l_fma(cplx<R> a, cplx<R> b, cplx<R> c)
{
    c.re = l_fma(a.re, b.re, c.re);
    c.re = l_fma(-a.im, b.im, c.re);
    c.im = l_fma(a.re, b.im, c.im);
    c.im = l_fma(a.im, b.re, c.im);
    return c;
}
__device__ __forceinline__ double l_add(double a, double b) { return a + b; }
__device__ __forceinline__ float  l_add(float a, float b) { return a + b; }
template <typename R>
__device__ __forceinline__ cplx<R> l_add(cplx<R> a, cplx<R> b) { return cplx<R>(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ double l_conj(double a, bool) { return a; }
__device__ __forceinline__ float  l_conj(float a, bool) { return a; }
template <typename R>
__device__ __forceinline__ cplx<R> l_conj(cplx<R> a, bool on) { return on ? cplx<R>(a.re, -a.im) : a; }
template <typename T> __device__ __forceinline__ T l_zero();
template <> __device__ __forceinline__ double  l_zero<double>() { return 0.0; }
template <> __device__ __forceinline__ float   l_zero<float>() { return 0.0f; }
template <> __device__ __forceinline__ cdouble l_zero<cdouble>() { return cdouble(0.0, 0.0); }
template <> __device__ __forceinline__ cfloat  l_zero<cfloat>() { return cfloat(0.0f, 0.0f); }

// position of entry i in y: indexed (indx != nullptr) or strided
__device__ __forceinline__ long long l_pos(const aoclsparse_int *indx, long long stride, aoclsparse_int i)
{
    return indx ? (long long)indx[i] : stride * (long long)i;
}
} // namespace

template <typename T>
__global__ void axpyi_kernel(aoclsparse_int nnz, T a, const T *__restrict__ x, const aoclsparse_int *__restrict__ indx, T *y)
{
    for(aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x)
    {
        const aoclsparse_int p = indx[i];
        y[p]                   = l_fma(a, x[i], y[p]);
    }
}

// mode 0: x[i] = y[p]; 1: also y[p] = 0 (gthrz); 2: y[p] = x[i] (sctr)
template <typename T>
__global__ void gather_scatter_kernel(aoclsparse_int nnz, T *x, const aoclsparse_int *__restrict__ indx, long long stride,
                                      T *y, int mode)
{
    for(aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x)
    {
        const long long p = l_pos(indx, stride, i);
        if(mode == 2)
            y[p] = x[i];
        else
        {
            x[i] = y[p];
            if(mode == 1)
                y[p] = l_zero<T>();
        }
    }
}

template <typename T>
__global__ void roti_kernel(aoclsparse_int nnz, T *x, const aoclsparse_int *__restrict__ indx, T *y, T c, T s)
{
    for(aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x)
    {
        const aoclsparse_int p  = indx[i];
        const T              xv = x[i], yv = y[p];
        x[i]                    = l_fma(c, xv, s * yv); // roti.hpp:51-52 with the compiler's contraction
        y[p]                    = l_fma(c, yv, -(s * xv));
    }
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *lds)
{
    lds[threadIdx.x] = v;
    __syncthreads();
    for(int w = L1_BLOCK / 2; w > 0; w >>= 1)
    {
        if((int)threadIdx.x < w)
            lds[threadIdx.x] = l_add(lds[threadIdx.x], lds[threadIdx.x + w]);
        __syncthreads();
    }
    return lds[0];
}

template <typename T>
__global__ __launch_bounds__(L1_BLOCK) void doti_partial_kernel(aoclsparse_int nnz, const T *__restrict__ x,
                                                               const aoclsparse_int *__restrict__ indx,
                                                               const T *__restrict__ y, bool conj, T *partial)
{
    __shared__ T lds[L1_BLOCK];
    T            acc = l_zero<T>();
    for(aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x)
        acc = l_fma(l_conj(x[i], conj), y[indx[i]], acc);
    const T tot = block_sum(acc, lds);
    if(threadIdx.x == 0)
        partial[blockIdx.x] = tot;
}

template <typename T>
__global__ __launch_bounds__(L1_BLOCK) void doti_final_kernel(int n, const T *__restrict__ partial, T *out)
{
    __shared__ T lds[L1_BLOCK];
    T            acc = l_zero<T>();
    for(int i = threadIdx.x; i < n; i += L1_BLOCK)
        acc = l_add(acc, partial[i]);
    const T tot = block_sum(acc, lds);
    if(threadIdx.x == 0)
        *out = tot;
}

static int l1_grid(aoclsparse_int nnz)
{
    long long b = ((long long)nnz + L1_BLOCK - 1) / L1_BLOCK;
    return (int)(b < 1 ? 1 : b > 16384 ? 16384 : b);
}

template <typename T>
aoclsparse_status launch_axpyi(hipStream_t s, aoclsparse_int nnz, T a, const T *x, const aoclsparse_int *indx, T *y)
{
    hipLaunchKernelGGL((axpyi_kernel<T>), dim3(l1_grid(nnz)), dim3(L1_BLOCK), 0, s, nnz, a, x, indx, y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template <typename T>
aoclsparse_status launch_gather_scatter(hipStream_t s, aoclsparse_int nnz, T *x, const aoclsparse_int *indx,
                                        long long stride, T *y, int mode)
{
    hipLaunchKernelGGL((gather_scatter_kernel<T>), dim3(l1_grid(nnz)), dim3(L1_BLOCK), 0, s, nnz, x, indx, stride, y, mode);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template <typename T>
aoclsparse_status launch_roti(hipStream_t s, aoclsparse_int nnz, T *x, const aoclsparse_int *indx, T *y, T c, T sn)
{
    hipLaunchKernelGGL((roti_kernel<T>), dim3(l1_grid(nnz)), dim3(L1_BLOCK), 0, s, nnz, x, indx, y, c, sn);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
// partial: room for L1_DOT_PARTIALS values of T; out: one T (device)
template <typename T>
aoclsparse_status launch_doti(hipStream_t s, aoclsparse_int nnz, const T *x, const aoclsparse_int *indx, const T *y,
                              bool conj, T *partial, T *out)
{
    int blocks = l1_grid(nnz);
    if(blocks > L1_DOT_BLOCKS)
        blocks = L1_DOT_BLOCKS;
    hipLaunchKernelGGL((doti_partial_kernel<T>), dim3(blocks), dim3(L1_BLOCK), 0, s, nnz, x, indx, y, conj, partial);
    hipLaunchKernelGGL((doti_final_kernel<T>), dim3(1), dim3(L1_BLOCK), 0, s, blocks, partial, out);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_INST_L1(T)                                                                                              \
    template aoclsparse_status launch_axpyi<T>(hipStream_t, aoclsparse_int, T, const T *, const aoclsparse_int *, T *); \
    template aoclsparse_status launch_gather_scatter<T>(hipStream_t, aoclsparse_int, T *, const aoclsparse_int *,     \
                                                        long long, T *, int);                                         \
    template aoclsparse_status launch_doti<T>(hipStream_t, aoclsparse_int, const T *, const aoclsparse_int *,         \
                                              const T *, bool, T *, T *);
MI355_INST_L1(double)
MI355_INST_L1(float)
MI355_INST_L1(cdouble)
MI355_INST_L1(cfloat)
template aoclsparse_status launch_roti<double>(hipStream_t, aoclsparse_int, double *, const aoclsparse_int *, double *,
                                               double, double);
template aoclsparse_status launch_roti<float>(hipStream_t, aoclsparse_int, float *, const aoclsparse_int *, float *, float,
                                              float);

} // namespace mi355
