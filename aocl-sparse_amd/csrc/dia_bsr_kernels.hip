// dia_bsr_kernels.hip -- y = alpha*A*x + beta*y for the DIA and BSR storage formats (aoclsparse_?diamv, ?bsrmv), gfx950.
//
// Reference: level2/aoclsparse_diamv.hpp:34-70 (reference kernel: y scaled first, then one pass per diagonal adding
// (alpha*v)*x with a contracted multiply-add) and level2/aoclsparse_bsrmv_kr.hpp:30-154 with the builders of
// aoclsparse_bsrmv_bldr.hpp:47-161 (per scalar row: one chain of multiply-adds over the blocks of the block row and
// the columns inside each block, then *alpha when alpha != 1, then fma(beta, y, .) when beta != 0).
// Both map to ONE LANE PER SCALAR ROW, which keeps exactly those chains, and both stream the value array with
// unit stride across lanes: DIA stores a diagonal as m consecutive values, BSR stores a block column-major, so the
// `dim` rows of a block are consecutive.  HBM-bound: 8 B per stored value (DIA: m per diagonal, padding included;
// BSR: dim^2 per block + 4 B per block index) plus x and y.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{
namespace
{
__device__ __forceinline__ double b_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float  b_fma(float a, float b, float c) { return fmaf(a, b, c); }
} // namespace

template <typename T>
__global__ __launch_bounds__(256) void diamv_kernel(T alpha, aoclsparse_int m, aoclsparse_int n,
                                                    const T *__restrict__ dia_val,
                                                    const aoclsparse_int *__restrict__ dia_offset, aoclsparse_int ndiag,
                                                    const T *__restrict__ x, T beta, T *__restrict__ y)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= m)
        return;
    T acc = beta == T(0) ? T(0) : beta == T(1) ? y[i] : beta * y[i]; // diamv.hpp:47-58
    for(aoclsparse_int d = 0; d < ndiag; d++)
    {
        const long long j = (long long)i + dia_offset[d];
        if(j >= 0 && j < n)
            acc = b_fma(alpha * dia_val[(size_t)d * m + i], x[j], acc);
    }
    y[i] = acc;
}

// DIM > 0: block size known at compile time (the sizes the reference has dedicated kernels for, bsrmv.cpp:120-136), so
// the loads of one block are all in flight before its multiply-add chain starts; DIM == 0: any size.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void bsrmv_kernel(T alpha, aoclsparse_int mb, aoclsparse_int dim_rt, int base,
                                                    const T *__restrict__ val, const aoclsparse_int *__restrict__ col,
                                                    const aoclsparse_int *__restrict__ row_ptr, const T *__restrict__ x,
                                                    T beta, T *__restrict__ y)
{
    const aoclsparse_int dim = DIM > 0 ? DIM : dim_rt;
    const long long      r = (long long)blockIdx.x * blockDim.x + threadIdx.x; // scalar row
    if(r >= (long long)mb * dim)
        return;
    const aoclsparse_int ai = (aoclsparse_int)(r / dim), bi = (aoclsparse_int)(r % dim);
    const size_t         sq = (size_t)dim * dim;
    T                    sum = T(0);
    for(aoclsparse_int aj = row_ptr[ai] - base; aj < row_ptr[ai + 1] - base; aj++)
    {
        const T *v  = val + sq * aj + bi;
        const T *xp = x + (size_t)dim * (col[aj] - base);
        if constexpr(DIM > 0)
        {
            T a[DIM], b[DIM];
#pragma unroll
            for(int bj = 0; bj < DIM; bj++)
                a[bj] = v[DIM * bj], b[bj] = xp[bj];
#pragma unroll
            for(int bj = 0; bj < DIM; bj++)
                sum = b_fma(a[bj], b[bj], sum);
        }
        else
            for(aoclsparse_int bj = 0; bj < dim; bj++)
                sum = b_fma(v[(size_t)dim * bj], xp[bj], sum);
    }
    if(alpha != T(1))
        sum = sum * alpha;
    if(beta != T(0))
        sum = b_fma(beta, y[r], sum);
    y[r] = sum;
}

template <typename T>
aoclsparse_status launch_diamv(hipStream_t s, T alpha, aoclsparse_int m, aoclsparse_int n, const T *dia_val,
                               const aoclsparse_int *dia_offset, aoclsparse_int ndiag, const T *x, T beta, T *y)
{
    if(m <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((diamv_kernel<T>), dim3((m + 255) / 256), dim3(256), 0, s, alpha, m, n, dia_val, dia_offset, ndiag,
                       x, beta, y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_bsrmv(hipStream_t s, T alpha, aoclsparse_int mb, aoclsparse_int dim, int base, const T *val,
                               const aoclsparse_int *col, const aoclsparse_int *row_ptr, const T *x, T beta, T *y)
{
    const long long rows = (long long)mb * dim;
    if(rows <= 0)
        return aoclsparse_status_success;
    const dim3 grid((unsigned)((rows + 255) / 256)), block(256);
#define MI355_BSR_CASE(D)                                                                                              \
    case D:                                                                                                            \
        hipLaunchKernelGGL((bsrmv_kernel<T, D>), grid, block, 0, s, alpha, mb, dim, base, val, col, row_ptr, x, beta, y); \
        break;
    switch(dim)
    {
        MI355_BSR_CASE(2)
        MI355_BSR_CASE(3)
        MI355_BSR_CASE(4)
        MI355_BSR_CASE(5)
        MI355_BSR_CASE(6)
        MI355_BSR_CASE(7)
        MI355_BSR_CASE(8)
        MI355_BSR_CASE(16)
    default:
        hipLaunchKernelGGL((bsrmv_kernel<T, 0>), grid, block, 0, s, alpha, mb, dim, base, val, col, row_ptr, x, beta, y);
    }
#undef MI355_BSR_CASE
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_INST_DB(T)                                                                                             \
    template aoclsparse_status launch_diamv<T>(hipStream_t, T, aoclsparse_int, aoclsparse_int, const T *,            \
                                               const aoclsparse_int *, aoclsparse_int, const T *, T, T *);           \
    template aoclsparse_status launch_bsrmv<T>(hipStream_t, T, aoclsparse_int, aoclsparse_int, int, const T *,       \
                                               const aoclsparse_int *, const aoclsparse_int *, const T *, T, T *);
MI355_INST_DB(double)
MI355_INST_DB(float)

} // namespace mi355
