// csrmm_kernels.hip -- C = alpha*A*B + beta*C, A sparse CSR (m x k), B/C dense, for gfx950.
//
// Arithmetic per output element follows the reference's column-major kernel
// (level3/aoclsparse_csrmm.hpp:69-85): sum = fma(a_ik, B_kj, sum) over the row in CSR order, then
// C = fma(beta, C, alpha*sum).  C is read even when beta == 0, as every reference csrmm kernel does
// (SURVEY.md Appendix B), so NaN/Inf already in C propagate exactly as on the CPU.
//
// HBM-bound (AI ~ 0.6 flop/B at 256 columns, 5 nnz/row): no MFMA -- the dense tiles a 5-point
// stencil would give an MFMA are >90 % zeros, so reshaping to GEMM only adds traffic.
//   row-major  : one lane owns 2 adjacent columns of one C row (16-B loads/stores); the B rows a
//                sparse row touches are contiguous 8*n-byte streams; val/col are wave-uniform loads.
//   column-major: one lane owns one row and a tile of 16 columns held in registers, so A is re-read
//                n/16 times (from L2 once the first pass has pulled it in) and C is written coalesced.
// Algorithmic bytes: (m+1+nnz)*4 + nnz*8 + 8*n*(k + m*(1+[beta!=0]))  (BASELINE.md section 2).
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

template <typename T>
struct vec2;
template <>
struct vec2<double>
{
    using type = double2;
};
template <>
struct vec2<float>
{
    using type = float2;
};

__device__ __forceinline__ double mm_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float mm_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}

// blockDim = (TX, TY): TX lanes span 2*TX columns, TY rows per workgroup
template <typename T, bool VEC2>
__global__ __launch_bounds__(256) void csrmm_row_kernel(int base, T alpha, aoclsparse_int m,
                                                        const T *__restrict__ val,
                                                        const aoclsparse_int *__restrict__ col,
                                                        const aoclsparse_int *__restrict__ row_ptr,
                                                        const T *__restrict__ B, aoclsparse_int n,
                                                        aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                        aoclsparse_int ldc)
{
    using V     = typename vec2<T>::type;
    const int i = blockIdx.x * blockDim.y + threadIdx.y; // rows on grid.x (no 65535 limit)
    if(i >= m)
        return;
    const int s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
    if constexpr(VEC2)
    {
        const int j = 2 * (blockIdx.y * blockDim.x + threadIdx.x);
        if(j >= n)
            return;
        T a0 = T(0), a1 = T(0);
        for(int p = s; p < e; p++)
        {
            const T a = val[p];
            const V b = *reinterpret_cast<const V *>(B + (size_t)(col[p] - base) * ldb + j);
            a0        = mm_fma(a, b.x, a0);
            a1        = mm_fma(a, b.y, a1);
        }
        V *cp = reinterpret_cast<V *>(C + (size_t)i * ldc + j);
        V  c  = *cp;
        c.x   = mm_fma(beta, c.x, alpha * a0);
        c.y   = mm_fma(beta, c.y, alpha * a1);
        *cp   = c;
    }
    else
    {
        const int j = blockIdx.y * blockDim.x + threadIdx.x;
        if(j >= n)
            return;
        T acc = T(0);
        for(int p = s; p < e; p++)
            acc = mm_fma(val[p], B[(size_t)(col[p] - base) * ldb + j], acc);
        T *cp = C + (size_t)i * ldc + j;
        *cp   = mm_fma(beta, *cp, alpha * acc);
    }
}

constexpr int CM_TILE = 16; // columns per lane in the column-major kernel

template <typename T>
__global__ __launch_bounds__(256) void csrmm_col_kernel(int base, T alpha, aoclsparse_int m,
                                                        const T *__restrict__ val,
                                                        const aoclsparse_int *__restrict__ col,
                                                        const aoclsparse_int *__restrict__ row_ptr,
                                                        const T *__restrict__ B, aoclsparse_int n,
                                                        aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                        aoclsparse_int ldc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= m)
        return;
    const int j0 = blockIdx.y * CM_TILE;
    const int nj = min(CM_TILE, n - j0);
    const int s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
    T         acc[CM_TILE];
#pragma unroll
    for(int jj = 0; jj < CM_TILE; jj++)
        acc[jj] = T(0);
    if(nj == CM_TILE)
    {
        for(int p = s; p < e; p++)
        {
            const T  a  = val[p];
            const T *bp = B + (size_t)(col[p] - base) + (size_t)j0 * ldb;
#pragma unroll
            for(int jj = 0; jj < CM_TILE; jj++)
                acc[jj] = mm_fma(a, bp[(size_t)jj * ldb], acc[jj]);
        }
#pragma unroll
        for(int jj = 0; jj < CM_TILE; jj++)
        {
            T *cp = C + (size_t)i + (size_t)(j0 + jj) * ldc;
            *cp   = mm_fma(beta, *cp, alpha * acc[jj]);
        }
    }
    else
    {
        for(int p = s; p < e; p++)
        {
            const T  a  = val[p];
            const T *bp = B + (size_t)(col[p] - base) + (size_t)j0 * ldb;
#pragma unroll
            for(int jj = 0; jj < CM_TILE; jj++)
                if(jj < nj)
                    acc[jj] = mm_fma(a, bp[(size_t)jj * ldb], acc[jj]);
        }
#pragma unroll
        for(int jj = 0; jj < CM_TILE; jj++)
            if(jj < nj)
            {
                T *cp = C + (size_t)i + (size_t)(j0 + jj) * ldc;
                *cp   = mm_fma(beta, *cp, alpha * acc[jj]);
            }
    }
}

// level3/aoclsparse_csrmm.hpp:361-427: beta == 0 stores exact zeros, otherwise C *= beta
template <typename T>
__global__ void scale_dense_kernel(T *C, aoclsparse_int inner, aoclsparse_int outer, aoclsparse_int ld, T beta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int o = blockIdx.y;
    if(i < inner && o < outer)
    {
        T *p = C + (size_t)o * ld + i;
        *p   = beta == T(0) ? T(0) : *p * beta;
    }
}

static int pow2_at_least(int v)
{
    int p = 1;
    while(p < v)
        p <<= 1;
    return p;
}

template <typename T>
aoclsparse_status launch_csrmm(hipStream_t s, aoclsparse_order order, int base, T alpha, aoclsparse_int m,
                               aoclsparse_int /*k*/, const T *val, const aoclsparse_int *col,
                               const aoclsparse_int *row_ptr, const T *B, aoclsparse_int n,
                               aoclsparse_int ldb, T beta, T *C, aoclsparse_int ldc)
{
    if(m <= 0 || n <= 0)
        return aoclsparse_status_success;
    if(order == aoclsparse_order_row)
    {
        const bool vec = (n % 2 == 0) && (ldb % 2 == 0) && (ldc % 2 == 0)
                         && (reinterpret_cast<uintptr_t>(B) % (2 * sizeof(T)) == 0)
                         && (reinterpret_cast<uintptr_t>(C) % (2 * sizeof(T)) == 0);
        const int lanes = vec ? n / 2 : n;
        const int tx    = lanes >= 128 ? 128 : pow2_at_least(lanes);
        const int ty    = 256 / tx;
        dim3      block(tx, ty), grid((m + ty - 1) / ty, (lanes + tx - 1) / tx);
        if(vec)
            hipLaunchKernelGGL((csrmm_row_kernel<T, true>), grid, block, 0, s, base, alpha, m, val, col, row_ptr,
                               B, n, ldb, beta, C, ldc);
        else
            hipLaunchKernelGGL((csrmm_row_kernel<T, false>), grid, block, 0, s, base, alpha, m, val, col,
                               row_ptr, B, n, ldb, beta, C, ldc);
    }
    else
    {
        dim3 block(256), grid((m + 255) / 256, (n + CM_TILE - 1) / CM_TILE);
        hipLaunchKernelGGL((csrmm_col_kernel<T>), grid, block, 0, s, base, alpha, m, val, col, row_ptr, B, n,
                           ldb, beta, C, ldc);
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_scale_dense(hipStream_t s, aoclsparse_order order, T *C, aoclsparse_int m,
                                     aoclsparse_int n, aoclsparse_int ld, T beta)
{
    const aoclsparse_int outer = order == aoclsparse_order_column ? n : m;
    const aoclsparse_int inner = order == aoclsparse_order_column ? m : n;
    if(outer <= 0 || inner <= 0)
        return aoclsparse_status_success;
    // gridDim.y is limited to 65535: walk the outer dimension in slabs
    for(aoclsparse_int o0 = 0; o0 < outer; o0 += 65535)
    {
        const aoclsparse_int cnt = outer - o0 < 65535 ? outer - o0 : 65535;
        hipLaunchKernelGGL((scale_dense_kernel<T>), dim3((inner + 255) / 256, cnt), dim3(256), 0, s,
                           C + (size_t)o0 * ld, inner, cnt, ld, beta);
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_INST_MM(T)                                                                                     \
    template aoclsparse_status launch_csrmm<T>(hipStream_t, aoclsparse_order, int, T, aoclsparse_int,        \
                                               aoclsparse_int, const T *, const aoclsparse_int *,             \
                                               const aoclsparse_int *, const T *, aoclsparse_int,             \
                                               aoclsparse_int, T, T *, aoclsparse_int);                       \
    template aoclsparse_status launch_scale_dense<T>(hipStream_t, aoclsparse_order, T *, aoclsparse_int,     \
                                                     aoclsparse_int, aoclsparse_int, T);
MI355_INST_MM(double)
MI355_INST_MM(float)

} // namespace mi355

extern "C" aoclsparse_status mi355_dcsrmm(void *stream, aoclsparse_int order, aoclsparse_int base, double alpha,
                                          aoclsparse_int m, aoclsparse_int k, const double *val,
                                          const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                                          const double *B, aoclsparse_int n, aoclsparse_int ldb, double beta,
                                          double *C, aoclsparse_int ldc)
{
    if(!val || !col || !row_ptr || !B || !C)
        return aoclsparse_status_invalid_pointer;
    if(m < 0 || n < 0 || k < 0)
        return aoclsparse_status_invalid_size;
    if((order != aoclsparse_order_row && order != aoclsparse_order_column) || (base != 0 && base != 1))
        return aoclsparse_status_invalid_value;
    return mi355::launch_csrmm<double>((hipStream_t)stream, (aoclsparse_order)order, base, alpha, m, k, val, col,
                                       row_ptr, B, n, ldb, beta, C, ldc);
}
