// spmv_kernels.hip -- CSR-Adaptive SpMV for gfx950 (wave64, 256 CUs / 8 XCDs, 160 KiB LDS per CU).
//
// One workgroup (256 threads = 4 wavefronts) owns one ROW BLOCK: consecutive whole rows whose
// non-zeros fit one LDS tile (SPMV_TILE = 2048).  Phase 1 streams val[] / col_ind[] of the block
// with fully coalesced loads, gathers x[col] (L2 / Infinity-Cache hits) and parks {val, x} in LDS.
// Phase 2 reduces each row out of LDS in EXACTLY the summation order of the reference CPU kernel
// the reference would dispatch:
//   order 0: one lane per row, left-to-right FMA chain     = ref_csrmv_gn        (csrmv_kr.hpp:448-513)
//   order 1: 4 lanes per row, j mod 4, (l0+l1)+(l2+l3), tail = ..._vectorized_avx2 (csrmv_kr.hpp:949-1040)
//   order 2: 8 lanes per row, j mod 8, AVX-512 tree, tail    = ..._vectorized_avx512 (csrmv_avx512.cpp:36-134)
//            (float: the AVX2 8-lane tree of csrmv_kr.hpp:734-831)
// so y is bit-identical to that CPU kernel (GPU fma == x86 vfmadd).  A row longer than a tile gets a
// workgroup of its own: STRICT keeps the reference order (tiles through LDS, owner lanes chain),
// otherwise a wavefront tree is used (documented componentwise bound).
//
// HBM traffic per launch = the algorithmic bytes: val 8 B + col 4 B per nnz, row_ptr 4 B + y 8 B per
// row, x 8 B per column once (re-reads are cache hits).  Roofline: HBM (AI ~ 0.16 flop/B).
#include "internal.hpp"

#include <hip/hip_runtime.h>

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

// workgroup-wide sum (any order) for the non-strict long-row path
template <typename T>
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
        for(int w = 0; w < SPMV_BLOCK / 64; w++)
            r += scratch[w];
    return r;
}

template <typename T, int ORDER, bool STRICT>
__global__ __launch_bounds__(SPMV_BLOCK) void csr_adaptive_kernel(const aoclsparse_int *__restrict__ rowblocks,
                                                                  const aoclsparse_int *__restrict__ row_ptr,
                                                                  const aoclsparse_int *__restrict__ col,
                                                                  const T *__restrict__ val,
                                                                  const T *__restrict__ x,
                                                                  T *__restrict__ y,
                                                                  T              alpha,
                                                                  T              beta,
                                                                  int            base,
                                                                  int            nblocks,
                                                                  int            chunk)
{
    __shared__ T s_val[SPMV_TILE];
    __shared__ T s_x[SPMV_TILE];
    constexpr int L   = lanes_of<ORDER>::value;
    const int     tid = threadIdx.x;
    // XCD-aware order: workgroups with equal blockIdx%8 share an XCD (one L2); give each XCD a
    // contiguous eighth of the row blocks so its x windows stay in its own L2.
    const int b = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if(b >= nblocks)
        return;
    const int r0  = rowblocks[b];
    const int r1  = rowblocks[b + 1];
    const int p0  = row_ptr[r0] - base;
    const int cnt = (row_ptr[r1] - base) - p0;

    if(cnt <= SPMV_TILE)
    {
        // ---- phase 1: coalesced stream of the block's non-zeros into LDS ------------------------
#pragma unroll
        for(int k = 0; k < SPMV_TILE / SPMV_BLOCK; k++)
        {
            const int i = tid + k * SPMV_BLOCK;
            if(i < cnt)
            {
                const T   v = val[p0 + i];
                const int c = col[p0 + i] - base;
                s_val[i]    = v;
                s_x[i]      = x[c];
            }
        }
        __syncthreads();
        // ---- phase 2: per-row reduction in the reference order --------------------------------------
        const int grp  = tid / L;
        const int lane = tid % L;
        for(int r = r0 + grp; r < r1; r += SPMV_BLOCK / L)
        {
            const int s = row_ptr[r] - base - p0;
            const int e = row_ptr[r + 1] - base - p0;
            T         acc = T(0);
            if constexpr(L == 1)
            {
                for(int j = s; j < e; j++)
                    acc = dev_fma(s_val[j], s_x[j], acc);
                y[r] = finish(acc, alpha, beta, &y[r]);
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
                    y[r] = finish(res, alpha, beta, &y[r]);
                }
            }
        }
    }
    else
    {
        // ---- long row: this workgroup owns the single row r0 ------------------------------------------
        const int n = cnt;
        if constexpr(STRICT)
        {
            const int nfull = n & ~(L - 1);
            T         acc   = T(0);
            for(int t0 = 0; t0 < nfull; t0 += SPMV_TILE)
            {
                const int tn = min(SPMV_TILE, nfull - t0);
                __syncthreads();
#pragma unroll
                for(int k = 0; k < SPMV_TILE / SPMV_BLOCK; k++)
                {
                    const int i = tid + k * SPMV_BLOCK;
                    if(i < tn)
                    {
                        s_val[i] = val[p0 + t0 + i];
                        s_x[i]   = x[col[p0 + t0 + i] - base];
                    }
                }
                __syncthreads();
                if(tid < L)
                    for(int j = tid; j < tn; j += L)
                        acc = dev_fma(s_val[j], s_x[j], acc);
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
                        res = dev_fma(val[p0 + j], x[col[p0 + j] - base], res);
                    y[r0] = finish(res, alpha, beta, &y[r0]);
                }
            }
        }
        else
        {
            // wavefront tree: same flops, order differs (bound stated in DESIGN.md / tests)
            T acc = T(0);
            for(int j = tid; j < n; j += SPMV_BLOCK)
                acc = dev_fma(val[p0 + j], x[col[p0 + j] - base], acc);
            T res = block_sum(acc, s_val);
            if(tid == 0)
                y[r0] = finish(res, alpha, beta, &y[r0]);
        }
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

template <typename T, int ORDER, bool STRICT>
static void launch_inst(hipStream_t s, int base, T alpha, const T *val, const aoclsparse_int *col,
                        const aoclsparse_int *row_ptr, const aoclsparse_int *rowblocks,
                        aoclsparse_int nblocks, const T *x, T beta, T *y)
{
    const int chunk = (nblocks + 7) / 8;
    hipLaunchKernelGGL((csr_adaptive_kernel<T, ORDER, STRICT>), dim3(chunk * 8), dim3(SPMV_BLOCK), 0, s,
                       rowblocks, row_ptr, col, val, x, y, alpha, beta, base, (int)nblocks, chunk);
}

template <typename T>
aoclsparse_status launch_csrmv(hipStream_t s, int order, bool strict, int base, T alpha,
                               aoclsparse_int m, const T *val, const aoclsparse_int *col,
                               const aoclsparse_int *row_ptr, const aoclsparse_int *rowblocks,
                               aoclsparse_int nblocks, const T *x, T beta, T *y)
{
    if(m <= 0 || nblocks <= 0)
        return aoclsparse_status_success;
#define MI355_CASE(O, S)                                                                            \
    launch_inst<T, O, S>(s, base, alpha, val, col, row_ptr, rowblocks, nblocks, x, beta, y);        \
    break
    switch(order * 2 + (strict ? 1 : 0))
    {
    case 0:
        MI355_CASE(0, false);
    case 1:
        MI355_CASE(0, true);
    case 2:
        MI355_CASE(1, false);
    case 3:
        MI355_CASE(1, true);
    case 4:
        MI355_CASE(2, false);
    case 5:
        MI355_CASE(2, true);
    default:
        return aoclsparse_status_invalid_kid;
    }
#undef MI355_CASE
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
    template aoclsparse_status launch_csrmv<T>(hipStream_t, int, bool, int, T, aoclsparse_int, const T *,      \
                                               const aoclsparse_int *, const aoclsparse_int *,                  \
                                               const aoclsparse_int *, aoclsparse_int, const T *, T, T *);      \
    template aoclsparse_status launch_scale<T>(hipStream_t, T *, aoclsparse_int, T);                           \
    template aoclsparse_status launch_strided_gather<T>(hipStream_t, const T *, aoclsparse_int,                 \
                                                        aoclsparse_int, T *);                                   \
    template aoclsparse_status launch_strided_scatter<T>(hipStream_t, const T *, aoclsparse_int, T *,           \
                                                         aoclsparse_int);
MI355_INSTANTIATE(double)
MI355_INSTANTIATE(float)

} // namespace mi355

using namespace mi355;

extern "C" {

aoclsparse_status mi355_dcsrmv(void *stream, aoclsparse_int order, aoclsparse_int strict, aoclsparse_int base,
                               double alpha, aoclsparse_int m, const double *val, const aoclsparse_int *col,
                               const aoclsparse_int *row_ptr, const aoclsparse_int *rowblocks,
                               aoclsparse_int nblocks, const double *x, double beta, double *y)
{
    if(!val || !col || !row_ptr || !rowblocks || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(m < 0 || nblocks < 0)
        return aoclsparse_status_invalid_size;
    if(base != 0 && base != 1)
        return aoclsparse_status_invalid_value;
    return launch_csrmv<double>((hipStream_t)stream, order, strict != 0, base, alpha, m, val, col, row_ptr,
                                rowblocks, nblocks, x, beta, y);
}

aoclsparse_status mi355_scsrmv(void *stream, aoclsparse_int order, aoclsparse_int strict, aoclsparse_int base,
                               float alpha, aoclsparse_int m, const float *val, const aoclsparse_int *col,
                               const aoclsparse_int *row_ptr, const aoclsparse_int *rowblocks,
                               aoclsparse_int nblocks, const float *x, float beta, float *y)
{
    if(!val || !col || !row_ptr || !rowblocks || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(m < 0 || nblocks < 0)
        return aoclsparse_status_invalid_size;
    if(base != 0 && base != 1)
        return aoclsparse_status_invalid_value;
    return launch_csrmv<float>((hipStream_t)stream, order, strict != 0, base, alpha, m, val, col, row_ptr,
                               rowblocks, nblocks, x, beta, y);
}

} // extern "C"
