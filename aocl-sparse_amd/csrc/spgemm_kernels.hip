// spgemm_kernels.hip -- C = A * B for two CSR matrices (the sp2m / spmm / csr2m path), gfx950.
//
// Reference: two-stage Gustavson with a dense per-thread accumulator of length n
// (level3/aoclsparse_csr2m.cpp:46-302 count, :310-543 finalize).  Row i of C lists its columns in
// FIRST-TOUCH order (walk row i of A left to right, for each entry walk the matching row of B left to
// right) and each value is accumulated in that same order: first product stored, later ones added with
// a contracted multiply-add (:489-498).  Both properties are reproduced exactly here:
//
//   one WAVEFRONT owns one row of C and consumes the products strictly in the reference's order; the 64
//   lanes only parallelise the membership test "is column c already in this row's list?" (one compare
//   per lane, one ballot per 64 list entries).  The list (and, in the fill pass, the partial sums) lives
//   in LDS when the row's upper bound (sum of the B-row lengths it touches) is <= SPGEMM_LDS_CAP, else
//   in a global scratch slab of exactly that upper bound.
//
// Stage 1 returns the per-row counts; the host prefix-sums them (64-bit, as the reference) and
// allocates C; stage 2 writes col_ind / val of C at row_ptr_C.  Integer output is bit-exact, fp output
// bit-identical to the reference's single-thread order.
//
// Bound: neither HBM nor MFMA -- irregular, latency/instruction bound; algorithmic traffic is
// 12 B per entry of A and of the touched B rows + 12 B per entry of C.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{


__device__ __forceinline__ double sp_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float sp_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}
// complex: the reference's std::complex multiply-add (csr2m.cpp:489-498 with T = std::complex); component-wise fma
template <typename R>
__device__ __forceinline__ cplx<R> sp_fma(cplx<R> a, cplx<R> b, cplx<R> c)
{
    c.re = sp_fma(a.re, b.re, c.re);
    c.re = sp_fma(-a.im, b.im, c.re);
    c.im = sp_fma(a.re, b.im, c.im);
    c.im = sp_fma(a.im, b.re, c.im);
    return c;
}
__device__ __forceinline__ double sp_mul(double a, double b)
{
    return a * b;
}
__device__ __forceinline__ float sp_mul(float a, float b)
{
    return a * b;
}
template <typename R>
__device__ __forceinline__ cplx<R> sp_mul(cplx<R> a, cplx<R> b)
{
    return sp_fma(a, b, cplx<R>(R(0), R(0)));
}
__device__ __forceinline__ double sp_conj(double a, bool)
{
    return a;
}
__device__ __forceinline__ float sp_conj(float a, bool)
{
    return a;
}
template <typename R>
__device__ __forceinline__ cplx<R> sp_conj(cplx<R> a, bool on)
{
    return on ? cplx<R>(a.re, -a.im) : a;
}

// position of c in list[0..len) or -1; wave-uniform result
__device__ __forceinline__ int wave_find(const int *list, int len, int c, int lane)
{
    for(int b0 = 0; b0 < len; b0 += 64)
    {
        const int                idx  = b0 + lane;
        const bool               hit  = idx < len && list[idx] == c;
        const unsigned long long mask = __ballot(hit);
        if(mask)
            return b0 + __ffsll((long long)mask) - 1;
    }
    return -1;
}

template <typename T, bool FILL>
__global__ __launch_bounds__(256) void spgemm_row_kernel(aoclsparse_int m, int base_a,
                                                         const aoclsparse_int *__restrict__ ptr_a,
                                                         const aoclsparse_int *__restrict__ ind_a,
                                                         const T *__restrict__ val_a, int base_b,
                                                         const aoclsparse_int *__restrict__ ptr_b,
                                                         const aoclsparse_int *__restrict__ ind_b,
                                                         const T *__restrict__ val_b,
                                                         const long long *__restrict__ slab_off, int *slab_idx,
                                                         T *slab_val, const aoclsparse_int *__restrict__ ptr_c,
                                                         aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a, bool conj_b)
{
    constexpr int  SPGEMM_LDS_CAP = spgemm_lds_cap<T>();
    __shared__ int s_idx[4][SPGEMM_LDS_CAP];
    __shared__ T   s_val[FILL ? 4 : 1][FILL ? SPGEMM_LDS_CAP : 1];
    const int      w    = threadIdx.x >> 6;
    const int      lane = threadIdx.x & 63;
    const int      i    = blockIdx.x * 4 + w;
    if(i >= m)
        return;
    const long long off = slab_off[i];
    const long long ub  = slab_off[i + 1] - off;
    const bool      in_lds = ub <= SPGEMM_LDS_CAP;
    int            *list = in_lds ? s_idx[w] : slab_idx + off;
    T              *acc  = nullptr;
    if constexpr(FILL)
        acc = in_lds ? s_val[w] : slab_val + off;
    int len = 0;
    for(int j = ptr_a[i] - base_a; j < ptr_a[i + 1] - base_a; j++)
    {
        const int ca = ind_a[j] - base_a;
        T         va = T(0);
        if constexpr(FILL)
            va = sp_conj(val_a[j], conj_a);
        for(int k = ptr_b[ca] - base_b; k < ptr_b[ca + 1] - base_b; k++)
        {
            const int c   = ind_b[k] - base_b;
            const int pos = wave_find(list, len, c, lane);
            if(pos < 0)
            {
                if(lane == 0)
                {
                    list[len] = c; // first touch: new entry of C (csr2m.cpp:489-496)
                    if constexpr(FILL)
                        acc[len] = sp_mul(va, sp_conj(val_b[k], conj_b));
                }
                len++;
            }
            else if constexpr(FILL)
            {
                if(lane == 0)
                    acc[pos] = sp_fma(va, sp_conj(val_b[k], conj_b), acc[pos]); // csr2m.cpp:498, contracted
            }
            if(!in_lds)
                __threadfence_block(); // lane 0's global store must be visible to the next compare
        }
    }
    if constexpr(FILL)
    {
        const int dst = ptr_c[i];
        for(int t = lane; t < len; t += 64)
        {
            cnt_or_ind_c[dst + t] = list[t];
            val_c[dst + t]        = acc[t];
        }
    }
    else if(lane == 0)
        cnt_or_ind_c[i] = len;
}

template <typename T>
aoclsparse_status launch_spgemm(hipStream_t s, bool fill, aoclsparse_int m, int base_a,
                                const aoclsparse_int *ptr_a, const aoclsparse_int *ind_a, const T *val_a,
                                int base_b, const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b,
                                const T *val_b, const long long *slab_off, int *slab_idx, T *slab_val,
                                const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a,
                                bool conj_b)
{
    if(m <= 0)
        return aoclsparse_status_success;
    dim3 grid((m + 3) / 4), block(256);
    if(fill)
        hipLaunchKernelGGL((spgemm_row_kernel<T, true>), grid, block, 0, s, m, base_a, ptr_a, ind_a, val_a, base_b,
                           ptr_b, ind_b, val_b, slab_off, slab_idx, slab_val, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b);
    else
        hipLaunchKernelGGL((spgemm_row_kernel<T, false>), grid, block, 0, s, m, base_a, ptr_a, ind_a, val_a, base_b,
                           ptr_b, ind_b, val_b, slab_off, slab_idx, slab_val, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_SPGEMM_INST(T)                                                                                        \
    template aoclsparse_status launch_spgemm<T>(hipStream_t, bool, aoclsparse_int, int, const aoclsparse_int *,     \
                                                const aoclsparse_int *, const T *, int, const aoclsparse_int *,     \
                                                const aoclsparse_int *, const T *, const long long *, int *, T *,   \
                                                const aoclsparse_int *, aoclsparse_int *, T *, bool, bool);
MI355_SPGEMM_INST(double)
MI355_SPGEMM_INST(float)
MI355_SPGEMM_INST(cdouble)
MI355_SPGEMM_INST(cfloat)

} // namespace mi355
