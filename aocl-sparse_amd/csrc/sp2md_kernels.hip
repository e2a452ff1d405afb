// sp2md_kernels.hip -- sparse x sparse with a DENSE result (aoclsparse_?sp2md / ?spmmd), CSR -> dense
// (aoclsparse_?csr2dense) and the sparse sum C = alpha*op(A) + B (aoclsparse_?add), gfx950.
//
// Reference: level3/aoclsparse_sp2md.hpp:40-168 (row / column layout kernels), conversion/aoclsparse_convert.hpp:658-929
// (csr2dense), level3/aoclsparse_csradd.hpp:33-281 (count + fill of the sum).
//
// sp2md: element C(i,c) receives its products in the order "walk row i of op(A) left to right, for each entry walk
// the matching row of op(B)"; every product is added with one contracted multiply-add onto alpha*a (rounded once).
// Here one WAVEFRONT owns one row of C and walks op(A)'s row serially; the 64 lanes spread over the entries of the
// current op(B) row, which hit distinct elements of C (a CSR row holds a column once), so each element sees the
// reference's chain.  Accesses of one wavefront to one address are performed in program order by the hardware
// (wavefront-scope fences emit no instructions on gfx9); the fence below only stops the compiler from moving them.
// All of it is HBM/L2 traffic on C: 2 * 8 B per product for fp64 plus 12 B per entry of A and of the touched B rows.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

namespace
{
__device__ __forceinline__ double d_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float d_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}
template <typename R>
__device__ __forceinline__ cplx<R> d_fma(cplx<R> a, cplx<R> b, cplx<R> c)
{
    c.re = d_fma(a.re, b.re, c.re);
    c.re = d_fma(-a.im, b.im, c.re);
    c.im = d_fma(a.re, b.im, c.im);
    c.im = d_fma(a.im, b.re, c.im);
    return c;
}
__device__ __forceinline__ double d_mul(double a, double b)
{
    return a * b;
}
__device__ __forceinline__ float d_mul(float a, float b)
{
    return a * b;
}
template <typename R>
__device__ __forceinline__ cplx<R> d_mul(cplx<R> a, cplx<R> b)
{
    return cplx<R>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
__device__ __forceinline__ double d_add(double a, double b)
{
    return a + b;
}
__device__ __forceinline__ float d_add(float a, float b)
{
    return a + b;
}
template <typename R>
__device__ __forceinline__ cplx<R> d_add(cplx<R> a, cplx<R> b)
{
    return cplx<R>(a.re + b.re, a.im + b.im);
}
__device__ __forceinline__ double d_conj(double a, bool)
{
    return a;
}
__device__ __forceinline__ float d_conj(float a, bool)
{
    return a;
}
template <typename R>
__device__ __forceinline__ cplx<R> d_conj(cplx<R> a, bool on)
{
    return on ? cplx<R>(a.re, -a.im) : a;
}
__device__ __forceinline__ bool d_is_zero(double a)
{
    return a == 0.0;
}
__device__ __forceinline__ bool d_is_zero(float a)
{
    return a == 0.0f;
}
template <typename R>
__device__ __forceinline__ bool d_is_zero(cplx<R> a)
{
    return a.re == R(0) && a.im == R(0);
}
template <typename T>
__device__ __forceinline__ T d_const(double v)
{
    return T(static_cast<decltype(T{}.re)>(v));
}
template <>
__device__ __forceinline__ double d_const<double>(double v)
{
    return v;
}
template <>
__device__ __forceinline__ float d_const<float>(double v)
{
    return (float)v;
}
} // namespace

// sp2md.hpp:366-379 / :409-420: beta == 0 stores zeros, beta == 1 is skipped by the launcher, otherwise C *= beta
template <typename T>
__global__ void dense_scale_kernel(T *C, aoclsparse_int inner, aoclsparse_int outer, long long ld, T beta, bool zero)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= inner)
        return;
    for(aoclsparse_int o = blockIdx.y; o < outer; o += gridDim.y)
    {
        T *p = C + (size_t)o * ld + i;
        *p   = zero ? d_const<T>(0.0) : d_mul(beta, *p);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sp2md_kernel(aoclsparse_int m, int base_a, const aoclsparse_int *__restrict__ ptr_a,
                                                    const aoclsparse_int *__restrict__ ind_a,
                                                    const T *__restrict__ val_a, bool conj_a, int base_b,
                                                    const aoclsparse_int *__restrict__ ptr_b,
                                                    const aoclsparse_int *__restrict__ ind_b,
                                                    const T *__restrict__ val_b, bool conj_b, T alpha, T *C,
                                                    long long rs, long long cs)
{
    const int lane   = threadIdx.x & 63;
    const int wave   = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for(aoclsparse_int row = wave; row < m; row += nwaves)
    {
        T                   *crow = C + (long long)row * rs;
        const aoclsparse_int a0 = ptr_a[row] - base_a, a1 = ptr_a[row + 1] - base_a;
        for(aoclsparse_int j = a0; j < a1; j++)
        {
            const T              v  = d_mul(alpha, d_conj(val_a[j], conj_a));
            const aoclsparse_int c  = ind_a[j] - base_a;
            const aoclsparse_int b0 = ptr_b[c] - base_b, b1 = ptr_b[c + 1] - base_b;
            for(aoclsparse_int k = b0 + lane; k < b1; k += 64)
            {
                T *p = crow + (long long)(ind_b[k] - base_b) * cs;
                *p   = d_fma(v, d_conj(val_b[k], conj_b), *p);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

template <typename T>
aoclsparse_status launch_dense_scale(hipStream_t s, T *C, aoclsparse_int inner, aoclsparse_int outer, long long ld,
                                     T beta, bool zero)
{
    if(inner <= 0 || outer <= 0)
        return aoclsparse_status_success;
    const int gy = outer < 32768 ? outer : 32768;
    hipLaunchKernelGGL((dense_scale_kernel<T>), dim3((inner + 255) / 256, gy), dim3(256), 0, s, C, inner, outer, ld, beta,
                       zero);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_sp2md(hipStream_t s, aoclsparse_int m, int base_a, const aoclsparse_int *ptr_a,
                               const aoclsparse_int *ind_a, const T *val_a, bool conj_a, int base_b,
                               const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b, const T *val_b, bool conj_b,
                               T alpha, T *C, long long rs, long long cs)
{
    if(m <= 0)
        return aoclsparse_status_success;
    long long blocks = ((long long)m + 3) / 4;
    if(blocks > 65536)
        blocks = 65536;
    hipLaunchKernelGGL((sp2md_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, m, base_a, ptr_a, ind_a, val_a, conj_a,
                       base_b, ptr_b, ind_b, val_b, conj_b, alpha, C, rs, cs);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// ---- csr2dense (convert.hpp:658-929) ---------------------------------------------------------------------------
// One thread per CSR row keeps the reference's "last writer wins" for repeated entries of a row.  mode: 0 general,
// 1 symmetric, 2 hermitian (mirror is conjugated), 3 triangular.  fill: 0 lower, 1 upper.  diag: 0 non-unit, 1 unit,
// 2 zero.  rs / cs are the element strides of a row / column step of the dense matrix.
template <typename T>
__global__ void csr2dense_kernel(aoclsparse_int m, int base, const aoclsparse_int *__restrict__ ptr,
                                 const aoclsparse_int *__restrict__ ind, const T *__restrict__ val, T *A, long long rs,
                                 long long cs, int mode, int fill, int diag)
{
    const aoclsparse_int row = blockIdx.x * blockDim.x + threadIdx.x;
    if(row >= m)
        return;
    if(mode != 0 && diag == 1)
        A[row * rs + row * cs] = d_const<T>(1.0);
    else if(mode != 0 && diag == 2)
        A[row * rs + row * cs] = d_const<T>(0.0);
    for(aoclsparse_int at = ptr[row] - base; at < ptr[row + 1] - base; at++)
    {
        const aoclsparse_int col = ind[at] - base;
        const T              v   = val[at];
        if(mode == 0)
        {
            A[row * rs + col * cs] = v;
            continue;
        }
        if(col == row)
        {
            if(diag == 0)
                A[row * rs + col * cs] = v;
            continue;
        }
        if((fill == 0 && col < row) || (fill == 1 && col > row))
        {
            A[row * rs + col * cs] = v;
            if(mode == 1)
                A[col * rs + row * cs] = v;
            else if(mode == 2)
                A[col * rs + row * cs] = d_conj(v, true);
        }
    }
}

template <typename T>
aoclsparse_status launch_csr2dense(hipStream_t s, aoclsparse_int m, int base, const aoclsparse_int *ptr,
                                   const aoclsparse_int *ind, const T *val, T *A, long long rs, long long cs, int mode,
                                   int fill, int diag)
{
    if(m <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((csr2dense_kernel<T>), dim3((m + 255) / 256), dim3(256), 0, s, m, base, ptr, ind, val, A, rs, cs,
                       mode, fill, diag);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// ---- C = alpha * op(A) + B (csradd.hpp:33-281) ------------------------------------------------------------------
// Row i of C lists row i of op(A) first (every stored entry, scaled), then the entries of row i of B whose column
// is not in that list, in B's order; a B entry whose column is present is added onto the matching A entry.
// One wavefront per row; lane l takes B entries l, l+64, ... and looks its column up in A's row.
// C carries A's base (csradd.hpp:245, :267-268); base_c is passed apart from base_a because the handle's cached
// transpose (op = T / H) is stored zero-based whatever the base of A.
__device__ __forceinline__ int add_find(const aoclsparse_int *ind_a, aoclsparse_int a0, aoclsparse_int a1, aoclsparse_int col)
{
    int f = -1;
    for(aoclsparse_int j = a0; j < a1; j++)
        if(ind_a[j] == col)
            f = j - a0; // the latest entry, as col_rec does
    return f;
}

template <typename T, bool FILL>
__global__ __launch_bounds__(256) void csradd_kernel(aoclsparse_int m, int base_a, const aoclsparse_int *__restrict__ ptr_a,
                                                     const aoclsparse_int *__restrict__ ind_a,
                                                     const T *__restrict__ val_a, bool conj_a, T alpha, int base_b,
                                                     const aoclsparse_int *__restrict__ ptr_b,
                                                     const aoclsparse_int *__restrict__ ind_b,
                                                     const T *__restrict__ val_b, int base_c, const aoclsparse_int *ptr_c,
                                                     aoclsparse_int *cnt_or_ind_c, T *val_c)
{
    const int lane   = threadIdx.x & 63;
    const int wave   = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for(aoclsparse_int row = wave; row < m; row += nwaves)
    {
        const aoclsparse_int a0 = ptr_a[row] - base_a, a1 = ptr_a[row + 1] - base_a;
        const aoclsparse_int b0 = ptr_b[row] - base_b, b1 = ptr_b[row + 1] - base_b;
        aoclsparse_int       out = a1 - a0; // next free slot of the row
        aoclsparse_int       c0  = 0;
        if(FILL)
        {
            c0 = ptr_c[row] - base_c;
            for(aoclsparse_int j = a0 + lane; j < a1; j += 64)
            {
                cnt_or_ind_c[c0 + j - a0] = ind_a[j] - base_a + base_c;
                val_c[c0 + j - a0]        = d_mul(alpha, d_conj(val_a[j], conj_a));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        for(aoclsparse_int kb = b0; kb < b1; kb += 64)
        {
            const aoclsparse_int k     = kb + lane;
            const bool           live  = k < b1;
            const aoclsparse_int col   = live ? ind_b[k] - base_b : -1; // zero-based
            const int            f     = live ? add_find(ind_a, a0, a1, col + base_a) : 0;
            const bool           fresh = live && f < 0;
            const unsigned long long mask = __ballot(fresh);
            if(FILL && live)
            {
                if(fresh)
                {
                    const int slot         = out + __popcll(mask & ((1ull << lane) - 1ull));
                    cnt_or_ind_c[c0 + slot] = col + base_c;
                    val_c[c0 + slot]        = val_b[k];
                }
                else
                    val_c[c0 + f] = d_add(val_c[c0 + f], val_b[k]);
            }
            out += __popcll(mask);
        }
        if(!FILL && lane == 0)
            cnt_or_ind_c[row] = out;
    }
}

template <typename T>
aoclsparse_status launch_csradd(hipStream_t s, bool fill, aoclsparse_int m, int base_a, const aoclsparse_int *ptr_a,
                                const aoclsparse_int *ind_a, const T *val_a, bool conj_a, T alpha, int base_b,
                                const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b, const T *val_b, int base_c,
                                const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c)
{
    if(m <= 0)
        return aoclsparse_status_success;
    long long blocks = ((long long)m + 3) / 4;
    if(blocks > 65536)
        blocks = 65536;
    if(fill)
        hipLaunchKernelGGL((csradd_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, m, base_a, ptr_a, ind_a,
                           val_a, conj_a, alpha, base_b, ptr_b, ind_b, val_b, base_c, ptr_c, cnt_or_ind_c, val_c);
    else
        hipLaunchKernelGGL((csradd_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, m, base_a, ptr_a, ind_a,
                           val_a, conj_a, alpha, base_b, ptr_b, ind_b, val_b, base_c, ptr_c, cnt_or_ind_c, val_c);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_INST_SP2MD(T)                                                                                             \
    template aoclsparse_status launch_dense_scale<T>(hipStream_t, T *, aoclsparse_int, aoclsparse_int, long long, T,    \
                                                     bool);                                                             \
    template aoclsparse_status launch_sp2md<T>(hipStream_t, aoclsparse_int, int, const aoclsparse_int *,                \
                                               const aoclsparse_int *, const T *, bool, int, const aoclsparse_int *,    \
                                               const aoclsparse_int *, const T *, bool, T, T *, long long, long long);  \
    template aoclsparse_status launch_csr2dense<T>(hipStream_t, aoclsparse_int, int, const aoclsparse_int *,            \
                                                   const aoclsparse_int *, const T *, T *, long long, long long, int,   \
                                                   int, int);                                                           \
    template aoclsparse_status launch_csradd<T>(hipStream_t, bool, aoclsparse_int, int, const aoclsparse_int *,         \
                                                const aoclsparse_int *, const T *, bool, T, int,                        \
                                                const aoclsparse_int *, const aoclsparse_int *, const T *, int,         \
                                                const aoclsparse_int *, aoclsparse_int *, T *);
MI355_INST_SP2MD(double)
MI355_INST_SP2MD(float)
MI355_INST_SP2MD(cdouble)
MI355_INST_SP2MD(cfloat)

} // namespace mi355
