// complex_kernels.hip -- SpMV for complex handles (aoclsparse_cmv / aoclsparse_zmv).
//
// One sub-wavefront group of G lanes per row (G = 4 .. 64 chosen from the mean row length): the lanes stride
// over the row's entries with coalesced 8 / 16-byte value loads, keep one complex partial sum each and are
// tree-reduced with shuffles; lane 0 applies alpha / beta.  `conj` multiplies by the conjugated matrix
// value, which turns the stored transpose into the conjugate transpose and a symmetric / hermitian expansion
// into its conjugate (op table in complex_api.cpp).  The reference computes these products with its
// vectorised KT kernels (level2/aoclsparse_csrmv_kt.cpp:30-329) whose lane assignment is an x86 register
// width: parity is the forward-error bound, not the bit pattern.  HBM-bound like the real kernels
// (16 + 4 B per non-zero for double complex); not tuned further.
#include "internal.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace mi355
{

namespace
{

template <typename R>
__device__ __forceinline__ R c_fma(R a, R b, R c);
template <>
__device__ __forceinline__ double c_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
template <>
__device__ __forceinline__ float c_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}

// acc += a * b
template <typename R>
__device__ __forceinline__ void c_mac(cplx<R> &acc, cplx<R> a, cplx<R> b)
{
    acc.re = c_fma(a.re, b.re, acc.re);
    acc.re = c_fma(-a.im, b.im, acc.re);
    acc.im = c_fma(a.re, b.im, acc.im);
    acc.im = c_fma(a.im, b.re, acc.im);
}
template <typename R>
__device__ __forceinline__ cplx<R> c_mul(cplx<R> a, cplx<R> b)
{
    cplx<R> r(R(0), R(0));
    c_mac(r, a, b);
    return r;
}

template <typename R, int G>
__global__ __launch_bounds__(256) void cspmv_kernel(int base, bool conj, cplx<R> alpha, aoclsparse_int m,
                                                    const cplx<R> *__restrict__ val,
                                                    const aoclsparse_int *__restrict__ col,
                                                    const aoclsparse_int *__restrict__ row_ptr,
                                                    const cplx<R> *__restrict__ x, cplx<R> beta,
                                                    cplx<R> *__restrict__ y)
{
    const int lane = threadIdx.x % G;
    const int row  = (int)(((long long)blockIdx.x * 256 + threadIdx.x) / G);
    if(row >= m)
        return; // whole groups leave together (256 % G == 0)
    const int s = row_ptr[row] - base, e = row_ptr[row + 1] - base;
    cplx<R>   acc(R(0), R(0));
    for(int p = s + lane; p < e; p += G)
    {
        cplx<R> a = val[p];
        if(conj)
            a.im = -a.im;
        c_mac(acc, a, x[col[p] - base]);
    }
#pragma unroll
    for(int o = G / 2; o > 0; o >>= 1)
    {
        acc.re += __shfl_down(acc.re, o, G);
        acc.im += __shfl_down(acc.im, o, G);
    }
    if(lane == 0)
    {
        cplx<R> r = acc;
        if(!(alpha.re == R(1) && alpha.im == R(0)))
            r = c_mul(alpha, acc);
        if(!(beta.re == R(0) && beta.im == R(0))) // beta == 0 never reads y
            c_mac(r, beta, y[row]);
        y[row] = r;
    }
}

template <typename R>
__global__ void cscale_kernel(cplx<R> *y, aoclsparse_int n, cplx<R> beta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        y[i] = (beta.re == R(0) && beta.im == R(0)) ? cplx<R>(R(0), R(0)) : c_mul(beta, y[i]);
}

// C = alpha op(A) B + beta C, dense B / C row- or column-major: one lane per output element, lanes of a workgroup
// along the contiguous direction of C (columns for row-major, rows for column-major).  readc: C is read and multiplied by
// beta (always for beta != 0; for beta == 0 unless the overwrite mode is on -- csrmm_reads_c, as for the real types: the
// reference's kernels compute beta * C for every beta).
template <typename R, bool COLMAJ>
__global__ __launch_bounds__(256) void ccsrmm_kernel(int base, bool conj, cplx<R> alpha, aoclsparse_int m,
                                                     const cplx<R> *__restrict__ val,
                                                     const aoclsparse_int *__restrict__ col,
                                                     const aoclsparse_int *__restrict__ row_ptr,
                                                     const cplx<R> *__restrict__ B, aoclsparse_int n,
                                                     aoclsparse_int ldb, cplx<R> beta, cplx<R> *__restrict__ C,
                                                     aoclsparse_int ldc, bool readc)
{
    const int i = COLMAJ ? blockIdx.x * blockDim.x + threadIdx.x : blockIdx.x * blockDim.y + threadIdx.y;
    const int j = COLMAJ ? blockIdx.y * blockDim.y + threadIdx.y : blockIdx.y * blockDim.x + threadIdx.x;
    if(i >= m || j >= n)
        return;
    cplx<R> acc(R(0), R(0));
    for(int p = row_ptr[i] - base; p < row_ptr[i + 1] - base; p++)
    {
        cplx<R> a = val[p];
        if(conj)
            a.im = -a.im;
        const size_t k = (size_t)(col[p] - base);
        c_mac(acc, a, COLMAJ ? B[k + (size_t)j * ldb] : B[k * ldb + j]);
    }
    cplx<R> *cp = COLMAJ ? C + i + (size_t)j * ldc : C + (size_t)i * ldc + j;
    cplx<R>  r  = c_mul(alpha, acc);
    if(readc)
        c_mac(r, beta, *cp);
    *cp = r;
}

template <typename R, bool COLMAJ>
__global__ void cscale_dense_kernel(cplx<R> *C, aoclsparse_int inner, aoclsparse_int outer, aoclsparse_int ld,
                                    cplx<R> beta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y;
    if(i < inner && o < outer)
    {
        cplx<R> *cp = C + (size_t)o * ld + i;
        *cp         = (beta.re == R(0) && beta.im == R(0)) ? cplx<R>(R(0), R(0)) : c_mul(beta, *cp);
    }
}


// ---- complex triangular solve --------------------------------------------------------------------------
// Same level-ordered layout as the real solve (trsv_kernels.hip: position k holds row rowmap[k], pind = positions,
// entries in the reference's chain order).  One lane per row: xi = alpha*b_i; xi -= a_ij * x_j ...; xi /= d_i --
// the order of ref_trsv_l / _u / _lth / _uth (level2/aoclsparse_trsv_kr.hpp:38-222) with complex operands.  The
// reference's complex multiply / divide are whatever std::complex compiles to, so parity here is the forward-error
// bound.  Column c of a multi-RHS solve (blockIdx.y) lives at b + c*b_off with element stride incb.
struct CRhsGeom
{
    long long b_off, x_off;
    int       incb, incx;
};

template <typename R>
__device__ __forceinline__ cplx<R> ctrsv_row(int k, const aoclsparse_int *__restrict__ rowmap,
                                             const aoclsparse_int *__restrict__ pptr,
                                             const aoclsparse_int *__restrict__ pind, const cplx<R> *__restrict__ pval,
                                             const cplx<R> *__restrict__ diag, const cplx<R> *__restrict__ b,
                                             const cplx<R> *xp, cplx<R> alpha, int unit, int conj_diag, int incb, int &row)
{
    const int i = rowmap[k];
    row         = i;
    cplx<R> xi  = c_mul(alpha, b[(size_t)i * incb]);
    for(int p = pptr[k], e = pptr[k + 1]; p < e; p++)
    {
        const cplx<R> a = pval[p], xj = xp[pind[p]];
        c_mac(xi, cplx<R>(-a.re, -a.im), xj);
    }
    if(!unit)
    {
        cplx<R> d = diag[i];
        if(conj_diag)
            d.im = -d.im;
        const R den = c_fma(d.re, d.re, d.im * d.im);
        const R re  = c_fma(xi.re, d.re, xi.im * d.im) / den;
        const R im  = c_fma(xi.im, d.re, -(xi.re * d.im)) / den;
        xi          = cplx<R>(re, im);
    }
    return xi;
}

template <typename R>
__global__ void ctrsv_level_kernel(aoclsparse_int first, aoclsparse_int count, aoclsparse_int m,
                                   const aoclsparse_int *__restrict__ rowmap, const aoclsparse_int *__restrict__ pptr,
                                   const aoclsparse_int *__restrict__ pind, const cplx<R> *__restrict__ pval,
                                   const cplx<R> *__restrict__ diag, const cplx<R> *__restrict__ b, cplx<R> *xp,
                                   cplx<R> *x, cplx<R> alpha, int unit, int conj_diag, CRhsGeom g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= count)
        return;
    const int c = blockIdx.y;
    b += c * g.b_off;
    x += c * g.x_off;
    xp += (size_t)c * m;
    int           i;
    const cplx<R> xi = ctrsv_row<R>(first + t, rowmap, pptr, pind, pval, diag, b, xp, alpha, unit, conj_diag, g.incb, i);
    xp[first + t]         = xi;
    x[(size_t)i * g.incx] = xi;
}

// a run of narrow levels [l0, l1) inside ONE workgroup (a lane per row, a workgroup barrier per level): what later
// levels read was written by this workgroup, i.e. through this CU's own L1
template <typename R>
__global__ __launch_bounds__(1024) void ctrsv_run_kernel(aoclsparse_int l0, aoclsparse_int l1, aoclsparse_int m,
                                                         const aoclsparse_int *__restrict__ levels,
                                                         const aoclsparse_int *__restrict__ rowmap,
                                                         const aoclsparse_int *__restrict__ pptr,
                                                         const aoclsparse_int *__restrict__ pind,
                                                         const cplx<R> *__restrict__ pval, const cplx<R> *__restrict__ diag,
                                                         const cplx<R> *__restrict__ b, cplx<R> *xp, cplx<R> *x,
                                                         cplx<R> alpha, int unit, int conj_diag, CRhsGeom g)
{
    const int c = blockIdx.y;
    b += c * g.b_off;
    x += c * g.x_off;
    xp += (size_t)c * m;
    for(int l = l0; l < l1; l++)
    {
        const int first = levels[l], count = levels[l + 1] - first;
        if((int)threadIdx.x < count)
        {
            int           i;
            const cplx<R> xi = ctrsv_row<R>(first + threadIdx.x, rowmap, pptr, pind, pval, diag, b, xp, alpha, unit,
                                            conj_diag, g.incb, i);
            xp[first + threadIdx.x] = xi;
            x[(size_t)i * g.incx]   = xi;
        }
        __threadfence_block();
        __syncthreads();
    }
}


// ---- conjugated dot product d = sum conj(x_i) * y_i (level1/aoclsparse_dense_dot.hpp:36-49), fixed two-stage tree
constexpr int CDOT_BLOCKS = 1024;
template <typename R>
__device__ __forceinline__ cplx<R> cdot_block_reduce(cplx<R> acc, cplx<R> *sh)
{
    for(int off = 32; off > 0; off >>= 1)
    {
        acc.re += __shfl_down(acc.re, off, 64);
        acc.im += __shfl_down(acc.im, off, 64);
    }
    if((threadIdx.x & 63) == 0)
        sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    return cplx<R>((sh[0].re + sh[1].re) + (sh[2].re + sh[3].re), (sh[0].im + sh[1].im) + (sh[2].im + sh[3].im));
}
template <typename R>
__global__ __launch_bounds__(256) void cdot_partial_kernel(const cplx<R> *__restrict__ x, const cplx<R> *__restrict__ y,
                                                           aoclsparse_int n, cplx<R> *partial, bool conj_x)
{
    __shared__ cplx<R> sh[4];
    cplx<R>            acc(R(0), R(0));
    for(long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    {
        const cplx<R> a = x[i];
        c_mac(acc, cplx<R>(a.re, conj_x ? -a.im : a.im), y[i]);
    }
    const cplx<R> r = cdot_block_reduce(acc, sh);
    if(threadIdx.x == 0)
        partial[blockIdx.x] = r;
}
template <typename R>
__global__ __launch_bounds__(256) void cdot_final_kernel(const cplx<R> *__restrict__ partial, int count, cplx<R> *d)
{
    __shared__ cplx<R> sh[4];
    cplx<R>            acc(R(0), R(0));
    for(int i = threadIdx.x; i < count; i += 256)
        acc.re += partial[i].re, acc.im += partial[i].im;
    const cplx<R> r = cdot_block_reduce(acc, sh);
    if(threadIdx.x == 0)
        *d = r;
}


// w = x - y (the residual updates of the complex symmetric Gauss-Seidel sweep)
template <typename R>
__global__ void cdiff_kernel(aoclsparse_int n, const cplx<R> *x, const cplx<R> *y, cplx<R> *w)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        w[i] = cplx<R>(x[i].re - y[i].re, x[i].im - y[i].im);
}


// w = a x + b y with complex scalars (the vector steps of the complex CG / GMRES; y may alias w, x may be null when a = 0)
template <typename R>
__global__ void caxpby_kernel(aoclsparse_int n, cplx<R> a, const cplx<R> *x, cplx<R> b, const cplx<R> *y, cplx<R> *w)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
    {
        cplx<R> r(R(0), R(0));
        if(x)
            c_mac(r, a, x[i]);
        if(y)
            c_mac(r, b, y[i]);
        w[i] = r;
    }
}


// y[i] *= d[i] (the diagonal scale between the two sweeps of the SymGS preconditioner)
template <typename R>
__global__ void cvec_mul_kernel(aoclsparse_int n, const cplx<R> *d, cplx<R> *y)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        y[i] = c_mul(d[i], y[i]);
}

} // namespace

template <typename R>
aoclsparse_status launch_ccsrmm(hipStream_t s, aoclsparse_order order, int base, bool conj, cplx<R> alpha,
                                aoclsparse_int m, const cplx<R> *val, const aoclsparse_int *col,
                                const aoclsparse_int *row_ptr, const cplx<R> *B, aoclsparse_int n, aoclsparse_int ldb,
                                cplx<R> beta, cplx<R> *C, aoclsparse_int ldc)
{
    if(m <= 0 || n <= 0)
        return aoclsparse_status_success;
    const bool readc = csrmm_reads_c(!(beta.re == R(0) && beta.im == R(0)));
    if(order == aoclsparse_order_column)
    {
        // rows on grid.x (no 65535 limit), columns on grid.y in chunks of 4 -> loop over y chunks if needed
        for(aoclsparse_int j0 = 0; j0 < n; j0 += 65535 * 4)
        {
            const aoclsparse_int nj = std::min<aoclsparse_int>(n - j0, 65535 * 4);
            hipLaunchKernelGGL((ccsrmm_kernel<R, true>), dim3((m + 63) / 64, (nj + 3) / 4), dim3(64, 4), 0, s, base,
                               conj, alpha, m, val, col, row_ptr, B + (size_t)j0 * ldb, nj, ldb, beta,
                               C + (size_t)j0 * ldc, ldc, readc);
        }
    }
    else
    {
        const int tx = n >= 64 ? 64 : (n >= 16 ? 16 : 4), ty = 256 / tx;
        for(aoclsparse_int j0 = 0; j0 < n; j0 += 65535 * tx)
        {
            const aoclsparse_int nj = std::min<aoclsparse_int>(n - j0, 65535 * tx);
            hipLaunchKernelGGL((ccsrmm_kernel<R, false>), dim3((m + ty - 1) / ty, (nj + tx - 1) / tx), dim3(tx, ty), 0,
                               s, base, conj, alpha, m, val, col, row_ptr, B + j0, nj, ldb, beta, C + j0, ldc, readc);
        }
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename R>
aoclsparse_status launch_cscale_dense(hipStream_t s, aoclsparse_order order, cplx<R> *C, aoclsparse_int m,
                                      aoclsparse_int n, aoclsparse_int ld, cplx<R> beta)
{
    const aoclsparse_int inner = order == aoclsparse_order_column ? m : n, outer = order == aoclsparse_order_column ? n : m;
    if(inner <= 0 || outer <= 0)
        return aoclsparse_status_success;
    for(aoclsparse_int o0 = 0; o0 < outer; o0 += 65535)
    {
        const aoclsparse_int no = std::min<aoclsparse_int>(outer - o0, 65535);
        hipLaunchKernelGGL((cscale_dense_kernel<R, true>), dim3((inner + 255) / 256, no), dim3(256), 0, s,
                           C + (size_t)o0 * ld, inner, no, ld, beta);
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename R>
aoclsparse_status launch_cspmv(hipStream_t s, int base, bool conj, cplx<R> alpha, aoclsparse_int m, aoclsparse_int nnz,
                               const cplx<R> *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                               const cplx<R> *x, cplx<R> beta, cplx<R> *y)
{
    if(m <= 0)
        return aoclsparse_status_success;
    const long long mean = nnz / (long long)m;
    auto            go   = [&](auto gtag) {
        constexpr int G    = decltype(gtag)::value;
        const long long nb = ((long long)m * G + 255) / 256;
        hipLaunchKernelGGL((cspmv_kernel<R, G>), dim3((unsigned)nb), dim3(256), 0, s, base, conj, alpha, m, val, col,
                           row_ptr, x, beta, y);
    };
    if(mean <= 6)
        go(std::integral_constant<int, 4>{});
    else if(mean <= 24)
        go(std::integral_constant<int, 8>{});
    else if(mean <= 96)
        go(std::integral_constant<int, 16>{});
    else
        go(std::integral_constant<int, 64>{});
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename R>
aoclsparse_status launch_cscale(hipStream_t s, cplx<R> *y, aoclsparse_int n, cplx<R> beta)
{
    if(n > 0)
        hipLaunchKernelGGL((cscale_kernel<R>), dim3((n + 255) / 256), dim3(256), 0, s, y, n, beta);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}


template <typename R>
aoclsparse_status launch_ctrsv(hipStream_t s, bool unit, bool conj_diag, cplx<R> alpha, aoclsparse_int m,
                               const TrsvPlan &plan, const cplx<R> *diag, const cplx<R> *b, cplx<R> *x, cplx<R> *xp,
                               aoclsparse_int nrhs, long long b_off, aoclsparse_int incb, long long x_off,
                               aoclsparse_int incx)
{
    if(m <= 0 || nrhs <= 0)
        return aoclsparse_status_success;
    const aoclsparse_int *rowmap = plan.rowmap.as<aoclsparse_int>(), *pptr = plan.pptr.as<aoclsparse_int>();
    const aoclsparse_int *pind = plan.pind.as<aoclsparse_int>(), *levels = plan.levels.as<aoclsparse_int>();
    const cplx<R>        *pval = plan.pval.as<cplx<R>>();
    const CRhsGeom        g{b_off, x_off, incb, incx};
    for(aoclsparse_int c0 = 0; c0 < nrhs; c0 += 65535)
    {
        const int nc = nrhs - c0 < 65535 ? nrhs - c0 : 65535;
        for(const TrsvSegment &sg : plan.segments)
        {
            if(sg.narrow)
            {
                hipLaunchKernelGGL((ctrsv_run_kernel<R>), dim3(1, nc), dim3(TRSV_NARROW), 0, s, sg.l0, sg.l1, m, levels,
                                   rowmap, pptr, pind, pval, diag, b + c0 * b_off, xp + (size_t)c0 * m, x + c0 * x_off,
                                   alpha, (int)unit, (int)conj_diag, g);
                continue;
            }
            for(aoclsparse_int l = sg.l0; l < sg.l1; l++)
            {
                const aoclsparse_int first = plan.level_ptr[l], count = plan.level_ptr[l + 1] - first;
                const int            bs = count >= 256 ? 256 : 64;
                hipLaunchKernelGGL((ctrsv_level_kernel<R>), dim3((count + bs - 1) / bs, nc), dim3(bs), 0, s, first,
                                   count, m, rowmap, pptr, pind, pval, diag, b + c0 * b_off, xp + (size_t)c0 * m,
                                   x + c0 * x_off, alpha, (int)unit, (int)conj_diag, g);
            }
        }
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_ctrsv<float>(hipStream_t, bool, bool, cfloat, aoclsparse_int, const TrsvPlan &,
                                               const cfloat *, const cfloat *, cfloat *, cfloat *, aoclsparse_int,
                                               long long, aoclsparse_int, long long, aoclsparse_int);
template aoclsparse_status launch_ctrsv<double>(hipStream_t, bool, bool, cdouble, aoclsparse_int, const TrsvPlan &,
                                                const cdouble *, const cdouble *, cdouble *, cdouble *, aoclsparse_int,
                                                long long, aoclsparse_int, long long, aoclsparse_int);

template <typename R>
aoclsparse_status launch_cdot(hipStream_t s, aoclsparse_int n, const cplx<R> *x, const cplx<R> *y, cplx<R> *partial,
                              cplx<R> *d, bool conj_x)
{
    const int blocks = n <= 0 ? 1 : (int)std::min<long long>(CDOT_BLOCKS, ((long long)n + 255) / 256);
    hipLaunchKernelGGL((cdot_partial_kernel<R>), dim3(blocks), dim3(256), 0, s, x, y, n, partial, conj_x);
    hipLaunchKernelGGL((cdot_final_kernel<R>), dim3(1), dim3(256), 0, s, partial, blocks, d);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template aoclsparse_status launch_cdot<float>(hipStream_t, aoclsparse_int, const cfloat *, const cfloat *, cfloat *, cfloat *,
                                              bool);
template aoclsparse_status launch_cdot<double>(hipStream_t, aoclsparse_int, const cdouble *, const cdouble *, cdouble *,
                                               cdouble *, bool);

template <typename R>
aoclsparse_status launch_cdiff(hipStream_t s, aoclsparse_int n, const cplx<R> *x, const cplx<R> *y, cplx<R> *w)
{
    if(n > 0)
        hipLaunchKernelGGL((cdiff_kernel<R>), dim3((n + 255) / 256), dim3(256), 0, s, n, x, y, w);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template aoclsparse_status launch_cdiff<float>(hipStream_t, aoclsparse_int, const cfloat *, const cfloat *, cfloat *);
template aoclsparse_status launch_cdiff<double>(hipStream_t, aoclsparse_int, const cdouble *, const cdouble *, cdouble *);

template <typename R>
aoclsparse_status launch_caxpby(hipStream_t s, aoclsparse_int n, cplx<R> a, const cplx<R> *x, cplx<R> b, const cplx<R> *y,
                                cplx<R> *w)
{
    if(n > 0)
        hipLaunchKernelGGL((caxpby_kernel<R>), dim3((n + 255) / 256), dim3(256), 0, s, n, a, x, b, y, w);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template aoclsparse_status launch_caxpby<float>(hipStream_t, aoclsparse_int, cfloat, const cfloat *, cfloat, const cfloat *,
                                                cfloat *);
template aoclsparse_status launch_caxpby<double>(hipStream_t, aoclsparse_int, cdouble, const cdouble *, cdouble,
                                                 const cdouble *, cdouble *);

template <typename R>
aoclsparse_status launch_cvec_mul(hipStream_t s, aoclsparse_int n, const cplx<R> *d, cplx<R> *y)
{
    if(n > 0)
        hipLaunchKernelGGL((cvec_mul_kernel<R>), dim3((n + 255) / 256), dim3(256), 0, s, n, d, y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template aoclsparse_status launch_cvec_mul<float>(hipStream_t, aoclsparse_int, const cfloat *, cfloat *);
template aoclsparse_status launch_cvec_mul<double>(hipStream_t, aoclsparse_int, const cdouble *, cdouble *);

template aoclsparse_status launch_cspmv<float>(hipStream_t, int, bool, cfloat, aoclsparse_int, aoclsparse_int,
                                               const cfloat *, const aoclsparse_int *, const aoclsparse_int *,
                                               const cfloat *, cfloat, cfloat *);
template aoclsparse_status launch_cspmv<double>(hipStream_t, int, bool, cdouble, aoclsparse_int, aoclsparse_int,
                                                const cdouble *, const aoclsparse_int *, const aoclsparse_int *,
                                                const cdouble *, cdouble, cdouble *);
template aoclsparse_status launch_ccsrmm<float>(hipStream_t, aoclsparse_order, int, bool, cfloat, aoclsparse_int,
                                                const cfloat *, const aoclsparse_int *, const aoclsparse_int *,
                                                const cfloat *, aoclsparse_int, aoclsparse_int, cfloat, cfloat *,
                                                aoclsparse_int);
template aoclsparse_status launch_ccsrmm<double>(hipStream_t, aoclsparse_order, int, bool, cdouble, aoclsparse_int,
                                                 const cdouble *, const aoclsparse_int *, const aoclsparse_int *,
                                                 const cdouble *, aoclsparse_int, aoclsparse_int, cdouble, cdouble *,
                                                 aoclsparse_int);
template aoclsparse_status launch_cscale_dense<float>(hipStream_t, aoclsparse_order, cfloat *, aoclsparse_int,
                                                      aoclsparse_int, aoclsparse_int, cfloat);
template aoclsparse_status launch_cscale_dense<double>(hipStream_t, aoclsparse_order, cdouble *, aoclsparse_int,
                                                       aoclsparse_int, aoclsparse_int, cdouble);
template aoclsparse_status launch_cscale<float>(hipStream_t, cfloat *, aoclsparse_int, cfloat);
template aoclsparse_status launch_cscale<double>(hipStream_t, cdouble *, aoclsparse_int, cdouble);

} // namespace mi355
