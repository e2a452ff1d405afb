// itsol_kernels.hip -- the dense-vector steps of the iterative solvers (CG, restarted GMRES) as HIP
// kernels, so that between two SpMVs / preconditioner applications the iterates never leave the GPU.
// Reference: solvers/aoclsparse_itsol_functions.hpp:632-875 (CG), :910-1367 (GMRES); there these are
// plain loops and AOCL-BLAS level-1 calls (nrm2, dot, axpby, scal).  All HBM-bound streaming kernels:
// bytes = (vectors read + vectors written) * n * sizeof(T); reductions are fixed two-stage trees
// (deterministic for a given n).
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

namespace
{

constexpr int RED_BLOCKS = 512; // partial sums per reduced quantity

__device__ __forceinline__ double v_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float v_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}

template <typename T>
__device__ __forceinline__ T block_reduce_256(T v, T *sh)
{
    for(int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o);
    const int w = threadIdx.x >> 6;
    if((threadIdx.x & 63) == 0)
        sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]); // valid on every lane
}

// r = -b, p = x  (CG start, :676-680)
template <typename T>
__global__ void cg_init_kernel(aoclsparse_int n, const T *b, const T *x, T *r, T *p)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        r[i] = -b[i], p[i] = x[i];
}

// dst (op)= src
template <typename T, int OP> // 0 copy, 1 add, 2 multiply
__global__ void ew_kernel(aoclsparse_int n, const T *src, T *dst)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        dst[i] = OP == 0 ? src[i] : (OP == 1 ? dst[i] + src[i] : dst[i] * src[i]);
}

template <typename T>
__global__ void fill_kernel(aoclsparse_int n, T *dst, T value)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        dst[i] = value;
}

// p = beta*p - z  (:805-806; one FMA under the reference's -ffp-contract=fast)
template <typename T>
__global__ void cg_dir_kernel(aoclsparse_int n, T beta, T *p, const T *z)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        p[i] = v_fma(beta, p[i], -z[i]);
}

// x += alpha p; r += alpha q; partial[b] = sum over the block's elements of r_new^2  (:838-845)
template <typename T>
__global__ __launch_bounds__(256) void cg_step_kernel(aoclsparse_int n, T alpha, const T *p, const T *q, T *x, T *r,
                                                       T *partial)
{
    __shared__ T sh[4];
    T            acc = T(0);
    for(aoclsparse_int i = blockIdx.x * 256 + threadIdx.x; i < n; i += (aoclsparse_int)gridDim.x * 256)
    {
        x[i]       = v_fma(alpha, p[i], x[i]);
        const T rn = v_fma(alpha, q[i], r[i]);
        r[i]       = rn;
        acc        = v_fma(rn, rn, acc);
    }
    const T s = block_reduce_256(acc, sh);
    if(threadIdx.x == 0)
        partial[blockIdx.x] = s;
}

// the same step with alpha = rz / pq formed on the device from the p.q the previous launch left in *pq_dev, so the
// host does not have to wait for that dot product; a p.q the reference refuses (<= 0.02 eps or 0, :815-822) makes
// the step a no-op and the host reports the error after the one synchronisation of the iteration
template <typename T>
__global__ __launch_bounds__(256) void cg_step_dev_kernel(aoclsparse_int n, T rz, const T *__restrict__ pq_dev, T tiny,
                                                           const T *p, const T *q, T *x, T *r, T *partial)
{
    __shared__ T sh[4];
    const T      pq    = *pq_dev;
    const T      alpha = (pq <= tiny || pq == T(0)) ? T(0) : rz / pq;
    T            acc   = T(0);
    for(aoclsparse_int i = blockIdx.x * 256 + threadIdx.x; i < n; i += (aoclsparse_int)gridDim.x * 256)
    {
        T xn = x[i], rn = r[i];
        if(alpha != T(0))
        {
            xn   = v_fma(alpha, p[i], xn);
            rn   = v_fma(alpha, q[i], rn);
            x[i] = xn, r[i] = rn;
        }
        acc = v_fma(rn, rn, acc);
    }
    const T s = block_reduce_256(acc, sh);
    if(threadIdx.x == 0)
        partial[blockIdx.x] = s;
}

// partial[y][b] = sum_i w[i] * V[y*ld + i] over block b's elements
template <typename T>
__global__ __launch_bounds__(256) void multidot_kernel(aoclsparse_int n, const T *V, long long ld, const T *w,
                                                        T *partial)
{
    __shared__ T sh[4];
    const T     *v   = V + (long long)blockIdx.y * ld;
    T            acc = T(0);
    for(aoclsparse_int i = blockIdx.x * 256 + threadIdx.x; i < n; i += (aoclsparse_int)gridDim.x * 256)
        acc = v_fma(w[i], v[i], acc);
    const T s = block_reduce_256(acc, sh);
    if(threadIdx.x == 0)
        partial[(size_t)blockIdx.y * RED_BLOCKS + blockIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_final_kernel(int nb, const T *partial, T *out)
{
    __shared__ T sh[4];
    const T     *p   = partial + (size_t)blockIdx.x * RED_BLOCKS;
    T            acc = T(0);
    for(int i = threadIdx.x; i < nb; i += 256)
        acc += p[i];
    const T s = block_reduce_256(acc, sh);
    if(threadIdx.x == 0)
        out[blockIdx.x] = s;
}

// SIGN = -1: w[i] -= sum_t c[t] V_t[i] (classical Gram-Schmidt step, :1101-1113)
// SIGN = +1: w[i] += sum_t c[t] V_t[i] (GMRES solution update, :1247-1273)
template <typename T, int SIGN>
__global__ void lincomb_kernel(aoclsparse_int n, int k, const T *c, const T *V, long long ld, T *w)
{
    const aoclsparse_int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n)
        return;
    T acc = T(0);
    for(int t = 0; t < k; t++)
        acc = v_fma(c[t], V[(long long)t * ld + i], acc);
    w[i] = SIGN < 0 ? w[i] - acc : w[i] + acc;
}

inline dim3 grid1(aoclsparse_int n)
{
    return dim3((unsigned)((n + 255) / 256));
}
inline int red_blocks(aoclsparse_int n)
{
    const long long nb = ((long long)n + 255) / 256;
    return (int)(nb < RED_BLOCKS ? (nb > 0 ? nb : 1) : RED_BLOCKS);
}

} // namespace

int vec_reduce_scratch_elems(int k)
{
    return RED_BLOCKS * (k > 0 ? k : 1);
}

#define MI355_VEC_DONE()                \
    MI355_HIP_TRY(hipGetLastError());   \
    return aoclsparse_status_success

template <typename T>
aoclsparse_status launch_cg_init(hipStream_t s, aoclsparse_int n, const T *b, const T *x, T *r, T *p)
{
    if(n > 0)
        hipLaunchKernelGGL((cg_init_kernel<T>), grid1(n), dim3(256), 0, s, n, b, x, r, p);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_vec_copy(hipStream_t s, aoclsparse_int n, const T *src, T *dst)
{
    if(n > 0)
        hipLaunchKernelGGL((ew_kernel<T, 0>), grid1(n), dim3(256), 0, s, n, src, dst);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_vec_add(hipStream_t s, aoclsparse_int n, const T *src, T *dst)
{
    if(n > 0)
        hipLaunchKernelGGL((ew_kernel<T, 1>), grid1(n), dim3(256), 0, s, n, src, dst);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_vec_mul(hipStream_t s, aoclsparse_int n, const T *src, T *dst)
{
    if(n > 0)
        hipLaunchKernelGGL((ew_kernel<T, 2>), grid1(n), dim3(256), 0, s, n, src, dst);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_vec_fill(hipStream_t s, aoclsparse_int n, T *dst, T value)
{
    if(n > 0)
        hipLaunchKernelGGL((fill_kernel<T>), grid1(n), dim3(256), 0, s, n, dst, value);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_cg_direction(hipStream_t s, aoclsparse_int n, T beta, T *p, const T *z)
{
    if(n > 0)
        hipLaunchKernelGGL((cg_dir_kernel<T>), grid1(n), dim3(256), 0, s, n, beta, p, z);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_cg_step(hipStream_t s, aoclsparse_int n, T alpha, const T *p, const T *q, T *x, T *r,
                                 T *partial, T *rr)
{
    const int nb = red_blocks(n);
    hipLaunchKernelGGL((cg_step_kernel<T>), dim3(nb), dim3(256), 0, s, n, alpha, p, q, x, r, partial);
    hipLaunchKernelGGL((reduce_final_kernel<T>), dim3(1), dim3(256), 0, s, nb, partial, rr);
    MI355_VEC_DONE();
}
// out2[0] = r_new . r_new, out2[1] = p.q (both stay on the device; partial holds 2 * RED_BLOCKS elements)
template <typename T>
aoclsparse_status launch_cg_step_dev(hipStream_t s, aoclsparse_int n, T rz, T tiny, const T *p, const T *q, T *x, T *r,
                                     T *partial, T *out2)
{
    const int nb = red_blocks(n);
    hipLaunchKernelGGL((multidot_kernel<T>), dim3(nb, 1), dim3(256), 0, s, n, q, 0LL, p, partial + RED_BLOCKS);
    hipLaunchKernelGGL((reduce_final_kernel<T>), dim3(1), dim3(256), 0, s, nb, partial + RED_BLOCKS, out2 + 1);
    hipLaunchKernelGGL((cg_step_dev_kernel<T>), dim3(nb), dim3(256), 0, s, n, rz, out2 + 1, tiny, p, q, x, r, partial);
    hipLaunchKernelGGL((reduce_final_kernel<T>), dim3(1), dim3(256), 0, s, nb, partial, out2);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_multidot(hipStream_t s, aoclsparse_int n, int k, const T *V, long long ld, const T *w,
                                  T *partial, T *out)
{
    if(k <= 0)
        return aoclsparse_status_success;
    const int nb = red_blocks(n);
    hipLaunchKernelGGL((multidot_kernel<T>), dim3(nb, k), dim3(256), 0, s, n, V, ld, w, partial);
    hipLaunchKernelGGL((reduce_final_kernel<T>), dim3(k), dim3(256), 0, s, nb, partial, out);
    MI355_VEC_DONE();
}
template <typename T>
aoclsparse_status launch_lincomb(hipStream_t s, int sign, aoclsparse_int n, int k, const T *c, const T *V,
                                 long long ld, T *w)
{
    if(n > 0 && k > 0)
    {
        if(sign < 0)
            hipLaunchKernelGGL((lincomb_kernel<T, -1>), grid1(n), dim3(256), 0, s, n, k, c, V, ld, w);
        else
            hipLaunchKernelGGL((lincomb_kernel<T, 1>), grid1(n), dim3(256), 0, s, n, k, c, V, ld, w);
    }
    MI355_VEC_DONE();
}

#define MI355_VEC_INSTANTIATE(T)                                                                                    \
    template aoclsparse_status launch_cg_init<T>(hipStream_t, aoclsparse_int, const T *, const T *, T *, T *);      \
    template aoclsparse_status launch_vec_copy<T>(hipStream_t, aoclsparse_int, const T *, T *);                     \
    template aoclsparse_status launch_vec_add<T>(hipStream_t, aoclsparse_int, const T *, T *);                      \
    template aoclsparse_status launch_vec_mul<T>(hipStream_t, aoclsparse_int, const T *, T *);                      \
    template aoclsparse_status launch_vec_fill<T>(hipStream_t, aoclsparse_int, T *, T);                             \
    template aoclsparse_status launch_cg_direction<T>(hipStream_t, aoclsparse_int, T, T *, const T *);              \
    template aoclsparse_status launch_cg_step<T>(hipStream_t, aoclsparse_int, T, const T *, const T *, T *, T *,    \
                                                 T *, T *);                                                         \
    template aoclsparse_status launch_cg_step_dev<T>(hipStream_t, aoclsparse_int, T, T, const T *, const T *, T *,  \
                                                     T *, T *, T *);                                                \
    template aoclsparse_status launch_multidot<T>(hipStream_t, aoclsparse_int, int, const T *, long long,           \
                                                  const T *, T *, T *);                                             \
    template aoclsparse_status launch_lincomb<T>(hipStream_t, int, aoclsparse_int, int, const T *, const T *,       \
                                                 long long, T *);
MI355_VEC_INSTANTIATE(double)
MI355_VEC_INSTANTIATE(float)

} // namespace mi355
