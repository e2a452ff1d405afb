// ilu_kernels.hip -- level-scheduled ILU(0) factorisation on the user's CSR pattern, in place in a device
// copy of the values.  Reference: solvers/aoclsparse_ilu0.hpp:34-111 (serial IKJ).
//
// Row i depends on the finished rows k < i of its own pattern, so rows are processed level by level (one
// launch per dependency level, rows of a level in parallel), one wavefront per row:
//   * the row's columns and values sit in LDS while it is eliminated (the wavefront both updates and re-reads
//     them; LDS operations of one wave are ordered);
//   * the k-loop walks the row's lower entries in stored order exactly like the reference; for each of them
//     the 64 lanes take the entries of row k's upper part and apply  a_iw = fma(-l_ik, a_kw, a_iw)  to the
//     matching position w of row i -- distinct entries of row k hit distinct positions, and the k steps are
//     sequential, so every a_iw sees its updates in the reference's order: results are bit-identical;
//   * a (near-)zero pivot or a row without its diagonal right after the lower part sets the error word.
// One-time analysis work per handle; bound by the dependency chain (levels x launch latency), not by HBM.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

namespace
{

template <typename T>
__device__ __forceinline__ bool ilu_near_zero(T v)
{
    // aoclsparse_is_nearzero, extra/aoclsparse_utils.hpp:598-613
    return fabs((double)v) <= 1e-2 * 2.0 * (sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07);
}

// s - l * a and a / b: one fma / one division for real types (the reference's contracted update, :80-87); for
// complex types component-wise fmas and the textbook quotient (the reference runs std::complex arithmetic there, so
// complex factors carry a tolerance instead of bit-equality)
__device__ __forceinline__ double ilu_nfma(double l, double a, double s)
{
    return fma(-l, a, s);
}
__device__ __forceinline__ float ilu_nfma(float l, float a, float s)
{
    return fmaf(-l, a, s);
}
template <typename R>
__device__ __forceinline__ cplx<R> ilu_nfma(cplx<R> l, cplx<R> a, cplx<R> s)
{
    s.re = ilu_nfma(l.re, a.re, s.re);
    s.re = ilu_nfma(-l.im, a.im, s.re);
    s.im = ilu_nfma(l.re, a.im, s.im);
    s.im = ilu_nfma(l.im, a.re, s.im);
    return s;
}
__device__ __forceinline__ double ilu_div(double a, double b)
{
    return a / b;
}
__device__ __forceinline__ float ilu_div(float a, float b)
{
    return a / b;
}
template <typename R>
__device__ __forceinline__ cplx<R> ilu_div(cplx<R> a, cplx<R> b)
{
    const R den = b.re * b.re + b.im * b.im;
    return cplx<R>((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
template <typename R>
__device__ __forceinline__ bool ilu_near_zero(cplx<R> v)
{
    return hypot((double)v.re, (double)v.im) <= 1e-2 * 2.0 * (sizeof(R) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07);
}

// one wavefront (= one workgroup) per row of the level; dynamic LDS: maxlen columns + maxlen values
template <typename T>
__global__ __launch_bounds__(64) void ilu0_level_kernel(int base, const aoclsparse_int *__restrict__ rows,
                                                        const aoclsparse_int *__restrict__ row_ptr,
                                                        const aoclsparse_int *__restrict__ col, T *val,
                                                        aoclsparse_int *__restrict__ diag, int maxlen, int *error)
{
    extern __shared__ unsigned char smem[];
    T              *sv   = reinterpret_cast<T *>(smem);
    aoclsparse_int *sc   = reinterpret_cast<aoclsparse_int *>(smem + sizeof(T) * (size_t)maxlen);
    const int       lane = threadIdx.x;
    const int       i    = rows[blockIdx.x];
    const int       s = row_ptr[i] - base, len = row_ptr[i + 1] - base - s;
    for(int t = lane; t < len; t += 64)
        sv[t] = val[s + t], sc[t] = col[s + t] - base;
    __syncthreads(); // single-wave workgroup: an LDS fence
    int  t = 0, k = -1;
    bool bad = false;
    for(; t < len; t++) // wave-uniform loop: every lane sees the same k, dk, pivot
    {
        k = sc[t];
        if(k >= i)
            break;
        const int dk    = diag[k]; // written by an earlier level (earlier launch)
        const T   pivot = val[dk];
        if(ilu_near_zero(pivot))
        {
            bad = true;
            break;
        }
        const T   lik = ilu_div(sv[t], pivot);
        const int ke  = row_ptr[k + 1] - base;
        for(int j0 = dk + 1; j0 < ke; j0 += 64) // upper part of row k, 64 entries per round
        {
            const int  jj   = j0 + lane;
            const bool have = jj < ke;
            const int  c    = have ? col[jj] - base : -1;
            T          akw  = T(0);
            if(have)
                akw = val[jj];
            for(int w = 0; w < len; w++) // the reference's column -> position map, as a scan of the LDS row
                if(have && sc[w] == c)
                    sv[w] = ilu_nfma(lik, akw, sv[w]);
        }
        if(lane == 0)
            sv[t] = lik;
        __syncthreads();
    }
    if(!bad && (t >= len || k != i || ilu_near_zero(sv[t])))
        bad = true; // no diagonal right after the lower part, or a (near-)zero pivot (:95-101)
    if(bad)
    {
        if(lane == 0)
            atomicExch(error, 1);
        return;
    }
    if(lane == 0)
        diag[i] = s + t;
    for(int w = lane; w < len; w += 64)
        val[s + w] = sv[w];
}

// ---- sync-free variant (real types): ONE launch over all rows in level order ------------------------------------------
// Same row routine, but instead of one launch per dependency level (5,505 launches = 390 ms on the shell-like stand-in,
// more than the CPU's serial loop) a row waits for the rows it needs: diag[k] -- the position of row k's diagonal, which
// the routine writes anyway -- doubles as row k's "finished" flag (-1 until then, set with release semantics after the
// row's values), and row k's values are read with agent-scope loads.  Rows are taken in LEVEL order through an atomic
// ticket, so a wavefront only ever waits for rows of wavefronts that have already started.  A row that fails (zero
// pivot, missing diagonal) still publishes its flag (-2): its dependants fail too instead of waiting forever.
template <typename T>
__global__ __launch_bounds__(64) void ilu0_syncfree_kernel(int base, aoclsparse_int n, const aoclsparse_int *__restrict__ rows,
                                                           const aoclsparse_int *__restrict__ row_ptr,
                                                           const aoclsparse_int *__restrict__ col, T *val, aoclsparse_int *diag,
                                                           int maxlen, int *error, unsigned int *ticket)
{
    extern __shared__ unsigned char smem[];
    T              *sv   = reinterpret_cast<T *>(smem);
    aoclsparse_int *sc   = reinterpret_cast<aoclsparse_int *>(smem + sizeof(T) * (size_t)maxlen);
    const int       lane = threadIdx.x;
    unsigned int    tk   = 0;
    if(lane == 0)
        tk = atomicAdd(ticket, 1u);
    const int pos = __builtin_amdgcn_readfirstlane((int)tk);
    if(pos >= n)
        return;
    const int i = rows[pos];
    const int s = row_ptr[i] - base, len = row_ptr[i + 1] - base - s;
    for(int t = lane; t < len; t += 64)
        sv[t] = val[s + t], sc[t] = col[s + t] - base; // row i's own values: nobody else writes them
    __syncthreads(); // single-wave workgroup: an LDS fence
    int  t = 0, k = -1;
    bool bad = false;
    for(; t < len; t++) // wave-uniform loop: every lane sees the same k, dk, pivot
    {
        k = sc[t];
        if(k >= i)
            break;
        // relaxed polls (an acquire load invalidates the L1 on every look: with thousands of resident wavefronts polling,
        // the first version ran at 2.4 s); row k's values are read with agent-scope loads below, after the flag was seen
        int dk = __hip_atomic_load(&diag[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (rows take their tickets in level order, so the row waited for is always running or finished; the wall-clock
        // bound -- 5 s of the 100 MHz clock -- only turns a lost device into an error instead of a hang)
        unsigned int       spins = 0;
        unsigned long long t0    = 0;
        while(dk == -1)
        {
            __builtin_amdgcn_s_sleep(2);
            dk = __hip_atomic_load(&diag[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if((++spins & 1023u) == 0)
            {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if(t0 == 0)
                    t0 = now;
                else if(now - t0 > 500000000ull)
                    dk = -2;
            }
        }
        if(dk < 0) // row k failed
        {
            bad = true;
            break;
        }
        // the pivot and the first 64 entries of row k's upper part are requested together (both depend on dk only)
        const int ke   = row_ptr[k + 1] - base;
        const int jj0  = dk + 1 + lane;
        const T   pivot = __hip_atomic_load(&val[dk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int       c0   = -1;
        T         akw0 = T(0);
        if(jj0 < ke)
        {
            c0   = col[jj0] - base;
            akw0 = __hip_atomic_load(&val[jj0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if(ilu_near_zero(pivot))
        {
            bad = true;
            break;
        }
        const T lik = ilu_div(sv[t], pivot);
        for(int j0 = dk + 1; j0 < ke; j0 += 64) // upper part of row k, 64 entries per round
        {
            const int  jj   = j0 + lane;
            const bool have = jj < ke;
            int        c    = c0;
            T          akw  = akw0;
            if(j0 != dk + 1)
            {
                c   = have ? col[jj] - base : -1;
                akw = T(0);
                if(have)
                    akw = __hip_atomic_load(&val[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for(int w = 0; w < len; w++) // the reference's column -> position map, as a scan of the LDS row
                if(have && sc[w] == c)
                    sv[w] = ilu_nfma(lik, akw, sv[w]);
        }
        if(lane == 0)
            sv[t] = lik;
        __syncthreads();
    }
    if(!bad && (t >= len || k != i || ilu_near_zero(sv[t])))
        bad = true; // no diagonal right after the lower part, or a (near-)zero pivot (:95-101)
    if(bad)
    {
        if(lane == 0)
        {
            atomicExch(error, 1);
            __hip_atomic_store(&diag[i], -2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    for(int w = lane; w < len; w += 64)
        __hip_atomic_store(&val[s + w], sv[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence(); // every lane's values before the flag
    __syncthreads();
    if(lane == 0)
        __hip_atomic_store(&diag[i], s + t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

} // namespace

template <typename T>
aoclsparse_status launch_ilu0_level(hipStream_t s, int base, aoclsparse_int nrows, const aoclsparse_int *rows,
                                    const aoclsparse_int *row_ptr, const aoclsparse_int *col, T *val,
                                    aoclsparse_int *diag, int maxlen, int *error)
{
    if(nrows <= 0)
        return aoclsparse_status_success;
    const size_t lds = (sizeof(T) + sizeof(aoclsparse_int)) * (size_t)maxlen;
    // the row lives in LDS: beyond the 64 KB a kernel gets by default the limit has to be raised explicitly, and a row
    // that does not fit the CU's 160 KB at all (> ~13,600 fp64 entries) cannot be factorised by this kernel
    constexpr size_t LDS_DEFAULT = 64u << 10, LDS_MAX = 160u << 10;
    if(lds > LDS_MAX)
        return aoclsparse_status_not_implemented;
    if(lds > LDS_DEFAULT)
        MI355_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ilu0_level_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
    hipLaunchKernelGGL((ilu0_level_kernel<T>), dim3(nrows), dim3(64), lds, s, base, rows, row_ptr, col, val, diag,
                       maxlen, error);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_ilu0_level<double>(hipStream_t, int, aoclsparse_int, const aoclsparse_int *,
                                                     const aoclsparse_int *, const aoclsparse_int *, double *,
                                                     aoclsparse_int *, int, int *);
template aoclsparse_status launch_ilu0_level<float>(hipStream_t, int, aoclsparse_int, const aoclsparse_int *,
                                                    const aoclsparse_int *, const aoclsparse_int *, float *,
                                                    aoclsparse_int *, int, int *);
template aoclsparse_status launch_ilu0_level<cdouble>(hipStream_t, int, aoclsparse_int, const aoclsparse_int *,
                                                      const aoclsparse_int *, const aoclsparse_int *, cdouble *,
                                                      aoclsparse_int *, int, int *);
template aoclsparse_status launch_ilu0_level<cfloat>(hipStream_t, int, aoclsparse_int, const aoclsparse_int *,
                                                     const aoclsparse_int *, const aoclsparse_int *, cfloat *,
                                                     aoclsparse_int *, int, int *);

// real types only; diag must hold -1 everywhere, *ticket 0, *error 0
template <typename T>
aoclsparse_status launch_ilu0_syncfree(hipStream_t s, int base, aoclsparse_int n, const aoclsparse_int *rows,
                                       const aoclsparse_int *row_ptr, const aoclsparse_int *col, T *val, aoclsparse_int *diag,
                                       int maxlen, int *error, unsigned int *ticket)
{
    if(n <= 0)
        return aoclsparse_status_success;
    const size_t     lds = (sizeof(T) + sizeof(aoclsparse_int)) * (size_t)maxlen;
    constexpr size_t LDS_DEFAULT = 64u << 10, LDS_MAX = 160u << 10;
    if(lds > LDS_MAX)
        return aoclsparse_status_not_implemented;
    if(lds > LDS_DEFAULT)
        MI355_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&ilu0_syncfree_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
    hipLaunchKernelGGL((ilu0_syncfree_kernel<T>), dim3(n), dim3(64), lds, s, base, n, rows, row_ptr, col, val, diag, maxlen,
                       error, ticket);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}
template aoclsparse_status launch_ilu0_syncfree<double>(hipStream_t, int, aoclsparse_int, const aoclsparse_int *,
                                                        const aoclsparse_int *, const aoclsparse_int *, double *,
                                                        aoclsparse_int *, int, int *, unsigned int *);
template aoclsparse_status launch_ilu0_syncfree<float>(hipStream_t, int, aoclsparse_int, const aoclsparse_int *,
                                                       const aoclsparse_int *, const aoclsparse_int *, float *,
                                                       aoclsparse_int *, int, int *, unsigned int *);

} // namespace mi355
