// sorv_kernels.hip -- one level of the forward SOR sweep (aoclsparse_?sorv), gfx950.
//
// Reference: solvers/aoclsparse_sorv.hpp:78-113: for i = 0..n-1, axi = sum over the row's off-diagonal entries IN
// STORAGE ORDER of a_ij * x_j (contracted multiply-adds; x_j already updated for j < i, still the scaled input for
// j > i), then x_i += omega * ((b_i - axi) / a_ii - x_i).
// Row i depends only on the rows j < i that appear in it, i.e. on the level structure of the strict lower triangle --
// the same level sets the triangular solve uses.  One lane per row of the level walks the row exactly as the
// reference does, reading updated entries from x and not-yet-updated ones from a snapshot of the scaled input
// (a later row may already have been updated when its level precedes this one), so every bit is the reference's.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

template <typename T>
__global__ void sorv_level_kernel(const aoclsparse_int *__restrict__ rows, aoclsparse_int count, int base,
                                  const aoclsparse_int *__restrict__ ptr, const aoclsparse_int *__restrict__ ind,
                                  const T *__restrict__ val, T omega, T *x, const T *__restrict__ xold,
                                  const T *__restrict__ b)
{
    const aoclsparse_int t = blockIdx.x * blockDim.x + threadIdx.x;
    if(t >= count)
        return;
    const aoclsparse_int i = rows[t];
    T                    axi = T(0), diag = T(1);
    for(aoclsparse_int j = ptr[i] - base; j < ptr[i + 1] - base; j++)
    {
        const aoclsparse_int c = ind[j] - base;
        if(c == i)
            diag = val[j];
        else
            axi = fma(val[j], c < i ? x[c] : xold[c], axi);
    }
    const T xi = xold[i];
    x[i]       = fma(omega, (b[i] - axi) / diag - xi, xi);
}

template <typename T>
aoclsparse_status launch_sorv_level(hipStream_t s, const aoclsparse_int *rows, aoclsparse_int count, int base,
                                    const aoclsparse_int *ptr, const aoclsparse_int *ind, const T *val, T omega, T *x,
                                    const T *xold, const T *b)
{
    if(count <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((sorv_level_kernel<T>), dim3((count + 255) / 256), dim3(256), 0, s, rows, count, base, ptr, ind, val,
                       omega, x, xold, b);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_sorv_level<double>(hipStream_t, const aoclsparse_int *, aoclsparse_int, int,
                                                     const aoclsparse_int *, const aoclsparse_int *, const double *, double,
                                                     double *, const double *, const double *);
template aoclsparse_status launch_sorv_level<float>(hipStream_t, const aoclsparse_int *, aoclsparse_int, int,
                                                    const aoclsparse_int *, const aoclsparse_int *, const float *, float,
                                                    float *, const float *, const float *);

} // namespace mi355
