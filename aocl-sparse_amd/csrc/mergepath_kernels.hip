// mergepath_kernels.hip -- merge-path CSR SpMV: the balanced-work companion of the CSR-Adaptive kernel.
//
// The work list of a CSR SpMV is the merge of the m row ends with the nnz non-zeros (Merrill & Garland's
// merge path).  It is cut into tiles of MP_ITEMS consecutive items -- whatever the row lengths, every
// workgroup gets the same number of rows + non-zeros, and a row longer than a tile is spread over several
// workgroups.  The tile coordinates {row ends consumed, non-zeros consumed} are found on the HOST at plan
// time (one binary search per tile, integer work: matrix.cpp build_merge_plan), so the kernel does no searching:
//   phase 1  the tile's values and gathered x are parked in LDS together with the tile's slice of row_ptr;
//   phase 2  a lane per row walks its entries out of LDS as one left-to-right FMA chain -- the reference's
//            scalar order (csrmv_kr.hpp:448-513), so every row that lies inside one tile is bit-identical;
//   carries  the piece of a row cut by a tile boundary goes to a carry record (at most one tail piece and one
//            head piece per tile) and mp_fixup_kernel adds the pieces of each cut row in tile order
//            (deterministic; such rows carry the forward-error bound instead of bit-exactness).
// Served: the scalar order only (nnz <= 10 m -- where irregular rows live) without a pinned kid.
// HBM bytes as CSR-Adaptive + 24 B per tile of carries.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

namespace
{

constexpr int MP_BLOCK = 256;

__device__ __forceinline__ double mp_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float mp_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}
template <typename T>
__device__ __forceinline__ T mp_finish(T r, T alpha, T beta, const T *yi)
{
    if(alpha != T(1))
        r = alpha * r;
    if(beta != T(0))
        r = mp_fma(beta, *yi, r);
    return r;
}

// tile w owns the row ends [i0, i1) and the non-zeros [j0, j1) (0-based).  Carry records, two per tile:
//   [2w]   tail piece: row i0 started in an earlier tile and ends here         (row = i0, else -1)
//   [2w+1] head piece: the non-zeros after the tile's last row end belong to row i1, which ends later
template <typename T>
__global__ __launch_bounds__(MP_BLOCK) void mp_kernel(const int2 *__restrict__ starts,
                                                      const aoclsparse_int *__restrict__ row_ptr,
                                                      const aoclsparse_int *__restrict__ col,
                                                      const T *__restrict__ val, const T *__restrict__ x,
                                                      T *__restrict__ y, T alpha, T beta, int base,
                                                      aoclsparse_int *__restrict__ carry_row, T *__restrict__ carry_val)
{
    __shared__ T              s_val[MP_ITEMS];
    __shared__ T              s_x[MP_ITEMS];
    __shared__ aoclsparse_int s_row[MP_ITEMS + 2];
    const int  w   = blockIdx.x;
    const int  tid = threadIdx.x;
    const int2 a = starts[w], b = starts[w + 1];
    const int  i0 = a.x, j0 = a.y, i1 = b.x, j1 = b.y;
    const int  nr = i1 - i0, nz = j1 - j0;
    for(int t = tid; t <= nr; t += MP_BLOCK)
        s_row[t] = row_ptr[i0 + t] - base; // start of row i0 .. start of row i1
    for(int t = tid; t < nz; t += MP_BLOCK)
    {
        s_val[t] = val[j0 + t];
        s_x[t]   = x[col[j0 + t] - base];
    }
    __syncthreads();
    for(int t = tid; t <= nr; t += MP_BLOCK)
    {
        const int rs = s_row[t];
        const int s  = max(rs, j0) - j0;
        const int e  = (t < nr ? s_row[t + 1] : j1) - j0;
        T         r  = T(0);
        int       p  = s;
        for(; p + 8 <= e; p += 8) // one chain; the LDS reads of 8 entries are issued together
        {
            T av[8], xv[8];
#pragma unroll
            for(int q = 0; q < 8; q++)
                av[q] = s_val[p + q], xv[q] = s_x[p + q];
#pragma unroll
            for(int q = 0; q < 8; q++)
                r = mp_fma(av[q], xv[q], r);
        }
        for(; p < e; p++)
            r = mp_fma(s_val[p], s_x[p], r);
        if(t < nr)
        {
            if(t == 0 && rs < j0) // tail piece of a row cut by the tile's left boundary
            {
                carry_row[2 * w] = i0;
                carry_val[2 * w] = r;
            }
            else
                y[i0 + t] = mp_finish(r, alpha, beta, y + i0 + t);
        }
        else // t == nr: what follows the last row end belongs to row i1
        {
            carry_row[2 * w + 1] = e > s ? i1 : -1;
            carry_val[2 * w + 1] = r;
        }
    }
    if(tid == 0 && !(nr > 0 && s_row[0] < j0))
        carry_row[2 * w] = -1;
}

// one lane per tile holding the END of a cut row: walk back over the head pieces of that row, add them in tile
// order, finish and write y
template <typename T>
__global__ void mp_fixup_kernel(int ntiles, const aoclsparse_int *__restrict__ carry_row,
                                const T *__restrict__ carry_val, T *__restrict__ y, T alpha, T beta)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if(w >= ntiles)
        return;
    const int row = carry_row[2 * w];
    if(row < 0)
        return;
    // the tiles [f, w) hold the head pieces of this row: "head row == row" is false ... false, true ... true over [0, w), so f is
    // found by bisection (8 dependent loads for a row cut into 170 pieces instead of 170), and the pieces are summed in tile
    // order as before, their loads issued 16 at a time (round 4: the lane-by-lane walk made this kernel 39 us on a matrix with
    // four 170 k-entry rows, more than mp_kernel itself: profiles/r4/legs_kernel_stats.csv)
    int f = 0;
    {
        int b = w;
        while(f < b)
        {
            const int mid = (f + b) >> 1;
            if(carry_row[2 * mid + 1] == row)
                b = mid;
            else
                f = mid + 1;
        }
    }
    T   r = T(0);
    int v = f;
    for(; v + 16 <= w; v += 16)
    {
        T t[16];
#pragma unroll
        for(int q = 0; q < 16; q++)
            t[q] = carry_val[2 * (v + q) + 1];
#pragma unroll
        for(int q = 0; q < 16; q++)
            r += t[q];
    }
    for(; v < w; v++)
        r += carry_val[2 * v + 1];
    r += carry_val[2 * w];
    y[row] = mp_finish(r, alpha, beta, y + row);
}

} // namespace

template <typename T>
aoclsparse_status launch_mergepath(hipStream_t s, int base, T alpha, aoclsparse_int ntiles, const aoclsparse_int *starts,
                                   const T *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr, const T *x,
                                   T beta, T *y, aoclsparse_int *carry_row, T *carry_val)
{
    if(ntiles <= 0)
        return aoclsparse_status_success;
    hipLaunchKernelGGL((mp_kernel<T>), dim3(ntiles), dim3(MP_BLOCK), 0, s, reinterpret_cast<const int2 *>(starts), row_ptr,
                       col, val, x, y, alpha, beta, base, carry_row, carry_val);
    hipLaunchKernelGGL((mp_fixup_kernel<T>), dim3((ntiles + 255) / 256), dim3(256), 0, s, (int)ntiles, carry_row,
                       carry_val, y, alpha, beta);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template aoclsparse_status launch_mergepath<double>(hipStream_t, int, double, aoclsparse_int, const aoclsparse_int *,
                                                    const double *, const aoclsparse_int *, const aoclsparse_int *,
                                                    const double *, double, double *, aoclsparse_int *, double *);
template aoclsparse_status launch_mergepath<float>(hipStream_t, int, float, aoclsparse_int, const aoclsparse_int *,
                                                   const float *, const aoclsparse_int *, const aoclsparse_int *,
                                                   const float *, float, float *, aoclsparse_int *, float *);

} // namespace mi355
