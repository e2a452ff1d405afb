// blk_kernels.hip -- SpMV on the reference's BLKCSR storage (1/2/4 x 8 blocks + one bit mask per sub-row).
//
// The CPU kernels (level2/aoclsparse_blkcsrmv_avx512.cpp:40-369) walk the packed value array with a running
// popcount.  Here the running count becomes data: three small launches (per-chunk popcount scan, scan of the chunk
// totals, add) turn the masks into the absolute value offset of every block, after which the blocks of a row are
// independent loads.  blk_mv_kernel gives every matrix row a group of 8 lanes = the 8 lanes of the reference's zmm
// accumulator: lane l owns column (window start + l), multiplies its value (or 0 where the mask has no bit,
// as the expand-load does) with x and chains the FMAs block after block; the group then reduces lo4+hi4,
// pairs, pairs -- the reference's extract/hadd/add -- so the result is bit-identical to the AVX-512 kernel.
// Bound: HBM, 8 B/nnz values + (8 + rows_blk) B/block (column, value offset, masks) + the x windows.
#include "internal.hpp"

#include <hip/hip_runtime.h>

namespace mi355
{

namespace
{

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_PER   = 4; // blocks per thread -> 1024 blocks per workgroup (= 1 << BLK_PART_SHIFT)

__device__ __forceinline__ int wg_exclusive_scan(int v, int *s_wave, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int       inc = v;
#pragma unroll
    for(int d = 1; d < 64; d <<= 1)
    {
        int t = __shfl_up(inc, d);
        if(lane >= d)
            inc += t;
    }
    if(lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
    for(int q = 0; q < SCAN_BLOCK / 64; q++)
    {
        if(q < wave)
            before += s_wave[q];
        all += s_wave[q];
    }
    __syncthreads();
    total = all;
    return before + inc - v;
}

// pass 1: valoff[b] = values stored before block b within its chunk of 1024 blocks; part[chunk] = values in the chunk
__global__ __launch_bounds__(SCAN_BLOCK) void blk_popc_kernel(aoclsparse_int nblk, int rows, const uint8_t *__restrict__ masks,
                                                              aoclsparse_int *__restrict__ valoff,
                                                              aoclsparse_int *__restrict__ part)
{
    __shared__ int s_wave[SCAN_BLOCK / 64];
    const long long b0 = ((long long)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_PER;
    int             c[SCAN_PER], mine = 0;
#pragma unroll
    for(int k = 0; k < SCAN_PER; k++)
    {
        c[k] = 0;
        if(b0 + k < nblk)
            for(int r = 0; r < rows; r++)
                c[k] += __popc((unsigned)masks[(b0 + k) * rows + r]);
        mine += c[k];
    }
    int total;
    int pre = wg_exclusive_scan(mine, s_wave, total);
#pragma unroll
    for(int k = 0; k < SCAN_PER; k++)
    {
        if(b0 + k < nblk)
            valoff[b0 + k] = pre;
        pre += c[k];
    }
    if(threadIdx.x == 0)
        part[blockIdx.x] = total;
}

// exclusive scan of the chunk totals, one workgroup, carry across passes
__global__ __launch_bounds__(SCAN_BLOCK) void blk_part_scan_kernel(aoclsparse_int nparts, aoclsparse_int *__restrict__ part)
{
    __shared__ int s_wave[SCAN_BLOCK / 64];
    int            carry = 0;
    for(aoclsparse_int p0 = 0; p0 < nparts; p0 += SCAN_BLOCK)
    {
        const aoclsparse_int p = p0 + threadIdx.x;
        const int            v = p < nparts ? part[p] : 0;
        int                  total;
        const int            pre = wg_exclusive_scan(v, s_wave, total);
        if(p < nparts)
            part[p] = carry + pre;
        carry += total;
    }
}

// valoff[b] += scanned total of the chunks before b's: absolute value offsets
__global__ __launch_bounds__(SCAN_BLOCK) void blk_add_part_kernel(aoclsparse_int nblk, aoclsparse_int *__restrict__ valoff,
                                                                  const aoclsparse_int *__restrict__ part)
{
    const long long b = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if(b < nblk)
        valoff[b] += part[b >> BLK_PART_SHIFT];
}

template <int ROWS>
__global__ __launch_bounds__(256) void blk_mv_kernel(int base, double alpha, aoclsparse_int m, const uint8_t *__restrict__ masks,
                                                     const double *__restrict__ val, const aoclsparse_int *__restrict__ col,
                                                     const aoclsparse_int *__restrict__ row_ptr,
                                                     const aoclsparse_int *__restrict__ valoff,
                                                     const double *__restrict__ x, double beta, double *__restrict__ y)
{
    const long long i    = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int       l    = threadIdx.x & 7;
    const bool      live = i < m;
    const int       r    = (int)(i % ROWS);
    const long long i0   = i - r;
    aoclsparse_int  b = 0, b1 = 0;
    if(live)
        b = row_ptr[i0] - base, b1 = row_ptr[i0 + 1] - base; // the row block's range is held by its first sub-row
    double acc = 0.0;
    // The blocks of a row are independent loads once their value offsets are known.  Per batch of 8 blocks, lane
    // k of the group fetches the metadata of block b+k (coalesced), the group exchanges it by shuffles, all
    // values and x windows of the batch are fetched together, then the FMAs are chained in block order.
    constexpr int BATCH = 8;
    for(; b < b1; b += BATCH)
    {
        const aoclsparse_int bb = min(b + l, b1 - 1); // clamped lanes repeat the last block and are masked off below
        const int            my_off = valoff[bb], my_col = col[bb] - base;
        unsigned             my_mw = 0; // the block's ROWS masks, sub-row q in byte q
#pragma unroll
        for(int q = 0; q < ROWS; q++)
            my_mw |= (unsigned)masks[(long long)bb * ROWS + q] << (8 * q);
        double v[BATCH], xv[BATCH];
#pragma unroll
        for(int k = 0; k < BATCH; k++)
        {
            const int      off = __shfl(my_off, k, 8), c = __shfl(my_col, k, 8);
            const unsigned mw = __shfl(my_mw, k, 8);
            const unsigned mk = (mw >> (8 * r)) & 0xffu;
            const int      at = off + __popc(mw & ((1u << (8 * r)) - 1u)) + __popc(mk & ((1u << l) - 1u));
            xv[k]             = x[(long long)c + l];
            v[k]              = (mk >> l) & 1u ? val[at] : 0.0;
        }
#pragma unroll
        for(int k = 0; k < BATCH; k++)
            if(b + k < b1)
                acc = fma(v[k], xv[k], acc);
    }
    // lo4 + hi4, then (v0 + v1), (v2 + v3), then their sum -- blkcsrmv_avx512.cpp:79-92
    double t = acc + __shfl_down(acc, 4, 8);
    double u = t + __shfl_down(t, 1, 8);
    double w = u + __shfl_down(u, 2, 8);
    if(live && l == 0)
    {
        double sum = 0.0 + w;
        if(alpha != 1.0)
            sum = alpha * sum;
        if(beta != 0.0)
            sum = fma(beta, y[i], sum);
        y[i] = sum;
    }
}

} // namespace

aoclsparse_status launch_blk_valoff(hipStream_t s, aoclsparse_int nblk, int rows, const uint8_t *masks,
                                    aoclsparse_int *valoff, aoclsparse_int *part)
{
    if(nblk <= 0)
        return aoclsparse_status_success;
    const aoclsparse_int nparts = (nblk + (1 << BLK_PART_SHIFT) - 1) >> BLK_PART_SHIFT;
    hipLaunchKernelGGL(blk_popc_kernel, dim3(nparts), dim3(SCAN_BLOCK), 0, s, nblk, rows, masks, valoff, part);
    hipLaunchKernelGGL(blk_part_scan_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, nparts, part);
    hipLaunchKernelGGL(blk_add_part_kernel, dim3((nblk + SCAN_BLOCK - 1) / SCAN_BLOCK), dim3(SCAN_BLOCK), 0, s, nblk, valoff, part);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

aoclsparse_status launch_blkcsrmv(hipStream_t s, int base, double alpha, aoclsparse_int m, int rows, const uint8_t *masks,
                                  const double *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                                  const aoclsparse_int *valoff, const double *x, double beta, double *y)
{
    if(m <= 0)
        return aoclsparse_status_success;
    const dim3 grid((unsigned)((m + 31) / 32)), block(256);
    if(rows == 1)
        hipLaunchKernelGGL(blk_mv_kernel<1>, grid, block, 0, s, base, alpha, m, masks, val, col, row_ptr, valoff, x, beta, y);
    else if(rows == 2)
        hipLaunchKernelGGL(blk_mv_kernel<2>, grid, block, 0, s, base, alpha, m, masks, val, col, row_ptr, valoff, x, beta, y);
    else
        hipLaunchKernelGGL(blk_mv_kernel<4>, grid, block, 0, s, base, alpha, m, masks, val, col, row_ptr, valoff, x, beta, y);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

} // namespace mi355
