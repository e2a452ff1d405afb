// transpose_kernels.hip -- B = A^T of a device CSR, in the order of the reference's counting sort (round 4).
//
// Reference: conversion/aoclsparse_convert.hpp:552-655 (aoclsparse_csr2csc_template): count the entries of every column, prefix
// sum, then walk the rows in order and drop each entry into the next free slot of its column -- a STABLE sort by column: inside
// a column the entries keep the order of the walk (ascending row; a repeated (i, c) keeps its CSR order).  The public ?csr2csc
// returns exactly that, the handles' transposes (?mv / ?csrmm with op = T, sp2m with op = T, whose first-touch order depends on
// it) are built the same way.  On the host it costs ~4.4 ns per entry on one core: 22 ms for 5 M entries, ~0.4 s for the 84 M of
// the headline matrix.
//
// Here: (1) column counts with atomics, (2) prefix sum (spg_scan_*), (3) every entry dropped into its column at an atomic
// cursor -- the order inside a column is then whatever the wavefronts' arrival order was -- together with its POSITION p in the
// CSR arrays and its row, (4) every column's segment sorted by p, which is the stable order (p grows with the row and, inside a
// row, with the CSR order): a thread per column for segments of <= 32 entries (insertion sort; arrival order is nearly sorted),
// a wavefront per column with an LDS counting rank for segments of <= 2,048, (5) values gathered by p.  A matrix with a longer
// column is declined (the caller keeps the host sort): the power-law graphs of config 2 are the only ones in the test set.
#include "internal.hpp"

#include <hip/hip_runtime.h>

#define MI355_TRY(expr)                            \
    do                                             \
    {                                              \
        aoclsparse_status st__ = (expr);           \
        if(st__ != aoclsparse_status_success)      \
            return st__;                           \
    } while(0)

namespace mi355
{

namespace
{
    constexpr int TR_ROW_WAVE = 128; // rows longer than this are walked by a wavefront
    // (a column index outside [0, n) raises *bad and is skipped: the caller then declines -- nothing is ever written out of range)
    __global__ __launch_bounds__(256) void tr_count_kernel(aoclsparse_int m, aoclsparse_int n, int base,
                                                           const aoclsparse_int *__restrict__ ptr,
                                                           const aoclsparse_int *__restrict__ ind, int *cnt, unsigned int *bad)
    {
        // a lane per row; a row of more than TR_ROW_WAVE entries is walked by its whole wavefront instead (as one lane's loop a
        // 190 k-entry row of a power-law matrix was the kernel: ADVICE r4)
        const int  i   = blockIdx.x * 256 + threadIdx.x;
        const bool act = i < m;
        const int  s = act ? ptr[i] - base : 0, e = act ? ptr[i + 1] - base : 0;
        const bool lng = e - s > TR_ROW_WAVE;
        if(!lng)
            for(int p = s; p < e; p++)
            {
                const int c = ind[p] - base;
                if((unsigned)c >= (unsigned)n)
                    atomicOr(bad, 1u);
                else
                    atomicAdd(&cnt[c], 1);
            }
        unsigned long long mask = __ballot(lng);
        while(mask)
        {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int ls = __builtin_amdgcn_readlane(s, l), le = __builtin_amdgcn_readlane(e, l);
            for(int p = ls + (int)(threadIdx.x & 63); p < le; p += 64)
            {
                const int c = ind[p] - base;
                if((unsigned)c >= (unsigned)n)
                    atomicOr(bad, 1u);
                else
                    atomicAdd(&cnt[c], 1);
            }
        }
    }

    __global__ __launch_bounds__(256) void tr_scatter_kernel(aoclsparse_int m, int base, const aoclsparse_int *__restrict__ ptr,
                                                             const aoclsparse_int *__restrict__ ind,
                                                             const aoclsparse_int *__restrict__ tptr, int *cursor,
                                                             int *__restrict__ tpos, aoclsparse_int *__restrict__ trow)
    {
        const int  i   = blockIdx.x * 256 + threadIdx.x;
        const bool act = i < m;
        const int  s = act ? ptr[i] - base : 0, e = act ? ptr[i + 1] - base : 0;
        const bool lng = e - s > TR_ROW_WAVE;
        if(!lng)
            for(int p = s; p < e; p++)
            {
                const int c = ind[p] - base;
                const int q = tptr[c] + atomicAdd(&cursor[c], 1);
                tpos[q] = p, trow[q] = i;
            }
        unsigned long long mask = __ballot(lng); // (long rows: the whole wavefront, as in tr_count_kernel)
        while(mask)
        {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int ls = __builtin_amdgcn_readlane(s, l), le = __builtin_amdgcn_readlane(e, l);
            const int li = __builtin_amdgcn_readlane(i, l);
            for(int p = ls + (int)(threadIdx.x & 63); p < le; p += 64)
            {
                const int c = ind[p] - base;
                const int q = tptr[c] + atomicAdd(&cursor[c], 1);
                tpos[q] = p, trow[q] = li;
            }
        }
    }

    // segments of <= 32 entries: a thread per column, insertion sort by position (the arrival order is nearly sorted already)
    __global__ __launch_bounds__(256) void tr_sort_short_kernel(aoclsparse_int n, const aoclsparse_int *__restrict__ tptr, int *tpos,
                                                                aoclsparse_int *trow)
    {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if(c >= n)
            return;
        const int s = tptr[c], e = tptr[c + 1];
        if(e - s > 32)
            return;
        for(int a = s + 1; a < e; a++)
        {
            const int kp = tpos[a], kr = trow[a];
            int       b  = a - 1;
            while(b >= s && tpos[b] > kp)
            {
                tpos[b + 1] = tpos[b], trow[b + 1] = trow[b];
                b--;
            }
            tpos[b + 1] = kp, trow[b + 1] = kr;
        }
    }

    // segments of 33 .. 2,048 entries: a wavefront per listed column; positions are distinct, so an entry's place is the number of
    // smaller positions in the segment
    constexpr int TR_WAVE_CAP = 2048;
    __global__ __launch_bounds__(64) void tr_sort_wave_kernel(aoclsparse_int ncols, const aoclsparse_int *__restrict__ cols,
                                                              const aoclsparse_int *__restrict__ tptr, int *tpos, aoclsparse_int *trow)
    {
        __shared__ int s_pos[TR_WAVE_CAP], s_row[TR_WAVE_CAP];
        const int      c = cols[blockIdx.x];
        const int      s = tptr[c], len = tptr[c + 1] - s, lane = threadIdx.x;
        if(len <= 32 || len > TR_WAVE_CAP)
            return;
        for(int t = lane; t < len; t += 64)
            s_pos[t] = tpos[s + t], s_row[t] = trow[s + t];
        __syncthreads();
        for(int t = lane; t < len; t += 64)
        {
            const int mine = s_pos[t];
            int       rank = 0;
            for(int u = 0; u < len; u++)
                rank += s_pos[u] < mine;
            tpos[s + rank] = mine, trow[s + rank] = s_row[t];
        }
    }

    template <typename V>
    __global__ __launch_bounds__(256) void tr_gather_kernel(aoclsparse_int nnz, const int *__restrict__ tpos, const V *__restrict__ val,
                                                            V *__restrict__ tval)
    {
        const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
        if(q < nnz)
            tval[q] = val[tpos[q]];
    }
} // namespace

// d_* : the CSR arrays of an m x n matrix in HBM (index base `base`); tptr (n + 1), tind (nnz), tval (nnz values of vsize bytes):
// the 0-based CSR of the transpose, ALLOCATED HERE once the matrix is accepted -- the column histogram comes first, so a declined
// matrix (aoclsparse_status_not_implemented: a column of more than 2,048 entries, i.e. every power-law graph; the caller sorts on
// the host) has cost n counters, not 12-20 bytes per entry of HBM held while the host sort runs.  On any failure the
// three output buffers are released again.
static aoclsparse_status device_transpose_run(hipStream_t s, aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz, int base,
                                              const aoclsparse_int *d_ptr, const aoclsparse_int *d_ind, const void *d_val,
                                              size_t vsize, DeviceBuffer &tptr_b, DeviceBuffer &tind_b, DeviceBuffer &tval_b)
{
    if(m <= 0 || n <= 0 || nnz <= 0)
        return aoclsparse_status_not_implemented;
    // temporaries: grow-only staging slots of the calling thread's runtime (round 5, ADVICE r4: five hipMalloc / hipFree per call
    // were five implicit device synchronisations), held under the stage lock until the stream has run the kernels
    Runtime                              &rt = Runtime::get();
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    void *p_cnt = nullptr, *p_cursor = nullptr, *p_tpos = nullptr, *p_scan = nullptr, *p_small = nullptr, *p_order = nullptr;
    // every way out of this function -- an allocation that fails half way included -- first lets the stream finish what was
    // enqueued on the slots: the next holder of the lock may grow (free) them
    struct DrainOnExit
    {
        hipStream_t s;
        ~DrainOnExit()
        {
            if(hipStreamSynchronize(s) != hipSuccess)
                (void)hipGetLastError();
        }
    } drain{s};
    MI355_TRY(rt.staging(40, sizeof(int) * (size_t)n, &p_cnt));
    MI355_TRY(rt.staging(41, 256, &p_small));
    MI355_HIP_TRY(hipMemsetAsync(p_cnt, 0, sizeof(int) * (size_t)n, s));
    MI355_HIP_TRY(hipMemsetAsync(p_small, 0, 256, s));
    int *const          cnt_p   = static_cast<int *>(p_cnt);
    unsigned int *const small_p = static_cast<unsigned int *>(p_small);
    unsigned int *d_bad = small_p + 32;
    hipLaunchKernelGGL(tr_count_kernel, dim3((m + 255) / 256), dim3(256), 0, s, m, n, base, d_ptr, d_ind, cnt_p, d_bad);
    // the columns by segment length: <= 32 (bin 0), <= 2,048 (bins 1, 2), longer (bins 3, 4: declined) -- the bins of the SpGEMM
    // analysis (32 / 256 / 2,048 / 8,192) serve as they are
    unsigned int *d_hist = small_p, *d_cursor = d_hist + 16;
    MI355_TRY(launch_spg_hist(s, n, cnt_p, nullptr, false, d_hist));
    unsigned int hist[SPGEMM_BINS + 1], bad = 0;
    MI355_HIP_TRY(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, s));
    MI355_HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s));
    MI355_HIP_TRY(hipStreamSynchronize(s));
    if(bad || hist[3] + hist[4] > 0)
        return aoclsparse_status_not_implemented;
    // accepted: now the outputs and the remaining temporaries
    MI355_TRY(tptr_b.alloc(sizeof(aoclsparse_int) * ((size_t)n + 1)));
    MI355_TRY(tind_b.alloc(sizeof(aoclsparse_int) * (size_t)nnz));
    MI355_TRY(tval_b.alloc(vsize * (size_t)nnz));
    MI355_TRY(rt.staging(42, sizeof(int) * (size_t)n, &p_cursor));
    MI355_TRY(rt.staging(43, sizeof(int) * (size_t)nnz, &p_tpos));
    MI355_TRY(rt.staging(44, spg_scan_scratch_bytes(n), &p_scan));
    MI355_HIP_TRY(hipMemsetAsync(p_cursor, 0, sizeof(int) * (size_t)n, s));
    int *const cursor_p = static_cast<int *>(p_cursor), *const tpos_p = static_cast<int *>(p_tpos);
    aoclsparse_int *tptr = tptr_b.as<aoclsparse_int>(), *tind = tind_b.as<aoclsparse_int>();
    void           *tval = tval_b.ptr;
    long long *total = nullptr;
    MI355_TRY(launch_spg_scan(s, n, cnt_p, tptr, static_cast<long long *>(p_scan), &total));
    hipLaunchKernelGGL(tr_scatter_kernel, dim3((m + 255) / 256), dim3(256), 0, s, m, base, d_ptr, d_ind, tptr, cursor_p,
                       tpos_p, tind);
    hipLaunchKernelGGL(tr_sort_short_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, tptr, tpos_p, tind);
    const aoclsparse_int nmid = (aoclsparse_int)(hist[1] + hist[2]);
    if(nmid > 0)
    {
        aoclsparse_int bounds[SPGEMM_BINS + 1];
        bounds[0] = 0;
        for(int b = 0; b < SPGEMM_BINS; b++)
            bounds[b + 1] = bounds[b] + (aoclsparse_int)hist[b];
        MI355_TRY(rt.staging(45, sizeof(aoclsparse_int) * (size_t)n, &p_order));
        aoclsparse_int *const order_p = static_cast<aoclsparse_int *>(p_order);
        MI355_TRY(launch_spg_order(s, n, cnt_p, false, bounds, d_cursor, order_p));
        hipLaunchKernelGGL(tr_sort_wave_kernel, dim3((unsigned)nmid), dim3(64), 0, s, nmid, order_p + bounds[1], tptr, tpos_p, tind);
    }
    const unsigned gq = (unsigned)(((long long)nnz + 255) / 256);
    if(vsize == 4)
        hipLaunchKernelGGL((tr_gather_kernel<float>), dim3(gq), dim3(256), 0, s, nnz, tpos_p, static_cast<const float *>(d_val),
                           static_cast<float *>(tval));
    else if(vsize == 8)
        hipLaunchKernelGGL((tr_gather_kernel<double>), dim3(gq), dim3(256), 0, s, nnz, tpos_p, static_cast<const double *>(d_val),
                           static_cast<double *>(tval));
    else if(vsize == 16)
        hipLaunchKernelGGL((tr_gather_kernel<double2>), dim3(gq), dim3(256), 0, s, nnz, tpos_p, static_cast<const double2 *>(d_val),
                           static_cast<double2 *>(tval));
    else
        return aoclsparse_status_not_implemented;
    MI355_HIP_TRY(hipGetLastError());
    MI355_HIP_TRY(hipStreamSynchronize(s)); // (the staging slots may be handed to the next caller once the lock is released)
    return aoclsparse_status_success; // (`drain` runs before the lock guard declared above it is released)
}

aoclsparse_status device_transpose(hipStream_t s, aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz, int base,
                                   const aoclsparse_int *d_ptr, const aoclsparse_int *d_ind, const void *d_val, size_t vsize,
                                   DeviceBuffer &tptr, DeviceBuffer &tind, DeviceBuffer &tval)
{
    const aoclsparse_status st = device_transpose_run(s, m, n, nnz, base, d_ptr, d_ind, d_val, vsize, tptr, tind, tval);
    if(st != aoclsparse_status_success)
        tptr.release(), tind.release(), tval.release();
    return st;
}

} // namespace mi355
