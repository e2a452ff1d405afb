// blk_api.cpp -- BLKCSR, the blocked format the reference's optimize step picks for rows of >= 10 non-zeros on
// an AVX-512 host: aoclsparse_opt_blksize, aoclsparse_csr2blkcsr (host, integer work -- they hand host arrays
// to the caller exactly as the reference does) and aoclsparse_dblkcsrmv (GPU).
//
//   block-size choice : conversion/aoclsparse_convert.cpp:36-147
//   conversion        : conversion/aoclsparse_convert.cpp:149-310
//   product, checks   : level2/aoclsparse_blkcsrmv.hpp:38-147; kernels level2/aoclsparse_blkcsrmv_avx512.cpp:40-369
#include "internal.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <vector>

using namespace mi355;

namespace
{

#define MI355_TRY(expr)                       \
    do                                        \
    {                                         \
        aoclsparse_status st__ = (expr);      \
        if(st__ != aoclsparse_status_success) \
            return st__;                      \
    } while(0)

constexpr aoclsparse_int WINDOW = 8; // columns per block

// Cursor over the sub-rows of one row block.  open() returns the first column nobody has consumed yet (or
// INT_MAX), take() consumes what falls inside the window [c, c+8) and reports every consumed entry.
struct BlockWalk
{
    aoclsparse_int        rows, i0, m;
    aoclsparse_index_base base;
    const aoclsparse_int *ptr, *col;
    aoclsparse_int        pos[4];

    BlockWalk(aoclsparse_int rows_, aoclsparse_int i0_, aoclsparse_int m_, aoclsparse_index_base base_,
              const aoclsparse_int *ptr_, const aoclsparse_int *col_)
        : rows(rows_), i0(i0_), m(m_), base(base_), ptr(ptr_), col(col_)
    {
        for(aoclsparse_int r = 0; r < 4; r++)
            pos[r] = (r < rows && i0 + r < m) ? ptr[i0 + r] - base : 0;
    }
    aoclsparse_int live() const
    {
        return std::min(rows, m - i0);
    }
    aoclsparse_int open() const
    {
        aoclsparse_int c = INT_MAX;
        for(aoclsparse_int r = 0; r < live(); r++)
            if(pos[r] < ptr[i0 + r + 1] - base)
                c = std::min(c, col[pos[r]] - base);
        return c;
    }
    template <typename F>
    void take(aoclsparse_int c, F &&entry)
    {
        for(aoclsparse_int r = 0; r < live(); r++)
            for(; pos[r] < ptr[i0 + r + 1] - base && col[pos[r]] - base < c + WINDOW; pos[r]++)
                entry(r, pos[r], col[pos[r]] - base - c);
    }
};

aoclsparse_int count_blocks(aoclsparse_int rows, aoclsparse_int m, aoclsparse_index_base base, const aoclsparse_int *ptr,
                            const aoclsparse_int *col)
{
    aoclsparse_int blocks = 0;
    for(aoclsparse_int i0 = 0; i0 < m; i0 += rows)
    {
        BlockWalk w(rows, i0, m, base, ptr, col);
        for(aoclsparse_int c = w.open(); c != INT_MAX; c = w.open(), blocks++)
            w.take(c, [](aoclsparse_int, aoclsparse_int, aoclsparse_int) {});
    }
    return blocks;
}

// The value offset of every block (a running popcount of the masks) is recomputed on every call by three small
// launches into scratch: a cache keyed on the arrays' device addresses (round 1) could be served stale after the
// caller freed and re-allocated its arrays, and the raw-array API has no handle to tie an analysis to.
} // namespace

extern "C" {

aoclsparse_int aoclsparse_opt_blksize(aoclsparse_int m, aoclsparse_int nnz, aoclsparse_index_base base,
                                      const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                      aoclsparse_int *total_blks)
{
    if(m <= 0 || nnz <= 0 || !csr_row_ptr || !csr_col_ind || !total_blks)
        return 0;
    const aoclsparse_int rows[3] = {1, 2, 4};
    aoclsparse_int       blocks[3];
    double               per_blk[3], util[3], gain[2] = {0.0, 0.0};
    const double         per_row = static_cast<double>(nnz) / m;
    for(int f = 0; f < 3; f++)
    {
        blocks[f] = count_blocks(rows[f], m, base, csr_row_ptr, csr_col_ind);
        if(blocks[f] == 0)
            return 0;
        per_blk[f] = double(nnz) / double(blocks[f]);
        util[f]    = per_blk[f] / (double(rows[f]) * WINDOW) * 100;
        if((per_row < 30 && util[0] < 40) || (per_row > 30 && util[0] < 50))
            return 0;
        if(f)
            gain[f - 1] = (per_blk[f] - per_blk[f - 1]) / per_blk[f - 1] * 100;
    }
    // the reference's unqualified abs() resolves to the integer one (convert.cpp:128-129 with its include
    // set), i.e. both differences are truncated towards zero before the comparison; kept
    const double d_gain = std::abs(static_cast<int>(gain[0] - gain[1]));
    const double d_util = std::abs(static_cast<int>(util[1] - util[2]));
    if(util[2] > 24 && (d_gain < 12.5 || d_util < 12.5) && gain[1] > 51)
    {
        *total_blks = blocks[2];
        return 4;
    }
    if(util[1] > 28)
    {
        *total_blks = blocks[1];
        return 2;
    }
    return 0;
}

aoclsparse_status aoclsparse_csr2blkcsr(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                        const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                        const double *csr_val, aoclsparse_int *blk_row_ptr, aoclsparse_int *blk_col_ind,
                                        double *blk_csr_val, uint8_t *masks, aoclsparse_int nRowsblk,
                                        aoclsparse_index_base base)
{
    if(m < 0 || n < WINDOW || nnz < 0)
        return aoclsparse_status_invalid_size;
    if(!csr_row_ptr || !csr_col_ind || !csr_val || !blk_row_ptr || !blk_col_ind || !blk_csr_val || !masks)
        return aoclsparse_status_invalid_pointer;
    if(nRowsblk != 1 && nRowsblk != 2 && nRowsblk != 4)
        return aoclsparse_status_invalid_size;
    aoclsparse_int blocks = 0;
    size_t         stored = 0;
    for(aoclsparse_int i0 = 0; i0 < m; i0 += nRowsblk)
    {
        BlockWalk            w(nRowsblk, i0, m, base, csr_row_ptr, csr_col_ind);
        const aoclsparse_int first = blocks;
        for(aoclsparse_int c = w.open(); c != INT_MAX; c = w.open(), blocks++)
        {
            // a window that would run past the last column is anchored at n-8 and its bits move up (:249-254)
            const aoclsparse_int slide = c + WINDOW > n ? c + WINDOW - n : 0;
            uint8_t              bits[4] = {0, 0, 0, 0};
            w.take(c, [&](aoclsparse_int r, aoclsparse_int at, aoclsparse_int lane) {
                blk_csr_val[stored++] = csr_val[at];
                bits[r] |= static_cast<uint8_t>(1u << lane);
            });
            blk_col_ind[blocks] = c - slide + base;
            for(aoclsparse_int r = 0; r < nRowsblk; r++)
                masks[(size_t)blocks * nRowsblk + r] = static_cast<uint8_t>(bits[r] << slide);
        }
        // the first sub-row holds the block range, the others point at its end (:287-293)
        blk_row_ptr[i0] = first + base;
        for(aoclsparse_int r = 1; r < nRowsblk && i0 + r < m; r++)
            blk_row_ptr[i0 + r] = blocks + base;
    }
    blk_row_ptr[m] = blocks + base;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_dblkcsrmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                       aoclsparse_int n, aoclsparse_int nnz, const uint8_t *masks,
                                       const double *blk_csr_val, const aoclsparse_int *blk_col_ind,
                                       const aoclsparse_int *blk_row_ptr, const aoclsparse_mat_descr descr,
                                       const double *x, const double *beta, double *y, aoclsparse_int nRowsblk)
{
    // blkcsrmv.hpp:64-143, in the reference's order
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != aoclsparse_index_base_zero && descr->base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_symmetric)
        return aoclsparse_status_not_implemented;
    if(trans != aoclsparse_operation_none)
        return aoclsparse_status_not_implemented;
    if(m < 0 || n < WINDOW || nnz < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0 || nnz == 0)
        return aoclsparse_status_success;
    if(!blk_csr_val || !blk_row_ptr || !blk_col_ind || !masks || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(nRowsblk != 1 && nRowsblk != 2 && nRowsblk != 4)
        return aoclsparse_status_invalid_size;
    if(!alpha || !beta) // dereferenced unchecked by the reference (its tests note the crash); refused here
        return aoclsparse_status_invalid_pointer;
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    const aoclsparse_int last = (m - 1) / nRowsblk * nRowsblk; // the last row block's first sub-row holds its range
    aoclsparse_int       nblk = 0;
    const void          *valoff = nullptr;
    if(rt.is_device_pointer(blk_row_ptr))
    {
        MI355_HIP_TRY(hipMemcpyAsync(&nblk, blk_row_ptr + last + 1, sizeof(nblk), hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    }
    else
        nblk = blk_row_ptr[last + 1];
    nblk -= descr->base;
    if(nblk < 0)
        return aoclsparse_status_invalid_value;
    StagedArg am, av, ac, ap, ax, ay;
    MI355_TRY(am.in(rt, 8, masks, (size_t)nblk * nRowsblk, true));
    MI355_TRY(av.in(rt, 9, blk_csr_val, sizeof(double) * (size_t)nnz, true));
    MI355_TRY(ac.in(rt, 10, blk_col_ind, sizeof(aoclsparse_int) * (size_t)nblk, true));
    MI355_TRY(ap.in(rt, 11, blk_row_ptr, sizeof(aoclsparse_int) * ((size_t)m + 1), true));
    MI355_TRY(ax.in(rt, 3, x, sizeof(double) * (size_t)n, true));
    MI355_TRY(ay.in(rt, 4, y, sizeof(double) * (size_t)m, *beta != 0.0));
    {
        const size_t nparts = ((size_t)nblk + (1u << BLK_PART_SHIFT) - 1) >> BLK_PART_SHIFT;
        const size_t vbytes = sizeof(aoclsparse_int) * std::max<size_t>(1, (size_t)nblk);
        const size_t pbytes = sizeof(aoclsparse_int) * std::max<size_t>(1, nparts);
        void        *vo = nullptr, *pt = nullptr;
        MI355_TRY(rt.staging(14, vbytes, &vo));
        MI355_TRY(rt.staging(15, pbytes, &pt));
        MI355_TRY(launch_blk_valoff(rt.stream(), nblk, (int)nRowsblk, static_cast<const uint8_t *>(am.dev),
                                    static_cast<aoclsparse_int *>(vo), static_cast<aoclsparse_int *>(pt)));
        valoff = vo;
    }
    MI355_TRY(launch_blkcsrmv(rt.stream(), descr->base, *alpha, m, (int)nRowsblk, static_cast<const uint8_t *>(am.dev),
                              static_cast<const double *>(av.dev), static_cast<const aoclsparse_int *>(ac.dev),
                              static_cast<const aoclsparse_int *>(ap.dev), static_cast<const aoclsparse_int *>(valoff),
                              static_cast<const double *>(ax.dev), *beta, static_cast<double *>(ay.dev)));
    MI355_TRY(ay.out(rt));
    if(ay.staged)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return aoclsparse_status_success;
}

} // extern "C"
