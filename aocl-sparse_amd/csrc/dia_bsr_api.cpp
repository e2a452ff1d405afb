// dia_bsr_api.cpp -- the DIA and BSR raw-array routines: aoclsparse_csr2dia_ndiag, aoclsparse_?csr2dia,
// aoclsparse_?diamv(_kid), aoclsparse_csr2bsr_nnz, aoclsparse_?csr2bsr, aoclsparse_?bsrmv.
//
// Conversions are host routines (the reference's are serial host loops over host arrays; a conversion is done once):
// conversion/aoclsparse_convert.cpp:510-566 (ndiag), conversion/aoclsparse_convert.hpp:291-387 (csr2dia),
// convert.cpp:596-729 (csr2bsr_nnz), convert.hpp:389-551 (csr2bsr, blocks sorted by column at the end).
// The products run on the GPU (dia_bsr_kernels.hip) with the checks of level2/aoclsparse_diamv.hpp:154-190 and
// level2/aoclsparse_bsrmv.cpp:84-141 in their order; arrays may be host or device memory.
#include "internal.hpp"

#include <algorithm>
#include <vector>

using namespace mi355;

namespace mi355
{
template <typename T>
aoclsparse_status launch_diamv(hipStream_t s, T alpha, aoclsparse_int m, aoclsparse_int n, const T *dia_val,
                               const aoclsparse_int *dia_offset, aoclsparse_int ndiag, const T *x, T beta, T *y);
template <typename T>
aoclsparse_status launch_bsrmv(hipStream_t s, T alpha, aoclsparse_int mb, aoclsparse_int dim, int base, const T *val,
                               const aoclsparse_int *col, const aoclsparse_int *row_ptr, const T *x, T beta, T *y);
}

namespace
{

bool valid_base(const aoclsparse_mat_descr d)
{
    return d->base == aoclsparse_index_base_zero || d->base == aoclsparse_index_base_one;
}

// marks which of the m+n-1 diagonals hold an entry; slot = col - row + m (convert.cpp:549-557)
aoclsparse_status mark_diagonals(aoclsparse_int m, aoclsparse_int n, int base, const aoclsparse_int *row_ptr,
                                 const aoclsparse_int *col_ind, std::vector<aoclsparse_int> &slot)
{
    try
    {
        slot.assign((size_t)m + (size_t)n, 0);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    for(aoclsparse_int i = 0; i < m; i++)
        for(aoclsparse_int p = row_ptr[i] - base; p < row_ptr[i + 1] - base; p++)
            slot[(size_t)(col_ind[p] - base - i + m)] = 1;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status csr2dia_t(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                            const aoclsparse_int *row_ptr, const aoclsparse_int *col_ind, const T *val,
                            aoclsparse_int ndiag, aoclsparse_int *dia_offset, T *dia_val)
{
    if(m < 0 || n < 0 || ndiag < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0 || ndiag == 0)
        return aoclsparse_status_success;
    if(!val || !row_ptr || !col_ind || !dia_val || !dia_offset || !descr)
        return aoclsparse_status_invalid_pointer;
    std::vector<aoclsparse_int> slot;
    const int                   base = descr->base;
    aoclsparse_status           st = mark_diagonals(m, n, base, row_ptr, col_ind, slot);
    if(st != aoclsparse_status_success)
        return st;
    aoclsparse_int d = 0;
    for(size_t k = 0; k < slot.size(); k++)
        if(slot[k])
        {
            slot[k]         = d; // rank of the diagonal, for the fill below
            dia_offset[d++] = (aoclsparse_int)k - m;
        }
    // entries land at [row + m * rank]; cells without an entry keep what the caller put there (convert.hpp:372-383)
    for(aoclsparse_int i = 0; i < m; i++)
        for(aoclsparse_int p = row_ptr[i] - base; p < row_ptr[i + 1] - base; p++)
            dia_val[(size_t)i + (size_t)m * slot[(size_t)(col_ind[p] - base - i + m)]] = val[p];
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status csr2bsr_t(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr, aoclsparse_order order,
                            const T *val, const aoclsparse_int *row_ptr, const aoclsparse_int *col_ind,
                            aoclsparse_int dim, T *bsr_val, aoclsparse_int *bsr_row_ptr, aoclsparse_int *bsr_col_ind)
{
    if(m < 0 || n < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0)
        return aoclsparse_status_success;
    if(dim <= 0)
        return aoclsparse_status_invalid_value;
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(!valid_base(descr))
        return aoclsparse_status_invalid_value;
    if(!val || !row_ptr || !col_ind || !bsr_val || !bsr_row_ptr || !bsr_col_ind)
        return aoclsparse_status_invalid_pointer;
    const aoclsparse_int        mb = (m + dim - 1) / dim, nb = (n + dim - 1) / dim;
    const int                   base = descr->base;
    const size_t                sq = (size_t)dim * dim;
    std::vector<long long>      where; // value offset of the block of each block column in the current block row
    std::vector<aoclsparse_int> perm;
    std::vector<T>              tmp;
    try
    {
        where.assign((size_t)nb, -1);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    for(aoclsparse_int bi = 0; bi < mb; bi++)
    {
        const aoclsparse_int first = bsr_row_ptr[bi] - base, last = bsr_row_ptr[bi + 1] - base;
        aoclsparse_int       next = first;
        for(aoclsparse_int i = 0; i < dim && bi * dim + i < m; i++)
        {
            const aoclsparse_int r = bi * dim + i;
            for(aoclsparse_int p = row_ptr[r] - base; p < row_ptr[r + 1] - base; p++)
            {
                const aoclsparse_int c = col_ind[p] - base, bc = c / dim, j = c % dim;
                if(where[bc] < 0)
                {
                    where[bc]           = (long long)next * (long long)sq;
                    bsr_col_ind[next++] = bc + base;
                }
                // element (i, j) of the block: row order i*dim + j, column order i + j*dim (convert.hpp:386, :493-496)
                bsr_val[where[bc] + (order == aoclsparse_order_row ? (size_t)i * dim + j : (size_t)i + (size_t)j * dim)] = val[p];
            }
        }
        for(aoclsparse_int k = first; k < last; k++)
            where[bsr_col_ind[k] - base] = -1;
        // blocks of the block row in ascending column order (the reference bubble-sorts them, convert.hpp:518-544;
        // block columns are distinct, so any stable sort gives the same arrangement)
        const aoclsparse_int cnt = last - first;
        if(cnt > 1 && !std::is_sorted(bsr_col_ind + first, bsr_col_ind + last))
        {
            try
            {
                perm.resize((size_t)cnt);
                tmp.resize((size_t)cnt * sq);
            }
            catch(const std::bad_alloc &)
            {
                return aoclsparse_status_memory_error;
            }
            for(aoclsparse_int k = 0; k < cnt; k++)
                perm[k] = k;
            std::sort(perm.begin(), perm.end(),
                      [&](aoclsparse_int a, aoclsparse_int b) { return bsr_col_ind[first + a] < bsr_col_ind[first + b]; });
            std::copy(bsr_val + (size_t)first * sq, bsr_val + (size_t)last * sq, tmp.begin());
            std::vector<aoclsparse_int> cols(bsr_col_ind + first, bsr_col_ind + last);
            for(aoclsparse_int k = 0; k < cnt; k++)
            {
                bsr_col_ind[first + k] = cols[perm[k]];
                std::copy(tmp.begin() + (size_t)perm[k] * sq, tmp.begin() + (size_t)(perm[k] + 1) * sq,
                          bsr_val + (size_t)(first + k) * sq);
            }
        }
    }
    return aoclsparse_status_success;
}

// device view of a host array for one call (slot of the runtime's scratch), or the pointer itself
template <typename U>
aoclsparse_status view_in(Runtime &rt, int slot, const U *p, size_t count, const U **out, bool copy = true)
{
    if(rt.is_device_pointer(p))
    {
        *out = p;
        return aoclsparse_status_success;
    }
    void             *d = nullptr;
    aoclsparse_status st = rt.staging(slot, sizeof(U) * std::max<size_t>(count, 1), &d);
    if(st != aoclsparse_status_success)
        return st;
    if(count && copy)
        MI355_HIP_TRY(hipMemcpyAsync(d, p, sizeof(U) * count, hipMemcpyHostToDevice, rt.stream()));
    *out = static_cast<const U *>(d);
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status finish_y(Runtime &rt, T *y, T *dy, size_t count)
{
    if(dy != y)
    {
        MI355_HIP_TRY(hipMemcpyAsync(y, dy, sizeof(T) * count, hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    }
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status diamv_t(aoclsparse_operation trans, const T *alpha, aoclsparse_int m, aoclsparse_int n,
                          const T *dia_val, const aoclsparse_int *dia_offset, aoclsparse_int ndiag,
                          const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y, aoclsparse_int kid)
{
    if(!alpha || !beta || !dia_val || !dia_offset || !x || !y || !descr)
        return aoclsparse_status_invalid_pointer;
    if(!valid_base(descr))
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_general || trans != aoclsparse_operation_none)
        return aoclsparse_status_not_implemented;
    if(m < 0 || n < 0 || ndiag < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0)
        return aoclsparse_status_success;
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();
    const T              *dv = nullptr, *dx = nullptr, *dy0 = nullptr;
    const aoclsparse_int *doff = nullptr;
    st = view_in(rt, 8, dia_val, (size_t)m * (size_t)ndiag, &dv);
    if(st == aoclsparse_status_success)
        st = view_in(rt, 9, dia_offset, (size_t)ndiag, &doff);
    if(st == aoclsparse_status_success)
        st = view_in(rt, 10, x, (size_t)n, &dx);
    if(st == aoclsparse_status_success)
        st = view_in(rt, 11, const_cast<const T *>(y), (size_t)m, &dy0, *beta != T(0)); // y is not read when beta == 0
    if(st != aoclsparse_status_success)
        return st;
    T *dy = const_cast<T *>(dy0);
    st    = launch_diamv<T>(rt.stream(), *alpha, m, n, dv, doff, ndiag, dx, *beta, dy);
    return st == aoclsparse_status_success ? finish_y(rt, y, dy, (size_t)m) : st;
}

template <typename T>
aoclsparse_status bsrmv_t(aoclsparse_operation trans, const T *alpha, aoclsparse_int mb, aoclsparse_int nb,
                          aoclsparse_int dim, const T *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                          const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y)
{
    if(!alpha || !beta)
        return aoclsparse_status_invalid_pointer;
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(!valid_base(descr))
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_general || trans != aoclsparse_operation_none)
        return aoclsparse_status_not_implemented;
    if(mb < 0 || nb < 0 || dim <= 0)
        return aoclsparse_status_invalid_size;
    if(mb == 0 || nb == 0)
        return aoclsparse_status_success;
    if(!val || !row_ptr || !col || !x || !y)
        return aoclsparse_status_invalid_pointer;
    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();
    const T              *dv = nullptr, *dx = nullptr, *dy0 = nullptr;
    const aoclsparse_int *dc = nullptr, *dp = nullptr;
    size_t                nblk = 0;
    if(!rt.is_device_pointer(row_ptr))
    {
        if(row_ptr[mb] < descr->base)
            return aoclsparse_status_invalid_value;
        nblk = (size_t)(row_ptr[mb] - descr->base);
    }
    else if(!rt.is_device_pointer(val) || !rt.is_device_pointer(col))
        return aoclsparse_status_invalid_value; // the block count lives in row_ptr: keep the three arrays together
    st = view_in(rt, 8, val, nblk * (size_t)dim * (size_t)dim, &dv);
    if(st == aoclsparse_status_success)
        st = view_in(rt, 9, col, nblk, &dc);
    if(st == aoclsparse_status_success)
        st = view_in(rt, 12, row_ptr, (size_t)mb + 1, &dp);
    if(st == aoclsparse_status_success)
        st = view_in(rt, 10, x, (size_t)nb * (size_t)dim, &dx);
    if(st == aoclsparse_status_success)
        st = view_in(rt, 11, const_cast<const T *>(y), (size_t)mb * (size_t)dim, &dy0, *beta != T(0));
    if(st != aoclsparse_status_success)
        return st;
    T *dy = const_cast<T *>(dy0);
    st    = launch_bsrmv<T>(rt.stream(), *alpha, mb, dim, descr->base, dv, dc, dp, dx, *beta, dy);
    return st == aoclsparse_status_success ? finish_y(rt, y, dy, (size_t)mb * (size_t)dim) : st;
}

} // namespace

extern "C" {

aoclsparse_status aoclsparse_csr2dia_ndiag(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                           aoclsparse_int nnz, const aoclsparse_int *csr_row_ptr,
                                           const aoclsparse_int *csr_col_ind, aoclsparse_int *dia_num_diag)
{
    if(m < 0 || n < 0 || nnz < 0)
        return aoclsparse_status_invalid_size;
    if(!dia_num_diag || !csr_row_ptr || !csr_col_ind || !descr)
        return aoclsparse_status_invalid_pointer;
    *dia_num_diag = 0;
    std::vector<aoclsparse_int> slot;
    aoclsparse_status           st = mark_diagonals(m, n, descr->base, csr_row_ptr, csr_col_ind, slot);
    if(st != aoclsparse_status_success)
        return st;
    aoclsparse_int cnt = 0;
    for(aoclsparse_int v : slot)
        cnt += v;
    *dia_num_diag = cnt;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_csr2bsr_nnz(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                         const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                         aoclsparse_int block_dim, aoclsparse_int *bsr_row_ptr, aoclsparse_int *bsr_nnz)
{
    if(m < 0 || n < 0 || block_dim <= 0)
        return aoclsparse_status_invalid_size;
    if(!csr_row_ptr || !csr_col_ind || !bsr_row_ptr || !bsr_nnz)
        return aoclsparse_status_invalid_pointer;
    if(m == 0 || n == 0)
    {
        *bsr_nnz = 0;
        return aoclsparse_status_success;
    }
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(!valid_base(descr))
        return aoclsparse_status_invalid_value;
    const int                   base = descr->base;
    const aoclsparse_int        mb = (m + block_dim - 1) / block_dim, nb = (n + block_dim - 1) / block_dim;
    std::vector<aoclsparse_int> seen; // last block row that touched a block column
    try
    {
        seen.assign((size_t)nb, -1);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    long long run = base;
    bsr_row_ptr[0] = base;
    for(aoclsparse_int bi = 0; bi < mb; bi++)
    {
        aoclsparse_int       blocks = 0;
        const aoclsparse_int r1 = std::min<long long>((long long)(bi + 1) * block_dim, m);
        for(aoclsparse_int r = bi * block_dim; r < r1; r++)
            for(aoclsparse_int p = csr_row_ptr[r] - base; p < csr_row_ptr[r + 1] - base; p++)
            {
                const aoclsparse_int bc = (csr_col_ind[p] - base) / block_dim;
                if(seen[bc] != bi)
                {
                    seen[bc] = bi;
                    blocks++;
                }
            }
        run += blocks;
        bsr_row_ptr[bi + 1] = (aoclsparse_int)run;
    }
    if(run > 2147483647LL)
        return aoclsparse_status_invalid_size; // convert.cpp:712-724
    *bsr_nnz = bsr_row_ptr[mb] - base;
    return aoclsparse_status_success;
}

#define MI355_DIA(P, T)                                                                                                 \
    aoclsparse_status aoclsparse_##P##csr2dia(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,     \
                                              const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,     \
                                              const T *csr_val, aoclsparse_int dia_num_diag, aoclsparse_int *dia_offset, \
                                              T *dia_val)                                                               \
    {                                                                                                                   \
        return csr2dia_t<T>(m, n, descr, csr_row_ptr, csr_col_ind, csr_val, dia_num_diag, dia_offset, dia_val);         \
    }                                                                                                                   \
    aoclsparse_status aoclsparse_##P##diamv(aoclsparse_operation trans, const T *alpha, aoclsparse_int m,               \
                                            aoclsparse_int n, aoclsparse_int nnz, const T *dia_val,                     \
                                            const aoclsparse_int *dia_offset, aoclsparse_int dia_num_diag,              \
                                            const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y)          \
    {                                                                                                                   \
        (void)nnz;                                                                                                      \
        return diamv_t<T>(trans, alpha, m, n, dia_val, dia_offset, dia_num_diag, descr, x, beta, y, -1);                \
    }                                                                                                                   \
    aoclsparse_status aoclsparse_##P##diamv_kid(aoclsparse_operation trans, const T *alpha, aoclsparse_int m,           \
                                                aoclsparse_int n, aoclsparse_int nnz, const T *dia_val,                 \
                                                const aoclsparse_int *dia_offset, aoclsparse_int dia_num_diag,          \
                                                const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y,      \
                                                aoclsparse_int diamv_mode, aoclsparse_int diamv_kid)                    \
    {                                                                                                                   \
        /* diamv.hpp:204-226: mode picks the CPU kernel family (0 = reference, which this kernel reproduces) */          \
        (void)nnz, (void)diamv_mode;                                                                                    \
        return diamv_t<T>(trans, alpha, m, n, dia_val, dia_offset, dia_num_diag, descr, x, beta, y, diamv_kid);         \
    }                                                                                                                   \
    aoclsparse_status aoclsparse_##P##bsrmv(aoclsparse_operation trans, const T *alpha, aoclsparse_int mb,              \
                                            aoclsparse_int nb, aoclsparse_int bsr_dim, const T *bsr_val,                \
                                            const aoclsparse_int *bsr_col_ind, const aoclsparse_int *bsr_row_ptr,       \
                                            const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y)          \
    {                                                                                                                   \
        return bsrmv_t<T>(trans, alpha, mb, nb, bsr_dim, bsr_val, bsr_col_ind, bsr_row_ptr, descr, x, beta, y);         \
    }
MI355_DIA(d, double)
MI355_DIA(s, float)

#define MI355_CSR2BSR(P, CT, T)                                                                                         \
    aoclsparse_status aoclsparse_##P##csr2bsr(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,     \
                                              const aoclsparse_order block_order, const CT *csr_val,                    \
                                              const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,     \
                                              aoclsparse_int block_dim, CT *bsr_val, aoclsparse_int *bsr_row_ptr,       \
                                              aoclsparse_int *bsr_col_ind)                                              \
    {                                                                                                                   \
        return csr2bsr_t<T>(m, n, descr, block_order, reinterpret_cast<const T *>(csr_val), csr_row_ptr, csr_col_ind,   \
                            block_dim, reinterpret_cast<T *>(bsr_val), bsr_row_ptr, bsr_col_ind);                       \
    }
MI355_CSR2BSR(d, double, double)
MI355_CSR2BSR(s, float, float)
MI355_CSR2BSR(z, aoclsparse_double_complex, cdouble)
MI355_CSR2BSR(c, aoclsparse_float_complex, cfloat)

} // extern "C"
