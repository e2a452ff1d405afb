// formats_api.cpp -- the data formats either side of the CSR path: CSC and COO handles,
// aoclsparse_convert_csr, aoclsparse_order_mat and the public ?csr2csc conversion.
//
//   create_?csc / create_?coo : extra/aoclsparse_auxiliary.cpp:373-560 (templates in the same file)
//   export_?csc / export_?coo : extra/aoclsparse_auxiliary.hpp:299-354, auxiliary.cpp:880-960
//   order_mat                 : extra/aoclsparse_auxiliary.cpp:840-878, :1259-1297
//   convert_csr               : conversion/aoclsparse_convert.cpp:1019-1215
//   ?csr2csc, coo2csr         : conversion/aoclsparse_convert.hpp:552-655, :657-750
//
// These are host-side structure conversions (integer / byte work), exactly as in the reference.  A CSC handle
// differs from the reference in ONE respect: the reference keeps the CSC arrays as "the CSR of A^T" and flips
// the operation at dispatch time; here the CSR of A is built once at creation (counting sort, owned by the
// handle) so that every executor of the path runs unchanged, and export_?csc hands back the caller's arrays.
#include "internal.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
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

// B = A^T by counting sort, independent index bases (convert.hpp:552-655); rows of B come out sorted when
// the rows of A are visited in order (stable)
template <typename T>
aoclsparse_status csr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz, int base_in, int base_out,
                          const aoclsparse_int *ptr, const aoclsparse_int *ind, const T *val, aoclsparse_int *oind,
                          aoclsparse_int *optr, T *oval)
{
    if(m < 0 || n < 0 || nnz < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0 || nnz == 0)
    {
        if(optr)
            for(aoclsparse_int i = 0; i < n + 1; i++)
                optr[i] = base_out;
        return aoclsparse_status_success;
    }
    if((base_in != 0 && base_in != 1) || (base_out != 0 && base_out != 1))
        return aoclsparse_status_invalid_value;
    if(!val || !ptr || !ind || !oval || !oind || !optr)
        return aoclsparse_status_invalid_pointer;
    // Large conversions run on the device (transpose_kernels.hip: the same stable order); the arrays cross PCIe both ways, which is
    // still several times faster than one core's counting sort.  Declined (a column of more than 2,048 entries) or failed: the host
    // loop below.
    if(nnz >= (1 << 20) && ptr[m] - base_in == nnz)
    {
        Runtime &rt = Runtime::get();
        if(rt.init() == aoclsparse_status_success)
        {
            std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
            hipStream_t                           s = rt.stream();
            DeviceBuffer                          ip, ii, iv, op, oi, ov;
            aoclsparse_status                     st = ip.upload(ptr, sizeof(aoclsparse_int) * ((size_t)m + 1), s);
            if(st == aoclsparse_status_success)
                st = ii.upload(ind, sizeof(aoclsparse_int) * (size_t)nnz, s);
            if(st == aoclsparse_status_success)
                st = iv.upload(val, sizeof(T) * (size_t)nnz, s);
            if(st == aoclsparse_status_success) // (allocates op / oi / ov once the column histogram has accepted the matrix)
                st = device_transpose(s, m, n, nnz, base_in, ip.as<aoclsparse_int>(), ii.as<aoclsparse_int>(), iv.ptr, sizeof(T), op,
                                      oi, ov);
            if(st == aoclsparse_status_success
               && hipMemcpyAsync(optr, op.ptr, sizeof(aoclsparse_int) * ((size_t)n + 1), hipMemcpyDeviceToHost, s) == hipSuccess
               && hipMemcpyAsync(oind, oi.ptr, sizeof(aoclsparse_int) * (size_t)nnz, hipMemcpyDeviceToHost, s) == hipSuccess
               && hipMemcpyAsync(oval, ov.ptr, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToHost, s) == hipSuccess
               && hipStreamSynchronize(s) == hipSuccess)
            {
                if(base_out != 0)
                {
                    parallel_for((long long)n + 1, 1 << 16, [&](long long a, long long b) {
                        for(long long i = a; i < b; i++)
                            optr[i] += base_out;
                    });
                    parallel_for(nnz, 1 << 18, [&](long long a, long long b) {
                        for(long long q = a; q < b; q++)
                            oind[q] += base_out;
                    });
                }
                return aoclsparse_status_success;
            }
            (void)hipGetLastError();
        }
    }
    std::fill(optr, optr + n + 1, 0);
    for(aoclsparse_int i = 0; i < nnz; i++)
        ++optr[ind[i] - base_in + 1];
    for(aoclsparse_int i = 0; i < n; i++)
        optr[i + 1] += optr[i];
    for(aoclsparse_int i = 0; i < m; i++)
        for(aoclsparse_int j = ptr[i] - base_in; j < ptr[i + 1] - base_in; j++)
        {
            const aoclsparse_int c = ind[j] - base_in, o = optr[c]++;
            oind[o] = i + base_out, oval[o] = val[j];
        }
    for(aoclsparse_int i = n; i > 0; i--)
        optr[i] = optr[i - 1] + base_out;
    optr[0] = base_out;
    return aoclsparse_status_success;
}

// convert.hpp:657-750: stable counting sort by row, entries of a row keep their COO order
template <typename T>
void coo2csr(aoclsparse_int M, aoclsparse_int nnz, int base, const aoclsparse_int *row, const aoclsparse_int *col,
             const T *val, aoclsparse_int *ptr, aoclsparse_int *ind, T *oval)
{
    std::fill(ptr, ptr + M + 1, 0);
    for(aoclsparse_int i = 0; i < nnz; i++)
        ++ptr[row[i] + 1 - base];
    for(aoclsparse_int i = 0; i < M; i++)
        ptr[i + 1] += ptr[i];
    for(aoclsparse_int i = 0; i < nnz; i++)
    {
        const aoclsparse_int o = ptr[row[i] - base]++;
        ind[o] = col[i], oval[o] = val[i];
    }
    for(aoclsparse_int i = M; i > 0; i--)
        ptr[i] = ptr[i - 1] + base;
    ptr[0] = base;
}

// a handle owning freshly allocated CSR arrays
aoclsparse_status new_owned_csr(aoclsparse_matrix *out, aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                aoclsparse_index_base base, aoclsparse_matrix_data_type vt)
{
    _aoclsparse_matrix *A = new(std::nothrow) _aoclsparse_matrix;
    if(!A)
        return aoclsparse_status_memory_error;
    try
    {
        A->user.ptr = new aoclsparse_int[(size_t)m + 1];
        A->user.ind = new aoclsparse_int[nnz > 0 ? nnz : 1];
        A->user.val = ::operator new(val_size(vt) * (size_t)(nnz > 0 ? nnz : 1));
    }
    catch(const std::bad_alloc &)
    {
        A->user.owned = true;
        delete A;
        return aoclsparse_status_memory_error;
    }
    A->user.owned = true;
    A->m = m, A->n = n, A->nnz = nnz, A->base = base, A->val_type = vt;
    A->user.m = m, A->user.n = n, A->user.nnz = nnz, A->user.base = base;
    A->owns_user_arrays = true;
    *out                = A;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status create_csc(aoclsparse_matrix *mat, aoclsparse_index_base base, aoclsparse_int M, aoclsparse_int N,
                             aoclsparse_int nnz, aoclsparse_int *col_ptr, aoclsparse_int *row_idx, T *val,
                             aoclsparse_matrix_data_type vt)
{
    if(!mat)
        return aoclsparse_status_invalid_pointer;
    *mat          = nullptr;
    int  sort     = 0;
    bool fulldiag = false;
    // the CSC arrays are checked as the CSR of the N x M transpose (auxiliary.cpp: create_csc_t)
    MI355_TRY(mat_check(N, M, nnz, col_ptr, row_idx, val, 0, base, sort, fulldiag));
    aoclsparse_matrix A = nullptr;
    MI355_TRY(new_owned_csr(&A, M, N, nnz, base, vt));
    aoclsparse_status st = csr2csc<T>(N, M, nnz, base, base, col_ptr, row_idx, val, A->user.ind, A->user.ptr,
                                      static_cast<T *>(A->user.val));
    if(st != aoclsparse_status_success)
    {
        aoclsparse_destroy(&A);
        return st;
    }
    // sort class / full diagonal of the CSR the executors run on (a stable transpose yields sorted rows)
    if(mat_check(M, N, nnz, A->user.ptr, A->user.ind, A->user.val, 0, base, sort, fulldiag) == aoclsparse_status_success)
        A->sort = sort, A->fulldiag = fulldiag;
    A->csc_ptr = col_ptr, A->csc_ind = row_idx, A->csc_val = val; // for export_?csc / order_mat / mutation
    *mat = A;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status create_coo(aoclsparse_matrix *mat, aoclsparse_index_base base, aoclsparse_int M, aoclsparse_int N,
                             aoclsparse_int nnz, aoclsparse_int *row_ind, aoclsparse_int *col_ind, T *val,
                             aoclsparse_matrix_data_type vt)
{
    if(!mat)
        return aoclsparse_status_invalid_pointer;
    *mat = nullptr;
    if(M < 0 || N < 0 || nnz < 0)
        return aoclsparse_status_invalid_size;
    if(!row_ind || !col_ind || !val)
        return aoclsparse_status_invalid_pointer;
    for(aoclsparse_int i = 0; i < nnz; i++)
        if(row_ind[i] < base || row_ind[i] >= M + base || col_ind[i] < base || col_ind[i] >= N + base)
            return aoclsparse_status_invalid_index_value;
    _aoclsparse_matrix *A = new(std::nothrow) _aoclsparse_matrix;
    if(!A)
        return aoclsparse_status_memory_error;
    A->m = M, A->n = N, A->nnz = nnz, A->base = base, A->val_type = vt;
    A->input_format = aoclsparse_coo_mat;
    A->coo_row = row_ind, A->coo_col = col_ind, A->coo_val = val; // aliased; no executor runs on COO
    *mat = A;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status export_csc(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                             aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **col_ptr,
                             aoclsparse_int **row_ind, T **val, aoclsparse_matrix_data_type vt)
{
    if(!mat || !base || !m || !n || !nnz || !col_ptr || !row_ind || !val)
        return aoclsparse_status_invalid_pointer;
    if(mat->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(!mat->csc_ptr)
        return aoclsparse_status_invalid_value; // not created from CSC arrays
    *col_ptr = mat->csc_ptr, *row_ind = mat->csc_ind, *val = static_cast<T *>(mat->csc_val);
    *m = mat->m, *n = mat->n, *nnz = mat->nnz, *base = mat->base;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status export_coo(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                             aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **row_ptr,
                             aoclsparse_int **col_ptr, T **val, aoclsparse_matrix_data_type vt)
{
    if(!mat || !base || !m || !n || !nnz || !row_ptr || !col_ptr || !val)
        return aoclsparse_status_invalid_pointer;
    if(mat->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(!mat->coo_row || !mat->coo_col || !mat->coo_val)
        return aoclsparse_status_invalid_value;
    *row_ptr = mat->coo_row, *col_ptr = mat->coo_col, *val = static_cast<T *>(mat->coo_val);
    *m = mat->m, *n = mat->n, *nnz = mat->nnz, *base = mat->base;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status convert_csr(const aoclsparse_matrix src, aoclsparse_operation op, aoclsparse_matrix *dest,
                              aoclsparse_matrix_data_type vt)
{
    const bool           tr = op != aoclsparse_operation_none;
    const aoclsparse_int md = tr ? src->n : src->m, nd = tr ? src->m : src->n;
    aoclsparse_matrix    D  = nullptr;
    MI355_TRY(new_owned_csr(&D, md, nd, src->nnz, src->base, vt));
    T *dv = static_cast<T *>(D->user.val);
    if(src->input_format == aoclsparse_coo_mat)
    {
        // transposing a COO matrix = swapping its index arrays (convert.cpp:1070-1096)
        coo2csr<T>(md, src->nnz, src->base, tr ? src->coo_col : src->coo_row, tr ? src->coo_row : src->coo_col,
                   static_cast<const T *>(src->coo_val), D->user.ptr, D->user.ind, dv);
    }
    else if(!tr)
    {
        std::memcpy(D->user.ptr, src->user.ptr, sizeof(aoclsparse_int) * ((size_t)src->m + 1));
        std::memcpy(D->user.ind, src->user.ind, sizeof(aoclsparse_int) * (size_t)src->nnz);
        std::memcpy(dv, src->user.val, sizeof(T) * (size_t)src->nnz);
    }
    else
    {
        aoclsparse_status st = csr2csc<T>(src->m, src->n, src->nnz, src->base, src->base, src->user.ptr, src->user.ind,
                                          static_cast<const T *>(src->user.val), D->user.ind, D->user.ptr, dv);
        if(st != aoclsparse_status_success)
        {
            aoclsparse_destroy(&D);
            return st;
        }
    }
    if(op == aoclsparse_operation_conjugate_transpose)
        for(aoclsparse_int i = 0; i < src->nnz; i++)
            dv[i] = conj_of(dv[i]); // identity for real types
    // the reference leaves sort / fulldiag at their defaults
    // (aoclsparse_init_mat): unknown, so that ILU-type routines re-check; here they are computed
    bool sorted = false, fd = false;
    int  sort   = 0;
    if(mat_check(md, nd, src->nnz, D->user.ptr, D->user.ind, dv, 0, src->base, sort, fd) == aoclsparse_status_success)
        D->sort = sort, D->fulldiag = fd;
    (void)sorted;
    *dest = D;
    return aoclsparse_status_success;
}

// per-row sort of (index, value) pairs in place; the same ordering rule as the clean copy (csr_util.hpp:100-159)
template <typename T>
aoclsparse_status sort_rows(aoclsparse_int m, aoclsparse_int base, const aoclsparse_int *ptr, aoclsparse_int *ind, T *val)
{
    std::vector<aoclsparse_int> perm, ti;
    std::vector<T>              tv;
    try
    {
        for(aoclsparse_int i = 0; i < m; i++)
        {
            const aoclsparse_int s = ptr[i] - base, len = ptr[i + 1] - base - s;
            if(len < 2 || std::is_sorted(ind + s, ind + s + len))
                continue;
            perm.resize((size_t)len);
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(), [&](aoclsparse_int a, aoclsparse_int c) { return ind[s + a] < ind[s + c]; });
            ti.assign(ind + s, ind + s + len);
            tv.assign(val + s, val + s + len);
            for(aoclsparse_int t = 0; t < len; t++)
                ind[s + t] = ti[perm[t]], val[s + t] = tv[perm[t]];
        }
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

} // namespace

namespace mi355
{

// extra/aoclsparse_auxiliary.hpp:356-386 (aoclsparse_set_coo_value): first matching coordinate
aoclsparse_status coo_set_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx, const void *val)
{
    const size_t vs = val_size(A->val_type);
    for(aoclsparse_int i = 0; i < A->nnz; i++)
        if(A->coo_row[i] == row_idx && A->coo_col[i] == col_idx)
        {
            std::memcpy(static_cast<char *>(A->coo_val) + vs * (size_t)i, val, vs);
            return aoclsparse_status_success;
        }
    return aoclsparse_status_invalid_index_value;
}

void csc_set_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx, const void *val)
{
    const size_t         vs = val_size(A->val_type);
    const aoclsparse_int c  = col_idx - A->base;
    for(aoclsparse_int p = A->csc_ptr[c] - A->base; p < A->csc_ptr[c + 1] - A->base; p++)
        if(A->csc_ind[p] == row_idx)
        {
            std::memcpy(static_cast<char *>(A->csc_val) + vs * (size_t)p, val, vs);
            return;
        }
}

aoclsparse_status csc_refresh_csr(aoclsparse_matrix A)
{
    return dispatch_value_type(A->val_type, [&](auto tag) {
        using T = decltype(tag);
        return csr2csc<T>(A->n, A->m, A->nnz, A->base, A->base, A->csc_ptr, A->csc_ind, static_cast<const T *>(A->csc_val),
                          A->user.ind, A->user.ptr, static_cast<T *>(A->user.val));
    });
}

} // namespace mi355

extern "C" {

aoclsparse_status aoclsparse_create_dcsc(aoclsparse_matrix *mat, aoclsparse_index_base base, aoclsparse_int M,
                                         aoclsparse_int N, aoclsparse_int nnz, aoclsparse_int *col_ptr,
                                         aoclsparse_int *row_idx, double *val)
{
    return create_csc<double>(mat, base, M, N, nnz, col_ptr, row_idx, val, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_create_scsc(aoclsparse_matrix *mat, aoclsparse_index_base base, aoclsparse_int M,
                                         aoclsparse_int N, aoclsparse_int nnz, aoclsparse_int *col_ptr,
                                         aoclsparse_int *row_idx, float *val)
{
    return create_csc<float>(mat, base, M, N, nnz, col_ptr, row_idx, val, aoclsparse_smat);
}
aoclsparse_status aoclsparse_create_dcoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                         const aoclsparse_int M, const aoclsparse_int N, const aoclsparse_int nnz,
                                         aoclsparse_int *row_ind, aoclsparse_int *col_ind, double *val)
{
    return create_coo<double>(mat, base, M, N, nnz, row_ind, col_ind, val, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_create_scoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                         const aoclsparse_int M, const aoclsparse_int N, const aoclsparse_int nnz,
                                         aoclsparse_int *row_ind, aoclsparse_int *col_ind, float *val)
{
    return create_coo<float>(mat, base, M, N, nnz, row_ind, col_ind, val, aoclsparse_smat);
}

aoclsparse_status aoclsparse_create_ccsc(aoclsparse_matrix *mat, aoclsparse_index_base base, aoclsparse_int M,
                                         aoclsparse_int N, aoclsparse_int nnz, aoclsparse_int *col_ptr,
                                         aoclsparse_int *row_idx, aoclsparse_float_complex *val)
{
    return create_csc<cfloat>(mat, base, M, N, nnz, col_ptr, row_idx, reinterpret_cast<cfloat *>(val), aoclsparse_cmat);
}
aoclsparse_status aoclsparse_create_zcsc(aoclsparse_matrix *mat, aoclsparse_index_base base, aoclsparse_int M,
                                         aoclsparse_int N, aoclsparse_int nnz, aoclsparse_int *col_ptr,
                                         aoclsparse_int *row_idx, aoclsparse_double_complex *val)
{
    return create_csc<cdouble>(mat, base, M, N, nnz, col_ptr, row_idx, reinterpret_cast<cdouble *>(val), aoclsparse_zmat);
}
aoclsparse_status aoclsparse_create_ccoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                         const aoclsparse_int M, const aoclsparse_int N, const aoclsparse_int nnz,
                                         aoclsparse_int *row_ind, aoclsparse_int *col_ind, aoclsparse_float_complex *val)
{
    return create_coo<cfloat>(mat, base, M, N, nnz, row_ind, col_ind, reinterpret_cast<cfloat *>(val), aoclsparse_cmat);
}
aoclsparse_status aoclsparse_create_zcoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                         const aoclsparse_int M, const aoclsparse_int N, const aoclsparse_int nnz,
                                         aoclsparse_int *row_ind, aoclsparse_int *col_ind, aoclsparse_double_complex *val)
{
    return create_coo<cdouble>(mat, base, M, N, nnz, row_ind, col_ind, reinterpret_cast<cdouble *>(val), aoclsparse_zmat);
}
aoclsparse_status aoclsparse_export_ccsc(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **col_ptr,
                                         aoclsparse_int **row_ind, aoclsparse_float_complex **val)
{
    return export_csc<cfloat>(mat, base, m, n, nnz, col_ptr, row_ind, reinterpret_cast<cfloat **>(val), aoclsparse_cmat);
}
aoclsparse_status aoclsparse_export_zcsc(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **col_ptr,
                                         aoclsparse_int **row_ind, aoclsparse_double_complex **val)
{
    return export_csc<cdouble>(mat, base, m, n, nnz, col_ptr, row_ind, reinterpret_cast<cdouble **>(val), aoclsparse_zmat);
}
aoclsparse_status aoclsparse_export_ccoo(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **row_ptr,
                                         aoclsparse_int **col_ptr, aoclsparse_float_complex **val)
{
    return export_coo<cfloat>(mat, base, m, n, nnz, row_ptr, col_ptr, reinterpret_cast<cfloat **>(val), aoclsparse_cmat);
}
aoclsparse_status aoclsparse_export_zcoo(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **row_ptr,
                                         aoclsparse_int **col_ptr, aoclsparse_double_complex **val)
{
    return export_coo<cdouble>(mat, base, m, n, nnz, row_ptr, col_ptr, reinterpret_cast<cdouble **>(val), aoclsparse_zmat);
}

aoclsparse_status aoclsparse_export_dcsc(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **col_ptr,
                                         aoclsparse_int **row_ind, double **val)
{
    return export_csc<double>(mat, base, m, n, nnz, col_ptr, row_ind, val, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_export_scsc(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **col_ptr,
                                         aoclsparse_int **row_ind, float **val)
{
    return export_csc<float>(mat, base, m, n, nnz, col_ptr, row_ind, val, aoclsparse_smat);
}
aoclsparse_status aoclsparse_export_dcoo(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **row_ptr,
                                         aoclsparse_int **col_ptr, double **val)
{
    return export_coo<double>(mat, base, m, n, nnz, row_ptr, col_ptr, val, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_export_scoo(const aoclsparse_matrix mat, aoclsparse_index_base *base, aoclsparse_int *m,
                                         aoclsparse_int *n, aoclsparse_int *nnz, aoclsparse_int **row_ptr,
                                         aoclsparse_int **col_ptr, float **val)
{
    return export_coo<float>(mat, base, m, n, nnz, row_ptr, col_ptr, val, aoclsparse_smat);
}

aoclsparse_status aoclsparse_convert_csr(const aoclsparse_matrix src_mat, const aoclsparse_operation op,
                                         aoclsparse_matrix *dest_mat)
{
    if(!src_mat || !dest_mat)
        return aoclsparse_status_invalid_pointer;
    *dest_mat = nullptr;
    if(src_mat->input_format == aoclsparse_coo_mat ? !src_mat->coo_row : !src_mat->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(src_mat->input_format != aoclsparse_coo_mat && src_mat->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    return dispatch_value_type(src_mat->val_type, [&](auto tag) {
        return convert_csr<decltype(tag)>(src_mat, op, dest_mat, src_mat->val_type);
    });
}

aoclsparse_status aoclsparse_order_mat(aoclsparse_matrix mat)
{
    if(!mat)
        return aoclsparse_status_invalid_pointer;
    if(mat->m < 0 || mat->n < 0 || mat->nnz < 0)
        return aoclsparse_status_invalid_value;
    if(mat->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(mat->m == 0 || mat->n == 0 || mat->nnz == 0)
        return aoclsparse_status_success;
    if(!mat->user.ptr || !mat->user.ind || !mat->user.val)
        return aoclsparse_status_invalid_pointer;
    std::unique_lock<std::shared_mutex> w(mat->guard);
    aoclsparse_status                   st = dispatch_value_type(mat->val_type, [&](auto tag) {
        using T              = decltype(tag);
        aoclsparse_status rc = aoclsparse_status_success;
        if(mat->csc_ptr) // the caller's CSC arrays are "the first representation" there (auxiliary.cpp:1268-1294)
            rc = sort_rows<T>(mat->n, mat->base, mat->csc_ptr, mat->csc_ind, static_cast<T *>(mat->csc_val));
        if(rc == aoclsparse_status_success)
            rc = sort_rows<T>(mat->m, mat->base, mat->user.ptr, mat->user.ind, static_cast<T *>(mat->user.val));
        return rc;
    });
    if(st != aoclsparse_status_success)
        return st;
    int  sort = 0;
    bool fd   = false;
    if(mat_check(mat->m, mat->n, mat->nnz, mat->user.ptr, mat->user.ind, mat->user.val, 0, mat->base, sort, fd)
       == aoclsparse_status_success)
        mat->sort = sort, mat->fulldiag = fd;
    drop_derived_state(mat); // every copy made of the unsorted arrays (clean CSR, device mirrors, plans)
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_dcsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                      const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                      const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                      const double *csr_val, aoclsparse_int *csc_row_ind, aoclsparse_int *csc_col_ptr,
                                      double *csc_val)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    return csr2csc<double>(m, n, nnz, descr->base, baseCSC, csr_row_ptr, csr_col_ind, csr_val, csc_row_ind, csc_col_ptr,
                           csc_val);
}
aoclsparse_status aoclsparse_scsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                      const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                      const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                      const float *csr_val, aoclsparse_int *csc_row_ind, aoclsparse_int *csc_col_ptr,
                                      float *csc_val)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    return csr2csc<float>(m, n, nnz, descr->base, baseCSC, csr_row_ptr, csr_col_ind, csr_val, csc_row_ind, csc_col_ptr,
                          csc_val);
}
aoclsparse_status aoclsparse_ccsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                      const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                      const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                      const aoclsparse_float_complex *csr_val, aoclsparse_int *csc_row_ind,
                                      aoclsparse_int *csc_col_ptr, aoclsparse_float_complex *csc_val)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    return csr2csc<cfloat>(m, n, nnz, descr->base, baseCSC, csr_row_ptr, csr_col_ind, reinterpret_cast<const cfloat *>(csr_val),
                           csc_row_ind, csc_col_ptr, reinterpret_cast<cfloat *>(csc_val));
}
aoclsparse_status aoclsparse_zcsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                      const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                      const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                      const aoclsparse_double_complex *csr_val, aoclsparse_int *csc_row_ind,
                                      aoclsparse_int *csc_col_ptr, aoclsparse_double_complex *csc_val)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    return csr2csc<cdouble>(m, n, nnz, descr->base, baseCSC, csr_row_ptr, csr_col_ind,
                            reinterpret_cast<const cdouble *>(csr_val), csc_row_ind, csc_col_ptr,
                            reinterpret_cast<cdouble *>(csc_val));
}

} // extern "C"
