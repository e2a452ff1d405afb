// solvers_api.cpp -- composite routines on top of the SpMV / TRSV executors: symmetric Gauss-Seidel
// (aoclsparse_?symgs, ?symgs_mv) and the ILU(0) smoother (aoclsparse_?ilu_smoother).
//
// Both keep every intermediate vector in HBM: host operands are staged once on entry, the executors are
// chained on the library stream under a DeviceScope, and only the results travel back.
//   symgs : solvers/aoclsparse_symgs.hpp:62-258 (algorithm), :264-394 (checks)
//   ilu   : solvers/aoclsparse_ilu.hpp:33-139, solvers/aoclsparse_ilu0.hpp:34-199, analysis.cpp:390-425
#include "internal.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

using namespace mi355;

namespace
{

// typed access to the public executors (the composites are written once for float and double)
inline aoclsparse_status exec_mv(aoclsparse_operation op, const double *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr d, const double *x, const double *beta, double *y)
{
    return aoclsparse_dmv(op, alpha, A, d, x, beta, y);
}
inline aoclsparse_status exec_mv(aoclsparse_operation op, const float *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr d, const float *x, const float *beta, float *y)
{
    return aoclsparse_smv(op, alpha, A, d, x, beta, y);
}
inline aoclsparse_status exec_mv(aoclsparse_operation op, const cdouble *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr d, const cdouble *x, const cdouble *beta, cdouble *y)
{
    return aoclsparse_zmv(op, reinterpret_cast<const aoclsparse_double_complex *>(alpha), A, d,
                          reinterpret_cast<const aoclsparse_double_complex *>(x),
                          reinterpret_cast<const aoclsparse_double_complex *>(beta),
                          reinterpret_cast<aoclsparse_double_complex *>(y));
}
inline aoclsparse_status exec_mv(aoclsparse_operation op, const cfloat *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr d, const cfloat *x, const cfloat *beta, cfloat *y)
{
    return aoclsparse_cmv(op, reinterpret_cast<const aoclsparse_float_complex *>(alpha), A, d,
                          reinterpret_cast<const aoclsparse_float_complex *>(x),
                          reinterpret_cast<const aoclsparse_float_complex *>(beta),
                          reinterpret_cast<aoclsparse_float_complex *>(y));
}
inline aoclsparse_status exec_trsv(aoclsparse_operation op, cdouble alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr d, const cdouble *b, cdouble *x, aoclsparse_int kid)
{
    return aoclsparse_ztrsv_kid(op, aoclsparse_double_complex{alpha.re, alpha.im}, A, d,
                                reinterpret_cast<const aoclsparse_double_complex *>(b),
                                reinterpret_cast<aoclsparse_double_complex *>(x), kid);
}
inline aoclsparse_status exec_trsv(aoclsparse_operation op, cfloat alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr d, const cfloat *b, cfloat *x, aoclsparse_int kid)
{
    return aoclsparse_ctrsv_kid(op, aoclsparse_float_complex{alpha.re, alpha.im}, A, d,
                                reinterpret_cast<const aoclsparse_float_complex *>(b),
                                reinterpret_cast<aoclsparse_float_complex *>(x), kid);
}
// w = x - y
template <typename T>
inline aoclsparse_status exec_diff(hipStream_t s, aoclsparse_int n, const T *x, const T *y, T *w)
{
    return launch_waxpby<T>(s, n, T(1), x, T(-1), y, w);
}
inline aoclsparse_status exec_diff(hipStream_t s, aoclsparse_int n, const cdouble *x, const cdouble *y, cdouble *w)
{
    return launch_cdiff<double>(s, n, x, y, w);
}
inline aoclsparse_status exec_diff(hipStream_t s, aoclsparse_int n, const cfloat *x, const cfloat *y, cfloat *w)
{
    return launch_cdiff<float>(s, n, x, y, w);
}
inline aoclsparse_status exec_trsv(aoclsparse_operation op, double alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr d, const double *b, double *x, aoclsparse_int kid)
{
    return aoclsparse_dtrsv_kid(op, alpha, A, d, b, x, kid);
}
inline aoclsparse_status exec_trsv(aoclsparse_operation op, float alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr d, const float *b, float *x, aoclsparse_int kid)
{
    return aoclsparse_strsv_kid(op, alpha, A, d, b, x, kid);
}
inline aoclsparse_status create_csr(aoclsparse_matrix *M, aoclsparse_index_base b, aoclsparse_int m, aoclsparse_int n,
                                    aoclsparse_int nnz, aoclsparse_int *p, aoclsparse_int *c, double *v)
{
    return aoclsparse_create_dcsr(M, b, m, n, nnz, p, c, v);
}
inline aoclsparse_status create_csr(aoclsparse_matrix *M, aoclsparse_index_base b, aoclsparse_int m, aoclsparse_int n,
                                    aoclsparse_int nnz, aoclsparse_int *p, aoclsparse_int *c, float *v)
{
    return aoclsparse_create_scsr(M, b, m, n, nnz, p, c, v);
}

inline aoclsparse_status create_csr(aoclsparse_matrix *M, aoclsparse_index_base b, aoclsparse_int m, aoclsparse_int n,
                                    aoclsparse_int nnz, aoclsparse_int *p, aoclsparse_int *c, cdouble *v)
{
    return aoclsparse_create_zcsr(M, b, m, n, nnz, p, c, reinterpret_cast<aoclsparse_double_complex *>(v));
}
inline aoclsparse_status create_csr(aoclsparse_matrix *M, aoclsparse_index_base b, aoclsparse_int m, aoclsparse_int n,
                                    aoclsparse_int nnz, aoclsparse_int *p, aoclsparse_int *c, cfloat *v)
{
    return aoclsparse_create_ccsr(M, b, m, n, nnz, p, c, reinterpret_cast<aoclsparse_float_complex *>(v));
}

// A caller's vector made device-resident for the duration of one composite call.
template <typename T>
struct Vec
{
    T    *dev  = nullptr;
    T    *host = nullptr;
    bool  staged = false;
    aoclsparse_status in(Runtime &rt, DeviceBuffer &buf, const T *p, aoclsparse_int n, bool copy)
    {
        if(rt.is_device_pointer(p))
        {
            dev = const_cast<T *>(p);
            return aoclsparse_status_success;
        }
        staged = true;
        host   = const_cast<T *>(p);
        aoclsparse_status st = buf.alloc(sizeof(T) * (size_t)n);
        if(st != aoclsparse_status_success)
            return st;
        dev = buf.as<T>();
        if(copy)
            MI355_HIP_TRY(hipMemcpyAsync(dev, p, sizeof(T) * (size_t)n, hipMemcpyHostToDevice, rt.stream()));
        return aoclsparse_status_success;
    }
    aoclsparse_status out(Runtime &rt, aoclsparse_int n)
    {
        if(staged)
            MI355_HIP_TRY(hipMemcpyAsync(host, dev, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost, rt.stream()));
        return aoclsparse_status_success;
    }
};

#define MI355_TRY(expr)                          \
    do                                           \
    {                                            \
        aoclsparse_status st__ = (expr);         \
        if(st__ != aoclsparse_status_success)    \
            return st__;                         \
    } while(0)

// ---- forward SOR sweep (solvers/aoclsparse_sorv.hpp:116-227) ------------------------------------------------------
// sorv.hpp:32-75: every row holds exactly one diagonal entry and it is non-zero
template <typename T>
bool sorv_full_diag(const HostCsr &h)
{
    const T *v = static_cast<const T *>(h.val);
    for(aoclsparse_int i = 0; i < h.m; i++)
    {
        bool found = false;
        for(aoclsparse_int j = h.ptr[i] - h.base; j < h.ptr[i + 1] - h.base; j++)
            if(h.ind[j] - h.base == i)
            {
                if(found || v[j] == T(0))
                    return false;
                found = true;
            }
        if(!found)
            return false;
    }
    return true;
}

template <typename T>
aoclsparse_status sorv_t(aoclsparse_sor_type sor_type, const aoclsparse_mat_descr descr, aoclsparse_matrix A, T omega,
                         T alpha, T *x, const T *b, aoclsparse_matrix_data_type vt)
{
    if(!A || !descr || !x || !b)
        return aoclsparse_status_invalid_pointer;
    if(!A->user.ptr && A->input_format == aoclsparse_csr_mat)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    if(A->m != A->n)
        return aoclsparse_status_invalid_size;
    if(A->m == 0)
        return aoclsparse_status_success;
    if(A->m < 0 || A->nnz < 0)
        return aoclsparse_status_invalid_value;
    if(A->input_format != aoclsparse_csr_mat || descr->type != aoclsparse_matrix_type_general)
        return aoclsparse_status_not_implemented;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(sor_type != aoclsparse_sor_forward)
        return sor_type == aoclsparse_sor_backward || sor_type == aoclsparse_sor_symmetric
                   ? aoclsparse_status_not_implemented
                   : aoclsparse_status_invalid_value;
    if(!sorv_full_diag<T>(A->user))
        return aoclsparse_status_invalid_value;

    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock); // the handle's workspaces are shared
    const aoclsparse_int m = A->m;
    // level sets of the strict lower triangle (the solve's analysis) and the user's CSR in HBM
    MI355_TRY(ensure_trsv(A, false, false));
    DeviceCsr *d = nullptr;
    SpmvPlan  *p = nullptr;
    MI355_TRY(ensure_spmv(A, false, d, p));
    Vec<T> vb, vx;
    MI355_TRY(workspace_stream_guard(A, rt.stream()));
    MI355_TRY(vb.in(rt, A->work[2], b, m, true));
    MI355_TRY(vx.in(rt, A->work[3], x, m, true));
    MI355_TRY(A->work[0].alloc(sizeof(T) * (size_t)m));
    hipStream_t s = rt.stream();
    // x = alpha * x (exact zeros for alpha == 0, sorv.hpp:212-222), snapshot, then level by level
    MI355_TRY(launch_dense_scale<T>(s, vx.dev, m, 1, m, alpha, alpha == T(0)));
    MI355_HIP_TRY(hipMemcpyAsync(A->work[0].ptr, vx.dev, sizeof(T) * (size_t)m, hipMemcpyDeviceToDevice, s));
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        const TrsvPlan                     &tp = A->trsv_plan[0];
        const aoclsparse_int               *rows = tp.rowmap.as<aoclsparse_int>();
        for(aoclsparse_int l = 0; l < tp.nlevels; l++)
            MI355_TRY(launch_sorv_level<T>(s, rows + tp.level_ptr[l], tp.level_ptr[l + 1] - tp.level_ptr[l], d->base,
                                           d->ptr.as<aoclsparse_int>(), d->ind.as<aoclsparse_int>(), d->val.as<T>(), omega,
                                           vx.dev, A->work[0].as<T>(), vb.dev));
    }
    MI355_TRY(vx.out(rt, m));
    if(vx.staged)
        MI355_HIP_TRY(hipStreamSynchronize(s));
    return aoclsparse_status_success;
}

// ---- symmetric Gauss-Seidel ------------------------------------------------------------------------
template <typename T>
aoclsparse_status symgs_t(aoclsparse_operation trans, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                          const T alpha, const T *b, T *x, T *y, aoclsparse_int kid, bool fuse_mv,
                          aoclsparse_matrix_data_type vt)
{
    // symgs.hpp:276-352, same order
    if(!x || !b)
        return aoclsparse_status_invalid_pointer;
    if(fuse_mv && !y)
        return aoclsparse_status_invalid_pointer;
    if(!A || !descr)
        return aoclsparse_status_invalid_pointer;
    if(!A->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(descr->base != aoclsparse_index_base_zero && descr->base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    if(trans != aoclsparse_operation_none && trans != aoclsparse_operation_transpose
       && trans != aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_invalid_value;
    if(descr->fill_mode != aoclsparse_fill_mode_lower && descr->fill_mode != aoclsparse_fill_mode_upper)
        return aoclsparse_status_invalid_value;
    if(descr->diag_type == aoclsparse_diag_type_unit)
        return aoclsparse_status_not_implemented;
    if(descr->type == aoclsparse_matrix_type_general && trans == aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_not_implemented;
    if(descr->type != aoclsparse_matrix_type_symmetric && descr->type != aoclsparse_matrix_type_triangular
       && descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_hermitian)
        return aoclsparse_status_invalid_value;
    if(A->m < 0 || A->nnz < 0 || A->n < 0)
        return aoclsparse_status_invalid_size;
    if(A->m == 0 || A->n == 0 || A->nnz == 0)
        return aoclsparse_status_success;
    if(A->m != A->n)
        return aoclsparse_status_invalid_size;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    MI355_TRY(csr_optimize(A));
    if(!A->opt_csr_full_diag) // unit diagonals were refused above
        return aoclsparse_status_invalid_value;
    (void)kid; // symgs.hpp:354-381: every kid runs the reference composition

    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock); // the handle's workspaces are shared
    const aoclsparse_int m = A->m;
    Vec<T>               vb, vx, vy;
    MI355_TRY(workspace_stream_guard(A, rt.stream()));
    MI355_TRY(vb.in(rt, A->work[2], b, m, true));
    MI355_TRY(vx.in(rt, A->work[3], x, m, true)); // x carries the initial guess
    if(fuse_mv)
        MI355_TRY(vy.in(rt, A->work[4], y, m, false));
    const T one = T(1), zero = T(0);
    {
        DeviceScope scope;
        if(descr->type == aoclsparse_matrix_type_triangular)
        {
            // symgs.hpp:128-149: a single solve with the given triangle (+ the product)
            MI355_TRY(exec_trsv(trans, one, A, descr, vb.dev, vx.dev, -1));
            if(fuse_mv)
                MI355_TRY(exec_mv(trans, &one, A, descr, vx.dev, &zero, vy.dev));
        }
        else
        {
            MI355_TRY(A->work[0].alloc(sizeof(T) * (size_t)m));
            MI355_TRY(A->work[1].alloc(sizeof(T) * (size_t)m));
            T *r = A->work[0].as<T>(), *q = A->work[1].as<T>();
            // which stored triangle plays L and which U, and under which operation (symgs.hpp:151-190)
            aoclsparse_operation u_trans = aoclsparse_operation_transpose, l_trans = aoclsparse_operation_none;
            aoclsparse_fill_mode u_fill = aoclsparse_fill_mode_lower, l_fill = aoclsparse_fill_mode_lower;
            if(descr->type == aoclsparse_matrix_type_hermitian)
                u_trans = aoclsparse_operation_conjugate_transpose;
            if(descr->type == aoclsparse_matrix_type_symmetric && descr->fill_mode == aoclsparse_fill_mode_upper)
            {
                u_fill = l_fill = aoclsparse_fill_mode_upper;
                u_trans = aoclsparse_operation_none, l_trans = aoclsparse_operation_transpose;
            }
            else if(descr->type == aoclsparse_matrix_type_general && trans == aoclsparse_operation_none)
            {
                u_trans = l_trans = aoclsparse_operation_none;
                u_fill            = aoclsparse_fill_mode_upper;
            }
            else if(descr->type == aoclsparse_matrix_type_general && trans == aoclsparse_operation_transpose)
            {
                u_trans = l_trans = aoclsparse_operation_transpose;
                l_fill = aoclsparse_fill_mode_upper, u_fill = aoclsparse_fill_mode_lower;
            }
            else if(descr->type == aoclsparse_matrix_type_hermitian && descr->fill_mode == aoclsparse_fill_mode_upper)
            {
                u_fill = l_fill = aoclsparse_fill_mode_upper;
                u_trans = aoclsparse_operation_none, l_trans = aoclsparse_operation_conjugate_transpose;
            }
            _aoclsparse_mat_descr d = *descr;
            d.type                  = aoclsparse_matrix_type_triangular;
            auto with = [&](aoclsparse_fill_mode f, aoclsparse_diag_type dt) {
                d.fill_mode = f, d.diag_type = dt;
                return &d;
            };
            // 1: (L + D) x1 = b - alpha U x0
            MI355_TRY(exec_mv(u_trans, &alpha, A, with(u_fill, aoclsparse_diag_type_zero), vx.dev, &zero, q));
            MI355_TRY(exec_diff(rt.stream(), m, vb.dev, q, r));
            MI355_TRY(exec_trsv(l_trans, one, A, with(l_fill, aoclsparse_diag_type_non_unit), r, q, -1));
            // 2: (U + D) x = b - L x1
            MI355_TRY(exec_mv(l_trans, &one, A, with(l_fill, aoclsparse_diag_type_zero), q, &zero, r));
            MI355_TRY(exec_diff(rt.stream(), m, vb.dev, r, q));
            MI355_TRY(exec_trsv(u_trans, one, A, with(u_fill, aoclsparse_diag_type_non_unit), q, vx.dev, -1));
            // 3: y = op(A) x
            if(fuse_mv)
                MI355_TRY(exec_mv(trans, &one, A, descr, vx.dev, &zero, vy.dev));
        }
    }
    MI355_TRY(vx.out(rt, m));
    if(fuse_mv)
        MI355_TRY(vy.out(rt, m));
    if(vx.staged || vy.staged)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return aoclsparse_status_success;
}

// ---- ILU(0) ------------------------------------------------------------------------------------------
template <typename T>
bool near_zero(T v)
{
    // aoclsparse_is_nearzero, extra/aoclsparse_utils.hpp:598-613: |v| <= 1e-2 * 2 * macheps
    return std::fabs(v) <= T(1e-2) * T(2) * std::numeric_limits<T>::epsilon();
}

// ILU(0) on the user's pattern (ilu0.hpp:34-111), on the GPU: dependency levels of the strictly lower
// pattern are computed on the host (integer work), then one launch per level eliminates the level's rows in
// place in a device copy of the values (ilu_kernels.hip); the factors come back to `host_val` for
// *precond_csr_val and for the factor handle.  Bit-identical to the serial IKJ loop.
template <typename T>
aoclsparse_status ilu0_factorize(aoclsparse_matrix A, T *host_val)
{
    const aoclsparse_int  n = A->n, base = A->base, nnz = A->nnz;
    const aoclsparse_int *ptr = A->user.ptr, *ind = A->user.ind;
    std::vector<aoclsparse_int> level, order, lptr;
    aoclsparse_int              nlev = 0, maxlen = 0;
    try
    {
        level.assign((size_t)n, 0);
        for(aoclsparse_int i = 0; i < n; i++)
        {
            aoclsparse_int lv = 0;
            for(aoclsparse_int j = ptr[i] - base; j < ptr[i + 1] - base; j++)
            {
                const aoclsparse_int k = ind[j] - base;
                if(k >= i)
                    break; // the k-loop of the reference stops at the first column >= i (:60-92)
                lv = std::max(lv, level[k] + 1);
            }
            level[i] = lv;
            nlev     = std::max(nlev, lv + 1);
            maxlen   = std::max(maxlen, ptr[i + 1] - ptr[i]);
        }
        lptr.assign((size_t)nlev + 1, 0);
        for(aoclsparse_int i = 0; i < n; i++)
            lptr[level[i] + 1]++;
        for(aoclsparse_int l = 0; l < nlev; l++)
            lptr[l + 1] += lptr[l];
        order.resize((size_t)n);
        std::vector<aoclsparse_int> fill(lptr.begin(), lptr.end() - 1);
        for(aoclsparse_int i = 0; i < n; i++)
            order[fill[level[i]]++] = i;
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    if((sizeof(T) + sizeof(aoclsparse_int)) * (size_t)maxlen > 60000) // one row must fit a workgroup's LDS
        return aoclsparse_status_not_implemented;

    Runtime &rt = Runtime::get();
    DeviceCsr *dcsr = nullptr;
    SpmvPlan  *plan = nullptr;
    MI355_TRY(ensure_spmv(A, false, dcsr, plan)); // the pattern (and current values) in HBM
    DeviceBuffer dval, ddiag, drows, derr;
    MI355_TRY(dval.alloc(sizeof(T) * (size_t)std::max<aoclsparse_int>(nnz, 1)));
    MI355_TRY(ddiag.alloc(sizeof(aoclsparse_int) * (size_t)n));
    MI355_TRY(derr.alloc(sizeof(int)));
    MI355_TRY(drows.upload(order.data(), sizeof(aoclsparse_int) * (size_t)n, rt.stream()));
    // the factor starts from the values captured by ilu_prepare (the matrix at hint/optimize time)
    MI355_HIP_TRY(hipMemcpyAsync(dval.ptr, host_val, sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice, rt.stream()));
    MI355_HIP_TRY(hipMemsetAsync(derr.ptr, 0, sizeof(int), rt.stream()));
    // real types, deep DAGs: one sync-free launch (rows wait for the rows they need; ilu_kernels.hip) instead of one launch per
    // level -- 5,505 launches cost 390 ms on the shell-like stand-in, the sync-free launch 139 ms; with two lower entries per row
    // (2-D Laplacian) the launches win, 20 vs 44 ms.
    bool syncfree = false;
    if constexpr(std::is_floating_point<T>::value)
        syncfree = nlev > 8 && (long long)nnz >= 16LL * n; // (short rows: a level's launch is cheaper than its hops)
    if(syncfree)
    {
        if constexpr(std::is_floating_point<T>::value)
        {
            DeviceBuffer dticket;
            MI355_TRY(dticket.alloc(sizeof(unsigned int)));
            MI355_HIP_TRY(hipMemsetAsync(dticket.ptr, 0, sizeof(unsigned int), rt.stream()));
            MI355_HIP_TRY(hipMemsetAsync(ddiag.ptr, 0xFF, sizeof(aoclsparse_int) * (size_t)n, rt.stream())); // -1: not finished
            MI355_TRY(launch_ilu0_syncfree<T>(rt.stream(), base, n, drows.as<aoclsparse_int>(), dcsr->ptr.as<aoclsparse_int>(),
                                              dcsr->ind.as<aoclsparse_int>(), dval.as<T>(), ddiag.as<aoclsparse_int>(),
                                              (int)maxlen, derr.as<int>(), dticket.as<unsigned int>()));
            MI355_HIP_TRY(hipStreamSynchronize(rt.stream())); // dticket goes out of scope
        }
    }
    else
        for(aoclsparse_int l = 0; l < nlev; l++)
            MI355_TRY(launch_ilu0_level<T>(rt.stream(), base, lptr[l + 1] - lptr[l], drows.as<aoclsparse_int>() + lptr[l],
                                           dcsr->ptr.as<aoclsparse_int>(), dcsr->ind.as<aoclsparse_int>(), dval.as<T>(),
                                           ddiag.as<aoclsparse_int>(), (int)maxlen, derr.as<int>()));
    int err = 0;
    MI355_HIP_TRY(hipMemcpyAsync(&err, derr.ptr, sizeof(int), hipMemcpyDeviceToHost, rt.stream()));
    MI355_HIP_TRY(hipMemcpyAsync(host_val, dval.ptr, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToHost, rt.stream()));
    MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return err ? aoclsparse_status_numerical_error : aoclsparse_status_success;
}

template <typename T>
aoclsparse_status ilu_smoother_t(aoclsparse_operation op, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                 T **precond_csr_val, T *x, const T *b, aoclsparse_matrix_data_type vt)
{
    // ilu.hpp:43-98, same order
    if(!descr || !A)
        return aoclsparse_status_invalid_pointer;
    if(!x || !b || !precond_csr_val)
        return aoclsparse_status_invalid_pointer;
    if(!A->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    if(op != aoclsparse_operation_none)
        return aoclsparse_status_not_implemented;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(descr->type != aoclsparse_matrix_type_general)
        return aoclsparse_status_not_implemented;
    if(A->sort != 1 && A->sort != 2) // fully or partially sorted (aoclsparse_matrix_sort)
        return aoclsparse_status_unsorted_input;
    if(!A->fulldiag)
        return aoclsparse_status_numerical_error;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(A->m < 0 || A->n < 0 || A->m != A->n)
        return aoclsparse_status_invalid_size;
    if(A->m == 0 || A->n == 0)
        return aoclsparse_status_success;
    {
        PhaseTimer pt("ilu: prepare");
        MI355_TRY(ilu_prepare(A));
    }

    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    *precond_csr_val = nullptr;
    if(!A->ilu_factorized)
    {
        {
            PhaseTimer pt("ilu: factorise on the GPU");
            MI355_TRY(ilu0_factorize<T>(A, static_cast<T *>(A->ilu_val)));
        }
        PhaseTimer pt("ilu: factor handle");
        // the factors as a matrix of their own: its level-scheduled TRSV plans are the smoother's solves
        MI355_TRY(create_csr(&A->ilu_factor, A->base, A->m, A->n, A->nnz, A->user.ptr, A->user.ind,
                             static_cast<T *>(A->ilu_val)));
        A->ilu_factorized = true;
    }
    *precond_csr_val = static_cast<T *>(A->ilu_val);

    const aoclsparse_int m = A->m;
    Vec<T>               vb, vx;
    MI355_TRY(workspace_stream_guard(A, rt.stream()));
    MI355_TRY(vb.in(rt, A->work[2], b, m, true));
    MI355_TRY(vx.in(rt, A->work[3], x, m, false));
    MI355_TRY(A->work[0].alloc(sizeof(T) * (size_t)m));
    {
        // ilu0.hpp:113-156: L y = b with the unit lower factor, then U x = y
        DeviceScope           scope;
        _aoclsparse_mat_descr d = *descr;
        d.type                  = aoclsparse_matrix_type_triangular;
        d.fill_mode = aoclsparse_fill_mode_lower, d.diag_type = aoclsparse_diag_type_unit;
        MI355_TRY(exec_trsv(aoclsparse_operation_none, T(1), A->ilu_factor, &d, vb.dev, A->work[0].as<T>(), -1));
        d.fill_mode = aoclsparse_fill_mode_upper, d.diag_type = aoclsparse_diag_type_non_unit;
        MI355_TRY(exec_trsv(aoclsparse_operation_none, T(1), A->ilu_factor, &d, A->work[0].as<T>(), vx.dev, -1));
    }
    MI355_TRY(vx.out(rt, m));
    if(vx.staged)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return aoclsparse_status_success;
}

} // namespace

namespace mi355
{

// analysis.cpp:390-425 (aoclsparse_optimize_ilu): the factor values start as a copy of A's values
aoclsparse_status ilu_prepare(aoclsparse_matrix A)
{
    if(!A)
        return aoclsparse_status_invalid_pointer;
    if(!A->user.val)
        return aoclsparse_status_invalid_pointer;
    if(A->ilu_ready)
        return aoclsparse_status_success;
    const size_t bytes = val_size(A->val_type) * (size_t)A->nnz;
    A->ilu_val         = std::malloc(bytes ? bytes : 1);
    if(!A->ilu_val)
        return aoclsparse_status_memory_error;
    std::memcpy(A->ilu_val, A->user.val, bytes);
    A->ilu_ready = true;
    return aoclsparse_status_success;
}

} // namespace mi355

extern "C" {

aoclsparse_status aoclsparse_cilu_smoother(aoclsparse_operation op, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                           aoclsparse_float_complex **precond_csr_val, const aoclsparse_float_complex *approx_inv_diag, aoclsparse_float_complex *x, const aoclsparse_float_complex *b)
{
    (void)approx_inv_diag; // unused by the reference as well
    return ilu_smoother_t<cfloat>(op, A, descr, reinterpret_cast<cfloat **>(precond_csr_val), reinterpret_cast<cfloat *>(x),
                                 reinterpret_cast<const cfloat *>(b), aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zilu_smoother(aoclsparse_operation op, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                           aoclsparse_double_complex **precond_csr_val, const aoclsparse_double_complex *approx_inv_diag, aoclsparse_double_complex *x, const aoclsparse_double_complex *b)
{
    (void)approx_inv_diag; // unused by the reference as well
    return ilu_smoother_t<cdouble>(op, A, descr, reinterpret_cast<cdouble **>(precond_csr_val), reinterpret_cast<cdouble *>(x),
                                 reinterpret_cast<const cdouble *>(b), aoclsparse_zmat);
}
aoclsparse_status aoclsparse_csymgs(aoclsparse_operation trans, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                    const aoclsparse_float_complex alpha, const aoclsparse_float_complex *b, aoclsparse_float_complex *x)
{
    return symgs_t<cfloat>(trans, A, descr, cfloat(alpha.real, alpha.imag), reinterpret_cast<const cfloat *>(b),
                          reinterpret_cast<cfloat *>(x), nullptr, -1, false, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_csymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                        const aoclsparse_mat_descr descr, const aoclsparse_float_complex alpha, const aoclsparse_float_complex *b, aoclsparse_float_complex *x,
                                        const aoclsparse_int kid)
{
    return symgs_t<cfloat>(trans, A, descr, cfloat(alpha.real, alpha.imag), reinterpret_cast<const cfloat *>(b),
                          reinterpret_cast<cfloat *>(x), nullptr, kid, false, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_csymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const aoclsparse_float_complex alpha, const aoclsparse_float_complex *b, aoclsparse_float_complex *x,
                                       aoclsparse_float_complex *y)
{
    return symgs_t<cfloat>(trans, A, descr, cfloat(alpha.real, alpha.imag), reinterpret_cast<const cfloat *>(b),
                          reinterpret_cast<cfloat *>(x), reinterpret_cast<cfloat *>(y), -1, true, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_csymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const aoclsparse_float_complex alpha, const aoclsparse_float_complex *b, aoclsparse_float_complex *x,
                                           aoclsparse_float_complex *y, const aoclsparse_int kid)
{
    return symgs_t<cfloat>(trans, A, descr, cfloat(alpha.real, alpha.imag), reinterpret_cast<const cfloat *>(b),
                          reinterpret_cast<cfloat *>(x), reinterpret_cast<cfloat *>(y), kid, true, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zsymgs(aoclsparse_operation trans, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                    const aoclsparse_double_complex alpha, const aoclsparse_double_complex *b, aoclsparse_double_complex *x)
{
    return symgs_t<cdouble>(trans, A, descr, cdouble(alpha.real, alpha.imag), reinterpret_cast<const cdouble *>(b),
                          reinterpret_cast<cdouble *>(x), nullptr, -1, false, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_zsymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                        const aoclsparse_mat_descr descr, const aoclsparse_double_complex alpha, const aoclsparse_double_complex *b, aoclsparse_double_complex *x,
                                        const aoclsparse_int kid)
{
    return symgs_t<cdouble>(trans, A, descr, cdouble(alpha.real, alpha.imag), reinterpret_cast<const cdouble *>(b),
                          reinterpret_cast<cdouble *>(x), nullptr, kid, false, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_zsymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const aoclsparse_double_complex alpha, const aoclsparse_double_complex *b, aoclsparse_double_complex *x,
                                       aoclsparse_double_complex *y)
{
    return symgs_t<cdouble>(trans, A, descr, cdouble(alpha.real, alpha.imag), reinterpret_cast<const cdouble *>(b),
                          reinterpret_cast<cdouble *>(x), reinterpret_cast<cdouble *>(y), -1, true, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_zsymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const aoclsparse_double_complex alpha, const aoclsparse_double_complex *b, aoclsparse_double_complex *x,
                                           aoclsparse_double_complex *y, const aoclsparse_int kid)
{
    return symgs_t<cdouble>(trans, A, descr, cdouble(alpha.real, alpha.imag), reinterpret_cast<const cdouble *>(b),
                          reinterpret_cast<cdouble *>(x), reinterpret_cast<cdouble *>(y), kid, true, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_dsymgs(aoclsparse_operation trans, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                    const double alpha, const double *b, double *x)
{
    return symgs_t<double>(trans, A, descr, alpha, b, x, nullptr, -1, false, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_ssymgs(aoclsparse_operation trans, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                    const float alpha, const float *b, float *x)
{
    return symgs_t<float>(trans, A, descr, alpha, b, x, nullptr, -1, false, aoclsparse_smat);
}
aoclsparse_status aoclsparse_dsymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                        const aoclsparse_mat_descr descr, const double alpha, const double *b,
                                        double *x, const aoclsparse_int kid)
{
    return symgs_t<double>(trans, A, descr, alpha, b, x, nullptr, kid, false, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_ssymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                        const aoclsparse_mat_descr descr, const float alpha, const float *b, float *x,
                                        const aoclsparse_int kid)
{
    return symgs_t<float>(trans, A, descr, alpha, b, x, nullptr, kid, false, aoclsparse_smat);
}
aoclsparse_status aoclsparse_dsymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const double alpha, const double *b,
                                       double *x, double *y)
{
    return symgs_t<double>(trans, A, descr, alpha, b, x, y, -1, true, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_ssymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const float alpha, const float *b, float *x,
                                       float *y)
{
    return symgs_t<float>(trans, A, descr, alpha, b, x, y, -1, true, aoclsparse_smat);
}
aoclsparse_status aoclsparse_dsymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const double alpha, const double *b,
                                           double *x, double *y, const aoclsparse_int kid)
{
    return symgs_t<double>(trans, A, descr, alpha, b, x, y, kid, true, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_ssymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const float alpha, const float *b,
                                           float *x, float *y, const aoclsparse_int kid)
{
    return symgs_t<float>(trans, A, descr, alpha, b, x, y, kid, true, aoclsparse_smat);
}

aoclsparse_status aoclsparse_dilu_smoother(aoclsparse_operation op, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, double **precond_csr_val,
                                           const double *approx_inv_diag, double *x, const double *b)
{
    (void)approx_inv_diag; // unused by the reference as well (solvers/aoclsparse_ilu.cpp:43-52)
    return ilu_smoother_t<double>(op, A, descr, precond_csr_val, x, b, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_silu_smoother(aoclsparse_operation op, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, float **precond_csr_val,
                                           const float *approx_inv_diag, float *x, const float *b)
{
    (void)approx_inv_diag;
    return ilu_smoother_t<float>(op, A, descr, precond_csr_val, x, b, aoclsparse_smat);
}


aoclsparse_status aoclsparse_dsorv(aoclsparse_sor_type sor_type, const aoclsparse_mat_descr descr, const aoclsparse_matrix A,
                                   double omega, double alpha, double *x, const double *b)
{
    return sorv_t<double>(sor_type, descr, A, omega, alpha, x, b, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_ssorv(aoclsparse_sor_type sor_type, const aoclsparse_mat_descr descr, const aoclsparse_matrix A,
                                   float omega, float alpha, float *x, const float *b)
{
    return sorv_t<float>(sor_type, descr, A, omega, alpha, x, b, aoclsparse_smat);
}
// sorv.hpp:128-131: the complex types are not implemented in the reference either
aoclsparse_status aoclsparse_csorv(aoclsparse_sor_type, const aoclsparse_mat_descr, const aoclsparse_matrix,
                                   aoclsparse_float_complex, aoclsparse_float_complex, aoclsparse_float_complex *,
                                   const aoclsparse_float_complex *)
{
    return aoclsparse_status_not_implemented;
}
aoclsparse_status aoclsparse_zsorv(aoclsparse_sor_type, const aoclsparse_mat_descr, const aoclsparse_matrix,
                                   aoclsparse_double_complex, aoclsparse_double_complex, aoclsparse_double_complex *,
                                   const aoclsparse_double_complex *)
{
    return aoclsparse_status_not_implemented;
}

} // extern "C"
