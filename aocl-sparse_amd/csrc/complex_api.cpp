// complex_api.cpp -- aoclsparse_cmv / aoclsparse_zmv: y = alpha op(A) x + beta y for complex handles, every
// descriptor type and operation the reference dispatches by doid (level2/aoclsparse_mv.cpp:41-349,
// include/aoclsparse_mtx_dispatcher.hpp:41-353).
//
// The operator is always an ordinary CSR in HBM -- the user's CSR, its transpose, or an expansion / slice
// derived from the clean CSR (derived.cpp; the mirrored half of a hermitian matrix is stored conjugated) --
// and the conjugation that an operation adds on top is a flag of the kernel:
//
//   descriptor   op = N                 op = T                    op = H
//   general      user                   transpose                 transpose, conj
//   symmetric    expansion              expansion (A^T = A)       expansion, conj (A^H = conj A)
//   hermitian    expansion              expansion, conj (A^T = conj A)   expansion (A^H = A)
//   triangular   slice                  transposed slice          transposed slice, conj
#include "internal.hpp"

#include <algorithm>

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

template <typename R>
aoclsparse_status cmv_t(aoclsparse_operation op, const cplx<R> *alpha, aoclsparse_matrix A,
                        const aoclsparse_mat_descr descr, const cplx<R> *x, const cplx<R> *beta, cplx<R> *y,
                        aoclsparse_matrix_data_type vt)
{
    using C = cplx<R>;
    // mv.cpp:55-97, same order
    if(!alpha || !beta || !A || !descr || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(A->input_format == aoclsparse_csr_mat && !A->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    if(op != aoclsparse_operation_none && op != aoclsparse_operation_transpose
       && op != aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_invalid_value;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(descr->type < aoclsparse_matrix_type_general || descr->type > aoclsparse_matrix_type_triangular)
        return aoclsparse_status_invalid_value;
    const bool sym  = descr->type == aoclsparse_matrix_type_symmetric;
    const bool herm = descr->type == aoclsparse_matrix_type_hermitian;
    if((sym || herm) && A->m != A->n)
        return aoclsparse_status_invalid_size;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;

    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    const bool                            tr = op != aoclsparse_operation_none;
    if(A->m == 0 || A->n == 0 || (A->nnz == 0 && descr->type == aoclsparse_matrix_type_general))
    {
        const aoclsparse_int dim = tr ? A->n : A->m; // mv.cpp:116-121: an empty matrix still scales y
        StagedArg            ay;
        const bool           b0 = beta->re == R(0) && beta->im == R(0);
        MI355_TRY(ay.in(rt, 4, y, sizeof(C) * (size_t)dim, !b0));
        MI355_TRY(launch_cscale<R>(rt.stream(), static_cast<C *>(ay.dev), dim, *beta));
        MI355_TRY(ay.out(rt));
        if(ay.staged)
            MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        return aoclsparse_status_success;
    }

    DeviceCsr            *dcsr = nullptr;
    SpmvPlan             *plan = nullptr;
    const aoclsparse_int *hptr = nullptr; // host row_ptr of the operator the product runs on
    bool                  conj = false;
    if(descr->type == aoclsparse_matrix_type_general)
    {
        MI355_TRY(ensure_spmv(A, tr, dcsr, plan));
        conj = op == aoclsparse_operation_conjugate_transpose;
        hptr = (tr ? *A->trans : A->user).ptr;
    }
    else
    {
        Derived *dv = nullptr;
        MI355_TRY(ensure_derived(A, descr->type, descr->fill_mode, descr->diag_type, tr, dv));
        dcsr = &dv->dev;
        plan = &dv->plan;
        hptr = dv->host.ptr;
        conj = sym    ? op == aoclsparse_operation_conjugate_transpose
               : herm ? op == aoclsparse_operation_transpose
                      : op == aoclsparse_operation_conjugate_transpose;
    }
    // The SELL-64 copy the real types run on (round 4: complex ?mv was a lane group per CSR row, 0.42-0.54 of the roofline on the
    // 2000^2 Laplacian): built for a handle that carries an mv hint, at its first product; never under
    // aoclsparse_memory_usage_minimal.  One chain per row (left to right, c_mac), conjugation at load.  No silent promotion after
    // N products as for the real types: there the two kernels give the same bits, here the lane-group kernel sums a row as a
    // tree, and a result that changes in the middle of a caller's loop would be worse than a slower kernel.
    if(plan && plan->valid && !plan->sell.valid && !plan->sell.tried && A->mem_policy == aoclsparse_memory_usage_unrestricted)
    {
        bool hinted = false;
        {
            std::shared_lock<std::shared_mutex> r(A->guard);
            for(const Hint &h : A->hints)
                hinted |= h.act == action_mv;
        }
        if(hinted)
        {
            std::unique_lock<std::shared_mutex> w(A->guard);
            MI355_TRY(build_sell(hptr, *dcsr, sizeof(C), *plan, true));
        }
    }
    std::shared_lock<std::shared_mutex> r(A->guard);
    StagedArg                           ax, ay;
    const bool                          b0 = beta->re == R(0) && beta->im == R(0);
    MI355_TRY(ax.in(rt, 3, x, sizeof(C) * (size_t)dcsr->n, true));
    MI355_TRY(ay.in(rt, 4, y, sizeof(C) * (size_t)dcsr->m, !b0));
    if(plan && plan->sell.valid)
        MI355_TRY(launch_sellmv_complex<R>(rt.stream(), conj, *alpha, dcsr->m, plan->sell.nslices, plan->sell.slice_ptr.as<long long>(),
                                           plan->sell.val.as<C>(), plan->sell.col.as<aoclsparse_int>(),
                                           plan->sell.rowlen.as<aoclsparse_int>(), static_cast<const C *>(ax.dev), *beta,
                                           static_cast<C *>(ay.dev), plan->sell.shared ? plan->sell.cptr.as<long long>() : nullptr,
                                           plan->sell.shared ? plan->sell.lead.as<unsigned short>() : nullptr, plan->max_row_nnz,
                                           plan->sell.next_direction()));
    else
        MI355_TRY(launch_cspmv<R>(rt.stream(), dcsr->base, conj, *alpha, dcsr->m, dcsr->nnz, dcsr->val.as<C>(),
                                  dcsr->ind.as<aoclsparse_int>(), dcsr->ptr.as<aoclsparse_int>(),
                                  static_cast<const C *>(ax.dev), *beta, static_cast<C *>(ay.dev)));
    MI355_TRY(ay.out(rt));
    if(ay.staged)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return aoclsparse_status_success;
}

// aoclsparse_{c,z}csrmm: checks as level3/aoclsparse_csrmm.hpp:429-640 (same order as csrmm_api.cpp), operator
// choice as in cmv_t
template <typename R>
aoclsparse_status ccsrmm_t(aoclsparse_operation op, const cplx<R> alpha, const aoclsparse_matrix A,
                           const aoclsparse_mat_descr descr, aoclsparse_order order, const cplx<R> *B, aoclsparse_int n,
                           aoclsparse_int ldb, const cplx<R> beta, cplx<R> *C, aoclsparse_int ldc, aoclsparse_int kid,
                           aoclsparse_matrix_data_type vt)
{
    using Cx = cplx<R>;
    if(!A || !B || !C || !descr)
        return aoclsparse_status_invalid_pointer;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(op != aoclsparse_operation_none && op != aoclsparse_operation_transpose
       && op != aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_invalid_value;
    const bool sym = descr->type == aoclsparse_matrix_type_symmetric, herm = descr->type == aoclsparse_matrix_type_hermitian;
    if(descr->type != aoclsparse_matrix_type_general && !sym && !herm)
        return aoclsparse_status_not_implemented;
    if((sym || herm) && A->m != A->n)
        return aoclsparse_status_invalid_size;
    if(order != aoclsparse_order_row && order != aoclsparse_order_column)
        return aoclsparse_status_invalid_value;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    const aoclsparse_int m = A->m, k = A->n;
    if(m < 0 || n < 0 || k < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0 || k == 0)
        return aoclsparse_status_success;
    const bool a0 = alpha.re == R(0) && alpha.im == R(0), b0 = beta.re == R(0) && beta.im == R(0);
    if(a0 && beta.re == R(1) && beta.im == R(0))
        return aoclsparse_status_success;
    if(!A->user.val || !A->user.ptr || !A->user.ind)
        return aoclsparse_status_invalid_pointer;
    const bool           tr     = op != aoclsparse_operation_none;
    const bool           colmaj = order == aoclsparse_order_column;
    const aoclsparse_int b_rows = tr ? m : k, m_c = tr ? k : m;
    const aoclsparse_int chk_b = colmaj ? b_rows : n, chk_c = colmaj ? m_c : n;
    if(ldb < (chk_b > 1 ? chk_b : 1) || ldc < (chk_c > 1 ? chk_c : 1))
        return aoclsparse_status_invalid_size;
    const long long c_outer = colmaj ? n : m_c, b_outer = colmaj ? n : b_rows;
    if(c_outer * (long long)ldc > 2147483647LL || b_outer * (long long)ldb > 2147483647LL)
        return aoclsparse_status_invalid_size;
    if(kid > 3)
        return aoclsparse_status_invalid_kid;

    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    StagedArg                             aB, aC;
    MI355_TRY(aC.in(rt, 4, C, sizeof(Cx) * (size_t)c_outer * (size_t)ldc, true)); // padding of ldc survives
    auto finish = [&]() -> aoclsparse_status {
        MI355_TRY(aC.out(rt));
        if(aC.staged)
            MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        return aoclsparse_status_success;
    };
    if(a0)
    {
        MI355_TRY(launch_cscale_dense<R>(rt.stream(), order, static_cast<Cx *>(aC.dev), m_c, n, ldc, beta));
        return finish();
    }
    (void)b0;
    MI355_TRY(aB.in(rt, 3, B, sizeof(Cx) * (size_t)b_outer * (size_t)ldb, true));
    DeviceCsr *d    = nullptr;
    bool       conj = false;
    if(sym || herm)
    {
        Derived *dv = nullptr;
        MI355_TRY(ensure_derived(const_cast<aoclsparse_matrix>(A), descr->type, descr->fill_mode, descr->diag_type,
                                 false, dv));
        d    = &dv->dev;
        conj = sym ? op == aoclsparse_operation_conjugate_transpose : op == aoclsparse_operation_transpose;
    }
    else
    {
        SpmvPlan *p = nullptr;
        MI355_TRY(ensure_spmv(const_cast<aoclsparse_matrix>(A), tr, d, p));
        conj = op == aoclsparse_operation_conjugate_transpose;
    }
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        MI355_TRY(launch_ccsrmm<R>(rt.stream(), order, d->base, conj, alpha, d->m, d->val.as<Cx>(),
                                   d->ind.as<aoclsparse_int>(), d->ptr.as<aoclsparse_int>(),
                                   static_cast<const Cx *>(aB.dev), n, ldb, beta, static_cast<Cx *>(aC.dev), ldc));
    }
    return finish();
}

// aoclsparse_{c,z}dotmv (level2/aoclsparse_dotmv.hpp:31-60): y = alpha op(A) x + beta y, then the CONJUGATED dot
// d = sum conj(x_i) y_i over min(m, n) entries (level1/aoclsparse_dense_dot.hpp:36-49)
template <typename R>
aoclsparse_status cdotmv_t(aoclsparse_operation op, cplx<R> alpha, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                           const cplx<R> *x, cplx<R> beta, cplx<R> *y, cplx<R> *d, aoclsparse_matrix_data_type vt)
{
    using C = cplx<R>;
    if(!d || !A)
        return aoclsparse_status_invalid_pointer;
    MI355_TRY(cmv_t<R>(op, &alpha, A, descr, x, &beta, y, vt));
    Runtime                              &rt = Runtime::get();
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    const aoclsparse_int                  n = std::min(A->m, A->n);
    StagedArg                             ax, ay;
    MI355_TRY(ax.in(rt, 3, x, sizeof(C) * (size_t)n, true));
    MI355_TRY(ay.in(rt, 4, y, sizeof(C) * (size_t)n, true));
    void      *part = nullptr, *dd = d;
    const bool ddev = rt.is_device_pointer(d);
    MI355_TRY(rt.staging(6, sizeof(C) * 1024, &part));
    if(!ddev)
        MI355_TRY(rt.staging(7, sizeof(C), &dd));
    MI355_TRY(launch_cdot<R>(rt.stream(), n, static_cast<const C *>(ax.dev), static_cast<const C *>(ay.dev),
                             static_cast<C *>(part), static_cast<C *>(dd)));
    if(!ddev)
    {
        MI355_HIP_TRY(hipMemcpyAsync(d, dd, sizeof(C), hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    }
    return aoclsparse_status_success;
}

} // namespace

extern "C" {

aoclsparse_status aoclsparse_cdotmv(const aoclsparse_operation op, const aoclsparse_float_complex alpha,
                                    aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                    const aoclsparse_float_complex *x, const aoclsparse_float_complex beta,
                                    aoclsparse_float_complex *y, aoclsparse_float_complex *d)
{
    return cdotmv_t<float>(op, cfloat(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cfloat *>(x),
                           cfloat(beta.real, beta.imag), reinterpret_cast<cfloat *>(y), reinterpret_cast<cfloat *>(d),
                           aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zdotmv(const aoclsparse_operation op, const aoclsparse_double_complex alpha,
                                    aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                    const aoclsparse_double_complex *x, const aoclsparse_double_complex beta,
                                    aoclsparse_double_complex *y, aoclsparse_double_complex *d)
{
    return cdotmv_t<double>(op, cdouble(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cdouble *>(x),
                            cdouble(beta.real, beta.imag), reinterpret_cast<cdouble *>(y), reinterpret_cast<cdouble *>(d),
                            aoclsparse_zmat);
}

aoclsparse_status aoclsparse_ccsrmm(aoclsparse_operation op, const aoclsparse_float_complex alpha,
                                    const aoclsparse_matrix A, const aoclsparse_mat_descr descr, aoclsparse_order order,
                                    const aoclsparse_float_complex *B, aoclsparse_int n, aoclsparse_int ldb,
                                    const aoclsparse_float_complex beta, aoclsparse_float_complex *C, aoclsparse_int ldc)
{
    return ccsrmm_t<float>(op, cfloat(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cfloat *>(B), n,
                           ldb, cfloat(beta.real, beta.imag), reinterpret_cast<cfloat *>(C), ldc, -1, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zcsrmm(aoclsparse_operation op, const aoclsparse_double_complex alpha,
                                    const aoclsparse_matrix A, const aoclsparse_mat_descr descr, aoclsparse_order order,
                                    const aoclsparse_double_complex *B, aoclsparse_int n, aoclsparse_int ldb,
                                    const aoclsparse_double_complex beta, aoclsparse_double_complex *C,
                                    aoclsparse_int ldc)
{
    return ccsrmm_t<double>(op, cdouble(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cdouble *>(B),
                            n, ldb, cdouble(beta.real, beta.imag), reinterpret_cast<cdouble *>(C), ldc, -1,
                            aoclsparse_zmat);
}
aoclsparse_status aoclsparse_ccsrmm_kid(aoclsparse_operation op, const aoclsparse_float_complex alpha,
                                        const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                        aoclsparse_order order, const aoclsparse_float_complex *B, aoclsparse_int n,
                                        aoclsparse_int ldb, const aoclsparse_float_complex beta,
                                        aoclsparse_float_complex *C, aoclsparse_int ldc, const aoclsparse_int kid)
{
    return ccsrmm_t<float>(op, cfloat(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cfloat *>(B), n,
                           ldb, cfloat(beta.real, beta.imag), reinterpret_cast<cfloat *>(C), ldc, kid, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zcsrmm_kid(aoclsparse_operation op, const aoclsparse_double_complex alpha,
                                        const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                        aoclsparse_order order, const aoclsparse_double_complex *B, aoclsparse_int n,
                                        aoclsparse_int ldb, const aoclsparse_double_complex beta,
                                        aoclsparse_double_complex *C, aoclsparse_int ldc, const aoclsparse_int kid)
{
    return ccsrmm_t<double>(op, cdouble(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cdouble *>(B),
                            n, ldb, cdouble(beta.real, beta.imag), reinterpret_cast<cdouble *>(C), ldc, kid,
                            aoclsparse_zmat);
}

aoclsparse_status aoclsparse_cmv(aoclsparse_operation op, const aoclsparse_float_complex *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr descr, const aoclsparse_float_complex *x,
                                 const aoclsparse_float_complex *beta, aoclsparse_float_complex *y)
{
    return cmv_t<float>(op, reinterpret_cast<const cfloat *>(alpha), A, descr, reinterpret_cast<const cfloat *>(x),
                        reinterpret_cast<const cfloat *>(beta), reinterpret_cast<cfloat *>(y), aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zmv(aoclsparse_operation op, const aoclsparse_double_complex *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr descr, const aoclsparse_double_complex *x,
                                 const aoclsparse_double_complex *beta, aoclsparse_double_complex *y)
{
    return cmv_t<double>(op, reinterpret_cast<const cdouble *>(alpha), A, descr, reinterpret_cast<const cdouble *>(x),
                         reinterpret_cast<const cdouble *>(beta), reinterpret_cast<cdouble *>(y), aoclsparse_zmat);
}

} // extern "C"
