// sp2md_api.cpp -- aoclsparse_?sp2md / ?spmmd (sparse x sparse, dense result), aoclsparse_?csr2dense and
// aoclsparse_?add (C = alpha*op(A) + B, sparse result).
//
// Drivers follow the reference's argument checks in order: level3/aoclsparse_sp2md.hpp:179-431,
// level3/aoclsparse_spmmd.cpp:39-151, conversion/aoclsparse_convert.hpp:658-745, level3/aoclsparse_csradd.hpp:283-532.
// The operands of sp2md / add are the handles' device CSR copies (op = T / H uses the cached stable transpose, the
// conjugation happens as the kernel loads a value), so repeated products do not move the matrices again; the dense
// result may live in host or device memory (pointer mode, as csrmm).
#include "internal.hpp"

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

using namespace mi355;

namespace
{

template <typename T>
constexpr bool is_cplx_v = !std::is_floating_point<T>::value;

template <typename T>
bool eq(T a, double v)
{
    if constexpr(is_cplx_v<T>)
        return a.re == v && a.im == 0;
    else
        return a == (T)v;
}

bool valid_op(aoclsparse_operation o)
{
    return o == aoclsparse_operation_none || o == aoclsparse_operation_transpose
           || o == aoclsparse_operation_conjugate_transpose;
}

bool csr_like(const aoclsparse_matrix A)
{
    // a handle created from CSC keeps the CSR of the same matrix in `user` (formats_api.cpp): both are served alike
    return A->input_format == aoclsparse_csr_mat && A->user.ptr;
}

template <typename T>
aoclsparse_status sp2md_t(aoclsparse_operation opA, const aoclsparse_mat_descr descrA, const aoclsparse_matrix A,
                          aoclsparse_operation opB, const aoclsparse_mat_descr descrB, const aoclsparse_matrix B, T alpha,
                          T beta, T *C, aoclsparse_order layout, aoclsparse_int ldc, aoclsparse_matrix_data_type vt)
{
    if(!descrA || !descrB)
        return aoclsparse_status_invalid_pointer;
    if(descrA->type != aoclsparse_matrix_type_general || descrB->type != aoclsparse_matrix_type_general)
        return aoclsparse_status_not_implemented;
    if(layout != aoclsparse_order_row && layout != aoclsparse_order_column)
        return aoclsparse_status_invalid_value;
    if(!A || !B || !C)
        return aoclsparse_status_invalid_pointer;
    if(!csr_like(A) || !csr_like(B))
        return aoclsparse_status_not_implemented;
    if(A->val_type != vt || B->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(!valid_op(opA) || !valid_op(opB))
        return aoclsparse_status_invalid_value;
    const bool           trA = opA != aoclsparse_operation_none, trB = opB != aoclsparse_operation_none;
    const aoclsparse_int m_c = trA ? A->n : A->m, inner_a = trA ? A->m : A->n;
    const aoclsparse_int n_c = trB ? B->m : B->n, inner_b = trB ? B->n : B->m;
    if(inner_a != inner_b)
        return aoclsparse_status_invalid_size;
    const bool rowmaj = layout == aoclsparse_order_row;
    if(ldc < (rowmaj ? n_c : m_c))
        return aoclsparse_status_invalid_size;
    const aoclsparse_int outer = rowmaj ? m_c : n_c, inner = rowmaj ? n_c : m_c;
    if((long long)outer * (long long)ldc > 2147483647LL)
        return aoclsparse_status_invalid_size;
    if(A->base != descrA->base || B->base != descrB->base)
        return aoclsparse_status_invalid_value;
    if(outer == 0 || inner == 0)
        return aoclsparse_status_success;

    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();
    const bool   cdev = rt.is_device_pointer(C);
    void        *dC = C;
    const size_t cbytes = sizeof(T) * (size_t)outer * (size_t)ldc;
    if(!cdev)
    {
        // the whole outer x ldc block travels both ways, so padding beyond `inner` keeps the caller's bytes
        st = rt.staging(4, cbytes, &dC);
        if(st != aoclsparse_status_success)
            return st;
        MI355_HIP_TRY(hipMemcpyAsync(dC, C, cbytes, hipMemcpyHostToDevice, rt.stream()));
    }
    if(!eq(beta, 1.0))
    {
        st = launch_dense_scale<T>(rt.stream(), static_cast<T *>(dC), inner, outer, ldc, beta, eq(beta, 0.0));
        if(st != aoclsparse_status_success)
            return st;
    }
    if(!eq(alpha, 0.0))
    {
        DeviceCsr *da = nullptr, *db = nullptr;
        SpmvPlan  *pa = nullptr, *pb = nullptr;
        st = ensure_spmv(const_cast<aoclsparse_matrix>(A), trA, da, pa);
        if(st == aoclsparse_status_success)
            st = ensure_spmv(const_cast<aoclsparse_matrix>(B), trB, db, pb);
        if(st != aoclsparse_status_success)
            return st;
        const bool conj_a = is_cplx_v<T> && opA == aoclsparse_operation_conjugate_transpose;
        const bool conj_b = is_cplx_v<T> && opB == aoclsparse_operation_conjugate_transpose;
        std::shared_lock<std::shared_mutex> ra(A->guard, std::defer_lock), rb(B->guard, std::defer_lock);
        ra.lock();
        if(B != A)
            rb.lock();
        st = launch_sp2md<T>(rt.stream(), da->m, da->base, da->ptr.as<aoclsparse_int>(), da->ind.as<aoclsparse_int>(),
                             da->val.as<T>(), conj_a, db->base, db->ptr.as<aoclsparse_int>(),
                             db->ind.as<aoclsparse_int>(), db->val.as<T>(), conj_b, alpha, static_cast<T *>(dC),
                             rowmaj ? (long long)ldc : 1LL, rowmaj ? 1LL : (long long)ldc);
        if(st != aoclsparse_status_success)
            return st;
    }
    if(!cdev)
    {
        MI355_HIP_TRY(hipMemcpyAsync(C, dC, cbytes, hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    }
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status spmmd_t(aoclsparse_operation op, const aoclsparse_matrix A, const aoclsparse_matrix B,
                          aoclsparse_order layout, T *C, aoclsparse_int ldc, aoclsparse_matrix_data_type vt)
{
    // spmmd.cpp:39-66: general descriptors in each matrix's base, alpha = 1, beta = 0, op(B) = B
    if(!A || !B)
        return aoclsparse_status_invalid_pointer;
    if(!csr_like(A) || !csr_like(B))
        return aoclsparse_status_not_implemented;
    _aoclsparse_mat_descr dA, dB;
    dA.base = A->base;
    dB.base = B->base;
    if constexpr(is_cplx_v<T>)
        return sp2md_t<T>(op, &dA, A, aoclsparse_operation_none, &dB, B, T(1, 0), T(0, 0), C, layout, ldc, vt);
    else
        return sp2md_t<T>(op, &dA, A, aoclsparse_operation_none, &dB, B, T(1), T(0), C, layout, ldc, vt);
}

template <typename T>
aoclsparse_status csr2dense_t(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr, const T *val,
                              const aoclsparse_int *row_ptr, const aoclsparse_int *col_ind, T *A, aoclsparse_int ld,
                              aoclsparse_order order)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != aoclsparse_index_base_zero && descr->base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    const aoclsparse_matrix_type ty = descr->type;
    if(ty != aoclsparse_matrix_type_general && ty != aoclsparse_matrix_type_symmetric
       && ty != aoclsparse_matrix_type_triangular && ty != aoclsparse_matrix_type_hermitian)
        return aoclsparse_status_invalid_value;
    if(order == aoclsparse_order_column
       && (ty == aoclsparse_matrix_type_triangular || ty == aoclsparse_matrix_type_hermitian))
        return aoclsparse_status_not_implemented;
    if(ty != aoclsparse_matrix_type_general)
    {
        if(descr->fill_mode != aoclsparse_fill_mode_lower && descr->fill_mode != aoclsparse_fill_mode_upper)
            return aoclsparse_status_invalid_value;
        if(descr->diag_type != aoclsparse_diag_type_non_unit && descr->diag_type != aoclsparse_diag_type_unit
           && descr->diag_type != aoclsparse_diag_type_zero)
            return aoclsparse_status_invalid_value;
    }
    if(m < 0 || n < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0)
        return aoclsparse_status_success;
    if(!val || !row_ptr || !col_ind || !A)
        return aoclsparse_status_invalid_pointer;
    const bool           colmaj = order == aoclsparse_order_column;
    const aoclsparse_int outer = colmaj ? n : m, inner = colmaj ? m : n;
    if((long long)outer * (long long)ld > 2147483647LL)
        return aoclsparse_status_invalid_size;

    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();
    hipStream_t s = rt.stream();
    // CSR arrays: device pointers are used in place, host arrays are staged (slots 0-2)
    const bool            sdev = rt.is_device_pointer(row_ptr);
    const aoclsparse_int *dptr = row_ptr, *dind = col_ind;
    const T              *dval = val;
    if(!sdev)
    {
        const aoclsparse_int nnz = row_ptr[m] - descr->base;
        if(nnz < 0)
            return aoclsparse_status_invalid_value;
        void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
        st = rt.staging(0, sizeof(aoclsparse_int) * ((size_t)m + 1), &p0);
        if(st == aoclsparse_status_success)
            st = rt.staging(1, sizeof(aoclsparse_int) * (size_t)std::max(nnz, 1), &p1);
        if(st == aoclsparse_status_success)
            st = rt.staging(2, sizeof(T) * (size_t)std::max(nnz, 1), &p2);
        if(st != aoclsparse_status_success)
            return st;
        MI355_HIP_TRY(hipMemcpyAsync(p0, row_ptr, sizeof(aoclsparse_int) * ((size_t)m + 1), hipMemcpyHostToDevice, s));
        if(nnz > 0)
        {
            MI355_HIP_TRY(hipMemcpyAsync(p1, col_ind, sizeof(aoclsparse_int) * (size_t)nnz, hipMemcpyHostToDevice, s));
            MI355_HIP_TRY(hipMemcpyAsync(p2, val, sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice, s));
        }
        dptr = static_cast<const aoclsparse_int *>(p0), dind = static_cast<const aoclsparse_int *>(p1);
        dval = static_cast<const T *>(p2);
    }
    const bool   adev = rt.is_device_pointer(A);
    void        *dA = A;
    const size_t abytes = sizeof(T) * (size_t)outer * (size_t)ld;
    if(!adev)
    {
        st = rt.staging(4, abytes, &dA);
        if(st != aoclsparse_status_success)
            return st;
        if(ld != inner) // padding keeps the caller's bytes
            MI355_HIP_TRY(hipMemcpyAsync(dA, A, abytes, hipMemcpyHostToDevice, s));
    }
    T zero{};
    std::memset(&zero, 0, sizeof(T));
    st = launch_dense_scale<T>(s, static_cast<T *>(dA), inner, outer, ld, zero, true);
    if(st != aoclsparse_status_success)
        return st;
    // The reference's column-major symmetric branch addresses the diagonal and both mirrors through
    // row*ld (convert.hpp:770-806), which for a symmetric result is the same matrix as the row-major walk.
    const int mode = ty == aoclsparse_matrix_type_general     ? 0
                     : ty == aoclsparse_matrix_type_symmetric ? 1
                     : ty == aoclsparse_matrix_type_hermitian ? 2
                                                              : 3;
    const long long rs = colmaj ? 1LL : (long long)ld, cs = colmaj ? (long long)ld : 1LL;
    st = launch_csr2dense<T>(s, m, descr->base, dptr, dind, dval, static_cast<T *>(dA), rs, cs, mode,
                             descr->fill_mode == aoclsparse_fill_mode_upper ? 1 : 0,
                             descr->diag_type == aoclsparse_diag_type_unit   ? 1
                             : descr->diag_type == aoclsparse_diag_type_zero ? 2
                                                                             : 0);
    if(st != aoclsparse_status_success)
        return st;
    if(!adev)
    {
        MI355_HIP_TRY(hipMemcpyAsync(A, dA, abytes, hipMemcpyDeviceToHost, s));
        MI355_HIP_TRY(hipStreamSynchronize(s));
    }
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status add_t(aoclsparse_operation op, const aoclsparse_matrix A, T alpha, const aoclsparse_matrix B,
                        aoclsparse_matrix *C, aoclsparse_matrix_data_type vt)
{
    if(!A || !B || !C)
        return aoclsparse_status_invalid_pointer;
    if(A->input_format != aoclsparse_csr_mat || B->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(A->val_type != vt || B->val_type != vt)
        return aoclsparse_status_wrong_type;
    const bool tr = op != aoclsparse_operation_none;
    if(!tr ? (A->m != B->m || A->n != B->n) : (A->m != B->n || A->n != B->m))
        return aoclsparse_status_invalid_size;
    if(!A->user.ptr || (A->nnz != 0 && (!A->user.ind || !A->user.val)))
        return aoclsparse_status_invalid_pointer;
    if(!B->user.ptr || (B->nnz != 0 && (!B->user.ind || !B->user.val)))
        return aoclsparse_status_invalid_pointer;
    *C = nullptr;
    const aoclsparse_int        M = B->m, N = B->n;
    const aoclsparse_index_base base_a = A->base;
    if(M == 0 || N == 0 || (A->nnz == 0 && B->nnz == 0))
        return new_csr_result(C, M, N, 0, vt, nullptr, base_a); // csradd.hpp:163-177

    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    DeviceCsr *da = nullptr, *db = nullptr;
    SpmvPlan  *pa = nullptr, *pb = nullptr;
    st = ensure_spmv(const_cast<aoclsparse_matrix>(A), tr, da, pa);
    if(st == aoclsparse_status_success)
        st = ensure_spmv(const_cast<aoclsparse_matrix>(B), false, db, pb);
    if(st != aoclsparse_status_success)
        return st;
    const bool   conj_a = is_cplx_v<T> && op == aoclsparse_operation_conjugate_transpose;
    hipStream_t  s = rt.stream();
    DeviceBuffer d_cnt, d_cptr, d_ci, d_cv;
    st = d_cnt.alloc(sizeof(aoclsparse_int) * (size_t)M);
    if(st != aoclsparse_status_success)
        return st;
    std::shared_lock<std::shared_mutex> ra(A->guard, std::defer_lock), rb(B->guard, std::defer_lock);
    ra.lock();
    if(B != A)
        rb.lock();
    st = launch_csradd<T>(s, false, M, da->base, da->ptr.as<aoclsparse_int>(), da->ind.as<aoclsparse_int>(), nullptr,
                          conj_a, alpha, db->base, db->ptr.as<aoclsparse_int>(), db->ind.as<aoclsparse_int>(), nullptr,
                          base_a, nullptr, d_cnt.as<aoclsparse_int>(), nullptr);
    if(st != aoclsparse_status_success)
        return st;
    std::vector<aoclsparse_int> cptr;
    try
    {
        cptr.assign((size_t)M + 1, 0);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    MI355_HIP_TRY(hipMemcpyAsync(cptr.data() + 1, d_cnt.ptr, sizeof(aoclsparse_int) * (size_t)M, hipMemcpyDeviceToHost, s));
    MI355_HIP_TRY(hipStreamSynchronize(s));
    long long run = base_a; // 64-bit running sum, overflow -> invalid_size (csradd.hpp:108-124)
    cptr[0]       = base_a;
    for(aoclsparse_int i = 1; i <= M; i++)
    {
        run += cptr[i];
        cptr[i] = (aoclsparse_int)run;
    }
    if(run > 2147483647LL)
        return aoclsparse_status_invalid_size;
    const aoclsparse_int nnz_c = (aoclsparse_int)(run - base_a);
    aoclsparse_matrix    c     = nullptr;
    st = new_csr_result(&c, M, N, nnz_c, vt, cptr.data(), base_a);
    if(st != aoclsparse_status_success)
        return st;
    st = d_cptr.upload(cptr.data(), sizeof(aoclsparse_int) * ((size_t)M + 1), s);
    if(st == aoclsparse_status_success)
        st = d_ci.alloc(sizeof(aoclsparse_int) * (size_t)std::max(nnz_c, 1));
    if(st == aoclsparse_status_success)
        st = d_cv.alloc(sizeof(T) * (size_t)std::max(nnz_c, 1));
    if(st == aoclsparse_status_success)
        st = launch_csradd<T>(s, true, M, da->base, da->ptr.as<aoclsparse_int>(), da->ind.as<aoclsparse_int>(),
                              da->val.as<T>(), conj_a, alpha, db->base, db->ptr.as<aoclsparse_int>(),
                              db->ind.as<aoclsparse_int>(), db->val.as<T>(), base_a, d_cptr.as<aoclsparse_int>(),
                              d_ci.as<aoclsparse_int>(), d_cv.as<T>());
    if(st == aoclsparse_status_success && nnz_c > 0)
    {
        hipError_t e = hipMemcpyAsync(c->user.ind, d_ci.ptr, sizeof(aoclsparse_int) * (size_t)nnz_c, hipMemcpyDeviceToHost, s);
        if(e == hipSuccess)
            e = hipMemcpyAsync(c->user.val, d_cv.ptr, sizeof(T) * (size_t)nnz_c, hipMemcpyDeviceToHost, s);
        if(e == hipSuccess)
            e = hipStreamSynchronize(s);
        if(e != hipSuccess)
            st = aoclsparse_status_internal_error;
    }
    if(st != aoclsparse_status_success)
    {
        aoclsparse_destroy(&c);
        return st;
    }
    *C = c;
    return aoclsparse_status_success;
}

inline cfloat  cv(aoclsparse_float_complex v) { return cfloat(v.real, v.imag); }
inline cdouble cv(aoclsparse_double_complex v) { return cdouble(v.real, v.imag); }

} // namespace

extern "C" {

#define MI355_SP2MD_REAL(P, T, VT)                                                                                       \
    aoclsparse_status aoclsparse_##P##sp2md(const aoclsparse_operation opA, const aoclsparse_mat_descr descrA,           \
                                            const aoclsparse_matrix A, const aoclsparse_operation opB,                   \
                                            const aoclsparse_mat_descr descrB, const aoclsparse_matrix B, const T alpha, \
                                            const T beta, T *C, const aoclsparse_order layout, const aoclsparse_int ldc) \
    {                                                                                                                    \
        return sp2md_t<T>(opA, descrA, A, opB, descrB, B, alpha, beta, C, layout, ldc, VT);                              \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##spmmd(const aoclsparse_operation op, const aoclsparse_matrix A,                    \
                                            const aoclsparse_matrix B, const aoclsparse_order layout, T *C,              \
                                            const aoclsparse_int ldc)                                                    \
    {                                                                                                                    \
        return spmmd_t<T>(op, A, B, layout, C, ldc, VT);                                                                 \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##csr2dense(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,    \
                                                const T *csr_val, const aoclsparse_int *csr_row_ptr,                     \
                                                const aoclsparse_int *csr_col_ind, T *A, aoclsparse_int ld,              \
                                                aoclsparse_order order)                                                  \
    {                                                                                                                    \
        return csr2dense_t<T>(m, n, descr, csr_val, csr_row_ptr, csr_col_ind, A, ld, order);                             \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##add(const aoclsparse_operation op, const aoclsparse_matrix A, const T alpha,       \
                                          const aoclsparse_matrix B, aoclsparse_matrix *C)                               \
    {                                                                                                                    \
        return add_t<T>(op, A, alpha, B, C, VT);                                                                         \
    }
MI355_SP2MD_REAL(d, double, aoclsparse_dmat)
MI355_SP2MD_REAL(s, float, aoclsparse_smat)

#define MI355_SP2MD_CPLX(P, CT, T, VT)                                                                                   \
    aoclsparse_status aoclsparse_##P##sp2md(const aoclsparse_operation opA, const aoclsparse_mat_descr descrA,           \
                                            const aoclsparse_matrix A, const aoclsparse_operation opB,                   \
                                            const aoclsparse_mat_descr descrB, const aoclsparse_matrix B, CT alpha,      \
                                            CT beta, CT *C, const aoclsparse_order layout, const aoclsparse_int ldc)     \
    {                                                                                                                    \
        return sp2md_t<T>(opA, descrA, A, opB, descrB, B, cv(alpha), cv(beta), reinterpret_cast<T *>(C), layout, ldc,    \
                          VT);                                                                                           \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##spmmd(const aoclsparse_operation op, const aoclsparse_matrix A,                    \
                                            const aoclsparse_matrix B, const aoclsparse_order layout, CT *C,             \
                                            const aoclsparse_int ldc)                                                    \
    {                                                                                                                    \
        return spmmd_t<T>(op, A, B, layout, reinterpret_cast<T *>(C), ldc, VT);                                          \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##csr2dense(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,    \
                                                const CT *csr_val, const aoclsparse_int *csr_row_ptr,                    \
                                                const aoclsparse_int *csr_col_ind, CT *A, aoclsparse_int ld,             \
                                                aoclsparse_order order)                                                  \
    {                                                                                                                    \
        return csr2dense_t<T>(m, n, descr, reinterpret_cast<const T *>(csr_val), csr_row_ptr, csr_col_ind,               \
                              reinterpret_cast<T *>(A), ld, order);                                                      \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##add(const aoclsparse_operation op, const aoclsparse_matrix A, const CT alpha,      \
                                          const aoclsparse_matrix B, aoclsparse_matrix *C)                               \
    {                                                                                                                    \
        return add_t<T>(op, A, cv(alpha), B, C, VT);                                                                     \
    }
MI355_SP2MD_CPLX(z, aoclsparse_double_complex, cdouble, aoclsparse_zmat)
MI355_SP2MD_CPLX(c, aoclsparse_float_complex, cfloat, aoclsparse_cmat)

} // extern "C"
