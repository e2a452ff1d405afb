/* aoclsparse.hpp -- the reference's C++ template entry points (library/include/aoclsparse.hpp:29-181:
 * aoclsparse::trsv<T>, mv<T>, create_csr<T>, sp2m<T>) as header-only forwards onto the C ABI of this library, so that
 * programs written against them (tests/examples/sample_mv_cpp.cpp, sample_trsv_cpp.cpp, sample_csr2m_cpp.cpp) build
 * unchanged.  T is float, double, std::complex<float/double> or aoclsparse_{float,double}_complex. */
#ifndef AOCLSPARSE_HPP_MI355_
#define AOCLSPARSE_HPP_MI355_

#include "aoclsparse.h"
#include "aoclsparse_mi355.h"

#include <complex>
#include <string>
#include <type_traits>

namespace aoclsparse
{
namespace mi355_detail
{
    template <typename T>
    struct is_c : std::integral_constant<bool, std::is_same<T, std::complex<float>>::value
                                                   || std::is_same<T, aoclsparse_float_complex>::value>
    {
    };
    template <typename T>
    struct is_z : std::integral_constant<bool, std::is_same<T, std::complex<double>>::value
                                                   || std::is_same<T, aoclsparse_double_complex>::value>
    {
    };
    template <typename T>
    inline aoclsparse_float_complex as_c(const T &v)
    {
        const float *p = reinterpret_cast<const float *>(&v);
        return aoclsparse_float_complex{p[0], p[1]};
    }
    template <typename T>
    inline aoclsparse_double_complex as_z(const T &v)
    {
        const double *p = reinterpret_cast<const double *>(&v);
        return aoclsparse_double_complex{p[0], p[1]};
    }
} // namespace mi355_detail

template <typename T>
aoclsparse_status trsv(const aoclsparse_operation trans, const T alpha, aoclsparse_matrix A,
                       const aoclsparse_mat_descr descr, const T *b, const aoclsparse_int incb, T *x,
                       const aoclsparse_int incx, aoclsparse_int kid = -1)
{
    using namespace mi355_detail;
    if constexpr(std::is_same<T, double>::value)
        return aoclsparse_mi355_dtrsv_full(trans, alpha, A, descr, b, incb, x, incx, kid);
    else if constexpr(std::is_same<T, float>::value)
        return aoclsparse_mi355_strsv_full(trans, alpha, A, descr, b, incb, x, incx, kid);
    else if constexpr(is_z<T>::value)
        return aoclsparse_mi355_ztrsv_full(trans, as_z(alpha), A, descr, reinterpret_cast<const aoclsparse_double_complex *>(b),
                                           incb, reinterpret_cast<aoclsparse_double_complex *>(x), incx, kid);
    else
    {
        static_assert(is_c<T>::value, "aoclsparse::trsv<T>: unsupported type");
        return aoclsparse_mi355_ctrsv_full(trans, as_c(alpha), A, descr, reinterpret_cast<const aoclsparse_float_complex *>(b),
                                           incb, reinterpret_cast<aoclsparse_float_complex *>(x), incx, kid);
    }
}

template <typename T>
aoclsparse_status mv(aoclsparse_operation op, const T *alpha, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                     const T *x, const T *beta, T *y)
{
    using namespace mi355_detail;
    if constexpr(std::is_same<T, double>::value)
        return aoclsparse_dmv(op, alpha, A, descr, x, beta, y);
    else if constexpr(std::is_same<T, float>::value)
        return aoclsparse_smv(op, alpha, A, descr, x, beta, y);
    else if constexpr(is_z<T>::value)
        return aoclsparse_zmv(op, reinterpret_cast<const aoclsparse_double_complex *>(alpha), A, descr,
                              reinterpret_cast<const aoclsparse_double_complex *>(x),
                              reinterpret_cast<const aoclsparse_double_complex *>(beta),
                              reinterpret_cast<aoclsparse_double_complex *>(y));
    else
    {
        static_assert(is_c<T>::value, "aoclsparse::mv<T>: unsupported type");
        return aoclsparse_cmv(op, reinterpret_cast<const aoclsparse_float_complex *>(alpha), A, descr,
                              reinterpret_cast<const aoclsparse_float_complex *>(x),
                              reinterpret_cast<const aoclsparse_float_complex *>(beta),
                              reinterpret_cast<aoclsparse_float_complex *>(y));
    }
}

/* fast_chck asks the reference to skip the O(nnz) validation; this library always validates */
template <typename T>
aoclsparse_status create_csr(aoclsparse_matrix *mat, aoclsparse_index_base base, aoclsparse_int M, aoclsparse_int N,
                             aoclsparse_int nnz, aoclsparse_int *row_ptr, aoclsparse_int *col_idx, T *val,
                             bool fast_chck = false)
{
    using namespace mi355_detail;
    (void)fast_chck;
    if constexpr(std::is_same<T, double>::value)
        return aoclsparse_create_dcsr(mat, base, M, N, nnz, row_ptr, col_idx, val);
    else if constexpr(std::is_same<T, float>::value)
        return aoclsparse_create_scsr(mat, base, M, N, nnz, row_ptr, col_idx, val);
    else if constexpr(is_z<T>::value)
        return aoclsparse_create_zcsr(mat, base, M, N, nnz, row_ptr, col_idx, reinterpret_cast<aoclsparse_double_complex *>(val));
    else
    {
        static_assert(is_c<T>::value, "aoclsparse::create_csr<T>: unsupported type");
        return aoclsparse_create_ccsr(mat, base, M, N, nnz, row_ptr, col_idx, reinterpret_cast<aoclsparse_float_complex *>(val));
    }
}

/* the C entry point dispatches on the handles' value type; the template argument is only checked by the library
 * through the handles (wrong_type when A and B disagree) */
template <typename T>
aoclsparse_status sp2m(aoclsparse_operation opA, const aoclsparse_mat_descr descrA, const aoclsparse_matrix A,
                       aoclsparse_operation opB, const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                       aoclsparse_request request, aoclsparse_matrix *C)
{
    return aoclsparse_sp2m(opA, descrA, A, opB, descrB, B, request, C);
}

} // namespace aoclsparse

#endif /* AOCLSPARSE_HPP_MI355_ */
