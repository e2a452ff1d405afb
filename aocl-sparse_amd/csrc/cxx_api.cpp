// cxx_api.cpp -- the reference's exported C++ template entry points.
//
// The reference library exports explicit instantiations of aoclsparse::mv / trsv / sp2m / create_csr for float, double,
// std::complex<float> and std::complex<double> (library/src/level2/aoclsparse_mv.cpp:351-360, aoclsparse_trsv.cpp:419-431,
// library/src/level3/aoclsparse_csr2m.cpp:863-873, library/src/create/aoclsparse_create.cpp:99-110), and its public header
// (library/include/aoclsparse.hpp:55-128) only DECLARES them: a program built against that header carries undefined references
// to the mangled names.  include/aoclsparse.hpp defines the same templates as forwards onto the C ABI; instantiating them here with
// default visibility gives this library the same sixteen symbols, so such a program links and runs unchanged
// (tests/golden/cxx_symbols.txt holds the names a compile against the reference header asks for).
#include "aoclsparse.hpp"

#define MI355_CXX_API(T)                                                                                                            \
    template DLL_PUBLIC aoclsparse_status aoclsparse::mv<T>(aoclsparse_operation, const T *, aoclsparse_matrix,                       \
                                                            const aoclsparse_mat_descr, const T *, const T *, T *);                  \
    template DLL_PUBLIC aoclsparse_status aoclsparse::trsv<T>(const aoclsparse_operation, const T, aoclsparse_matrix,                 \
                                                              const aoclsparse_mat_descr, const T *, const aoclsparse_int, T *,      \
                                                              const aoclsparse_int, aoclsparse_int);                                 \
    template DLL_PUBLIC aoclsparse_status aoclsparse::sp2m<T>(aoclsparse_operation, const aoclsparse_mat_descr,                       \
                                                              const aoclsparse_matrix, aoclsparse_operation,                         \
                                                              const aoclsparse_mat_descr, const aoclsparse_matrix,                   \
                                                              aoclsparse_request, aoclsparse_matrix *);                              \
    template DLL_PUBLIC aoclsparse_status aoclsparse::create_csr<T>(aoclsparse_matrix *, aoclsparse_index_base, aoclsparse_int,       \
                                                                    aoclsparse_int, aoclsparse_int, aoclsparse_int *,                \
                                                                    aoclsparse_int *, T *, bool);

MI355_CXX_API(float)
MI355_CXX_API(double)
MI355_CXX_API(std::complex<float>)
MI355_CXX_API(std::complex<double>)
