#!/bin/bash
# Regenerates tests/golden/cxx_symbols.txt: the mangled names a translation unit compiled against the REFERENCE's public C++ header
# (library/include/aoclsparse.hpp:55-128, declarations only) leaves undefined for aoclsparse::mv / trsv / sp2m / create_csr with
# T = float, double, std::complex<float>, std::complex<double> -- the sixteen symbols the reference library exports
# (aoclsparse_mv.cpp:351-360, aoclsparse_trsv.cpp:419-431, aoclsparse_csr2m.cpp:863-873, aoclsparse_create.cpp:99-110).
# Runs only where /root/reference exists (this container); the output is data (symbol names), not reference source.
set -e
REF=${REF:-/root/reference}
T=$(mktemp -d)
printf '#define AOCLSPARSE_VERSION_MAJOR 5\n#define AOCLSPARSE_VERSION_MINOR 3\n#define AOCLSPARSE_VERSION_PATCH 2\n' > $T/aoclsparse_version.h
cat > $T/t.cpp <<'EOF'
#include "aoclsparse.hpp"
#include <complex>
template <typename T> aoclsparse_status f(){
  aoclsparse_matrix A=nullptr; aoclsparse_mat_descr d=nullptr; T a{}; T*p=nullptr; aoclsparse_int *ip=nullptr;
  aoclsparse::mv<T>(aoclsparse_operation_none,&a,A,d,p,&a,p);
  aoclsparse::trsv<T>(aoclsparse_operation_none,a,A,d,p,1,p,1,-1);
  aoclsparse::sp2m<T>(aoclsparse_operation_none,d,A,aoclsparse_operation_none,d,A,aoclsparse_stage_full_computation,&A);
  return aoclsparse::create_csr<T>(&A,aoclsparse_index_base_zero,1,1,1,ip,ip,p,false);
}
template aoclsparse_status f<float>(); template aoclsparse_status f<double>();
template aoclsparse_status f<std::complex<float>>(); template aoclsparse_status f<std::complex<double>>();
EOF
g++ -std=c++17 -c $T/t.cpp -I$REF/library/include -I$T -o $T/t.o
nm $T/t.o | awk '$1=="U" && $2 ~ /^_ZN10aoclsparse/ {print $2}' | sort > "$(dirname "$0")/cxx_symbols.txt"
rm -rf $T
