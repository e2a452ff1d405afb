// kt_order.hpp -- the horizontal-sum trees of the reference's vector "kernel templates" (device side).
// kt_hsum_p for 256-bit registers: library/src/include/kernel-templates/kt_l0_avx2.hpp:331-351; for 512-bit registers
// kt_l0_avx512.hpp:367-376 = _mm512_reduce_add_pd/ps, which the compiler header expands to lane-wise additions of the
// register halves.  oracle/oracle.c restates the same trees and tests/test_oracle_kt.py pins them bit for bit on the
// reference's own templates (oracle/_ref/libktref.so).
#pragma once
#include <type_traits>

namespace mi355
{
template <typename T, int PSZ>
__device__ __forceinline__ T kt_hsum(const T (&p)[PSZ])
{
    if constexpr(std::is_same<T, double>::value && PSZ == 4)
        return (p[0] + p[1]) + (p[2] + p[3]);
    else if constexpr(std::is_same<T, double>::value && PSZ == 8)
        return ((p[4] + p[0]) + (p[6] + p[2])) + ((p[5] + p[1]) + (p[7] + p[3]));
    else if constexpr(std::is_same<T, float>::value && PSZ == 8)
        return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    else
    {
        static_assert(std::is_same<T, float>::value && PSZ == 16, "vector widths of the reference: 256 and 512 bits");
        T t3[8], t6[4];
#pragma unroll
        for(int i = 0; i < 8; i++)
            t3[i] = p[8 + i] + p[i];
#pragma unroll
        for(int i = 0; i < 4; i++)
            t6[i] = t3[4 + i] + t3[i];
        return (t6[0] + t6[2]) + (t6[1] + t6[3]);
    }
}
} // namespace mi355
