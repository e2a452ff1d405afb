/*
 * ref_kt_driver.cpp -- TEST INFRASTRUCTURE ONLY (checker of the checker).
 *
 * Exposes the reference's OWN vector micro-kernels ("kernel templates", level 0 / level 1) through a C ABI so that the
 * summation trees restated in oracle.c (orc_kt_hsum_*, the KT TRSV / csrmm rows) can be pinned bit for bit against the
 * code the reference really runs.  The header set library/src/include/kernel-templates/ of /root/reference is
 * self-contained (standard headers + <immintrin.h> only), so it is compiled FROM WHERE IT LIES by oracle/Makefile
 * (target ktref) into oracle/_ref/libktref.so -- no reference source is copied, no stand-in header is written.
 * (The kernels that CALL these templates -- trsv_kt.cpp, csrmm_kt.cpp -- include the cmake-generated version header
 * and AOCL-Utils and stay unbuildable here; the loops below compose the same micro-kernel calls in the order
 * trsv_kt.cpp:92-137 and csrmm_kt.cpp:127-191 make them, with the reference's flags: -O3 -ffp-contract=fast, so the
 * scalar tails are contracted by the compiler exactly as in the reference build.)
 *
 * BUILT only in the build container (needs /root/reference); tests/golden/make_kt_vectors.py turns its outputs into the
 * committed fixture tests/golden/kt_vectors.json.  The built oracle/_ref/libktref.so travels to the GPU box with the snapshot
 * (git-ignored, not gpurun-ignored, as the task prescribes for oracle/_ref) and tests/test_oracle_kt.py loads it there too when
 * present: a checker of the checker, never linked or called by the product.
 */
#include "kernel-templates/kernel_templates.hpp"

#include <cmath>

using namespace kernel_templates;

namespace
{
    bool have512()
    {
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")
               && __builtin_cpu_supports("avx512dq");
    }

    // one row of kt_trsv_l / kt_trsv_u: the micro-kernel sequence of trsv_kt.cpp:96-137
    template <bsz SZ, typename SUF, kt_avxext EXT>
    SUF trsv_row(SUF xi, int cnt, const SUF *a, const SUF *x, const int *icol)
    {
        constexpr int        tsz = tsz_v<SZ, SUF>;
        avxvector_t<SZ, SUF> avec, xvec, pvec;
        int                  rem = cnt % tsz, idx;
        pvec                     = kt_setzero_p<SZ, SUF>();
        for(idx = 0; idx < cnt - rem; idx += tsz)
        {
            avec = kt_loadu_p<SZ, SUF>(&a[idx]);
            xvec = kt_set_p<SZ, SUF>(x, &icol[idx]);
            pvec = kt_fmadd_p<SZ, SUF>(avec, xvec, pvec);
        }
        if(cnt - tsz >= 0)
            xi -= kt_hsum_p<SZ, SUF>(pvec);
        if(rem == tsz - 1)
        {
            idx  = cnt - rem;
            avec = kt_maskz_set_p<SZ, SUF, EXT, tsz - 1>(a, idx);
            xvec = kt_maskz_set_p<SZ, SUF, EXT, tsz - 1>(x, &icol[idx]);
            xi -= kt_dot_p<SZ, SUF>(avec, xvec);
        }
        else
            for(idx = cnt - rem; idx < cnt; idx++)
                xi -= a[idx] * x[icol[idx]];
        return xi;
    }

    // one element of csrmm_col_kt: csrmm_kt.cpp:127-191 for a single column
    template <bsz SZ, typename SUF>
    SUF csrmm_col_elem(int nnz, const SUF *a, const SUF *bcol, const int *icol, SUF alpha, SUF beta, SUF c)
    {
        constexpr int        psz = tsz_v<SZ, SUF>;
        avxvector_t<SZ, SUF> avec, bvec, cvec;
        SUF                  cij = 0.0f;
        int                  mul = nnz / psz, rem = nnz - psz * mul;
        if(mul)
        {
            cvec = kt_setzero_p<SZ, SUF>();
            for(int idx = 0; idx < nnz - rem; idx += psz)
            {
                avec = kt_loadu_p<SZ, SUF>(&a[idx]);
                bvec = kt_set_p<SZ, SUF>(bcol, &icol[idx]);
                cvec = kt_fmadd_p<SZ, SUF>(avec, bvec, cvec);
            }
            cij += kt_hsum_p<SZ, SUF>(cvec);
        }
        if(rem)
            for(int idx = nnz - rem; idx < nnz; idx++)
                cij += a[idx] * bcol[icol[idx]];
        cij *= alpha;
        cij += beta * c;
        return cij;
    }

    // one row of csrmm_row_kt over n columns: csrmm_kt.cpp:244-356 (one entry at a time: groups of four only share loads)
    template <bsz SZ, typename SUF>
    void csrmm_row(int nnz, const SUF *a, const SUF *B, int ldb, const int *icol, int n, SUF alpha, SUF beta, SUF *c)
    {
        constexpr int        psz = tsz_v<SZ, SUF>;
        avxvector_t<SZ, SUF> avec, bvec, cvec;
        int                  rem = n - psz * (n / psz);
        for(int j = 0; j < n; j++)
            c[j] = c[j] * beta;
        for(int k = 0; k < nnz; k++)
        {
            const SUF  sv   = a[k];
            const SUF *brow = B + (size_t)icol[k] * ldb;
            avec            = kt_set1_p<SZ, SUF>(alpha * sv);
            for(int j = 0; j < n - rem; j += psz)
            {
                cvec = kt_loadu_p<SZ, SUF>(&c[j]);
                bvec = kt_loadu_p<SZ, SUF>(&brow[j]);
                cvec = kt_fmadd_p<SZ, SUF>(avec, bvec, cvec);
                kt_storeu_p<SZ, SUF>(&c[j], cvec);
            }
            for(int j = n - rem; j < n; j++)
                c[j] += sv * brow[j] * alpha;
        }
    }
}

extern "C" {

int ktref_have_avx512(void)
{
    return have512() ? 1 : 0;
}

// kt_hsum_p of one register; bits = 256 or 512; returns NaN when the CPU lacks the ISA
double ktref_hsum_d(int bits, const double *v)
{
    if(bits == 256)
        return kt_hsum_p<bsz::b256, double>(kt_loadu_p<bsz::b256, double>(v));
    if(!have512())
        return std::nan("");
    return kt_hsum_p<bsz::b512, double>(kt_loadu_p<bsz::b512, double>(v));
}

float ktref_hsum_s(int bits, const float *v)
{
    if(bits == 256)
        return kt_hsum_p<bsz::b256, float>(kt_loadu_p<bsz::b256, float>(v));
    if(!have512())
        return std::nanf("");
    return kt_hsum_p<bsz::b512, float>(kt_loadu_p<bsz::b512, float>(v));
}

// kt_dot_p of two registers
double ktref_dot_d(int bits, const double *a, const double *b)
{
    if(bits == 256)
        return kt_dot_p<bsz::b256, double>(kt_loadu_p<bsz::b256, double>(a), kt_loadu_p<bsz::b256, double>(b));
    if(!have512())
        return std::nan("");
    return kt_dot_p<bsz::b512, double>(kt_loadu_p<bsz::b512, double>(a), kt_loadu_p<bsz::b512, double>(b));
}

float ktref_dot_s(int bits, const float *a, const float *b)
{
    if(bits == 256)
        return kt_dot_p<bsz::b256, float>(kt_loadu_p<bsz::b256, float>(a), kt_loadu_p<bsz::b256, float>(b));
    if(!have512())
        return std::nanf("");
    return kt_dot_p<bsz::b512, float>(kt_loadu_p<bsz::b512, float>(a), kt_loadu_p<bsz::b512, float>(b));
}

// xi after one KT TRSV row: a[cnt], x gathered through icol[cnt] (zero-based), kid: 1 = b256/AVX2, 2 = b256/AVX512VL, 3 = b512
double ktref_trsv_row_d(int kid, double xi, int cnt, const double *a, const double *x, const int *icol)
{
    if(kid == 1)
        return trsv_row<bsz::b256, double, kt_avxext::AVX2>(xi, cnt, a, x, icol);
    if(!have512())
        return std::nan("");
    if(kid == 2)
        return trsv_row<bsz::b256, double, kt_avxext::AVX512VL>(xi, cnt, a, x, icol);
    return trsv_row<bsz::b512, double, kt_avxext::AVX512F>(xi, cnt, a, x, icol);
}

float ktref_trsv_row_s(int kid, float xi, int cnt, const float *a, const float *x, const int *icol)
{
    if(kid == 1)
        return trsv_row<bsz::b256, float, kt_avxext::AVX2>(xi, cnt, a, x, icol);
    if(!have512())
        return std::nanf("");
    if(kid == 2)
        return trsv_row<bsz::b256, float, kt_avxext::AVX512VL>(xi, cnt, a, x, icol);
    return trsv_row<bsz::b512, float, kt_avxext::AVX512F>(xi, cnt, a, x, icol);
}

double ktref_csrmm_col_elem_d(int bits, int nnz, const double *a, const double *bcol, const int *icol, double alpha,
                              double beta, double c)
{
    if(bits == 256)
        return csrmm_col_elem<bsz::b256, double>(nnz, a, bcol, icol, alpha, beta, c);
    if(!have512())
        return std::nan("");
    return csrmm_col_elem<bsz::b512, double>(nnz, a, bcol, icol, alpha, beta, c);
}

int ktref_csrmm_row_d(int bits, int nnz, const double *a, const double *B, int ldb, const int *icol, int n,
                      double alpha, double beta, double *c)
{
    if(bits == 256)
    {
        csrmm_row<bsz::b256, double>(nnz, a, B, ldb, icol, n, alpha, beta, c);
        return 0;
    }
    if(!have512())
        return 1;
    csrmm_row<bsz::b512, double>(nnz, a, B, ldb, icol, n, alpha, beta, c);
    return 0;
}

float ktref_csrmm_col_elem_s(int bits, int nnz, const float *a, const float *bcol, const int *icol, float alpha, float beta,
                             float c)
{
    if(bits == 256)
        return csrmm_col_elem<bsz::b256, float>(nnz, a, bcol, icol, alpha, beta, c);
    if(!have512())
        return std::nanf("");
    return csrmm_col_elem<bsz::b512, float>(nnz, a, bcol, icol, alpha, beta, c);
}

int ktref_csrmm_row_s(int bits, int nnz, const float *a, const float *B, int ldb, const int *icol, int n, float alpha,
                      float beta, float *c)
{
    if(bits == 256)
    {
        csrmm_row<bsz::b256, float>(nnz, a, B, ldb, icol, n, alpha, beta, c);
        return 0;
    }
    if(!have512())
        return 1;
    csrmm_row<bsz::b512, float>(nnz, a, B, ldb, icol, n, alpha, beta, c);
    return 0;
}
}
