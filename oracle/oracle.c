/*
 * oracle.c -- CPU restatement of the AOCL-Sparse CSR SpMV / TRSV / csrmm / clean-CSR path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity pinned on the reference's own
 * known-answer vectors (tests/golden/); the reference library is unbuildable here.
 *
 * Floating point: the reference is compiled with -ffp-contract=fast (CMakeLists.txt:190)
 * for FMA-capable x86, so every "acc += a*b" below is an explicit fma(); this file is
 * compiled with -ffp-contract=off so nothing else is contracted behind our back.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* Which build of the reference?  Found in round 3 by compiling the reference's kernel    */
/* templates with its own flags: "-O3 -ffp-contract=fast -march=znver2" does NOT fuse a   */
/* LOOP-CARRIED scalar accumulation under GCC (>= 9; checked with 11.4): -march=znver2    */
/* implies -mtune=znver2, whose X86_TUNE_AVOID_128FMA_CHAINS makes tree-ssa-math-opts     */
/* leave "acc += a*b" chains as vmulsd + vaddsd; with -mtune=generic, or with clang /     */
/* AOCC (contraction in the front end), the same statement is one vfmadd.  Everything     */
/* that is not such a chain (beta*y + r, C[k] += a*b*alpha, the explicit _mm*_fmadd       */
/* intrinsics) is fused by both.  So the reference has TWO sets of bits for its scalar    */
/* loops (ref_csrmv_gn, the tails of the AVX kernels, ref_trsv_l/u, the sum loop of       */
/* csrmm_col_major_ref, the KT tails), one per compiler.  orc_set_contract(1) (default):  */
/* fused = clang/AOCC build (what the GPU kernels reproduce bit for bit);                 */
/* orc_set_contract(0): the GCC -march=znver2 build (what oracle/_ref/libktref.so, built  */
/* here with GCC and those flags, produces: tests/golden/kt_vectors.json).                */
/* ------------------------------------------------------------------------------------ */
static int g_fused = 1;
void orc_set_contract(int fused)
{
    g_fused = fused ? 1 : 0;
}
int orc_get_contract(void)
{
    return g_fused;
}
/* a loop-carried scalar accumulation acc = a*b + acc */
#define CH_D(a, b, c) (g_fused ? fma((a), (b), (c)) : ((a) * (b) + (c)))
#define CH_S(a, b, c) (g_fused ? fmaf((a), (b), (c)) : ((a) * (b) + (c)))

/* ------------------------------------------------------------------------------------ */
/* SpMV row kernels.  Each computes one row's dot product in the reference's order.      */
/* ------------------------------------------------------------------------------------ */

/* csrmv_kr.hpp:493-496: result += val[j]*x[col[j]] left to right. */
static inline double row_ref_d(const double *val, const oint *col, const double *x, oint s,
                               oint e, int base)
{
    double r = 0.0;
    for(oint j = s; j < e; j++)
        r = CH_D(val[j - base], x[col[j - base] - base], r);
    return r;
}

static inline float row_ref_s(const float *val, const oint *col, const float *x, oint s,
                              oint e, int base)
{
    float r = 0.0f;
    for(oint j = s; j < e; j++)
        r = CH_S(val[j - base], x[col[j - base] - base], r);
    return r;
}

/* csrmv_kr.hpp:974-1020: 4 lanes FMA over the first floor(n/4)*4 entries, hadd
 * (l0+l1)+(l2+l3), then the scalar tail continues on the reduced value. */
static inline double row_lane4_d(const double *val, const oint *col, const double *x, oint s,
                                 oint e, int base)
{
    oint   n    = e - s;
    oint   krem = n % 4;
    double l[4] = {0.0, 0.0, 0.0, 0.0};
    double r    = 0.0;
    oint   j;
    for(j = s; j < e - krem; j += 4)
        for(int k = 0; k < 4; k++)
            l[k] = fma(val[j + k - base], x[col[j + k - base] - base], l[k]);
    if(n / 4)
        r = (l[0] + l[1]) + (l[2] + l[3]);
    for(j = e - krem; j < e; j++)
        r = CH_D(val[j - base], x[col[j - base] - base], r);
    return r;
}

/* csrmv_avx512.cpp:64-113: 8 lanes; v[k] = l[k]+l[k+4]; (v0+v1)+(v2+v3); scalar tail. */
static inline double row_lane8_d(const double *val, const oint *col, const double *x, oint s,
                                 oint e, int base)
{
    oint   n    = e - s;
    oint   krem = n % 8;
    double l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double r    = 0.0;
    oint   j;
    for(j = s; j < e - krem; j += 8)
        for(int k = 0; k < 8; k++)
            l[k] = fma(val[j + k - base], x[col[j + k - base] - base], l[k]);
    if(n / 8)
    {
        double v0 = l[0] + l[4], v1 = l[1] + l[5], v2 = l[2] + l[6], v3 = l[3] + l[7];
        r = (v0 + v1) + (v2 + v3);
    }
    for(j = e - krem; j < e; j++)
        r = CH_D(val[j - base], x[col[j - base] - base], r);
    return r;
}

/* csrmv_kr.hpp:766-812: float, 8 lanes; ((x0+x4)+(x2+x6)) + ((x1+x5)+(x3+x7)); tail. */
static inline float row_lane8_s(const float *val, const oint *col, const float *x, oint s,
                                oint e, int base)
{
    oint  n    = e - s;
    oint  krem = n % 8;
    float l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float r    = 0.0f;
    oint  j;
    for(j = s; j < e - krem; j += 8)
        for(int k = 0; k < 8; k++)
            l[k] = fmaf(val[j + k - base], x[col[j + k - base] - base], l[k]);
    if(n / 8)
    {
        float q0 = l[0] + l[4], q1 = l[1] + l[5], q2 = l[2] + l[6], q3 = l[3] + l[7];
        float d0 = q0 + q2, d1 = q1 + q3;
        r = d0 + d1;
    }
    for(j = e - krem; j < e; j++)
        r = CH_S(val[j - base], x[col[j - base] - base], r);
    return r;
}

/* csrmv_kr.hpp:497-509: if(alpha!=1) r=alpha*r; if(beta!=0) r += beta*y (contracted). */
static inline double finish_d(double r, double alpha, double beta, const double *yi)
{
    if(alpha != 1.0)
        r = alpha * r;
    if(beta != 0.0)
        r = fma(beta, *yi, r);
    return r;
}

static inline float finish_s(float r, float alpha, float beta, const float *yi)
{
    if(alpha != 1.0f)
        r = alpha * r;
    if(beta != 0.0f)
        r = fmaf(beta, *yi, r);
    return r;
}

#define DEF_DCSRMV(NAME, ROWFN)                                                              \
    int NAME(int base, double alpha, oint m, const double *val, const oint *col,             \
             const oint *row, const double *x, double beta, double *y)                       \
    {                                                                                        \
        for(oint i = 0; i < m; i++)                                                          \
            y[i] = finish_d(ROWFN(val, col, x, row[i], row[i + 1], base), alpha, beta, &y[i]); \
        return ORC_SUCCESS;                                                                  \
    }

DEF_DCSRMV(orc_dcsrmv_ref, row_ref_d)
DEF_DCSRMV(orc_dcsrmv_lane4, row_lane4_d)
DEF_DCSRMV(orc_dcsrmv_lane8, row_lane8_d)

int orc_scsrmv_ref(int base, float alpha, oint m, const float *val, const oint *col,
                   const oint *row, const float *x, float beta, float *y)
{
    for(oint i = 0; i < m; i++)
        y[i] = finish_s(row_ref_s(val, col, x, row[i], row[i + 1], base), alpha, beta, &y[i]);
    return ORC_SUCCESS;
}

int orc_scsrmv_lane8(int base, float alpha, oint m, const float *val, const oint *col,
                     const oint *row, const float *x, float beta, float *y)
{
    for(oint i = 0; i < m; i++)
        y[i] = finish_s(row_lane8_s(val, col, x, row[i], row[i + 1], base), alpha, beta, &y[i]);
    return ORC_SUCCESS;
}

/* csrmv.hpp:322-355: the KAT has kid 0 (ref), 1 and 2 (AVX2), 3 (AVX-512); nnz<=10*m
 * overrides kid to 0; auto (kid<0) resolves to the AVX-512 kernel on an AVX-512 host. */
static int resolve_kid(int kid, oint m, oint nnz)
{
    if((long long)nnz <= 10LL * (long long)m)
        return 0;
    if(kid < 0)
        return 3;
    return kid;
}

int orc_dcsrmv(int kid, int base, double alpha, oint m, oint nnz, const double *val,
               const oint *col, const oint *row, const double *x, double beta, double *y)
{
    switch(resolve_kid(kid, m, nnz))
    {
    case 0:
        return orc_dcsrmv_ref(base, alpha, m, val, col, row, x, beta, y);
    case 1:
    case 2:
        return orc_dcsrmv_lane4(base, alpha, m, val, col, row, x, beta, y);
    case 3:
        return orc_dcsrmv_lane8(base, alpha, m, val, col, row, x, beta, y);
    default:
        return ORC_INVALID_KID;
    }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int orc_dcsrmv_omp(int kid, int base, double alpha, oint m, oint nnz, const double *val,
                   const oint *col, const oint *row, const double *x, double beta, double *y,
                   int nthreads)
{
    int k = resolve_kid(kid, m, nnz);
    if(k > 3)
        return ORC_INVALID_KID;
    if(nthreads < 1)
        nthreads = 1;
    /* csrmv_kr.hpp:483-488: omp parallel for over rows (static schedule). */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for(oint i = 0; i < m; i++)
    {
        double r;
        if(k == 0)
            r = row_ref_d(val, col, x, row[i], row[i + 1], base);
        else if(k == 3)
            r = row_lane8_d(val, col, x, row[i], row[i + 1], base);
        else
            r = row_lane4_d(val, col, x, row[i], row[i + 1], base);
        y[i] = finish_d(r, alpha, beta, &y[i]);
    }
    return ORC_SUCCESS;
}

/* CPU-baseline leg of bench.py: the same OpenMP row split as orc_dcsrmv_omp, but on
 * copies of the arrays that were FIRST TOUCHED by the threads that will read them (same static schedule), so that on a
 * multi-socket host every thread streams from its own NUMA node as a tuned run of the reference would
 * (OMP_PROC_BIND=close / OMP_PLACES=cores are set by the caller before this library is loaded).  Each pass is timed on
 * its own; y of the last pass is returned for the parity check.  The arithmetic is orc_dcsrmv_omp's. */
int orc_dcsrmv_bench(int kid, int base, oint m, oint n, oint nnz, const double *val, const oint *col,
                     const oint *row, const double *x, int nthreads, int passes, double *seconds, double *y_out)
{
    int k = resolve_kid(kid, m, nnz);
    if(k > 3)
        return ORC_INVALID_KID;
    if(nthreads < 1)
        nthreads = 1;
    double *v2 = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    oint   *c2 = (oint *)malloc(sizeof(oint) * (size_t)(nnz > 0 ? nnz : 1));
    oint   *r2 = (oint *)malloc(sizeof(oint) * ((size_t)m + 1));
    double *x2 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *y2 = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    if(!v2 || !c2 || !r2 || !x2 || !y2)
    {
        free(v2), free(c2), free(r2), free(x2), free(y2);
        return ORC_MEMORY_ERROR;
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for(oint i = 0; i < m; i++)
    {
        r2[i] = row[i];
        if(i == m - 1)
            r2[m] = row[m];
        for(oint p = row[i] - base; p < row[i + 1] - base; p++)
            v2[p] = val[p], c2[p] = col[p];
        y2[i] = 0.0;
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for(oint j = 0; j < n; j++)
        x2[j] = x[j];
    if(m == 0)
        r2[0] = row[0];
    for(int t = 0; t < passes; t++)
    {
#ifdef _OPENMP
        const double t0 = omp_get_wtime();
#endif
        orc_dcsrmv_omp(k, base, 1.0, m, nnz, v2, c2, r2, x2, 0.0, y2, nthreads);
#ifdef _OPENMP
        seconds[t] = omp_get_wtime() - t0;
#else
        seconds[t] = 0.0;
#endif
    }
    if(y_out)
        memcpy(y_out, y2, sizeof(double) * (size_t)m);
    free(v2), free(c2), free(r2), free(x2), free(y2);
    return ORC_SUCCESS;
}

/* csrmv_kt.cpp:96-214 with one thread: scale y (beta==0 writes zeros, beta==1 untouched),
 * then row by row y[col] += val * (alpha*x[i]).  The vector body multiplies then adds
 * (kt_mul_p then +=, :184-193); the tail "y += aval*alpha*x[i]" is contracted to an fma of
 * (aval*alpha) and x[i] (:195-199).  With one thread there is no merge step. */
int orc_dcsrmvt(int base, double alpha, oint m, oint n, const double *val, const oint *col,
                const oint *row, const double *x, double beta, double *y)
{
    if(beta == 0.0)
        for(oint i = 0; i < n; i++)
            y[i] = 0.0;
    else if(beta != 1.0)
        for(oint i = 0; i < n; i++)
            y[i] = beta * y[i];
    /* the OpenMP build accumulates into a zeroed per-thread buffer and merges it into y
     * afterwards (:150-160, :203-210); with one thread that is buf then y += buf. */
    double *buf = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    if(!buf)
        return ORC_MEMORY_ERROR;
    const oint tsz = 8; /* b512 doubles on an AVX-512 host */
    for(oint i = 0; i < m; i++)
    {
        oint   s = row[i], e = row[i + 1];
        oint   krem = (e - s) % tsz;
        double ax   = alpha * x[i];
        oint   j;
        for(j = s; j < e - krem; j++)
            buf[col[j - base] - base] += val[j - base] * ax;
        for(j = e - krem; j < e; j++)
            buf[col[j - base] - base]
                = fma(val[j - base] * alpha, x[i], buf[col[j - base] - base]);
    }
    for(oint i = 0; i < n; i++)
        y[i] += buf[i];
    free(buf);
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Symmetric / triangular SpMV reference kernels.                                        */
/* ------------------------------------------------------------------------------------ */
static void scale_y_d(double *y, oint n, double beta)
{
    if(beta == 0.0)
        for(oint i = 0; i < n; i++)
            y[i] = 0.0;
    else if(beta != 1.0)
        for(oint i = 0; i < n; i++)
            y[i] = beta * y[i];
}

/* csrmv_kr.hpp:41-92 (aoclsparse_csrmv_symm, the raw-array csrmv with a symmetric descriptor):
 * every stored entry except a diagonal in LAST position of its row is applied twice. */
int orc_dcsrmv_symm_raw(int base, double alpha, oint m, const double *val, const oint *col,
                        const oint *row, const double *x, double beta, double *y)
{
    scale_y_d(y, m, beta);
    for(oint i = 0; i < m; i++)
    {
        oint didx = row[i + 1] - base - 1;
        int  last = (row[i + 1] > row[i]) && (col[didx] - base == i); /* reference reads col[-1] on empty rows */
        if(last)
            y[i] = fma(alpha * val[didx], x[i], y[i]); /* last*alpha*val*x[i], contracted */
        oint end = row[i + 1] - base - last;
        for(oint j = row[i] - base; j < end; j++)
        {
            oint c = col[j] - base;
            y[i]   = fma(alpha * val[j], x[c], y[i]);
            y[c]   = fma(alpha * val[j], x[i], y[c]);
        }
    }
    return ORC_SUCCESS;
}

/* csrmv_kr.hpp:107-186 (aoclsparse_csrmv_symm_internal) on the clean CSR: fill 0 lower / 1 upper,
 * diag 0 non_unit / 1 unit / 2 zero; istart/iend select the strict triangle via idiag / iurow. */
int orc_dcsrmv_symm(int base, double alpha, oint m, int diag, int fill, const double *val,
                    const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
                    const double *x, double beta, double *y)
{
    scale_y_d(y, m, beta);
    for(oint i = 0; i < m; i++)
    {
        oint   s = fill == 0 ? ptr[i] : iurow[i], e = fill == 0 ? idiag[i] : ptr[i + 1];
        double xv = x[i], sum = 0.0;
        for(oint j = s; j < e; j++)
        {
            oint   c = col[j - base] - base;
            double v = alpha * val[j - base];
            sum      = fma(v, x[c], sum);
            y[c]     = fma(v, xv, y[c]);
        }
        if(diag == 0)
            sum = fma(alpha * val[idiag[i] - base], xv, sum);
        else if(diag == 1)
            sum = fma(alpha, xv, sum);
        y[i] += sum;
    }
    return ORC_SUCCESS;
}

/* csrmv_kr.hpp:658-728 (ref_csrmv_tri): rows [rs[i], re[i]) of the clean CSR, i.e. strict triangle
 * plus the stored diagonal; unit/zero diag drop the stored diagonal, unit adds x[i]. */
int orc_dcsrmv_tri(int base, double alpha, oint m, int diag, int fill, const double *val,
                   const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
                   const double *x, double beta, double *y)
{
    (void)iurow;
    scale_y_d(y, m, beta);
    for(oint i = 0; i < m; i++)
    {
        /* lower: [ptr[i], iurow[i]) ; upper: [idiag[i], ptr[i+1]) (csrmv.hpp:110-123) */
        oint rs = fill == 0 ? ptr[i] : idiag[i], re = fill == 0 ? iurow[i] : ptr[i + 1];
        int  so = 0, eo = 0;
        if(diag != 0)
        {
            if(fill == 0)
                eo = -1;
            else
                so = 1;
        }
        double r = 0.0;
        if(so && diag == 1)
            r += x[i];
        for(oint j = rs + so; j < re + eo; j++)
            r = CH_D(val[j - base], x[col[j - base] - base], r);
        if(eo && diag == 1)
            r += x[i];
        y[i] = fma(alpha, r, y[i]);
    }
    return ORC_SUCCESS;
}

/* csrmv_kr.hpp:577-649 (ref_csrmv_tri_th): transposed triangular SpMV, column sweep. */
int orc_dcsrmv_tri_t(int base, double alpha, oint m, oint n, int diag, int fill, const double *val,
                     const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
                     const double *x, double beta, double *y)
{
    scale_y_d(y, n, beta);
    for(oint i = 0; i < m; i++)
    {
        oint rs = fill == 0 ? ptr[i] : idiag[i], re = fill == 0 ? iurow[i] : ptr[i + 1];
        int  so = 0, eo = 0;
        if(diag != 0)
        {
            if(fill == 0)
                eo = -1;
            else
                so = 1;
        }
        double axi = alpha * x[i];
        if(so && diag == 1)
            y[i] += axi;
        for(oint j = rs + so; j < re + eo; j++)
        {
            oint c = col[j - base] - base;
            y[c]   = fma(val[j - base], axi, y[c]);
        }
        if(eo && diag == 1)
            y[i] += axi;
    }
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* TRSV reference kernels, trsv_kr.hpp:38-222.  "xi -= a*x" contracts to fma(-a, x, xi). */
/* ------------------------------------------------------------------------------------ */
#define DEF_TRSV(T, SUF, FMA, CH)                                                                \
    int orc_##SUF##trsv_l(T alpha, oint m, int base, const T *a, const oint *icol,           \
                          const oint *ilrow, const oint *idiag, const T *b, oint incb, T *x, \
                          oint incx, int unit)                                               \
    {                                                                                        \
        for(oint i = 0; i < m; i++)                                                          \
        {                                                                                    \
            T xi = alpha * b[(size_t)i * incb];                                              \
            for(oint idx = ilrow[i]; idx < idiag[i]; idx++)                                  \
                xi = CH(-a[idx - base], x[(size_t)(icol[idx - base] - base) * incx], xi);    \
            if(!unit)                                                                        \
                xi /= a[idiag[i] - base];                                                    \
            x[(size_t)i * incx] = xi;                                                        \
        }                                                                                    \
        return ORC_SUCCESS;                                                                  \
    }                                                                                        \
    int orc_##SUF##trsv_u(T alpha, oint m, int base, const T *a, const oint *icol,           \
                          const oint *ilrow, const oint *iurow, const T *b, oint incb, T *x, \
                          oint incx, int unit)                                               \
    {                                                                                        \
        for(oint i = m - 1; i >= 0; i--)                                                     \
        {                                                                                    \
            T xi = alpha * b[(size_t)i * incb];                                              \
            for(oint idx = iurow[i]; idx <= ilrow[i + 1] - 1; idx++)                         \
                xi = CH(-a[idx - base], x[(size_t)(icol[idx - base] - base) * incx], xi);    \
            if(!unit)                                                                        \
                xi /= a[iurow[i] - 1 - base];                                                \
            x[(size_t)i * incx] = xi;                                                        \
        }                                                                                    \
        return ORC_SUCCESS;                                                                  \
    }

DEF_TRSV(double, d, fma, CH_D)
DEF_TRSV(float, s, fmaf, CH_S)

/* trsv_kr.hpp:101-120: x = alpha*b; for i = m-1..0: x[i] /= d; x[col] -= a*x[i]. */
int orc_dtrsv_lt(double alpha, oint m, int base, const double *a, const oint *icol,
                 const oint *ilrow, const oint *idiag, const double *b, oint incb, double *x,
                 oint incx, int unit)
{
    for(oint i = 0; i < m; i++)
        x[(size_t)i * incx] = alpha * b[(size_t)i * incb];
    for(oint i = m - 1; i >= 0; i--)
    {
        if(!unit)
            x[(size_t)i * incx] /= a[idiag[i] - base];
        double xi = x[(size_t)i * incx];
        for(oint idx = ilrow[i]; idx < idiag[i]; idx++)
        {
            size_t c = (size_t)(icol[idx - base] - base) * incx;
            x[c]     = fma(-a[idx - base], xi, x[c]);
        }
    }
    return ORC_SUCCESS;
}

/* trsv_kr.hpp:196-221: x = alpha*b; for i = 0..m-1: x[i] /= d; x[col] -= a*x[i]. */
int orc_dtrsv_ut(double alpha, oint m, int base, const double *a, const oint *icol,
                 const oint *ilrow, const oint *iurow, const double *b, oint incb, double *x,
                 oint incx, int unit)
{
    for(oint i = 0; i < m; i++)
        x[(size_t)i * incx] = alpha * b[(size_t)i * incb];
    for(oint i = 0; i < m; i++)
    {
        if(!unit)
            x[(size_t)i * incx] /= a[iurow[i] - 1 - base];
        double xi = x[(size_t)i * incx];
        for(oint idx = iurow[i]; idx <= ilrow[i + 1] - 1; idx++)
        {
            size_t c = (size_t)(icol[idx - base] - base) * incx;
            x[c]     = fma(-a[idx - base], xi, x[c]);
        }
    }
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* KT ("kernel template") TRSV kernels, level2/aoclsparse_trsv_kt.cpp:64-531 -- what the  */
/* reference dispatches for kid 1/2 (256-bit vectors) and kid 3 / auto on an AVX-512 host  */
/* (512-bit vectors), trsv.cpp:321-353.  tsz = lanes per vector: double 4 / 8, float 8 / 16.*/
/* ------------------------------------------------------------------------------------ */

/* kt_hsum_p: horizontal sum of one vector register.
 *  b256 double (kernel-templates/kt_l0_avx2.hpp:333-340): hadd -> (v0+v1, v2+v3), lo + hi.
 *  b256 float  (:342-350): hadd, hadd, lo + hi = ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7)).
 *  b512 (kt_l0_avx512.hpp:369-376) = _mm512_reduce_add_pd/ps, a compiler-header sequence, not an
 *  instruction; GCC's (avx512fintrin.h __MM512_REDUCE_OP): halves added lane-wise until two lanes
 *  are left: double ((v0+v4)+(v2+v6)) + ((v1+v5)+(v3+v7)); float 16 -> 8 -> 4 -> 2 -> 1 likewise.
 * Pinned bit for bit against the reference's own templates compiled from /root/reference
 * (oracle/_ref/libktref.so, tests/golden/kt_vectors.json). */
double orc_kt_hsum_d(int tsz, const double *v)
{
    if(tsz == 4)
        return (v[0] + v[1]) + (v[2] + v[3]);
    double t3[4], t6[2];
    for(int i = 0; i < 4; i++)
        t3[i] = v[4 + i] + v[i];
    for(int i = 0; i < 2; i++)
        t6[i] = t3[2 + i] + t3[i];
    return t6[0] + t6[1];
}

float orc_kt_hsum_s(int tsz, const float *v)
{
    if(tsz == 8)
        return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    float t3[8], t6[4], t8[2];
    for(int i = 0; i < 8; i++)
        t3[i] = v[8 + i] + v[i];
    for(int i = 0; i < 4; i++)
        t6[i] = t3[4 + i] + t3[i];
    for(int i = 0; i < 2; i++)
        t8[i] = t6[i] + t6[i + 2];
    return t8[0] + t8[1];
}

/* One row of kt_trsv_l / kt_trsv_u (trsv_kt.cpp:92-137, :324-371): the entries idx0 .. idx0+cnt-1 (already
 * offset by the base) are consumed as
 *   full groups of tsz:  pvec[l] = fma(a, x, pvec[l])            (kt_fmadd_p, :109)
 *   if cnt >= tsz:       xi -= hsum(pvec)                        (:111-114)
 *   rem == tsz-1:        xi -= hsum(a .* x, last lane = 0*0)     (kt_maskz_set_p + kt_dot_p = mul, hsum; :121-131)
 *   otherwise:           xi = fma(-a, x, xi) left to right       (the contracted "xi -= a*x", :136-137)
 */
#define DEF_KT_TRSV(T, SUF, FMA, CH, MAXL)                                                                \
    static T kt_row_##SUF(int tsz, T xi, oint idx0, oint cnt, const T *a, const oint *icol, int base,  \
                          const T *x, oint incx)                                                       \
    {                                                                                                  \
        T    p[MAXL];                                                                                  \
        oint rem = cnt % tsz, idx = idx0;                                                              \
        for(int l = 0; l < tsz; l++)                                                                   \
            p[l] = 0;                                                                                  \
        for(; idx < idx0 + cnt - rem; idx += tsz)                                                      \
            for(int l = 0; l < tsz; l++)                                                               \
                p[l] = FMA(a[idx + l], x[(size_t)(icol[idx + l] - base) * incx], p[l]);                \
        if(cnt - tsz >= 0)                                                                             \
            xi -= orc_kt_hsum_##SUF(tsz, p);                                                           \
        if(rem == tsz - 1)                                                                             \
        {                                                                                              \
            for(int l = 0; l < tsz - 1; l++)                                                           \
                p[l] = a[idx + l] * x[(size_t)(icol[idx + l] - base) * incx];                          \
            p[tsz - 1] = (T)0 * (T)0;                                                                  \
            xi -= orc_kt_hsum_##SUF(tsz, p);                                                           \
        }                                                                                              \
        else                                                                                           \
            for(; idx < idx0 + cnt; idx++)                                                             \
                xi = CH(-a[idx], x[(size_t)(icol[idx] - base) * incx], xi);                            \
        return xi;                                                                                     \
    }                                                                                                  \
    /* kt_trsv_l, trsv_kt.cpp:64-150: forward rows; division after the chain (:140-148) */            \
    int orc_##SUF##trsv_kt_l(int tsz, T alpha, oint m, int base, const T *a, const oint *icol,         \
                             const oint *ilrow, const oint *idiag, const T *b, oint incb, T *x,        \
                             oint incx, int unit)                                                      \
    {                                                                                                  \
        for(oint i = 0; i < m; i++)                                                                    \
        {                                                                                              \
            T xi = alpha * b[(size_t)i * incb];                                                        \
            xi   = kt_row_##SUF(tsz, xi, ilrow[i] - base, idiag[i] - ilrow[i], a, icol, base, x, incx);\
            if(!unit)                                                                                  \
                xi /= a[idiag[i] - base];                                                              \
            x[(size_t)i * incx] = xi;                                                                  \
        }                                                                                              \
        return ORC_SUCCESS;                                                                            \
    }                                                                                                  \
    /* kt_trsv_u, trsv_kt.cpp:297-383: backward rows over [iurow[i], ilrow[i+1]-1] */                 \
    int orc_##SUF##trsv_kt_u(int tsz, T alpha, oint m, int base, const T *a, const oint *icol,         \
                             const oint *ilrow, const oint *iurow, const T *b, oint incb, T *x,        \
                             oint incx, int unit)                                                      \
    {                                                                                                  \
        for(oint i = m - 1; i >= 0; i--)                                                               \
        {                                                                                              \
            T xi = alpha * b[(size_t)i * incb];                                                        \
            xi   = kt_row_##SUF(tsz, xi, iurow[i] - base, ilrow[i + 1] - iurow[i], a, icol, base, x,   \
                                incx);                                                                 \
            if(!unit)                                                                                  \
                xi /= a[iurow[i] - 1 - base];                                                          \
            x[(size_t)i * incx] = xi;                                                                  \
        }                                                                                              \
        return ORC_SUCCESS;                                                                            \
    }                                                                                                  \
    /* kt_trsv_lt / kt_trsv_ut, trsv_kt.cpp:183-268, :416-503: x = alpha*b (only when alpha != 0, :208-210); per row */ \
    /* x[i] /= d, then every entry x[col] = fma(a, -x[i], x[col]) -- vector groups (:236-238), the masked group  */   \
    /* (:254-258) and the contracted scalar tail "x[col] -= a*xi" (:264-266) are the same per-element operation, */   \
    /* so the tsz argument changes nothing: these two equal ref_trsv_lth / ref_trsv_uth bit for bit.             */   \
    int orc_##SUF##trsv_kt_lt(int tsz, T alpha, oint m, int base, const T *a, const oint *icol,        \
                              const oint *ilrow, const oint *idiag, const T *b, oint incb, T *x,       \
                              oint incx, int unit)                                                     \
    {                                                                                                  \
        (void)tsz;                                                                                     \
        if(alpha != (T)0)                                                                              \
            for(oint i = 0; i < m; i++)                                                                \
                x[(size_t)i * incx] = alpha * b[(size_t)i * incb];                                     \
        for(oint i = m - 1; i >= 0; i--)                                                               \
        {                                                                                              \
            if(!unit)                                                                                  \
                x[(size_t)i * incx] /= a[idiag[i] - base];                                             \
            T mxi = -x[(size_t)i * incx];                                                              \
            for(oint idx = ilrow[i] - base; idx < idiag[i] - base; idx++)                              \
            {                                                                                          \
                size_t c = (size_t)(icol[idx] - base) * incx;                                          \
                x[c]     = FMA(a[idx], mxi, x[c]);                                                     \
            }                                                                                          \
        }                                                                                              \
        return ORC_SUCCESS;                                                                            \
    }                                                                                                  \
    int orc_##SUF##trsv_kt_ut(int tsz, T alpha, oint m, int base, const T *a, const oint *icol,        \
                              const oint *ilrow, const oint *iurow, const T *b, oint incb, T *x,       \
                              oint incx, int unit)                                                     \
    {                                                                                                  \
        (void)tsz;                                                                                     \
        if(alpha != (T)0)                                                                              \
            for(oint i = 0; i < m; i++)                                                                \
                x[(size_t)i * incx] = alpha * b[(size_t)i * incb];                                     \
        for(oint i = 0; i < m; i++)                                                                    \
        {                                                                                              \
            if(!unit)                                                                                  \
                x[(size_t)i * incx] = x[(size_t)i * incx] / a[iurow[i] - 1 - base];                    \
            T mxi = -x[(size_t)i * incx];                                                              \
            for(oint idx = iurow[i] - base; idx <= ilrow[i + 1] - 1 - base; idx++)                     \
            {                                                                                          \
                size_t c = (size_t)(icol[idx] - base) * incx;                                          \
                x[c]     = FMA(a[idx], mxi, x[c]);                                                     \
            }                                                                                          \
        }                                                                                              \
        return ORC_SUCCESS;                                                                            \
    }

DEF_KT_TRSV(double, d, fma, CH_D, 8)
DEF_KT_TRSV(float, s, fmaf, CH_S, 16)

/* ------------------------------------------------------------------------------------ */
/* csrmm reference kernels, csrmm.hpp:36-144.                                            */
/* ------------------------------------------------------------------------------------ */

/* csrmm.hpp:69-85: sum = aval*B + sum (fma chain); C = (beta*C) + (alpha*sum), which the
 * compiler contracts to fma(beta, C, alpha*sum). */
int orc_dcsrmm_col(double alpha, int base, const double *val, const oint *col,
                   const oint *row, oint m, const double *B, oint n, oint ldb, double beta,
                   double *C, oint ldc)
{
    for(oint j = 0; j < n; j++)
        for(oint i = 0; i < m; i++)
        {
            double sum = 0.0;
            for(oint k = row[i]; k < row[i + 1]; k++)
                sum = CH_D(val[k - base], B[(size_t)(col[k - base] - base) + (size_t)j * ldb], sum);
            size_t ic = (size_t)i + (size_t)j * ldc;
            C[ic]     = fma(beta, C[ic], alpha * sum);
        }
    return ORC_SUCCESS;
}

/* csrmm.hpp:123-139: C_row *= beta; then per nnz in CSR order C += (aval*B)*alpha,
 * contracted to fma(aval*B, alpha, C). */
int orc_dcsrmm_row(double alpha, int base, const double *val, const oint *col,
                   const oint *row, oint m, const double *B, oint n, oint ldb, double beta,
                   double *C, oint ldc)
{
    for(oint i = 0; i < m; i++)
    {
        double *c = C + (size_t)i * ldc;
        for(oint k = 0; k < n; k++)
            c[k] = c[k] * beta;
        for(oint j = row[i]; j < row[i + 1]; j++)
        {
            const double *brow = B + (size_t)(col[j - base] - base) * ldb;
            double        av   = val[j - base];
            for(oint k = 0; k < n; k++)
                c[k] = fma(av * brow[k], alpha, c[k]);
        }
    }
    return ORC_SUCCESS;
}

/* csrmm_col_kt, level3/aoclsparse_csrmm_kt.cpp:31-197 (kid 1/2: psz = 4 lanes, kid 3 / auto on an AVX-512 host: 8).
 * Per element (i, j), four columns at a time share the loads but not the arithmetic:
 *   nnz >= psz: cvec[l] = fma(a, b, cvec[l]) over the full groups (:144-147), cij = 0 + hsum(cvec) (:149-156);
 *   tail:       cij = fma(a, b, cij) left to right (the contracted "cij += aval * b", :165-168);
 *   cij *= alpha; C = beta*C + cij (:172-191) -- one fma, which product is fused depends on the compiler (below).
 * C is read even when beta == 0. */
#define DEF_CSRMM_COL_KT(NAME, T, FMA, HSUM, CH, MAXP)                                                          \
    int NAME(int psz, T alpha, int base, const T *val, const oint *col, const oint *row, oint m, const T *B, oint n, \
             oint ldb, T beta, T *C, oint ldc)                                                                      \
    {                                                                                                               \
        for(oint j = 0; j < n; j++)                                                                                 \
            for(oint i = 0; i < m; i++)                                                                             \
            {                                                                                                       \
                const T *bc  = B + (size_t)j * ldb;                                                                 \
                oint     s   = row[i] - base, e = row[i + 1] - base;                                                \
                oint     nnz = e - s, mul = nnz / psz, rem = nnz - psz * mul;                                       \
                T        cij = 0;                                                                                   \
                if(mul)                                                                                             \
                {                                                                                                   \
                    T p[MAXP];                                                                                      \
                    for(int l = 0; l < MAXP; l++)                                                                   \
                        p[l] = 0;                                                                                   \
                    for(oint k = s; k < e - rem; k += psz)                                                          \
                        for(int l = 0; l < psz; l++)                                                                \
                            p[l] = FMA(val[k + l], bc[col[k + l] - base], p[l]);                                    \
                    cij += HSUM(psz, p);                                                                            \
                }                                                                                                   \
                for(oint k = e - rem; k < e; k++)                                                                   \
                    cij = CH(val[k], bc[col[k] - base], cij);                                                       \
                size_t ic = (size_t)i + (size_t)j * ldc;                                                            \
                /* "cij *= alpha; C = beta*C + cij": two products feed one addition and the compiler picks which    \
                 * one it fuses.  Statement-wise contraction (clang -ffp-contract=on style):          \
                 * fma(beta, C, alpha*cij); GCC's widening_mul pass takes the first product it meets:               \
                 * fma(cij, alpha, beta*C) -- measured on oracle/_ref/libktref.so (tests/golden/kt_vectors.json). */ \
                if(g_fused)                                                                                         \
                    C[ic] = FMA(beta, C[ic], cij * alpha);                                                          \
                else                                                                                                \
                    C[ic] = FMA(cij, alpha, beta * C[ic]);                                                          \
            }                                                                                                       \
        return ORC_SUCCESS;                                                                                         \
    }
DEF_CSRMM_COL_KT(orc_dcsrmm_col_kt, double, fma, orc_kt_hsum_d, CH_D, 8)
DEF_CSRMM_COL_KT(orc_scsrmm_col_kt, float, fmaf, orc_kt_hsum_s, CH_S, 16)

/* csrmm_row_kt, csrmm_kt.cpp:199-363.  C_row = C_row * beta first (:244-247, a multiplication also for beta == 0);
 * then the row's entries in CSR order (groups of four only share loads): columns j < n - n % psz take
 * c = fma(alpha*a_k, b_kj, c) (kt_set1_p(alpha*sv), kt_fmadd_p, :289-322), the last n % psz columns the scalar
 * statement "C += sv * B * alpha" (:335-356) = fma(sv*b, alpha, c) after contraction. */
#define DEF_CSRMM_ROW_KT(NAME, T, FMA)                                                                              \
    int NAME(int psz, T alpha, int base, const T *val, const oint *col, const oint *row, oint m, const T *B, oint n, \
             oint ldb, T beta, T *C, oint ldc)                                                                      \
    {                                                                                                               \
        oint rem = n % psz;                                                                                         \
        for(oint i = 0; i < m; i++)                                                                                 \
        {                                                                                                           \
            T *c = C + (size_t)i * ldc;                                                                             \
            for(oint j = 0; j < n; j++)                                                                             \
                c[j] = c[j] * beta;                                                                                 \
            for(oint k = row[i] - base; k < row[i + 1] - base; k++)                                                 \
            {                                                                                                       \
                const T *brow = B + (size_t)(col[k] - base) * ldb;                                                  \
                T        sv = val[k], av = alpha * sv;                                                              \
                for(oint j = 0; j < n - rem; j++)                                                                   \
                    c[j] = FMA(av, brow[j], c[j]);                                                                  \
                for(oint j = n - rem; j < n; j++)                                                                   \
                    c[j] = FMA(sv * brow[j], alpha, c[j]);                                                          \
            }                                                                                                       \
        }                                                                                                           \
        return ORC_SUCCESS;                                                                                         \
    }
DEF_CSRMM_ROW_KT(orc_dcsrmm_row_kt, double, fma)
DEF_CSRMM_ROW_KT(orc_scsrmm_row_kt, float, fmaf)

/* csrmm.hpp:361-427: beta==0 writes exact zeros, otherwise C *= beta. */
int orc_dscale_dense(int order, double *C, oint m, oint n, oint ld, double beta)
{
    oint outer = order == 1 ? n : m;
    oint inner = order == 1 ? m : n;
    for(oint o = 0; o < outer; o++)
        for(oint i = 0; i < inner; i++)
        {
            size_t idx = (size_t)o * ld + i;
            C[idx]     = beta == 0.0 ? 0.0 : C[idx] * beta;
        }
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* clean-CSR path                                                                        */
/* ------------------------------------------------------------------------------------ */

/* csr_util.cpp:124-279 */
int orc_mat_check(oint maj, oint mind, oint nnz, const oint *ptr, const oint *ind,
                  const void *val, int shape, int base, int *sort_out, int *fulldiag_out)
{
    if(!ptr || !ind || !val)
        return ORC_INVALID_POINTER;
    if(mind < 0 || maj < 0 || nnz < 0)
        return ORC_INVALID_SIZE;
    if(ptr[0] - base != 0)
        return ORC_INVALID_VALUE;
    if(ptr[maj] - base != nnz)
        return ORC_INVALID_VALUE;
    for(oint i = 1; i <= maj; i++)
        if(ptr[i - 1] > ptr[i])
            return ORC_INVALID_VALUE;

    int  sort     = ORC_FULLY_SORTED;
    int  fulldiag = 1;
    oint jmin = 0, jmax = mind - 1;
    for(oint i = 0; i < maj; i++)
    {
        oint idxend = ptr[i + 1] - base, idxstart = ptr[i] - base;
        if(shape == 1)
        {
            jmin = 0;
            jmax = i;
        }
        else if(shape == 2)
        {
            jmin = i;
            jmax = mind - 1;
        }
        int  diagonal = 0, upper = 0;
        oint prev = -1;
        for(oint idx = idxstart; idx < idxend; idx++)
        {
            oint j = ind[idx] - base;
            if(j < jmin || j > jmax)
                return ORC_INVALID_INDEX_VALUE;
            if(sort != ORC_UNSORTED)
            {
                if(prev > j)
                    sort = ORC_PARTIALLY_SORTED;
                else
                    prev = j;
                if((j <= i && upper) || (j < i && diagonal))
                    sort = ORC_UNSORTED;
            }
            if(j > i)
                upper = 1;
            else if(j == i)
            {
                if(diagonal)
                    return ORC_INVALID_VALUE;
                diagonal = 1;
            }
        }
        if(!diagonal && i < mind)
            fulldiag = 0;
    }
    *sort_out     = sort;
    *fulldiag_out = fulldiag;
    return ORC_SUCCESS;
}

/* csr_util.cpp:290-364 */
int orc_check_sort_diag(oint m, oint n, int base, const oint *ptr, const oint *ind,
                        int *sorted, int *fulldiag)
{
    *sorted   = 0;
    *fulldiag = 0;
    if(m < 0 || n < 0)
        return ORC_INVALID_SIZE;
    if(!ptr || !ind)
        return ORC_INVALID_POINTER;
    *sorted   = 1;
    *fulldiag = 1;
    for(oint i = 0; i < m; i++)
    {
        int  lower = 1, found = 0;
        oint idxend = ptr[i + 1] - base;
        for(oint idx = ptr[i] - base; idx < idxend; idx++)
        {
            oint j = ind[idx] - base;
            if(j == i)
            {
                if(found)
                    return ORC_INVALID_VALUE;
                found   = 1;
                *sorted = lower;
                lower   = 0;
            }
            else
            {
                if(lower)
                    lower = j < i;
                else
                    *sorted = *sorted && (j > i);
            }
            if(!*sorted)
            {
                *fulldiag = 0;
                return ORC_SUCCESS;
            }
        }
        if(!found && i < n)
            *fulldiag = 0;
        if(!*sorted)
        {
            *fulldiag = 0;
            return ORC_SUCCESS;
        }
    }
    return ORC_SUCCESS;
}

/* csr_util.cpp:389-458 */
int orc_csr_indices(oint m, int base, const oint *ptr, const oint *ind, oint *idiag,
                    oint *iurow)
{
    if(m < 0)
        return ORC_INVALID_SIZE;
    if(!ptr || !ind || !idiag || !iurow)
        return ORC_INVALID_POINTER;
    for(oint i = 0; i < m; i++)
    {
        int  found  = 0;
        oint idxend = ptr[i + 1] - base;
        for(oint idx = ptr[i] - base; idx < idxend; idx++)
        {
            oint j = ind[idx] - base;
            if(j >= i)
            {
                oint adj = idx + base;
                idiag[i] = adj;
                iurow[i] = j == i ? adj + 1 : adj;
                found    = 1;
                break;
            }
        }
        if(!found)
        {
            idiag[i] = idxend + base;
            iurow[i] = idxend + base;
        }
    }
    return ORC_SUCCESS;
}

/* csr_util.hpp:100-159: per-row sort of (index,value) by index.  The reference sorts a
 * permutation with std::sort and a "<=" comparator, so the relative order of duplicate
 * column indices is implementation-defined; rows without duplicates have a unique result,
 * which is what this stable insertion/merge sort produces. */
typedef struct
{
    oint   c;
    double v;
} cv_t;

static int cv_cmp(const void *a, const void *b)
{
    oint ca = ((const cv_t *)a)->c, cb = ((const cv_t *)b)->c;
    return (ca > cb) - (ca < cb);
}

/* csr_util.hpp:765-967 */
int orc_dcsr_optimize(oint m, oint n, oint nnz, int base, const oint *ptr, const oint *ind,
                      const double *val, oint *optr, oint *oind, double *oval, oint *onnz,
                      oint *idiag, oint *iurow, int *is_internal, int *fulldiag_out)
{
    int sort, fd, sorted, fulldiag;
    int st = orc_mat_check(m, n, nnz, ptr, ind, val, 0, base, &sort, &fd);
    if(st != ORC_SUCCESS)
        return st;
    st = orc_check_sort_diag(m, n, base, ptr, ind, &sorted, &fulldiag);
    if(st != ORC_SUCCESS)
        return ORC_INTERNAL_ERROR;
    if(sorted && fulldiag)
    {
        /* user's memory is used as is (base preserved) */
        *is_internal  = 0;
        *onnz         = nnz;
        *fulldiag_out = fulldiag;
        return orc_csr_indices(m, base, ptr, ind, idiag, iurow);
    }
    *is_internal = 1;
    /* 0-based copy (:893-902) */
    oint   *tptr = (oint *)malloc(sizeof(oint) * ((size_t)m + 1));
    oint   *tind = (oint *)malloc(sizeof(oint) * ((size_t)nnz + 1));
    double *tval = (double *)malloc(sizeof(double) * ((size_t)nnz + 1));
    if(!tptr || !tind || !tval)
    {
        free(tptr);
        free(tind);
        free(tval);
        return ORC_MEMORY_ERROR;
    }
    for(oint i = 0; i <= m; i++)
        tptr[i] = ptr[i] - base;
    for(oint i = 0; i < nnz; i++)
    {
        tind[i] = ind[i] - base;
        tval[i] = val[i];
    }
    if(!sorted)
    {
        /* aoclsparse_sort_idx_val (:904-916) then re-check (:918-924) */
        oint  maxrow = 0;
        for(oint i = 0; i < m; i++)
            if(tptr[i + 1] - tptr[i] > maxrow)
                maxrow = tptr[i + 1] - tptr[i];
        cv_t *buf = (cv_t *)malloc(sizeof(cv_t) * ((size_t)maxrow + 1));
        if(!buf)
        {
            free(tptr);
            free(tind);
            free(tval);
            return ORC_MEMORY_ERROR;
        }
        for(oint i = 0; i < m; i++)
        {
            oint s = tptr[i], len = tptr[i + 1] - tptr[i];
            for(oint k = 0; k < len; k++)
            {
                buf[k].c = tind[s + k];
                buf[k].v = tval[s + k];
            }
            qsort(buf, (size_t)len, sizeof(cv_t), cv_cmp);
            for(oint k = 0; k < len; k++)
            {
                tind[s + k] = buf[k].c;
                tval[s + k] = buf[k].v;
            }
        }
        free(buf);
        st = orc_check_sort_diag(m, n, 0, tptr, tind, &sorted, &fulldiag);
        if(st != ORC_SUCCESS)
        {
            free(tptr);
            free(tind);
            free(tval);
            return st;
        }
    }
    oint newnnz = nnz;
    if(!fulldiag)
    {
        /* aoclsparse_csr_csc_fill_diag (csr_util.hpp:167-279): insert explicit zeros for
         * missing diagonals of rows i < n, keeping each row sorted. */
        oint w = 0;
        for(oint i = 0; i < m; i++)
        {
            oint s = tptr[i], e = tptr[i + 1];
            optr[i] = w;
            int placed = (i >= n); /* rows beyond n have no diagonal */
            for(oint idx = s; idx < e; idx++)
            {
                oint j = tind[idx];
                if(!placed && j >= i)
                {
                    if(j != i)
                    {
                        oind[w] = i;
                        oval[w] = 0.0;
                        w++;
                    }
                    placed = 1;
                }
                oind[w] = j;
                oval[w] = tval[idx];
                w++;
            }
            if(!placed)
            {
                oind[w] = i;
                oval[w] = 0.0;
                w++;
            }
        }
        optr[m] = w;
        newnnz  = w;
    }
    else
    {
        memcpy(optr, tptr, sizeof(oint) * ((size_t)m + 1));
        memcpy(oind, tind, sizeof(oint) * (size_t)nnz);
        memcpy(oval, tval, sizeof(double) * (size_t)nnz);
    }
    free(tptr);
    free(tind);
    free(tval);
    *onnz         = newnnz;
    *fulldiag_out = fulldiag;
    return orc_csr_indices(m, 0, optr, oind, idiag, iurow);
}

/* convert.hpp:552-655: counting-sort transpose, stable in row order. */
int orc_dcsr2csc(oint m, oint n, oint nnz, int base_csr, int base_csc, const oint *row_ptr,
                 const oint *col_ind, const double *val, oint *csc_row_ind, oint *csc_col_ptr,
                 double *csc_val)
{
    if(m < 0 || n < 0 || nnz < 0)
        return ORC_INVALID_SIZE;
    if(m == 0 || n == 0 || nnz == 0)
    {
        for(oint i = 0; i < n + 1; i++)
            csc_col_ptr[i] = base_csc;
        return ORC_SUCCESS;
    }
    if((base_csr != 0 && base_csr != 1) || (base_csc != 0 && base_csc != 1))
        return ORC_INVALID_VALUE;
    if(!val || !row_ptr || !col_ind || !csc_val || !csc_row_ind || !csc_col_ptr)
        return ORC_INVALID_POINTER;
    for(oint i = 0; i < n + 1; i++)
        csc_col_ptr[i] = 0;
    for(oint i = 0; i < nnz; i++)
        ++csc_col_ptr[col_ind[i] - base_csr + 1];
    for(oint i = 0; i < n; i++)
        csc_col_ptr[i + 1] += csc_col_ptr[i];
    for(oint i = 0; i < m; i++)
        for(oint j = row_ptr[i] - base_csr; j < row_ptr[i + 1] - base_csr; j++)
        {
            oint c           = col_ind[j] - base_csr;
            oint idx         = csc_col_ptr[c];
            csc_row_ind[idx] = i + base_csc;
            csc_val[idx]     = val[j];
            ++csc_col_ptr[c];
        }
    for(oint i = n; i > 0; i--)
        csc_col_ptr[i] = csc_col_ptr[i - 1] + base_csc;
    csc_col_ptr[0] = base_csc;
    return ORC_SUCCESS;
}

/* ilu0.hpp:35-107: IKJ ILU(0) in place; lu_diag_ptr[i] = 0-based position of the diagonal.
 * Restated with the reference's mapper convention (a stored position of 0 means "absent",
 * :83-86), so the entry at array position 0 is never updated -- kept for fidelity. */
int orc_dilu0(oint n, int base, oint *lu_diag_ptr, double *val, const oint *row_ptr,
              const oint *col_ind)
{
    oint *mapper = (oint *)calloc((size_t)(n > 0 ? n : 1), sizeof(oint));
    if(!mapper)
        return ORC_MEMORY_ERROR;
    for(oint i = 0; i < n; i++)
    {
        oint j1 = row_ptr[i] - base, j2 = row_ptr[i + 1] - base, j, k = -1;
        for(j = j1; j < j2; j++)
            mapper[col_ind[j] - base] = j;
        for(j = j1; j < j2; j++)
        {
            k = col_ind[j] - base;
            if(k >= i)
                break;
            double d = val[lu_diag_ptr[k]];
            if(fabs(d) <= 1e-2 * 2.0 * 2.220446049250313e-16) /* aoclsparse_is_nearzero, extra/aoclsparse_utils.hpp:598-613 */
            {
                free(mapper);
                return ORC_NUMERICAL_ERROR;
            }
            val[j] = val[j] / d;
            for(oint jj = lu_diag_ptr[k] + 1; jj < row_ptr[k + 1] - base; jj++)
            {
                oint jw = mapper[col_ind[jj] - base];
                if(jw != 0)
                    val[jw] = fma(-val[j], val[jj], val[jw]);
            }
        }
        lu_diag_ptr[i] = j;
        if(j >= j2 || k != i || fabs(val[j]) <= 1e-2 * 2.0 * 2.220446049250313e-16)
        {
            free(mapper);
            return ORC_NUMERICAL_ERROR;
        }
        for(oint mn = j1; mn < j2; mn++)
            mapper[col_ind[mn] - base] = 0;
    }
    free(mapper);
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Symmetric Gauss-Seidel sweep, solvers/aoclsparse_symgs.hpp:62-258 (symgs_ref), built   */
/* from the triangular SpMV and TRSV restatements above exactly as the reference chains   */
/* aoclsparse::mv / aoclsparse::trsv on the clean CSR.  type: 0 general, 1 symmetric,     */
/* 3 triangular; fill 0 lower / 1 upper; trans 0 none / 1 transpose.                      */
/* ------------------------------------------------------------------------------------ */
static int symgs_mv(int tr, int base, double alpha, oint m, int diag, int fill, const double *val,
                    const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
                    const double *x, double *y)
{
    /* beta = 0: the triangular kernels zero y first (csrmv_kr.hpp:535-542) */
    return tr ? orc_dcsrmv_tri_t(base, alpha, m, m, diag, fill, val, col, ptr, idiag, iurow, x, 0.0, y)
              : orc_dcsrmv_tri(base, alpha, m, diag, fill, val, col, ptr, idiag, iurow, x, 0.0, y);
}
static int symgs_sv(int tr, int fill, int base, oint m, const double *val, const oint *col,
                    const oint *ptr, const oint *idiag, const oint *iurow, const double *b, double *x)
{
    if(fill == 0)
        return tr ? orc_dtrsv_lt(1.0, m, base, val, col, ptr, idiag, b, 1, x, 1, 0)
                  : orc_dtrsv_l(1.0, m, base, val, col, ptr, idiag, b, 1, x, 1, 0);
    return tr ? orc_dtrsv_ut(1.0, m, base, val, col, ptr, iurow, b, 1, x, 1, 0)
              : orc_dtrsv_u(1.0, m, base, val, col, ptr, iurow, b, 1, x, 1, 0);
}
int orc_dsymgs(int type, int fill, int trans, int base, double alpha, oint m, const double *val,
               const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
               const double *b, double *x, double *y, int fuse_mv)
{
    if(type == 3) /* :128-149 */
        return symgs_sv(trans, fill, base, m, val, col, ptr, idiag, iurow, b, x);
    int u_tr = 1, l_tr = 0, u_fill = 0, l_fill = 0; /* symmetric, lower stored (:151-163) */
    if(type == 1 && fill == 1)
        u_fill = l_fill = 1, u_tr = 0, l_tr = 1;
    else if(type == 0 && trans == 0)
        u_tr = l_tr = 0, u_fill = 1;
    else if(type == 0 && trans == 1)
        u_tr = l_tr = 1, l_fill = 1, u_fill = 0;
    double *r = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    double *q = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    if(!r || !q)
    {
        free(r), free(q);
        return ORC_MEMORY_ERROR;
    }
    symgs_mv(u_tr, base, alpha, m, 2, u_fill, val, col, ptr, idiag, iurow, x, q); /* q = alpha U x0 */
    for(oint i = 0; i < m; i++)
        r[i] = b[i] - q[i];
    symgs_sv(l_tr, l_fill, base, m, val, col, ptr, idiag, iurow, r, q); /* (L+D) x1 = r */
    symgs_mv(l_tr, base, 1.0, m, 2, l_fill, val, col, ptr, idiag, iurow, q, r); /* r = L x1 */
    for(oint i = 0; i < m; i++)
        q[i] = b[i] - r[i];
    symgs_sv(u_tr, u_fill, base, m, val, col, ptr, idiag, iurow, q, x); /* (U+D) x = q */
    free(r), free(q);
    (void)y, (void)fuse_mv; /* the closing product is checked by the callers with orc_dcsrmv* */
    return ORC_SUCCESS;
}

/* ILU(0) solve, solvers/aoclsparse_ilu0.hpp:113-156: L y = b (unit lower), U x = y; "sum - val*x"  */
/* contracts to an FMA under the reference's -ffp-contract=fast.                                    */
int orc_dilu_solve(oint n, int base, const oint *lu_diag_ptr, const double *val, const oint *row_ptr,
                   const oint *col_ind, double *x, const double *b)
{
    for(oint i = 0; i < n; i++)
    {
        double sum = b[i];
        for(oint k = row_ptr[i] - base; k < lu_diag_ptr[i]; k++)
            sum = fma(-val[k], x[col_ind[k] - base], sum);
        x[i] = sum;
    }
    for(oint i = n - 1; i >= 0; i--)
    {
        for(oint k = lu_diag_ptr[i] + 1; k < row_ptr[i + 1] - base; k++)
            x[i] = fma(-val[k], x[col_ind[k] - base], x[i]);
        double d = val[lu_diag_ptr[i]];
        if(!(fabs(d) <= 1e-2 * 2.0 * 2.220446049250313e-16))
            x[i] = x[i] / d;
    }
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* ELL family, level2/aoclsparse_ellmv.hpp.  Padding of row-major ELL = column -1.        */
/* ------------------------------------------------------------------------------------ */
static double ell_finish_d(double r, double alpha, double beta, double y)
{
    if(alpha != 1.0)
        r = alpha * r;
    if(beta != 0.0)
        r = fma(beta, y, r); /* "result += beta * y[i]" under -ffp-contract=fast */
    return r;
}
/* aoclsparse_dellmv_avx2, ellmv.hpp:90-208: 4 lanes over full groups (a group whose 4th column is padding
 * goes to the scalar tail), hadd reduction (l0+l1)+(l2+l3), scalar tail that stops at padding. */
int orc_dellmv(int base, double alpha, oint m, const double *val, const oint *col, oint width,
               const double *x, double beta, double *y)
{
    for(oint i = 0; i < m; i++)
    {
        const double *v = val + (size_t)i * width;
        const oint   *c = col + (size_t)i * width;
        oint          k_iter = width / 4, k_rem = width % 4, p = 0;
        double        l[4] = {0, 0, 0, 0}, r = 0.0;
        for(oint it = 0; it < k_iter; it++)
        {
            if(c[p + 3] - base < 0)
            {
                k_rem = 4;
                break;
            }
            for(int q = 0; q < 4; q++)
                l[q] = fma(v[p + q], x[c[p + q] - base], l[q]);
            p += 4;
        }
        if(k_iter)
            r = (l[0] + l[1]) + (l[2] + l[3]);
        for(oint q = 0; q < k_rem; q++)
        {
            oint cc = c[p + q] - base;
            if(cc < 0)
                break;
            r = fma(v[p + q], x[cc], r);
        }
        y[i] = ell_finish_d(r, alpha, beta, y[i]);
    }
    return ORC_SUCCESS;
}
/* aoclsparse_ellmv_ref<float>, ellmv.hpp:34-85 (what aoclsparse_sellmv runs) */
int orc_sellmv(int base, float alpha, oint m, const float *val, const oint *col, oint width,
               const float *x, float beta, float *y)
{
    for(oint i = 0; i < m; i++)
    {
        float r = 0.0f;
        for(oint p = 0; p < width; p++)
        {
            oint cc = col[(size_t)i * width + p] - base;
            if(cc < 0)
                break;
            r = fmaf(val[(size_t)i * width + p], x[cc], r);
        }
        if(alpha != 1.0f)
            r = alpha * r;
        if(beta != 0.0f)
            r = fmaf(beta, y[i], r);
        y[i] = r;
    }
    return ORC_SUCCESS;
}
/* aoclsparse_elltmv_avx2 / _ref, ellmv.hpp:316-444: one FMA chain per row over the column-major cells */
int orc_delltmv(int base, double alpha, oint m, const double *val, const oint *col, oint width,
                const double *x, double beta, double *y)
{
    for(oint j = 0; j < m; j++)
    {
        double r = 0.0;
        for(oint i = 0; i < width; i++)
            r = fma(val[(size_t)i * m + j], x[col[(size_t)i * m + j] - base], r);
        y[j] = ell_finish_d(r, alpha, beta, y[j]);
    }
    return ORC_SUCCESS;
}
/* aoclsparse_ellthybmv_avx2, ellmv.hpp:554-757: ELLT pass over all rows, then the listed rows again
 * from the CSR arrays in the 4-lane order, with the beta term taken from the ORIGINAL y. */
int orc_dellthybmv(int base, double alpha, oint m, const double *ell_val, const oint *ell_col,
                   oint width, oint ell_m, const double *csr_val, const oint *csr_row,
                   const oint *csr_col, const oint *map, const double *x, double beta, double *y)
{
    oint    nlong = m - ell_m;
    double *ytmp  = (double *)malloc(sizeof(double) * (size_t)(nlong > 0 ? nlong : 1));
    if(!ytmp)
        return ORC_MEMORY_ERROR;
    for(oint i = 0; i < nlong; i++)
        ytmp[i] = y[map[i]];
    orc_delltmv(base, alpha, m, ell_val, ell_col, width, x, beta, y);
    for(oint i = 0; i < nlong; i++)
    {
        oint   row = map[i], s = csr_row[row] - base, e = csr_row[row + 1] - base;
        oint   full = (e - s) / 4 * 4;
        double l[4] = {0, 0, 0, 0}, r = 0.0;
        for(oint p = s; p < s + full; p++)
            l[(p - s) & 3] = fma(csr_val[p], x[csr_col[p] - base], l[(p - s) & 3]);
        if(full)
            r = (l[0] + l[1]) + (l[2] + l[3]);
        for(oint p = s + full; p < e; p++)
            r = fma(csr_val[p], x[csr_col[p] - base], r);
        y[row] = ell_finish_d(r, alpha, beta, ytmp[i]);
    }
    free(ytmp);
    return ORC_SUCCESS;
}
/* aoclsparse_csr_lsolve / _usolve, csrsv.hpp:88-187 (zero-based only, :47-51).  Lower: the row is walked left to
 * right and CUT at the first entry on or right of the diagonal; upper: every entry right of the diagonal is applied.
 * y -= a*y is contracted to an fma like every other chain; the diagonal used is the last one seen (diag_j persists
 * across rows, as in the reference). */
int orc_dcsrsv(int lower, int unit, double alpha, oint m, const double *val, const oint *col, const oint *row_ptr,
               const double *x, double *y)
{
    oint diag_j = 0;
    if(lower)
        for(oint r = 0; r < m; r++)
        {
            double yr = alpha * x[r];
            for(oint j = row_ptr[r]; j < row_ptr[r + 1]; j++)
            {
                if(col[j] < r)
                    yr = fma(-val[j], y[col[j]], yr);
                else
                {
                    if(!unit && col[j] == r)
                        diag_j = j;
                    break;
                }
            }
            y[r] = unit ? yr : yr / val[diag_j];
        }
    else
        for(oint r = m - 1; r >= 0; r--)
        {
            double yr = alpha * x[r];
            for(oint j = row_ptr[r]; j < row_ptr[r + 1]; j++)
            {
                if(col[j] > r)
                    yr = fma(-val[j], y[col[j]], yr);
                if(!unit && col[j] == r)
                    diag_j = j;
            }
            y[r] = unit ? yr : yr / val[diag_j];
        }
    return ORC_SUCCESS;
}

/* ---- BLKCSR ------------------------------------------------------------------------------------------
 * walk of one row block, shared by the block count (convert.cpp:71-107) and the conversion (:214-283):
 * every pass opens a window of 8 columns at the smallest unread column of the block's sub-rows and consumes
 * every entry of every sub-row that falls inside it. */
static oint blk_min_col(oint rows, oint i0, oint m, int base, const oint *row_ptr, const oint *col, const oint *pos)
{
    oint c = 0x7fffffff;
    for(oint r = 0; r < rows && i0 + r < m; r++)
        if(pos[r] < row_ptr[i0 + r + 1] - base && col[pos[r]] - base < c)
            c = col[pos[r]] - base;
    return c;
}
/* aoclsparse_opt_blksize, convert.cpp:36-147 */
oint orc_opt_blksize(oint m, oint nnz, int base, const oint *row_ptr, const oint *col_ind, oint *total_blks)
{
    if(m <= 0 || nnz <= 0 || !row_ptr || !col_ind || !total_blks)
        return 0;
    static const oint factor[3] = {1, 2, 4};
    oint   total[3];
    double per[3], util[3], inc[2] = {0, 0};
    double nnzpr = (double)nnz / m;
    for(int f = 0; f < 3; f++)
    {
        oint rows = factor[f], blocks = 0;
        for(oint i0 = 0; i0 < m; i0 += rows)
        {
            oint pos[4] = {0, 0, 0, 0};
            for(oint r = 0; r < rows && i0 + r < m; r++)
                pos[r] = row_ptr[i0 + r] - base;
            for(;;)
            {
                oint c = blk_min_col(rows, i0, m, base, row_ptr, col_ind, pos);
                if(c == 0x7fffffff)
                    break;
                for(oint r = 0; r < rows && i0 + r < m; r++)
                    while(pos[r] < row_ptr[i0 + r + 1] - base && col_ind[pos[r]] - base < c + 8)
                        pos[r]++;
                blocks++;
            }
        }
        total[f] = blocks;
        if(blocks == 0)
            return 0;
        per[f]  = (double)nnz / (double)blocks;
        util[f] = per[f] / ((double)rows * 8) * 100;
        if((nnzpr < 30 && util[0] < 40) || (nnzpr > 30 && util[0] < 50))
            return 0;
        if(f)
            inc[f - 1] = (per[f] - per[f - 1]) / per[f - 1] * 100;
    }
    /* the reference calls the INTEGER abs() on both differences (convert.cpp:128-129): they are truncated */
    double d_blks = (double)abs((int)(inc[0] - inc[1]));
    double d_util = (double)abs((int)(util[1] - util[2]));
    if(util[2] > 24 && (d_blks < 12.5 || d_util < 12.5) && inc[1] > 51)
    {
        *total_blks = total[2];
        return 4;
    }
    if(util[1] > 28)
    {
        *total_blks = total[1];
        return 2;
    }
    return 0;
}
/* aoclsparse_csr2blkcsr, convert.cpp:149-310.  Values are appended sub-row by sub-row inside a block; a
 * window that would run past column n is re-anchored at n-8 and its masks shifted left accordingly. */
int orc_dcsr2blkcsr(oint m, oint n, oint nnz, const oint *row_ptr, const oint *col_ind, const double *val,
                    oint *blk_row_ptr, oint *blk_col, double *blk_val, unsigned char *masks, oint rows,
                    int base, oint *nblk)
{
    if(m < 0 || n < 8 || nnz < 0)
        return ORC_INVALID_SIZE;
    if(!row_ptr || !col_ind || !val || !blk_row_ptr || !blk_col || !blk_val || !masks)
        return ORC_INVALID_POINTER;
    if(rows != 1 && rows != 2 && rows != 4)
        return ORC_INVALID_SIZE;
    oint blocks = 0, nv = 0;
    for(oint i0 = 0; i0 < m; i0 += rows)
    {
        oint pos[4] = {0, 0, 0, 0}, first = blocks;
        for(oint r = 0; r < rows && i0 + r < m; r++)
            pos[r] = row_ptr[i0 + r] - base;
        for(;;)
        {
            oint c = blk_min_col(rows, i0, m, base, row_ptr, col_ind, pos);
            if(c == 0x7fffffff)
                break;
            unsigned char mk[4] = {0, 0, 0, 0}, shifted[4] = {0, 0, 0, 0};
            for(oint r = 0; r < rows && i0 + r < m; r++)
                while(pos[r] < row_ptr[i0 + r + 1] - base && col_ind[pos[r]] - base < c + 8)
                {
                    blk_val[nv++] = val[pos[r]];
                    mk[r] |= (unsigned char)(1u << (col_ind[pos[r]] - base - c));
                    if(c + 8 > n)
                        shifted[r] = (unsigned char)(mk[r] << (8 - (n - c)));
                    pos[r]++;
                }
            const int past = c + 8 > n;
            blk_col[blocks] = (past ? n - 8 : c) + base;
            for(oint r = 0; r < rows; r++)
                masks[(size_t)blocks * rows + r] = past ? shifted[r] : mk[r];
            blocks++;
        }
        blk_row_ptr[i0] = first + base;
        for(oint r = 1; r < rows && i0 + r < m; r++)
            blk_row_ptr[i0 + r] = blocks + base;
    }
    blk_row_ptr[m] = blocks + base;
    if(nblk)
        *nblk = blocks;
    return ORC_SUCCESS;
}
/* aoclsparse_blkcsrmv_{1,2,4}x8_vectorized_avx512, blkcsrmv_avx512.cpp:40-369: per sub-row eight lane
 * accumulators (lane = column offset inside the window), one fmadd per block with zeros where the mask has no
 * bit (so x is multiplied by 0 there, as the expand-load does), reduction lo4+hi4 -> hadd -> add, then
 * sum = 0 + that, alpha if != 1, fma(beta, y, sum) if beta != 0.  Values are consumed in storage order. */
int orc_dblkcsrmv(int base, double alpha, oint m, const unsigned char *masks, const double *blk_val,
                  const oint *blk_col, const oint *blk_row_ptr, const double *x, double beta, double *y, oint rows)
{
    if(rows != 1 && rows != 2 && rows != 4)
        return ORC_INVALID_SIZE;
    size_t iv = 0;
    for(oint i0 = 0; i0 < m; i0 += rows)
    {
        double acc[4][8];
        for(int r = 0; r < 4; r++)
            for(int l = 0; l < 8; l++)
                acc[r][l] = 0.0;
        for(oint b = blk_row_ptr[i0] - base; b < blk_row_ptr[i0 + 1] - base; b++)
        {
            const double *xw = x + (blk_col[b] - base);
            for(oint r = 0; r < rows; r++)
            {
                unsigned mk = masks[(size_t)b * rows + r];
                for(int l = 0; l < 8; l++)
                {
                    double v  = (mk >> l) & 1 ? blk_val[iv++] : 0.0;
                    acc[r][l] = fma(v, xw[l], acc[r][l]);
                }
            }
        }
        for(oint r = 0; r < rows && i0 + r < m; r++)
        {
            double v0 = acc[r][0] + acc[r][4], v1 = acc[r][1] + acc[r][5];
            double v2 = acc[r][2] + acc[r][6], v3 = acc[r][3] + acc[r][7];
            double sum = 0.0;
            sum += (v0 + v1) + (v2 + v3);
            if(alpha != 1.0)
                sum = alpha * sum;
            if(beta != 0.0)
                sum = fma(beta, y[i0 + r], sum);
            y[i0 + r] = sum;
        }
    }
    return ORC_SUCCESS;
}
/* conversion/aoclsparse_convert.hpp:41-107 (ELL), :110-175 (ELLT), :178-289 (ELLT-HYB) and
 * conversion/aoclsparse_convert.cpp:311-412 (widths).  layout: 0 ELL, 1 ELLT. */
int orc_csr2ell_width(oint m, const oint *row_ptr, oint *width)
{
    *width = 0;
    for(oint i = 0; i < m; i++)
        if(row_ptr[i + 1] - row_ptr[i] > *width)
            *width = row_ptr[i + 1] - row_ptr[i];
    return ORC_SUCCESS;
}
int orc_csr2ellthyb_width(oint m, oint nnz, const oint *row_ptr, oint *ell_m, oint *width)
{
    *width = 0, *ell_m = 0;
    if(m == 0)
        return ORC_SUCCESS;
    oint mx = 0, mn = nnz, cmn = 0, cmx = 0, avg = nnz / m;
    for(oint i = 0; i < m; i++)
    {
        oint len = row_ptr[i + 1] - row_ptr[i];
        if(len > mx && len <= avg)
            mx = len;
        if(len < mn && len > avg)
            mn = len;
        if(len <= avg)
            cmx++;
        else
            cmn++;
    }
    *width = cmx >= cmn ? mx : mn;
    for(oint i = 0; i < m; i++)
        if(row_ptr[i + 1] - row_ptr[i] <= *width)
            (*ell_m)++;
    return ORC_SUCCESS;
}
int orc_dcsr2ell(int layout, oint m, int base, const oint *row_ptr, const oint *col_ind,
                 const double *val, oint *ell_col, double *ell_val, oint width)
{
    for(oint i = 0; i < m; i++)
    {
        oint s = row_ptr[i] - base, e = row_ptr[i + 1] - base, k = 0;
        for(oint j = s; j < e; j++, k++)
        {
            size_t o   = layout ? (size_t)k * m + i : (size_t)i * width + k;
            ell_col[o] = col_ind[j], ell_val[o] = val[j];
        }
        for(; k < width; k++)
        {
            size_t o   = layout ? (size_t)k * m + i : (size_t)i * width + k;
            ell_col[o] = layout ? (e > s ? col_ind[e - 1] : base) : -1; /* empty row: see ell_api.cpp */
            ell_val[o] = 0.0;
        }
    }
    return ORC_SUCCESS;
}
int orc_dcsr2ellthyb(oint m, int base, oint *ell_m, const oint *row_ptr, const oint *col_ind,
                     const double *val, oint *map, oint *ell_col, double *ell_val, oint width)
{
    oint nlong = 0;
    *ell_m     = 0;
    for(oint i = 0; i < m; i++)
    {
        oint s = row_ptr[i] - base, e = row_ptr[i + 1] - base, k = 0;
        oint pad = e > s ? col_ind[e - 1] : base;
        if(e - s > width)
            map[nlong++] = i;
        else
        {
            (*ell_m)++;
            for(oint j = s; j < e; j++, k++)
                ell_col[(size_t)k * m + i] = col_ind[j], ell_val[(size_t)k * m + i] = val[j];
        }
        for(; k < width; k++)
            ell_col[(size_t)k * m + i] = pad, ell_val[(size_t)k * m + i] = 0.0;
    }
    return ORC_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Iterative solvers, solvers/aoclsparse_itsol_functions.hpp: CG :632-875 (+ the built-in */
/* SymGS preconditioner :390-479), restarted GMRES :910-1367 (+ ILU(0) preconditioner).   */
/* The reference's level-1 steps are AOCL-BLAS calls (not vendored): plain loops here.    */
/* A is a general clean CSR holding the whole (for CG: symmetric) matrix.                 */
/* precond: CG 0 none / 3 SymGS; GMRES 0 none / 2 ILU0.  Returns the reference's status;   */
/* rinfo[0] residual norm, rinfo[1] ||b|| (GMRES: rtol*||b||), rinfo[30] iterations.       */
/* ------------------------------------------------------------------------------------ */
static double orc_nrm2(oint n, const double *v)
{
    double s = 0.0;
    for(oint i = 0; i < n; i++)
        s += v[i] * v[i];
    return sqrt(s);
}
static void orc_mv(oint n, int base, const oint *ptr, const oint *col, const double *val,
                   const double *x, double *y)
{
    orc_dcsrmv_ref(base, 1.0, n, val, col, ptr, x, 0.0, y);
}
int orc_dcg(oint n, int base, const oint *ptr, const oint *col, const double *val,
            const oint *idiag, const oint *iurow, const double *b, double *x, double rtol,
            double atol, oint maxit, int precond, double *rinfo)
{
    const double tiny = 1e-2 * 2.0 * 2.220446049250313e-16;
    double      *w    = (double *)calloc(5 * (size_t)(n > 0 ? n : 1), sizeof(double));
    if(!w)
        return ORC_MEMORY_ERROR;
    double *r = w, *z = w + n, *p = w + 2 * (size_t)n, *q = w + 3 * (size_t)n, *y = w + 4 * (size_t)n;
    int     status = ORC_SUCCESS;
    for(int i = 0; i < 100; i++)
        rinfo[i] = 0.0;
    for(oint i = 0; i < n; i++)
        r[i] = -b[i], p[i] = x[i];
    double bnorm = orc_nrm2(n, b), brtol = rtol * bnorm;
    rinfo[1]     = bnorm;
    orc_mv(n, base, ptr, col, val, p, q);
    for(oint i = 0; i < n; i++)
        r[i] += q[i], p[i] = 0.0;
    double rnorm = orc_nrm2(n, r), rz = 1.0;
    rinfo[0]     = rnorm;
    oint niter   = 0;
    for(;;)
    {
        if((0.0 < atol && rnorm <= atol) || (0.0 < rtol && rnorm <= brtol))
            break;
        if(maxit > 0 && niter > maxit)
        {
            status = 7; /* aoclsparse_status_maxit */
            break;
        }
        niter++;
        rinfo[30] = (double)niter;
        if(precond == 3)
        {
            /* (L+D) y = r ; y = D y ; (U+D) z = y */
            orc_dtrsv_l(1.0, n, base, val, col, ptr, idiag, r, 1, y, 1, 0);
            for(oint i = 0; i < n; i++)
                y[i] *= val[idiag[i] - base];
            orc_dtrsv_u(1.0, n, base, val, col, ptr, iurow, y, 1, z, 1, 0);
        }
        else
            for(oint i = 0; i < n; i++)
                z[i] = r[i];
        double rz_new = 0.0;
        for(oint i = 0; i < n; i++)
            rz_new += r[i] * z[i];
        if(rz <= tiny)
        {
            status = ORC_NUMERICAL_ERROR;
            break;
        }
        double beta = rz_new / rz;
        rz          = rz_new;
        for(oint i = 0; i < n; i++)
            p[i] = beta * p[i] - z[i];
        orc_mv(n, base, ptr, col, val, p, q);
        double pq = 0.0;
        for(oint i = 0; i < n; i++)
            pq += p[i] * q[i];
        if(pq <= tiny)
        {
            status = ORC_NUMERICAL_ERROR;
            break;
        }
        double alpha = rz / pq;
        for(oint i = 0; i < n; i++)
            x[i] += alpha * p[i], r[i] += alpha * q[i];
        rnorm    = orc_nrm2(n, r);
        rinfo[0] = rnorm;
    }
    free(w);
    return status;
}

/* LAPACK 3.10 dlartg, unscaled branch (the values met here are far from the over/underflow limits) */
static void orc_lartg(double f, double g, double *c, double *s, double *r)
{
    if(g == 0.0)
        *c = 1.0, *s = 0.0, *r = f;
    else if(f == 0.0)
        *c = 0.0, *s = copysign(1.0, g), *r = fabs(g);
    else
    {
        double d = sqrt(f * f + g * g);
        *c = fabs(f) / d, *r = copysign(d, f), *s = g / *r;
    }
}
int orc_dgmres(oint n, int base, const oint *ptr, const oint *col, const double *val,
               const double *b, double *x, oint m, double rtol, double atol, oint maxit,
               int precond, double *rinfo)
{
    const double tiny = 1e-2 * 2.0 * 2.220446049250313e-16;
    size_t       nn = (size_t)(n > 0 ? n : 1), mm = (size_t)m;
    double      *V = (double *)calloc((mm + 1) * nn, sizeof(double)), *Z = (double *)calloc((mm + 1) * nn, sizeof(double));
    double      *h = (double *)calloc(mm * mm, sizeof(double)), *g = (double *)calloc(mm + 1, sizeof(double));
    double      *c = (double *)calloc(mm, sizeof(double)), *s = (double *)calloc(mm, sizeof(double));
    double      *lu = NULL;
    oint        *ludiag = NULL;
    int          status = ORC_SUCCESS;
    if(!V || !Z || !h || !g || !c || !s)
    {
        status = ORC_MEMORY_ERROR;
        goto done;
    }
    if(precond == 2)
    {
        oint nnz = ptr[n] - base;
        lu       = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
        ludiag   = (oint *)malloc(sizeof(oint) * nn);
        if(!lu || !ludiag)
        {
            status = ORC_MEMORY_ERROR;
            goto done;
        }
        memcpy(lu, val, sizeof(double) * (size_t)nnz);
        status = orc_dilu0(n, base, ludiag, lu, ptr, col);
        if(status != ORC_SUCCESS)
            goto done;
    }
    oint niter = 0;
    for(;;) /* one restart cycle per pass */
    {
        orc_mv(n, base, ptr, col, val, x, V);
        double bnorm = orc_nrm2(n, b), brtol = rtol * bnorm;
        rinfo[1]     = brtol;
        if(fabs(atol) <= tiny && fabs(brtol) <= tiny)
        {
            status = 5; /* invalid_value */
            goto done;
        }
        for(oint i = 0; i < n; i++)
            V[i] = b[i] - V[i];
        double rnorm = orc_nrm2(n, V);
        g[0] = rnorm, rinfo[0] = rnorm;
        if((0.0 < rnorm && (rnorm <= atol || rnorm <= brtol)) || rnorm == 0.0)
        {
            rinfo[30] = (double)niter;
            goto done;
        }
        for(oint i = 0; i < n; i++)
            V[i] *= 1.0 / rnorm;
        oint j = 0;
        for(; j < m; j++)
        {
            double *vj = V + (size_t)j * nn, *w = V + (size_t)(j + 1) * nn, *zj = Z + (size_t)j * nn;
            if(precond == 2)
                orc_dilu_solve(n, base, ludiag, lu, ptr, col, zj, vj);
            orc_mv(n, base, ptr, col, val, precond ? zj : vj, w);
            for(oint i = 0; i <= j; i++)
            {
                double d = 0.0;
                for(oint k = 0; k < n; k++)
                    d += w[k] * V[(size_t)i * nn + k];
                h[(size_t)i * mm + j] = d;
            }
            for(oint k = 0; k < n; k++)
            {
                double hv = 0.0;
                for(oint i = 0; i <= j; i++)
                    hv += h[(size_t)i * mm + j] * V[(size_t)i * nn + k];
                w[k] -= hv;
            }
            double hh = orc_nrm2(n, w);
            if(hh < atol || hh < brtol)
            {
                niter += j + 1;
                rinfo[30] = (double)niter, rinfo[0] = hh;
                goto done;
            }
            for(oint k = 0; k < n; k++)
                w[k] *= 1.0 / hh;
            for(oint i = 0; i < j; i++)
            {
                double r1 = h[(size_t)i * mm + j], r2 = h[(size_t)(i + 1) * mm + j];
                h[(size_t)i * mm + j]       = c[i] * r1 - s[i] * r2;
                h[(size_t)(i + 1) * mm + j] = s[i] * r1 + c[i] * r2;
            }
            double rr = h[(size_t)j * mm + j];
            orc_lartg(rr, -hh, &c[j], &s[j], &h[(size_t)j * mm + j]);
            double g0 = g[j];
            g[j] = c[j] * g0, g[j + 1] = s[j] * g0;
            rinfo[0] = fabs(g[j]);
        }
        for(oint jj = m - 1; jj >= 0; jj--)
        {
            double yj = g[jj];
            for(oint i = jj + 1; i < m; i++)
                yj -= h[(size_t)jj * mm + i] * s[i];
            if(fabs(h[(size_t)jj * mm + jj]) <= tiny)
            {
                status = ORC_NUMERICAL_ERROR;
                goto done;
            }
            s[jj] = yj / h[(size_t)jj * mm + jj];
        }
        for(oint k = 0; k < n; k++)
        {
            double acc = 0.0;
            for(oint t = 0; t < m; t++)
                acc += (precond ? Z : V)[(size_t)t * nn + k] * s[t];
            x[k] += acc;
        }
        rnorm = fabs(g[m]);
        niter += m;
        rinfo[30] = (double)niter, rinfo[0] = rnorm;
        if((0.0 < atol && rnorm <= atol) || (0.0 < rnorm && rnorm <= brtol))
            goto done;
        if(maxit > 0 && niter >= maxit)
        {
            status = 7;
            goto done;
        }
    }
done:
    free(V), free(Z), free(h), free(g), free(c), free(s), free(lu), free(ludiag);
    return status;
}

/* ------------------------------------------------------------------------------------ */
/* sp2m = two-stage Gustavson, csr2m.cpp:46-302 (count) and :310-543 (finalize).         */
/* ------------------------------------------------------------------------------------ */
int orc_csr2m_nnz(oint m, oint n, int base_a, const oint *ptr_a, const oint *ind_a,
                  int base_b, const oint *ptr_b, const oint *ind_b, oint *ptr_c, oint *nnz_c)
{
    if(!ptr_a || !ind_a || !ptr_b || !ind_b)
        return ORC_INVALID_POINTER;
    oint *mark = (oint *)malloc(sizeof(oint) * ((size_t)n + 1));
    if(!mark)
        return ORC_MEMORY_ERROR;
    for(oint i = 0; i < n; i++)
        mark[i] = -1;
    long long total = 0;
    ptr_c[0]        = 0;
    for(oint i = 0; i < m; i++)
    {
        oint cnt = 0;
        for(oint j = ptr_a[i] - base_a; j < ptr_a[i + 1] - base_a; j++)
        {
            oint ca = ind_a[j] - base_a;
            for(oint k = ptr_b[ca] - base_b; k < ptr_b[ca + 1] - base_b; k++)
            {
                oint cb = ind_b[k] - base_b;
                if(mark[cb] != i)
                {
                    mark[cb] = i;
                    cnt++;
                }
            }
        }
        total += cnt;
        ptr_c[i + 1] = (oint)total;
    }
    free(mark);
    if(total > 2147483647LL)
        return ORC_INVALID_SIZE;
    *nnz_c = (oint)total;
    return ORC_SUCCESS;
}

int orc_dcsr2m_fill(oint m, oint n, int base_a, const oint *ptr_a, const oint *ind_a,
                    const double *val_a, int base_b, const oint *ptr_b, const oint *ind_b,
                    const double *val_b, const oint *ptr_c, oint *ind_c, double *val_c)
{
    oint   *aidx = (oint *)malloc(sizeof(oint) * ((size_t)n + 1));
    double *asum = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    if(!aidx || !asum)
    {
        free(aidx);
        free(asum);
        return ORC_MEMORY_ERROR;
    }
    for(oint i = 0; i < n; i++)
    {
        aidx[i] = -1;
        asum[i] = 0.0;
    }
    int st = ORC_SUCCESS;
    for(oint i = 0; i < m; i++)
    {
        oint w = ptr_c[i];
        for(oint j = ptr_a[i] - base_a; j < ptr_a[i + 1] - base_a; j++)
        {
            oint   ca = ind_a[j] - base_a;
            double va = val_a[j];
            for(oint k = ptr_b[ca] - base_b; k < ptr_b[ca + 1] - base_b; k++)
            {
                oint cb = ind_b[k] - base_b;
                if(aidx[cb] != i)
                {
                    ind_c[w] = cb; /* first-touch order, csr2m.cpp:489-496 */
                    aidx[cb] = i;
                    asum[cb] = va * val_b[k];
                    w++;
                }
                else
                    asum[cb] = fma(va, val_b[k], asum[cb]); /* :498, contracted */
            }
        }
        if(w != ptr_c[i + 1])
            st = ORC_INTERNAL_ERROR;
        else
            for(w = ptr_c[i]; w < ptr_c[i + 1]; w++)
                val_c[w] = asum[ind_c[w]];
    }
    free(aidx);
    free(asum);
    return st;
}


/* ---- sparse x sparse with a dense result: level3/aoclsparse_sp2md.hpp:40-168 ---------------------------------
 * (A, B) are the operands AFTER op handling (the driver transposes with the stable csr2csc, :286-347); rs / cs are
 * the element strides of a row / column step of C, so both layout kernels (:40-99 column, :101-168 row) are this
 * one loop nest.  C(i,c) += (alpha*a) * b, contracted. */
void orc_dsp2md(oint m, int base_a, const oint *ptr_a, const oint *ind_a, const double *val_a, int base_b,
                const oint *ptr_b, const oint *ind_b, const double *val_b, double alpha, double *C, long long rs,
                long long cs)
{
    for(oint i = 0; i < m; i++)
        for(oint j = ptr_a[i] - base_a; j < ptr_a[i + 1] - base_a; j++)
        {
            const double v  = alpha * val_a[j];
            const oint   ca = ind_a[j] - base_a;
            for(oint k = ptr_b[ca] - base_b; k < ptr_b[ca + 1] - base_b; k++)
            {
                double *p = C + (long long)i * rs + (long long)(ind_b[k] - base_b) * cs;
                *p        = fma(v, val_b[k], *p);
            }
        }
}

/* sp2md.hpp:362-379 / :407-420: beta == 0 stores zeros, beta == 1 leaves C alone, otherwise C *= beta */
void orc_dsp2md_scale(oint outer, oint inner, oint ld, double beta, double *C)
{
    if(beta == 1.0)
        return;
    for(oint o = 0; o < outer; o++)
        for(oint i = 0; i < inner; i++)
            C[(size_t)o * ld + i] = beta == 0.0 ? 0.0 : beta * C[(size_t)o * ld + i];
}

/* ---- CSR -> dense: conversion/aoclsparse_convert.hpp:658-929 ---------------------------------------------------
 * mode 0 general, 1 symmetric, 2 hermitian, 3 triangular; fill 0 lower / 1 upper; diag 0 non-unit, 1 unit, 2 zero.
 * The reference's column-major symmetric branch addresses through row*ld for the diagonal and both mirrors
 * (:770-806); with rs/cs strides that is this same walk. */
void orc_dcsr2dense(oint m, oint n, int base, const oint *ptr, const oint *ind, const double *val, double *A,
                    oint ld, int colmajor, int mode, int fill, int diag)
{
    const long long rs = colmajor ? 1 : ld, cs = colmajor ? ld : 1;
    for(oint r = 0; r < m; r++)
        for(oint c = 0; c < n; c++)
            A[r * rs + c * cs] = 0.0;
    for(oint r = 0; r < m; r++)
    {
        if(mode != 0 && diag == 1)
            A[r * rs + r * cs] = 1.0;
        else if(mode != 0 && diag == 2)
            A[r * rs + r * cs] = 0.0;
        for(oint at = ptr[r] - base; at < ptr[r + 1] - base; at++)
        {
            const oint c = ind[at] - base;
            if(mode == 0)
                A[r * rs + c * cs] = val[at];
            else if(c == r)
            {
                if(diag == 0)
                    A[r * rs + c * cs] = val[at];
            }
            else if((fill == 0 && c < r) || (fill == 1 && c > r))
            {
                A[r * rs + c * cs] = val[at];
                if(mode == 1 || mode == 2)
                    A[c * rs + r * cs] = val[at];
            }
        }
    }
}

/* ---- C = alpha*A + B: level3/aoclsparse_csradd.hpp:137-281, single-thread branch ------------------------------
 * A is the operand after op handling.  Row i of C: every entry of A's row (scaled), then B's entries whose column is
 * new, in B's order; the others are added onto the recorded slot.  C's indices carry A's base.  ptr_c has m+1
 * entries, ind_c / val_c room for nnz_a + nnz_b.  Returns nnz(C). */
oint orc_dcsradd(oint m, oint n, int base_a, const oint *ptr_a, const oint *ind_a, const double *val_a, double alpha,
                 int base_b, const oint *ptr_b, const oint *ind_b, const double *val_b, oint *ptr_c, oint *ind_c,
                 double *val_c)
{
    oint *mark = (oint *)malloc(sizeof(oint) * ((size_t)n + 2));
    oint *rec  = (oint *)malloc(sizeof(oint) * ((size_t)n + 2));
    if(!mark || !rec)
    {
        free(mark);
        free(rec);
        return -1;
    }
    for(oint i = 0; i < n + 2; i++)
        mark[i] = rec[i] = -1;
    oint w   = 0;
    ptr_c[0] = base_a;
    for(oint i = 0; i < m; i++)
    {
        for(oint j = ptr_a[i] - base_a; j < ptr_a[i + 1] - base_a; j++)
        {
            const oint c = ind_a[j]; /* with A's base in place, :245 */
            mark[c]      = i;
            rec[c]       = w;
            ind_c[w]     = c;
            val_c[w++]   = alpha * val_a[j];
        }
        for(oint j = ptr_b[i] - base_b; j < ptr_b[i + 1] - base_b; j++)
        {
            const oint c = ind_b[j] - base_b + base_a;
            if(mark[c] != i)
            {
                ind_c[w]   = c;
                val_c[w++] = val_b[j];
                mark[c]    = i;
            }
            else
                val_c[rec[c]] += val_b[j];
        }
        ptr_c[i + 1] = w + base_a;
    }
    free(mark);
    free(rec);
    return w;
}


/* ---- level 1: level1/aoclsparse_axpyi.hpp:35-50, aoclsparse_dot.hpp:33-61, aoclsparse_roti.hpp:36-55 (reference
 * kernels, kid 0).  Return 6 (invalid_index_value) at the first negative index, entries before it already applied. */
int orc_daxpyi(oint nnz, double a, const double *x, const oint *indx, double *y)
{
    for(oint i = 0; i < nnz; i++)
    {
        if(indx[i] < 0)
            return 6;
        y[indx[i]] = fma(a, x[i], y[indx[i]]);
    }
    return ORC_SUCCESS;
}

double orc_ddoti(oint nnz, const double *x, const oint *indx, const double *y)
{
    double dot = 0.0;
    for(oint i = 0; i < nnz; i++)
        dot = fma(x[i], y[indx[i]], dot);
    return dot;
}

int orc_droti(oint nnz, double *x, const oint *indx, double *y, double c, double s)
{
    for(oint i = 0; i < nnz; i++)
    {
        if(indx[i] < 0)
            return 6;
        const double t = x[i], u = y[indx[i]];
        x[i]           = fma(c, t, s * u);
        y[indx[i]]     = fma(c, u, -(s * t));
    }
    return ORC_SUCCESS;
}


/* ---- DIA: conversion/aoclsparse_convert.cpp:510-566, conversion/aoclsparse_convert.hpp:291-387,
 * level2/aoclsparse_diamv.hpp:34-70 (reference kernel) -------------------------------------------------------- */
oint orc_csr2dia_ndiag(oint m, oint n, int base, const oint *ptr, const oint *ind)
{
    char *seen = (char *)calloc((size_t)m + (size_t)n + 1, 1);
    oint  cnt  = 0;
    if(!seen)
        return -1;
    for(oint i = 0; i < m; i++)
        for(oint j = ptr[i] - base; j < ptr[i + 1] - base; j++)
        {
            const oint off = ind[j] - base - i + m;
            if(!seen[off])
            {
                seen[off] = 1;
                cnt++;
            }
        }
    free(seen);
    return cnt;
}

/* dia_val must come in zero-filled (m * ndiag): the reference only writes the stored entries (:372-383) */
int orc_dcsr2dia(oint m, oint n, int base, const oint *ptr, const oint *ind, const double *val, oint ndiag,
                 oint *dia_offset, double *dia_val)
{
    oint *rank = (oint *)calloc((size_t)m + (size_t)n + 1, sizeof(oint));
    if(!rank)
        return ORC_MEMORY_ERROR;
    for(oint i = 0; i < m; i++)
        for(oint j = ptr[i] - base; j < ptr[i + 1] - base; j++)
            rank[ind[j] - base - i + m] = 1;
    oint d = 0;
    for(oint k = 0; k < m + n; k++)
        if(rank[k])
        {
            rank[k] = d;
            if(d < ndiag)
                dia_offset[d] = k - m;
            d++;
        }
    for(oint i = 0; i < m; i++)
        for(oint j = ptr[i] - base; j < ptr[i + 1] - base; j++)
            dia_val[i + (size_t)m * rank[ind[j] - base - i + m]] = val[j];
    free(rank);
    return ORC_SUCCESS;
}

void orc_ddiamv(double alpha, oint m, oint n, const double *dia_val, const oint *dia_offset, oint ndiag,
                const double *x, double beta, double *y)
{
    if(beta == 0.0)
        for(oint i = 0; i < m; i++)
            y[i] = 0.0;
    else if(beta != 1.0)
        for(oint i = 0; i < m; i++)
            y[i] = beta * y[i];
    for(oint d = 0; d < ndiag; d++)
    {
        const oint off = dia_offset[d];
        const oint i0 = off < 0 ? -off : 0, j0 = off > 0 ? off : 0;
        const oint cnt = (m - i0) < (n - j0) ? (m - i0) : (n - j0);
        for(oint j = 0; j < cnt; j++)
            y[i0 + j] = fma(alpha * dia_val[i0 + (size_t)d * m + j], x[j + j0], y[i0 + j]);
    }
}

/* ---- BSR: conversion/aoclsparse_convert.cpp:596-729, conversion/aoclsparse_convert.hpp:389-551,
 * level2/aoclsparse_bsrmv_kr.hpp:30-94 with aoclsparse_bsrmv_bldr.hpp:47-161 ----------------------------------- */
oint orc_csr2bsr_nnz(oint m, oint n, int base, const oint *ptr, const oint *ind, oint dim, oint *bsr_ptr)
{
    const oint mb = (m + dim - 1) / dim, nb = (n + dim - 1) / dim;
    char      *hit = (char *)calloc((size_t)nb + 1, 1);
    oint      *undo = (oint *)malloc(sizeof(oint) * ((size_t)nb + 1));
    if(!hit || !undo)
    {
        free(hit);
        free(undo);
        return -1;
    }
    bsr_ptr[0] = base;
    for(oint bi = 0; bi < mb; bi++)
    {
        oint k = 0;
        for(oint i = 0; i < dim && bi * dim + i < m; i++)
            for(oint j = ptr[bi * dim + i] - base; j < ptr[bi * dim + i + 1] - base; j++)
            {
                const oint bc = (ind[j] - base) / dim;
                if(!hit[bc])
                {
                    hit[bc]   = 1;
                    undo[k++] = bc;
                }
            }
        bsr_ptr[bi + 1] = bsr_ptr[bi] + k;
        while(k > 0)
            hit[undo[--k]] = 0;
    }
    free(hit);
    free(undo);
    return bsr_ptr[mb] - base;
}

/* bsr_val must come in zero-filled; rowmajor: element (i,j) of a block at i*dim+j, else i+j*dim */
int orc_dcsr2bsr(oint m, oint n, int base, int rowmajor, const double *val, const oint *ptr, const oint *ind, oint dim,
                 double *bsr_val, const oint *bsr_ptr, oint *bsr_ind)
{
    const oint   mb = (m + dim - 1) / dim, nb = (n + dim - 1) / dim;
    const size_t sq = (size_t)dim * dim;
    long long   *at = (long long *)malloc(sizeof(long long) * ((size_t)nb + 1));
    double      *sw = (double *)malloc(sizeof(double) * sq);
    if(!at || !sw)
    {
        free(at);
        free(sw);
        return ORC_MEMORY_ERROR;
    }
    for(oint k = 0; k < nb; k++)
        at[k] = -1;
    for(oint bi = 0; bi < mb; bi++)
    {
        oint w = bsr_ptr[bi] - base;
        for(oint i = 0; i < dim && bi * dim + i < m; i++)
            for(oint p = ptr[bi * dim + i] - base; p < ptr[bi * dim + i + 1] - base; p++)
            {
                const oint c = ind[p] - base, bc = c / dim, j = c % dim;
                if(at[bc] < 0)
                {
                    at[bc]       = (long long)w * (long long)sq;
                    bsr_ind[w++] = bc + base;
                }
                bsr_val[at[bc] + (rowmajor ? (size_t)i * dim + j : (size_t)i + (size_t)j * dim)] = val[p];
            }
        for(oint k = bsr_ptr[bi] - base; k < bsr_ptr[bi + 1] - base; k++)
            at[bsr_ind[k] - base] = -1;
    }
    /* the reference's bubble sort of every block row by block column (:518-544) */
    for(oint bi = 0; bi < mb; bi++)
    {
        const oint b = bsr_ptr[bi] - base, e = bsr_ptr[bi + 1] - base;
        for(oint pass = b; pass < e; pass++)
            for(oint k = b; k < e - 1; k++)
                if(bsr_ind[k] > bsr_ind[k + 1])
                {
                    memcpy(sw, bsr_val + sq * k, sizeof(double) * sq);
                    memcpy(bsr_val + sq * k, bsr_val + sq * (k + 1), sizeof(double) * sq);
                    memcpy(bsr_val + sq * (k + 1), sw, sizeof(double) * sq);
                    const oint t   = bsr_ind[k];
                    bsr_ind[k]     = bsr_ind[k + 1];
                    bsr_ind[k + 1] = t;
                }
    }
    free(at);
    free(sw);
    return ORC_SUCCESS;
}

void orc_dbsrmv(double alpha, oint mb, oint dim, int base, const double *val, const oint *col, const oint *ptr,
                const double *x, double beta, double *y)
{
    const size_t sq = (size_t)dim * dim;
    for(oint ai = 0; ai < mb; ai++)
        for(oint bi = 0; bi < dim; bi++)
        {
            double sum = 0.0;
            for(oint aj = ptr[ai] - base; aj < ptr[ai + 1] - base; aj++)
            {
                const double *v  = val + sq * aj + bi;
                const double *xp = x + (size_t)dim * (col[aj] - base);
                for(oint bj = 0; bj < dim; bj++)
                    sum = fma(v[(size_t)dim * bj], xp[bj], sum);
            }
            if(alpha != 1.0)
                sum = sum * alpha;
            if(beta != 0.0)
                sum = fma(beta, y[(size_t)ai * dim + bi], sum);
            y[(size_t)ai * dim + bi] = sum;
        }
}


/* ---- forward SOR sweep: solvers/aoclsparse_sorv.hpp:78-113 and :212-226 (x = alpha*x first; exact zeros for
 * alpha == 0).  Returns 5 (invalid_value) when a row lacks a single non-zero diagonal entry (:32-75). */
int orc_dsorv(oint n, int base, const oint *ptr, const oint *ind, const double *val, double omega, double alpha,
              double *x, const double *b)
{
    for(oint i = 0; i < n; i++)
    {
        int found = 0;
        for(oint j = ptr[i] - base; j < ptr[i + 1] - base; j++)
            if(ind[j] - base == i)
            {
                if(found || val[j] == 0.0)
                    return ORC_INVALID_VALUE;
                found = 1;
            }
        if(!found)
            return ORC_INVALID_VALUE;
    }
    for(oint i = 0; i < n; i++)
        x[i] = alpha != 0.0 ? alpha * x[i] : 0.0;
    for(oint i = 0; i < n; i++)
    {
        double axi = 0.0, d = 1.0;
        for(oint j = ptr[i] - base; j < ptr[i + 1] - base; j++)
        {
            const oint c = ind[j] - base;
            if(c != i)
                axi = fma(val[j], x[c], axi);
            else
                d = val[j];
        }
        x[i] = fma(omega, (b[i] - axi) / d - x[i], x[i]);
    }
    return ORC_SUCCESS;
}
