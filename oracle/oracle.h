/*
 * oracle.h -- CPU restatement of the AOCL-Sparse hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: plain C99, scalar loops with explicit fma() wherever the
 * reference (built with -ffp-contract=fast, CMakeLists.txt:190) contracts a*b+c.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (aocl-sparse_amd/) never links, loads or calls anything here.
 *
 * Pinning: the reference library itself is UNBUILDABLE in this image without stand-ins
 * (it needs the absent AOCL-Utils header Au/Cpuid/X86Cpu.hh, library/src/include/
 * aoclsparse_context.hpp:38, and the cmake-generated aoclsparse_version.h,
 * library/include/aoclsparse.h:61), so oracle/_ref does not exist.  The oracle is
 * pinned instead against the known-answer vectors the reference's own unit tests
 * hold for this path (the JSON files under tests/golden, produced by tests/golden/make_fixtures.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/library/src unless noted).  Status codes follow
 * library/include/aoclsparse_types.h:304-324.
 */
#ifndef ORACLE_H_
#define ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t oint; /* aoclsparse_int in the LP64 build (aoclsparse_types.h:54-58) */

enum
{
    ORC_SUCCESS             = 0,
    ORC_NOT_IMPLEMENTED     = 1,
    ORC_INVALID_POINTER     = 2,
    ORC_INVALID_SIZE        = 3,
    ORC_INTERNAL_ERROR      = 4,
    ORC_INVALID_VALUE       = 5,
    ORC_INVALID_INDEX_VALUE = 6,
    ORC_WRONG_TYPE          = 9,
    ORC_MEMORY_ERROR        = 10,
    ORC_NUMERICAL_ERROR     = 11,
    ORC_INVALID_KID         = 14
};

/* sort classes, aoclsparse_types.h:376-384 */
enum
{
    ORC_UNKNOWN_SORT     = 0,
    ORC_FULLY_SORTED     = 1,
    ORC_PARTIALLY_SORTED = 2,
    ORC_UNSORTED         = 3
};

/* ---- SpMV, y = alpha*A*x + beta*y, general non-transposed CSR ---------------------- */
/* level2/aoclsparse_csrmv_kr.hpp:448-513 (ref_csrmv_gn): scalar left-to-right FMA chain. */
int orc_dcsrmv_ref(int base, double alpha, oint m, const double *val, const oint *col,
                   const oint *row, const double *x, double beta, double *y);
int orc_scsrmv_ref(int base, float alpha, oint m, const float *val, const oint *col,
                   const oint *row, const float *x, float beta, float *y);
/* level2/aoclsparse_csrmv_kr.hpp:949-1040 (AVX2, 4 lanes, kid 1/2). */
int orc_dcsrmv_lane4(int base, double alpha, oint m, const double *val, const oint *col,
                     const oint *row, const double *x, double beta, double *y);
/* level2/aoclsparse_csrmv_avx512.cpp:36-134 (AVX-512, 8 lanes, kid 3). */
int orc_dcsrmv_lane8(int base, double alpha, oint m, const double *val, const oint *col,
                     const oint *row, const double *x, double beta, double *y);
/* level2/aoclsparse_csrmv_kr.hpp:734-831 (float, AVX2 8 lanes; the only float gn kernel). */
int orc_scsrmv_lane8(int base, float alpha, oint m, const float *val, const oint *col,
                     const oint *row, const float *x, float beta, float *y);
/* Dispatch rule of level2/aoclsparse_csrmv.hpp:322-355: nnz<=10*m forces kid 0;
 * kid<0 (auto) on an AVX-512 host resolves to kid 3.  Returns ORC_INVALID_KID for kid>3. */
int orc_dcsrmv(int kid, int base, double alpha, oint m, oint nnz, const double *val,
               const oint *col, const oint *row, const double *x, double beta, double *y);
/* Transposed SpMV, single-thread order of level2/aoclsparse_csrmv_kt.cpp:96-214 with one
 * thread (y scaled first, then y[col] += val*(alpha*x[i]) row by row). */
int orc_dcsrmvt(int base, double alpha, oint m, oint n, const double *val, const oint *col,
                const oint *row, const double *x, double beta, double *y);
/* Symmetric / triangular SpMV of the reference (serial kernels):
 * csrmv_kr.hpp:41-92 (raw csrmv, symmetric descriptor), :107-186 (clean CSR, symmetric),
 * :658-728 (triangular), :577-649 (triangular transposed).  fill 0 lower/1 upper; diag 0/1/2. */
int orc_dcsrmv_symm_raw(int base, double alpha, oint m, const double *val, const oint *col,
                        const oint *row, const double *x, double beta, double *y);
int orc_dcsrmv_symm(int base, double alpha, oint m, int diag, int fill, const double *val,
                    const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
                    const double *x, double beta, double *y);
int orc_dcsrmv_tri(int base, double alpha, oint m, int diag, int fill, const double *val,
                   const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
                   const double *x, double beta, double *y);
int orc_dcsrmv_tri_t(int base, double alpha, oint m, oint n, int diag, int fill, const double *val,
                     const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
                     const double *x, double beta, double *y);
/* OpenMP static-row version of orc_dcsrmv (same per-row arithmetic); CPU baseline leg. */
int orc_dcsrmv_omp(int kid, int base, double alpha, oint m, oint nnz, const double *val,
                   const oint *col, const oint *row, const double *x, double beta, double *y,
                   int nthreads);
int orc_max_threads(void);
int orc_dcsrmv_bench(int kid, int base, oint m, oint n, oint nnz, const double *val, const oint *col,
                     const oint *row, const double *x, int nthreads, int passes, double *seconds, double *y_out);

/* ---- TRSV, level2/aoclsparse_trsv_kr.hpp:38-222 ------------------------------------- */
/* ilend = idiag for the L kernels, iurow for the U kernels (trsv.cpp:381-400). */
int orc_dtrsv_l(double alpha, oint m, int base, const double *a, const oint *icol,
                const oint *ilrow, const oint *idiag, const double *b, oint incb, double *x,
                oint incx, int unit);
int orc_dtrsv_lt(double alpha, oint m, int base, const double *a, const oint *icol,
                 const oint *ilrow, const oint *idiag, const double *b, oint incb, double *x,
                 oint incx, int unit);
int orc_dtrsv_u(double alpha, oint m, int base, const double *a, const oint *icol,
                const oint *ilrow, const oint *iurow, const double *b, oint incb, double *x,
                oint incx, int unit);
int orc_dtrsv_ut(double alpha, oint m, int base, const double *a, const oint *icol,
                 const oint *ilrow, const oint *iurow, const double *b, oint incb, double *x,
                 oint incx, int unit);
int orc_strsv_l(float alpha, oint m, int base, const float *a, const oint *icol,
                const oint *ilrow, const oint *idiag, const float *b, oint incb, float *x,
                oint incx, int unit);
int orc_strsv_u(float alpha, oint m, int base, const float *a, const oint *icol,
                const oint *ilrow, const oint *iurow, const float *b, oint incb, float *x,
                oint incx, int unit);

/* 1 (default): every "acc += a*b" of the reference is one fma (clang / AOCC build, or GCC without znver tuning);
 * 0: loop-carried scalar accumulations are a multiplication and an addition (GCC build with the reference's -march=znver2). */
void orc_set_contract(int fused);
int  orc_get_contract(void);

/* ---- KT TRSV kernels, level2/aoclsparse_trsv_kt.cpp:64-531 (kid 1/2: tsz 4 double / 8 float; kid 3: 8 / 16) ---- */
double orc_kt_hsum_d(int tsz, const double *v);
float  orc_kt_hsum_s(int tsz, const float *v);
#define ORC_DECL_KT_TRSV(T, SUF, K)                                                                      \
    int orc_##SUF##trsv_kt_##K(int tsz, T alpha, oint m, int base, const T *a, const oint *icol,         \
                               const oint *ilrow, const oint *ilend, const T *b, oint incb, T *x,        \
                               oint incx, int unit);
ORC_DECL_KT_TRSV(double, d, l)
ORC_DECL_KT_TRSV(double, d, lt)
ORC_DECL_KT_TRSV(double, d, u)
ORC_DECL_KT_TRSV(double, d, ut)
ORC_DECL_KT_TRSV(float, s, l)
ORC_DECL_KT_TRSV(float, s, lt)
ORC_DECL_KT_TRSV(float, s, u)
ORC_DECL_KT_TRSV(float, s, ut)

/* ---- csrmm, C = alpha*A*B + beta*C, level3/aoclsparse_csrmm.hpp:36-144, 361-427 ------ */
int orc_dcsrmm_col(double alpha, int base, const double *val, const oint *col,
                   const oint *row, oint m, const double *B, oint n, oint ldb, double beta,
                   double *C, oint ldc);
int orc_dcsrmm_row(double alpha, int base, const double *val, const oint *col,
                   const oint *row, oint m, const double *B, oint n, oint ldb, double beta,
                   double *C, oint ldc);
/* KT kernels (what kid 1/2/3 and the auto dispatch of an AVX host run): level3/aoclsparse_csrmm_kt.cpp:31-363;
 * psz = lanes per vector (4: kid 1/2, 8: kid 3). */
int orc_dcsrmm_col_kt(int psz, double alpha, int base, const double *val, const oint *col,
                      const oint *row, oint m, const double *B, oint n, oint ldb, double beta,
                      double *C, oint ldc);
int orc_dcsrmm_row_kt(int psz, double alpha, int base, const double *val, const oint *col,
                      const oint *row, oint m, const double *B, oint n, oint ldb, double beta,
                      double *C, oint ldc);
/* float twins: psz = 8 (256-bit, kid 1/2) or 16 (512-bit, kid 3) */
int orc_scsrmm_col_kt(int psz, float alpha, int base, const float *val, const oint *col, const oint *row, oint m,
                      const float *B, oint n, oint ldb, float beta, float *C, oint ldc);
int orc_scsrmm_row_kt(int psz, float alpha, int base, const float *val, const oint *col, const oint *row, oint m,
                      const float *B, oint n, oint ldb, float beta, float *C, oint ldc);
/* order: 0 row-major, 1 column-major (aoclsparse_types.h:289-293). */
int orc_dscale_dense(int order, double *C, oint m, oint n, oint ld, double beta);

/* ---- clean-CSR / analysis path, analysis/aoclsparse_csr_util.{cpp,hpp} --------------- */
/* csr_util.cpp:124-279; shape: 0 general, 1 lower, 2 upper. */
int orc_mat_check(oint maj, oint mind, oint nnz, const oint *ptr, const oint *ind,
                  const void *val, int shape, int base, int *sort, int *fulldiag);
/* csr_util.cpp:290-364 */
int orc_check_sort_diag(oint m, oint n, int base, const oint *ptr, const oint *ind,
                        int *sorted, int *fulldiag);
/* csr_util.cpp:389-458; idiag/iurow (length m) are written in the matrix's base. */
int orc_csr_indices(oint m, int base, const oint *ptr, const oint *ind, oint *idiag,
                    oint *iurow);
/* csr_util.hpp:765-967.  Builds the clean CSR.  If the input is already group-sorted with a
 * full diagonal, *is_internal=0 and optr/oind/oval are NOT written (the reference aliases
 * the user's arrays, base preserved).  Otherwise a 0-based sorted, diagonal-filled copy is
 * written; buffers must hold m+1 / nnz+min(m,n) / nnz+min(m,n) entries.  *onnz = new nnz.
 * idiag/iurow (length m) always written.  *fulldiag_out = A->opt_csr_full_diag. */
int orc_dcsr_optimize(oint m, oint n, oint nnz, int base, const oint *ptr, const oint *ind,
                      const double *val, oint *optr, oint *oind, double *oval, oint *onnz,
                      oint *idiag, oint *iurow, int *is_internal, int *fulldiag_out);
/* conversion/aoclsparse_convert.hpp:552-655 */
int orc_dcsr2csc(oint m, oint n, oint nnz, int base_csr, int base_csc, const oint *row_ptr,
                 const oint *col_ind, const double *val, oint *csc_row_ind, oint *csc_col_ptr,
                 double *csc_val);

/* ---- ILU(0), solvers/aoclsparse_ilu0.hpp:35-107 (test-input generator for TRSV) ------ */
int orc_dilu0(oint n, int base, oint *lu_diag_ptr, double *val, const oint *row_ptr,
              const oint *col_ind);

/* solvers/aoclsparse_ilu0.hpp:113-156: x = U^-1 L^-1 b on the factors orc_dilu0 produced */
int orc_dilu_solve(oint n, int base, const oint *lu_diag_ptr, const double *val, const oint *row_ptr,
                   const oint *col_ind, double *x, const double *b);

/* ---- symmetric Gauss-Seidel, solvers/aoclsparse_symgs.hpp:62-258 --------------------- */
/* clean CSR + idiag/iurow; type 0 general / 1 symmetric / 3 triangular; x in: guess, out: sweep */
int orc_dsymgs(int type, int fill, int trans, int base, double alpha, oint m, const double *val,
               const oint *col, const oint *ptr, const oint *idiag, const oint *iurow,
               const double *b, double *x, double *y, int fuse_mv);

/* ---- ELL family, level2/aoclsparse_ellmv.hpp + conversion/aoclsparse_convert.{hpp,cpp} ----- */
int orc_dellmv(int base, double alpha, oint m, const double *val, const oint *col, oint width,
               const double *x, double beta, double *y);
int orc_sellmv(int base, float alpha, oint m, const float *val, const oint *col, oint width,
               const float *x, float beta, float *y);
int orc_delltmv(int base, double alpha, oint m, const double *val, const oint *col, oint width,
                const double *x, double beta, double *y);
int orc_dellthybmv(int base, double alpha, oint m, const double *ell_val, const oint *ell_col,
                   oint width, oint ell_m, const double *csr_val, const oint *csr_row,
                   const oint *csr_col, const oint *map, const double *x, double beta, double *y);
int orc_csr2ell_width(oint m, const oint *row_ptr, oint *width);
int orc_csr2ellthyb_width(oint m, oint nnz, const oint *row_ptr, oint *ell_m, oint *width);
int orc_dcsr2ell(int layout, oint m, int base, const oint *row_ptr, const oint *col_ind,
                 const double *val, oint *ell_col, double *ell_val, oint width);
int orc_dcsr2ellthyb(oint m, int base, oint *ell_m, const oint *row_ptr, const oint *col_ind,
                     const double *val, oint *map, oint *ell_col, double *ell_val, oint width);

/* ---- aoclsparse_dcsrsv, level2/aoclsparse_csrsv.hpp:28-190 (no reference test holds vectors for it: cross-checked
 *      against the pinned TRSV restatement on sorted full-diagonal inputs, where both walk the same chain) */
int orc_dcsrsv(int lower, int unit, double alpha, oint m, const double *val, const oint *col, const oint *row_ptr,
               const double *x, double *y);

/* ---- BLKCSR (1/2/4 x 8 blocks + bit masks), conversion/aoclsparse_convert.cpp:36-310,
 *      level2/aoclsparse_blkcsrmv_avx512.cpp:40-369 */
oint orc_opt_blksize(oint m, oint nnz, int base, const oint *row_ptr, const oint *col_ind, oint *total_blks);
/* returns the number of blocks in *nblk; outputs sized by the caller (blk_col >= nnz, masks >= nnz*rows_blk) */
int orc_dcsr2blkcsr(oint m, oint n, oint nnz, const oint *row_ptr, const oint *col_ind, const double *val,
                    oint *blk_row_ptr, oint *blk_col, double *blk_val, unsigned char *masks, oint rows_blk,
                    int base, oint *nblk);
int orc_dblkcsrmv(int base, double alpha, oint m, const unsigned char *masks, const double *blk_val,
                  const oint *blk_col, const oint *blk_row_ptr, const double *x, double beta, double *y,
                  oint rows_blk);

/* ---- iterative solvers, solvers/aoclsparse_itsol_functions.hpp:632-1367 (level-1 steps as plain loops) */
int orc_dcg(oint n, int base, const oint *ptr, const oint *col, const double *val,
            const oint *idiag, const oint *iurow, const double *b, double *x, double rtol,
            double atol, oint maxit, int precond, double *rinfo);
int orc_dgmres(oint n, int base, const oint *ptr, const oint *col, const double *val,
               const double *b, double *x, oint m, double rtol, double atol, oint maxit,
               int precond, double *rinfo);

/* ---- sp2m (C = A*B, both general CSR), level3/aoclsparse_csr2m.cpp:46-543 ------------ */
/* stage 1: row_ptr_C (0-based, length m+1).  Returns nnz_C in *nnz_c. */
int orc_csr2m_nnz(oint m, oint n, int base_a, const oint *ptr_a, const oint *ind_a,
                  int base_b, const oint *ptr_b, const oint *ind_b, oint *ptr_c,
                  oint *nnz_c);
/* stage 2: fills ind_c/val_c in the reference's first-touch column order. */
int orc_dcsr2m_fill(oint m, oint n, int base_a, const oint *ptr_a, const oint *ind_a,
                    const double *val_a, int base_b, const oint *ptr_b, const oint *ind_b,
                    const double *val_b, const oint *ptr_c, oint *ind_c, double *val_c);

/* dense-result product, CSR -> dense, sparse sum (sp2md.hpp, convert.hpp:658-929, csradd.hpp) */
void orc_dsp2md(oint m, int base_a, const oint *ptr_a, const oint *ind_a, const double *val_a, int base_b,
                const oint *ptr_b, const oint *ind_b, const double *val_b, double alpha, double *C, long long rs,
                long long cs);
void orc_dsp2md_scale(oint outer, oint inner, oint ld, double beta, double *C);
void orc_dcsr2dense(oint m, oint n, int base, const oint *ptr, const oint *ind, const double *val, double *A,
                    oint ld, int colmajor, int mode, int fill, int diag);
oint orc_dcsradd(oint m, oint n, int base_a, const oint *ptr_a, const oint *ind_a, const double *val_a, double alpha,
                 int base_b, const oint *ptr_b, const oint *ind_b, const double *val_b, oint *ptr_c, oint *ind_c,
                 double *val_c);

/* level 1 (reference kernels of level1/aoclsparse_axpyi.hpp, aoclsparse_dot.hpp, aoclsparse_roti.hpp) */
int    orc_daxpyi(oint nnz, double a, const double *x, const oint *indx, double *y);
double orc_ddoti(oint nnz, const double *x, const oint *indx, const double *y);
int    orc_droti(oint nnz, double *x, const oint *indx, double *y, double c, double s);

/* DIA and BSR raw-array formats (convert.cpp:510-729, convert.hpp:291-551, diamv.hpp:34-70, bsrmv_kr.hpp:30-94) */
oint orc_csr2dia_ndiag(oint m, oint n, int base, const oint *ptr, const oint *ind);
int  orc_dcsr2dia(oint m, oint n, int base, const oint *ptr, const oint *ind, const double *val, oint ndiag,
                  oint *dia_offset, double *dia_val);
void orc_ddiamv(double alpha, oint m, oint n, const double *dia_val, const oint *dia_offset, oint ndiag,
                const double *x, double beta, double *y);
oint orc_csr2bsr_nnz(oint m, oint n, int base, const oint *ptr, const oint *ind, oint dim, oint *bsr_ptr);
int  orc_dcsr2bsr(oint m, oint n, int base, int rowmajor, const double *val, const oint *ptr, const oint *ind, oint dim,
                  double *bsr_val, const oint *bsr_ptr, oint *bsr_ind);
void orc_dbsrmv(double alpha, oint mb, oint dim, int base, const double *val, const oint *col, const oint *ptr,
                const double *x, double beta, double *y);

/* forward SOR sweep (solvers/aoclsparse_sorv.hpp:78-113, :212-226) */
int orc_dsorv(oint n, int base, const oint *ptr, const oint *ind, const double *val, double omega, double alpha,
              double *x, const double *b);

#ifdef __cplusplus
}
#endif
#endif
