/*
 * aoclsparse.h -- C ABI of the MI355X-native CSR SpMV / SpMM / TRSV engine.
 *
 * Drop-in boundary: every declaration below has the same name, argument order, argument
 * meaning, enum values and status codes as the AOCL-Sparse (v5.3.2) entry point it
 * replaces; the reference declaration is cited as file:line relative to the reference's
 * library/include/.  Only the hot path named in DESIGN.md is provided (LP64:
 * aoclsparse_int is int32).  Pointers may be host pointers (reference semantics: staged
 * over PCIe, synchronous) or device pointers of the current HIP device (extension:
 * stream-ordered, see aoclsparse_mi355.h).
 */
#ifndef AOCLSPARSE_H_
#define AOCLSPARSE_H_

#include <stdbool.h>
#include <stdint.h>

#if defined(__GNUC__)
#define DLL_PUBLIC __attribute__((visibility("default")))
#else
#define DLL_PUBLIC
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types: aoclsparse_types.h ---------------------------------------------------- */
typedef int32_t aoclsparse_int; /* :54-58 (LP64 build) */

typedef struct aoclsparse_float_complex_ /* :77-87 */
{
    float real, imag;
} aoclsparse_float_complex;
typedef struct aoclsparse_double_complex_ /* :89-99 */
{
    double real, imag;
} aoclsparse_double_complex;

typedef struct _aoclsparse_mat_descr *aoclsparse_mat_descr; /* :114 */
typedef struct _aoclsparse_matrix    *aoclsparse_matrix; /* :140 */

typedef enum aoclsparse_operation_ /* :151-156 */
{
    aoclsparse_operation_none                = 111,
    aoclsparse_operation_transpose           = 112,
    aoclsparse_operation_conjugate_transpose = 113
} aoclsparse_operation;

typedef enum aoclsparse_index_base_ /* :162-166 */
{
    aoclsparse_index_base_zero = 0,
    aoclsparse_index_base_one  = 1
} aoclsparse_index_base;

typedef enum aoclsparse_matrix_type_ /* :172-185 */
{
    aoclsparse_matrix_type_general    = 0,
    aoclsparse_matrix_type_symmetric  = 1,
    aoclsparse_matrix_type_hermitian  = 2,
    aoclsparse_matrix_type_triangular = 3
} aoclsparse_matrix_type;

typedef enum aoclsparse_matrix_data_type_ /* :191-197 */
{
    aoclsparse_dmat = 0,
    aoclsparse_smat = 1,
    aoclsparse_cmat = 2,
    aoclsparse_zmat = 3
} aoclsparse_matrix_data_type;

typedef enum aoclsparse_matrix_format_type_ /* :214-239; only csr is produced here */
{
    aoclsparse_csr_mat           = 0,
    aoclsparse_ell_mat           = 1,
    aoclsparse_ellt_mat          = 2,
    aoclsparse_ellt_csr_hyb_mat  = 3,
    aoclsparse_ell_csr_hyb_mat   = 4,
    aoclsparse_dia_mat           = 5,
    aoclsparse_csr_mat_br4       = 6,
    aoclsparse_coo_mat           = 7,
    aoclsparse_tcsr_mat          = 8,
    aoclsparse_blkcsr_mat        = 9,
    aoclsparse_bsr_mat           = 10,
    aoclsparse_uninitialized_mat = 11
} aoclsparse_matrix_format_type;

typedef enum aoclsparse_diag_type_ /* :248-257 */
{
    aoclsparse_diag_type_non_unit = 0,
    aoclsparse_diag_type_unit     = 1,
    aoclsparse_diag_type_zero     = 2
} aoclsparse_diag_type;

typedef enum aoclsparse_fill_mode_ /* :266-270 */
{
    aoclsparse_fill_mode_lower = 0,
    aoclsparse_fill_mode_upper = 1
} aoclsparse_fill_mode;

typedef enum aoclsparse_order_ /* :289-293 */
{
    aoclsparse_order_row    = 0,
    aoclsparse_order_column = 1
} aoclsparse_order;

typedef enum aoclsparse_sor_type_ /* :356-361 */
{
    aoclsparse_sor_forward   = 0,
    aoclsparse_sor_backward  = 1,
    aoclsparse_sor_symmetric = 2
} aoclsparse_sor_type;

typedef enum aoclsparse_status_ /* :304-324 */
{
    aoclsparse_status_success             = 0,
    aoclsparse_status_not_implemented     = 1,
    aoclsparse_status_invalid_pointer     = 2,
    aoclsparse_status_invalid_size        = 3,
    aoclsparse_status_internal_error      = 4,
    aoclsparse_status_invalid_value       = 5,
    aoclsparse_status_invalid_index_value = 6,
    aoclsparse_status_maxit               = 7,
    aoclsparse_status_user_stop           = 8,
    aoclsparse_status_wrong_type          = 9,
    aoclsparse_status_memory_error        = 10,
    aoclsparse_status_numerical_error     = 11,
    aoclsparse_status_invalid_operation   = 12,
    aoclsparse_status_unsorted_input      = 13,
    aoclsparse_status_invalid_kid         = 14
} aoclsparse_status;

typedef enum aoclsparse_request_ /* :335-347 */
{
    aoclsparse_stage_nnz_count        = 0,
    aoclsparse_stage_finalize         = 1,
    aoclsparse_stage_full_computation = 2
} aoclsparse_request;

typedef enum aoclsparse_memory_usage_ /* :369-374 */
{
    aoclsparse_memory_usage_minimal      = 0,
    aoclsparse_memory_usage_unrestricted = 1
} aoclsparse_memory_usage;

/* ---- auxiliary: aoclsparse_auxiliary.h -------------------------------------------- */
DLL_PUBLIC const char *aoclsparse_get_version(void); /* :47 */
DLL_PUBLIC aoclsparse_status aoclsparse_enable_instructions(const char isa_preference[]); /* :89 */
DLL_PUBLIC aoclsparse_status aoclsparse_debug_get(char            isa_preference[], /* :110-114 */
                                                  aoclsparse_int *num_threads,
                                                  char            tl_isa_preference[],
                                                  bool           *is_isa_updated,
                                                  char            arch[]);
DLL_PUBLIC aoclsparse_int aoclsparse_is_avx512_build(void); /* :1123 */

DLL_PUBLIC aoclsparse_status aoclsparse_create_mat_descr(aoclsparse_mat_descr *descr); /* :131 */
DLL_PUBLIC aoclsparse_status aoclsparse_copy_mat_descr(aoclsparse_mat_descr       dest, /* :148 */
                                                       const aoclsparse_mat_descr src);
DLL_PUBLIC aoclsparse_status aoclsparse_destroy_mat_descr(aoclsparse_mat_descr descr); /* :165 */
DLL_PUBLIC aoclsparse_status aoclsparse_set_mat_index_base(aoclsparse_mat_descr  descr, /* :184 */
                                                           aoclsparse_index_base base);
DLL_PUBLIC aoclsparse_index_base aoclsparse_get_mat_index_base(const aoclsparse_mat_descr descr);
DLL_PUBLIC aoclsparse_status aoclsparse_set_mat_type(aoclsparse_mat_descr   descr, /* :222 */
                                                     aoclsparse_matrix_type type);
DLL_PUBLIC aoclsparse_matrix_type aoclsparse_get_mat_type(const aoclsparse_mat_descr descr);
DLL_PUBLIC aoclsparse_status aoclsparse_set_mat_fill_mode(aoclsparse_mat_descr descr, /* :258 */
                                                          aoclsparse_fill_mode fill_mode);
DLL_PUBLIC aoclsparse_fill_mode aoclsparse_get_mat_fill_mode(const aoclsparse_mat_descr descr);
DLL_PUBLIC aoclsparse_status aoclsparse_set_mat_diag_type(aoclsparse_mat_descr descr, /* :293 */
                                                          aoclsparse_diag_type diag_type);
DLL_PUBLIC aoclsparse_diag_type aoclsparse_get_mat_diag_type(const aoclsparse_mat_descr descr);

/* The arrays are aliased, not copied (:364-370); they must outlive the handle. */
DLL_PUBLIC aoclsparse_status aoclsparse_create_scsr(aoclsparse_matrix    *mat, /* :400-407 */
                                                    aoclsparse_index_base base,
                                                    aoclsparse_int        M,
                                                    aoclsparse_int        N,
                                                    aoclsparse_int        nnz,
                                                    aoclsparse_int       *row_ptr,
                                                    aoclsparse_int       *col_idx,
                                                    float                *val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_dcsr(aoclsparse_matrix    *mat, /* :410-417 */
                                                    aoclsparse_index_base base,
                                                    aoclsparse_int        M,
                                                    aoclsparse_int        N,
                                                    aoclsparse_int        nnz,
                                                    aoclsparse_int       *row_ptr,
                                                    aoclsparse_int       *col_idx,
                                                    double               *val);
/* Returns internal pointers (no copy), the optimized CSR if one exists. */
DLL_PUBLIC aoclsparse_status aoclsparse_export_scsr(const aoclsparse_matrix mat, /* :786-793 */
                                                    aoclsparse_index_base  *base,
                                                    aoclsparse_int         *m,
                                                    aoclsparse_int         *n,
                                                    aoclsparse_int         *nnz,
                                                    aoclsparse_int        **row_ptr,
                                                    aoclsparse_int        **col_ind,
                                                    float                 **val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_dcsr(const aoclsparse_matrix mat, /* :795-802 */
                                                    aoclsparse_index_base  *base,
                                                    aoclsparse_int         *m,
                                                    aoclsparse_int         *n,
                                                    aoclsparse_int         *nnz,
                                                    aoclsparse_int        **row_ptr,
                                                    aoclsparse_int        **col_ind,
                                                    double                **val);
DLL_PUBLIC aoclsparse_status aoclsparse_destroy(aoclsparse_matrix *mat); /* :836 */
/* Value mutation writes through to the aliased user arrays and drops every derived / device copy
 * (:340-356, :735-746); aoclsparse_copy makes a deep copy that owns its arrays (:1010-1012). */
DLL_PUBLIC aoclsparse_status aoclsparse_sset_value(aoclsparse_matrix A,
                                                   aoclsparse_int    row_idx,
                                                   aoclsparse_int    col_idx,
                                                   float             val);
DLL_PUBLIC aoclsparse_status aoclsparse_dset_value(aoclsparse_matrix A,
                                                   aoclsparse_int    row_idx,
                                                   aoclsparse_int    col_idx,
                                                   double            val);
DLL_PUBLIC aoclsparse_status aoclsparse_supdate_values(aoclsparse_matrix A, aoclsparse_int len, float *val);
DLL_PUBLIC aoclsparse_status aoclsparse_dupdate_values(aoclsparse_matrix A, aoclsparse_int len, double *val);
DLL_PUBLIC aoclsparse_status aoclsparse_copy(const aoclsparse_matrix    src,
                                             const aoclsparse_mat_descr descr,
                                             aoclsparse_matrix         *dest);

/* ---- analysis: aoclsparse_analysis.h ----------------------------------------------- */
DLL_PUBLIC aoclsparse_status aoclsparse_optimize(aoclsparse_matrix mat); /* :56 */
DLL_PUBLIC aoclsparse_status aoclsparse_set_mv_hint(aoclsparse_matrix          mat, /* :88-92 */
                                                    aoclsparse_operation       trans,
                                                    const aoclsparse_mat_descr descr,
                                                    aoclsparse_int expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_mv_hint_kid(aoclsparse_matrix          mat,
                                                        aoclsparse_operation       trans,
                                                        const aoclsparse_mat_descr descr,
                                                        aoclsparse_int expected_no_of_calls,
                                                        aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_set_sv_hint(aoclsparse_matrix          mat,
                                                    aoclsparse_operation       trans,
                                                    const aoclsparse_mat_descr descr,
                                                    aoclsparse_int expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_mm_hint(aoclsparse_matrix          mat,
                                                    aoclsparse_operation       trans,
                                                    const aoclsparse_mat_descr descr,
                                                    aoclsparse_int expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_2m_hint(aoclsparse_matrix          mat,
                                                    aoclsparse_operation       trans,
                                                    const aoclsparse_mat_descr descr,
                                                    aoclsparse_int expected_no_of_calls);
/* aoclsparse_analysis.h: the remaining hint setters record the action in the same list. */
DLL_PUBLIC aoclsparse_status aoclsparse_set_dotmv_hint(aoclsparse_matrix          mat,
                                                       aoclsparse_operation       trans,
                                                       const aoclsparse_mat_descr descr,
                                                       aoclsparse_int expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_lu_smoother_hint(aoclsparse_matrix          mat,
                                                             aoclsparse_operation       trans,
                                                             const aoclsparse_mat_descr descr,
                                                             aoclsparse_int expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_sm_hint(aoclsparse_matrix          mat,
                                                    aoclsparse_operation       trans,
                                                    const aoclsparse_mat_descr descr,
                                                    const aoclsparse_order     order,
                                                    aoclsparse_int             expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_symgs_hint(aoclsparse_matrix          mat,
                                                       aoclsparse_operation       trans,
                                                       const aoclsparse_mat_descr descr,
                                                       aoclsparse_int expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_sorv_hint(aoclsparse_matrix          mat,
                                                      const aoclsparse_mat_descr descr,
                                                      const aoclsparse_sor_type  type,
                                                      const aoclsparse_int expected_no_of_calls);
DLL_PUBLIC aoclsparse_status aoclsparse_set_memory_hint(aoclsparse_matrix             mat,
                                                        const aoclsparse_memory_usage policy);

/* ---- level 2: aoclsparse_functions.h ----------------------------------------------- */
/* y = alpha*op(A)*x + beta*y on raw CSR arrays; scalars BY POINTER (:695-721). */
DLL_PUBLIC aoclsparse_status aoclsparse_scsrmv(aoclsparse_operation       trans,
                                               const float               *alpha,
                                               aoclsparse_int             m,
                                               aoclsparse_int             n,
                                               aoclsparse_int             nnz,
                                               const float               *csr_val,
                                               const aoclsparse_int      *csr_col_ind,
                                               const aoclsparse_int      *csr_row_ptr,
                                               const aoclsparse_mat_descr descr,
                                               const float               *x,
                                               const float               *beta,
                                               float                     *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsrmv(aoclsparse_operation       trans,
                                               const double              *alpha,
                                               aoclsparse_int             m,
                                               aoclsparse_int             n,
                                               aoclsparse_int             nnz,
                                               const double              *csr_val,
                                               const aoclsparse_int      *csr_col_ind,
                                               const aoclsparse_int      *csr_row_ptr,
                                               const aoclsparse_mat_descr descr,
                                               const double              *x,
                                               const double              *beta,
                                               double                    *y);
/* Inspector-executor SpMV on a handle (:1298-1313). */
DLL_PUBLIC aoclsparse_status aoclsparse_smv(aoclsparse_operation       op,
                                            const float               *alpha,
                                            aoclsparse_matrix          A,
                                            const aoclsparse_mat_descr descr,
                                            const float               *x,
                                            const float               *beta,
                                            float                     *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dmv(aoclsparse_operation       op,
                                            const double              *alpha,
                                            aoclsparse_matrix          A,
                                            const aoclsparse_mat_descr descr,
                                            const double              *x,
                                            const double              *beta,
                                            double                    *y);
/* y = alpha*op(A)*x + beta*y and d = x.y over min(m,n) entries; scalars BY VALUE (:1782-1801). */
DLL_PUBLIC aoclsparse_status aoclsparse_sdotmv(const aoclsparse_operation op,
                                               const float                alpha,
                                               aoclsparse_matrix          A,
                                               const aoclsparse_mat_descr descr,
                                               const float               *x,
                                               const float                beta,
                                               float                     *y,
                                               float                     *d);
DLL_PUBLIC aoclsparse_status aoclsparse_ddotmv(const aoclsparse_operation op,
                                               const double               alpha,
                                               aoclsparse_matrix          A,
                                               const aoclsparse_mat_descr descr,
                                               const double              *x,
                                               const double               beta,
                                               double                    *y,
                                               double                    *d);
/* op(A)*x = alpha*b on the triangle fill_mode selects; alpha BY VALUE (:1525-1539). */
DLL_PUBLIC aoclsparse_status aoclsparse_strsv(aoclsparse_operation       trans,
                                              const float                alpha,
                                              aoclsparse_matrix          A,
                                              const aoclsparse_mat_descr descr,
                                              const float               *b,
                                              float                     *x);
DLL_PUBLIC aoclsparse_status aoclsparse_dtrsv(aoclsparse_operation       trans,
                                              const double               alpha,
                                              aoclsparse_matrix          A,
                                              const aoclsparse_mat_descr descr,
                                              const double              *b,
                                              double                    *x);
DLL_PUBLIC aoclsparse_status aoclsparse_strsv_kid(aoclsparse_operation       trans, /* :1602-1618 */
                                                  const float                alpha,
                                                  aoclsparse_matrix          A,
                                                  const aoclsparse_mat_descr descr,
                                                  const float               *b,
                                                  float                     *x,
                                                  aoclsparse_int             kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dtrsv_kid(aoclsparse_operation       trans,
                                                  const double               alpha,
                                                  aoclsparse_matrix          A,
                                                  const aoclsparse_mat_descr descr,
                                                  const double              *b,
                                                  double                    *x,
                                                  aoclsparse_int             kid);
DLL_PUBLIC aoclsparse_status aoclsparse_strsv_strided(aoclsparse_operation       trans, /* :1647-1665 */
                                                      const float                alpha,
                                                      aoclsparse_matrix          A,
                                                      const aoclsparse_mat_descr descr,
                                                      const float               *b,
                                                      const aoclsparse_int       incb,
                                                      float                     *x,
                                                      const aoclsparse_int       incx);
DLL_PUBLIC aoclsparse_status aoclsparse_dtrsv_strided(aoclsparse_operation       trans,
                                                      const double               alpha,
                                                      aoclsparse_matrix          A,
                                                      const aoclsparse_mat_descr descr,
                                                      const double              *b,
                                                      const aoclsparse_int       incb,
                                                      double                    *x,
                                                      const aoclsparse_int       incx);

/* ---- level 3 ------------------------------------------------------------------------ */
/* op(A)*X = alpha*B for n right-hand sides (dense B, X in the stated order) (:1968-2101). */
DLL_PUBLIC aoclsparse_status aoclsparse_strsm(const aoclsparse_operation trans,
                                              const float                alpha,
                                              aoclsparse_matrix          A,
                                              const aoclsparse_mat_descr descr,
                                              aoclsparse_order           order,
                                              const float               *B,
                                              aoclsparse_int             n,
                                              aoclsparse_int             ldb,
                                              float                     *X,
                                              aoclsparse_int             ldx);
DLL_PUBLIC aoclsparse_status aoclsparse_dtrsm(const aoclsparse_operation trans,
                                              const double               alpha,
                                              aoclsparse_matrix          A,
                                              const aoclsparse_mat_descr descr,
                                              aoclsparse_order           order,
                                              const double              *B,
                                              aoclsparse_int             n,
                                              aoclsparse_int             ldb,
                                              double                    *X,
                                              aoclsparse_int             ldx);
DLL_PUBLIC aoclsparse_status aoclsparse_strsm_kid(const aoclsparse_operation trans,
                                                  const float                alpha,
                                                  aoclsparse_matrix          A,
                                                  const aoclsparse_mat_descr descr,
                                                  aoclsparse_order           order,
                                                  const float               *B,
                                                  aoclsparse_int             n,
                                                  aoclsparse_int             ldb,
                                                  float                     *X,
                                                  aoclsparse_int             ldx,
                                                  const aoclsparse_int       kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dtrsm_kid(const aoclsparse_operation trans,
                                                  const double               alpha,
                                                  aoclsparse_matrix          A,
                                                  const aoclsparse_mat_descr descr,
                                                  aoclsparse_order           order,
                                                  const double              *B,
                                                  aoclsparse_int             n,
                                                  aoclsparse_int             ldb,
                                                  double                    *X,
                                                  aoclsparse_int             ldx,
                                                  const aoclsparse_int       kid);
/* ---- complex handles: creation, export, mutation and y = alpha op(A) x + beta y for every
 * descriptor type (general / symmetric / hermitian / triangular) and operation (N / T / H)
 * (aoclsparse_auxiliary.h:340-345,419-436,735-741,804-822; aoclsparse_functions.h:1280-1296).  The other complex
 * executors (dotmv, csrmm, trsv / trsm, symgs, ilu_smoother, itsol, sp2m) are declared with their real twins below;
 * an executor of one value type returns wrong_type for a handle of another. */
DLL_PUBLIC aoclsparse_status aoclsparse_create_ccsr(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                                    aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                                    aoclsparse_int *row_ptr, aoclsparse_int *col_idx,
                                                    aoclsparse_float_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_zcsr(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                                    aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                                    aoclsparse_int *row_ptr, aoclsparse_int *col_idx,
                                                    aoclsparse_double_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_ccsr(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **row_ptr, aoclsparse_int **col_ind,
                                                    aoclsparse_float_complex **val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_zcsr(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **row_ptr, aoclsparse_int **col_ind,
                                                    aoclsparse_double_complex **val);
DLL_PUBLIC aoclsparse_status aoclsparse_cset_value(aoclsparse_matrix A, aoclsparse_int row_idx,
                                                   aoclsparse_int col_idx, aoclsparse_float_complex val);
DLL_PUBLIC aoclsparse_status aoclsparse_zset_value(aoclsparse_matrix A, aoclsparse_int row_idx,
                                                   aoclsparse_int col_idx, aoclsparse_double_complex val);
DLL_PUBLIC aoclsparse_status aoclsparse_cupdate_values(aoclsparse_matrix A, aoclsparse_int len,
                                                       aoclsparse_float_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_zupdate_values(aoclsparse_matrix A, aoclsparse_int len,
                                                       aoclsparse_double_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_cmv(aoclsparse_operation op, const aoclsparse_float_complex *alpha,
                                            aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                            const aoclsparse_float_complex *x,
                                            const aoclsparse_float_complex *beta, aoclsparse_float_complex *y);
DLL_PUBLIC aoclsparse_status aoclsparse_zmv(aoclsparse_operation op, const aoclsparse_double_complex *alpha,
                                            aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                            const aoclsparse_double_complex *x,
                                            const aoclsparse_double_complex *beta, aoclsparse_double_complex *y);
/* C = alpha op(A) B + beta C for complex handles (aoclsparse_functions.h:2461-2484, :3398-3425): general,
 * symmetric and hermitian descriptors, alpha / beta BY VALUE */
DLL_PUBLIC aoclsparse_status aoclsparse_ccsrmm(aoclsparse_operation op, const aoclsparse_float_complex alpha,
                                               const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               aoclsparse_order order, const aoclsparse_float_complex *B,
                                               aoclsparse_int n, aoclsparse_int ldb,
                                               const aoclsparse_float_complex beta, aoclsparse_float_complex *C,
                                               aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_zcsrmm(aoclsparse_operation op, const aoclsparse_double_complex alpha,
                                               const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               aoclsparse_order order, const aoclsparse_double_complex *B,
                                               aoclsparse_int n, aoclsparse_int ldb,
                                               const aoclsparse_double_complex beta, aoclsparse_double_complex *C,
                                               aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_ccsrmm_kid(aoclsparse_operation op, const aoclsparse_float_complex alpha,
                                                   const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                   aoclsparse_order order, const aoclsparse_float_complex *B,
                                                   aoclsparse_int n, aoclsparse_int ldb,
                                                   const aoclsparse_float_complex beta, aoclsparse_float_complex *C,
                                                   aoclsparse_int ldc, const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zcsrmm_kid(aoclsparse_operation op, const aoclsparse_double_complex alpha,
                                                   const aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                   aoclsparse_order order, const aoclsparse_double_complex *B,
                                                   aoclsparse_int n, aoclsparse_int ldb,
                                                   const aoclsparse_double_complex beta,
                                                   aoclsparse_double_complex *C, aoclsparse_int ldc,
                                                   const aoclsparse_int kid);

/* y = alpha op(A) x + beta y followed by the conjugated dot d = sum conj(x_i) y_i (aoclsparse_functions.h:1925-1990) */
DLL_PUBLIC aoclsparse_status aoclsparse_cdotmv(const aoclsparse_operation op, const aoclsparse_float_complex alpha,
                                               aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               const aoclsparse_float_complex *x, const aoclsparse_float_complex beta,
                                               aoclsparse_float_complex *y, aoclsparse_float_complex *d);
DLL_PUBLIC aoclsparse_status aoclsparse_zdotmv(const aoclsparse_operation op, const aoclsparse_double_complex alpha,
                                               aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               const aoclsparse_double_complex *x,
                                               const aoclsparse_double_complex beta, aoclsparse_double_complex *y,
                                               aoclsparse_double_complex *d);
/* complex symmetric Gauss-Seidel sweeps (aoclsparse_solvers.h: aoclsparse_?symgs(_mv)(_kid)): same checks and
 * composition as the real ones; a hermitian descriptor uses the conjugate transpose of the stored triangle. */
DLL_PUBLIC aoclsparse_status aoclsparse_csymgs(aoclsparse_operation trans, aoclsparse_matrix A,
                                                const aoclsparse_mat_descr descr, const aoclsparse_float_complex alpha,
                                                const aoclsparse_float_complex *b, aoclsparse_float_complex *x);
DLL_PUBLIC aoclsparse_status aoclsparse_csymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                    const aoclsparse_mat_descr descr, const aoclsparse_float_complex alpha,
                                                    const aoclsparse_float_complex *b, aoclsparse_float_complex *x, const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_csymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                                   const aoclsparse_mat_descr descr, const aoclsparse_float_complex alpha,
                                                   const aoclsparse_float_complex *b, aoclsparse_float_complex *x, aoclsparse_float_complex *y);
DLL_PUBLIC aoclsparse_status aoclsparse_csymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                       const aoclsparse_mat_descr descr, const aoclsparse_float_complex alpha,
                                                       const aoclsparse_float_complex *b, aoclsparse_float_complex *x, aoclsparse_float_complex *y, const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zsymgs(aoclsparse_operation trans, aoclsparse_matrix A,
                                                const aoclsparse_mat_descr descr, const aoclsparse_double_complex alpha,
                                                const aoclsparse_double_complex *b, aoclsparse_double_complex *x);
DLL_PUBLIC aoclsparse_status aoclsparse_zsymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                    const aoclsparse_mat_descr descr, const aoclsparse_double_complex alpha,
                                                    const aoclsparse_double_complex *b, aoclsparse_double_complex *x, const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zsymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                                   const aoclsparse_mat_descr descr, const aoclsparse_double_complex alpha,
                                                   const aoclsparse_double_complex *b, aoclsparse_double_complex *x, aoclsparse_double_complex *y);
DLL_PUBLIC aoclsparse_status aoclsparse_zsymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                       const aoclsparse_mat_descr descr, const aoclsparse_double_complex alpha,
                                                       const aoclsparse_double_complex *b, aoclsparse_double_complex *x, aoclsparse_double_complex *y, const aoclsparse_int kid);
/* raw-array triangular solve y = inv(T) * alpha * x, T = the triangle of the CSR arrays named by descr->fill_mode
 * (aoclsparse_functions.h:1318-1402): zero-based, op = none, general / symmetric descriptor type, host arrays. */
DLL_PUBLIC aoclsparse_status aoclsparse_scsrsv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                               const float *csr_val, const aoclsparse_int *csr_col_ind,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_mat_descr descr,
                                               const float *x, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsrsv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                               const double *csr_val, const aoclsparse_int *csr_col_ind,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_mat_descr descr,
                                               const double *x, double *y);
/* complex triangular solves (aoclsparse_functions.h:1541-1598,1620-1697,2620-2760): same checks and semantics as the
 * real ones; op = conjugate_transpose conjugates the triangle.  kid is validated and otherwise ignored. */
DLL_PUBLIC aoclsparse_status aoclsparse_ctrsv(aoclsparse_operation trans, const aoclsparse_float_complex alpha,
                                               aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               const aoclsparse_float_complex *b, aoclsparse_float_complex *x);
DLL_PUBLIC aoclsparse_status aoclsparse_ctrsv_kid(aoclsparse_operation trans, const aoclsparse_float_complex alpha,
                                                   aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                   const aoclsparse_float_complex *b, aoclsparse_float_complex *x, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_ctrsv_strided(aoclsparse_operation trans, const aoclsparse_float_complex alpha,
                                                       aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                       const aoclsparse_float_complex *b, const aoclsparse_int incb,
                                                       aoclsparse_float_complex *x, const aoclsparse_int incx);
DLL_PUBLIC aoclsparse_status aoclsparse_ctrsm(const aoclsparse_operation trans, const aoclsparse_float_complex alpha,
                                               aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               aoclsparse_order order, const aoclsparse_float_complex *B, aoclsparse_int n,
                                               aoclsparse_int ldb, aoclsparse_float_complex *X, aoclsparse_int ldx);
DLL_PUBLIC aoclsparse_status aoclsparse_ctrsm_kid(const aoclsparse_operation trans, const aoclsparse_float_complex alpha,
                                                   aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                   aoclsparse_order order, const aoclsparse_float_complex *B, aoclsparse_int n,
                                                   aoclsparse_int ldb, aoclsparse_float_complex *X, aoclsparse_int ldx,
                                                   const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_ztrsv(aoclsparse_operation trans, const aoclsparse_double_complex alpha,
                                               aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               const aoclsparse_double_complex *b, aoclsparse_double_complex *x);
DLL_PUBLIC aoclsparse_status aoclsparse_ztrsv_kid(aoclsparse_operation trans, const aoclsparse_double_complex alpha,
                                                   aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                   const aoclsparse_double_complex *b, aoclsparse_double_complex *x, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_ztrsv_strided(aoclsparse_operation trans, const aoclsparse_double_complex alpha,
                                                       aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                       const aoclsparse_double_complex *b, const aoclsparse_int incb,
                                                       aoclsparse_double_complex *x, const aoclsparse_int incx);
DLL_PUBLIC aoclsparse_status aoclsparse_ztrsm(const aoclsparse_operation trans, const aoclsparse_double_complex alpha,
                                               aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                               aoclsparse_order order, const aoclsparse_double_complex *B, aoclsparse_int n,
                                               aoclsparse_int ldb, aoclsparse_double_complex *X, aoclsparse_int ldx);
DLL_PUBLIC aoclsparse_status aoclsparse_ztrsm_kid(const aoclsparse_operation trans, const aoclsparse_double_complex alpha,
                                                   aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                                   aoclsparse_order order, const aoclsparse_double_complex *B, aoclsparse_int n,
                                                   aoclsparse_int ldb, aoclsparse_double_complex *X, aoclsparse_int ldx,
                                                   const aoclsparse_int kid);

/* ---- other input formats and structure conversions (aoclsparse_auxiliary.h:674-1095, aoclsparse_convert.h:494-660).
 * A CSC handle behaves like the CSR handle of the same matrix in every executor (its CSR is built at creation);
 * a COO handle can be exported, mutated and converted (aoclsparse_convert_csr), executors return not_implemented. */
DLL_PUBLIC aoclsparse_status aoclsparse_create_scsc(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                                    aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                                    aoclsparse_int *col_ptr, aoclsparse_int *row_idx, float *val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_dcsc(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                                    aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                                    aoclsparse_int *col_ptr, aoclsparse_int *row_idx, double *val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_scoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                                    const aoclsparse_int M, const aoclsparse_int N,
                                                    const aoclsparse_int nnz, aoclsparse_int *row_ind,
                                                    aoclsparse_int *col_ind, float *val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_dcoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                                    const aoclsparse_int M, const aoclsparse_int N,
                                                    const aoclsparse_int nnz, aoclsparse_int *row_ind,
                                                    aoclsparse_int *col_ind, double *val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_scsc(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **col_ptr, aoclsparse_int **row_ind, float **val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_dcsc(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **col_ptr, aoclsparse_int **row_ind, double **val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_scoo(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **row_ptr, aoclsparse_int **col_ptr, float **val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_dcoo(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **row_ptr, aoclsparse_int **col_ptr, double **val);
/* new CSR handle (owning its arrays) holding op(src); src may be a CSR, CSC or COO handle */
DLL_PUBLIC aoclsparse_status aoclsparse_convert_csr(const aoclsparse_matrix src_mat, const aoclsparse_operation op,
                                                    aoclsparse_matrix *dest_mat);
/* sort the indices (and values) of every row of the caller's arrays in place */
DLL_PUBLIC aoclsparse_status aoclsparse_order_mat(aoclsparse_matrix mat);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                 const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                                 const aoclsparse_int *csr_row_ptr,
                                                 const aoclsparse_int *csr_col_ind, const float *csr_val,
                                                 aoclsparse_int *csc_row_ind, aoclsparse_int *csc_col_ptr,
                                                 float *csc_val);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                 const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                                 const aoclsparse_int *csr_row_ptr,
                                                 const aoclsparse_int *csr_col_ind, const double *csr_val,
                                                 aoclsparse_int *csc_row_ind, aoclsparse_int *csc_col_ptr,
                                                 double *csc_val);
/* complex twins of the above (aoclsparse_auxiliary.h:438-560,748-870; aoclsparse_convert.h:528-560) */
DLL_PUBLIC aoclsparse_status aoclsparse_create_ccsc(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                                    aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                                    aoclsparse_int *col_ptr, aoclsparse_int *row_idx,
                                                    aoclsparse_float_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_ccoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                                    const aoclsparse_int M, const aoclsparse_int N,
                                                    const aoclsparse_int nnz, aoclsparse_int *row_ind,
                                                    aoclsparse_int *col_ind, aoclsparse_float_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_ccsc(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **col_ptr, aoclsparse_int **row_ind,
                                                    aoclsparse_float_complex **val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_ccoo(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **row_ptr, aoclsparse_int **col_ptr,
                                                    aoclsparse_float_complex **val);
DLL_PUBLIC aoclsparse_status aoclsparse_ccsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                 const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                                 const aoclsparse_int *csr_row_ptr,
                                                 const aoclsparse_int *csr_col_ind,
                                                 const aoclsparse_float_complex *csr_val, aoclsparse_int *csc_row_ind,
                                                 aoclsparse_int *csc_col_ptr, aoclsparse_float_complex *csc_val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_zcsc(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                                    aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                                    aoclsparse_int *col_ptr, aoclsparse_int *row_idx,
                                                    aoclsparse_double_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_create_zcoo(aoclsparse_matrix *mat, const aoclsparse_index_base base,
                                                    const aoclsparse_int M, const aoclsparse_int N,
                                                    const aoclsparse_int nnz, aoclsparse_int *row_ind,
                                                    aoclsparse_int *col_ind, aoclsparse_double_complex *val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_zcsc(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **col_ptr, aoclsparse_int **row_ind,
                                                    aoclsparse_double_complex **val);
DLL_PUBLIC aoclsparse_status aoclsparse_export_zcoo(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                                    aoclsparse_int **row_ptr, aoclsparse_int **col_ptr,
                                                    aoclsparse_double_complex **val);
DLL_PUBLIC aoclsparse_status aoclsparse_zcsr2csc(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                 const aoclsparse_mat_descr descr, aoclsparse_index_base baseCSC,
                                                 const aoclsparse_int *csr_row_ptr,
                                                 const aoclsparse_int *csr_col_ind,
                                                 const aoclsparse_double_complex *csr_val, aoclsparse_int *csc_row_ind,
                                                 aoclsparse_int *csc_col_ptr, aoclsparse_double_complex *csc_val);

/* ---- ELL family: the formats the reference's optimize step stores, as raw-array products
 * (aoclsparse_functions.h:789-885) and their CSR conversions (aoclsparse_convert.h).  Only general
 * descriptors and op = none exist in the reference (anything else: not_implemented).  ELL is row-major
 * with column -1 as padding; ELLT is column-major (ell[p*m + i]); ELLT-HYB = ELLT over all rows plus the
 * rows listed in csr_row_idx_map (0-based) recomputed from the CSR arrays.  ?ellthybmv exists for double. */
DLL_PUBLIC aoclsparse_status aoclsparse_sellmv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                               aoclsparse_int n, aoclsparse_int nnz, const float *ell_val,
                                               const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                               const aoclsparse_mat_descr descr, const float *x,
                                               const float *beta, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dellmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                               aoclsparse_int n, aoclsparse_int nnz, const double *ell_val,
                                               const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                               const aoclsparse_mat_descr descr, const double *x,
                                               const double *beta, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_selltmv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                                aoclsparse_int n, aoclsparse_int nnz, const float *ell_val,
                                                const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                                const aoclsparse_mat_descr descr, const float *x,
                                                const float *beta, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_delltmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                                aoclsparse_int n, aoclsparse_int nnz, const double *ell_val,
                                                const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                                const aoclsparse_mat_descr descr, const double *x,
                                                const double *beta, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_sellthybmv(aoclsparse_operation trans, const float *alpha,
                                                   aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                   const float *ell_val, const aoclsparse_int *ell_col_ind,
                                                   aoclsparse_int ell_width, const aoclsparse_int ell_m,
                                                   const float *csr_val, const aoclsparse_int *csr_row_ind,
                                                   const aoclsparse_int *csr_col_ind, aoclsparse_int *row_idx_map,
                                                   aoclsparse_int *csr_row_idx_map,
                                                   const aoclsparse_mat_descr descr, const float *x,
                                                   const float *beta, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dellthybmv(aoclsparse_operation trans, const double *alpha,
                                                   aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                   const double *ell_val, const aoclsparse_int *ell_col_ind,
                                                   aoclsparse_int ell_width, const aoclsparse_int ell_m,
                                                   const double *csr_val, const aoclsparse_int *csr_row_ind,
                                                   const aoclsparse_int *csr_col_ind, aoclsparse_int *row_idx_map,
                                                   aoclsparse_int *csr_row_idx_map,
                                                   const aoclsparse_mat_descr descr, const double *x,
                                                   const double *beta, double *y);
/* BLKCSR: 1/2/4 x 8 blocks with one bit mask per sub-row (reference: aoclsparse_functions.h:887-900,
 * aoclsparse_convert.h:560-626; convert.cpp:36-310).  opt_blksize / csr2blkcsr are host routines. */
DLL_PUBLIC aoclsparse_status aoclsparse_dblkcsrmv(aoclsparse_operation trans, const double *alpha,
                                                  aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                  const uint8_t *masks, const double *blk_csr_val,
                                                  const aoclsparse_int *blk_col_ind,
                                                  const aoclsparse_int *blk_row_ptr,
                                                  const aoclsparse_mat_descr descr, const double *x,
                                                  const double *beta, double *y, aoclsparse_int nRowsblk);
DLL_PUBLIC aoclsparse_int aoclsparse_opt_blksize(aoclsparse_int m, aoclsparse_int nnz, aoclsparse_index_base base,
                                                 const aoclsparse_int *csr_row_ptr,
                                                 const aoclsparse_int *csr_col_ind, aoclsparse_int *total_blks);
DLL_PUBLIC aoclsparse_status aoclsparse_csr2blkcsr(aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                                   const aoclsparse_int *csr_row_ptr,
                                                   const aoclsparse_int *csr_col_ind, const double *csr_val,
                                                   aoclsparse_int *blk_row_ptr, aoclsparse_int *blk_col_ind,
                                                   double *blk_csr_val, uint8_t *masks, aoclsparse_int nRowsblk,
                                                   aoclsparse_index_base base);
DLL_PUBLIC aoclsparse_status aoclsparse_csr2ell_width(aoclsparse_int m, aoclsparse_int nnz,
                                                      const aoclsparse_int *csr_row_ptr, aoclsparse_int *ell_width);
DLL_PUBLIC aoclsparse_status aoclsparse_csr2ellthyb_width(aoclsparse_int m, aoclsparse_int nnz,
                                                          const aoclsparse_int *csr_row_ptr, aoclsparse_int *ell_m,
                                                          aoclsparse_int *ell_width);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2ell(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                                 const aoclsparse_int *csr_row_ptr,
                                                 const aoclsparse_int *csr_col_ind, const float *csr_val,
                                                 aoclsparse_int *ell_col_ind, float *ell_val,
                                                 aoclsparse_int ell_width);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2ell(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                                 const aoclsparse_int *csr_row_ptr,
                                                 const aoclsparse_int *csr_col_ind, const double *csr_val,
                                                 aoclsparse_int *ell_col_ind, double *ell_val,
                                                 aoclsparse_int ell_width);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2ellt(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                                  const aoclsparse_int *csr_row_ptr,
                                                  const aoclsparse_int *csr_col_ind, const float *csr_val,
                                                  aoclsparse_int *ell_col_ind, float *ell_val,
                                                  aoclsparse_int ell_width);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2ellt(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                                  const aoclsparse_int *csr_row_ptr,
                                                  const aoclsparse_int *csr_col_ind, const double *csr_val,
                                                  aoclsparse_int *ell_col_ind, double *ell_val,
                                                  aoclsparse_int ell_width);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2ellthyb(aoclsparse_int m, aoclsparse_index_base base,
                                                     aoclsparse_int *ell_m, const aoclsparse_int *csr_row_ptr,
                                                     const aoclsparse_int *csr_col_ind, const float *csr_val,
                                                     aoclsparse_int *row_idx_map, aoclsparse_int *csr_row_idx_map,
                                                     aoclsparse_int *ell_col_ind, float *ell_val,
                                                     aoclsparse_int ell_width);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2ellthyb(aoclsparse_int m, aoclsparse_index_base base,
                                                     aoclsparse_int *ell_m, const aoclsparse_int *csr_row_ptr,
                                                     const aoclsparse_int *csr_col_ind, const double *csr_val,
                                                     aoclsparse_int *row_idx_map, aoclsparse_int *csr_row_idx_map,
                                                     aoclsparse_int *ell_col_ind, double *ell_val,
                                                     aoclsparse_int ell_width);

/* ---- composite solvers kept device-resident (aoclsparse_solvers.h) -----------------------------
 * Symmetric Gauss-Seidel sweep (:824-1070): x holds the initial guess on entry; ?symgs_mv also returns
 * y = op(A) x.  ILU(0) smoother (:1136-1151): factorises once per handle, then x = U^-1 L^-1 b;
 * *precond_csr_val points at the library-owned factor values (host), approx_inv_diag is unused. */
DLL_PUBLIC aoclsparse_status aoclsparse_ssymgs(aoclsparse_operation trans, aoclsparse_matrix A,
                                               const aoclsparse_mat_descr descr, const float alpha,
                                               const float *b, float *x);
DLL_PUBLIC aoclsparse_status aoclsparse_dsymgs(aoclsparse_operation trans, aoclsparse_matrix A,
                                               const aoclsparse_mat_descr descr, const double alpha,
                                               const double *b, double *x);
DLL_PUBLIC aoclsparse_status aoclsparse_ssymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                   const aoclsparse_mat_descr descr, const float alpha,
                                                   const float *b, float *x, const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dsymgs_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                   const aoclsparse_mat_descr descr, const double alpha,
                                                   const double *b, double *x, const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_ssymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                                  const aoclsparse_mat_descr descr, const float alpha,
                                                  const float *b, float *x, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dsymgs_mv(aoclsparse_operation trans, aoclsparse_matrix A,
                                                  const aoclsparse_mat_descr descr, const double alpha,
                                                  const double *b, double *x, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_ssymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                      const aoclsparse_mat_descr descr, const float alpha,
                                                      const float *b, float *x, float *y,
                                                      const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dsymgs_mv_kid(aoclsparse_operation trans, aoclsparse_matrix A,
                                                      const aoclsparse_mat_descr descr, const double alpha,
                                                      const double *b, double *x, double *y,
                                                      const aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_silu_smoother(aoclsparse_operation op, aoclsparse_matrix A,
                                                      const aoclsparse_mat_descr descr,
                                                      float **precond_csr_val, const float *approx_inv_diag,
                                                      float *x, const float *b);
DLL_PUBLIC aoclsparse_status aoclsparse_dilu_smoother(aoclsparse_operation op, aoclsparse_matrix A,
                                                      const aoclsparse_mat_descr descr,
                                                      double **precond_csr_val, const double *approx_inv_diag,
                                                      double *x, const double *b);
DLL_PUBLIC aoclsparse_status aoclsparse_cilu_smoother(aoclsparse_operation op, aoclsparse_matrix A,
                                                      const aoclsparse_mat_descr descr,
                                                      aoclsparse_float_complex **precond_csr_val,
                                                      const aoclsparse_float_complex *approx_inv_diag,
                                                      aoclsparse_float_complex *x, const aoclsparse_float_complex *b);
DLL_PUBLIC aoclsparse_status aoclsparse_zilu_smoother(aoclsparse_operation op, aoclsparse_matrix A,
                                                      const aoclsparse_mat_descr descr,
                                                      aoclsparse_double_complex **precond_csr_val,
                                                      const aoclsparse_double_complex *approx_inv_diag,
                                                      aoclsparse_double_complex *x, const aoclsparse_double_complex *b);

/* ---- iterative solvers (aoclsparse_solvers.h:104-560): CG and restarted GMRES, options by name
 * ("iterative method", "cg iteration limit", "cg rel tolerance", "cg abs tolerance", "cg preconditioner",
 * "gmres iteration limit", "gmres rel tolerance", "gmres abs tolerance", "gmres preconditioner",
 * "gmres restart iterations").  rinfo[0] = residual norm, rinfo[1] = ||b|| (GMRES: rtol*||b||), rinfo[30] =
 * iterations.  The direct interface keeps every iterate in HBM; see DESIGN.md 5.8 for the RCI workspaces. */
typedef struct _aoclsparse_itsol_handle *aoclsparse_itsol_handle;
typedef enum aoclsparse_itsol_rci_job_ /* :115-133 */
{
    aoclsparse_rci_interrupt = -1,
    aoclsparse_rci_stop      = 0,
    aoclsparse_rci_start,
    aoclsparse_rci_mv,
    aoclsparse_rci_precond,
    aoclsparse_rci_stopping_criterion
} aoclsparse_itsol_rci_job;
DLL_PUBLIC void              aoclsparse_itsol_handle_prn_options(aoclsparse_itsol_handle handle);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_option_set(aoclsparse_itsol_handle handle, const char *option,
                                                         const char *value);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_d_init(aoclsparse_itsol_handle *handle);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_s_init(aoclsparse_itsol_handle *handle);
DLL_PUBLIC void              aoclsparse_itsol_destroy(aoclsparse_itsol_handle *handle);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_d_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n,
                                                          const double *b);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_s_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n,
                                                          const float *b);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_d_rci_solve(aoclsparse_itsol_handle   handle,
                                                          aoclsparse_itsol_rci_job *ircomm, double **u,
                                                          double **v, double *x, double rinfo[100]);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_s_rci_solve(aoclsparse_itsol_handle   handle,
                                                          aoclsparse_itsol_rci_job *ircomm, float **u, float **v,
                                                          float *x, float rinfo[100]);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_d_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const double *b, double *x, double rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const double *u, double *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const double *x, const double *r, double rinfo[100], void *udata),
    void *udata);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_s_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const float *b, float *x, float rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const float *u, float *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const float *x, const float *r, float rinfo[100], void *udata),
    void *udata);
/* complex handles (aoclsparse_solvers.h: aoclsparse_itsol_{c,z}_*): norms, tolerances and rinfo are real */
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_c_init(aoclsparse_itsol_handle *handle);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_c_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n,
                                                          const aoclsparse_float_complex *b);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_c_rci_solve(aoclsparse_itsol_handle   handle,
                                                          aoclsparse_itsol_rci_job *ircomm, aoclsparse_float_complex **u,
                                                          aoclsparse_float_complex **v, aoclsparse_float_complex *x, float rinfo[100]);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_c_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const aoclsparse_float_complex *b, aoclsparse_float_complex *x, float rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const aoclsparse_float_complex *u, aoclsparse_float_complex *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const aoclsparse_float_complex *x, const aoclsparse_float_complex *r, float rinfo[100], void *udata),
    void *udata);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_z_init(aoclsparse_itsol_handle *handle);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_z_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n,
                                                          const aoclsparse_double_complex *b);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_z_rci_solve(aoclsparse_itsol_handle   handle,
                                                          aoclsparse_itsol_rci_job *ircomm, aoclsparse_double_complex **u,
                                                          aoclsparse_double_complex **v, aoclsparse_double_complex *x, double rinfo[100]);
DLL_PUBLIC aoclsparse_status aoclsparse_itsol_z_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const aoclsparse_double_complex *b, aoclsparse_double_complex *x, double rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const aoclsparse_double_complex *u, aoclsparse_double_complex *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const aoclsparse_double_complex *x, const aoclsparse_double_complex *r, double rinfo[100], void *udata),
    void *udata);

/* C = alpha*op(A)*B + beta*C, dense B/C in the stated order; alpha, beta BY VALUE (:2487-2511). */
DLL_PUBLIC aoclsparse_status aoclsparse_scsrmm(aoclsparse_operation       op,
                                               const float                alpha,
                                               const aoclsparse_matrix    A,
                                               const aoclsparse_mat_descr descr,
                                               aoclsparse_order           order,
                                               const float               *B,
                                               aoclsparse_int             n,
                                               aoclsparse_int             ldb,
                                               const float                beta,
                                               float                     *C,
                                               aoclsparse_int             ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsrmm(aoclsparse_operation       op,
                                               const double               alpha,
                                               const aoclsparse_matrix    A,
                                               const aoclsparse_mat_descr descr,
                                               aoclsparse_order           order,
                                               const double              *B,
                                               aoclsparse_int             n,
                                               aoclsparse_int             ldb,
                                               const double               beta,
                                               double                    *C,
                                               aoclsparse_int             ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_scsrmm_kid(aoclsparse_operation       op, /* :3426-3452 */
                                                   const float                alpha,
                                                   const aoclsparse_matrix    A,
                                                   const aoclsparse_mat_descr descr,
                                                   aoclsparse_order           order,
                                                   const float               *B,
                                                   aoclsparse_int             n,
                                                   aoclsparse_int             ldb,
                                                   const float                beta,
                                                   float                     *C,
                                                   aoclsparse_int             ldc,
                                                   const aoclsparse_int       kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsrmm_kid(aoclsparse_operation       op,
                                                   const double               alpha,
                                                   const aoclsparse_matrix    A,
                                                   const aoclsparse_mat_descr descr,
                                                   aoclsparse_order           order,
                                                   const double              *B,
                                                   aoclsparse_int             n,
                                                   aoclsparse_int             ldb,
                                                   const double               beta,
                                                   double                    *C,
                                                   aoclsparse_int             ldc,
                                                   const aoclsparse_int       kid);
/* C = op(A)*op(B), sparse result allocated by the library (:2202-2209, :2258-2261, :2795-2813). */
DLL_PUBLIC aoclsparse_status aoclsparse_sp2m(aoclsparse_operation       opA,
                                             const aoclsparse_mat_descr descrA,
                                             const aoclsparse_matrix    A,
                                             aoclsparse_operation       opB,
                                             const aoclsparse_mat_descr descrB,
                                             const aoclsparse_matrix    B,
                                             const aoclsparse_request   request,
                                             aoclsparse_matrix         *C);
DLL_PUBLIC aoclsparse_status aoclsparse_spmm(aoclsparse_operation    opA,
                                             const aoclsparse_matrix A,
                                             const aoclsparse_matrix B,
                                             aoclsparse_matrix      *C);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2m(aoclsparse_operation       trans_A,
                                               const aoclsparse_mat_descr descrA,
                                               const aoclsparse_matrix    csrA,
                                               aoclsparse_operation       trans_B,
                                               const aoclsparse_mat_descr descrB,
                                               const aoclsparse_matrix    csrB,
                                               const aoclsparse_request   request,
                                               aoclsparse_matrix         *csrC);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2m(aoclsparse_operation       trans_A,
                                               const aoclsparse_mat_descr descrA,
                                               const aoclsparse_matrix    csrA,
                                               aoclsparse_operation       trans_B,
                                               const aoclsparse_mat_descr descrB,
                                               const aoclsparse_matrix    csrB,
                                               const aoclsparse_request   request,
                                               aoclsparse_matrix         *csrC);

/* ---- sparse x sparse with a dense result, CSR -> dense, sparse sum ------------------------------------------
 * Replaces library/include/aoclsparse_functions.h:2546-2586 (?spmmd), :2674-2720 (?sp2md), :2856-2882 (?add) and
 * library/include/aoclsparse_convert.h:566-610 (?csr2dense).  C / A may be host or device memory. */
DLL_PUBLIC aoclsparse_status aoclsparse_ssp2md(const aoclsparse_operation opA, const aoclsparse_mat_descr descrA,
                                               const aoclsparse_matrix A, const aoclsparse_operation opB,
                                               const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                                               const float alpha, const float beta, float *C,
                                               const aoclsparse_order layout, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_dsp2md(const aoclsparse_operation opA, const aoclsparse_mat_descr descrA,
                                               const aoclsparse_matrix A, const aoclsparse_operation opB,
                                               const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                                               const double alpha, const double beta, double *C,
                                               const aoclsparse_order layout, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_csp2md(const aoclsparse_operation opA, const aoclsparse_mat_descr descrA,
                                               const aoclsparse_matrix A, const aoclsparse_operation opB,
                                               const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                                               const aoclsparse_float_complex alpha, const aoclsparse_float_complex beta, aoclsparse_float_complex *C,
                                               const aoclsparse_order layout, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_zsp2md(const aoclsparse_operation opA, const aoclsparse_mat_descr descrA,
                                               const aoclsparse_matrix A, const aoclsparse_operation opB,
                                               const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                                               const aoclsparse_double_complex alpha, const aoclsparse_double_complex beta, aoclsparse_double_complex *C,
                                               const aoclsparse_order layout, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_sspmmd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                               const aoclsparse_matrix B, const aoclsparse_order layout,
                                               float *C, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_dspmmd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                               const aoclsparse_matrix B, const aoclsparse_order layout,
                                               double *C, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_cspmmd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                               const aoclsparse_matrix B, const aoclsparse_order layout,
                                               aoclsparse_float_complex *C, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_zspmmd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                               const aoclsparse_matrix B, const aoclsparse_order layout,
                                               aoclsparse_double_complex *C, const aoclsparse_int ldc);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2dense(aoclsparse_int m, aoclsparse_int n,
                                                   const aoclsparse_mat_descr descr, const float *csr_val,
                                                   const aoclsparse_int *csr_row_ptr,
                                                   const aoclsparse_int *csr_col_ind, float *A,
                                                   aoclsparse_int ld, aoclsparse_order order);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2dense(aoclsparse_int m, aoclsparse_int n,
                                                   const aoclsparse_mat_descr descr, const double *csr_val,
                                                   const aoclsparse_int *csr_row_ptr,
                                                   const aoclsparse_int *csr_col_ind, double *A,
                                                   aoclsparse_int ld, aoclsparse_order order);
DLL_PUBLIC aoclsparse_status aoclsparse_ccsr2dense(aoclsparse_int m, aoclsparse_int n,
                                                   const aoclsparse_mat_descr descr, const aoclsparse_float_complex *csr_val,
                                                   const aoclsparse_int *csr_row_ptr,
                                                   const aoclsparse_int *csr_col_ind, aoclsparse_float_complex *A,
                                                   aoclsparse_int ld, aoclsparse_order order);
DLL_PUBLIC aoclsparse_status aoclsparse_zcsr2dense(aoclsparse_int m, aoclsparse_int n,
                                                   const aoclsparse_mat_descr descr, const aoclsparse_double_complex *csr_val,
                                                   const aoclsparse_int *csr_row_ptr,
                                                   const aoclsparse_int *csr_col_ind, aoclsparse_double_complex *A,
                                                   aoclsparse_int ld, aoclsparse_order order);
DLL_PUBLIC aoclsparse_status aoclsparse_sadd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                             const float alpha, const aoclsparse_matrix B,
                                             aoclsparse_matrix *C);
DLL_PUBLIC aoclsparse_status aoclsparse_dadd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                             const double alpha, const aoclsparse_matrix B,
                                             aoclsparse_matrix *C);
DLL_PUBLIC aoclsparse_status aoclsparse_cadd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                             const aoclsparse_float_complex alpha, const aoclsparse_matrix B,
                                             aoclsparse_matrix *C);
DLL_PUBLIC aoclsparse_status aoclsparse_zadd(const aoclsparse_operation op, const aoclsparse_matrix A,
                                             const aoclsparse_double_complex alpha, const aoclsparse_matrix B,
                                             aoclsparse_matrix *C);

/* ---- level 1: compressed sparse vector (x, indx) against a dense vector y -----------------------------------------
 * Replaces library/include/aoclsparse_functions.h:83-99 (?axpyi), :141-151 / :190-200 (?dotci / ?dotui), :236-246
 * (?doti), :295-312 (?sctr), :344-357 (?sctrs), :416-430 (?roti), :492-505 (?gthr), :558-572 (?gthrz), :612-626 (?gthrs)
 * and the _kid twins at :3220-3395.  Vectors may be host or device memory. */
DLL_PUBLIC aoclsparse_status aoclsparse_saxpyi(const aoclsparse_int nnz, const float a, const float *x, const aoclsparse_int *indx, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_daxpyi(const aoclsparse_int nnz, const double a, const double *x, const aoclsparse_int *indx, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_caxpyi(const aoclsparse_int nnz, const void *a, const void *x, const aoclsparse_int *indx, void *y);
DLL_PUBLIC aoclsparse_status aoclsparse_zaxpyi(const aoclsparse_int nnz, const void *a, const void *x, const aoclsparse_int *indx, void *y);
DLL_PUBLIC float aoclsparse_sdoti(const aoclsparse_int nnz, const float *x, const aoclsparse_int *indx, const float *y);
DLL_PUBLIC double aoclsparse_ddoti(const aoclsparse_int nnz, const double *x, const aoclsparse_int *indx, const double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_cdotci(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot);
DLL_PUBLIC aoclsparse_status aoclsparse_cdotui(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot);
DLL_PUBLIC aoclsparse_status aoclsparse_zdotci(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot);
DLL_PUBLIC aoclsparse_status aoclsparse_zdotui(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot);
DLL_PUBLIC aoclsparse_status aoclsparse_sgthr(aoclsparse_int nnz, const float *y, float *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_dgthr(aoclsparse_int nnz, const double *y, double *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_cgthr(aoclsparse_int nnz, const void *y, void *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_zgthr(aoclsparse_int nnz, const void *y, void *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_sgthrz(aoclsparse_int nnz, float *y, float *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_dgthrz(aoclsparse_int nnz, double *y, double *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_cgthrz(aoclsparse_int nnz, void *y, void *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_zgthrz(aoclsparse_int nnz, void *y, void *x, const aoclsparse_int *indx);
DLL_PUBLIC aoclsparse_status aoclsparse_sgthrs(aoclsparse_int nnz, const float *y, float *x, aoclsparse_int stride);
DLL_PUBLIC aoclsparse_status aoclsparse_dgthrs(aoclsparse_int nnz, const double *y, double *x, aoclsparse_int stride);
DLL_PUBLIC aoclsparse_status aoclsparse_cgthrs(aoclsparse_int nnz, const void *y, void *x, aoclsparse_int stride);
DLL_PUBLIC aoclsparse_status aoclsparse_zgthrs(aoclsparse_int nnz, const void *y, void *x, aoclsparse_int stride);
DLL_PUBLIC aoclsparse_status aoclsparse_ssctr(const aoclsparse_int nnz, const float *x, const aoclsparse_int *indx, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dsctr(const aoclsparse_int nnz, const double *x, const aoclsparse_int *indx, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_csctr(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, void *y);
DLL_PUBLIC aoclsparse_status aoclsparse_zsctr(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, void *y);
DLL_PUBLIC aoclsparse_status aoclsparse_ssctrs(const aoclsparse_int nnz, const float *x, aoclsparse_int stride, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dsctrs(const aoclsparse_int nnz, const double *x, aoclsparse_int stride, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_csctrs(const aoclsparse_int nnz, const void *x, aoclsparse_int stride, void *y);
DLL_PUBLIC aoclsparse_status aoclsparse_zsctrs(const aoclsparse_int nnz, const void *x, aoclsparse_int stride, void *y);
DLL_PUBLIC aoclsparse_status aoclsparse_sroti(const aoclsparse_int nnz, float *x, const aoclsparse_int *indx, float *y, const float c, const float s);
DLL_PUBLIC aoclsparse_status aoclsparse_droti(const aoclsparse_int nnz, double *x, const aoclsparse_int *indx, double *y, const double c, const double s);
DLL_PUBLIC aoclsparse_status aoclsparse_saxpyi_kid(const aoclsparse_int nnz, const float a, const float *x, const aoclsparse_int *indx, float *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_daxpyi_kid(const aoclsparse_int nnz, const double a, const double *x, const aoclsparse_int *indx, double *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_caxpyi_kid(const aoclsparse_int nnz, const void *a, const void *x, const aoclsparse_int *indx, void *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zaxpyi_kid(const aoclsparse_int nnz, const void *a, const void *x, const aoclsparse_int *indx, void *y, aoclsparse_int kid);
DLL_PUBLIC float aoclsparse_sdoti_kid(const aoclsparse_int nnz, const float *x, const aoclsparse_int *indx, const float *y, aoclsparse_int kid);
DLL_PUBLIC double aoclsparse_ddoti_kid(const aoclsparse_int nnz, const double *x, const aoclsparse_int *indx, const double *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_cdotci_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_cdotui_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zdotci_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zdotui_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, const void *y, void *dot, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_sgthr_kid(aoclsparse_int nnz, const float *y, float *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dgthr_kid(aoclsparse_int nnz, const double *y, double *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_cgthr_kid(aoclsparse_int nnz, const void *y, void *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zgthr_kid(aoclsparse_int nnz, const void *y, void *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_sgthrz_kid(aoclsparse_int nnz, float *y, float *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dgthrz_kid(aoclsparse_int nnz, double *y, double *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_cgthrz_kid(aoclsparse_int nnz, void *y, void *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zgthrz_kid(aoclsparse_int nnz, void *y, void *x, const aoclsparse_int *indx, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_sgthrs_kid(aoclsparse_int nnz, const float *y, float *x, aoclsparse_int stride, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dgthrs_kid(aoclsparse_int nnz, const double *y, double *x, aoclsparse_int stride, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_cgthrs_kid(aoclsparse_int nnz, const void *y, void *x, aoclsparse_int stride, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zgthrs_kid(aoclsparse_int nnz, const void *y, void *x, aoclsparse_int stride, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_ssctr_kid(const aoclsparse_int nnz, const float *x, const aoclsparse_int *indx, float *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dsctr_kid(const aoclsparse_int nnz, const double *x, const aoclsparse_int *indx, double *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_csctr_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, void *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zsctr_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, void *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_ssctrs_kid(const aoclsparse_int nnz, const float *x, aoclsparse_int stride, float *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dsctrs_kid(const aoclsparse_int nnz, const double *x, aoclsparse_int stride, double *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_csctrs_kid(const aoclsparse_int nnz, const void *x, aoclsparse_int stride, void *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_zsctrs_kid(const aoclsparse_int nnz, const void *x, aoclsparse_int stride, void *y, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_sroti_kid(const aoclsparse_int nnz, float *x, const aoclsparse_int *indx, float *y, const float c, const float s, aoclsparse_int kid);
DLL_PUBLIC aoclsparse_status aoclsparse_droti_kid(const aoclsparse_int nnz, double *x, const aoclsparse_int *indx, double *y, const double c, const double s, aoclsparse_int kid);

/* ---- DIA and BSR raw-array formats -----------------------------------------------------------------------------
 * Replaces library/include/aoclsparse_convert.h:215-221 (csr2dia_ndiag), :266-285 (?csr2dia), :324-331 (csr2bsr_nnz),
 * :380-430 (?csr2bsr) and library/include/aoclsparse_functions.h:1040-1106 (?diamv, ?diamv_kid), ?bsrmv.
 * Conversions work on host arrays; the products accept host or device arrays. */
DLL_PUBLIC aoclsparse_status aoclsparse_csr2dia_ndiag(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                                      aoclsparse_int nnz, const aoclsparse_int *csr_row_ptr,
                                                      const aoclsparse_int *csr_col_ind, aoclsparse_int *dia_num_diag);
DLL_PUBLIC aoclsparse_status aoclsparse_csr2bsr_nnz(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                                    const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                                    aoclsparse_int block_dim, aoclsparse_int *bsr_row_ptr,
                                                    aoclsparse_int *bsr_nnz);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2dia(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                               const float *csr_val, aoclsparse_int dia_num_diag,
                                               aoclsparse_int *dia_offset, float *dia_val);
DLL_PUBLIC aoclsparse_status aoclsparse_sdiamv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                             aoclsparse_int n, aoclsparse_int nnz, const float *dia_val,
                                             const aoclsparse_int *dia_offset, aoclsparse_int dia_num_diag,
                                             const aoclsparse_mat_descr descr, const float *x, const float *beta, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_sdiamv_kid(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                                 aoclsparse_int n, aoclsparse_int nnz, const float *dia_val,
                                                 const aoclsparse_int *dia_offset, aoclsparse_int dia_num_diag,
                                                 const aoclsparse_mat_descr descr, const float *x, const float *beta,
                                                 float *y, aoclsparse_int diamv_mode, aoclsparse_int diamv_kid);
DLL_PUBLIC aoclsparse_status aoclsparse_sbsrmv(aoclsparse_operation trans, const float *alpha, aoclsparse_int mb,
                                             aoclsparse_int nb, aoclsparse_int bsr_dim, const float *bsr_val,
                                             const aoclsparse_int *bsr_col_ind, const aoclsparse_int *bsr_row_ptr,
                                             const aoclsparse_mat_descr descr, const float *x, const float *beta, float *y);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2dia(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                               const double *csr_val, aoclsparse_int dia_num_diag,
                                               aoclsparse_int *dia_offset, double *dia_val);
DLL_PUBLIC aoclsparse_status aoclsparse_ddiamv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                             aoclsparse_int n, aoclsparse_int nnz, const double *dia_val,
                                             const aoclsparse_int *dia_offset, aoclsparse_int dia_num_diag,
                                             const aoclsparse_mat_descr descr, const double *x, const double *beta, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_ddiamv_kid(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                                 aoclsparse_int n, aoclsparse_int nnz, const double *dia_val,
                                                 const aoclsparse_int *dia_offset, aoclsparse_int dia_num_diag,
                                                 const aoclsparse_mat_descr descr, const double *x, const double *beta,
                                                 double *y, aoclsparse_int diamv_mode, aoclsparse_int diamv_kid);
DLL_PUBLIC aoclsparse_status aoclsparse_dbsrmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int mb,
                                             aoclsparse_int nb, aoclsparse_int bsr_dim, const double *bsr_val,
                                             const aoclsparse_int *bsr_col_ind, const aoclsparse_int *bsr_row_ptr,
                                             const aoclsparse_mat_descr descr, const double *x, const double *beta, double *y);
DLL_PUBLIC aoclsparse_status aoclsparse_scsr2bsr(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                               const aoclsparse_order block_order, const float *csr_val,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                               aoclsparse_int block_dim, float *bsr_val, aoclsparse_int *bsr_row_ptr,
                                               aoclsparse_int *bsr_col_ind);
DLL_PUBLIC aoclsparse_status aoclsparse_dcsr2bsr(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                               const aoclsparse_order block_order, const double *csr_val,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                               aoclsparse_int block_dim, double *bsr_val, aoclsparse_int *bsr_row_ptr,
                                               aoclsparse_int *bsr_col_ind);
DLL_PUBLIC aoclsparse_status aoclsparse_ccsr2bsr(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                               const aoclsparse_order block_order, const aoclsparse_float_complex *csr_val,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                               aoclsparse_int block_dim, aoclsparse_float_complex *bsr_val, aoclsparse_int *bsr_row_ptr,
                                               aoclsparse_int *bsr_col_ind);
DLL_PUBLIC aoclsparse_status aoclsparse_zcsr2bsr(aoclsparse_int m, aoclsparse_int n, const aoclsparse_mat_descr descr,
                                               const aoclsparse_order block_order, const aoclsparse_double_complex *csr_val,
                                               const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                               aoclsparse_int block_dim, aoclsparse_double_complex *bsr_val, aoclsparse_int *bsr_row_ptr,
                                               aoclsparse_int *bsr_col_ind);

/* ---- forward SOR sweep: replaces library/include/aoclsparse_solvers.h:640-682 (backward / symmetric and the complex
 * types return not_implemented, as in the reference) */
DLL_PUBLIC aoclsparse_status aoclsparse_ssorv(aoclsparse_sor_type sor_type, const aoclsparse_mat_descr descr,
                                            const aoclsparse_matrix A, float omega, float alpha,
                                            float *x, const float *b);
DLL_PUBLIC aoclsparse_status aoclsparse_dsorv(aoclsparse_sor_type sor_type, const aoclsparse_mat_descr descr,
                                            const aoclsparse_matrix A, double omega, double alpha,
                                            double *x, const double *b);
DLL_PUBLIC aoclsparse_status aoclsparse_csorv(aoclsparse_sor_type sor_type, const aoclsparse_mat_descr descr,
                                            const aoclsparse_matrix A, aoclsparse_float_complex omega, aoclsparse_float_complex alpha,
                                            aoclsparse_float_complex *x, const aoclsparse_float_complex *b);
DLL_PUBLIC aoclsparse_status aoclsparse_zsorv(aoclsparse_sor_type sor_type, const aoclsparse_mat_descr descr,
                                            const aoclsparse_matrix A, aoclsparse_double_complex omega, aoclsparse_double_complex alpha,
                                            aoclsparse_double_complex *x, const aoclsparse_double_complex *b);

#ifdef __cplusplus
}
#endif
#endif /* AOCLSPARSE_H_ */
