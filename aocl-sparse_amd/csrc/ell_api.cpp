// ell_api.cpp -- the stored formats the reference's optimize step produces, as public raw-array entry
// points: aoclsparse_?ellmv, ?elltmv, ?ellthybmv and the CSR -> ELL / ELLT / ELLT-HYB conversions.
//
//   checks      : level2/aoclsparse_ellmv.hpp:211-310 (ell), :447-546 (ellt), :759-832 (ellthyb: none)
//   conversions : conversion/aoclsparse_convert.cpp:311-412, conversion/aoclsparse_convert.hpp:41-290
//
// The conversions are host routines in the reference and here (they produce host arrays the caller owns);
// the products run on the GPU, on device arrays directly or on host arrays staged for the call.
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

template <typename T>
aoclsparse_status ell_checks(aoclsparse_operation trans, aoclsparse_int m, aoclsparse_int n, const T *ell_val,
                             const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                             const aoclsparse_mat_descr descr, const T *x, T *y, bool &nothing)
{
    nothing = false;
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != aoclsparse_index_base_zero && descr->base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_general)
        return aoclsparse_status_not_implemented;
    if(trans != aoclsparse_operation_none)
        return aoclsparse_status_not_implemented;
    if(m < 0 || n < 0 || ell_width < 0)
        return aoclsparse_status_invalid_size;
    if((m == 0 || n == 0) && ell_width != 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0 || ell_width == 0)
    {
        nothing = true; // quick return: y is left as it is (:268-272)
        return aoclsparse_status_success;
    }
    if(!ell_val || !ell_col_ind || !x || !y)
        return aoclsparse_status_invalid_pointer;
    return aoclsparse_status_success;
}

template <typename T, bool TRANSPOSED_LAYOUT>
aoclsparse_status ellmv_t(aoclsparse_operation trans, const T *alpha, aoclsparse_int m, aoclsparse_int n,
                          const T *ell_val, const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                          const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y)
{
    bool nothing;
    MI355_TRY(ell_checks(trans, m, n, ell_val, ell_col_ind, ell_width, descr, x, y, nothing));
    if(nothing)
        return aoclsparse_status_success;
    if(!alpha || !beta) // the reference dereferences them unchecked; refuse instead of crashing
        return aoclsparse_status_invalid_pointer;
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    const size_t                          cells = (size_t)m * (size_t)ell_width;
    StagedArg                             av, ac, ax, ay;
    MI355_TRY(av.in(rt, 8, ell_val, sizeof(T) * cells, true));
    MI355_TRY(ac.in(rt, 9, ell_col_ind, sizeof(aoclsparse_int) * cells, true));
    MI355_TRY(ax.in(rt, 3, x, sizeof(T) * (size_t)n, true));
    MI355_TRY(ay.in(rt, 4, y, sizeof(T) * (size_t)m, *beta != T(0)));
    if(TRANSPOSED_LAYOUT)
        MI355_TRY(launch_elltmv<T>(rt.stream(), descr->base, *alpha, m, static_cast<const T *>(av.dev),
                                   static_cast<const aoclsparse_int *>(ac.dev), ell_width,
                                   static_cast<const T *>(ax.dev), *beta, static_cast<T *>(ay.dev)));
    else
        MI355_TRY(launch_ellmv<T>(rt.stream(), descr->base, *alpha, m, static_cast<const T *>(av.dev),
                                  static_cast<const aoclsparse_int *>(ac.dev), ell_width,
                                  static_cast<const T *>(ax.dev), *beta, static_cast<T *>(ay.dev)));
    MI355_TRY(ay.out(rt));
    if(ay.staged)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return aoclsparse_status_success;
}

aoclsparse_status ellthybmv_d(const double *alpha, aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                              const double *ell_val, const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                              aoclsparse_int ell_m, const double *csr_val, const aoclsparse_int *csr_row_ind,
                              const aoclsparse_int *csr_col_ind, const aoclsparse_int *csr_row_idx_map,
                              const aoclsparse_mat_descr descr, const double *x, const double *beta, double *y)
{
    // the reference performs no checks here (ellmv.hpp:759-832); keep the ones that stop a crash
    if(!alpha || !beta || !descr || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(m < 0 || n < 0 || nnz < 0 || ell_width < 0 || ell_m < 0 || ell_m > m)
        return aoclsparse_status_invalid_size;
    if(m == 0 || n == 0)
        return aoclsparse_status_success;
    if((ell_width > 0 && (!ell_val || !ell_col_ind)) || (ell_m < m && (!csr_val || !csr_row_ind || !csr_col_ind || !csr_row_idx_map)))
        return aoclsparse_status_invalid_pointer;
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    const aoclsparse_int                  nlong = m - ell_m;
    const size_t                          cells = (size_t)m * (size_t)ell_width;
    StagedArg                             av, ac, ax, ay, cv, cr, cc, mp;
    MI355_TRY(av.in(rt, 8, ell_val, sizeof(double) * cells, true));
    MI355_TRY(ac.in(rt, 9, ell_col_ind, sizeof(aoclsparse_int) * cells, true));
    MI355_TRY(ax.in(rt, 3, x, sizeof(double) * (size_t)n, true));
    MI355_TRY(ay.in(rt, 4, y, sizeof(double) * (size_t)m, *beta != 0.0));
    void *ytmp = nullptr;
    if(nlong > 0)
    {
        MI355_TRY(cv.in(rt, 10, csr_val, sizeof(double) * (size_t)nnz, true));
        MI355_TRY(cc.in(rt, 11, csr_col_ind, sizeof(aoclsparse_int) * (size_t)nnz, true));
        MI355_TRY(cr.in(rt, 12, csr_row_ind, sizeof(aoclsparse_int) * ((size_t)m + 1), true));
        MI355_TRY(mp.in(rt, 13, csr_row_idx_map, sizeof(aoclsparse_int) * (size_t)nlong, true));
        MI355_TRY(rt.staging(14, sizeof(double) * (size_t)nlong, &ytmp));
        if(*beta != 0.0) // the long rows' y survives the ELL pass in a side buffer (:585-601, :652-660)
            MI355_TRY(launch_gather_rows<double>(rt.stream(), nlong, static_cast<const aoclsparse_int *>(mp.dev),
                                                 static_cast<const double *>(ay.dev), static_cast<double *>(ytmp)));
    }
    if(ell_width > 0)
        MI355_TRY(launch_elltmv<double>(rt.stream(), descr->base, *alpha, m, static_cast<const double *>(av.dev),
                                        static_cast<const aoclsparse_int *>(ac.dev), ell_width,
                                        static_cast<const double *>(ax.dev), *beta, static_cast<double *>(ay.dev)));
    else if(ell_m == m) // width 0 and no long rows: the ELL pass alone defines y
        MI355_TRY(launch_scale<double>(rt.stream(), static_cast<double *>(ay.dev), m, *beta));
    if(nlong > 0)
        MI355_TRY(launch_csr_rows<double>(rt.stream(), descr->base, *alpha, nlong,
                                          static_cast<const aoclsparse_int *>(mp.dev),
                                          static_cast<const double *>(cv.dev),
                                          static_cast<const aoclsparse_int *>(cc.dev),
                                          static_cast<const aoclsparse_int *>(cr.dev),
                                          static_cast<const double *>(ax.dev), *beta,
                                          static_cast<const double *>(ytmp), static_cast<double *>(ay.dev)));
    MI355_TRY(ay.out(rt));
    if(ay.staged)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return aoclsparse_status_success;
}

// ---- conversions (host) --------------------------------------------------------------------------------
template <typename T>
aoclsparse_status csr2ell(aoclsparse_int m, const aoclsparse_mat_descr descr, const aoclsparse_int *csr_row_ptr,
                          const aoclsparse_int *csr_col_ind, const T *csr_val, aoclsparse_int *ell_col_ind,
                          T *ell_val, aoclsparse_int ell_width, bool transposed)
{
    // convert.hpp:50-79 / :119-148
    if(m < 0 || ell_width < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || ell_width == 0)
        return aoclsparse_status_success;
    if(!csr_val || !csr_row_ptr || !csr_col_ind || !ell_val || !ell_col_ind)
        return aoclsparse_status_invalid_pointer;
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    const aoclsparse_int base = descr->base;
    for(aoclsparse_int i = 0; i < m; i++)
    {
        const aoclsparse_int s = csr_row_ptr[i] - base, e = csr_row_ptr[i + 1] - base;
        aoclsparse_int       k = 0;
        if(!transposed)
        {
            const size_t o = (size_t)i * (size_t)ell_width;
            for(aoclsparse_int j = s; j < e && k < ell_width; j++, k++)
                ell_col_ind[o + k] = csr_col_ind[j], ell_val[o + k] = csr_val[j]; // index base is kept (:90-92)
            for(; k < ell_width; k++)
                ell_col_ind[o + k] = -1, ell_val[o + k] = T(0);
        }
        else
        {
            for(aoclsparse_int j = s; j < e && k < ell_width; j++, k++)
                ell_col_ind[(size_t)k * m + i] = csr_col_ind[j], ell_val[(size_t)k * m + i] = csr_val[j];
            // padding repeats the row's last column so that the gather stays in bounds (:166-171); an empty
            // row has no such column (the reference reads csr_col_ind[row_end - 1] regardless): use `base`
            const aoclsparse_int pad = e > s ? csr_col_ind[e - 1] : base;
            for(; k < ell_width; k++)
                ell_col_ind[(size_t)k * m + i] = pad, ell_val[(size_t)k * m + i] = T(0);
        }
    }
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status csr2ellthyb(aoclsparse_int m, aoclsparse_index_base base, aoclsparse_int *ell_m,
                              const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind, const T *csr_val,
                              aoclsparse_int *csr_row_idx_map, aoclsparse_int *ell_col_ind, T *ell_val,
                              aoclsparse_int ell_width)
{
    // convert.hpp:190-289
    if(m < 0 || ell_width < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0)
        return aoclsparse_status_success;
    if(!csr_val || !csr_row_ptr || !csr_col_ind || !ell_val || !ell_col_ind || !csr_row_idx_map || !ell_m)
        return aoclsparse_status_invalid_pointer;
    aoclsparse_int in_ell = 0, nlong = 0;
    for(aoclsparse_int i = 0; i < m; i++)
    {
        const aoclsparse_int s = csr_row_ptr[i] - base, e = csr_row_ptr[i + 1] - base;
        const aoclsparse_int pad = e > s ? csr_col_ind[e - 1] : (aoclsparse_int)base;
        aoclsparse_int       k   = 0;
        if(e - s > ell_width)
            csr_row_idx_map[nlong++] = i; // 0-based whatever the index base (:243-247); its ELL cells are padding
        else
        {
            in_ell++;
            for(aoclsparse_int j = s; j < e; j++, k++)
                ell_col_ind[(size_t)k * m + i] = csr_col_ind[j], ell_val[(size_t)k * m + i] = csr_val[j];
        }
        for(; k < ell_width; k++)
            ell_col_ind[(size_t)k * m + i] = pad, ell_val[(size_t)k * m + i] = T(0);
    }
    *ell_m = in_ell;
    return aoclsparse_status_success;
}

} // namespace

extern "C" {

aoclsparse_status aoclsparse_dellmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                    aoclsparse_int n, aoclsparse_int nnz, const double *ell_val,
                                    const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                    const aoclsparse_mat_descr descr, const double *x, const double *beta, double *y)
{
    (void)nnz;
    return ellmv_t<double, false>(trans, alpha, m, n, ell_val, ell_col_ind, ell_width, descr, x, beta, y);
}
aoclsparse_status aoclsparse_sellmv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                    aoclsparse_int n, aoclsparse_int nnz, const float *ell_val,
                                    const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                    const aoclsparse_mat_descr descr, const float *x, const float *beta, float *y)
{
    (void)nnz;
    return ellmv_t<float, false>(trans, alpha, m, n, ell_val, ell_col_ind, ell_width, descr, x, beta, y);
}
aoclsparse_status aoclsparse_delltmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                     aoclsparse_int n, aoclsparse_int nnz, const double *ell_val,
                                     const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                     const aoclsparse_mat_descr descr, const double *x, const double *beta, double *y)
{
    (void)nnz;
    return ellmv_t<double, true>(trans, alpha, m, n, ell_val, ell_col_ind, ell_width, descr, x, beta, y);
}
aoclsparse_status aoclsparse_selltmv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                     aoclsparse_int n, aoclsparse_int nnz, const float *ell_val,
                                     const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                     const aoclsparse_mat_descr descr, const float *x, const float *beta, float *y)
{
    (void)nnz;
    return ellmv_t<float, true>(trans, alpha, m, n, ell_val, ell_col_ind, ell_width, descr, x, beta, y);
}
aoclsparse_status aoclsparse_dellthybmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                        aoclsparse_int n, aoclsparse_int nnz, const double *ell_val,
                                        const aoclsparse_int *ell_col_ind, aoclsparse_int ell_width,
                                        const aoclsparse_int ell_m, const double *csr_val,
                                        const aoclsparse_int *csr_row_ind, const aoclsparse_int *csr_col_ind,
                                        aoclsparse_int *row_idx_map, aoclsparse_int *csr_row_idx_map,
                                        const aoclsparse_mat_descr descr, const double *x, const double *beta,
                                        double *y)
{
    (void)trans, (void)row_idx_map; // both unused by the reference as well (:759, :567)
    return ellthybmv_d(alpha, m, n, nnz, ell_val, ell_col_ind, ell_width, ell_m, csr_val, csr_row_ind, csr_col_ind,
                       csr_row_idx_map, descr, x, beta, y);
}
aoclsparse_status aoclsparse_sellthybmv(aoclsparse_operation, const float *, aoclsparse_int, aoclsparse_int,
                                        aoclsparse_int, const float *, const aoclsparse_int *, aoclsparse_int,
                                        const aoclsparse_int, const float *, const aoclsparse_int *,
                                        const aoclsparse_int *, aoclsparse_int *, aoclsparse_int *,
                                        const aoclsparse_mat_descr, const float *, const float *, float *)
{
    return aoclsparse_status_not_implemented; // :781-786: only double exists
}

aoclsparse_status aoclsparse_csr2ell_width(aoclsparse_int m, aoclsparse_int nnz, const aoclsparse_int *csr_row_ptr,
                                           aoclsparse_int *ell_width)
{
    (void)nnz;
    if(m < 0)
        return aoclsparse_status_invalid_size;
    if(!ell_width || !csr_row_ptr)
        return aoclsparse_status_invalid_pointer;
    aoclsparse_int w = 0;
    for(aoclsparse_int i = 0; i < m; i++)
        w = std::max(w, csr_row_ptr[i + 1] - csr_row_ptr[i]);
    *ell_width = w;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_csr2ellthyb_width(aoclsparse_int m, aoclsparse_int nnz, const aoclsparse_int *csr_row_ptr,
                                               aoclsparse_int *ell_m, aoclsparse_int *ell_width)
{
    // convert.cpp:344-412: the width is the row length next to the average on the side where most rows are
    if(m < 0)
        return aoclsparse_status_invalid_size;
    if(!ell_width || !ell_m)
        return aoclsparse_status_invalid_pointer;
    if(m == 0)
    {
        *ell_width = 0, *ell_m = 0;
        return aoclsparse_status_success;
    }
    if(!csr_row_ptr)
        return aoclsparse_status_invalid_pointer;
    const aoclsparse_int avg = nnz / m;
    aoclsparse_int       below = 0, above = nnz, n_below = 0, n_above = 0;
    for(aoclsparse_int i = 0; i < m; i++)
    {
        const aoclsparse_int len = csr_row_ptr[i + 1] - csr_row_ptr[i];
        if(len <= avg)
            below = std::max(below, len), n_below++;
        else
            above = std::min(above, len), n_above++;
    }
    *ell_width = n_below >= n_above ? below : above;
    aoclsparse_int fit = 0;
    for(aoclsparse_int i = 0; i < m; i++)
        fit += (csr_row_ptr[i + 1] - csr_row_ptr[i] <= *ell_width) ? 1 : 0;
    *ell_m = fit;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_dcsr2ell(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                      const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                      const double *csr_val, aoclsparse_int *ell_col_ind, double *ell_val,
                                      aoclsparse_int ell_width)
{
    return csr2ell(m, descr, csr_row_ptr, csr_col_ind, csr_val, ell_col_ind, ell_val, ell_width, false);
}
aoclsparse_status aoclsparse_scsr2ell(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                      const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                      const float *csr_val, aoclsparse_int *ell_col_ind, float *ell_val,
                                      aoclsparse_int ell_width)
{
    return csr2ell(m, descr, csr_row_ptr, csr_col_ind, csr_val, ell_col_ind, ell_val, ell_width, false);
}
aoclsparse_status aoclsparse_dcsr2ellt(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                       const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                       const double *csr_val, aoclsparse_int *ell_col_ind, double *ell_val,
                                       aoclsparse_int ell_width)
{
    return csr2ell(m, descr, csr_row_ptr, csr_col_ind, csr_val, ell_col_ind, ell_val, ell_width, true);
}
aoclsparse_status aoclsparse_scsr2ellt(aoclsparse_int m, const aoclsparse_mat_descr descr,
                                       const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                       const float *csr_val, aoclsparse_int *ell_col_ind, float *ell_val,
                                       aoclsparse_int ell_width)
{
    return csr2ell(m, descr, csr_row_ptr, csr_col_ind, csr_val, ell_col_ind, ell_val, ell_width, true);
}
aoclsparse_status aoclsparse_dcsr2ellthyb(aoclsparse_int m, aoclsparse_index_base base, aoclsparse_int *ell_m,
                                          const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                          const double *csr_val, aoclsparse_int *row_idx_map,
                                          aoclsparse_int *csr_row_idx_map, aoclsparse_int *ell_col_ind,
                                          double *ell_val, aoclsparse_int ell_width)
{
    (void)row_idx_map;
    return csr2ellthyb(m, base, ell_m, csr_row_ptr, csr_col_ind, csr_val, csr_row_idx_map, ell_col_ind, ell_val,
                       ell_width);
}
aoclsparse_status aoclsparse_scsr2ellthyb(aoclsparse_int m, aoclsparse_index_base base, aoclsparse_int *ell_m,
                                          const aoclsparse_int *csr_row_ptr, const aoclsparse_int *csr_col_ind,
                                          const float *csr_val, aoclsparse_int *row_idx_map,
                                          aoclsparse_int *csr_row_idx_map, aoclsparse_int *ell_col_ind,
                                          float *ell_val, aoclsparse_int ell_width)
{
    (void)row_idx_map;
    return csr2ellthyb(m, base, ell_m, csr_row_ptr, csr_col_ind, csr_val, csr_row_idx_map, ell_col_ind, ell_val,
                       ell_width);
}

} // extern "C"
