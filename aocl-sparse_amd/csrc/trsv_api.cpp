// trsv_api.cpp -- aoclsparse_?trsv(_kid)(_strided): checks, level-set analysis, dispatch.
//
// Argument checks and their order: level2/aoclsparse_trsv.cpp:59-137 of the reference.  The solve
// itself is the level-scheduled HIP path of trsv_kernels.hip; the analysis (level sets of the
// hinted triangle) runs once per (fill, op) at aoclsparse_optimize after aoclsparse_set_sv_hint,
// or lazily on the first solve, mirroring the reference's lazy aoclsparse_csr_csc_optimize (:128).
#include "internal.hpp"

#include <algorithm>
#include <cstring>

using namespace mi355;

namespace mi355
{

// level[i] = 1 + max level of the rows row i depends on; rows bucketed by level (counting sort,
// ascending row index inside a level).
static aoclsparse_status level_sets(aoclsparse_int m, const aoclsparse_int *rs, const aoclsparse_int *re,
                                    const aoclsparse_int *ind, int base, bool descending, TrsvPlan &plan)
{
    std::vector<aoclsparse_int> level((size_t)m, 0);
    aoclsparse_int              nlev = 0;
    for(aoclsparse_int t = 0; t < m; t++)
    {
        const aoclsparse_int i  = descending ? m - 1 - t : t;
        aoclsparse_int       lv = 0;
        for(aoclsparse_int p = rs[i] - base; p < re[i] - base; p++)
            lv = std::max(lv, level[ind[p] - base] + 1);
        level[i] = lv;
        nlev     = std::max(nlev, lv + 1);
    }
    plan.level_ptr.assign((size_t)nlev + 1, 0);
    for(aoclsparse_int i = 0; i < m; i++)
        plan.level_ptr[level[i] + 1]++;
    plan.max_width = 0;
    for(aoclsparse_int l = 0; l < nlev; l++)
    {
        plan.max_width = std::max(plan.max_width, plan.level_ptr[l + 1]);
        plan.level_ptr[l + 1] += plan.level_ptr[l];
    }
    std::vector<aoclsparse_int> next(plan.level_ptr.begin(), plan.level_ptr.end() - 1);
    std::vector<aoclsparse_int> rowmap((size_t)m);
    for(aoclsparse_int i = 0; i < m; i++)
        rowmap[next[level[i]]++] = i;
    plan.nlevels = nlev;
    return plan.rowmap.upload(rowmap.data(), sizeof(aoclsparse_int) * (size_t)m, Runtime::get().stream());
}

template <typename T>
static aoclsparse_status build_transposed_triangle(const HostCsr &c, bool upper, TrsvPlan &plan,
                                                   std::vector<aoclsparse_int> &tptr,
                                                   std::vector<aoclsparse_int> &tind)
{
    // strict triangle of the clean CSR, transposed by counting sort (stable: ascending source row)
    const aoclsparse_int  m = c.m, b = c.base;
    const aoclsparse_int *s = upper ? c.iurow : c.ptr; // positions in base b
    const aoclsparse_int *e = upper ? c.ptr + 1 : c.idiag;
    const T              *v = static_cast<const T *>(c.val);
    tptr.assign((size_t)m + 1, 0);
    for(aoclsparse_int i = 0; i < m; i++)
        for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
            tptr[c.ind[p] - b + 1]++;
    for(aoclsparse_int j = 0; j < m; j++)
        tptr[j + 1] += tptr[j];
    const aoclsparse_int tnnz = tptr[m];
    tind.assign((size_t)std::max(tnnz, 1), 0);
    std::vector<T>              tval((size_t)std::max(tnnz, 1));
    std::vector<aoclsparse_int> next(tptr.begin(), tptr.end() - 1);
    for(aoclsparse_int i = 0; i < m; i++)
        for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
        {
            const aoclsparse_int q = next[c.ind[p] - b]++;
            tind[q]                = i;
            tval[q]                = v[p];
        }
    hipStream_t       st = Runtime::get().stream();
    aoclsparse_status rc = plan.own_ptr.upload(tptr.data(), sizeof(aoclsparse_int) * ((size_t)m + 1), st);
    if(rc == aoclsparse_status_success)
        rc = plan.own_ind.upload(tind.data(), sizeof(aoclsparse_int) * (size_t)tnnz, st);
    if(rc == aoclsparse_status_success)
        rc = plan.own_val.upload(tval.data(), sizeof(T) * (size_t)tnnz, st);
    return rc;
}

aoclsparse_status ensure_trsv(aoclsparse_matrix A, bool upper, bool transposed)
{
    aoclsparse_status st = csr_optimize(A);
    if(st != aoclsparse_status_success)
        return st;
    TrsvPlan &plan = A->trsv_plan[(upper ? 2 : 0) + (transposed ? 1 : 0)];
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if(plan.valid)
            return aoclsparse_status_success;
    }
    std::unique_lock<std::shared_mutex> w(A->guard);
    if(plan.valid)
        return aoclsparse_status_success;
    const HostCsr &c  = *A->opt;
    const size_t   vs = val_size(A->val_type);
    const aoclsparse_int m = c.m;
    Runtime       &rt = Runtime::get();
    try
    {
        if(!A->dev_opt.valid)
        {
            st = upload_csr(c, vs, A->dev_opt);
            if(st == aoclsparse_status_success)
                st = A->dev_opt_idiag.upload(c.idiag, sizeof(aoclsparse_int) * (size_t)m, rt.stream());
            if(st == aoclsparse_status_success)
                st = A->dev_opt_iurow.upload(c.iurow, sizeof(aoclsparse_int) * (size_t)m, rt.stream());
            if(st != aoclsparse_status_success)
                return st;
            // diagonal values (only read for non-unit solves, which require a full diagonal)
            std::vector<char> dv(vs * (size_t)std::max(m, 1), 0);
            for(aoclsparse_int i = 0; i < std::min(c.m, c.n); i++)
                if(c.iurow[i] == c.idiag[i] + 1)
                    std::memcpy(&dv[vs * (size_t)i], static_cast<const char *>(c.val) + vs * (size_t)(c.idiag[i] - c.base),
                                vs);
            st = A->dev_diag.upload(dv.data(), vs * (size_t)m, rt.stream());
            if(st != aoclsparse_status_success)
                return st;
        }
        st = A->trsv_scratch.alloc(2 * sizeof(unsigned int));
        if(st != aoclsparse_status_success)
            return st;
        if(!transposed)
        {
            plan.rs   = upper ? A->dev_opt_iurow.as<aoclsparse_int>() : A->dev_opt.ptr.as<aoclsparse_int>();
            plan.re   = upper ? A->dev_opt.ptr.as<aoclsparse_int>() + 1 : A->dev_opt_idiag.as<aoclsparse_int>();
            plan.ind  = A->dev_opt.ind.as<aoclsparse_int>();
            plan.val  = A->dev_opt.val.ptr;
            plan.base = c.base;
            plan.reverse = false;
            // L: row i depends on smaller rows (ascending sweep); U: on larger rows (descending)
            st = level_sets(m, upper ? c.iurow : c.ptr, upper ? c.ptr + 1 : c.idiag, c.ind, c.base, upper, plan);
        }
        else
        {
            std::vector<aoclsparse_int> tptr, tind;
            st = A->val_type == aoclsparse_smat ? build_transposed_triangle<float>(c, upper, plan, tptr, tind)
                                                : build_transposed_triangle<double>(c, upper, plan, tptr, tind);
            if(st != aoclsparse_status_success)
                return st;
            plan.rs   = plan.own_ptr.as<aoclsparse_int>();
            plan.re   = plan.own_ptr.as<aoclsparse_int>() + 1;
            plan.ind  = plan.own_ind.as<aoclsparse_int>();
            plan.val  = plan.own_val.ptr;
            plan.base = 0;
            // L^T is upper triangular: x_c needs x_i for i > c, applied in DESCENDING i
            // (ref_trsv_lth sweeps i = m-1..0); U^T is lower triangular, ascending.
            plan.reverse = !upper;
            st = level_sets(m, tptr.data(), tptr.data() + 1, tind.data(), 0, !upper, plan);
        }
        if(st != aoclsparse_status_success)
            return st;
        plan.valid = true;
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

} // namespace mi355

namespace
{

template <typename T>
aoclsparse_status trsv_t(aoclsparse_operation trans, const T alpha, aoclsparse_matrix A,
                         const aoclsparse_mat_descr descr, const T *b, aoclsparse_int incb, T *x,
                         aoclsparse_int incx, aoclsparse_int kid, aoclsparse_matrix_data_type vt)
{
    // trsv.cpp:59-113
    if(!A || !x || !b || !descr)
        return aoclsparse_status_invalid_pointer;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    const aoclsparse_int m = A->m;
    if(m <= 0 || A->nnz <= 0)
        return aoclsparse_status_invalid_size;
    if(m != A->n || incb <= 0 || incx <= 0)
        return aoclsparse_status_invalid_value;
    if(!A->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    if(descr->base != aoclsparse_index_base_zero && descr->base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    if(trans != aoclsparse_operation_none && trans != aoclsparse_operation_transpose
       && trans != aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_not_implemented;
    if(descr->type != aoclsparse_matrix_type_symmetric && descr->type != aoclsparse_matrix_type_triangular)
        return aoclsparse_status_invalid_value;
    if(descr->diag_type == aoclsparse_diag_type_zero)
        return aoclsparse_status_invalid_value;
    if(descr->fill_mode != aoclsparse_fill_mode_lower && descr->fill_mode != aoclsparse_fill_mode_upper)
        return aoclsparse_status_not_implemented;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;

    // trsv.cpp:128-137: lazy clean CSR, then the rank check
    aoclsparse_status st = csr_optimize(A);
    if(st != aoclsparse_status_success)
        return st;
    const bool unit = descr->diag_type == aoclsparse_diag_type_unit;
    if(!A->opt_csr_full_diag && !unit)
        return aoclsparse_status_invalid_value;
    // KAT of trsv.cpp:315-376 has kernels 0..3 per doid
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    // (m-1)*inc must not overflow, trsv.cpp:407-411
    if((long long)(m - 1) * incb > 2147483647LL || (long long)(m - 1) * incx > 2147483647LL)
        return aoclsparse_status_invalid_size;

    Runtime &rt = Runtime::get();
    st          = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();

    const bool upper = descr->fill_mode == aoclsparse_fill_mode_upper;
    const bool tr    = trans != aoclsparse_operation_none;
    st               = ensure_trsv(A, upper, tr);
    if(st != aoclsparse_status_success)
        return st;
    std::shared_lock<std::shared_mutex> r(A->guard);
    const TrsvPlan                     &plan = A->trsv_plan[(upper ? 2 : 0) + (tr ? 1 : 0)];

    // schedule: kid 0 = one launch per level, kid 1..3 = sync-free single launch;
    // auto: per-level launches while the DAG is shallow, sync-free once launches would dominate
    const int schedule = kid == 0 ? 0 : (kid > 0 ? 1 : (plan.nlevels <= 48 ? 0 : 1));

    const bool bdev = rt.is_device_pointer(b), xdev = rt.is_device_pointer(x);
    const T   *db   = nullptr;
    T         *dx   = nullptr;
    void      *tmp  = nullptr;
    const size_t nb = (size_t)(m - 1) * incb + 1, nx = (size_t)(m - 1) * incx + 1;
    // b: contiguous device vector
    if(bdev && incb == 1)
        db = b;
    else
    {
        st = rt.staging(0, sizeof(T) * (size_t)m, &tmp);
        if(st != aoclsparse_status_success)
            return st;
        T *cb = static_cast<T *>(tmp);
        if(bdev)
            st = launch_strided_gather<T>(rt.stream(), b, incb, m, cb);
        else if(incb == 1)
            MI355_HIP_TRY(hipMemcpyAsync(cb, b, sizeof(T) * (size_t)m, hipMemcpyHostToDevice, rt.stream()));
        else
        {
            void *raw = nullptr;
            st        = rt.staging(1, sizeof(T) * nb, &raw);
            if(st != aoclsparse_status_success)
                return st;
            MI355_HIP_TRY(hipMemcpyAsync(raw, b, sizeof(T) * nb, hipMemcpyHostToDevice, rt.stream()));
            st = launch_strided_gather<T>(rt.stream(), static_cast<const T *>(raw), incb, m, cb);
        }
        if(st != aoclsparse_status_success)
            return st;
        db = cb;
    }
    // x: contiguous device vector the kernels write
    const bool xdirect = xdev && incx == 1;
    if(xdirect)
        dx = x;
    else
    {
        st = rt.staging(2, sizeof(T) * (size_t)m, &tmp);
        if(st != aoclsparse_status_success)
            return st;
        dx = static_cast<T *>(tmp);
    }
    st = launch_trsv<T>(rt.stream(), schedule, plan.reverse, unit, plan.base, alpha, m, plan.rs, plan.re,
                        plan.ind, static_cast<const T *>(plan.val), A->dev_diag.as<T>(), plan, db, dx,
                        A->trsv_scratch.as<unsigned int>());
    if(st != aoclsparse_status_success)
        return st;
    if(!xdirect)
    {
        if(xdev)
            st = launch_strided_scatter<T>(rt.stream(), dx, m, x, incx);
        else if(incx == 1)
            MI355_HIP_TRY(hipMemcpyAsync(x, dx, sizeof(T) * (size_t)m, hipMemcpyDeviceToHost, rt.stream()));
        else
        {
            // strided host x: only the strided slots may change -> read back compact, scatter on host
            std::vector<T> hx((size_t)m);
            MI355_HIP_TRY(hipMemcpyAsync(hx.data(), dx, sizeof(T) * (size_t)m, hipMemcpyDeviceToHost, rt.stream()));
            MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
            for(aoclsparse_int i = 0; i < m; i++)
                x[(size_t)i * incx] = hx[i];
        }
        if(st != aoclsparse_status_success)
            return st;
    }
    if(!xdev || schedule == 1)
    {
        // host semantics, and the sync-free path reports a (never expected) spin timeout
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        if(schedule == 1)
        {
            unsigned int words[2] = {0, 0};
            MI355_HIP_TRY(hipMemcpy(words, A->trsv_scratch.ptr, sizeof(words), hipMemcpyDeviceToHost));
            if(words[1])
                return aoclsparse_status_internal_error;
        }
    }
    (void)nx;
    return aoclsparse_status_success;
}

} // namespace

extern "C" {

aoclsparse_status aoclsparse_dtrsv(aoclsparse_operation trans, const double alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, const double *b, double *x)
{
    return trsv_t<double>(trans, alpha, A, descr, b, 1, x, 1, -1, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_strsv(aoclsparse_operation trans, const float alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, const float *b, float *x)
{
    return trsv_t<float>(trans, alpha, A, descr, b, 1, x, 1, -1, aoclsparse_smat);
}

aoclsparse_status aoclsparse_dtrsv_kid(aoclsparse_operation trans, const double alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const double *b, double *x,
                                       aoclsparse_int kid)
{
    return trsv_t<double>(trans, alpha, A, descr, b, 1, x, 1, kid, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_strsv_kid(aoclsparse_operation trans, const float alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const float *b, float *x,
                                       aoclsparse_int kid)
{
    return trsv_t<float>(trans, alpha, A, descr, b, 1, x, 1, kid, aoclsparse_smat);
}

aoclsparse_status aoclsparse_dtrsv_strided(aoclsparse_operation trans, const double alpha, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const double *b,
                                           const aoclsparse_int incb, double *x, const aoclsparse_int incx)
{
    return trsv_t<double>(trans, alpha, A, descr, b, incb, x, incx, -1, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_strsv_strided(aoclsparse_operation trans, const float alpha, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const float *b,
                                           const aoclsparse_int incb, float *x, const aoclsparse_int incx)
{
    return trsv_t<float>(trans, alpha, A, descr, b, incb, x, incx, -1, aoclsparse_smat);
}

aoclsparse_status aoclsparse_mi355_get_trsv_levels(const aoclsparse_matrix A, aoclsparse_fill_mode fill,
                                                   aoclsparse_operation op, aoclsparse_int *levels)
{
    if(!A || !levels)
        return aoclsparse_status_invalid_pointer;
    std::shared_lock<std::shared_mutex> r(A->guard);
    const TrsvPlan &p = A->trsv_plan[(fill == aoclsparse_fill_mode_upper ? 2 : 0)
                                     + (op != aoclsparse_operation_none ? 1 : 0)];
    *levels = p.valid ? p.nlevels : -1;
    return aoclsparse_status_success;
}

} // extern "C"
