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

// Host view of the strict triangle one (fill, op) variant walks: row i depends on the rows listed in
// [ptr[i], ptr[i+1]) of ind (0-based), val in the order the reference's chain applies them.
template <typename T>
struct Triangle
{
    std::vector<aoclsparse_int> ptr, ind;
    std::vector<T>              val;
    bool                        descending = false; // solve order m-1..0 (dependencies point to larger rows)
};

template <typename T>
static void build_triangle(const HostCsr &c, bool upper, bool transposed, Triangle<T> &t)
{
    const aoclsparse_int  m = c.m, b = c.base;
    const aoclsparse_int *s = upper ? c.iurow : c.ptr; // strict triangle of row i: [s[i], e[i]) in base b
    const aoclsparse_int *e = upper ? c.ptr + 1 : c.idiag;
    const T              *v = static_cast<const T *>(c.val);
    t.ptr.assign((size_t)m + 1, 0);
    if(!transposed)
    {
        // L: rows ascending, entries left to right (ref_trsv_l); U: rows descending (ref_trsv_u)
        for(aoclsparse_int i = 0; i < m; i++)
            t.ptr[i + 1] = t.ptr[i] + (e[i] - s[i]);
        t.ind.resize((size_t)std::max(t.ptr[m], 1));
        t.val.resize((size_t)std::max(t.ptr[m], 1));
        for(aoclsparse_int i = 0; i < m; i++)
            for(aoclsparse_int p = s[i] - b, q = t.ptr[i]; p < e[i] - b; p++, q++)
            {
                t.ind[q] = c.ind[p] - b;
                t.val[q] = v[p];
            }
        t.descending = upper;
        return;
    }
    // transposed solves are column sweeps (ref_trsv_lth / _uth): x_c receives a_ic * x_i from every
    // stored (i, c).  Row form on the transposed triangle: row c lists the i's.  L^T: sweep i = m-1..0,
    // so x_c is updated in DESCENDING i; U^T: i = 0..m-1, ascending.
    for(aoclsparse_int i = 0; i < m; i++)
        for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
            t.ptr[c.ind[p] - b + 1]++;
    for(aoclsparse_int j = 0; j < m; j++)
        t.ptr[j + 1] += t.ptr[j];
    t.ind.resize((size_t)std::max(t.ptr[m], 1));
    t.val.resize((size_t)std::max(t.ptr[m], 1));
    std::vector<aoclsparse_int> next(t.ptr.begin(), t.ptr.end() - 1);
    const bool                  desc_fill = !upper; // L^T: fill from the largest source row down
    for(aoclsparse_int ii = 0; ii < m; ii++)
    {
        const aoclsparse_int i = desc_fill ? m - 1 - ii : ii;
        for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
        {
            const aoclsparse_int q = next[c.ind[p] - b]++;
            t.ind[q]               = i;
            t.val[q]               = v[p];
        }
    }
    t.descending = !upper; // L^T is upper triangular: x_c needs x_i, i > c
}

// level[i] = 1 + max level of the rows row i depends on; rows bucketed by level (counting sort,
// ascending row index inside a level); then the triangle is re-laid out in that order and the hybrid
// schedule (runs of narrow levels vs. wide levels) is derived.
template <typename T>
static aoclsparse_status build_levels(aoclsparse_int m, const Triangle<T> &t, TrsvPlan &plan)
{
    std::vector<aoclsparse_int> level((size_t)m, 0);
    aoclsparse_int              nlev = 0;
    for(aoclsparse_int k = 0; k < m; k++)
    {
        const aoclsparse_int i  = t.descending ? m - 1 - k : k;
        aoclsparse_int       lv = 0;
        for(aoclsparse_int p = t.ptr[i]; p < t.ptr[i + 1]; p++)
            lv = std::max(lv, level[t.ind[p]] + 1);
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
    plan.nnz_tri = t.ptr[m];

    // level-ordered copy of the triangle; dependencies are rewritten as POSITIONS in that order
    std::vector<aoclsparse_int> pos((size_t)m);
    for(aoclsparse_int k = 0; k < m; k++)
        pos[rowmap[k]] = k;
    std::vector<aoclsparse_int> pptr((size_t)m + 1, 0), pind(t.ind.size());
    std::vector<T>              pval(t.val.size());
    for(aoclsparse_int k = 0; k < m; k++)
    {
        const aoclsparse_int i = rowmap[k], len = t.ptr[i + 1] - t.ptr[i];
        pptr[k + 1]            = pptr[k] + len;
        for(aoclsparse_int j = 0; j < len; j++)
            pind[pptr[k] + j] = pos[t.ind[t.ptr[i] + j]];
        std::copy(t.val.begin() + t.ptr[i], t.val.begin() + t.ptr[i + 1], pval.begin() + pptr[k]);
    }
    // hybrid schedule
    plan.segments.clear();
    plan.launches = 0;
    for(aoclsparse_int l = 0; l < nlev;)
    {
        const bool     narrow = plan.level_ptr[l + 1] - plan.level_ptr[l] <= TRSV_NARROW;
        aoclsparse_int e      = l + 1;
        while(e < nlev && ((plan.level_ptr[e + 1] - plan.level_ptr[e] <= TRSV_NARROW) == narrow))
            e++;
        // a lone narrow level between wide ones is cheaper as an ordinary launch
        plan.segments.push_back({l, e, narrow && e - l > 1});
        plan.launches += (narrow && e - l > 1) ? 1 : e - l;
        l = e;
    }
    hipStream_t       st = Runtime::get().stream();
    aoclsparse_status rc = plan.rowmap.upload(rowmap.data(), sizeof(aoclsparse_int) * (size_t)m, st);
    if(rc == aoclsparse_status_success)
        rc = plan.levels.upload(plan.level_ptr.data(), sizeof(aoclsparse_int) * ((size_t)nlev + 1), st);
    if(rc == aoclsparse_status_success)
        rc = plan.pptr.upload(pptr.data(), sizeof(aoclsparse_int) * ((size_t)m + 1), st);
    if(rc == aoclsparse_status_success)
        rc = plan.pind.upload(pind.data(), sizeof(aoclsparse_int) * pind.size(), st);
    if(rc == aoclsparse_status_success)
        rc = plan.pval.upload(pval.data(), sizeof(T) * pval.size(), st);
    if(rc == aoclsparse_status_success)
        rc = plan.xp.alloc(sizeof(T) * (size_t)std::max(m, 1));
    return rc;
}

template <typename T>
static aoclsparse_status build_plan_t(const HostCsr &c, bool upper, bool transposed, TrsvPlan &plan)
{
    Triangle<T> t;
    build_triangle<T>(c, upper, transposed, t);
    return build_levels<T>(c.m, t, plan);
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
    const HostCsr       &c  = *A->opt;
    const size_t         vs = val_size(A->val_type);
    const aoclsparse_int m  = c.m;
    Runtime             &rt = Runtime::get();
    try
    {
        if(!A->dev_diag.ptr)
        {
            // diagonal values (only read for non-unit solves, which require a full diagonal)
            std::vector<char> dv(vs * (size_t)std::max(m, 1), 0);
            for(aoclsparse_int i = 0; i < std::min(c.m, c.n); i++)
                if(c.iurow[i] == c.idiag[i] + 1)
                    std::memcpy(&dv[vs * (size_t)i],
                                static_cast<const char *>(c.val) + vs * (size_t)(c.idiag[i] - c.base), vs);
            st = A->dev_diag.upload(dv.data(), vs * (size_t)m, rt.stream());
            if(st == aoclsparse_status_success)
                st = A->trsv_scratch.alloc(2 * sizeof(unsigned int));
            if(st != aoclsparse_status_success)
                return st;
        }
        st = A->val_type == aoclsparse_smat ? build_plan_t<float>(c, upper, transposed, plan)
                                            : build_plan_t<double>(c, upper, transposed, plan);
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

    // schedule (all three give the same bits): kid 0 = one launch per level, kid 1/2 = hybrid (narrow
    // level runs inside one workgroup), kid 3 = sync-free single launch.  auto: a shallow DAG of wide
    // levels is cheapest as plain launches; otherwise sync-free, which measured fastest on both the
    // 2-D Laplacian and the shell-like ILU(0) factors (profiles/r1).
    const int schedule = kid == 0 ? 0 : (kid == 3 ? 2 : (kid > 0 ? 1 : (plan.nlevels <= 32 ? 0 : 2)));

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
    st = launch_trsv<T>(rt.stream(), schedule, unit, alpha, m, plan, A->dev_diag.as<T>(), db, dx,
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
    if(!xdev || schedule == 2)
    {
        // host semantics, and the sync-free path reports a (never expected) spin timeout
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        if(schedule == 2)
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
