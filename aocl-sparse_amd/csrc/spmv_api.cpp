// spmv_api.cpp -- aoclsparse_?csrmv (raw arrays) and aoclsparse_?mv (handle) on the HIP engine.
//
// Argument checks and their order follow level2/aoclsparse_csrmv.hpp:61-103 and
// level2/aoclsparse_mv.cpp:55-121 of the reference; kernel choice follows csrmv.hpp:322-355
// (nnz <= 10*m -> scalar order, kid 1/2 -> 4-lane order, kid 3 / auto -> 8-lane order).
#include "internal.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace mi355;

namespace
{

// A host or device array made addressable by the GPU for one call.
struct Arg
{
    void       *dev   = nullptr;
    void       *host  = nullptr;
    size_t      bytes = 0;
    bool        staged = false;
    aoclsparse_status in(Runtime &rt, int slot, const void *p, size_t nbytes, bool is_dev, bool copy)
    {
        bytes = nbytes;
        if(is_dev)
        {
            dev = const_cast<void *>(p);
            return aoclsparse_status_success;
        }
        staged = true;
        host   = const_cast<void *>(p);
        aoclsparse_status st = rt.staging(slot, nbytes, &dev);
        if(st != aoclsparse_status_success)
            return st;
        if(copy && nbytes)
            return rt.h2d(dev, p, nbytes); // pipelined through pinned memory when large
        return aoclsparse_status_success;
    }
    aoclsparse_status out(Runtime &rt)
    {
        if(staged && bytes)
            return rt.d2h(host, dev, bytes);
        return aoclsparse_status_success;
    }
};

inline bool valid_op(aoclsparse_operation op)
{
    return op == aoclsparse_operation_none || op == aoclsparse_operation_transpose
           || op == aoclsparse_operation_conjugate_transpose;
}
inline bool valid_base(int b)
{
    return b == aoclsparse_index_base_zero || b == aoclsparse_index_base_one;
}
inline bool valid_type(int t)
{
    return t >= aoclsparse_matrix_type_general && t <= aoclsparse_matrix_type_triangular;
}

// csrmv.hpp:322-355.  Returns the summation order (0 scalar, 1 lane4, 2 lane8) and whether the
// caller pinned a kernel (strict order also for rows longer than a tile).
template <typename T>
aoclsparse_status resolve_order(aoclsparse_int kid, aoclsparse_int m, aoclsparse_int nnz,
                                aoclsparse_int max_row_nnz, int &order, bool &strict)
{
    strict = kid >= 0;
    if constexpr(sizeof(T) == 4)
    {
        order = 2; // the only float kernel for general/no-trans is the 8-lane one (:317-321)
    }
    else
    {
        if((long long)nnz <= 10LL * (long long)m)
            kid = 0;
        if(kid < 0 || kid == 3)
            order = 2;
        else if(kid == 0)
            order = 0;
        else if(kid == 1 || kid == 2)
            order = 1;
        else
            return aoclsparse_status_invalid_kid;
    }
    // a lane-grouped kernel on rows shorter than the lane count IS the scalar chain (the vector
    // loop never runs): use one lane per row, same bits, no idle lanes
    if((order == 2 && max_row_nnz < 8) || (order == 1 && max_row_nnz < 4))
        order = 0;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status run_on_device_csr(Runtime &rt, aoclsparse_int kid, const DeviceCsr &d,
                                    const SpmvPlan &plan, T alpha, const T *x, T beta, T *y,
                                    aoclsparse_int nx, aoclsparse_int ny, unsigned int *stale = nullptr)
{
    int               order;
    bool              strict;
    aoclsparse_status st = resolve_order<T>(kid, d.m, d.nnz, plan.max_row_nnz, order, strict);
    // aoclsparse_mi355_set_option(spmv_strict, 1): every row in the reference's order without pinning a kid (the automatic
    // order stays) -- bit-exact everywhere, at the price of one lane's serial chain per long row (web-like, 91 rows of up to
    // 2,908 entries: 44.5 vs 28.5 us in round 3).  The default leaves rows of >= SPMV_TREE_MIN entries to a wavefront tree.
    if(!strict && plan_option(aoclsparse_mi355_option_spmv_strict) == 1)
        strict = true;
    if(st != aoclsparse_status_success)
        return st;
    const bool xdev = rt.is_device_pointer(x), ydev = rt.is_device_pointer(y);
    Arg        ax, ay;
    st = ax.in(rt, 3, x, sizeof(T) * (size_t)nx, xdev, true);
    if(st != aoclsparse_status_success)
        return st;
    st = ay.in(rt, 4, y, sizeof(T) * (size_t)ny, ydev, beta != T(0));
    if(st != aoclsparse_status_success)
        return st;
    if(plan.sell.valid) // format chosen by optimize for an mv hint: every order is exact there
        st = launch_sellmv<T>(rt.stream(), order, plan.sell.pack, alpha, d.m, plan.sell.nslices, plan.sell.slice_ptr.as<long long>(),
                              plan.sell.val.as<T>(), plan.sell.col.as<aoclsparse_int>(),
                              plan.sell.rowlen.as<aoclsparse_int>(), static_cast<const T *>(ax.dev), beta,
                              static_cast<T *>(ay.dev), plan.sell.shared ? plan.sell.cptr.as<long long>() : nullptr,
                              plan.sell.shared ? plan.sell.lead.as<unsigned short>() : nullptr, plan.max_row_nnz,
                              plan.sell.next_direction());
    else if(plan.merge.valid && order == 0 && !strict) // balanced tiles for irregular rows (scalar order, no pinned kid)
    {
        // one launch; the pieces of cut rows meet in the piece set of this stream (internal.hpp, MergePlan).  Finding the set
        // and enqueueing are one step.
        const MergePlan            &mp = plan.merge;
        std::lock_guard<std::mutex> g(mp.launch_lock);
        MergePlan::GranuleSet      *gs  = nullptr;
        const unsigned long long    uid = stream_uid(rt.stream());
        for(auto &c : mp.sets)
            if(c->stream == (void *)rt.stream())
                gs = c.get();
        const size_t gbytes = sizeof(unsigned long long) * 3 * (size_t)mp.ntiles;
        if(gs && gs->uid != uid) // the address of a stream the caller has destroyed, handed out again: its launches may still run
        {
            MI355_HIP_TRY(hipDeviceSynchronize());
            MI355_HIP_TRY(hipMemsetAsync(gs->granules.ptr, 0, gbytes, rt.stream()));
            gs->uid = uid;
        }
        if(!gs)
        {
            if(mp.sets.size() >= MergePlan::MAX_SETS) // (a caller cycling through streams: everything enqueued so far completes first)
            {
                MI355_HIP_TRY(hipDeviceSynchronize());
                mp.sets.clear();
            }
            try
            {
                mp.sets.emplace_back(new MergePlan::GranuleSet);
            }
            catch(const std::bad_alloc &)
            {
                return aoclsparse_status_memory_error;
            }
            gs                   = mp.sets.back().get();
            aoclsparse_status sa = gs->granules.alloc(gbytes);
            if(sa == aoclsparse_status_success && hipMemsetAsync(gs->granules.ptr, 0, gbytes, rt.stream()) != hipSuccess)
                sa = aoclsparse_status_internal_error;
            if(sa != aoclsparse_status_success)
            {
                mp.sets.pop_back();
                return sa;
            }
            gs->stream = (void *)rt.stream();
            gs->uid    = uid;
        }
        st = launch_mergepath<T>(rt.stream(), d.base, alpha, mp.ntiles, mp.starts.as<aoclsparse_int>(), mp.first.as<aoclsparse_int>(),
                                 d.val.as<T>(), d.ind.as<aoclsparse_int>(), d.ptr.as<aoclsparse_int>(),
                                 static_cast<const T *>(ax.dev), beta, static_cast<T *>(ay.dev), gs->granules.ptr);
    }
    else
    {
        // (bit 1 of the tile word: blocks in descending order -- every second product of a plan whose blocks are in row order;
        // the heavy-first order of irregular matrices is left alone)
        const bool by_rows = !(plan.heavy_first && (plan.tile & 1) == 0);
        const int  rev     = by_rows && plan_option(aoclsparse_mi355_option_alternate_sweeps) != 0
                                 ? (int)(plan.sweeps.fetch_add(1u, std::memory_order_relaxed) & 1u)
                                 : 0;
        st = launch_csrmv<T>(rt.stream(), order, strict, plan.tile | (rev ? 2 : 0), d.base, alpha, d.m, d.val.as<T>(),
                             d.ind.as<aoclsparse_int>(), d.ptr.as<aoclsparse_int>(),
                             plan.rowblocks.as<aoclsparse_int>(), plan.nblocks, static_cast<const T *>(ax.dev),
                             beta, static_cast<T *>(ay.dev),
                             (plan.heavy_first && (plan.tile & 1) == 0) ? plan.rowblocks4.as<aoclsparse_int>() : nullptr,
                             plan.max_row_nnz, stale);
    }
    if(st != aoclsparse_status_success)
        return st;
    st = ay.out(rt);
    if(st != aoclsparse_status_success)
        return st;
    if(ay.staged) // host-pointer semantics: result visible on return
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status scale_y(Runtime &rt, T *y, aoclsparse_int n, T beta)
{
    if(n <= 0)
        return aoclsparse_status_success;
    const bool ydev = rt.is_device_pointer(y);
    Arg        ay;
    aoclsparse_status st = ay.in(rt, 4, y, sizeof(T) * (size_t)n, ydev, beta != T(0));
    if(st != aoclsparse_status_success)
        return st;
    st = launch_scale<T>(rt.stream(), static_cast<T *>(ay.dev), n, beta);
    if(st != aoclsparse_status_success)
        return st;
    st = ay.out(rt);
    if(st == aoclsparse_status_success && ay.staged)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    return st;
}

// ---- handle path ---------------------------------------------------------------------------------
template <typename T>
aoclsparse_status mv_t(aoclsparse_operation op, const T *alpha, aoclsparse_matrix A,
                       const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y,
                       aoclsparse_matrix_data_type vt)
{
    if(!alpha || !beta || !A || !descr || !x || !y)
        return aoclsparse_status_invalid_pointer;
    if(A->input_format == aoclsparse_csr_mat && !A->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    if(!valid_op(op))
        return aoclsparse_status_invalid_value;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(!valid_type(descr->type))
        return aoclsparse_status_invalid_value;
    if((descr->type == aoclsparse_matrix_type_symmetric || descr->type == aoclsparse_matrix_type_hermitian)
       && A->m != A->n)
        return aoclsparse_status_invalid_size;
    if(A->input_format != aoclsparse_csr_mat) // mv.cpp:96-97 (COO handles: convert with aoclsparse_convert_csr)
        return aoclsparse_status_not_implemented;
    if(op == aoclsparse_operation_conjugate_transpose)
        op = aoclsparse_operation_transpose;
    if(descr->type == aoclsparse_matrix_type_hermitian)
        return aoclsparse_status_not_implemented;

    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();

    // mv.cpp:116-121: empty matrix still scales y
    if(A->m == 0 || A->n == 0 || (A->nnz == 0 && descr->type == aoclsparse_matrix_type_general))
        return scale_y<T>(rt, y, op == aoclsparse_operation_none ? A->m : A->n, *beta);

    const doid     id  = get_doid(descr, op);
    aoclsparse_int kid = -1; // magic_box.hpp:34-53: first hint with matching action + doid
    for(const Hint &h : A->hints)
        if(h.act == action_mv && h.id == id)
        {
            kid = h.kid;
            break;
        }
    const bool tr   = op != aoclsparse_operation_none;
    DeviceCsr *dcsr = nullptr;
    SpmvPlan  *plan = nullptr;
    if(descr->type != aoclsparse_matrix_type_general)
    {
        // symmetric / triangular operator: general CSR derived from the clean CSR (derived.cpp); the
        // reference optimises the CSR on the fly here too (mv.cpp:134-149)
        Derived *dv = nullptr;
        st          = ensure_derived(A, descr->type, descr->fill_mode, descr->diag_type, tr, dv);
        if(st != aoclsparse_status_success)
            return st;
        dcsr = &dv->dev;
        plan = &dv->plan;
    }
    else
    {
        st = ensure_spmv(A, tr, dcsr, plan);
        if(st != aoclsparse_status_success)
            return st;
        // A handle that keeps being multiplied without ever having been given an mv hint is promoted to the SELL-64
        // copy an optimize would have built (same summation orders, same bits; the copy costs about three products).
        // aoclsparse_memory_usage_minimal and aoclsparse_mi355_set_option(sell, 0) forbid it.
        const bool promote = !plan->sell.valid && !plan->sell.tried && !plan->merge.valid && !is_complex_type(A->val_type)
                             && A->mem_policy == aoclsparse_memory_usage_unrestricted
                             && plan->mv_calls.fetch_add(1, std::memory_order_relaxed) + 1 >= SELL_PROMOTE_CALLS;
        if(promote || (plan->sell.wanted && !plan->sell.valid)) // or: values changed since the SELL copy was built
        {
            std::unique_lock<std::shared_mutex> w(A->guard);
            st = build_sell((tr ? *A->trans : A->user).ptr, *dcsr, val_size(A->val_type), *plan);
            if(st != aoclsparse_status_success)
                return st;
        }
    }
    std::shared_lock<std::shared_mutex> r(A->guard);
    return run_on_device_csr<T>(rt, kid, *dcsr, *plan, *alpha, x, *beta, y, dcsr->n, dcsr->m);
}

// ---- raw-array path --------------------------------------------------------------------------------
// Plans for device-resident raw arrays are cached on (row_ptr address, m, nnz, base): the one-shot
// API has no handle to hang an analysis on ("raw csrmv").  The key says nothing about the CONTENTS (a
// caller may free its arrays and get the same address back for another matrix of the same size), so every hit is
// validated ON THE DEVICE, INSIDE the product: each workgroup of csr_adaptive_kernel checks its own block entry against the
// live row_ptr values it loads anyway, computes its rows from the live arrays when the entry is stale (same chains: the
// result is right either way) and raises a pinned word that the NEXT raw call sees -- it then drops the cache and
// rebuilds.  No check kernel, no stream round trip per call (round 3: ~70 us of each 0.334 ms call on the 4096^2 Laplacian).  A plan is handed out as a shared_ptr copied under the lock, so a concurrent call
// that evicts the slot cannot free or rewrite a plan that is still in use.
struct RawPlan
{
    const void               *key = nullptr;
    aoclsparse_int            m = -1, nnz = -1, base = -1;
    std::shared_ptr<SpmvPlan> plan;
};
constexpr int RAW_CACHE = 8;
RawPlan       g_raw[RAW_CACHE];
int           g_raw_next = 0;

template <typename T>
aoclsparse_status csrmv_t(aoclsparse_operation trans, const T *alpha, aoclsparse_int m,
                          aoclsparse_int n, aoclsparse_int nnz, const T *val,
                          const aoclsparse_int *col, const aoclsparse_int *row,
                          const aoclsparse_mat_descr descr, const T *x, const T *beta, T *y,
                          aoclsparse_matrix_data_type vt)
{
    // csrmv.hpp:61-103, same order
    if(!alpha || !beta)
        return aoclsparse_status_invalid_pointer;
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(!valid_base(descr->base))
        return aoclsparse_status_invalid_value;
    if(!valid_type(descr->type))
        return aoclsparse_status_invalid_value;
    if(!valid_op(trans))
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_symmetric)
        return aoclsparse_status_not_implemented;
    if(descr->type == aoclsparse_matrix_type_symmetric && m != n)
        return aoclsparse_status_invalid_size;
    if(m < 0 || n < 0 || nnz < 0)
        return aoclsparse_status_invalid_size;
    if(!val || !row || !col || !x || !y)
        return aoclsparse_status_invalid_pointer;
    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();
    const bool tr = trans != aoclsparse_operation_none;
    const bool mdev = rt.is_device_pointer(row);

    if(descr->type == aoclsparse_matrix_type_symmetric)
    {
        // aoclsparse_csrmv_symm (csrmv_kr.hpp:41-92): the arrays hold ONE triangle; every stored
        // entry except a diagonal sitting LAST in its row is applied as the pair (i,c),(c,i); op is
        // irrelevant for a symmetric operator.  One-shot path: expand on the host, run general.
        if(m == 0)
            return aoclsparse_status_success;
        std::vector<aoclsparse_int> hrow, hcol;
        std::vector<T>              hval;
        const aoclsparse_int       *prow = row, *pcol = col;
        const T                    *pval = val;
        const int                   b    = descr->base;
        try
        {
            if(mdev)
            {
                hrow.resize((size_t)m + 1), hcol.resize(nnz), hval.resize(nnz);
                MI355_HIP_TRY(hipMemcpy(hrow.data(), row, sizeof(aoclsparse_int) * ((size_t)m + 1),
                                        hipMemcpyDeviceToHost));
                MI355_HIP_TRY(hipMemcpy(hcol.data(), col, sizeof(aoclsparse_int) * (size_t)nnz,
                                        hipMemcpyDeviceToHost));
                MI355_HIP_TRY(hipMemcpy(hval.data(), val, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToHost));
                prow = hrow.data(), pcol = hcol.data(), pval = hval.data();
            }
            std::vector<aoclsparse_int> eptr((size_t)m + 1, 0);
            auto                        last_is_diag = [&](aoclsparse_int i) {
                return prow[i + 1] > prow[i] && pcol[prow[i + 1] - b - 1] - b == i;
            };
            for(aoclsparse_int i = 0; i < m; i++)
            {
                const aoclsparse_int ld = last_is_diag(i) ? 1 : 0;
                eptr[i + 1] += prow[i + 1] - prow[i]; // the row itself (diagonal included once)
                for(aoclsparse_int p = prow[i] - b; p < prow[i + 1] - b - ld; p++)
                    eptr[pcol[p] - b + 1]++; // mirrored entry
            }
            for(aoclsparse_int i = 0; i < m; i++)
                eptr[i + 1] += eptr[i];
            const aoclsparse_int        ennz = eptr[m];
            std::vector<aoclsparse_int> eind((size_t)std::max(ennz, 1)), next(eptr.begin(), eptr.end() - 1);
            std::vector<T>              eval((size_t)std::max(ennz, 1));
            for(aoclsparse_int i = 0; i < m; i++)
            {
                const aoclsparse_int ld = last_is_diag(i) ? 1 : 0;
                for(aoclsparse_int p = prow[i] - b; p < prow[i + 1] - b; p++)
                {
                    aoclsparse_int q = next[i]++;
                    eind[q] = pcol[p] - b, eval[q] = pval[p];
                    if(p < prow[i + 1] - b - ld)
                    {
                        q       = next[pcol[p] - b]++;
                        eind[q] = i, eval[q] = pval[p];
                    }
                }
            }
            _aoclsparse_matrix tmp;
            tmp.m = m, tmp.n = m, tmp.nnz = ennz, tmp.base = aoclsparse_index_base_zero, tmp.val_type = vt;
            tmp.user.m = m, tmp.user.n = m, tmp.user.nnz = ennz, tmp.user.base = aoclsparse_index_base_zero;
            tmp.user.ptr = eptr.data(), tmp.user.ind = eind.data(), tmp.user.val = eval.data();
            DeviceCsr *dcsr = nullptr;
            SpmvPlan  *plan = nullptr;
            st              = ensure_spmv(&tmp, false, dcsr, plan);
            if(st != aoclsparse_status_success)
                return st;
            st = run_on_device_csr<T>(rt, -1, *dcsr, *plan, *alpha, x, *beta, y, m, m);
            if(st == aoclsparse_status_success)
                MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
            return st;
        }
        catch(const std::bad_alloc &)
        {
            return aoclsparse_status_memory_error;
        }
    }

    if(tr)
    {
        // transposed one-shot call: gather the CSR on the host, transpose there, run as a handle
        std::vector<aoclsparse_int> hrow, hcol;
        std::vector<T>              hval;
        const aoclsparse_int       *prow = row, *pcol = col;
        const T                    *pval = val;
        try
        {
            if(mdev)
            {
                hrow.resize((size_t)m + 1), hcol.resize(nnz), hval.resize(nnz);
                MI355_HIP_TRY(hipMemcpy(hrow.data(), row, sizeof(aoclsparse_int) * ((size_t)m + 1),
                                        hipMemcpyDeviceToHost));
                MI355_HIP_TRY(hipMemcpy(hcol.data(), col, sizeof(aoclsparse_int) * (size_t)nnz,
                                        hipMemcpyDeviceToHost));
                MI355_HIP_TRY(hipMemcpy(hval.data(), val, sizeof(T) * (size_t)nnz, hipMemcpyDeviceToHost));
                prow = hrow.data(), pcol = hcol.data(), pval = hval.data();
            }
        }
        catch(const std::bad_alloc &)
        {
            return aoclsparse_status_memory_error;
        }
        _aoclsparse_matrix tmp;
        tmp.m = m, tmp.n = n, tmp.nnz = nnz, tmp.base = descr->base, tmp.val_type = vt;
        tmp.user.m = m, tmp.user.n = n, tmp.user.nnz = nnz, tmp.user.base = descr->base;
        tmp.user.ptr = const_cast<aoclsparse_int *>(prow);
        tmp.user.ind = const_cast<aoclsparse_int *>(pcol);
        tmp.user.val = const_cast<T *>(pval);
        if(m == 0 || n == 0 || nnz == 0)
            return scale_y<T>(rt, y, n, *beta);
        DeviceCsr *dcsr = nullptr;
        SpmvPlan  *plan = nullptr;
        st              = ensure_spmv(&tmp, true, dcsr, plan);
        if(st != aoclsparse_status_success)
            return st;
        st = run_on_device_csr<T>(rt, -1, *dcsr, *plan, *alpha, x, *beta, y, m, n);
        if(st == aoclsparse_status_success) // tmp's device buffers die with it: drain first
            MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        return st;
    }

    if(m == 0)
        return aoclsparse_status_success;

    // non-transposed: make the three CSR arrays device-addressable
    Arg aval, acol, arow;
    st = arow.in(rt, 2, row, sizeof(aoclsparse_int) * ((size_t)m + 1), mdev, true);
    if(st == aoclsparse_status_success)
        st = acol.in(rt, 1, col, sizeof(aoclsparse_int) * (size_t)nnz, rt.is_device_pointer(col), true);
    if(st == aoclsparse_status_success)
        st = aval.in(rt, 0, val, sizeof(T) * (size_t)nnz, rt.is_device_pointer(val), true);
    if(st != aoclsparse_status_success)
        return st;

    SpmvPlan                  local;
    SpmvPlan                 *plan = &local;
    bool                      fresh_plan = false; // built in this call from this row_ptr
    std::shared_ptr<SpmvPlan> held; // keeps a cached plan alive for this call whatever other threads evict
    try
    {
        if(mdev)
        {
            {
                std::lock_guard<std::mutex> g(rt.lock);
                for(auto &e : g_raw)
                    if(e.key == row && e.m == m && e.nnz == nnz && e.base == descr->base && e.plan && e.plan->valid)
                        held = e.plan;
            }
            if(rt.plan_stale_host && *rt.plan_stale_host)
            {
                // an earlier product found a cached plan stale (it computed the right result from the live arrays): forget
                // every cached plan, they are rebuilt from the live row_ptr as their keys come back
                std::lock_guard<std::mutex> g(rt.lock);
                *rt.plan_stale_host = 0;
                for(auto &e : g_raw)
                    e = RawPlan();
                held.reset();
            }
            if(held && !rt.plan_stale_dev)
                held.reset(); // no pinned word to report a stale plan with: never trust the cache
            if(!held)
            {
                fresh_plan = true;
                std::vector<aoclsparse_int> hrow((size_t)m + 1);
                MI355_HIP_TRY(hipMemcpy(hrow.data(), row, sizeof(aoclsparse_int) * ((size_t)m + 1),
                                        hipMemcpyDeviceToHost));
                held = std::make_shared<SpmvPlan>();
                st   = build_spmv_plan(m, nnz, descr->base, hrow.data(), *held, sizeof(T));
                if(st != aoclsparse_status_success)
                    return st;
                std::lock_guard<std::mutex> g(rt.lock);
                RawPlan                    *slot = nullptr;
                for(auto &e : g_raw) // replace a stale entry of the same key in place, else round robin
                    if(e.key == row && e.m == m && e.nnz == nnz && e.base == descr->base)
                        slot = &e;
                if(!slot)
                {
                    slot       = &g_raw[g_raw_next];
                    g_raw_next = (g_raw_next + 1) % RAW_CACHE;
                }
                slot->plan = held;
                slot->key = row, slot->m = m, slot->nnz = nnz, slot->base = descr->base;
            }
            plan = held.get();
        }
        else
        {
            st = build_spmv_plan(m, nnz, descr->base, row, local, sizeof(T));
            if(st != aoclsparse_status_success)
                return st;
        }
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }

    DeviceCsr view; // non-owning view for the shared launcher
    view.m = m, view.n = n, view.nnz = nnz, view.base = descr->base;
    view.ptr.ptr = arow.dev, view.ind.ptr = acol.dev, view.val.ptr = aval.dev;
    // (a cached plan is validated inside the kernel; a plan built in this call from this row_ptr needs no validation)
    st = run_on_device_csr<T>(rt, -1, view, *plan, *alpha, x, *beta, y, n, m, mdev && fresh_plan == false ? rt.plan_stale_dev : nullptr);
    view.ptr.ptr = view.ind.ptr = view.val.ptr = nullptr; // not ours to free
    if(st == aoclsparse_status_success && plan == &local)
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream())); // local plan buffer is freed on return
    return st;
}

// level2/aoclsparse_dotmv.hpp:31-70: y = alpha*op(A)*x + beta*y, then d = x . y over min(m, n) entries
template <typename T>
aoclsparse_status dotmv_t(aoclsparse_operation op, T alpha, aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                          const T *x, T beta, T *y, T *d, aoclsparse_matrix_data_type vt)
{
    if(!d || !A)
        return aoclsparse_status_invalid_pointer;
    Runtime &rt = Runtime::get();
    // keep y on the device between the two steps when the caller's y is a host array
    aoclsparse_status st = mv_t<T>(op, &alpha, A, descr, x, &beta, y, vt);
    if(st != aoclsparse_status_success)
        return st;
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    const aoclsparse_int                  n = std::min(A->m, A->n);
    const bool xdev = rt.is_device_pointer(x), ydev = rt.is_device_pointer(y), ddev = rt.is_device_pointer(d);
    Arg        ax, ay;
    st = ax.in(rt, 3, x, sizeof(T) * (size_t)n, xdev, true);
    if(st == aoclsparse_status_success)
        st = ay.in(rt, 4, y, sizeof(T) * (size_t)n, ydev, true);
    void *part = nullptr, *dd = d;
    if(st == aoclsparse_status_success)
        st = rt.staging(6, sizeof(T) * 1024, &part);
    if(st == aoclsparse_status_success && !ddev)
        st = rt.staging(7, sizeof(T), &dd);
    if(st != aoclsparse_status_success)
        return st;
    st = launch_dot<T>(rt.stream(), n, static_cast<const T *>(ax.dev), static_cast<const T *>(ay.dev),
                       static_cast<T *>(part), static_cast<T *>(dd));
    if(st != aoclsparse_status_success)
        return st;
    if(!ddev)
    {
        MI355_HIP_TRY(hipMemcpyAsync(d, dd, sizeof(T), hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
    }
    return aoclsparse_status_success;
}

} // namespace

extern "C" {

aoclsparse_status aoclsparse_ddotmv(const aoclsparse_operation op, const double alpha, aoclsparse_matrix A,
                                    const aoclsparse_mat_descr descr, const double *x, const double beta,
                                    double *y, double *d)
{
    return dotmv_t<double>(op, alpha, A, descr, x, beta, y, d, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_sdotmv(const aoclsparse_operation op, const float alpha, aoclsparse_matrix A,
                                    const aoclsparse_mat_descr descr, const float *x, const float beta, float *y,
                                    float *d)
{
    return dotmv_t<float>(op, alpha, A, descr, x, beta, y, d, aoclsparse_smat);
}

aoclsparse_status aoclsparse_dcsrmv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                    aoclsparse_int n, aoclsparse_int nnz, const double *csr_val,
                                    const aoclsparse_int *csr_col_ind, const aoclsparse_int *csr_row_ptr,
                                    const aoclsparse_mat_descr descr, const double *x, const double *beta,
                                    double *y)
{
    return csrmv_t<double>(trans, alpha, m, n, nnz, csr_val, csr_col_ind, csr_row_ptr, descr, x, beta, y,
                           aoclsparse_dmat);
}

aoclsparse_status aoclsparse_scsrmv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                    aoclsparse_int n, aoclsparse_int nnz, const float *csr_val,
                                    const aoclsparse_int *csr_col_ind, const aoclsparse_int *csr_row_ptr,
                                    const aoclsparse_mat_descr descr, const float *x, const float *beta,
                                    float *y)
{
    return csrmv_t<float>(trans, alpha, m, n, nnz, csr_val, csr_col_ind, csr_row_ptr, descr, x, beta, y,
                          aoclsparse_smat);
}

aoclsparse_status aoclsparse_dmv(aoclsparse_operation op, const double *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr descr, const double *x, const double *beta,
                                 double *y)
{
    return mv_t<double>(op, alpha, A, descr, x, beta, y, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_smv(aoclsparse_operation op, const float *alpha, aoclsparse_matrix A,
                                 const aoclsparse_mat_descr descr, const float *x, const float *beta,
                                 float *y)
{
    return mv_t<float>(op, alpha, A, descr, x, beta, y, aoclsparse_smat);
}

aoclsparse_status aoclsparse_mi355_get_spmv_info(const aoclsparse_matrix A, aoclsparse_operation op,
                                                 aoclsparse_mi355_spmv_info *info)
{
    if(!A || !info)
        return aoclsparse_status_invalid_pointer;
    std::shared_lock<std::shared_mutex> r(A->guard);
    const bool                          tr = op != aoclsparse_operation_none;
    const SpmvPlan                     &p  = tr ? A->plan_trans : A->plan_user;
    const DeviceCsr                    &d  = tr ? A->dev_trans : A->dev_user;
    std::memset(info, 0, sizeof(*info));
    info->device_resident = d.valid;
    if(!p.valid)
        return aoclsparse_status_success;
    info->kernel      = p.sell.valid ? (p.sell.shared ? 4 : 3) : (p.merge.valid ? 2 : 1);
    info->row_blocks  = p.sell.valid ? (p.sell.nslices < 2048 ? p.sell.nslices : (p.sell.nslices + 1) / 2) : p.nblocks;
    info->tile        = p.tile & ~1;
    info->sell_slices  = p.sell.valid ? p.sell.nslices : 0;
    info->stored_cells = p.sell.valid ? p.sell.cells : 0;
    info->mm_groups    = p.mm.valid ? p.mm.ngroups : 0;
    info->mm_window_rows = p.mm.win ? p.mm.win_rows : 0;
    info->mm_bell_width  = p.bell.valid ? p.bell.width : 0;
    info->mm_bell_fill_permille = p.bell.valid ? (aoclsparse_int)(p.bell.fill * 1000.0 + 0.5) : 0;
    info->mm_bell_xcd_chunk     = p.bell.valid ? p.bell.xcd_chunk : 0;
    info->mm_bell_lattice_line  = p.bell.valid ? p.bell.lattice[0] : 0;
    info->mm_bell_lattice_lines = p.bell.valid ? p.bell.lattice[1] : 0;
    info->mm_bell_region_a      = p.bell.valid ? p.bell.region[0] : 0;
    info->mm_bell_region_b      = p.bell.valid ? p.bell.region[1] : 0;
    info->mm_bell_model_fetches_permille              = p.bell.valid ? (aoclsparse_int)(p.bell.model_fetches * 1000.0 + 0.5) : 0;
    info->mm_bell_model_fetches_launch_order_permille = p.bell.valid ? (aoclsparse_int)(p.bell.model_fetches_launch_order * 1000.0 + 0.5) : 0;
    info->long_rows   = p.long_rows;
    info->max_row_nnz = p.max_row_nnz;
    aoclsparse_int kid = -1;
    for(const Hint &h : A->hints)
        if(h.act == action_mv && h.id == (tr ? doid::gt : doid::gn))
        {
            kid = h.kid;
            break;
        }
    int  order  = 0;
    bool strict = false;
    if(A->val_type == aoclsparse_smat)
        resolve_order<float>(kid, d.m, d.nnz, p.max_row_nnz, order, strict);
    else
        resolve_order<double>(kid, d.m, d.nnz, p.max_row_nnz, order, strict);
    info->order = order;
    if(plan_option(aoclsparse_mi355_option_spmv_strict) == 1)
        strict = true;
    info->tree_min = ((info->kernel == 1 || info->kernel == 2) && order == 0 && !strict) ? SPMV_TREE_MIN : 0;
    return aoclsparse_status_success;
}

} // extern "C"
