// matrix.cpp -- descriptor + handle lifecycle, clean-CSR analysis, hints/optimize, device mirrors.
//
// Behaviour follows the reference (cited per function, paths relative to
// /root/reference/library/src); the data layout behind the opaque handles is our own.
#include "internal.hpp"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

using namespace mi355;

namespace mi355
{

doid get_doid(const _aoclsparse_mat_descr *d, aoclsparse_operation op)
{
    // include/aoclsparse_mtx_dispatcher.hpp:79-143 for real types (conjugate == transpose)
    const bool tr = op != aoclsparse_operation_none;
    switch(d->type)
    {
    case aoclsparse_matrix_type_general:
        return tr ? doid::gt : doid::gn;
    case aoclsparse_matrix_type_symmetric:
    case aoclsparse_matrix_type_hermitian:
        return d->fill_mode == aoclsparse_fill_mode_lower ? doid::sl : doid::su;
    case aoclsparse_matrix_type_triangular:
        if(d->fill_mode == aoclsparse_fill_mode_lower)
            return tr ? doid::tlt : doid::tln;
        return tr ? doid::tut : doid::tun;
    }
    return doid::len;
}

// ---- validity / sort class / full diagonal: analysis/aoclsparse_csr_util.cpp:124-279 ---------
aoclsparse_status mat_check(aoclsparse_int maj, aoclsparse_int mind, aoclsparse_int nnz,
                            const aoclsparse_int *ptr, const aoclsparse_int *ind, const void *val,
                            int shape, aoclsparse_index_base base, int &sort, bool &fulldiag)
{
    if(!ptr || !ind || !val)
        return aoclsparse_status_invalid_pointer;
    if(mind < 0 || maj < 0 || nnz < 0)
        return aoclsparse_status_invalid_size;
    if(ptr[0] != base || ptr[maj] - base != nnz)
        return aoclsparse_status_invalid_value;
    for(aoclsparse_int i = 0; i < maj; i++)
        if(ptr[i] > ptr[i + 1])
            return aoclsparse_status_invalid_value;

    // Rows are independent (the sort class is the worst class of any row, the diagonal flag an AND, and the reference
    // returns the error of the FIRST offending row): chunks of rows in parallel, merged in row order.  52 M entries: 50 -> a few ms
    // on the GPU box's host (round 3: this pass, check_sort_diag and csr_indices were half of aoclsparse_optimize's time for
    // an sv hint).
    std::atomic<int>       cls_all{1}; // fully sorted until proven otherwise
    std::atomic<bool>      full_all{true};
    std::atomic<long long> first_err{LLONG_MAX}; // (row << 8) | status of the earliest offending row
    parallel_for(maj, 1 << 15, [&](long long i0, long long i1) {
        int  cls  = 1;
        bool full = true;
        for(aoclsparse_int i = (aoclsparse_int)i0; i < (aoclsparse_int)i1; i++)
        {
            const aoclsparse_int lo = shape == 2 ? i : 0;
            const aoclsparse_int hi = shape == 1 ? i : mind - 1;
            bool           seen_diag = false, seen_upper = false;
            aoclsparse_int prev = -1;
            int            err  = 0;
            for(aoclsparse_int p = ptr[i] - base; p < ptr[i + 1] - base && !err; p++)
            {
                const aoclsparse_int j = ind[p] - base;
                if(j < lo || j > hi)
                {
                    err = (int)aoclsparse_status_invalid_index_value;
                    break;
                }
                if(cls != 3)
                {
                    if(prev > j)
                        cls = 2; // order inside a group broken: partially sorted
                    else
                        prev = j;
                    if((j <= i && seen_upper) || (j < i && seen_diag))
                        cls = 3; // L | D | U group order broken: unsorted
                }
                if(j > i)
                    seen_upper = true;
                else if(j == i)
                {
                    if(seen_diag)
                        err = (int)aoclsparse_status_invalid_value; // duplicate diagonal
                    seen_diag = true;
                }
            }
            if(err)
            {
                const long long key = ((long long)i << 8) | err;
                long long       cur = first_err.load();
                while(key < cur && !first_err.compare_exchange_weak(cur, key))
                {
                }
                return; // later rows of this chunk cannot be the first offender
            }
            if(!seen_diag && i < mind)
                full = false;
        }
        int c0 = cls_all.load();
        while(cls > c0 && !cls_all.compare_exchange_weak(c0, cls))
        {
        }
        if(!full)
            full_all.store(false);
    });
    if(first_err.load() != LLONG_MAX)
        return (aoclsparse_status)(first_err.load() & 0xff);
    const int  cls  = cls_all.load();
    const bool full = full_all.load();
    sort     = cls;
    fulldiag = full;
    return aoclsparse_status_success;
}

// ---- group order (L | D | U) + diagonal presence: csr_util.cpp:290-364 -------------------------
aoclsparse_status check_sort_diag(aoclsparse_int m, aoclsparse_int n, aoclsparse_index_base base,
                                  const aoclsparse_int *ptr, const aoclsparse_int *ind, bool &sorted,
                                  bool &fulldiag)
{
    sorted = fulldiag = false;
    if(m < 0 || n < 0)
        return aoclsparse_status_invalid_size;
    if(!ptr || !ind)
        return aoclsparse_status_invalid_pointer;
    // the reference stops at the first row that is out of order (sorted = fulldiag = false) or holds two diagonals
    // (invalid_value): rows are scanned in parallel chunks and the earliest such row decides
    std::atomic<long long> first_evt{LLONG_MAX}; // (row << 1) | (1: duplicate diagonal, 0: unsorted)
    std::atomic<bool>      full_all{true};
    parallel_for(m, 1 << 15, [&](long long i0, long long i1) {
        bool full = true;
        for(aoclsparse_int i = (aoclsparse_int)i0; i < (aoclsparse_int)i1; i++)
        {
            bool in_lower = true, have_diag = false, ok = true;
            int  evt = -1;
            for(aoclsparse_int p = ptr[i] - base; p < ptr[i + 1] - base; p++)
            {
                const aoclsparse_int j = ind[p] - base;
                if(j == i)
                {
                    if(have_diag)
                    {
                        evt = 1;
                        break;
                    }
                    have_diag = true;
                    ok        = in_lower;
                    in_lower  = false;
                }
                else if(in_lower)
                    in_lower = j < i;
                else
                    ok = ok && j > i;
                if(!ok)
                {
                    evt = 0;
                    break;
                }
            }
            if(evt >= 0)
            {
                const long long key = ((long long)i << 1) | evt;
                long long       cur = first_evt.load();
                while(key < cur && !first_evt.compare_exchange_weak(cur, key))
                {
                }
                return;
            }
            if(!have_diag && i < n)
                full = false;
        }
        if(!full)
            full_all.store(false);
    });
    if(first_evt.load() != LLONG_MAX)
    {
        sorted = fulldiag = false;
        return (first_evt.load() & 1) ? aoclsparse_status_invalid_value : aoclsparse_status_success;
    }
    sorted   = true;
    fulldiag = full_all.load();
    return aoclsparse_status_success;
}

// ---- idiag / iurow in the matrix's base: csr_util.cpp:389-458 ----------------------------------
aoclsparse_status csr_indices(aoclsparse_int m, aoclsparse_index_base base,
                              const aoclsparse_int *ptr, const aoclsparse_int *ind,
                              aoclsparse_int **idiag, aoclsparse_int **iurow)
{
    if(m < 0)
        return aoclsparse_status_invalid_size;
    if(!ptr || !ind || !idiag || !iurow)
        return aoclsparse_status_invalid_pointer;
    aoclsparse_int *d = new(std::nothrow) aoclsparse_int[m > 0 ? m : 1];
    aoclsparse_int *u = new(std::nothrow) aoclsparse_int[m > 0 ? m : 1];
    if(!d || !u)
    {
        delete[] d;
        delete[] u;
        return aoclsparse_status_memory_error;
    }
    parallel_for(m, 1 << 15, [&](long long i0, long long i1) {
        for(aoclsparse_int i = (aoclsparse_int)i0; i < (aoclsparse_int)i1; i++)
        {
            const aoclsparse_int e = ptr[i + 1] - base;
            aoclsparse_int       p = ptr[i] - base;
            while(p < e && ind[p] - base < i)
                p++;
            // p: first entry at or right of the diagonal (or row end); positions keep the base
            d[i] = p + base;
            u[i] = (p < e && ind[p] - base == i) ? p + base + 1 : p + base;
        }
    });
    *idiag = d;
    *iurow = u;
    return aoclsparse_status_success;
}

template <typename T>
static aoclsparse_status make_clean_copy(aoclsparse_matrix A, bool sorted, bool &fulldiag,
                                         std::unique_ptr<HostCsr> &out)
{
    // csr_util.hpp:875-951: 0-based copy, per-row sort (csr_util.hpp:100-159), explicit zero
    // diagonals for rows i < n (csr_util.hpp:167-279).
    const HostCsr       &u    = A->user;
    const aoclsparse_int m    = u.m, n = u.n, nnz = A->nnz, b = u.base;
    const T             *uval = static_cast<const T *>(u.val);
    std::vector<aoclsparse_int> tptr(m + 1), tind(nnz);
    std::vector<T>              tval(nnz);
    for(aoclsparse_int i = 0; i <= m; i++)
        tptr[i] = u.ptr[i] - b;
    for(aoclsparse_int p = 0; p < nnz; p++)
    {
        tind[p] = u.ind[p] - b;
        tval[p] = uval[p];
    }
    if(!sorted)
    {
        std::vector<aoclsparse_int> perm;
        for(aoclsparse_int i = 0; i < m; i++)
        {
            const aoclsparse_int s = tptr[i], len = tptr[i + 1] - s;
            perm.resize(len);
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(), [&](aoclsparse_int a, aoclsparse_int c) {
                return u.ind[s + a] < u.ind[s + c];
            });
            for(aoclsparse_int k = 0; k < len; k++)
            {
                tind[s + k] = u.ind[s + perm[k]] - b;
                tval[s + k] = uval[s + perm[k]];
            }
        }
        bool              s2;
        aoclsparse_status st = check_sort_diag(m, n, aoclsparse_index_base_zero, tptr.data(),
                                               tind.data(), s2, fulldiag);
        if(st != aoclsparse_status_success)
            return st;
    }
    aoclsparse_int missing = 0;
    if(!fulldiag)
        for(aoclsparse_int i = 0; i < std::min(m, n); i++)
        {
            bool have = false;
            for(aoclsparse_int p = tptr[i]; p < tptr[i + 1] && !have; p++)
                have = tind[p] == i;
            missing += !have;
        }
    std::unique_ptr<HostCsr> c(new HostCsr);
    c->m = m, c->n = n, c->nnz = nnz + missing, c->base = aoclsparse_index_base_zero;
    c->owned = true;
    c->ptr   = new aoclsparse_int[m + 1];
    c->ind   = new aoclsparse_int[c->nnz > 0 ? c->nnz : 1];
    c->val   = ::operator new(sizeof(T) * (c->nnz > 0 ? c->nnz : 1));
    T             *cval = static_cast<T *>(c->val);
    aoclsparse_int w    = 0;
    for(aoclsparse_int i = 0; i < m; i++)
    {
        c->ptr[i]   = w;
        bool placed = fulldiag || i >= n;
        for(aoclsparse_int p = tptr[i]; p < tptr[i + 1]; p++)
        {
            if(!placed && tind[p] >= i)
            {
                if(tind[p] != i)
                {
                    c->ind[w] = i;
                    cval[w++] = T(0);
                }
                placed = true;
            }
            c->ind[w] = tind[p];
            cval[w++] = tval[p];
        }
        if(!placed)
        {
            c->ind[w] = i;
            cval[w++] = T(0);
        }
    }
    c->ptr[m] = w;
    c->nnz    = w;
    out       = std::move(c);
    return aoclsparse_status_success;
}

// ---- clean CSR, analysis/aoclsparse_csr_util.hpp:765-967 -----------------------------------------
aoclsparse_status csr_optimize(aoclsparse_matrix A)
{
    if(!A)
        return aoclsparse_status_invalid_pointer;
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if(A->opt)
            return aoclsparse_status_success;
    }
    std::unique_lock<std::shared_mutex> w(A->guard);
    if(A->opt)
        return aoclsparse_status_success;
    HostCsr &u = A->user;
    if(!u.ptr || !u.ind || !u.val)
        return aoclsparse_status_invalid_pointer;
    try
    {
        aoclsparse_status st
            = mat_check(u.m, u.n, A->nnz, u.ptr, u.ind, u.val, 0, u.base, A->sort, A->fulldiag);
        if(st != aoclsparse_status_success)
            return st;
        bool sorted, fulldiag;
        st = check_sort_diag(u.m, u.n, u.base, u.ptr, u.ind, sorted, fulldiag);
        if(st != aoclsparse_status_success)
            return aoclsparse_status_internal_error;
        if(sorted && fulldiag)
        {
            // already clean: keep using the caller's memory, only add idiag / iurow
            st = csr_indices(u.m, u.base, u.ptr, u.ind, &u.idiag, &u.iurow);
            if(st != aoclsparse_status_success)
                return st;
            u.is_optimized = true;
            A->opt         = &u;
        }
        else
        {
            std::unique_ptr<HostCsr> c;
            st = dispatch_value_type(A->val_type, [&](auto tag) {
                return make_clean_copy<decltype(tag)>(A, sorted, fulldiag, c);
            });
            if(st != aoclsparse_status_success)
                return st;
            st = csr_indices(c->m, c->base, c->ptr, c->ind, &c->idiag, &c->iurow);
            if(st != aoclsparse_status_success)
                return st;
            c->is_optimized = true;
            A->opt_copy     = std::move(c);
            A->opt          = A->opt_copy.get();
        }
        A->opt_csr_full_diag = fulldiag;
        A->optimized         = true;
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

// ---- transpose of the user CSR, conversion/aoclsparse_convert.hpp:552-655 (counting sort) ---------
template <typename T>
static void transpose_into(const HostCsr &u, aoclsparse_int nnz, HostCsr &t)
{
    const aoclsparse_int m = u.m, n = u.n, b = u.base;
    const T             *uv = static_cast<const T *>(u.val);
    T                   *tv = static_cast<T *>(t.val);
    std::fill(t.ptr, t.ptr + n + 1, 0);
    for(aoclsparse_int p = 0; p < nnz; p++)
        t.ptr[u.ind[p] - b + 1]++;
    for(aoclsparse_int j = 0; j < n; j++)
        t.ptr[j + 1] += t.ptr[j];
    std::vector<aoclsparse_int> next(t.ptr, t.ptr + n);
    for(aoclsparse_int i = 0; i < m; i++)
        for(aoclsparse_int p = u.ptr[i] - b; p < u.ptr[i + 1] - b; p++)
        {
            const aoclsparse_int q = next[u.ind[p] - b]++;
            t.ind[q]               = i;
            tv[q]                  = uv[p];
        }
}

// from this many entries on the handles' transposes are sorted on the device
constexpr aoclsparse_int DEVICE_TRANSPOSE_MIN_NNZ = 1 << 20;

aoclsparse_status build_transpose(aoclsparse_matrix A)
{
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if(A->trans)
            return aoclsparse_status_success;
    }
    // (the device sort takes its temporaries from the staging slots: the stage lock comes BEFORE the handle's guard, as everywhere)
    std::lock_guard<std::recursive_mutex> sl(Runtime::get().stage_lock);
    std::unique_lock<std::shared_mutex>   w(A->guard);
    if(A->trans)
        return aoclsparse_status_success;
    try
    {
        std::unique_ptr<HostCsr> t(new HostCsr);
        const size_t             vs = val_size(A->val_type);
        const size_t             nz = (size_t)(A->nnz > 0 ? A->nnz : 1);
        t->m = A->n, t->n = A->m, t->nnz = A->nnz, t->base = aoclsparse_index_base_zero;
        t->owned = t->result_arrays = true;
        t->ptr   = new aoclsparse_int[t->m + 1];
        t->ind   = static_cast<aoclsparse_int *>(host_result_alloc(sizeof(aoclsparse_int) * nz));
        t->val   = host_result_alloc(vs * nz);
        if(!t->ind || !t->val)
            return aoclsparse_status_memory_error;
        // Large matrices: the sort runs on the device (transpose_kernels.hip: same order, 22 -> ~3 ms for 5 M entries) and leaves
        // the transpose in HBM as the handle's dev_trans; the host copy every analysis reads follows over PCIe.  Small ones, and
        // matrices with a column of more than 2,048 entries, take the host sort.
        bool on_device = false;
        if(A->nnz >= DEVICE_TRANSPOSE_MIN_NNZ && A->user.ptr[A->m] - A->base == A->nnz)
        {
            Runtime          &rt = Runtime::get();
            aoclsparse_status st = rt.init();
            if(st == aoclsparse_status_success && !A->dev_user.valid)
                st = upload_csr(A->user, vs, A->dev_user);
            DeviceCsr &dt = A->dev_trans;
            if(st == aoclsparse_status_success) // (dt.ptr / ind / val are allocated by the call once the matrix is accepted, and
                                                // released again on decline or failure: nothing is held while the host sorts)
                st = device_transpose(rt.stream(), A->m, A->n, A->nnz, A->base, A->dev_user.ptr.as<aoclsparse_int>(),
                                      A->dev_user.ind.as<aoclsparse_int>(), A->dev_user.val.ptr, vs, dt.ptr, dt.ind, dt.val);
            if(st == aoclsparse_status_success)
            {
                host_result_touch(t->ind, sizeof(aoclsparse_int) * nz);
                host_result_touch(t->val, vs * nz);
                hipStream_t s = rt.stream();
                if(hipMemcpyAsync(t->ptr, dt.ptr.ptr, sizeof(aoclsparse_int) * ((size_t)t->m + 1), hipMemcpyDeviceToHost, s) == hipSuccess
                   && hipMemcpyAsync(t->ind, dt.ind.ptr, sizeof(aoclsparse_int) * (size_t)A->nnz, hipMemcpyDeviceToHost, s) == hipSuccess
                   && hipMemcpyAsync(t->val, dt.val.ptr, vs * (size_t)A->nnz, hipMemcpyDeviceToHost, s) == hipSuccess
                   && hipStreamSynchronize(s) == hipSuccess)
                {
                    dt.m = t->m, dt.n = t->n, dt.nnz = A->nnz, dt.base = aoclsparse_index_base_zero;
                    dt.valid  = true;
                    on_device = true;
                }
            }
            if(!on_device)
            {
                (void)hipGetLastError(); // (declined or failed: the host sort below serves the handle)
                dt.ptr.release(), dt.ind.release(), dt.val.release();
            }
        }
        if(!on_device)
            dispatch_value_type(A->val_type, [&](auto tag) {
                transpose_into<decltype(tag)>(A->user, A->nnz, *t);
                return 0;
            });
        A->trans = std::move(t);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

// ---- device mirrors ---------------------------------------------------------------------------------
aoclsparse_status upload_csr(const HostCsr &h, size_t vsize, DeviceCsr &d)
{
    Runtime          &rt = Runtime::get();
    const aoclsparse_int nnz = h.ptr[h.m] - h.base;
    aoclsparse_status st = d.ptr.upload(h.ptr, sizeof(aoclsparse_int) * (size_t)(h.m + 1), rt.stream());
    if(st == aoclsparse_status_success)
        st = d.ind.upload(h.ind, sizeof(aoclsparse_int) * (size_t)nnz, rt.stream());
    if(st == aoclsparse_status_success)
        st = d.val.upload(h.val, vsize * (size_t)nnz, rt.stream());
    if(st != aoclsparse_status_success)
        return st;
    d.m = h.m, d.n = h.n, d.nnz = nnz, d.base = h.base;
    d.valid = true;
    return aoclsparse_status_success;
}

// CSR-Adaptive row blocks (host, O(m)): consecutive rows are packed into a block while their
// non-zeros fit one LDS tile and the row count stays <= spmv_maxrows(tile); a row longer than a tile gets a
// block of its own.  Entry b = {first row, first non-zero (0-based)}; entry nb closes the last block.
static aoclsparse_int plan_rows(aoclsparse_int m, aoclsparse_index_base base, aoclsparse_int tile,
                                const aoclsparse_int *row_ptr, aoclsparse_int *blocks,
                                aoclsparse_int *long_rows, aoclsparse_int *max_row)
{
    aoclsparse_int nb = 0, lr = 0, mx = 0, i = 0;
    while(i < m)
    {
        const aoclsparse_int start = row_ptr[i] - base;
        aoclsparse_int       j     = i;
        while(j < m && j - i < spmv_maxrows(tile) && (row_ptr[j + 1] - base) - start <= tile)
            j++;
        if(j == i) // single row longer than a tile
        {
            j = i + 1;
            lr++;
        }
        for(aoclsparse_int r = i; r < j; r++)
            mx = std::max(mx, row_ptr[r + 1] - row_ptr[r]);
        blocks[2 * nb]     = i;
        blocks[2 * nb + 1] = start;
        nb++;
        i = j;
    }
    blocks[2 * nb]     = m;
    blocks[2 * nb + 1] = row_ptr[m] - base;
    if(long_rows)
        *long_rows = lr;
    if(max_row)
        *max_row = mx;
    return nb;
}

static aoclsparse_int choose_tile(aoclsparse_int /*m*/, aoclsparse_int nnz, size_t vsize)
{
    // float, large: 2048 entries per block (the LDS footprint of 1024 doubles; a float load instruction moves half the bytes, so a
    // block needs twice the entries in flight: raw scsrmv on the 4096^2 Laplacian 0.216 / 0.202 / 0.182 ms at 512 / 1024 / 2048,
    // double 0.267 / 0.261 / 0.284 on the same box -- profiles/r4/float_headline.txt)
    if(vsize == 4 && (long long)nnz >= 1024LL * 256 * 16)
        return 2048;
    // 1024: 18 KiB of LDS per workgroup -> 8 workgroups (32 wavefronts) per CU hide the
    // load -> gather -> reduce latency chain better than 4 fatter ones (0.257 vs 0.290 ms on the 4096^2
    // Laplacian).  A matrix too small to give every CU ~16 such blocks gets 512-entry tiles and 128-lane
    // workgroups instead (web-like stand-in: 0.039 vs 0.056 ms).  profiles/r1, DESIGN.md 5.1.
    // (the XCD-contiguous block order -- bit 0 of the tile word, which the kernel still decodes -- measured slower than launch
    // order on the Laplacian and 6 % faster at most on the graph stand-ins: profiles/r3/irregular_locality_pmc.txt; the round 1-3
    // switches that forced a tile size or that order are gone, the losing sides are recorded under profiles/)
    // (diagnostics, counter passes of round 6: AOCLSPARSE_MI355_SPMV_XCD_ORDER=1 sets that bit; profiles/r6/csrmv_xcd_order_pmc.txt)
    static const bool xcd_order = getenv("AOCLSPARSE_MI355_SPMV_XCD_ORDER") && getenv("AOCLSPARSE_MI355_SPMV_XCD_ORDER")[0] == '1';
    return ((long long)nnz < 1024LL * 256 * 16 ? 512 : 1024) | (xcd_order ? 1 : 0);
}

// plan_rows on the row range [r0, r1): block entries are appended to `out` (no terminal entry)
static void plan_rows_range(aoclsparse_int r0, aoclsparse_int r1, aoclsparse_index_base base, aoclsparse_int tile,
                            const aoclsparse_int *row_ptr, std::vector<aoclsparse_int> &out, aoclsparse_int &lr,
                            aoclsparse_int &mx)
{
    aoclsparse_int i = r0;
    while(i < r1)
    {
        const aoclsparse_int start = row_ptr[i] - base;
        aoclsparse_int       j     = i;
        while(j < r1 && j - i < spmv_maxrows(tile) && (row_ptr[j + 1] - base) - start <= tile)
            j++;
        if(j == i) // single row longer than a tile
        {
            j = i + 1;
            lr++;
        }
        for(aoclsparse_int r = i; r < j; r++)
            mx = std::max(mx, row_ptr[r + 1] - row_ptr[r]);
        out.push_back(i);
        out.push_back(start);
        i = j;
    }
}

aoclsparse_status build_spmv_plan(aoclsparse_int m, aoclsparse_int nnz, aoclsparse_index_base base,
                                  const aoclsparse_int *row_ptr_host, SpmvPlan &plan, size_t vsize)
{
    try
    {
        // Big matrices: the greedy packing runs on 16 fixed row chunks in parallel (the chunking does not depend on
        // the thread count, so the plan is reproducible; a chunk edge merely ends a block early).  The one-shot raw
        // aoclsparse_?csrmv on host arrays builds a plan per call: a sequential pass over 16.8 M rows into a zeroed
        // 2(m+2)-int buffer cost ~50 ms there, more than sending the matrix over PCIe (profiles/r2/h2d_probe.jsonl).
        plan.tile                 = choose_tile(m, nnz, vsize);
        const aoclsparse_int tile = plan.tile & ~1;
        const int            nchunks = m >= (1 << 20) ? 16 : 1;
        std::vector<std::vector<aoclsparse_int>> part(nchunks);
        std::vector<aoclsparse_int>              lrs(nchunks, 0), mxs(nchunks, 0);
        auto work = [&](int c) {
            const aoclsparse_int r0 = (aoclsparse_int)((long long)m * c / nchunks);
            const aoclsparse_int r1 = (aoclsparse_int)((long long)m * (c + 1) / nchunks);
            part[c].reserve((size_t)(r1 - r0) / 64 + 16);
            plan_rows_range(r0, r1, base, tile, row_ptr_host, part[c], lrs[c], mxs[c]);
        };
        // parallel_for (internal.hpp) carries a worker's exception (bad_alloc inside reserve / push_back) to this thread and
        // runs a chunk inline when a thread cannot be started, so nothing can reach std::terminate below the C ABI
        // (ADVICE r2); the 16 chunks are fixed, so the plan does not depend on how many threads took them.
        mi355::parallel_for(nchunks, 1, [&](long long c0, long long c1) {
            for(long long c = c0; c < c1; c++)
                work((int)c);
        });
        std::vector<aoclsparse_int> blk;
        size_t                      total = 2;
        for(auto &p : part)
            total += p.size();
        blk.reserve(total);
        plan.long_rows = 0, plan.max_row_nnz = 0;
        for(int c = 0; c < nchunks; c++)
        {
            blk.insert(blk.end(), part[c].begin(), part[c].end());
            plan.long_rows += lrs[c];
            plan.max_row_nnz = std::max(plan.max_row_nnz, mxs[c]);
        }
        plan.nblocks = (aoclsparse_int)(blk.size() / 2);
        blk.push_back(m);
        blk.push_back(row_ptr_host[m] - base);
        aoclsparse_status st = plan.rowblocks.upload(
            blk.data(), sizeof(aoclsparse_int) * 2 * (size_t)(plan.nblocks + 1), Runtime::get().stream());
        if(st != aoclsparse_status_success)
            return st;
        // Heavy blocks first.  A row is ONE lane's serial chain (scalar order), so a block that holds a row of a few
        // hundred entries -- or is a single row longer than a tile -- runs 2-4 x as long as the others; workgroups start
        // over 2.5-4.5 us (circuit-like: 1,900 of them), and a heavy block started last is the kernel's tail
        // (tools/spmv_trace.py).  Same blocks, same rows, same chains: only the workgroup that takes them changes.
        plan.heavy_first = false;
        if(plan.nblocks >= 512 && plan.max_row_nnz >= 64)
        {
            const aoclsparse_int        nb = plan.nblocks;
            std::vector<aoclsparse_int> weight((size_t)nb), order((size_t)nb);
            aoclsparse_int              heavy = 0;
            for(aoclsparse_int b = 0; b < nb; b++)
            {
                aoclsparse_int w = 0;
                for(aoclsparse_int r = blk[2 * b]; r < blk[2 * b + 2]; r++)
                    w = std::max(w, row_ptr_host[r + 1] - row_ptr_host[r]);
                weight[b] = w >= 64 ? w : 0;
                heavy += w >= 64;
                order[b] = b;
            }
            if(heavy > 0) // (circuit-like: most blocks hold a row >= 64; the 300-entry ones still have to go first)
            {
                std::stable_sort(order.begin(), order.end(), [&](aoclsparse_int a, aoclsparse_int c) { return weight[a] > weight[c]; });
                // (Tried: the heaviest eighth on the XCD a lone launch reaches first -- workgroup i runs on XCD i % 8 and a
                // synchronous launch reaches the XCDs staggered by up to 4.5 us, tools/spmv_trace.py -- the traced span fell
                // 13.8 -> 11.0 us, but back-to-back calls got SLOWER (14.8 vs 13.6 us, web-like 37 vs 28 us): in a stream of
                // launches the stagger is not there and one XCD ends up with all the long chains.)
                std::vector<aoclsparse_int> b4(4 * (size_t)nb);
                for(aoclsparse_int k = 0; k < nb; k++)
                {
                    const aoclsparse_int b = order[k];
                    b4[4 * k] = blk[2 * b], b4[4 * k + 1] = blk[2 * b + 1];
                    b4[4 * k + 2] = blk[2 * b + 2] - blk[2 * b], b4[4 * k + 3] = blk[2 * b + 3] - blk[2 * b + 1];
                }
                st = plan.rowblocks4.upload(b4.data(), sizeof(aoclsparse_int) * b4.size(), Runtime::get().stream());
                if(st != aoclsparse_status_success)
                    return st;
                plan.heavy_first = true;
            }
        }
        plan.valid = true;
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    catch(const std::exception &)
    {
        return aoclsparse_status_internal_error;
    }
    return aoclsparse_status_success;
}

aoclsparse_status build_sell(const aoclsparse_int *row_ptr_host, const DeviceCsr &d, size_t vsize, SpmvPlan &plan, bool complex_values)
{
    SellPlan &sp = plan.sell;
    if(sp.valid || sp.tried)
        return aoclsparse_status_success;
    sp.tried = true;
    const int mode = plan_option(aoclsparse_mi355_option_sell); // -1 automatic (default), 0 never, 1 whatever the padding
    if(mode == 0 || d.m <= 0 || d.nnz <= 0 || !d.valid)
        return aoclsparse_status_success;
    const aoclsparse_int   m = d.m, nslices = (m + 63) / 64;
    std::vector<long long> sptr;
    try
    {
        sptr.resize((size_t)nslices + 1);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    // PACK 4 (four consecutive cells of a row adjacent, width rounded up to a multiple of 4) for long rows:
    // contiguous 2 KB wavefront loads; PACK 1 otherwise (a 5-wide slice must not be padded to 8)
    // (complex values: PACK 1 only -- their kernels are the scalar-chain ones)
    const int pack = (!complex_values && (long long)d.nnz >= 16LL * m) ? 4 : 1;
    sptr[0] = 0;
    for(aoclsparse_int s = 0; s < nslices; s++)
    {
        aoclsparse_int w = 0;
        for(aoclsparse_int i = s * 64; i < std::min<aoclsparse_int>(m, s * 64 + 64); i++)
            w = std::max(w, row_ptr_host[i + 1] - row_ptr_host[i]);
        w           = (w + pack - 1) / pack * pack;
        sptr[s + 1] = sptr[s] + 64LL * w;
    }
    const long long cells = sptr[nslices];
    // Padding budget: 1.35 cells per non-zero.  Round 1 set 1.15 from the two kernels' rates then (0.78 vs 0.66 of peak); the SELL
    // kernel has gained since.  Round-3 measurement on the unstructured flan-like variant (padding 1.21: tools/history/exp_r3_sellpad.sh,
    // profiles/r3/sell_padding_budget.txt): SELL-64 0.259 ms vs CSR-Adaptive 0.352 ms, i.e. break-even near 1.21 * 0.352 / 0.259 =
    // 1.64 cells per non-zero; 1.35 keeps a margin for matrices with shorter rows.
    if(mode != 1 && (double)cells > 1.35 * (double)d.nnz + 64.0)
        return aoclsparse_status_success;
    Runtime          &rt = Runtime::get();
    aoclsparse_status st = sp.slice_ptr.upload(sptr.data(), sizeof(long long) * sptr.size(), rt.stream());
    // Shared column lists: rows that repeat the list of the row before them -- as it is (the dofs of a mesh node) or
    // shifted by one (the rows of a stencil) -- keep ONE copy per slice.  Leaders are found on the device (one compare
    // pass over the CSR arrays); used when the column stream shrinks to <= 70 %.
    sp.shared = false, sp.ccells = cells;
    std::vector<long long> cptr;
    // (not for matrices that live in the caches anyway: the leader / shift words are one more dependent load, and the 10k x 10k
    // Laplacian of BASELINE configs[0] -- 50 k non-zeros, launch-bound -- ran at 5.4 instead of 4.6 us per call with them)
    if(st == aoclsparse_status_success && d.nnz >= (1 << 17))
    {
        DeviceBuffer nl;
        st = sp.lead.alloc(sizeof(unsigned short) * (size_t)m);
        if(st == aoclsparse_status_success)
            st = nl.alloc(sizeof(aoclsparse_int) * (size_t)nslices);
        if(st == aoclsparse_status_success)
            st = launch_sell_leaders(rt.stream(), m, d.base, d.ptr.as<aoclsparse_int>(), d.ind.as<aoclsparse_int>(), nslices,
                                     sp.lead.as<unsigned short>(), nl.as<aoclsparse_int>());
        if(st != aoclsparse_status_success)
            return st;
        std::vector<aoclsparse_int> nlh((size_t)nslices);
        MI355_HIP_TRY(hipMemcpyAsync(nlh.data(), nl.ptr, sizeof(aoclsparse_int) * (size_t)nslices, hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        cptr.resize((size_t)nslices + 1);
        cptr[0] = 0;
        for(aoclsparse_int s = 0; s < nslices; s++)
            cptr[s + 1] = cptr[s] + (long long)(nlh[s] & 0xff) * ((sptr[s + 1] - sptr[s]) >> 6);
        const long long ctotal = cptr[nslices];
        if(PhaseTimer::on())
            std::fprintf(stderr, "[mi355 timing] sell: %lld cells, %lld column cells with shared lists (%d slices)\n", cells, ctotal, (int)nslices);
        if((double)ctotal <= 0.7 * (double)cells)
        {
            for(aoclsparse_int s = 0; s < nslices; s++) // the slice's mode (sell_kernels.hip) rides in the top byte
                cptr[s] |= (long long)(nlh[s] >> 8) << 56;
            sp.shared = true, sp.ccells = ctotal;
            st        = sp.cptr.upload(cptr.data(), sizeof(long long) * cptr.size(), rt.stream());
        }
        else
            sp.lead.release();
    }
    if(st == aoclsparse_status_success)
        st = sp.val.alloc(vsize * (size_t)std::max<long long>(cells, 1));
    if(st == aoclsparse_status_success)
        st = sp.col.alloc(sizeof(aoclsparse_int) * (size_t)std::max<long long>(sp.ccells, 1));
    if(st == aoclsparse_status_success)
        st = sp.rowlen.alloc(sizeof(aoclsparse_int) * (size_t)m);
    if(st != aoclsparse_status_success)
        return st;
    const long long     *cp = sp.shared ? sp.cptr.as<long long>() : nullptr;
    const unsigned short *ld = sp.shared ? sp.lead.as<unsigned short>() : nullptr;
    if(vsize == sizeof(cdouble))
        st = launch_sell_fill<cdouble>(rt.stream(), pack, m, d.base, d.ptr.as<aoclsparse_int>(), d.ind.as<aoclsparse_int>(),
                                       d.val.as<cdouble>(), nslices, sp.slice_ptr.as<long long>(), sp.val.as<cdouble>(),
                                       sp.col.as<aoclsparse_int>(), sp.rowlen.as<aoclsparse_int>(), cp, ld);
    else if(vsize == sizeof(float))
        st = launch_sell_fill<float>(rt.stream(), pack, m, d.base, d.ptr.as<aoclsparse_int>(), d.ind.as<aoclsparse_int>(),
                                     d.val.as<float>(), nslices, sp.slice_ptr.as<long long>(), sp.val.as<float>(),
                                     sp.col.as<aoclsparse_int>(), sp.rowlen.as<aoclsparse_int>(), cp, ld);
    else
        st = launch_sell_fill<double>(rt.stream(), pack, m, d.base, d.ptr.as<aoclsparse_int>(), d.ind.as<aoclsparse_int>(),
                                      d.val.as<double>(), nslices, sp.slice_ptr.as<long long>(), sp.val.as<double>(),
                                      sp.col.as<aoclsparse_int>(), sp.rowlen.as<aoclsparse_int>(), cp, ld);
    if(st != aoclsparse_status_success)
        return st;
    MI355_HIP_TRY(hipStreamSynchronize(rt.stream())); // sptr / cptr (host) are read by the uploads until here
    sp.nslices = nslices, sp.cells = cells, sp.pack = pack, sp.valid = sp.wanted = true;
    return aoclsparse_status_success;
}

constexpr int MERGE_AUTO_TILES = 16; // auto: merge-path once the longest row spans this many LDS tiles (32 until round 5)
// 0 auto, 1 CSR-Adaptive always, 2 merge-path whenever it can serve the request
// (read at plan-build time, i.e. once per handle and operator)
static int spmv_kernel_choice()
{
    return plan_option(aoclsparse_mi355_option_spmv_kernel);
}

aoclsparse_status build_merge_plan(aoclsparse_int m, aoclsparse_int nnz, aoclsparse_index_base base,
                                   const aoclsparse_int *ptr, size_t vsize, SpmvPlan &plan)
{
    MergePlan &mp = plan.merge;
    if(mp.valid || mp.tried)
        return aoclsparse_status_success;
    mp.tried = true;
    const int choice = spmv_kernel_choice();
    if(m <= 0 || nnz <= 0)
        return aoclsparse_status_success;
    // Automatic choice (aoclsparse_optimize / first product, from the row-length statistics of the row-block plan):
    // the row-block kernel gives a row longer than one LDS tile to ONE workgroup, which walks it tile by tile --
    // fine for rows of a few tiles (web-like: longest row 2,908 = 6 tiles; merge-path loses there, 29.8 vs 25.0 us,
    // and on circuit-like, 8.8 vs 7.3 us), a serial tail once a row spans tens of tiles.  Merge-path cuts such rows
    // into 1,024-item pieces spread over the chip, so it is selected when the longest row exceeds MERGE_AUTO_TILES tiles.
    // Round 5 (one launch, wavefront trees for the pieces; tools/exp_arrow.py, profiles/r5/merge_vs_adaptive.jsonl; adaptive /
    // merge, us): 1 row of 300 k 125 / 9.8; 4 x 250 k 114 / 12.4; 16 x 64 k 48 / 14.4; 64 x 16 k 24.9 / 16.9; 256 x 4 k
    // 15.6 / 18.8; 1,024 x 1 k 15.5 / 19.1 -- the crossover lies between rows of 8 and 32 tiles.
    constexpr int auto_tiles = MERGE_AUTO_TILES;
    const bool auto_pick = choice == 0 && auto_tiles > 0 && plan.long_rows > 0
                           && (long long)plan.max_row_nnz >= (long long)auto_tiles * (plan.tile & ~1);
    if(choice != 2 && !auto_pick)
        return aoclsparse_status_success;
    const long long             items  = (long long)m + nnz;
    const aoclsparse_int        ntiles = (aoclsparse_int)((items + MP_ITEMS - 1) / MP_ITEMS);
    std::vector<aoclsparse_int> st;
    try
    {
        st.resize(2 * ((size_t)ntiles + 1));
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    for(aoclsparse_int w = 0; w <= ntiles; w++)
    {
        // merge-path search on diagonal d: the largest i with (row end i-1) <= (d - i) consumed non-zeros
        const long long d  = std::min<long long>((long long)w * MP_ITEMS, items);
        long long       lo = std::max<long long>(0, d - nnz), hi = std::min<long long>(d, m);
        while(lo < hi)
        {
            const long long mid = (lo + hi) / 2;
            if((long long)(ptr[mid + 1] - base) <= d - mid - 1)
                lo = mid + 1;
            else
                hi = mid;
        }
        st[2 * (size_t)w]     = (aoclsparse_int)lo;
        st[2 * (size_t)w + 1] = (aoclsparse_int)(d - lo);
    }
    // first[w]: tile w holds the END of a row that started in an earlier tile -> the first tile with a (non-empty) head piece of
    // that row; -1 otherwise.  Tile v's head piece belongs to the row its successor starts in (st[2(v+1)]), and is non-empty
    // when the tile ends past that row's first entry.  Each tile is visited by at most one row's walk: O(ntiles).
    std::vector<aoclsparse_int> first;
    try
    {
        first.assign(2 * (size_t)ntiles, -1);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    for(aoclsparse_int w = 1; w < ntiles; w++)
    {
        const aoclsparse_int i0 = st[2 * (size_t)w], j0 = st[2 * (size_t)w + 1];
        if(st[2 * (size_t)w + 2] == i0 || i0 >= m) // no row ends in this tile
            continue;
        const aoclsparse_int rowstart = ptr[i0] - base;
        if(rowstart >= j0) // row i0 starts here
            continue;
        aoclsparse_int f = w - 1;
        while(f > 0 && st[2 * (size_t)f] == i0 && st[2 * (size_t)f + 1] > rowstart)
            f--;
        first[w] = f;
    }
    // endt[v] (stored behind first[]): the tile in which the row of tile v's head piece ends -- the inverse of first[]
    for(aoclsparse_int w = 1; w < ntiles; w++)
        for(aoclsparse_int v = first[w]; v >= 0 && v < w; v++)
            first[(size_t)ntiles + v] = w;
    // (the kernel decides "tile v has a head piece" from row_ptr: every such tile must know its end tile)
    for(aoclsparse_int v = 0; v < ntiles; v++)
    {
        const aoclsparse_int i1 = st[2 * (size_t)v + 2], j0 = st[2 * (size_t)v + 1], j1 = st[2 * (size_t)v + 3];
        const bool           has_head = i1 < m && std::max<aoclsparse_int>(ptr[i1] - base, j0) < j1;
        if(has_head != (first[(size_t)ntiles + v] >= 0))
            return aoclsparse_status_success; // (never seen) leave the matrix to the row-block kernel
    }
    (void)vsize;
    Runtime          &rt = Runtime::get();
    aoclsparse_status s1 = mp.starts.upload(st.data(), sizeof(aoclsparse_int) * st.size(), rt.stream());
    if(s1 == aoclsparse_status_success)
        s1 = mp.first.upload(first.data(), sizeof(aoclsparse_int) * first.size(), rt.stream());
    if(s1 != aoclsparse_status_success)
        return s1;
    MI355_HIP_TRY(hipStreamSynchronize(rt.stream())); // (uploads read the host vectors until the stream has run them)
    {
        std::lock_guard<std::mutex> g(mp.launch_lock);
        mp.sets.clear(); // piece sets are made by the first launch on each stream (spmv_api.cpp)
    }
    mp.ntiles = ntiles;
    mp.valid  = true;
    return aoclsparse_status_success;
}

// the value size build_spmv_plan sizes its row blocks by: a float handle plans 2,048-entry blocks for its products with vectors --
// unless it carries an mm hint: csrmm_tile_kernel walks the same blocks, and the 32-column float slab of the 1000^2 Laplacian
// measured 0.094-0.096 ms over 1,024-entry blocks against 0.101 over 2,048 (tools/history/exp_float_slab.py)
static size_t plan_value_size(const _aoclsparse_matrix &A)
{
    if(A.val_type != aoclsparse_smat)
        return 8;
    for(const Hint &h : A.hints)
        if(h.act == action_mm)
            return 8;
    return 4;
}

aoclsparse_status ensure_spmv(aoclsparse_matrix A, bool transposed, DeviceCsr *&dcsr, SpmvPlan *&plan)
{
    dcsr = transposed ? &A->dev_trans : &A->dev_user;
    plan = transposed ? &A->plan_trans : &A->plan_user;
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if(dcsr->valid && plan->valid)
            return aoclsparse_status_success;
    }
    if(transposed)
    {
        aoclsparse_status st = build_transpose(A);
        if(st != aoclsparse_status_success)
            return st;
    }
    std::unique_lock<std::shared_mutex> w(A->guard);
    const HostCsr                      &h = transposed ? *A->trans : A->user;
    if(!dcsr->valid)
    {
        aoclsparse_status st = upload_csr(h, val_size(A->val_type), *dcsr);
        if(st != aoclsparse_status_success)
            return st;
    }
    if(!plan->valid)
    {
        aoclsparse_status st = build_spmv_plan(h.m, h.ptr[h.m] - h.base, h.base, h.ptr, *plan, plan_value_size(*A));
        if(st != aoclsparse_status_success)
            return st;
    }
    if(!is_complex_type(A->val_type))
    {
        aoclsparse_status st = build_merge_plan(h.m, h.ptr[h.m] - h.base, h.base, h.ptr, val_size(A->val_type), *plan);
        if(st != aoclsparse_status_success)
            return st;
    }
    return aoclsparse_status_success;
}

} // namespace mi355

// =====================================================================================================
extern "C" {

aoclsparse_int mi355_csrmv_plan_bound(aoclsparse_int m, aoclsparse_int /*nnz*/)
{
    return 2 * (m + 2);
}

aoclsparse_int mi355_csrmv_plan_host(aoclsparse_int m, aoclsparse_int base, aoclsparse_int tile,
                                     const aoclsparse_int *row_ptr_host, aoclsparse_int *blocks_host)
{
    if(m < 0 || !row_ptr_host || !blocks_host || (base != 0 && base != 1) || (tile != 512 && tile != 1024 && tile != 2048))
        return -1;
    return plan_rows(m, (aoclsparse_index_base)base, tile, row_ptr_host, blocks_host, nullptr, nullptr);
}

// ---- descriptor: extra/aoclsparse_auxiliary.cpp:191-360 -------------------------------------------------
aoclsparse_status aoclsparse_create_mat_descr(aoclsparse_mat_descr *descr)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    *descr = new(std::nothrow) _aoclsparse_mat_descr;
    return *descr ? aoclsparse_status_success : aoclsparse_status_memory_error;
}

aoclsparse_status aoclsparse_copy_mat_descr(aoclsparse_mat_descr dest, const aoclsparse_mat_descr src)
{
    if(!dest || !src)
        return aoclsparse_status_invalid_pointer;
    *dest = *src;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_destroy_mat_descr(aoclsparse_mat_descr descr)
{
    delete descr; // NULL is a no-op success (:238-246)
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_set_mat_index_base(aoclsparse_mat_descr descr, aoclsparse_index_base base)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(base != aoclsparse_index_base_zero && base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    descr->base = base;
    return aoclsparse_status_success;
}

aoclsparse_index_base aoclsparse_get_mat_index_base(const aoclsparse_mat_descr descr)
{
    return descr ? descr->base : aoclsparse_index_base_zero;
}

aoclsparse_status aoclsparse_set_mat_type(aoclsparse_mat_descr descr, aoclsparse_matrix_type type)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(type != aoclsparse_matrix_type_general && type != aoclsparse_matrix_type_symmetric
       && type != aoclsparse_matrix_type_hermitian && type != aoclsparse_matrix_type_triangular)
        return aoclsparse_status_invalid_value;
    descr->type = type;
    return aoclsparse_status_success;
}

aoclsparse_matrix_type aoclsparse_get_mat_type(const aoclsparse_mat_descr descr)
{
    return descr ? descr->type : aoclsparse_matrix_type_general;
}

aoclsparse_status aoclsparse_set_mat_fill_mode(aoclsparse_mat_descr descr, aoclsparse_fill_mode fill_mode)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(fill_mode != aoclsparse_fill_mode_lower && fill_mode != aoclsparse_fill_mode_upper)
        return aoclsparse_status_invalid_value;
    descr->fill_mode = fill_mode;
    return aoclsparse_status_success;
}

aoclsparse_fill_mode aoclsparse_get_mat_fill_mode(const aoclsparse_mat_descr descr)
{
    return descr ? descr->fill_mode : aoclsparse_fill_mode_lower;
}

aoclsparse_status aoclsparse_set_mat_diag_type(aoclsparse_mat_descr descr, aoclsparse_diag_type diag_type)
{
    if(!descr)
        return aoclsparse_status_invalid_pointer;
    if(diag_type != aoclsparse_diag_type_unit && diag_type != aoclsparse_diag_type_non_unit
       && diag_type != aoclsparse_diag_type_zero)
        return aoclsparse_status_invalid_value;
    descr->diag_type = diag_type;
    return aoclsparse_status_success;
}

aoclsparse_diag_type aoclsparse_get_mat_diag_type(const aoclsparse_mat_descr descr)
{
    return descr ? descr->diag_type : aoclsparse_diag_type_non_unit;
}

// ---- create / destroy / export: create/aoclsparse_create.cpp:34-97, auxiliary.cpp:657-671, 1303-1353 ---
static aoclsparse_status create_csr(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                    aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                    aoclsparse_int *row_ptr, aoclsparse_int *col_idx, void *val,
                                    aoclsparse_matrix_data_type vt)
{
    if(!mat)
        return aoclsparse_status_invalid_pointer;
    *mat = nullptr;
    int               sort     = 0;
    bool              fulldiag = false;
    aoclsparse_status st = mat_check(M, N, nnz, row_ptr, col_idx, val, 0, base, sort, fulldiag);
    if(st != aoclsparse_status_success)
        return st;
    _aoclsparse_matrix *A = new(std::nothrow) _aoclsparse_matrix;
    if(!A)
        return aoclsparse_status_memory_error;
    A->m = M, A->n = N, A->nnz = nnz, A->base = base, A->val_type = vt;
    A->sort = sort, A->fulldiag = fulldiag;
    A->user.m = M, A->user.n = N, A->user.nnz = nnz, A->user.base = base;
    A->user.ptr = row_ptr, A->user.ind = col_idx, A->user.val = val;
    A->user.owned = false;
    *mat          = A;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_create_dcsr(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                         aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                         aoclsparse_int *row_ptr, aoclsparse_int *col_idx, double *val)
{
    return create_csr(mat, base, M, N, nnz, row_ptr, col_idx, val, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_create_scsr(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                         aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                         aoclsparse_int *row_ptr, aoclsparse_int *col_idx, float *val)
{
    return create_csr(mat, base, M, N, nnz, row_ptr, col_idx, val, aoclsparse_smat);
}

aoclsparse_status aoclsparse_create_ccsr(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                         aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                         aoclsparse_int *row_ptr, aoclsparse_int *col_idx,
                                         aoclsparse_float_complex *val)
{
    return create_csr(mat, base, M, N, nnz, row_ptr, col_idx, val, aoclsparse_cmat);
}

aoclsparse_status aoclsparse_create_zcsr(aoclsparse_matrix *mat, aoclsparse_index_base base,
                                         aoclsparse_int M, aoclsparse_int N, aoclsparse_int nnz,
                                         aoclsparse_int *row_ptr, aoclsparse_int *col_idx,
                                         aoclsparse_double_complex *val)
{
    return create_csr(mat, base, M, N, nnz, row_ptr, col_idx, val, aoclsparse_zmat);
}

aoclsparse_status aoclsparse_destroy(aoclsparse_matrix *mat)
{
    if(!mat)
        return aoclsparse_status_success; // :657-671 accepts NULL
    if(*mat)
    {
        if((*mat)->owns_user_arrays)
            (*mat)->user.owned = true; // sp2m results: the handle owns its CSR
        if((*mat)->ilu_factor)
            aoclsparse_destroy(&(*mat)->ilu_factor); // aliases ptr/ind/ilu_val: frees only its own plans
        for(auto &r : (*mat)->replicas) // multi-device replicas alias the same host arrays: only their device side goes
            if(r)
                aoclsparse_destroy(&r);
        std::free((*mat)->ilu_val);
        if((*mat)->trsv_timeout_host)
            (void)hipHostFree(const_cast<unsigned int *>((*mat)->trsv_timeout_host));
        delete *mat;
        *mat = nullptr;
    }
    return aoclsparse_status_success;
}

} // extern "C"

template <typename T>
static aoclsparse_status export_csr(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                    aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                    aoclsparse_int **row_ptr, aoclsparse_int **col_ind, T **val,
                                    aoclsparse_matrix_data_type vt)
{
    if(!mat || !base || !m || !n || !nnz || !row_ptr || !col_ind || !val)
        return aoclsparse_status_invalid_pointer;
    if(mat->val_type != vt)
        return aoclsparse_status_wrong_type;
    std::shared_lock<std::shared_mutex> r(mat->guard);
    const HostCsr *c = mat->opt ? mat->opt : &mat->user; // optimized CSR preferred (:1326-1347)
    if(!c->ptr || !c->ind || !c->val)
        return aoclsparse_status_invalid_value;
    *row_ptr = c->ptr;
    *col_ind = c->ind;
    *val     = static_cast<T *>(c->val);
    *nnz     = c->ptr[mat->m] - c->base;
    *base    = c->base;
    *m       = mat->m;
    *n       = mat->n;
    return aoclsparse_status_success;
}

extern "C" {

aoclsparse_status aoclsparse_export_dcsr(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                         aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                         aoclsparse_int **row_ptr, aoclsparse_int **col_ind, double **val)
{
    return export_csr<double>(mat, base, m, n, nnz, row_ptr, col_ind, val, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_export_scsr(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                         aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                         aoclsparse_int **row_ptr, aoclsparse_int **col_ind, float **val)
{
    return export_csr<float>(mat, base, m, n, nnz, row_ptr, col_ind, val, aoclsparse_smat);
}

aoclsparse_status aoclsparse_export_ccsr(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                         aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                         aoclsparse_int **row_ptr, aoclsparse_int **col_ind,
                                         aoclsparse_float_complex **val)
{
    return export_csr<aoclsparse_float_complex>(mat, base, m, n, nnz, row_ptr, col_ind, val, aoclsparse_cmat);
}

aoclsparse_status aoclsparse_export_zcsr(const aoclsparse_matrix mat, aoclsparse_index_base *base,
                                         aoclsparse_int *m, aoclsparse_int *n, aoclsparse_int *nnz,
                                         aoclsparse_int **row_ptr, aoclsparse_int **col_ind,
                                         aoclsparse_double_complex **val)
{
    return export_csr<aoclsparse_double_complex>(mat, base, m, n, nnz, row_ptr, col_ind, val, aoclsparse_zmat);
}

aoclsparse_status aoclsparse_mi355_export_diag(const aoclsparse_matrix A, aoclsparse_int **idiag,
                                               aoclsparse_int **iurow, aoclsparse_int *is_internal)
{
    if(!A || !idiag || !iurow)
        return aoclsparse_status_invalid_pointer;
    std::shared_lock<std::shared_mutex> r(A->guard);
    if(!A->opt)
        return aoclsparse_status_invalid_operation;
    *idiag = A->opt->idiag;
    *iurow = A->opt->iurow;
    if(is_internal)
        *is_internal = A->opt != &A->user;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_invalidate(aoclsparse_matrix A)
{
    if(!A)
        return aoclsparse_status_invalid_pointer;
    std::unique_lock<std::shared_mutex> w(A->guard);
    for(auto &r : A->replicas) // the replicas on other devices mirror the same arrays: rebuilt on the next multi-device call
        if(r)
            aoclsparse_destroy(&r);
    A->replicas_cloned = 0;
    A->dev_user.valid = A->dev_trans.valid = false;
    A->plan_user.valid = A->plan_trans.valid = false;
    A->plan_user.sell.valid = A->plan_user.sell.tried = false;
    A->plan_trans.sell.valid = A->plan_trans.sell.tried = false;
    A->plan_user.merge.valid = A->plan_user.merge.tried = false;
    A->plan_trans.merge.valid = A->plan_trans.merge.tried = false;
    A->plan_user.mm.valid = A->plan_user.mm.tried = false;
    A->plan_trans.mm.valid = A->plan_trans.mm.tried = false;
    A->plan_user.mm.pairs = A->plan_user.mm.pairs_tried = false;
    A->plan_trans.mm.pairs = A->plan_trans.mm.pairs_tried = false;
    A->plan_user.bell.valid = A->plan_user.bell.tried = false;
    A->plan_trans.bell.valid = A->plan_trans.bell.tried = false;
    A->plan_user.mm.win = A->plan_user.mm.win_tried = false;
    A->plan_trans.mm.win = A->plan_trans.mm.win_tried = false;
    A->plan_user.mm.row_runs = A->plan_user.mm.runs_tried = false;
    A->plan_trans.mm.row_runs = A->plan_trans.mm.runs_tried = false;
    for(auto &p : A->trsv_plan)
        p.valid = p.rows_valid = false, p.nlevels = -1, p.blk.valid = p.blk.tried = false, p.blk.chunk.valid = p.blk.chunk.tried = false;
    A->trans.reset();
    return aoclsparse_status_success;
}

} // extern "C"

// ---- value mutation + copy: extra/aoclsparse_auxiliary.hpp:216-270, 388-470; auxiliary.cpp:775-835 -------
// The reference writes into the FIRST representation (the caller's arrays, which the handle aliases) and
// deletes every derived copy.  Same here, device mirrors included: they are rebuilt lazily.
namespace mi355
{
void drop_derived_state(aoclsparse_matrix A)
{
    if(A->opt != &A->user)
    {
        A->opt_copy.reset();
        A->opt       = nullptr;
        A->optimized = false;
    }
    A->trans.reset();
    A->derived.clear();
    for(auto &r : A->replicas) // multi-device replicas hold device copies of the old values
        if(r)
            aoclsparse_destroy(&r);
    A->replicas_cloned = 0;
    A->dev_user.valid = A->dev_trans.valid = false; // row-block plans stay valid: structure is unchanged
    A->plan_user.sell.valid = A->plan_user.sell.tried = false; // the SELL copies hold values: rebuilt on optimize
    A->plan_trans.sell.valid = A->plan_trans.sell.tried = false;
    // ... and so does the blocked-ELL copy of csrmm (round 6: it was left standing, and a product after aoclsparse_?set_value /
    // ?update_values on a block-dense handle used the OLD values -- found by tests/test_gpu_r6.py)
    A->plan_user.bell.valid = A->plan_user.bell.tried = false;
    A->plan_trans.bell.valid = A->plan_trans.bell.tried = false;
    A->dev_diag.release();
    for(auto &p : A->trsv_plan)
        p.valid = p.rows_valid = false, p.nlevels = -1, p.blk.valid = p.blk.tried = false, p.blk.chunk.valid = p.blk.chunk.tried = false;
}
} // namespace mi355

template <typename T>
static aoclsparse_status set_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx, T val,
                                   aoclsparse_matrix_data_type vt)
{
    const bool coo = A && A->input_format == aoclsparse_coo_mat;
    if(!A || (coo ? (!A->coo_row || !A->coo_col || !A->coo_val) : (!A->user.ptr || !A->user.ind || !A->user.val)))
        return aoclsparse_status_invalid_pointer;
    const aoclsparse_int b = A->base;
    if(A->m + b <= row_idx || row_idx < b || A->n + b <= col_idx || col_idx < b)
        return aoclsparse_status_invalid_value;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    std::unique_lock<std::shared_mutex> w(A->guard);
    if(coo)
        return coo_set_value(A, row_idx, col_idx, &val);
    const aoclsparse_int r = row_idx - b;
    for(aoclsparse_int p = A->user.ptr[r] - b; p < A->user.ptr[r + 1] - b; p++)
        if(A->user.ind[p] == col_idx)
        {
            static_cast<T *>(A->user.val)[p] = val;
            if(A->csc_ptr)
                csc_set_value(A, row_idx, col_idx, &val);
            drop_derived_state(A);
            return aoclsparse_status_success;
        }
    return aoclsparse_status_invalid_index_value;
}

template <typename T>
static aoclsparse_status update_values(aoclsparse_matrix A, aoclsparse_int len, T *val, aoclsparse_matrix_data_type vt)
{
    const bool coo = A && A->input_format == aoclsparse_coo_mat;
    if(!A || !val || (coo ? !A->coo_val : !A->user.ptr))
        return aoclsparse_status_invalid_pointer;
    if(len != A->nnz)
        return aoclsparse_status_invalid_size;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(!coo && !A->user.val)
        return aoclsparse_status_invalid_pointer;
    std::unique_lock<std::shared_mutex> w(A->guard);
    if(coo) // the values follow the order of the caller's arrays (the first representation)
    {
        std::memcpy(A->coo_val, val, sizeof(T) * (size_t)len);
        return aoclsparse_status_success;
    }
    if(A->csc_ptr)
    {
        std::memcpy(A->csc_val, val, sizeof(T) * (size_t)len);
        aoclsparse_status st = csc_refresh_csr(A);
        if(st != aoclsparse_status_success)
            return st;
    }
    else
        std::memcpy(A->user.val, val, sizeof(T) * (size_t)len);
    drop_derived_state(A);
    return aoclsparse_status_success;
}

extern "C" {

aoclsparse_status aoclsparse_dset_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx, double val)
{
    return set_value<double>(A, row_idx, col_idx, val, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_sset_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx, float val)
{
    return set_value<float>(A, row_idx, col_idx, val, aoclsparse_smat);
}
aoclsparse_status aoclsparse_cset_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx,
                                        aoclsparse_float_complex val)
{
    return set_value<aoclsparse_float_complex>(A, row_idx, col_idx, val, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zset_value(aoclsparse_matrix A, aoclsparse_int row_idx, aoclsparse_int col_idx,
                                        aoclsparse_double_complex val)
{
    return set_value<aoclsparse_double_complex>(A, row_idx, col_idx, val, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_cupdate_values(aoclsparse_matrix A, aoclsparse_int len, aoclsparse_float_complex *val)
{
    return update_values<aoclsparse_float_complex>(A, len, val, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_zupdate_values(aoclsparse_matrix A, aoclsparse_int len, aoclsparse_double_complex *val)
{
    return update_values<aoclsparse_double_complex>(A, len, val, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_dupdate_values(aoclsparse_matrix A, aoclsparse_int len, double *val)
{
    return update_values<double>(A, len, val, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_supdate_values(aoclsparse_matrix A, aoclsparse_int len, float *val)
{
    return update_values<float>(A, len, val, aoclsparse_smat);
}

aoclsparse_status aoclsparse_copy(const aoclsparse_matrix src, const aoclsparse_mat_descr /*descr*/,
                                  aoclsparse_matrix *dest)
{
    if(!src || !dest)
        return aoclsparse_status_invalid_pointer;
    if(src->m < 0 || src->n < 0 || src->nnz < 0)
        return aoclsparse_status_invalid_size;
    if(src == *dest)
        return aoclsparse_status_invalid_pointer;
    if(src->val_type < aoclsparse_dmat || src->val_type > aoclsparse_zmat)
        return aoclsparse_status_wrong_type;
    if(!src->user.ptr || !src->user.ind || !src->user.val)
        return aoclsparse_status_invalid_pointer;
    _aoclsparse_matrix *c = new(std::nothrow) _aoclsparse_matrix;
    if(!c)
        return aoclsparse_status_memory_error;
    const size_t         vs  = val_size(src->val_type);
    const aoclsparse_int nnz = src->nnz;
    c->m = src->m, c->n = src->n, c->nnz = nnz, c->base = src->base, c->val_type = src->val_type;
    c->sort = src->sort, c->fulldiag = src->fulldiag;
    c->user.m = src->m, c->user.n = src->n, c->user.nnz = nnz, c->user.base = src->base;
    c->user.ptr = new(std::nothrow) aoclsparse_int[(size_t)src->m + 1];
    c->user.ind = new(std::nothrow) aoclsparse_int[(size_t)std::max(nnz, 1)];
    c->user.val = ::operator new(vs * (size_t)std::max(nnz, 1), std::nothrow);
    c->user.owned = true, c->owns_user_arrays = true; // deep copy: the new handle owns its arrays
    if(!c->user.ptr || !c->user.ind || !c->user.val)
    {
        delete c;
        return aoclsparse_status_memory_error;
    }
    std::memcpy(c->user.ptr, src->user.ptr, sizeof(aoclsparse_int) * ((size_t)src->m + 1));
    std::memcpy(c->user.ind, src->user.ind, sizeof(aoclsparse_int) * (size_t)nnz);
    std::memcpy(c->user.val, src->user.val, vs * (size_t)nnz);
    *dest = c;
    return aoclsparse_status_success;
}

} // extern "C"

extern "C" {

// ---- hints: analysis/aoclsparse_analysis.cpp:568-747 -----------------------------------------------------
static aoclsparse_status set_hint(aoclsparse_matrix mat, hinted_action act, aoclsparse_operation trans,
                                  const aoclsparse_mat_descr descr, aoclsparse_int ncalls,
                                  aoclsparse_int kid = -1)
{
    if(!mat || !mat->user.ptr || !descr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != aoclsparse_index_base_zero && descr->base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    if(descr->base != mat->base) // is_descr_matching, mat_structures.hpp:808-814
        return aoclsparse_status_invalid_value;
    if(trans != aoclsparse_operation_none && trans != aoclsparse_operation_transpose
       && trans != aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_invalid_value;
    if(descr->fill_mode != aoclsparse_fill_mode_lower && descr->fill_mode != aoclsparse_fill_mode_upper)
        return aoclsparse_status_invalid_value;
    if(descr->diag_type != aoclsparse_diag_type_non_unit && descr->diag_type != aoclsparse_diag_type_unit
       && descr->diag_type != aoclsparse_diag_type_zero)
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_symmetric
       && descr->type != aoclsparse_matrix_type_triangular && descr->type != aoclsparse_matrix_type_hermitian)
        return aoclsparse_status_invalid_value;
    if(ncalls < 0 || (ncalls == 0 && kid == -1))
        return aoclsparse_status_invalid_value;
    if(act <= action_none || act >= action_max)
        return aoclsparse_status_invalid_operation;
    try
    {
        Hint h{act, get_doid(descr, trans), trans, descr->type, descr->fill_mode, ncalls, kid, false};
        mat->hints.insert(mat->hints.begin(), h); // newest first
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_set_mv_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                         const aoclsparse_mat_descr descr, aoclsparse_int n)
{
    return set_hint(mat, action_mv, trans, descr, n);
}

aoclsparse_status aoclsparse_set_mv_hint_kid(aoclsparse_matrix mat, aoclsparse_operation trans,
                                             const aoclsparse_mat_descr descr, aoclsparse_int n,
                                             aoclsparse_int kid)
{
    return set_hint(mat, action_mv, trans, descr, n, kid);
}

aoclsparse_status aoclsparse_set_sv_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                         const aoclsparse_mat_descr descr, aoclsparse_int n)
{
    return set_hint(mat, action_sv, trans, descr, n);
}

aoclsparse_status aoclsparse_set_mm_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                         const aoclsparse_mat_descr descr, aoclsparse_int n)
{
    return set_hint(mat, action_mm, trans, descr, n);
}

aoclsparse_status aoclsparse_set_2m_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                         const aoclsparse_mat_descr descr, aoclsparse_int n)
{
    return set_hint(mat, action_2m, trans, descr, n);
}

aoclsparse_status aoclsparse_set_dotmv_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                            const aoclsparse_mat_descr descr, aoclsparse_int n)
{
    return set_hint(mat, action_dotmv, trans, descr, n);
}

aoclsparse_status aoclsparse_set_lu_smoother_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                                  const aoclsparse_mat_descr descr, aoclsparse_int n)
{
    return set_hint(mat, action_ilu0, trans, descr, n);
}

aoclsparse_status aoclsparse_set_sm_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                         const aoclsparse_mat_descr descr, const aoclsparse_order order,
                                         aoclsparse_int n)
{
    if(order != aoclsparse_order_row && order != aoclsparse_order_column) // analysis.cpp:686-689
        return aoclsparse_status_invalid_value;
    return set_hint(mat, order == aoclsparse_order_row ? action_sm_row : action_sm_col, trans, descr, n);
}

aoclsparse_status aoclsparse_set_symgs_hint(aoclsparse_matrix mat, aoclsparse_operation trans,
                                            const aoclsparse_mat_descr descr, aoclsparse_int n)
{
    return set_hint(mat, action_symgs, trans, descr, n);
}

aoclsparse_status aoclsparse_set_sorv_hint(aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
                                           const aoclsparse_sor_type type, const aoclsparse_int n)
{
    if(type != aoclsparse_sor_forward && type != aoclsparse_sor_backward && type != aoclsparse_sor_symmetric)
        return aoclsparse_status_invalid_value; // analysis.cpp:713-717
    const hinted_action act = type == aoclsparse_sor_forward    ? action_sorv_forward
                              : type == aoclsparse_sor_backward ? action_sorv_backward
                                                                : action_sorv_symm;
    return set_hint(mat, act, aoclsparse_operation_none, descr, n);
}

aoclsparse_status aoclsparse_set_memory_hint(aoclsparse_matrix mat, const aoclsparse_memory_usage policy)
{
    // analysis.cpp:725-747
    if(!mat)
        return aoclsparse_status_invalid_pointer;
    if(policy != aoclsparse_memory_usage_minimal && policy != aoclsparse_memory_usage_unrestricted)
        return aoclsparse_status_invalid_value;
    mat->mem_policy = policy;
    return aoclsparse_status_success;
}

} // extern "C"
