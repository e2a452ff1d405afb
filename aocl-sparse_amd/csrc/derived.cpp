// derived.cpp -- general CSR matrices derived from the clean CSR for non-general descriptors.
//
// The reference serves symmetric / triangular descriptors with dedicated serial CPU kernels that walk
// one triangle of the clean CSR through idiag / iurow (level2/aoclsparse_csrmv_kr.hpp:93-444, 658-728;
// level2/aoclsparse_csrmv_kt.cpp:217-329; level3/aoclsparse_csrmm.hpp:148-358).  On the GPU the same
// operators are materialised once per (type, fill, diag, op) as an ordinary CSR in HBM --
//   symmetric : strict triangle + D' + its transpose     (D' = D, I or nothing for non_unit/unit/zero)
//   triangular: strict triangle + D', or its transpose
// -- and then run through the general SpMV / csrmm kernels.  288 GB of HBM make the extra copy a
// non-issue; it is built at aoclsparse_optimize for a matching hint or on first use, exactly where the
// reference calls aoclsparse_csr_csc_optimize on the fly (mv.cpp:134-149).
#include "internal.hpp"

#include <algorithm>

namespace mi355
{

template <typename T>
static void build_derived(const HostCsr &c, aoclsparse_matrix_type type, bool upper, aoclsparse_diag_type diag,
                          bool transposed, HostCsr &out)
{
    const aoclsparse_int  m = c.m, b = c.base;
    const aoclsparse_int *s = upper ? c.iurow : c.ptr; // strict triangle of row i: [s[i], e[i]) (base b)
    const aoclsparse_int *e = upper ? c.ptr + 1 : c.idiag;
    const T              *v = static_cast<const T *>(c.val);
    const bool            sym = type == aoclsparse_matrix_type_symmetric || type == aoclsparse_matrix_type_hermitian;
    const aoclsparse_int  dim = std::min(c.m, c.n);
    auto has_diag = [&](aoclsparse_int i) {
        if(i >= dim || diag == aoclsparse_diag_type_zero)
            return false;
        return diag == aoclsparse_diag_type_unit || c.iurow[i] == c.idiag[i] + 1;
    };
    auto diag_val = [&](aoclsparse_int i) { return diag == aoclsparse_diag_type_unit ? T(1) : v[c.idiag[i] - b]; };
    // which pieces does row i of the result get?
    //   direct  = strict entries of row i           (not for a transposed triangular operator)
    //   mirror  = strict entries (r, i) of other rows r, as (i, r)   (symmetric, or transposed triangular)
    const bool direct = sym || !transposed;
    const bool mirror = sym || transposed;
    const aoclsparse_int rows = (sym || !transposed) ? c.m : c.n, cols = (sym || !transposed) ? c.n : c.m;
    std::vector<aoclsparse_int> cnt((size_t)rows + 1, 0);
    for(aoclsparse_int i = 0; i < m; i++)
    {
        if(direct)
            cnt[i + 1] += e[i] - s[i];
        if(mirror)
            for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
                if(c.ind[p] - b < rows)
                    cnt[c.ind[p] - b + 1]++;
    }
    for(aoclsparse_int i = 0; i < std::min(rows, dim); i++)
        if(has_diag(i))
            cnt[i + 1]++;
    for(aoclsparse_int i = 0; i < rows; i++)
        cnt[i + 1] += cnt[i];
    const aoclsparse_int nnz = cnt[rows];
    out.m = rows, out.n = cols, out.nnz = nnz, out.base = aoclsparse_index_base_zero;
    out.owned = true;
    out.ptr   = new aoclsparse_int[(size_t)rows + 1];
    out.ind   = new aoclsparse_int[(size_t)std::max(nnz, 1)];
    out.val   = ::operator new(sizeof(T) * (size_t)std::max(nnz, 1));
    T *ov     = static_cast<T *>(out.val);
    std::copy(cnt.begin(), cnt.end(), out.ptr);
    std::vector<aoclsparse_int> next(cnt.begin(), cnt.end() - 1);
    // Fill so that every row ends up sorted by column.  Columns of row i come from three groups:
    //   lower fill: [direct strict (cols < i)] [diag] [mirror (cols > i)]
    //   upper fill: [mirror (cols < i)] [diag] [direct strict (cols > i)]
    // Mirror entries of row i arrive in ascending source row when source rows are visited ascending.
    // the mirrored half of a hermitian matrix is the conjugate of the stored one (identity for real types)
    const bool herm = type == aoclsparse_matrix_type_hermitian;
    auto       mir  = [&](T val) { return herm ? conj_of(val) : val; };
    auto put = [&](aoclsparse_int r, aoclsparse_int col, T val) {
        const aoclsparse_int q = next[r]++;
        out.ind[q]             = col;
        ov[q]                  = val;
    };
    if(!upper)
    {
        // pass 1: direct strict parts and diagonals, row by row; pass 2: mirrors (cols > row)
        for(aoclsparse_int i = 0; i < rows; i++)
        {
            if(direct && i < m)
                for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
                    put(i, c.ind[p] - b, v[p]);
            if(i < dim && has_diag(i))
                put(i, i, diag_val(i));
        }
        if(mirror)
            for(aoclsparse_int i = 0; i < m; i++)
                for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
                    if(c.ind[p] - b < rows)
                        put(c.ind[p] - b, i, mir(v[p]));
    }
    else
    {
        // pass 1: mirrors (cols < row); pass 2: diagonals then direct strict parts
        if(mirror)
            for(aoclsparse_int i = 0; i < m; i++)
                for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
                    if(c.ind[p] - b < rows)
                        put(c.ind[p] - b, i, mir(v[p]));
        for(aoclsparse_int i = 0; i < rows; i++)
        {
            if(i < dim && has_diag(i))
                put(i, i, diag_val(i));
            if(direct && i < m)
                for(aoclsparse_int p = s[i] - b; p < e[i] - b; p++)
                    put(i, c.ind[p] - b, v[p]);
        }
    }
}

aoclsparse_status ensure_derived(aoclsparse_matrix A, aoclsparse_matrix_type type, aoclsparse_fill_mode fill,
                                 aoclsparse_diag_type diag, bool transposed, Derived *&out)
{
    aoclsparse_status st = csr_optimize(A);
    if(st != aoclsparse_status_success)
        return st;
    const bool sym = type != aoclsparse_matrix_type_triangular;
    if(sym)
        transposed = false; // (A^T = A)
    auto find = [&]() -> Derived * {
        for(auto &d : A->derived)
            if(d->type == (int)type && d->fill == (int)fill && d->diag == (int)diag && d->trans == (int)transposed)
                return d.get();
        return nullptr;
    };
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if((out = find()))
            return aoclsparse_status_success;
    }
    std::unique_lock<std::shared_mutex> w(A->guard);
    if((out = find()))
        return aoclsparse_status_success;
    try
    {
        std::unique_ptr<Derived> d(new Derived);
        d->type = type, d->fill = fill, d->diag = diag, d->trans = transposed;
        dispatch_value_type(A->val_type, [&](auto tag) {
            build_derived<decltype(tag)>(*A->opt, type, fill == aoclsparse_fill_mode_upper, diag, transposed, d->host);
            return 0;
        });
        st = upload_csr(d->host, val_size(A->val_type), d->dev);
        if(st == aoclsparse_status_success)
            st = build_spmv_plan(d->host.m, d->host.nnz, d->host.base, d->host.ptr, d->plan, val_size(A->val_type));
        // a derived operator exists because products with it were asked for: give it the SELL-64 twin too
        // (when its padding is small); both kernels realise the same summation order on the derived rows
        if(st == aoclsparse_status_success && A->mem_policy == aoclsparse_memory_usage_unrestricted
           && !is_complex_type(A->val_type))
            st = build_sell(d->host.ptr, d->dev, val_size(A->val_type), d->plan);
        if(st != aoclsparse_status_success)
            return st;
        out = d.get();
        A->derived.push_back(std::move(d));
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

} // namespace mi355
