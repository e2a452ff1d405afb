// sp2m_api.cpp -- aoclsparse_sp2m / aoclsparse_spmm / aoclsparse_?csr2m: C = op(A) * op(B), sparse result.
//
// Driver logic follows level3/aoclsparse_csr2m.cpp:592-861 of the reference: argument checks in the
// same order, empty-product quick return that still allocates an empty C, op handling (A^T / B^T by an
// explicit counting-sort transpose, A^T B^T as (B A)^T with the product transposed back at finalize),
// the two-stage request protocol (stage_nnz_count allocates C and fills row_ptr; stage_finalize fills
// col_ind / val of that same C; full_computation does both), C always 0-based.  The product itself runs
// on the GPU (spgemm_kernels.hip); C's arrays are host arrays owned by the new handle, like any handle
// created by aoclsparse_create_?csr, so export / destroy work unchanged.
#include "internal.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

using namespace mi355;

namespace mi355
{
template <typename T>
aoclsparse_status launch_spgemm_heavy(hipStream_t s, bool fill, aoclsparse_int nrows, const SpgHeavy *heavy, int *g_key, int *g_pos,
                                      int *g_list, T *g_acc, int base_a, const aoclsparse_int *ptr_a, const aoclsparse_int *ind_a,
                                      const T *val_a, int base_b, const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b,
                                      const T *val_b, const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c,
                                      bool conj_a, bool conj_b, unsigned int *bad);
template <typename T>
aoclsparse_status launch_spgemm_bin(hipStream_t s, bool fill, int bin, aoclsparse_int nrows, const aoclsparse_int *rows, int base_a,
                                    const aoclsparse_int *ptr_a, const aoclsparse_int *ind_a, const T *val_a, int base_b,
                                    const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b, const T *val_b,
                                    const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a, bool conj_b,
                                    unsigned int *bad);

aoclsparse_status new_csr_result(aoclsparse_matrix *C, aoclsparse_int m, aoclsparse_int n, aoclsparse_int nnz,
                                 aoclsparse_matrix_data_type vt, const aoclsparse_int *row_ptr, aoclsparse_index_base base)
{
    _aoclsparse_matrix *c = new(std::nothrow) _aoclsparse_matrix;
    if(!c)
        return aoclsparse_status_memory_error;
    const size_t vs = val_size(vt);
    c->m = m, c->n = n, c->nnz = nnz, c->base = base, c->val_type = vt;
    c->user.m = m, c->user.n = n, c->user.nnz = nnz, c->user.base = base;
    c->user.ptr = new(std::nothrow) aoclsparse_int[(size_t)m + 1];
    c->user.ind = static_cast<aoclsparse_int *>(host_result_alloc(sizeof(aoclsparse_int) * (size_t)std::max(nnz, 1)));
    c->user.val = host_result_alloc(vs * (size_t)std::max(nnz, 1));
    c->user.owned = c->user.result_arrays = true; // freed by the handle
    c->owns_user_arrays = true;
    if(!c->user.ptr || !c->user.ind || !c->user.val)
    {
        delete c;
        return aoclsparse_status_memory_error;
    }
    if(row_ptr)
        std::memcpy(c->user.ptr, row_ptr, sizeof(aoclsparse_int) * ((size_t)m + 1));
    else
        std::fill(c->user.ptr, c->user.ptr + m + 1, (aoclsparse_int)base);
    *C = c;
    return aoclsparse_status_success;
}
}

namespace
{

// host CSR operand, either a view of a handle's user arrays or an owned transpose
template <typename T>
struct Operand
{
    aoclsparse_int              m = 0, n = 0, nnz = 0, base = 0;
    const aoclsparse_int       *ptr = nullptr, *ind = nullptr;
    const T                    *val = nullptr;
    std::vector<aoclsparse_int> optr, oind;
    std::vector<T>              oval;
};

template <typename T>
void view_of(const aoclsparse_matrix A, Operand<T> &o)
{
    o.m = A->m, o.n = A->n, o.nnz = A->nnz, o.base = A->base;
    o.ptr = A->user.ptr, o.ind = A->user.ind, o.val = static_cast<const T *>(A->user.val);
}

// conversion/aoclsparse_convert.hpp:552-655 with equal in/out bases (csr2m.cpp:771-781)
template <typename T>
void transpose_of(const Operand<T> &a, Operand<T> &t)
{
    const aoclsparse_int b = a.base;
    t.m = a.n, t.n = a.m, t.nnz = a.nnz, t.base = b;
    t.optr.assign((size_t)t.m + 1, 0);
    t.oind.resize((size_t)std::max(a.nnz, 1));
    t.oval.resize((size_t)std::max(a.nnz, 1));
    for(aoclsparse_int p = 0; p < a.nnz; p++)
        t.optr[a.ind[p] - b + 1]++;
    for(aoclsparse_int j = 0; j < t.m; j++)
        t.optr[j + 1] += t.optr[j];
    std::vector<aoclsparse_int> next(t.optr.begin(), t.optr.end() - 1);
    for(aoclsparse_int i = 0; i < a.m; i++)
        for(aoclsparse_int p = a.ptr[i] - b; p < a.ptr[i + 1] - b; p++)
        {
            const aoclsparse_int q = next[a.ind[p] - b]++;
            t.oind[q]              = i + b;
            t.oval[q]              = a.val[p];
        }
    for(aoclsparse_int j = 0; j <= t.m; j++)
        t.optr[j] += b;
    t.ptr = t.optr.data(), t.ind = t.oind.data(), t.val = t.oval.data();
}

// the rows of a product grouped by the bin of spgemm_hash_kernel that serves them (device arrays: staging slots of the runtime,
// grown on demand and reused by the next product -- a hipMalloc / hipFree pair per temporary cost more than the kernels)
struct Binned
{
    aoclsparse_int        bounds[SPGEMM_BINS + 1] = {};
    const aoclsparse_int *d_order = nullptr; // null: one bin holds every row (no list needed)
    // the rows of the last bin (tables in a global slab), in batches whose tables fit SPG_SLAB_SLOTS: batch b = records
    // [batch[b], batch[b + 1]) with offsets that start at 0
    std::vector<SpgHeavy>       heavy;
    std::vector<aoclsparse_int> batch;
    long long                   slots = 0, entries = 0; // of the largest batch
    const SpgHeavy             *d_heavy = nullptr;
};
constexpr long long SPG_SLAB_SLOTS = 1LL << 28; // 1 GiB of keys (and of slots in the fill pass) per batch
enum
{
    SLOT_XP = 16,
    SLOT_XI,
    SLOT_XV,
    SLOT_YP,
    SLOT_YI,
    SLOT_YV,
    SLOT_CNT,
    SLOT_ORDER_COUNT,
    SLOT_ORDER_FILL,
    SLOT_HEAVY_COUNT,
    SLOT_HEAVY_FILL,
    SLOT_G_KEY,
    SLOT_G_POS,
    SLOT_G_LIST,
    SLOT_G_ACC,
    SLOT_CAP,
    SLOT_SMALL,
    SLOT_SCAN,
    SLOT_HEAVY_KEYS,
    SLOT_END
};
static_assert(SLOT_END <= 40, "Runtime::stage_ is too small");
constexpr int SLOT_KEY = SLOT_CNT; // the list size of every row: the counts themselves

template <typename T>
aoclsparse_status sp2m_t(aoclsparse_operation opA, const aoclsparse_mat_descr descrA, const aoclsparse_matrix A,
                         aoclsparse_operation opB, const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                         aoclsparse_request request, aoclsparse_matrix *C, aoclsparse_matrix_data_type vt)
{
    if(!descrA || !descrB)
        return aoclsparse_status_invalid_pointer;
    if(!A || !B || !C)
        return aoclsparse_status_invalid_pointer;
    if(request != aoclsparse_stage_finalize)
        *C = nullptr;
    if(A->input_format != aoclsparse_csr_mat || B->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(A->val_type != vt || B->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(!A->user.ptr || !B->user.ptr)
        return aoclsparse_status_invalid_pointer;
    auto valid_base = [](int b) { return b == 0 || b == 1; };
    if(!valid_base(descrA->base) || !valid_base(descrB->base))
        return aoclsparse_status_invalid_value;
    if(A->base != descrA->base || B->base != descrB->base)
        return aoclsparse_status_invalid_value;
    if(descrA->type != aoclsparse_matrix_type_general || descrB->type != aoclsparse_matrix_type_general)
        return aoclsparse_status_not_implemented;
    auto is_tr = [](aoclsparse_operation o, bool &ok) {
        ok = o == aoclsparse_operation_none || o == aoclsparse_operation_transpose
             || o == aoclsparse_operation_conjugate_transpose;
        return o != aoclsparse_operation_none;
    };
    bool       okA, okB;
    const bool trA = is_tr(opA, okA), trB = is_tr(opB, okB);
    if(!okA || !okB)
        return aoclsparse_status_invalid_value;
    const aoclsparse_int m_a = trA ? A->n : A->m, n_a = trA ? A->m : A->n;
    const aoclsparse_int m_b = trB ? B->n : B->m, n_b = trB ? B->m : B->n;
    if(n_a != m_b)
        return aoclsparse_status_invalid_size;
    if(request != aoclsparse_stage_nnz_count && request != aoclsparse_stage_finalize
       && request != aoclsparse_stage_full_computation)
        return aoclsparse_status_invalid_value;
    if(m_a == 0 || n_a == 0 || n_b == 0 || A->nnz == 0 || B->nnz == 0)
    {
        if(*C == nullptr) // csr2m.cpp:705-735: valid empty result
            return new_csr_result(C, m_a, n_b, 0, vt, nullptr);
        return aoclsparse_status_success;
    }

    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    try
    {
        // operands in computation orientation (csr2m.cpp:743-830)
        const int  opflag = (trA ? 1 : 0) | (trB ? 2 : 0);
        Operand<T> va, vb, ta, tb;
        view_of<T>(A, va);
        view_of<T>(B, vb);
        const Operand<T> *X = &va, *Y = &vb;
        // op = H on a complex operand: the values are conjugated as the kernel loads them (csr2m.cpp:748-835)
        constexpr bool is_cplx = !std::is_floating_point<T>::value;
        bool           conj_x = is_cplx && opA == aoclsparse_operation_conjugate_transpose;
        bool           conj_y = is_cplx && opB == aoclsparse_operation_conjugate_transpose;
        if(opflag == 3)
        {
            X = &vb, Y = &va; // (B A)^T
            std::swap(conj_x, conj_y);
        }
        else
        {
            // op = T / H: the explicit transpose of csr2m.cpp:771-781.  A handle keeps its transpose (the same stable counting
            // sort, 0-based: build_transpose -- ?mv with op = T uses it too) and its device copy, so that a chain of products with
            // P^T pays for it once; handles whose `trans` slot holds something else (results of a (B A)^T product) transpose here
            auto transposed = [&](const aoclsparse_matrix H, const Operand<T> &v, Operand<T> &t) -> aoclsparse_status {
                if(H->owns_user_arrays)
                {
                    transpose_of<T>(v, t);
                    return aoclsparse_status_success;
                }
                const aoclsparse_status rc = build_transpose(H);
                if(rc != aoclsparse_status_success)
                    return rc;
                std::shared_lock<std::shared_mutex> r(H->guard);
                const HostCsr &h = *H->trans;
                t.m = h.m, t.n = h.n, t.nnz = h.nnz, t.base = h.base;
                t.ptr = h.ptr, t.ind = h.ind, t.val = static_cast<const T *>(h.val);
                return aoclsparse_status_success;
            };
            if(trA)
            {
                st = transposed(A, va, ta);
                X  = &ta;
            }
            if(trB && st == aoclsparse_status_success)
            {
                st = transposed(B, vb, tb);
                Y  = &tb;
            }
            if(st != aoclsparse_status_success)
                return st;
        }
        const aoclsparse_int m = X->m, n = Y->n; // product D = X * Y is m x n
        // diagnostic: AOCLSPARSE_MI355_SP2M_TRACE=1 prints the wall time of every phase of the call (synchronising after each)
        static const bool trace = getenv("AOCLSPARSE_MI355_SP2M_TRACE") != nullptr;
        auto              t_last = std::chrono::steady_clock::now();
        auto              phase  = [&](const char *what) {
            if(!trace)
                return;
            (void)hipStreamSynchronize(rt.stream());
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "sp2m %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        phase("operands (transposes)");

        hipStream_t  s = rt.stream();
        // operands on the device: the handles' own device copies; transposed operands (built on the host above) go through staging
        // slots; A * A sends A once
        DeviceBuffer d_cptr, d_ci, d_cv; // the result's own arrays
        const bool   count = request != aoclsparse_stage_finalize, fill = request != aoclsparse_stage_nnz_count;
        struct DevOp
        {
            const aoclsparse_int *ptr = nullptr, *ind = nullptr;
            const void           *val = nullptr;
        } dx, dy;
        auto resident = [&](const Operand<T> *o, const aoclsparse_matrix H, DevOp &dv) -> aoclsparse_status {
            // an operand that IS the handle's own arrays, or the handle's kept transpose, is used from the handle's device copy,
            // which is made here when no product or optimize has made it yet: the next product with this handle -- a Galerkin
            // chain multiplies the same matrices again and again -- sends nothing
            std::unique_lock<std::shared_mutex> w(H->guard);
            const bool own = o->ptr == H->user.ptr, own_t = H->trans && o->ptr == H->trans->ptr;
            if(!own && !own_t)
                return aoclsparse_status_not_implemented;
            DeviceCsr &dc = own ? H->dev_user : H->dev_trans;
            if(!dc.valid)
            {
                const aoclsparse_status rc = upload_csr(own ? H->user : *H->trans, sizeof(T), dc);
                if(rc != aoclsparse_status_success)
                    return rc;
            }
            dv.ptr = dc.ptr.template as<aoclsparse_int>(), dv.ind = dc.ind.template as<aoclsparse_int>(), dv.val = dc.val.ptr;
            return aoclsparse_status_success;
        };
        const aoclsparse_matrix HX = opflag == 3 ? B : A, HY = opflag == 3 ? A : B;
        auto send = [&](const Operand<T> *o, const aoclsparse_matrix H, DevOp &dv, int slot) {
            aoclsparse_status rc = resident(o, H, dv);
            if(rc != aoclsparse_status_not_implemented)
                return rc;
            void *pp = nullptr, *pi = nullptr, *pv = nullptr;
            rc       = rt.staging(slot, sizeof(aoclsparse_int) * ((size_t)o->m + 1), &pp);
            if(rc == aoclsparse_status_success)
                rc = rt.h2d(pp, o->ptr, sizeof(aoclsparse_int) * ((size_t)o->m + 1));
            if(rc == aoclsparse_status_success)
                rc = rt.staging(slot + 1, sizeof(aoclsparse_int) * (size_t)o->nnz, &pi);
            if(rc == aoclsparse_status_success)
                rc = rt.h2d(pi, o->ind, sizeof(aoclsparse_int) * (size_t)o->nnz);
            if(rc == aoclsparse_status_success && fill)
            {
                rc = rt.staging(slot + 2, sizeof(T) * (size_t)o->nnz, &pv);
                if(rc == aoclsparse_status_success)
                    rc = rt.h2d(pv, o->val, sizeof(T) * (size_t)o->nnz);
            }
            dv.ptr = static_cast<const aoclsparse_int *>(pp), dv.ind = static_cast<const aoclsparse_int *>(pi), dv.val = pv;
            return rc;
        };
        st = send(X, HX, dx, SLOT_XP);
        if(st == aoclsparse_status_success)
        {
            if(Y->ptr == X->ptr && Y->ind == X->ind && Y->val == X->val)
                dy = dx;
            else
                st = send(Y, HY, dy, SLOT_YP);
        }
        if(st != aoclsparse_status_success)
            return st;
        phase("operands to the device");
        // ---- analysis on the device: the upper bound of every row's list, the bins, the rows of every bin (spgemm_kernels.hip) ----
        void *p_cap = nullptr, *p_key = nullptr, *p_small = nullptr;
        st          = rt.staging(SLOT_CAP, sizeof(int) * (size_t)m, &p_cap);
        if(st == aoclsparse_status_success)
            st = rt.staging(SLOT_KEY, sizeof(int) * (size_t)m, &p_key);
        if(st == aoclsparse_status_success)
            st = rt.staging(SLOT_SMALL, 256, &p_small);
        if(st == aoclsparse_status_success)
            st = launch_spg_bounds(s, m, n, X->base, dx.ptr, dx.ind, dy.ptr, static_cast<int *>(p_cap));
        if(st != aoclsparse_status_success)
            return st;
        int          *d_cap  = static_cast<int *>(p_cap);
        unsigned int *d_hist = static_cast<unsigned int *>(p_small), *d_cursor = d_hist + 16, *d_bad = d_hist + 32;
        MI355_HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), s)); // raised by a row whose list does not fit what it was given
        // key = the size of every row's list (count pass: d_cap; fill pass: the exact counts, checked against d_cap)
        auto bin_rows = [&](Binned &bn, bool for_fill, const int *key, const int *limit) -> aoclsparse_status {
            aoclsparse_status rc = launch_spg_hist(s, m, key, limit, for_fill, d_hist);
            if(rc != aoclsparse_status_success)
                return rc;
            unsigned int hist[SPGEMM_BINS + 1];
            MI355_HIP_TRY(hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, s));
            MI355_HIP_TRY(hipStreamSynchronize(s));
            if(hist[SPGEMM_BINS])
                return aoclsparse_status_invalid_value; // a row_ptr that is not this product's (csr2m.cpp:397-399 checks dimensions only)
            bn.bounds[0] = 0;
            bool one_bin = false;
            for(int b = 0; b < SPGEMM_BINS; b++)
            {
                bn.bounds[b + 1] = bn.bounds[b] + (aoclsparse_int)hist[b];
                one_bin |= (aoclsparse_int)hist[b] == m;
            }
            const aoclsparse_int nheavy = (aoclsparse_int)hist[SPGEMM_BINS - 1];
            if(one_bin && nheavy == 0)
                return aoclsparse_status_success; // (no list: workgroup g serves row g)
            void *po = nullptr;
            rc       = rt.staging(for_fill ? SLOT_ORDER_FILL : SLOT_ORDER_COUNT, sizeof(aoclsparse_int) * (size_t)m, &po);
            if(rc == aoclsparse_status_success)
                rc = launch_spg_order(s, m, key, for_fill, bn.bounds, d_cursor, static_cast<aoclsparse_int *>(po));
            if(rc != aoclsparse_status_success)
                return rc;
            bn.d_order = static_cast<const aoclsparse_int *>(po);
            if(nheavy == 0)
                return rc;
            // the rows of the last bin: ids and list sizes to the host, slab offsets back (few rows: the dense ones of a power-law matrix)
            void *pk = nullptr;
            rc       = rt.staging(SLOT_HEAVY_KEYS, sizeof(int) * (size_t)nheavy, &pk);
            if(rc == aoclsparse_status_success)
                rc = launch_spg_gather(s, nheavy, bn.d_order + bn.bounds[SPGEMM_BINS - 1], key, static_cast<int *>(pk));
            if(rc != aoclsparse_status_success)
                return rc;
            std::vector<aoclsparse_int> ids((size_t)nheavy);
            std::vector<int>            caps((size_t)nheavy);
            MI355_HIP_TRY(hipMemcpyAsync(ids.data(), bn.d_order + bn.bounds[SPGEMM_BINS - 1], sizeof(aoclsparse_int) * (size_t)nheavy,
                                         hipMemcpyDeviceToHost, s));
            MI355_HIP_TRY(hipMemcpyAsync(caps.data(), pk, sizeof(int) * (size_t)nheavy, hipMemcpyDeviceToHost, s));
            MI355_HIP_TRY(hipStreamSynchronize(s));
            std::vector<std::pair<aoclsparse_int, int>> rows((size_t)nheavy);
            for(aoclsparse_int j = 0; j < nheavy; j++)
                rows[(size_t)j] = {ids[(size_t)j], caps[(size_t)j]};
            std::sort(rows.begin(), rows.end()); // (the device appends them in arrival order)
            bn.heavy.reserve((size_t)nheavy);
            long long hs = 0, cs = 0;
            bn.batch.push_back(0);
            for(const auto &r : rows)
            {
                const long long cap = r.second;
                if(cap > (1LL << 29))
                    return aoclsparse_status_memory_error;
                int logh = 6;
                while((1LL << logh) < 2 * cap)
                    logh++;
                if(hs + (1LL << logh) > SPG_SLAB_SLOTS && hs > 0)
                {
                    bn.batch.push_back((aoclsparse_int)bn.heavy.size());
                    hs = cs = 0;
                }
                bn.heavy.push_back(SpgHeavy{r.first, logh, hs, cs});
                hs += 1LL << logh, cs += cap;
                bn.slots = std::max(bn.slots, hs), bn.entries = std::max(bn.entries, cs);
            }
            bn.batch.push_back((aoclsparse_int)bn.heavy.size());
            void *ph = nullptr;
            rc       = rt.staging(for_fill ? SLOT_HEAVY_FILL : SLOT_HEAVY_COUNT, sizeof(SpgHeavy) * bn.heavy.size(), &ph);
            if(rc == aoclsparse_status_success)
                rc = rt.h2d(ph, bn.heavy.data(), sizeof(SpgHeavy) * bn.heavy.size());
            bn.d_heavy = static_cast<const SpgHeavy *>(ph);
            return rc;
        };
        phase("upper bounds (device)");
        auto run_pass = [&](bool pass_fill, Binned &bn, const aoclsparse_int *ptr_c, aoclsparse_int *out_i, T *out_v) -> aoclsparse_status {
            aoclsparse_status rc = aoclsparse_status_success;
            for(int b = 0; b < SPGEMM_BINS - 1 && rc == aoclsparse_status_success; b++)
                rc = launch_spgemm_bin<T>(s, pass_fill, b, bn.bounds[b + 1] - bn.bounds[b], bn.d_order ? bn.d_order + bn.bounds[b] : nullptr,
                                          X->base, dx.ptr, dx.ind, static_cast<const T *>(dx.val), Y->base, dy.ptr, dy.ind,
                                          static_cast<const T *>(dy.val), ptr_c, out_i, out_v, conj_x, conj_y, d_bad);
            if(rc != aoclsparse_status_success || bn.heavy.empty())
                return rc;
            void *gk = nullptr, *gp = nullptr, *gl = nullptr, *ga = nullptr;
            rc       = rt.staging(SLOT_G_KEY, sizeof(int) * (size_t)bn.slots, &gk);
            if(pass_fill)
            {
                if(rc == aoclsparse_status_success)
                    rc = rt.staging(SLOT_G_POS, sizeof(int) * (size_t)bn.slots, &gp);
                if(rc == aoclsparse_status_success)
                    rc = rt.staging(SLOT_G_LIST, sizeof(int) * (size_t)bn.entries, &gl);
                if(rc == aoclsparse_status_success)
                    rc = rt.staging(SLOT_G_ACC, sizeof(T) * (size_t)bn.entries, &ga);
            }
            for(size_t b = 0; b + 1 < bn.batch.size() && rc == aoclsparse_status_success; b++) // (stream order: a batch reuses the slab)
                rc = launch_spgemm_heavy<T>(s, pass_fill, bn.batch[b + 1] - bn.batch[b], bn.d_heavy + bn.batch[b], static_cast<int *>(gk),
                                            static_cast<int *>(gp), static_cast<int *>(gl), static_cast<T *>(ga), X->base, dx.ptr, dx.ind,
                                            static_cast<const T *>(dx.val), Y->base, dy.ptr, dy.ind, static_cast<const T *>(dy.val), ptr_c,
                                            out_i, out_v, conj_x, conj_y, d_bad);
            return rc;
        };

        bool ptr_on_device = false; // d_cptr holds D's row_ptr (a full computation goes from the count pass to the fill pass in HBM)
        if(count)
        {
            Binned by_bound;
            st = bin_rows(by_bound, false, d_cap, nullptr);
            phase("count: bins");
            if(st == aoclsparse_status_success)
                st = run_pass(false, by_bound, nullptr, static_cast<aoclsparse_int *>(p_key), nullptr);
            if(st != aoclsparse_status_success)
                return st;
            phase("count: kernels");
            // row_ptr of D = prefix sum of the counts, on the device; the host needs the total now (to allocate) and the array itself
            // for the handle (64-bit sums: a product of more than 2^31 - 1 entries is refused, csr2m.cpp:221-236)
            void      *p_scan = nullptr;
            long long *d_total = nullptr, total = 0;
            st = rt.staging(SLOT_SCAN, spg_scan_scratch_bytes(m), &p_scan);
            if(st == aoclsparse_status_success)
                st = d_cptr.alloc(sizeof(aoclsparse_int) * ((size_t)m + 1));
            if(st == aoclsparse_status_success)
                st = launch_spg_scan(s, m, static_cast<const int *>(p_key), d_cptr.as<aoclsparse_int>(), static_cast<long long *>(p_scan),
                                     &d_total);
            if(st != aoclsparse_status_success)
                return st;
            std::vector<aoclsparse_int> cptr((size_t)m + 1);
            MI355_HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof(long long), hipMemcpyDeviceToHost, s));
            MI355_HIP_TRY(hipMemcpyAsync(cptr.data(), d_cptr.ptr, sizeof(aoclsparse_int) * ((size_t)m + 1), hipMemcpyDeviceToHost, s));
            MI355_HIP_TRY(hipStreamSynchronize(s));
            if(total > 2147483647LL)
                return aoclsparse_status_invalid_size;
            const aoclsparse_int nnz_c = (aoclsparse_int)total;
            ptr_on_device              = true;
            phase("count: scan, row_ptr to host");
            if(opflag == 3)
            {
                // C is n x m; keep D's row_ptr in the handle's transposed-product scratch until finalize
                st = new_csr_result(C, n, m, nnz_c, vt, nullptr);
                if(st != aoclsparse_status_success)
                    return st;
                (*C)->trans.reset(new HostCsr);
                HostCsr &d = *(*C)->trans;
                d.m = m, d.n = n, d.nnz = nnz_c, d.base = aoclsparse_index_base_zero, d.owned = true;
                d.ptr = new aoclsparse_int[(size_t)m + 1];
                d.ind = new aoclsparse_int[(size_t)std::max(nnz_c, 1)];
                d.val = ::operator new(sizeof(T) * (size_t)std::max(nnz_c, 1));
                std::memcpy(d.ptr, cptr.data(), sizeof(aoclsparse_int) * ((size_t)m + 1));
            }
            else
            {
                st = new_csr_result(C, m, n, nnz_c, vt, cptr.data());
                if(st != aoclsparse_status_success)
                    return st;
            }
        }
        phase("result handle");
        if(fill)
        {
            if(*C == nullptr)
                return aoclsparse_status_invalid_pointer;
            _aoclsparse_matrix *c = *C;
            if(c->val_type != vt || !c->owns_user_arrays)
                return aoclsparse_status_invalid_pointer;
            HostCsr *d = opflag == 3 ? c->trans.get() : &c->user; // where the product D lands
            if(!d || !d->ptr || !d->ind || !d->val)
                return aoclsparse_status_invalid_pointer;
            if(d->m != m || d->n != n)
                return aoclsparse_status_invalid_size; // csr2m.cpp:397-399
            const aoclsparse_int nnz_c = d->ptr[m];
            // rows binned by their exact count: the counts of stage 1 are still in HBM after a full computation, a finalize call
            // sends the handle's row_ptr; either way they are checked against the upper bounds on the device (bin_rows)
            if(!ptr_on_device)
            {
                st = d_cptr.upload(d->ptr, sizeof(aoclsparse_int) * ((size_t)m + 1), s);
                if(st == aoclsparse_status_success)
                    st = launch_spg_diff(s, m, d_cptr.as<aoclsparse_int>(), static_cast<int *>(p_key));
                if(st != aoclsparse_status_success)
                    return st;
            }
            Binned by_count;
            st = bin_rows(by_count, true, static_cast<const int *>(p_key), d_cap);
            if(st == aoclsparse_status_success)
                st = d_ci.alloc(sizeof(aoclsparse_int) * (size_t)std::max(nnz_c, 1));
            if(st == aoclsparse_status_success)
                st = d_cv.alloc(sizeof(T) * (size_t)std::max(nnz_c, 1));
            if(st != aoclsparse_status_success)
                return st;
            phase("fill: bins, buffers");
            st = run_pass(true, by_count, d_cptr.as<aoclsparse_int>(), d_ci.as<aoclsparse_int>(), d_cv.as<T>());
            if(st != aoclsparse_status_success)
                return st;
            unsigned int bad = 0;
            if(!ptr_on_device)
            {
                // finalize-only call: the row_ptr is whatever the caller's handle holds.  The verdict of the device checks is read
                // BEFORE anything of *C is touched, so a refused call (the row_ptr of another product) leaves the caller's handle --
                // its arrays, its device copy, its plans -- exactly as it was
                MI355_HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s));
                MI355_HIP_TRY(hipStreamSynchronize(s));
                if(bad)
                    return aoclsparse_status_invalid_value;
            }
            // whatever the handle derived from an earlier fill (device copies, plans, SELL twin, replicas) mirrors the old values
            // (the (B A)^T scratch of stage 1 lives in c->trans, which invalidate would drop: carried across)
            {
                auto keep = std::move(c->trans);
                (void)aoclsparse_mi355_invalidate(c);
                c->trans = std::move(keep);
            }
            // while the kernels run: the result's host arrays are first-touched by several threads (host_result_alloc; faulted in
            // by the copy itself they cost more than the copy: 7-13 ms for 156 MB against ~3); the values' pages are touched while
            // the column indices already travel
            host_result_touch(d->ind, sizeof(aoclsparse_int) * (size_t)nnz_c);
            phase("fill: kernels (+ host pages of col_ind)");
            std::thread toucher;
            try
            {
                toucher = std::thread([&] { host_result_touch(d->val, sizeof(T) * (size_t)nnz_c); });
            }
            catch(const std::system_error &)
            {
                host_result_touch(d->val, sizeof(T) * (size_t)nnz_c);
            }
            const hipError_t e1 = hipMemcpyAsync(d->ind, d_ci.ptr, sizeof(aoclsparse_int) * (size_t)nnz_c, hipMemcpyDeviceToHost, s);
            if(toucher.joinable())
                toucher.join();
            MI355_HIP_TRY(e1);
            MI355_HIP_TRY(hipMemcpyAsync(d->val, d_cv.ptr, sizeof(T) * (size_t)nnz_c, hipMemcpyDeviceToHost, s));
            if(ptr_on_device) // (full computation: the counts are this product's own, the word is read with the result)
                MI355_HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s));
            MI355_HIP_TRY(hipStreamSynchronize(s));
            if(bad) // the row_ptr of *C is not the one stage 1 of THIS product returned (a row ended short of, or beyond, its segment)
                return aoclsparse_status_invalid_value;
            phase("fill: result to host");
            if(opflag != 3)
            {
                // the result stays resident: a product, solve or another sp2m with this handle starts from HBM
                std::unique_lock<std::shared_mutex> w(c->guard);
                DeviceCsr                          &dc = c->dev_user;
                dc.ptr.adopt(d_cptr), dc.ind.adopt(d_ci), dc.val.adopt(d_cv);
                dc.m = m, dc.n = n, dc.nnz = nnz_c, dc.base = aoclsparse_index_base_zero;
                dc.valid = true;
            }
            if(opflag == 3)
            {
                // C = D^T by the reference's counting-sort transpose (csr2m.cpp:520-538)
                Operand<T> dv, dt;
                dv.m = m, dv.n = n, dv.nnz = nnz_c, dv.base = 0;
                dv.ptr = d->ptr, dv.ind = d->ind, dv.val = static_cast<const T *>(d->val);
                transpose_of<T>(dv, dt);
                std::memcpy(c->user.ptr, dt.optr.data(), sizeof(aoclsparse_int) * ((size_t)n + 1));
                std::memcpy(c->user.ind, dt.oind.data(), sizeof(aoclsparse_int) * (size_t)nnz_c);
                std::memcpy(c->user.val, dt.oval.data(), sizeof(T) * (size_t)nnz_c);
            }
        }
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

aoclsparse_status sp2m_any(aoclsparse_operation opA, const aoclsparse_mat_descr descrA, const aoclsparse_matrix A,
                           aoclsparse_operation opB, const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                           aoclsparse_request request, aoclsparse_matrix *C)
{
    // level3/aoclsparse_sp2m.cpp:27-50 dispatches on A's value type
    if(!A || !B || !C || !descrA || !descrB)
        return aoclsparse_status_invalid_pointer;
    if(A->val_type == aoclsparse_dmat)
        return sp2m_t<double>(opA, descrA, A, opB, descrB, B, request, C, aoclsparse_dmat);
    if(A->val_type == aoclsparse_smat)
        return sp2m_t<float>(opA, descrA, A, opB, descrB, B, request, C, aoclsparse_smat);
    if(A->val_type == aoclsparse_zmat)
        return sp2m_t<cdouble>(opA, descrA, A, opB, descrB, B, request, C, aoclsparse_zmat);
    if(A->val_type == aoclsparse_cmat)
        return sp2m_t<cfloat>(opA, descrA, A, opB, descrB, B, request, C, aoclsparse_cmat);
    return aoclsparse_status_not_implemented;
}

} // namespace

extern "C" {

aoclsparse_status aoclsparse_sp2m(aoclsparse_operation opA, const aoclsparse_mat_descr descrA,
                                  const aoclsparse_matrix A, aoclsparse_operation opB,
                                  const aoclsparse_mat_descr descrB, const aoclsparse_matrix B,
                                  const aoclsparse_request request, aoclsparse_matrix *C)
{
    return sp2m_any(opA, descrA, A, opB, descrB, B, request, C);
}

aoclsparse_status aoclsparse_spmm(aoclsparse_operation opA, const aoclsparse_matrix A,
                                  const aoclsparse_matrix B, aoclsparse_matrix *C)
{
    // level3/aoclsparse_spmm.cpp:27-67: general descriptors in each matrix's base, full computation
    if(!A || !B || !C)
        return aoclsparse_status_invalid_pointer;
    _aoclsparse_mat_descr dA, dB;
    dA.base = A->base;
    dB.base = B->base;
    return sp2m_any(opA, &dA, A, aoclsparse_operation_none, &dB, B, aoclsparse_stage_full_computation, C);
}

aoclsparse_status aoclsparse_dcsr2m(aoclsparse_operation trans_A, const aoclsparse_mat_descr descrA,
                                    const aoclsparse_matrix csrA, aoclsparse_operation trans_B,
                                    const aoclsparse_mat_descr descrB, const aoclsparse_matrix csrB,
                                    const aoclsparse_request request, aoclsparse_matrix *csrC)
{
    if(!descrA || !descrB || !csrA || !csrB || !csrC)
        return aoclsparse_status_invalid_pointer;
    return sp2m_t<double>(trans_A, descrA, csrA, trans_B, descrB, csrB, request, csrC, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_scsr2m(aoclsparse_operation trans_A, const aoclsparse_mat_descr descrA,
                                    const aoclsparse_matrix csrA, aoclsparse_operation trans_B,
                                    const aoclsparse_mat_descr descrB, const aoclsparse_matrix csrB,
                                    const aoclsparse_request request, aoclsparse_matrix *csrC)
{
    if(!descrA || !descrB || !csrA || !csrB || !csrC)
        return aoclsparse_status_invalid_pointer;
    return sp2m_t<float>(trans_A, descrA, csrA, trans_B, descrB, csrB, request, csrC, aoclsparse_smat);
}

} // extern "C"
