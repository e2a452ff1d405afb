// trsv_api.cpp -- aoclsparse_?trsv(_kid)(_strided): checks, level-set analysis, dispatch.
//
// Argument checks and their order: level2/aoclsparse_trsv.cpp:59-137 of the reference.  The solve
// itself is the level-scheduled HIP path of trsv_kernels.hip; the analysis (level sets of the
// hinted triangle) runs once per (fill, op) at aoclsparse_optimize after aoclsparse_set_sv_hint,
// or lazily on the first solve, mirroring the reference's lazy aoclsparse_csr_csc_optimize (:128).
#include "internal.hpp"

#include <system_error>
#include <thread>
#include <tuple>

#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <type_traits>

using namespace mi355;

namespace mi355
{

// Host view of the strict triangle one (fill, op) variant walks: row i depends on the rows listed in
// [ptr[i], ptr[i+1]) of ind (0-based), val in the order the reference's chain applies them.
// nnz-sized scratch that every element of is written before it is read: NOT zero-filled (a std::vector of 25 M entries costs
// ~30 ms of single-threaded zeroing and page faults per array; here the first touch happens in the parallel fill loops)
template <typename U>
struct RawArray
{
    std::unique_ptr<U[]> p;
    size_t               n = 0;
    RawArray()             = default;
    explicit RawArray(size_t count) { resize(count); }
    void resize(size_t count)
    {
        p.reset(new U[count]); // default-initialised: no fill for arithmetic types
        n = count;
    }
    U       *data() { return p.get(); }
    const U *data() const { return p.get(); }
    U       *begin() { return p.get(); }
    const U *begin() const { return p.get(); }
    U       *end() { return p.get() + n; }
    const U *end() const { return p.get() + n; }
    size_t   size() const { return n; }
    U       &operator[](size_t i) { return p[i]; }
    const U &operator[](size_t i) const { return p[i]; }
};

// Hundreds of MB of analysis scratch take tens of ms to give back to the kernel (munmap of page-faulted memory): the 25 M-entry
// shell-like factor spent 40 of its 95 ms of block-plan time in destructors.  The arrays are moved into a box that a detached
// thread deletes, off the caller's critical path (if no thread can be started they are freed here, as before).
template <typename... Ts>
static void free_later(Ts &&...xs)
{
    auto *box = new(std::nothrow) std::tuple<std::decay_t<Ts>...>(std::move(xs)...);
    if(!box)
        return; // (the arguments are destroyed by their owners)
    try
    {
        std::thread([box] { delete box; }).detach();
    }
    catch(const std::system_error &)
    {
        delete box;
    }
}

template <typename T>
struct Triangle
{
    std::vector<aoclsparse_int> ptr;
    RawArray<aoclsparse_int>    ind;
    RawArray<T>                 val;
    bool                        descending = false; // solve order m-1..0 (dependencies point to larger rows)
};

template <typename T>
static void build_triangle(const HostCsr &c, bool upper, bool transposed, bool conj, Triangle<T> &t)
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
        parallel_for(m, 1 << 16, [&](long long i0, long long i1) {
            for(aoclsparse_int i = (aoclsparse_int)i0; i < (aoclsparse_int)i1; i++)
                for(aoclsparse_int p = s[i] - b, q = t.ptr[i]; p < e[i] - b; p++, q++)
                {
                    t.ind[q] = c.ind[p] - b;
                    t.val[q] = v[p];
                }
        });
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
    if(conj) // op = H: the column sweep applies conj(a_ic) (ref_trsv_lth / _uth with the conjugating accessor)
        for(auto &a : t.val)
            a = conj_of(a);
}

// level[i] = 1 + max level of the rows row i depends on; rows bucketed by level (counting sort,
// ascending row index inside a level); then the triangle is re-laid out in that order and the hybrid
// schedule (runs of narrow levels vs. wide levels) is derived.
template <typename T>
static aoclsparse_status build_levels(aoclsparse_int m, const Triangle<T> &t, TrsvPlan &plan, bool do_layout)
{
    LapTimer                    lt;
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

    lt.lap("levels: level pass + buckets");
    if(!do_layout)
        return aoclsparse_status_success;
    // level-ordered copy of the triangle; dependencies are rewritten as POSITIONS in that order
    std::vector<aoclsparse_int> pos((size_t)m);
    for(aoclsparse_int k = 0; k < m; k++)
        pos[rowmap[k]] = k;
    std::vector<aoclsparse_int> pptr((size_t)m + 1, 0);
    RawArray<aoclsparse_int>    pind(t.ind.size());
    RawArray<T>                 pval(t.val.size());
    for(aoclsparse_int k = 0; k < m; k++)
        pptr[k + 1] = pptr[k] + (t.ptr[rowmap[k] + 1] - t.ptr[rowmap[k]]);
    parallel_for(m, 1 << 16, [&](long long k0, long long k1) {
        for(aoclsparse_int k = (aoclsparse_int)k0; k < (aoclsparse_int)k1; k++)
        {
            const aoclsparse_int i = rowmap[k], len = t.ptr[i + 1] - t.ptr[i];
            for(aoclsparse_int j = 0; j < len; j++)
                pind[pptr[k] + j] = pos[t.ind[t.ptr[i] + j]];
            std::copy(t.val.begin() + t.ptr[i], t.val.begin() + t.ptr[i + 1], pval.begin() + pptr[k]);
        }
    });
    lt.lap("levels: layout");
    // level slices (<= 64 positions, inside one level) for the slice-per-wavefront sync-free kernel
    std::vector<aoclsparse_int> slices;
    slices.reserve((size_t)m / 48 + (size_t)nlev + 2);
    for(aoclsparse_int l = 0; l < nlev; l++)
        for(aoclsparse_int k = plan.level_ptr[l]; k < plan.level_ptr[l + 1]; k += 64)
            slices.push_back(k);
    slices.push_back(m);
    plan.nslices = (aoclsparse_int)slices.size() - 1;
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
    lt.lap("levels: slices + segments");
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
        rc = plan.slices.upload(slices.data(), sizeof(aoclsparse_int) * slices.size(), st);
    lt.lap("levels: upload");
    plan.rows_valid = rc == aoclsparse_status_success;
    free_later(std::move(pind), std::move(pval), std::move(rowmap), std::move(pos), std::move(pptr), std::move(level));
    return rc;
}

// Blocked (supernodal) plan: see TrsvBlockPlan and trsv_block_kernel.  A row joins the block of the row its chain applies last
// (round 4: wherever that row is numbered; rounds 2-3: only the predecessor in solve order) when its dependency list -- in the order the reference's chain applies it
// -- is exactly the predecessor's list with the predecessor itself
//   * appended at the END  (L, L^T, U^T: the chain runs over the far rows first, the nearest last), or
//   * put at the FRONT     (U: ref_trsv_u walks the row left to right, so the row solved last comes first);
// the block stays within TRSV_BLK_ROWS rows, TRSV_BLK_EXT external dependencies and TRSV_BLK_NV entries.  One triangle
// uses one of the two forms (whichever groups more rows).  Built only when it pays: >= 1.6 rows per block on average.
template <typename T>
static aoclsparse_status build_blocked(aoclsparse_int m, const Triangle<T> &t, TrsvBlockPlan &bp)
{
    bp.tried = true;
    bp.chunk.valid = bp.chunk.tried = false; // (rebuilt below where it applies: never a leftover of an earlier plan)
    if(m < 2)
        return aoclsparse_status_success;
    LapTimer lt;
    auto     row_at = [&](aoclsparse_int k) { return t.descending ? m - 1 - k : k; }; // k-th row in solve order
    auto len_of = [&](aoclsparse_int i) { return t.ptr[i + 1] - t.ptr[i]; };
    // does `row` (lj entries) chain onto `prev` (lq entries)?  front = the predecessor is the FIRST entry
    auto chains = [&](aoclsparse_int row, aoclsparse_int prev, bool front) {
        const aoclsparse_int  lq = len_of(prev), lj = len_of(row);
        const aoclsparse_int *a = &t.ind[t.ptr[row]], *q = &t.ind[t.ptr[prev]];
        if(lj != lq + 1 || a[front ? 0 : lq] != prev)
            return false;
        return lq == 0 || std::memcmp(a + (front ? 1 : 0), q, sizeof(aoclsparse_int) * (size_t)lq) == 0;
    };
    // 1. blocks = CHAINS of the dependency structure (round 4; rounds 2-3 took ranges of the solve order, which only finds the
    // chains of a matrix whose chained rows are numbered consecutively -- the dofs of a mesh node in natural order -- and lost
    // them all on a renumbered mesh: 7,315 row levels at 0.62 us instead of ~1,100 block levels).  Row j can continue row p's
    // block when p is the dependency its chain applies LAST (or FIRST, form `front`) and the rest of its list is exactly p's
    // list: then everything j waits for outside the block, p's block has already waited for.  Every row has at most one
    // follower (the first candidate in solve order).  A block's first row is solved before every other row of it, and the
    // blocks' dependencies point to blocks with an earlier first row only: numbered by first row they are in topological order.
    // The kernel is compiled for blocks of up to 5 rows (every value in registers across the wait) and up to TRSV_BLK_ROWS
    // (row by row, values in LDS): when only a few chains run longer than 5 rows they are cut at 5, so that one long chain does
    // not put the whole solve on the slower shape.
    // One grouping = blocks (bptr / brows), their levels, the widest block and the most external dependencies a multi-row
    // block has.  ext_cap: a block is only grown from a first row with at most that many entries.
    struct Grouping
    {
        std::vector<aoclsparse_int> bptr, brows, blev;
        aoclsparse_int              nlev = 0;
        int                         max_rows = 1, max_ext = 0;
        bool                        front = false;
    };
    std::vector<aoclsparse_int> follower((size_t)m);
    std::vector<char>           taken((size_t)m);
    auto group = [&](int ext_cap, Grouping &G) {
        std::vector<aoclsparse_int> bptr, brows;
        for(int form = 0; form < 2; form++)
        {
            const bool front = form == 1;
            std::fill(follower.begin(), follower.end(), (aoclsparse_int)-1);
            for(aoclsparse_int k = 0; k < m; k++)
            {
                const aoclsparse_int j = row_at(k), lj = len_of(j);
                if(lj == 0)
                    continue;
                const aoclsparse_int pr = t.ind[t.ptr[j] + (front ? 0 : lj - 1)];
                if(follower[pr] < 0 && chains(j, pr, front))
                    follower[pr] = j;
            }
            for(int cap : {TRSV_BLK_ROWS, 5})
            {
                bptr.clear(), brows.clear();
                bptr.reserve((size_t)m / 2 + 2), brows.reserve((size_t)m);
                std::fill(taken.begin(), taken.end(), 0);
                aoclsparse_int longer = 0;
                for(aoclsparse_int k = 0; k < m; k++)
                {
                    aoclsparse_int j = row_at(k);
                    if(taken[j])
                        continue;
                    bptr.push_back((aoclsparse_int)brows.size());
                    brows.push_back(j), taken[j] = 1;
                    const aoclsparse_int n0 = len_of(j);
                    aoclsparse_int       rows = 1, total = n0;
                    if(n0 <= ext_cap)
                        while(rows < cap)
                        {
                            const aoclsparse_int f = follower[j];
                            if(f < 0 || taken[f] || total + len_of(f) > TRSV_BLK_NV)
                                break;
                            brows.push_back(f), taken[f] = 1;
                            total += len_of(f), rows++, j = f;
                        }
                    longer += (rows > 5);
                }
                if(longer == 0 || longer * 10 >= (aoclsparse_int)bptr.size())
                    break; // nothing to cut, or long chains are the rule: keep them
            }
            bptr.push_back((aoclsparse_int)brows.size());
            if(G.bptr.empty() || bptr.size() < G.bptr.size())
                G.bptr = bptr, G.brows = brows, G.front = front;
            if((G.bptr.size() - 1) * 16 <= (size_t)m * 10)
                break; // this form already groups the rows
        }
        const aoclsparse_int nb = (aoclsparse_int)G.bptr.size() - 1;
        G.max_rows = 1, G.max_ext = 0;
        for(aoclsparse_int bq = 0; bq < nb; bq++)
        {
            const int rows = G.bptr[bq + 1] - G.bptr[bq];
            G.max_rows     = std::max(G.max_rows, rows);
            // a single row longer than the cap is served by the kernel's tail loop: it does not widen the unrolled part
            if(rows > 1 || len_of(G.brows[G.bptr[bq]]) <= ext_cap)
                G.max_ext = std::max(G.max_ext, std::min<int>(len_of(G.brows[G.bptr[bq]]), ext_cap));
        }
        // block levels (a block's external dependencies are those of its first-solved row; blocks are numbered by first row in
        // solve order, so every dependency's block is already levelled)
        std::vector<aoclsparse_int> bof((size_t)m);
        G.blev.assign((size_t)nb, 0);
        for(aoclsparse_int bq = 0; bq < nb; bq++)
            for(aoclsparse_int k = G.bptr[bq]; k < G.bptr[bq + 1]; k++)
                bof[G.brows[k]] = bq;
        G.nlev = 0;
        for(aoclsparse_int bq = 0; bq < nb; bq++)
        {
            const aoclsparse_int r  = G.brows[G.bptr[bq]];
            aoclsparse_int       lv = 0;
            for(aoclsparse_int p = t.ptr[r]; p < t.ptr[r + 1]; p++)
                lv = std::max(lv, G.blev[bof[t.ind[p]]] + 1);
            G.blev[bq] = lv;
            G.nlev     = std::max(G.nlev, lv + 1);
        }
    };
    Grouping G;
    group(TRSV_BLK_EXT, G);
    lt.lap("blocks: chains + levels");
    // (Growing blocks only from rows of <= 16 entries -- so that every block fits the in-register shape of the kernel -- was tried
    // on the unstructured shell-like factor, 45 % of whose rows have 16-24 entries: 534,653 blocks in 3,061 levels instead of
    // 359,873 in 1,784; not a trade.)
    const std::vector<aoclsparse_int> &bptr = G.bptr, &brows = G.brows, &blev = G.blev;
    const bool                         front = G.front;
    const aoclsparse_int               nb = (aoclsparse_int)bptr.size() - 1, nlev = G.nlev;
    if((long long)nb * 16 > (long long)m * 10)
        return aoclsparse_status_success; // fewer than 1.6 rows per block: the row-level schedules are as good
    const int max_rows = G.max_rows, max_ext = G.max_ext;
    // 3. blocks in level order (stable), positions of their rows (in solve order inside a block)
    std::vector<aoclsparse_int> lptr((size_t)nlev + 1, 0);
    for(aoclsparse_int bq = 0; bq < nb; bq++)
        lptr[blev[bq] + 1]++;
    for(aoclsparse_int l = 0; l < nlev; l++)
        lptr[l + 1] += lptr[l];
    std::vector<aoclsparse_int> order((size_t)nb), next(lptr.begin(), lptr.end() - 1);
    for(aoclsparse_int bq = 0; bq < nb; bq++)
        order[next[blev[bq]]++] = bq;
    // ... and inside a level, by the place of a block's LAST dependency in the level below (stable: the natural order where that says
    // nothing -- a mesh numbered line by line is left as it is).  Neighbours in a slice then wait for the same producer slices: fan-in of a
    // slice 15.7 -> 13.5 on the unstructured shell-like factor, L 3.15 -> 3.00 ms, U 3.25 -> 3.06 (profiles/r6/trsv_chunk_experiments.txt
    // v17; the first, the mean or two levels of dependencies as the key: the same within 1 %).  Positions change, chains do not: same bits.
    {
        std::vector<aoclsparse_int> bofx((size_t)m), rank((size_t)nb, 0);
        for(aoclsparse_int bq = 0; bq < nb; bq++)
            for(aoclsparse_int k = bptr[bq]; k < bptr[bq + 1]; k++)
                bofx[brows[k]] = bq;
        for(aoclsparse_int k = lptr[0]; k < lptr[1]; k++)
            rank[order[k]] = k - lptr[0];
        std::vector<std::pair<aoclsparse_int, aoclsparse_int>> keyed;
        for(aoclsparse_int l = 1; l < nlev; l++)
        {
            keyed.clear();
            for(aoclsparse_int k = lptr[l]; k < lptr[l + 1]; k++)
            {
                const aoclsparse_int bq = order[k], r = brows[bptr[bq]];
                aoclsparse_int       key = -1;
                for(aoclsparse_int p = t.ptr[r]; p < t.ptr[r + 1]; p++)
                {
                    const aoclsparse_int d = bofx[t.ind[p]];
                    if(blev[d] == l - 1)
                        key = std::max(key, rank[d]);
                }
                keyed.push_back({key, bq});
            }
            std::stable_sort(keyed.begin(), keyed.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
            for(size_t i = 0; i < keyed.size(); i++)
                order[lptr[l] + (aoclsparse_int)i] = keyed[i].second, rank[keyed[i].second] = (aoclsparse_int)i;
        }
    }
    std::vector<aoclsparse_int> bfirst((size_t)nb + 1, 0), rowmap((size_t)m), pos((size_t)m);
    for(aoclsparse_int k = 0; k < nb; k++)
    {
        const aoclsparse_int bq = order[k];
        bfirst[k + 1]           = bfirst[k] + (bptr[bq + 1] - bptr[bq]);
        for(aoclsparse_int kk = bptr[bq], q = bfirst[k]; kk < bptr[bq + 1]; kk++, q++)
            rowmap[q] = brows[kk], pos[brows[kk]] = q;
    }
    // 4. the triangle in that order (entries in chain order), dependencies as positions
    std::vector<aoclsparse_int> pptr((size_t)m + 1, 0);
    RawArray<aoclsparse_int>    pind(t.ind.size());
    RawArray<T>                 pval(t.val.size());
    for(aoclsparse_int k = 0; k < m; k++)
        pptr[k + 1] = pptr[k] + len_of(rowmap[k]);
    parallel_for(m, 1 << 16, [&](long long k0, long long k1) {
        for(aoclsparse_int k = (aoclsparse_int)k0; k < (aoclsparse_int)k1; k++)
        {
            const aoclsparse_int i = rowmap[k], len = len_of(i);
            for(aoclsparse_int j = 0; j < len; j++)
                pind[pptr[k] + j] = pos[t.ind[t.ptr[i] + j]];
            std::copy(t.val.begin() + t.ptr[i], t.val.begin() + t.ptr[i + 1], pval.begin() + pptr[k]);
        }
    });
    lt.lap("blocks: order + layout");
    // 5. slices of <= 64 blocks inside one block level
    std::vector<aoclsparse_int> slices;
    // How many blocks share a wavefront.  A slice starts when the LAST dependency of its 64 blocks is in; when those dependencies come from
    // many producer slices (an irregular numbering: 15.7 on average on the unstructured shell-like factor, 3.6 on the structured one --
    // there a slice waits for the slices at the same place one and two levels down), narrower slices wait for less: slices of 32 blocks
    // 3.42 -> 3.14 ms on the unstructured factor (48: 3.30, 40: 3.22, 24: 3.16, 16: 3.41), and 1.86 -> 2.19 on the structured one, whose
    // slices are full either way (profiles/r6/trsv_chunk_experiments.txt).  Fan-in above 8: 32 blocks per slice.
    int SW = 64;
    {
        std::vector<aoclsparse_int> blk_of_pos((size_t)m), slice_of_blk((size_t)nb), seen;
        for(aoclsparse_int k = 0; k < nb; k++)
            for(aoclsparse_int q = bfirst[k]; q < bfirst[k + 1]; q++)
                blk_of_pos[q] = k;
        aoclsparse_int ns = 0;
        for(aoclsparse_int l = 0; l < nlev; l++)
            for(aoclsparse_int k = lptr[l]; k < lptr[l + 1]; k += 64, ns++)
                for(aoclsparse_int kk = k; kk < std::min<aoclsparse_int>(k + 64, lptr[l + 1]); kk++)
                    slice_of_blk[kk] = ns;
        long long fan = 0;
        for(aoclsparse_int l = 0; l < nlev; l++)
            for(aoclsparse_int k = lptr[l]; k < lptr[l + 1]; k += 64)
            {
                seen.clear();
                for(aoclsparse_int kk = k; kk < std::min<aoclsparse_int>(k + 64, lptr[l + 1]); kk++)
                    for(aoclsparse_int p = pptr[bfirst[kk]]; p < pptr[bfirst[kk] + 1]; p++)
                        seen.push_back(slice_of_blk[blk_of_pos[pind[p]]]);
                std::sort(seen.begin(), seen.end());
                fan += std::unique(seen.begin(), seen.end()) - seen.begin();
            }
        bp.slice_fan_in = ns > 0 ? (double)fan / (double)ns : 0.0;
        if(bp.slice_fan_in > 8.0)
            SW = 32;
    }
    for(aoclsparse_int l = 0; l < nlev; l++)
        for(aoclsparse_int k = lptr[l]; k < lptr[l + 1]; k += SW)
            slices.push_back(k);
    slices.push_back(nb);
    {
        // ... followed by each slice's block level, and the first slice of each level (nlev + 1 entries): the kernel's
        // gate counts finished slices per level
        const size_t ns = slices.size() - 1;
        for(size_t q = 0; q < ns; q++)
            slices.push_back(blev[order[slices[q]]]);
        aoclsparse_int first = 0;
        for(aoclsparse_int l = 0; l < nlev; l++)
        {
            slices.push_back(first);
            first += (lptr[l + 1] - lptr[l] + SW - 1) / SW;
        }
        slices.push_back(first);
    }
    hipStream_t       st = Runtime::get().stream();
    aoclsparse_status rc = bp.rowmap.upload(rowmap.data(), sizeof(aoclsparse_int) * (size_t)m, st);
    if(rc == aoclsparse_status_success)
        rc = bp.pptr.upload(pptr.data(), sizeof(aoclsparse_int) * ((size_t)m + 1), st);
    if(rc == aoclsparse_status_success)
        rc = bp.pind.upload(pind.data(), sizeof(aoclsparse_int) * pind.size(), st);
    if(rc == aoclsparse_status_success)
        rc = bp.pval.upload(pval.data(), sizeof(T) * pval.size(), st);
    if(rc == aoclsparse_status_success)
        rc = bp.bfirst.upload(bfirst.data(), sizeof(aoclsparse_int) * bfirst.size(), st);
    if(rc == aoclsparse_status_success)
        rc = bp.slices.upload(slices.data(), sizeof(aoclsparse_int) * slices.size(), st);
    if(rc != aoclsparse_status_success)
        return rc;
    lt.lap("blocks: upload");
    bp.nblocks = nb, bp.nslices = (aoclsparse_int)(slices.size() - 2 - (size_t)nlev) / 2, bp.nlevels = nlev;
    bp.max_rows = max_rows, bp.max_ext = max_ext;
    bp.front = front;
    bp.valid = true;
    // 6. the two-level schedule (TrsvChunkPlan): chunks of consecutive blocks in natural order, each walked in block-level order;
    // only for shapes the kernel is compiled for.
    bp.chunk.tried = true;
    if(max_rows <= TRSV_CHUNK_LANES && max_ext <= TRSV_BLK_EXT && nb >= 64)
    {
        TrsvChunkPlan &cp = bp.chunk;
        constexpr int  NBS = 64 / TRSV_CHUNK_LANES; // blocks per step
        constexpr int  NCW = TRSV_CHUNK_WAVES - 1; // wavefronts that take steps (the last one fetches the halo)
        // rows per chunk: as many as the LDS holds next to the chunk's halo (the rows of EARLIER chunks it depends on, copied into
        // LDS by the fetching wavefront), but at least ~32 chunks on large triangles (a chunk streams its part of the matrix with
        // one workgroup)
        // (the kernel is compiled for four shapes; the larger ones leave less LDS for the chunk's words: trsv_chunk_slots)
        const aoclsparse_int slots_max = max_rows <= 5 ? (max_ext <= 16 ? trsv_chunk_slots(5, 16) : trsv_chunk_slots(5, TRSV_BLK_EXT))
                                                       : (max_ext <= 16 ? trsv_chunk_slots(TRSV_CHUNK_LANES, 16)
                                                                        : trsv_chunk_slots(TRSV_CHUNK_LANES, TRSV_BLK_EXT));
        static const char   *cap_env = getenv("AOCLSPARSE_MI355_TRSV_CHUNK_ROWS"); // (diagnostics: rows per chunk)
        const aoclsparse_int cap     = cap_env ? std::min<aoclsparse_int>(slots_max, std::max(64, atoi(cap_env)))
                                               : std::min<aoclsparse_int>(slots_max, std::max<aoclsparse_int>(2048, m / 32));
        std::vector<aoclsparse_int> kof((size_t)nb), chunk_of((size_t)nb), bof2((size_t)m); // bof2: natural block index of a row
        for(aoclsparse_int k = 0; k < nb; k++)
            kof[order[k]] = k;
        for(aoclsparse_int bq = 0; bq < nb; bq++)
            for(aoclsparse_int kk = bptr[bq]; kk < bptr[bq + 1]; kk++)
                bof2[brows[kk]] = bq;
        std::vector<aoclsparse_int> cfirst; // first block (natural index) of every chunk
        std::vector<aoclsparse_int> stamp((size_t)m, -1); // row counted in the halo of chunk stamp[row]
        bool                        fits = true;
        {
            aoclsparse_int rows = 0, halo = 0, c = 0, bq0 = 0;
            cfirst.push_back(0);
            for(aoclsparse_int bq = 0; bq < nb && fits; bq++)
            {
                const aoclsparse_int r = bptr[bq + 1] - bptr[bq], first = brows[bptr[bq]];
                for(;;)
                {
                    aoclsparse_int add = 0;
                    for(aoclsparse_int p = t.ptr[first]; p < t.ptr[first + 1]; p++)
                        if(bof2[t.ind[p]] < bq0 && stamp[t.ind[p]] != c)
                            stamp[t.ind[p]] = c, add++;
                    if(rows + r <= cap && rows + r + halo + add <= slots_max)
                    {
                        rows += r, halo += add;
                        break;
                    }
                    if(rows == 0) // a block that does not fit a chunk of its own (a row with ~15,000 dependencies)
                    {
                        fits = false;
                        break;
                    }
                    cfirst.push_back(bq), c++, bq0 = bq, rows = 0, halo = 0; // close the chunk in front of this block; count again
                }
                chunk_of[bq] = c;
            }
            cfirst.push_back(nb);
        }
        const aoclsparse_int        nch = (aoclsparse_int)cfirst.size() - 1;
        std::vector<aoclsparse_int> steps, cptr((size_t)nch + 1, 0), crows((size_t)nch, 0), hptr((size_t)nch + 1, 0), slot_of((size_t)m);
        std::vector<aoclsparse_int> hind, ks, hslot((size_t)(fits ? m : 0)), step_of((size_t)(fits ? nb : 0)), hcount((size_t)nch, 0);
        std::vector<std::pair<aoclsparse_int, aoclsparse_int>> hl; // (first step that needs it, position)
        steps.reserve((size_t)nb);
        aoclsparse_int maxslots = 0;
        for(aoclsparse_int c = 0; c < nch && fits; c++)
        {
            ks.clear();
            for(aoclsparse_int bq = cfirst[c]; bq < cfirst[c + 1]; bq++)
                ks.push_back(kof[bq]);
            std::sort(ks.begin(), ks.end()); // = (block level, natural index): the level order is stable
            aoclsparse_int slot = 0;
            hl.clear();
            for(size_t i = 0; i < ks.size();)
            {
                size_t               j  = i + 1;
                const aoclsparse_int lv = blev[order[ks[i]]];
                while(j < ks.size() && j - i < (size_t)NBS && ks[j] == ks[j - 1] + 1 && blev[order[ks[j]]] == lv)
                    j++;
                // header of the step, 8 words: first block (index in block-level order), position of its first row, LDS slot of
                // that row, rows of the (<= 8) blocks as nibbles; rows in front of block j as bytes (2 words), rows of the step, blocks
                unsigned cw = 0, pre[2] = {0, 0};
                int      rows_before = 0;
                const aoclsparse_int sidx = (aoclsparse_int)(steps.size() / 8);
                for(size_t q = i; q < j; q++)
                {
                    const int rws = bfirst[ks[q] + 1] - bfirst[ks[q]];
                    cw |= (unsigned)rws << (4 * (q - i));
                    pre[(q - i) / 4] |= (unsigned)rows_before << (8 * ((q - i) % 4));
                    rows_before += rws;
                    step_of[ks[q]] = sidx;
                    const aoclsparse_int first = rowmap[bfirst[ks[q]]];
                    for(aoclsparse_int p = t.ptr[first]; p < t.ptr[first + 1]; p++)
                        if(chunk_of[bof2[t.ind[p]]] != c && stamp[t.ind[p]] != -2 - c) // the halo: first use decides the order
                            stamp[t.ind[p]] = -2 - c, hl.emplace_back(sidx, pos[t.ind[p]]);
                }
                steps.push_back(ks[i]), steps.push_back(bfirst[ks[i]]), steps.push_back(slot), steps.push_back((aoclsparse_int)cw);
                steps.push_back((aoclsparse_int)pre[0]), steps.push_back((aoclsparse_int)pre[1]), steps.push_back(rows_before);
                steps.push_back((aoclsparse_int)(j - i));
                for(aoclsparse_int q = bfirst[ks[i]]; q < bfirst[ks[j - 1] + 1]; q++)
                    slot_of[q] = slot++;
                i = j;
            }
            std::sort(hl.begin(), hl.end());
            for(size_t i = 0; i < hl.size(); i++)
                hind.push_back(hl[i].second);
            hcount[c] = (aoclsparse_int)hl.size();
            while(hind.size() % 4)
                hind.push_back(0); // (the fetching wavefront reads four positions per lane with one load)
            crows[c]    = slot;
            hptr[c + 1] = (aoclsparse_int)hind.size();
            maxslots    = std::max<aoclsparse_int>(maxslots, slot + (aoclsparse_int)hl.size());
            cptr[c + 1] = (aoclsparse_int)(steps.size() / 8);
            // this chunk's dependency lists right away (hslot is per chunk): LDS slots of its own rows and of its halo
        }
        // external dependency lists (those of a block's first-solved row), indexed like bfirst; every entry an LDS slot of the
        // block's chunk.  hslot[] of a position is valid for ONE chunk at a time, so the lists are written chunk by chunk.
        std::vector<aoclsparse_int> eptr((size_t)nb + 1, 0);
        for(aoclsparse_int k = 0; k < nb; k++)
            eptr[k + 1] = eptr[k] + len_of(rowmap[bfirst[k]]);
        std::vector<aoclsparse_int> cind((size_t)eptr[nb] + 256, 0); // (padded: the kernel reads whole rounds of 64 words)
        for(aoclsparse_int c = 0; c < nch && fits; c++)
        {
            for(aoclsparse_int i = hptr[c]; i < hptr[c] + hcount[c]; i++)
                hslot[hind[i]] = crows[c] + (i - hptr[c]);
            for(aoclsparse_int bq = cfirst[c]; bq < cfirst[c + 1]; bq++)
            {
                const aoclsparse_int k = kof[bq], r = rowmap[bfirst[k]];
                for(aoclsparse_int jj = 0; jj < len_of(r); jj++)
                {
                    const aoclsparse_int dep = t.ind[t.ptr[r] + jj];
                    cind[eptr[k] + jj]       = chunk_of[bof2[dep]] == c ? slot_of[pos[dep]] : hslot[pos[dep]];
                }
            }
        }
        // plan-time model of both schedules (costs in us from the traces: profiles/r5/trsv_experiments.txt, profiles/r6/): a step
        // = the later of {its wavefront free + the latency of its values, its last dependency + the hand-off} + the work
        // (round 6, profiles/r6/trsv_chunk_trace*.txt: a hand-off through LDS 0.45, solving a step 0.4, a wavefront's loads for a step
        // 1.7, a value of another chunk 2.5 us after it was produced)
        // (FRONT: the rows of a block are phases one after the other, ~0.08 us each on top)
        const double work = front ? 0.4 + 0.08 * max_rows : 0.4, local = 0.45, remote = 2.5, vals = 1.7;
        double       total = 0.0;
        if(fits)
        {
            std::vector<double> fin((size_t)nb, 0.0);
            for(aoclsparse_int c = 0; c < nch; c++)
            {
                double wfree[NCW] = {0};
                for(aoclsparse_int sidx = cptr[c]; sidx < cptr[c + 1]; sidx++)
                {
                    const int            w    = (int)((sidx - cptr[c]) % NCW);
                    double               when = wfree[w] + vals;
                    const aoclsparse_int kf = steps[8 * (size_t)sidx], kn = kf + steps[8 * (size_t)sidx + 7];
                    for(aoclsparse_int k = kf; k < kn; k++)
                    {
                        const aoclsparse_int r = rowmap[bfirst[k]];
                        for(aoclsparse_int jj = t.ptr[r]; jj < t.ptr[r + 1]; jj++)
                        {
                            const aoclsparse_int d = bof2[t.ind[jj]];
                            when = std::max(when, fin[d] + (chunk_of[d] == c ? local : remote));
                        }
                    }
                    const double f = when + work;
                    for(aoclsparse_int k = kf; k < kn; k++)
                        fin[order[k]] = f;
                    wfree[w] = f;
                    total    = std::max(total, f);
                }
            }
        }
        cp.model_us       = total;
        // (the lane-per-block schedule, measured per block level: 1.69 us with every block in registers, 2.51 us with the larger shape)
        cp.model_block_us = (double)nlev * (max_rows <= 5 ? (max_ext <= 16 ? 1.7 : (max_ext <= 20 ? (bp.slice_fan_in > 8.0 ? 2.15 : 2.35) : 2.55)) : 2.55);
        lt.lap("chunks: steps + dependency lists + model");
        // aoclsparse_mi355_set_option(trsv_chunks, ...): -1 the model decides (default), 0 never, 1 whenever the plan can be built
        const int want = plan_option(aoclsparse_mi355_option_trsv_chunks);
        if(fits && want != 0 && (cp.model_us < 0.9 * cp.model_block_us || want == 1))
        {
            // cptr: first step of every chunk (nch + 1), rows of every chunk (nch), first halo entry of every chunk (nch, each a
            // multiple of 4), halo entries of every chunk (nch)
            std::vector<aoclsparse_int> cp2(cptr);
            cp2.insert(cp2.end(), crows.begin(), crows.end());
            cp2.insert(cp2.end(), hptr.begin(), hptr.end() - 1);
            cp2.insert(cp2.end(), hcount.begin(), hcount.end());
            hind.resize(hind.size() + 256 * 4, 0); // (a round reads 256 positions whatever is left of the list)
            rc = cp.steps.upload(steps.data(), sizeof(aoclsparse_int) * steps.size(), st);
            if(rc == aoclsparse_status_success)
                rc = cp.cptr.upload(cp2.data(), sizeof(aoclsparse_int) * cp2.size(), st);
            if(rc == aoclsparse_status_success)
                rc = cp.eptr.upload(eptr.data(), sizeof(aoclsparse_int) * eptr.size(), st);
            if(rc == aoclsparse_status_success)
                rc = cp.cind.upload(cind.data(), sizeof(aoclsparse_int) * cind.size(), st);
            if(rc == aoclsparse_status_success)
                rc = cp.hind.upload(hind.data(), sizeof(aoclsparse_int) * hind.size(), st);
            if(rc == aoclsparse_status_success)
                rc = hipStreamSynchronize(st) == hipSuccess ? rc : aoclsparse_status_internal_error; // (the host vectors die here)
            if(rc == aoclsparse_status_success)
            {
                cp.nchunks = nch, cp.nsteps = (aoclsparse_int)(steps.size() / 8), cp.max_rows = maxslots;
                cp.valid = true;
            }
            lt.lap("chunks: upload");
        }
    }
    free_later(std::move(pind), std::move(pval), std::move(rowmap), std::move(pos), std::move(pptr), std::move(G.brows));
    return aoclsparse_status_success;
}

template <typename T>
static aoclsparse_status build_plan_t(const HostCsr &c, bool upper, bool transposed, bool conj, TrsvPlan &plan, bool need_rows)
{
    Triangle<T> t;
    {
        PhaseTimer pt("trsv plan: triangle");
        build_triangle<T>(c, upper, transposed, conj, t);
    }
    aoclsparse_status st = aoclsparse_status_success;
    constexpr bool    real = std::is_floating_point<T>::value;
    if(!plan.valid)
    {
        // levels first (cheap; every schedule choice needs nlevels), then the block plan, then -- only if asked for, or if
        // the triangle has no blocks -- the level-ordered row layout.  On the 25 M-entry shell-like factor the row layout is
        // ~75 ms of page-faulting, filling, uploading and freeing 300 MB that the automatic schedule (blocks) never reads.
        {
            PhaseTimer pt("trsv plan: level pass");
            st = build_levels<T>(c.m, t, plan, false);
        }
        if constexpr(real)
            if(st == aoclsparse_status_success && !plan.blk.tried)
            {
                PhaseTimer pt("trsv plan: blocks + layout + upload");
                st = build_blocked<T>(c.m, t, plan.blk);
            }
    }
    if(st == aoclsparse_status_success && !plan.rows_valid && (need_rows || !real || !plan.blk.valid))
    {
        PhaseTimer pt("trsv plan: levels + layout + upload");
        st = build_levels<T>(c.m, t, plan, true);
    }
    free_later(std::move(t.ptr), std::move(t.ind), std::move(t.val));
    return st;
}

aoclsparse_status ensure_trsv(aoclsparse_matrix A, bool upper, bool transposed, bool conj, bool need_rows)
{
    conj = conj && transposed && is_complex_type(A->val_type);
    aoclsparse_status st;
    {
        PhaseTimer pt("trsv plan: csr_optimize (if needed)");
        st = csr_optimize(A);
    }
    if(st != aoclsparse_status_success)
        return st;
    TrsvPlan &plan = A->trsv_plan[conj ? 4 + (upper ? 1 : 0) : (upper ? 2 : 0) + (transposed ? 1 : 0)];
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if(plan.valid && (plan.rows_valid || !need_rows))
            return aoclsparse_status_success;
    }
    std::unique_lock<std::shared_mutex> w(A->guard);
    if(plan.valid && (plan.rows_valid || !need_rows))
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
            if(st != aoclsparse_status_success)
                return st;
        }
        st = dispatch_value_type(A->val_type, [&](auto tag) {
            return build_plan_t<decltype(tag)>(c, upper, transposed, conj, plan, need_rows);
        });
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

#define MI355_TRY(expr)                       \
    do                                        \
    {                                         \
        aoclsparse_status st__ = (expr);      \
        if(st__ != aoclsparse_status_success) \
            return st__;                      \
    } while(0)

// Shared tail of trsv / trsm: plan lookup, schedule choice, staging of host operands, launch.
// b / x describe nrhs right-hand sides: column c at b + c*b_off (element stride incb), likewise x.
// span_b / span_x = number of elements of the caller's arrays that the strided views cover.
template <typename T>
aoclsparse_status solve_core(aoclsparse_operation trans, T alpha, aoclsparse_matrix A,
                             const aoclsparse_mat_descr descr, aoclsparse_int kid, const T *b, T *x,
                             aoclsparse_int nrhs, long long b_off, aoclsparse_int incb, size_t span_b,
                             long long x_off, aoclsparse_int incx, size_t span_x, bool x_partial)
{
    const aoclsparse_int m  = A->m;
    Runtime             &rt = Runtime::get();
    aoclsparse_status    st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    const bool upper = descr->fill_mode == aoclsparse_fill_mode_upper;
    const bool tr    = trans != aoclsparse_operation_none;
    const bool unit  = descr->diag_type == aoclsparse_diag_type_unit;
    constexpr bool is_cplx = !std::is_floating_point<T>::value;
    const bool     conj    = is_cplx && trans == aoclsparse_operation_conjugate_transpose;
    st                     = ensure_trsv(A, upper, tr, conj, /*need_rows=*/false);
    if(st != aoclsparse_status_success)
        return st;
    {
        // the level-ordered row layout is needed by the per-level, hybrid and slice schedules (and by complex types); the
        // lane-per-position kernel runs on the block plan's layout too (any topological order of the rows will do)
        bool rows_needed;
        {
            std::shared_lock<std::shared_mutex> r0(A->guard);
            const TrsvPlan &p0 = A->trsv_plan[conj ? 4 + (upper ? 1 : 0) : (upper ? 2 : 0) + (tr ? 1 : 0)];
            const int       f0 = Runtime::primary().trsv_schedule;
            rows_needed = !p0.rows_valid
                          && (is_cplx || !p0.blk.valid || f0 == 0 || f0 == 1 || f0 == 3 || (f0 < 0 && p0.nlevels <= 32));
        }
        if(rows_needed)
        {
            st = ensure_trsv(A, upper, tr, conj, true);
            if(st != aoclsparse_status_success)
                return st;
        }
    }
    // solves on one handle share its workspaces: serialise their enqueue (kernels are stream-ordered)
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    std::shared_lock<std::shared_mutex>   r(A->guard);
    const TrsvPlan &plan = A->trsv_plan[conj ? 4 + (upper ? 1 : 0) : (upper ? 2 : 0) + (tr ? 1 : 0)];

    // schedule: every one of the five gives the same bits.  A shallow DAG of wide levels is cheapest as plain launches;
    // otherwise sync-free, which measured fastest on both the 2-D Laplacian and the shell-like ILU(0) factors.
    // (complex handles always run the hybrid schedule: their 8 / 16-byte x cannot be the one-word ready flag)
    // The sync-free choice is the slice-per-wavefront kernel (3) for one right-hand side -- unless the level slices
    // would leave most lanes idle (average level narrower than 16 rows: deep chains), where the lane-per-position
    // kernel (2) packs better; aoclsparse_mi355_set_trsv_schedule forces one of them.
    // measured (profiles/r2/trsv_schedules.txt): the slice kernel wins on short rows (ILU(0) of the 2-D Laplacian:
    // 1.69 vs 2.09 ms), the lane-per-position kernel on rows of ~17 entries (shell-like factor)
    // -- except when a row's chain STARTS with the row solved last (U, upper && !transposed): there every entry behind
    // the first would be polled one round trip at a time (45 ms), and the slice kernel's batch re-read wins (17.9 ms)
    const bool packed = nrhs == 1 && plan.nslices > 0 && (long long)plan.nslices * 16 <= (long long)m;
    const int  sf     = (packed && ((long long)plan.nnz_tri <= 10LL * m || (upper && !tr && !conj))) ? 3 : 2;
    // chained rows (the dofs of a node) solved back to back by one lane: one hop per BLOCK level instead of per row level
    // ... and, where the plan-time model says it pays, the two-level schedule: chunks of consecutive blocks, hand-offs inside a
    // chunk through LDS (schedule 5; the reference chain only)
    const int sfb = plan.blk.valid ? (plan.blk.chunk.valid ? 5 : 4) : sf; // (trsm too: one grid column per right-hand side)
    // Round 3: the kid selects the ARITHMETIC, as it does in the reference (trsv.cpp:321-353), not the schedule.  kid 0 and auto:
    // the chain of ref_trsv_* -- every schedule reproduces it, so the fastest one runs; kid 1 / 2: the order of the 256-bit KT
    // kernels, kid 3: of the 512-bit ones (kt_trsv_l / kt_trsv_u, trsv_kt.cpp:64-150, :297-383), bit for bit, served by the
    // block kernel with run-time KT loops (triangles with chains), the per-level launches and the lane-per-position kernel.  The transposed KT kernels apply the same per-element fma
    // as the reference kernels (trsv_kt.cpp:183-268, :416-503), so for op != none every kid has the same bits.
    // aoclsparse_mi355_set_trsv_schedule forces a schedule (tests, measurements).
    const int kt_bits  = (!is_cplx && !tr && kid >= 1) ? (kid == 3 ? 512 : 256) : 0;
    const int forced   = Runtime::primary().trsv_schedule;
    const int schedule = is_cplx ? 1 : (forced >= 0 && forced <= 5) ? forced : (plan.nlevels <= 32 ? 0 : sfb);
    // the handle's own timeout word (pinned, device-mapped): allocated once per handle
    if(!is_cplx && !A->trsv_timeout_dev)
    {
        void *tw = nullptr, *twd = nullptr;
        if(hipHostMalloc(&tw, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&twd, tw, 0) == hipSuccess)
        {
            A->trsv_timeout_host  = static_cast<volatile unsigned int *>(tw);
            *A->trsv_timeout_host = 0;
            A->trsv_timeout_dev   = static_cast<unsigned int *>(twd);
        }
        else
        {
            (void)hipGetLastError(); // the solve then keeps its device-side word and a blocking read
            if(tw)
                (void)hipHostFree(tw);
        }
    }
    // a wait that expired in an EARLIER asynchronous (device-pointer) solve OF THIS HANDLE is reported now; callers that
    // never solve again ask aoclsparse_mi355_trsv_status(A) after their own stream synchronisation
    if(A->trsv_timeout_host && *A->trsv_timeout_host)
    {
        (void)hipStreamSynchronize(rt.stream());
        *A->trsv_timeout_host = 0;
        return aoclsparse_status_internal_error;
    }

    // (+ TRSV_XP_PAD elements: the block kernel parks the stores of lanes / rows that own nothing there)
    st = A->trsv_xp.alloc(sizeof(T) * ((size_t)m * (size_t)nrhs + TRSV_XP_PAD));
    if(st == aoclsparse_status_success)
        st = A->trsv_scratch.alloc(sizeof(unsigned int) * ((size_t)nrhs + 1 + (size_t)nrhs * (size_t)std::max<aoclsparse_int>(plan.blk.nlevels, 0)));
    if(st != aoclsparse_status_success)
        return st;

    // the workspaces are reused in stream order; after a change of stream (aoclsparse_mi355_set_stream) nothing orders this
    // solve behind the previous one, so everything enqueued so far completes first
    st = workspace_stream_guard(A, rt.stream());
    if(st != aoclsparse_status_success)
        return st;

    const bool bdev = rt.is_device_pointer(b), xdev = rt.is_device_pointer(x);
    const T   *db   = b;
    T         *dx   = x;
    void      *tmp  = nullptr;
    if(!bdev)
    {
        st = rt.staging(0, sizeof(T) * span_b, &tmp);
        if(st != aoclsparse_status_success)
            return st;
        MI355_HIP_TRY(hipMemcpyAsync(tmp, b, sizeof(T) * span_b, hipMemcpyHostToDevice, rt.stream()));
        db = static_cast<const T *>(tmp);
    }
    if(!xdev)
    {
        st = rt.staging(2, sizeof(T) * span_x, &tmp);
        if(st != aoclsparse_status_success)
            return st;
        if(x_partial) // strided / padded x: the untouched slots must survive the round trip
            MI355_HIP_TRY(hipMemcpyAsync(tmp, x, sizeof(T) * span_x, hipMemcpyHostToDevice, rt.stream()));
        dx = static_cast<T *>(tmp);
    }
    if constexpr(is_cplx)
        st = launch_ctrsv(rt.stream(), unit, conj, alpha, m, plan, A->dev_diag.as<T>(), db, dx, A->trsv_xp.as<T>(), nrhs,
                          b_off, incb, x_off, incx);
    else
        st = launch_trsv<T>(rt.stream(), schedule, unit, alpha, m, plan, A->dev_diag.as<T>(), db, dx,
                            A->trsv_xp.as<T>(), A->trsv_scratch.as<unsigned int>(), nrhs, b_off, incb, x_off, incx,
                            A->trsv_timeout_dev, kt_bits);
    if(st != aoclsparse_status_success)
        return st;
    if(!xdev)
        MI355_HIP_TRY(hipMemcpyAsync(x, dx, sizeof(T) * span_x, hipMemcpyDeviceToHost, rt.stream()));
    const bool syncfree = !is_cplx && (schedule >= 2 || (schedule == 1 && (nrhs != 1 || incb != 1 || incx != 1 || kt_bits != 0)));
    const bool pinned_word = A->trsv_timeout_dev != nullptr;
    if(!xdev || (syncfree && !pinned_word))
    {
        // host semantics (the result must be in the caller's memory on return); without the pinned word the
        // sync-free path also has to fetch its device-side timeout word
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        if(syncfree && !pinned_word)
        {
            unsigned int word = 0;
            MI355_HIP_TRY(hipMemcpy(&word, A->trsv_scratch.as<unsigned int>() + nrhs, sizeof(word),
                                    hipMemcpyDeviceToHost));
            if(word)
                return aoclsparse_status_internal_error;
        }
    }
    // device-pointer solves stay asynchronous: an expired wait (never expected) is seen at the next solve; a
    // host-pointer solve has just synchronised and reports it now
    if(syncfree && pinned_word && !xdev && *A->trsv_timeout_host)
    {
        *A->trsv_timeout_host = 0;
        return aoclsparse_status_internal_error;
    }
    return aoclsparse_status_success;
}

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
    if(!A->opt_csr_full_diag && descr->diag_type != aoclsparse_diag_type_unit)
        return aoclsparse_status_invalid_value;
    // KAT of trsv.cpp:315-376 has kernels 0..3 per doid
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    // (m-1)*inc must not overflow, trsv.cpp:407-411
    if((long long)(m - 1) * incb > 2147483647LL || (long long)(m - 1) * incx > 2147483647LL)
        return aoclsparse_status_invalid_size;
    return solve_core<T>(trans, alpha, A, descr, kid, b, x, 1, 0, incb, (size_t)(m - 1) * incb + 1, 0, incx,
                         (size_t)(m - 1) * incx + 1, incx != 1);
}

// level3/aoclsparse_trsm.hpp:40-160: X = alpha * inv(op(A)) * B column by column.  The reference loops
// aoclsparse::trsv over the n columns (OpenMP over columns); here all columns run in ONE launch (the
// level structure is shared, blockIdx.y is the column), each bit-identical to the single-RHS solve.
template <typename T>
aoclsparse_status trsm_t(aoclsparse_operation trans, const T alpha, aoclsparse_matrix A,
                         const aoclsparse_mat_descr descr, aoclsparse_order order, const T *B, aoclsparse_int n,
                         aoclsparse_int ldb, T *X, aoclsparse_int ldx, aoclsparse_int kid,
                         aoclsparse_matrix_data_type vt)
{
    if(!A || !X || !B || !descr)
        return aoclsparse_status_invalid_pointer;
    if(!A->user.ptr)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    const aoclsparse_int m = A->m;
    if(m < 0 || A->nnz < 0 || n < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0 || A->n == 0 || A->nnz == 0 || n == 0)
        return aoclsparse_status_success;
    if(m != A->n)
        return aoclsparse_status_invalid_size;
    if(ldb < 0 || ldx < 0)
        return aoclsparse_status_invalid_size;
    if(descr->base != aoclsparse_index_base_zero && descr->base != aoclsparse_index_base_one)
        return aoclsparse_status_invalid_value;
    if(trans != aoclsparse_operation_none && trans != aoclsparse_operation_transpose
       && trans != aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_symmetric && descr->type != aoclsparse_matrix_type_triangular)
        return aoclsparse_status_invalid_value;
    if(descr->fill_mode != aoclsparse_fill_mode_lower && descr->fill_mode != aoclsparse_fill_mode_upper)
        return aoclsparse_status_not_implemented;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    aoclsparse_status st = csr_optimize(A);
    if(st != aoclsparse_status_success)
        return st;
    aoclsparse_int incb, incx;
    long long      b_off, x_off;
    if(order == aoclsparse_order_row)
        incb = ldb, incx = ldx, b_off = 1, x_off = 1;
    else if(order == aoclsparse_order_column)
        incb = 1, incx = 1, b_off = ldb, x_off = ldx;
    else
        return aoclsparse_status_invalid_value;
    if((long long)n * b_off > 2147483647LL || (long long)n * x_off > 2147483647LL)
        return aoclsparse_status_invalid_size;
    // what every per-column trsv of the reference checks (trsv.cpp:72-137)
    if(incb <= 0 || incx <= 0 || descr->diag_type == aoclsparse_diag_type_zero)
        return aoclsparse_status_invalid_value;
    if(!A->opt_csr_full_diag && descr->diag_type != aoclsparse_diag_type_unit)
        return aoclsparse_status_invalid_value;
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    const size_t span_b = (size_t)(m - 1) * incb + (size_t)(n - 1) * b_off + 1;
    const size_t span_x = (size_t)(m - 1) * incx + (size_t)(n - 1) * x_off + 1;
    if(span_b > 2147483647ULL || span_x > 2147483647ULL)
        return aoclsparse_status_invalid_size;
    const bool dense_x = order == aoclsparse_order_row ? ldx == n : ldx == m;
    return solve_core<T>(trans, alpha, A, descr, kid, B, X, n, b_off, incb, span_b, x_off, incx, span_x, !dense_x);
}

// aoclsparse_?csrsv (level2/aoclsparse_csrsv.hpp:28-190): the older raw-array solve, y = inv(T) * alpha * x with T the
// lower / upper triangle of the CSR arrays chosen by descr->fill_mode.  Its row loop is the chain of ref_trsv_l / _u
// (alpha*x_i, subtract in storage order, divide by the diagonal), so it is served by the TRSV path on a one-shot
// handle over the caller's arrays (kid 0 order: bit-identical for sorted rows with a full diagonal).
// Deviations, both on input the reference mishandles: rows are sorted by the clean-CSR step instead of being cut at
// the first upper entry, and a missing diagonal of a non-unit solve is invalid_value instead of a division by a
// stale entry.  Host arrays only (the analysis walks them).
template <typename T>
aoclsparse_status csrsv_t(aoclsparse_operation trans, const T *alpha, aoclsparse_int m, const T *csr_val,
                          const aoclsparse_int *csr_col_ind, const aoclsparse_int *csr_row_ptr,
                          const aoclsparse_mat_descr descr, const T *x, T *y)
{
    if(!csr_val || !csr_row_ptr || !csr_col_ind || !x || !y || !descr || !alpha)
        return aoclsparse_status_invalid_pointer;
    if(descr->base != aoclsparse_index_base_zero)
        return aoclsparse_status_not_implemented;
    if(descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_symmetric)
        return aoclsparse_status_not_implemented;
    if(trans != aoclsparse_operation_none)
        return aoclsparse_status_not_implemented;
    if(m < 0)
        return aoclsparse_status_invalid_size;
    if(m == 0)
        return aoclsparse_status_success;
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    if(rt.is_device_pointer(csr_row_ptr) || rt.is_device_pointer(csr_col_ind) || rt.is_device_pointer(csr_val))
        return aoclsparse_status_not_implemented;
    aoclsparse_matrix A  = nullptr;
    aoclsparse_status st = std::is_same<T, float>::value
                               ? aoclsparse_create_scsr(&A, descr->base, m, m, csr_row_ptr[m] - descr->base,
                                                        const_cast<aoclsparse_int *>(csr_row_ptr),
                                                        const_cast<aoclsparse_int *>(csr_col_ind),
                                                        reinterpret_cast<float *>(const_cast<T *>(csr_val)))
                               : aoclsparse_create_dcsr(&A, descr->base, m, m, csr_row_ptr[m] - descr->base,
                                                        const_cast<aoclsparse_int *>(csr_row_ptr),
                                                        const_cast<aoclsparse_int *>(csr_col_ind),
                                                        reinterpret_cast<double *>(const_cast<T *>(csr_val)));
    if(st != aoclsparse_status_success)
        return st;
    _aoclsparse_mat_descr tri = *descr;
    tri.type                  = aoclsparse_matrix_type_triangular;
    if(tri.fill_mode != aoclsparse_fill_mode_lower)
        tri.fill_mode = aoclsparse_fill_mode_upper; // csrsv.hpp:77-86: anything but lower runs the upper solve
    st = trsv_t<T>(aoclsparse_operation_none, *alpha, A, &tri, x, 1, y, 1, 0,
                   std::is_same<T, float>::value ? aoclsparse_smat : aoclsparse_dmat);
    aoclsparse_destroy(&A);
    return st;
}

} // namespace

extern "C" {

aoclsparse_status aoclsparse_dcsrsv(aoclsparse_operation trans, const double *alpha, aoclsparse_int m,
                                    const double *csr_val, const aoclsparse_int *csr_col_ind,
                                    const aoclsparse_int *csr_row_ptr, const aoclsparse_mat_descr descr, const double *x,
                                    double *y)
{
    return csrsv_t<double>(trans, alpha, m, csr_val, csr_col_ind, csr_row_ptr, descr, x, y);
}
aoclsparse_status aoclsparse_scsrsv(aoclsparse_operation trans, const float *alpha, aoclsparse_int m,
                                    const float *csr_val, const aoclsparse_int *csr_col_ind,
                                    const aoclsparse_int *csr_row_ptr, const aoclsparse_mat_descr descr, const float *x,
                                    float *y)
{
    return csrsv_t<float>(trans, alpha, m, csr_val, csr_col_ind, csr_row_ptr, descr, x, y);
}

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

aoclsparse_status aoclsparse_dtrsm(const aoclsparse_operation trans, const double alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, aoclsparse_order order, const double *B,
                                   aoclsparse_int n, aoclsparse_int ldb, double *X, aoclsparse_int ldx)
{
    return trsm_t<double>(trans, alpha, A, descr, order, B, n, ldb, X, ldx, -1, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_strsm(const aoclsparse_operation trans, const float alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, aoclsparse_order order, const float *B,
                                   aoclsparse_int n, aoclsparse_int ldb, float *X, aoclsparse_int ldx)
{
    return trsm_t<float>(trans, alpha, A, descr, order, B, n, ldb, X, ldx, -1, aoclsparse_smat);
}

aoclsparse_status aoclsparse_dtrsm_kid(const aoclsparse_operation trans, const double alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, aoclsparse_order order, const double *B,
                                       aoclsparse_int n, aoclsparse_int ldb, double *X, aoclsparse_int ldx,
                                       const aoclsparse_int kid)
{
    return trsm_t<double>(trans, alpha, A, descr, order, B, n, ldb, X, ldx, kid, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_strsm_kid(const aoclsparse_operation trans, const float alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, aoclsparse_order order, const float *B,
                                       aoclsparse_int n, aoclsparse_int ldb, float *X, aoclsparse_int ldx,
                                       const aoclsparse_int kid)
{
    return trsm_t<float>(trans, alpha, A, descr, order, B, n, ldb, X, ldx, kid, aoclsparse_smat);
}

aoclsparse_status aoclsparse_ctrsv(aoclsparse_operation trans, const aoclsparse_float_complex alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, const aoclsparse_float_complex *b, aoclsparse_float_complex *x)
{
    return trsv_t<cfloat>(trans, cfloat(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cfloat *>(b), 1,
                         reinterpret_cast<cfloat *>(x), 1, -1, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_ctrsv_kid(aoclsparse_operation trans, const aoclsparse_float_complex alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const aoclsparse_float_complex *b, aoclsparse_float_complex *x, aoclsparse_int kid)
{
    return trsv_t<cfloat>(trans, cfloat(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cfloat *>(b), 1,
                         reinterpret_cast<cfloat *>(x), 1, kid, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_ctrsv_strided(aoclsparse_operation trans, const aoclsparse_float_complex alpha, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const aoclsparse_float_complex *b, const aoclsparse_int incb,
                                           aoclsparse_float_complex *x, const aoclsparse_int incx)
{
    return trsv_t<cfloat>(trans, cfloat(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cfloat *>(b), incb,
                         reinterpret_cast<cfloat *>(x), incx, -1, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_ctrsm(const aoclsparse_operation trans, const aoclsparse_float_complex alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, aoclsparse_order order, const aoclsparse_float_complex *B,
                                   aoclsparse_int n, aoclsparse_int ldb, aoclsparse_float_complex *X, aoclsparse_int ldx)
{
    return trsm_t<cfloat>(trans, cfloat(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cfloat *>(B), n, ldb,
                         reinterpret_cast<cfloat *>(X), ldx, -1, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_ctrsm_kid(const aoclsparse_operation trans, const aoclsparse_float_complex alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, aoclsparse_order order, const aoclsparse_float_complex *B,
                                       aoclsparse_int n, aoclsparse_int ldb, aoclsparse_float_complex *X, aoclsparse_int ldx,
                                       const aoclsparse_int kid)
{
    return trsm_t<cfloat>(trans, cfloat(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cfloat *>(B), n, ldb,
                         reinterpret_cast<cfloat *>(X), ldx, kid, aoclsparse_cmat);
}

aoclsparse_status aoclsparse_ztrsv(aoclsparse_operation trans, const aoclsparse_double_complex alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, const aoclsparse_double_complex *b, aoclsparse_double_complex *x)
{
    return trsv_t<cdouble>(trans, cdouble(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cdouble *>(b), 1,
                         reinterpret_cast<cdouble *>(x), 1, -1, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_ztrsv_kid(aoclsparse_operation trans, const aoclsparse_double_complex alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, const aoclsparse_double_complex *b, aoclsparse_double_complex *x, aoclsparse_int kid)
{
    return trsv_t<cdouble>(trans, cdouble(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cdouble *>(b), 1,
                         reinterpret_cast<cdouble *>(x), 1, kid, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_ztrsv_strided(aoclsparse_operation trans, const aoclsparse_double_complex alpha, aoclsparse_matrix A,
                                           const aoclsparse_mat_descr descr, const aoclsparse_double_complex *b, const aoclsparse_int incb,
                                           aoclsparse_double_complex *x, const aoclsparse_int incx)
{
    return trsv_t<cdouble>(trans, cdouble(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cdouble *>(b), incb,
                         reinterpret_cast<cdouble *>(x), incx, -1, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_ztrsm(const aoclsparse_operation trans, const aoclsparse_double_complex alpha, aoclsparse_matrix A,
                                   const aoclsparse_mat_descr descr, aoclsparse_order order, const aoclsparse_double_complex *B,
                                   aoclsparse_int n, aoclsparse_int ldb, aoclsparse_double_complex *X, aoclsparse_int ldx)
{
    return trsm_t<cdouble>(trans, cdouble(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cdouble *>(B), n, ldb,
                         reinterpret_cast<cdouble *>(X), ldx, -1, aoclsparse_zmat);
}
aoclsparse_status aoclsparse_ztrsm_kid(const aoclsparse_operation trans, const aoclsparse_double_complex alpha, aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, aoclsparse_order order, const aoclsparse_double_complex *B,
                                       aoclsparse_int n, aoclsparse_int ldb, aoclsparse_double_complex *X, aoclsparse_int ldx,
                                       const aoclsparse_int kid)
{
    return trsm_t<cdouble>(trans, cdouble(alpha.real, alpha.imag), A, descr, order, reinterpret_cast<const cdouble *>(B), n, ldb,
                         reinterpret_cast<cdouble *>(X), ldx, kid, aoclsparse_zmat);
}

aoclsparse_status aoclsparse_mi355_strsv_full(aoclsparse_operation trans, float alpha, aoclsparse_matrix A,
                                              const aoclsparse_mat_descr descr, const float *b, aoclsparse_int incb,
                                              float *x, aoclsparse_int incx, aoclsparse_int kid)
{
    return trsv_t<float>(trans, alpha, A, descr, b, incb, x, incx, kid, aoclsparse_smat);
}
aoclsparse_status aoclsparse_mi355_dtrsv_full(aoclsparse_operation trans, double alpha, aoclsparse_matrix A,
                                              const aoclsparse_mat_descr descr, const double *b, aoclsparse_int incb,
                                              double *x, aoclsparse_int incx, aoclsparse_int kid)
{
    return trsv_t<double>(trans, alpha, A, descr, b, incb, x, incx, kid, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_mi355_ctrsv_full(aoclsparse_operation trans, aoclsparse_float_complex alpha,
                                              aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                              const aoclsparse_float_complex *b, aoclsparse_int incb,
                                              aoclsparse_float_complex *x, aoclsparse_int incx, aoclsparse_int kid)
{
    return trsv_t<cfloat>(trans, cfloat(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cfloat *>(b), incb,
                          reinterpret_cast<cfloat *>(x), incx, kid, aoclsparse_cmat);
}
aoclsparse_status aoclsparse_mi355_ztrsv_full(aoclsparse_operation trans, aoclsparse_double_complex alpha,
                                              aoclsparse_matrix A, const aoclsparse_mat_descr descr,
                                              const aoclsparse_double_complex *b, aoclsparse_int incb,
                                              aoclsparse_double_complex *x, aoclsparse_int incx, aoclsparse_int kid)
{
    return trsv_t<cdouble>(trans, cdouble(alpha.real, alpha.imag), A, descr, reinterpret_cast<const cdouble *>(b), incb,
                           reinterpret_cast<cdouble *>(x), incx, kid, aoclsparse_zmat);
}

aoclsparse_status aoclsparse_mi355_set_trsv_schedule(aoclsparse_int schedule)
{
    if(schedule < -1 || schedule > 5)
        return aoclsparse_status_invalid_value;
    Runtime::primary().trsv_schedule = (int)schedule;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_trsv_status(aoclsparse_matrix A)
{
    if(!A)
        return aoclsparse_status_invalid_pointer;
    if(A->trsv_timeout_host && *A->trsv_timeout_host)
    {
        *A->trsv_timeout_host = 0;
        return aoclsparse_status_internal_error;
    }
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_get_trsv_levels(const aoclsparse_matrix A, aoclsparse_fill_mode fill,
                                                   aoclsparse_operation op, aoclsparse_int *levels)
{
    if(!A || !levels)
        return aoclsparse_status_invalid_pointer;
    std::shared_lock<std::shared_mutex> r(A->guard);
    const bool      up = fill == aoclsparse_fill_mode_upper;
    const bool      cj = op == aoclsparse_operation_conjugate_transpose && is_complex_type(A->val_type);
    const TrsvPlan &p  = A->trsv_plan[cj ? 4 + (up ? 1 : 0) : (up ? 2 : 0) + (op != aoclsparse_operation_none ? 1 : 0)];
    *levels = p.valid ? p.nlevels : -1;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_get_trsv_info(const aoclsparse_matrix A, aoclsparse_fill_mode fill, aoclsparse_operation op,
                                                 aoclsparse_mi355_trsv_info *info)
{
    if(!A || !info)
        return aoclsparse_status_invalid_pointer;
    std::shared_lock<std::shared_mutex> r(A->guard);
    const bool      up = fill == aoclsparse_fill_mode_upper;
    const bool      cj = op == aoclsparse_operation_conjugate_transpose && is_complex_type(A->val_type);
    const TrsvPlan &p  = A->trsv_plan[cj ? 4 + (up ? 1 : 0) : (up ? 2 : 0) + (op != aoclsparse_operation_none ? 1 : 0)];
    *info              = aoclsparse_mi355_trsv_info{};
    if(!p.valid)
        return aoclsparse_status_success;
    info->levels = p.nlevels;
    if(p.blk.valid)
        info->blocks = p.blk.nblocks, info->block_levels = p.blk.nlevels, info->slices = p.blk.nslices,
        info->slice_fan_in_permille = (aoclsparse_int)(p.blk.slice_fan_in * 1000.0 + 0.5);
    const TrsvChunkPlan &c = p.blk.chunk;
    info->model_chunk_us = (aoclsparse_int)c.model_us, info->model_block_us = (aoclsparse_int)c.model_block_us;
    if(p.blk.valid && c.valid)
        info->chunks = c.nchunks, info->steps = c.nsteps, info->lds_slots = c.max_rows;
    const int forced = Runtime::primary().trsv_schedule;
    const int autos  = p.nlevels <= 32 ? 0 : (p.blk.valid ? (c.valid ? 5 : 4) : 2);
    int       sched  = is_complex_type(A->val_type) ? 1 : (forced >= 0 && forced <= 5 ? forced : autos);
    if(sched == 5 && !(p.blk.valid && c.valid))
        sched = 4;
    if(sched == 4 && !p.blk.valid)
        sched = 3;
    info->schedule = sched;
    return aoclsparse_status_success;
}

} // extern "C"
