// csrmm_api.cpp -- aoclsparse_?csrmm(_kid): checks of level3/aoclsparse_csrmm.hpp:448-618, then the
// HIP kernels of csrmm_kernels.hip on the device-resident CSR (A^T copy for op != none).
#include "internal.hpp"
#include <climits>
#include <cmath>

#include <system_error>
#include <condition_variable>
#include <functional>
#include <thread>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

using namespace mi355;

namespace
{

constexpr int CM_DETOUR_NNZ_PER_ROW = 8; // = the column kernel's register cache (CM_K)

// Host <-> staging copy of a strided dense operand: ONLY the `inner` elements of each of the `outer` lines move; the ld
// padding -- which for a column shard of a row-major matrix is the other shards' columns -- is neither sent nor, on the way
// back, written (round 3: the contiguous-span copy of round 2 rewrote it with the values it had read, which is a lost update
// as soon as another thread -- another device's worker of ?csrmm_multi -- owns those columns, and sent 8 x the bytes a
// 1/8 shard needs).
template <typename T>
aoclsparse_status copy_dense(Runtime &rt, void *dst, const void *src, aoclsparse_int outer, aoclsparse_int inner,
                             aoclsparse_int ld, hipMemcpyKind kind)
{
    if(outer <= 0 || inner <= 0)
        return aoclsparse_status_success;
    if(ld == inner || outer == 1)
        MI355_HIP_TRY(hipMemcpyAsync(dst, src, sizeof(T) * ((size_t)(outer - 1) * (size_t)ld + (size_t)inner), kind, rt.stream()));
    else
        MI355_HIP_TRY(hipMemcpy2DAsync(dst, sizeof(T) * (size_t)ld, src, sizeof(T) * (size_t)ld, sizeof(T) * (size_t)inner,
                                       (size_t)outer, kind, rt.stream()));
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status stage_dense(Runtime &rt, int slot, const T *host, aoclsparse_int outer, aoclsparse_int inner,
                              aoclsparse_int ld, bool copy, void **dev, size_t *bytes)
{
    // exact extent of a strided dense matrix: the last row / column carries no ld padding (BLAS convention), and a
    // column shard of a row-major matrix starts j0 elements into its first row
    *bytes               = outer > 0 ? sizeof(T) * ((size_t)(outer - 1) * (size_t)ld + (size_t)inner) : 0;
    aoclsparse_status st = rt.staging(slot, *bytes, dev);
    if(st != aoclsparse_status_success)
        return st;
    if(copy && *bytes)
        return copy_dense<T>(rt, *dev, host, outer, inner, ld, hipMemcpyHostToDevice);
    return aoclsparse_status_success;
}

// Row groups for the row-major kernel: maximal runs (<= CSRMM_GROUP) of consecutive rows whose column arrays are
// equal entry for entry.  Kept only when they pay: at least 1.5 rows per group on average.
aoclsparse_status build_mm_groups(const HostCsr &h, SpmvPlan &plan)
{
    MmGroups &g = plan.mm;
    if(g.valid || g.tried)
        return aoclsparse_status_success;
    g.tried = true;
    if(h.m < 2)
        return aoclsparse_status_success;
    std::vector<aoclsparse_int> first;
    int                         max_rows = 1;
    try
    {
        first.reserve((size_t)h.m / 2 + 2);
        aoclsparse_int i = 0;
        while(i < h.m)
        {
            first.push_back(i);
            const aoclsparse_int s = h.ptr[i] - h.base, len = h.ptr[i + 1] - h.base - s;
            aoclsparse_int       e = i + 1;
            while(e < h.m && e - i < CSRMM_GROUP && len > 0 && h.ptr[e + 1] - h.ptr[e] == len
                  && !memcmp(h.ind + s, h.ind + (h.ptr[e] - h.base), sizeof(aoclsparse_int) * (size_t)len))
                e++;
            max_rows = std::max(max_rows, (int)(e - i));
            i = e;
        }
        first.push_back(h.m);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    const aoclsparse_int ng = (aoclsparse_int)first.size() - 1;
    if((long long)h.m * 2 < (long long)ng * 3)
        return aoclsparse_status_success;
    aoclsparse_status st = g.first.upload(first.data(), sizeof(aoclsparse_int) * first.size(), Runtime::get().stream());
    if(st != aoclsparse_status_success)
        return st;
    g.ngroups = ng, g.max_rows = max_rows;
    g.valid = true;
    // the band of the groups: how far right of its first row a group's list reaches, when most groups agree (majority vote, then a
    // count -- as detect_row_runs does for single rows): what the launcher deals to the XCDs by
    {
        auto reach = [&](aoclsparse_int gi) {
            const aoclsparse_int i = first[(size_t)gi], e = h.ptr[i + 1] - h.base;
            return e > h.ptr[i] - h.base ? h.ind[e - 1] - h.base - i : 0;
        };
        aoclsparse_int cand = 0;
        long long      votes = 0, same = 0;
        for(aoclsparse_int gi = 0; gi < ng; gi++)
        {
            const aoclsparse_int d = reach(gi);
            if(votes == 0)
                cand = d, votes = 1;
            else
                votes += d == cand ? 1 : -1;
        }
        for(aoclsparse_int gi = 0; gi < ng; gi++)
            same += reach(gi) == cand;
        g.band = same * 2 >= (long long)ng && cand >= 256 && (long long)cand * 4 <= (long long)h.m ? cand : 0;
    }
    return aoclsparse_status_success;
}

// Column-major pairs: consecutive rows (i, i+1) of equal length <= the register cache where row i+1's columns are row
// i's + 1, chosen greedily front to back; every other row is a "single".  The pair kernel is used when at least 80 %
// of the rows found a partner (singles go through the generic kernel with a row list).  One pass over the host CSR,
// once per handle.
// Stencil-like matrices (row-major csrmm, n >= 128): when most rows repeat the list of the row before shifted by one and have
// <= 8 entries, the row-run kernel pays off (entry k of a row reuses the B row that entry k + 1 of the previous row loaded).
aoclsparse_status detect_row_runs(const HostCsr &h, SpmvPlan &plan)
{
    MmGroups &g = plan.mm;
    if(g.runs_tried)
        return aoclsparse_status_success;
    g.runs_tried = true;
    if(h.m < 64)
        return aoclsparse_status_success;
    long long      followers = 0;
    aoclsparse_int longest = 0;
    for(aoclsparse_int i = 1; i < h.m; i++)
    {
        const aoclsparse_int s = h.ptr[i] - h.base, len = h.ptr[i + 1] - h.base - s, sp = h.ptr[i - 1] - h.base;
        longest                = std::max(longest, len);
        if(len == 0 || len > 8 || s - sp != len)
            continue;
        bool ok = true;
        for(aoclsparse_int k = 0; k < len && ok; k++)
            ok = h.ind[s + k] == h.ind[sp + k] + 1;
        followers += ok;
    }
    g.row_runs = followers * 2 >= (long long)h.m && longest <= 64;
    g.band     = 0;
    if(!g.row_runs)
        return aoclsparse_status_success;
    // the band: the distance from the diagonal to a row's last entry that most rows share (majority vote, then a count)
    aoclsparse_int cand = 0;
    long long      votes = 0;
    auto           reach = [&](aoclsparse_int i) {
        const aoclsparse_int e = h.ptr[i + 1] - h.base;
        return e > h.ptr[i] - h.base ? h.ind[e - 1] - h.base - i : 0;
    };
    for(aoclsparse_int i = 0; i < h.m; i++)
    {
        const aoclsparse_int d = reach(i);
        if(votes == 0)
            cand = d, votes = 1;
        else
            votes += d == cand ? 1 : -1;
    }
    long long same = 0;
    for(aoclsparse_int i = 0; i < h.m; i++)
        same += reach(i) == cand;
    if(same * 2 < (long long)h.m || cand < 256 || (long long)cand * 4 > (long long)h.m)
        return aoclsparse_status_success;
    constexpr aoclsparse_int RUN = 8, STRIP_MAX = 256;
    const aoclsparse_int     band = cand;
    const aoclsparse_int     sw   = std::min<aoclsparse_int>(((band + 7) / 8 + RUN - 1) / RUN * RUN, STRIP_MAX);
    const aoclsparse_int     qg   = 1; // band lines per workgroup-sized group of blocks
    const aoclsparse_int     nb   = (h.m + RUN - 1) / RUN;
    try
    {
        // blocks sorted by (strip, group of qg band lines, position inside the strip, line inside the group): the row
        // order inside a strip when qg = 1
        std::vector<std::pair<unsigned long long, aoclsparse_int>> keyed((size_t)nb);
        const unsigned long long nq = (unsigned long long)(h.m / band + 1), nsb = (unsigned long long)(sw / RUN + 1);
        for(aoclsparse_int b = 0; b < nb; b++)
        {
            const aoclsparse_int i0 = b * RUN, q = i0 / band, sp = i0 % band;
            const unsigned long long key
                = ((((unsigned long long)(sp / sw) * nq + (unsigned long long)(q / qg)) * nsb + (unsigned long long)((sp % sw) / RUN))
                   * (unsigned long long)qg)
                  + (unsigned long long)(q % qg);
            keyed[(size_t)b] = {key, i0};
        }
        std::sort(keyed.begin(), keyed.end());
        std::vector<aoclsparse_int> order((size_t)nb);
        for(aoclsparse_int b = 0; b < nb; b++)
            order[(size_t)b] = keyed[(size_t)b].second;
        const aoclsparse_status st
            = g.run_order.upload(order.data(), sizeof(aoclsparse_int) * order.size(), Runtime::get().stream());
        if(st != aoclsparse_status_success)
            return st;
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    g.band = band;
    // row blocks along the lines of the band for the narrow kernel: 8 per line, as long as each fits the kernel's LDS tile
    g.slab_nblocks = 0;
    try
    {
        const aoclsparse_int tile = plan.tile & ~1;
        std::vector<aoclsparse_int> blk;
        bool                        fits = tile >= 512;
        for(aoclsparse_int q0 = 0; q0 < h.m && fits; q0 += band)
            for(int k = 0; k < 8 && fits; k++)
            {
                const aoclsparse_int ra = q0 + (aoclsparse_int)((long long)band * k / 8), rb = q0 + (aoclsparse_int)((long long)band * (k + 1) / 8);
                if(ra >= h.m)
                    break;
                const aoclsparse_int re = std::min(rb, h.m);
                fits = h.ptr[re] - h.ptr[ra] <= tile && re - ra <= spmv_maxrows(tile);
                blk.push_back(ra), blk.push_back(h.ptr[ra] - h.base);
            }
        if(fits && !blk.empty())
        {
            const aoclsparse_int nb = (aoclsparse_int)(blk.size() / 2);
            blk.push_back(h.m), blk.push_back(h.ptr[h.m] - h.base);
            if(g.slab_blocks.upload(blk.data(), sizeof(aoclsparse_int) * blk.size(), Runtime::get().stream()) == aoclsparse_status_success)
                g.slab_nblocks = nb;
        }
    }
    catch(const std::bad_alloc &)
    {
        g.slab_nblocks = 0; // (optional: the SpMV plan's row blocks serve the kernel)
    }
    return aoclsparse_status_success;
}

// Blocked-ELL copy for the MFMA kernel (csrmm_bell_kernels.hip).  The reference chooses its blocked CSR the same way: count
// the non-empty blocks a block size would give and keep the format when enough of their cells are real entries
// (conversion/aoclsparse_convert.cpp:36-147 aoclsparse_opt_blksize, analysis.cpp:146-160: 40-50 % thresholds).  Here: 16 x 16
// blocks (the MFMA tile), kept when fill = nnz / (256 * blocks) >= 0.5, the ELL padding (width * block rows over blocks) stays
// under 1.35 and every row is sorted and duplicate-free (the tile walks k upwards: only then is the sum the CSR-order chain).
// Which XCD works through which block rows, and in which order (BellPlan::order).  A block row's wavefronts fetch `width` stretches of B
// (16 rows each); the same stretch is wanted by every block row that stores that block column, and it crosses the fabric once per L2 that
// does not hold it any more.  Model: workgroup w runs on XCD w % 8, an XCD works through its list in order, and a stretch is still in that
// XCD's L2 when the XCD touched it at most BELL_L2_WINDOW list positions ago (about what an XCD has in flight: 32 CUs x 3 workgroups;
// beyond that the 0.3 MB every block row moves have pushed it out).  Calibration, 7-point stand-ins of 32^3 / 40^3 nodes, FETCH_SIZE per
// distinct stretch against the model (profiles/r6/bell_experiments.txt): launch order 5.2 vs 4.9, chunks of 4 / 5 block rows 3.5 vs 3.4.
// Candidates:
//  * chunks of c consecutive block rows dealt to the XCDs in turn (c = 1 is launch order).  The SMALLEST chunk within 20 % of the fewest
//    misses is kept: small chunks keep all eight XCDs inside one neighbourhood of B, C and the values, which is worth more than the last
//    misses (32^3: chunks of 4 / 64 model 3.4 / 2.9 fetches and take 1.00 / 1.02 ms; a contiguous eighth per XCD has the fewest of the
//    chunked orders and measures 5-12 % slower than chunks of 4: eight distant DRAM streams).
//  * when the block columns sit at constant offsets 1, n1, n1 n2 from the diagonal (a structured grid numbered line by line): every XCD
//    takes regions of a x b block rows of the cross-section and follows each region through all the planes, so that the neighbours in the
//    next plane are a x b positions away instead of a plane's worth.  Taken when the model counts under 0.8 of the best chunk's misses.
// O(stored blocks) per candidate, once per handle.  AOCLSPARSE_MI355_BELL_XCD_CHUNK (diagnostic, read at analysis time): c >= 1 forces a
// chunk, 0 launch order without a list, -1 the lattice sweep whenever a lattice is found.
constexpr int BELL_L2_WINDOW = 96;
struct BellLattice
{
    aoclsparse_int n1 = 0, n2 = 0, n3 = 0; // block rows per line, lines per plane (1: a 2-D grid), planes
};
// the offsets block column - block row that at least a quarter of the block rows store
BellLattice detect_bell_lattice(const aoclsparse_int *bcol, aoclsparse_int nbr, aoclsparse_int width)
{
    BellLattice         L;
    std::vector<int>    hist((size_t)nbr, 0);
    for(aoclsparse_int b = 0; b < nbr; b++)
        for(aoclsparse_int s = 0; s < width; s++)
        {
            const aoclsparse_int bc = bcol[(size_t)b * width + s];
            if(bc < 0)
                break;
            if(bc > b && bc - b < nbr)
                hist[(size_t)(bc - b)]++;
        }
    auto in = [&](long long o) { return o >= 1 && o < nbr && hist[(size_t)o] >= nbr / 4; };
    if(!in(1))
        return L;
    // a line length / plane size is the CENTRE of a cluster of offsets (7-point: n1 alone; 27-point: n1 - 1, n1, n1 + 1)
    auto centre = [&](long long o) { return in(o) && in(o - 1) == in(o + 1); };
    long long n1 = 0, pl = 0;
    for(long long o = 2; o < nbr && !n1; o++)
        if(centre(o))
            n1 = o;
    if(!n1 || n1 > nbr / 4)
        return L;
    // (the plane size is also the centre of its clusters ALONG the lines: 27-point stores n1 n2 - n1, n1 n2, n1 n2 + n1)
    for(long long o = 2 * n1; o < nbr && !pl; o += n1)
        if(centre(o) && o > n1 + 1 && in(o - n1) == in(o + n1))
            pl = o;
    L.n1 = (aoclsparse_int)n1;
    if(pl && pl <= nbr / 2)
        L.n2 = (aoclsparse_int)(pl / n1), L.n3 = (aoclsparse_int)((nbr + pl - 1) / pl);
    else
        L.n2 = 1, L.n3 = (aoclsparse_int)((nbr + n1 - 1) / n1);
    return L;
}
// the lists of the eight XCDs: the line in nsx pieces, b lines of them = a region of the cross-section, followed through the planes of a
// z segment (nseg of them); the regions of a segment go to the XCDs in turn (eight neighbours at a time work side by side)
void bell_lattice_lists(const BellLattice &L, aoclsparse_int nbr, int nsx, int b, int nseg, std::vector<aoclsparse_int> (&list)[8])
{
    for(auto &l : list)
        l.clear();
    long long       r  = 0;
    const long long pl = (long long)L.n1 * L.n2;
    for(int sg = 0; sg < nseg; sg++)
    {
        const aoclsparse_int za = (aoclsparse_int)((long long)L.n3 * sg / nseg), zb = (aoclsparse_int)((long long)L.n3 * (sg + 1) / nseg);
        for(aoclsparse_int y0 = 0; y0 < L.n2; y0 += b)
            for(int sx = 0; sx < nsx; sx++, r++)
            {
                const aoclsparse_int xa = (aoclsparse_int)((long long)L.n1 * sx / nsx), xb = (aoclsparse_int)((long long)L.n1 * (sx + 1) / nsx);
                auto                &l = list[r & 7];
                for(aoclsparse_int z = za; z < zb; z++)
                    for(aoclsparse_int y = y0; y < std::min<aoclsparse_int>(L.n2, y0 + b); y++)
                        for(aoclsparse_int x = xa; x < xb; x++)
                        {
                            const long long br = x + (long long)L.n1 * y + pl * z;
                            if(br < nbr)
                                l.push_back((aoclsparse_int)br);
                        }
            }
    }
}
void bell_chunk_lists(aoclsparse_int nbr, int ch, std::vector<aoclsparse_int> (&list)[8])
{
    for(auto &l : list)
        l.clear();
    for(aoclsparse_int b = 0; b < nbr; b++)
        list[(b / ch) & 7].push_back(b);
}
constexpr int BELL_ORDER_AUTO = INT_MIN;
int bell_order_forced()
{
    const char *e = std::getenv("AOCLSPARSE_MI355_BELL_XCD_CHUNK");
    return e ? std::atoi(e) : BELL_ORDER_AUTO;
}
void choose_bell_order(const aoclsparse_int *bcol, size_t bcol_size, aoclsparse_int nbr, aoclsparse_int width, aoclsparse_int nbc, BellPlan &bp,
                       std::vector<aoclsparse_int> &order, int forced)
{
    bp.xcd_chunk = 1, bp.order_len = 0, bp.model_fetches = bp.model_fetches_launch_order = 0.0;
    bp.lattice[0] = bp.lattice[1] = bp.lattice[2] = 0, bp.region[0] = bp.region[1] = 0;
    order.clear();
    static const int cand[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32, 40, 48, 64};
    if(nbr < 8 * 64 || nbc <= 0)
        return;
    std::vector<long long>      last((size_t)nbc);
    std::vector<aoclsparse_int> list[8];
    auto misses = [&]() {
        std::fill(last.begin(), last.end(), LLONG_MIN / 2);
        long long miss = 0, t0 = 0;
        for(const auto &l : list) // (one L2 per XCD: the lists are independent; a jump of the clock separates them)
        {
            for(size_t p = 0; p < l.size(); p++)
                for(aoclsparse_int s = 0; s < width; s++)
                {
                    const aoclsparse_int bc = bcol[(size_t)l[p] * width + s];
                    if(bc < 0)
                        break;
                    miss += t0 + (long long)p - last[(size_t)bc] > BELL_L2_WINDOW;
                    last[(size_t)bc] = t0 + (long long)p;
                }
            t0 += (long long)l.size() + 4 * BELL_L2_WINDOW;
        }
        return miss;
    };
    // every block row exactly once, and no XCD with more than 1.06 of its share (the longest list is the launch's length)
    auto balanced = [&]() {
        size_t total = 0, longest = 0;
        for(const auto &l : list)
            total += l.size(), longest = std::max(longest, l.size());
        return total == (size_t)nbr && (double)longest <= 1.06 * (double)nbr / 8.0 + 1.0;
    };
    long long distinct = 0;
    {
        std::vector<char> seen((size_t)nbc, 0);
        for(size_t i = 0; i < bcol_size; i++)
            if(bcol[i] >= 0 && !seen[(size_t)bcol[i]])
                seen[(size_t)bcol[i]] = 1, distinct++;
    }
    if(distinct == 0)
        return;
    bell_chunk_lists(nbr, 1, list);
    const long long m1 = misses();
    bp.model_fetches_launch_order = bp.model_fetches = (double)m1 / (double)distinct;
    if(forced == 0)
        return;
    int       pick = 1;
    long long mpick = m1;
    if(forced >= 1 && forced <= nbr / 8)
    {
        bell_chunk_lists(nbr, pick = forced, list);
        mpick = misses();
    }
    else
    {
        std::vector<long long> mc;
        long long              best = m1;
        for(int ch : cand)
        {
            if(ch > 1)
                bell_chunk_lists(nbr, ch, list);
            mc.push_back(ch == 1 ? m1 : (balanced() ? misses() : LLONG_MAX)), best = std::min(best, mc.back());
        }
        for(size_t i = 0; i < mc.size(); i++)
            if((double)mc[i] <= 1.2 * (double)best)
            {
                pick = cand[i], mpick = mc[i];
                break;
            }
        if((double)mpick > 0.9 * (double)m1) // nothing worth leaving launch order for
            pick = 1, mpick = m1;
        // the lattice sweep: regions of about 40 block rows (the next plane's neighbours then sit inside the window)
        const BellLattice L = detect_bell_lattice(bcol, nbr, width);
        if(L.n1 > 0 && L.n3 >= 4)
        {
            // the line in 1, 2, 4, 8, ... pieces (eight consecutive regions then cover whole lines: pieces of 5 or 6 on the 32-node line
            // measured 6-8 % slower than pieces of 4 or 8) of about 6 block rows, b lines of them: about 40 block rows per region
            int nsx = 1;
            while(2 * nsx <= L.n1 && std::abs((double)L.n1 / (2 * nsx) - 6.0) <= std::abs((double)L.n1 / nsx - 6.0))
                nsx *= 2;
            const int a = (int)((L.n1 + nsx - 1) / nsx);
            int       b = (int)std::max<aoclsparse_int>(1, std::min<aoclsparse_int>(L.n2, 40 / a));
            const int ntiles = (int)((L.n2 + b - 1) / b);
            b                = (int)((L.n2 + ntiles - 1) / ntiles); // (tiles of equal height: the XCDs' lists of equal length)
            // z segments: as many as it takes for the regions to go round the eight XCDs evenly (segments of at least 2 planes)
            const long long regions = (long long)nsx * ntiles;
            int             nseg    = 1;
            while(regions * nseg % 8 != 0 && 2 * nseg <= L.n3 / 2)
                nseg *= 2;
            bell_lattice_lists(L, nbr, nsx, b, nseg, list);
            const long long ml = balanced() ? misses() : LLONG_MAX;
            if(ml != LLONG_MAX && (forced == -1 || (double)ml < 0.8 * (double)mpick))
            {
                pick = 0, mpick = ml;
                bp.lattice[0] = L.n1, bp.lattice[1] = L.n2, bp.lattice[2] = L.n3, bp.region[0] = a, bp.region[1] = b;
                bp.region_cut[0] = nsx, bp.region_cut[1] = nseg;
            }
        }
    }
    bp.xcd_chunk = pick, bp.model_fetches = (double)mpick / (double)distinct;
    if(pick == 1)
        return;
    if(pick > 1)
        bell_chunk_lists(nbr, pick, list);
    else
        bell_lattice_lists(BellLattice{bp.lattice[0], bp.lattice[1], bp.lattice[2]}, nbr, bp.region_cut[0], bp.region[1], bp.region_cut[1], list);
    size_t len = 0;
    for(const auto &l : list)
        len = std::max(len, l.size());
    order.assign(8 * len, -1);
    for(int x = 0; x < 8; x++)
        for(size_t p = 0; p < list[x].size(); p++)
            order[8 * p + x] = list[x][p];
    bp.order_len = (aoclsparse_int)len;
}
} // namespace
aoclsparse_status mi355::build_bell(const HostCsr &h, const DeviceCsr &d, SpmvPlan &plan, aoclsparse_matrix_data_type vt)
{
    BellPlan &bp = plan.bell;
    if(bp.tried)
        return aoclsparse_status_success;
    bp.tried = true;
    constexpr int BS = BELL_BS;
    // cheap rejections first: at least 8 entries per row on average (a half-full tile row), a matrix worth the copy
    if(vt != aoclsparse_dmat || h.m < 64 * BS || (long long)h.nnz < 8LL * h.m)
        return aoclsparse_status_success;
    const aoclsparse_int nbr = (aoclsparse_int)(((long long)h.m + BS - 1) / BS);
    std::vector<aoclsparse_int> cnt;
    try
    {
        cnt.assign((size_t)nbr + 1, 0);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_success; // (optional format)
    }
    std::atomic<bool> sorted{true}, nomem{false};
    // pass 1: distinct block columns per block row
    parallel_for(nbr, 64, [&](long long b0, long long b1) {
        std::vector<aoclsparse_int> bc;
        try
        {
            for(long long b = b0; b < b1 && sorted.load(std::memory_order_relaxed); b++)
            {
                bc.clear();
                const long long ra = b * BS, rb = std::min<long long>(h.m, ra + BS);
                for(long long i = ra; i < rb; i++)
                {
                    aoclsparse_int prev = -1, prevb = -1;
                    for(aoclsparse_int p = h.ptr[i] - h.base; p < h.ptr[i + 1] - h.base; p++)
                    {
                        const aoclsparse_int c = h.ind[p] - h.base;
                        if(c <= prev)
                        {
                            sorted.store(false, std::memory_order_relaxed);
                            break;
                        }
                        prev = c;
                        if(c / BS != prevb)
                            bc.push_back(prevb = c / BS);
                    }
                }
                std::sort(bc.begin(), bc.end());
                cnt[(size_t)b + 1] = (aoclsparse_int)(std::unique(bc.begin(), bc.end()) - bc.begin());
            }
        }
        catch(const std::bad_alloc &)
        {
            nomem.store(true);
        }
    });
    if(nomem.load())
        return aoclsparse_status_success; // (optional format)
    if(!sorted.load())
        return aoclsparse_status_success;
    long long      nblk = 0;
    aoclsparse_int width = 0;
    for(aoclsparse_int b = 0; b < nbr; b++)
        nblk += cnt[(size_t)b + 1], width = std::max(width, cnt[(size_t)b + 1]);
    if(nblk == 0)
        return aoclsparse_status_success;
    const double    fill  = (double)h.nnz / (256.0 * (double)nblk);
    const long long slots = (long long)nbr * width;
    if(fill < 0.5 || (double)slots > 1.35 * (double)nblk || slots * 256 > (1LL << 30)) // (the copy stays under 8 GiB of values)
        return aoclsparse_status_success;
    // pass 2: the block columns of every block row, ascending, empty slots (-1) last; the VALUES are scattered on the device from
    // the CSR arrays already in HBM (bell_fill_kernel) -- no host copy of the blocked values, nothing but bcol crosses PCIe
    if(!d.valid || d.m != h.m)
        return aoclsparse_status_success;
    std::vector<aoclsparse_int> bcol;
    try
    {
        bcol.assign((size_t)slots, -1);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_success; // the blocked copy is optional: without it the CSR kernels run
    }
    parallel_for(nbr, 64, [&](long long b0, long long b1) {
        std::vector<aoclsparse_int> bc;
        try
        {
            for(long long b = b0; b < b1; b++)
            {
                bc.clear();
                const long long ra = b * BS, rb = std::min<long long>(h.m, ra + BS);
                for(long long i = ra; i < rb; i++)
                {
                    aoclsparse_int prevb = -1;
                    for(aoclsparse_int p = h.ptr[i] - h.base; p < h.ptr[i + 1] - h.base; p++)
                        if((h.ind[p] - h.base) / BS != prevb)
                            bc.push_back(prevb = (h.ind[p] - h.base) / BS);
                }
                std::sort(bc.begin(), bc.end());
                bc.erase(std::unique(bc.begin(), bc.end()), bc.end());
                std::copy(bc.begin(), bc.end(), bcol.data() + (size_t)b * width);
            }
        }
        catch(const std::bad_alloc &)
        {
            nomem.store(true);
        }
    });
    if(nomem.load())
        return aoclsparse_status_success; // (optional format)
    hipStream_t       st = Runtime::get().stream();
    aoclsparse_status rc = bp.bcol.upload(bcol.data(), sizeof(aoclsparse_int) * bcol.size(), st);
    if(rc == aoclsparse_status_success)
        rc = bp.val.alloc(sizeof(double) * (size_t)slots * 256);
    if(rc == aoclsparse_status_success)
        rc = launch_bell_fill(st, h.m, d.base, d.ptr.as<aoclsparse_int>(), d.ind.as<aoclsparse_int>(), d.val.as<double>(), nbr, width,
                              bp.bcol.as<aoclsparse_int>(), bp.val.as<double>());
    if(rc != aoclsparse_status_success || hipStreamSynchronize(st) != hipSuccess) // (the host buffer above goes away)
    {
        // no room in HBM for a second copy of the matrix: not an error, the CSR kernels serve the handle
        (void)hipGetLastError();
        bp.val.release(), bp.bcol.release();
        return aoclsparse_status_success;
    }
    bp.nbr = nbr, bp.width = width, bp.nblocks = nblk, bp.fill = fill;
    // the order of the block rows over the XCDs: optional (without the list the kernels run in launch order)
    try
    {
        std::vector<aoclsparse_int> order;
        choose_bell_order(bcol.data(), bcol.size(), nbr, width, (aoclsparse_int)(((long long)h.n + BS - 1) / BS), bp, order, bell_order_forced());
        if(bp.order_len > 0
           && (bp.order.upload(order.data(), sizeof(aoclsparse_int) * order.size(), st) != aoclsparse_status_success
               || hipStreamSynchronize(st) != hipSuccess))
        {
            (void)hipGetLastError();
            bp.order.release();
            bp.order_len = 0, bp.xcd_chunk = 1, bp.model_fetches = bp.model_fetches_launch_order;
        }
    }
    catch(const std::bad_alloc &)
    {
        bp.order.release();
        bp.order_len = 0, bp.xcd_chunk = 1, bp.model_fetches = bp.model_fetches_launch_order;
    }
    bp.valid = true;
    return aoclsparse_status_success;
}
namespace
{

// Column windows for csrmm_colwin_kernel: for every block of R consecutive rows, the stretch [lo, hi] of columns its entries
// touch, rounded out to 16-byte granules.  The kernel applies when every stretch fits its LDS buffer and the stretches add up
// to at most 3.2 x the rows (what is fetched per column of B: a band of half-width g and R = 2048 rows gives 1 + 2g / R; past
// ~3 x the lane-per-row kernels' L1 / L2 reuse is as good).  O(nnz) once per handle.
aoclsparse_status detect_windows(const HostCsr &h, SpmvPlan &plan, size_t elem)
{
    MmGroups &g = plan.mm;
    if(g.win_tried)
        return aoclsparse_status_success;
    g.win_tried = true;
    const int R = csrmm_window_rows(plan.max_row_nnz, elem), epp = (int)(16 / elem), maxp = csrmm_window_max_pieces();
    if(h.m < 4 * R)
        return aoclsparse_status_success; // small matrices: nothing to win
    if(plan.max_row_nnz > 9)
    {
        // rows longer than the register cache (9 entries) finish their chain from the CSR arrays, once per column: fine for a
        // few boundary / constraint rows, not as the rule
        long long longer = 0;
        for(aoclsparse_int i = 0; i < h.m; i++)
            longer += h.ptr[i + 1] - h.ptr[i] > 9;
        if(longer * 100 > h.m)
            return aoclsparse_status_success;
    }
    const aoclsparse_int nb = (aoclsparse_int)(((long long)h.m + R - 1) / R);
    std::vector<aoclsparse_int> win;
    try
    {
        win.assign((size_t)nb * 2, 0);
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    std::atomic<bool>      fits{true};
    std::atomic<long long> total{0};
    parallel_for(nb, 16, [&](long long b0, long long b1) {
        long long sum = 0;
        for(long long b = b0; b < b1 && fits.load(std::memory_order_relaxed); b++)
        {
            const long long ra = b * R, rb = std::min<long long>(h.m, ra + R);
            aoclsparse_int  lo = INT32_MAX, hi = -1;
            for(aoclsparse_int p = h.ptr[ra] - h.base; p < h.ptr[rb] - h.base; p++)
            {
                const aoclsparse_int c = h.ind[p] - h.base;
                lo = std::min(lo, c), hi = std::max(hi, c);
            }
            if(hi < 0) // a block of empty rows
            {
                win[(size_t)b * 2] = 0, win[(size_t)b * 2 + 1] = 0;
                continue;
            }
            lo -= lo % epp;
            const long long pcs = ((long long)hi - lo + epp) / epp;
            if(pcs > maxp)
            {
                fits.store(false, std::memory_order_relaxed);
                break;
            }
            win[(size_t)b * 2] = lo, win[(size_t)b * 2 + 1] = (aoclsparse_int)pcs;
            sum += pcs * epp;
        }
        total.fetch_add(sum, std::memory_order_relaxed);
    });
    if(!fits.load() || total.load() * 10 > (long long)h.m * 32)
        return aoclsparse_status_success;
    const aoclsparse_status rc = g.windows.upload(win.data(), sizeof(aoclsparse_int) * win.size(), Runtime::get().stream());
    if(rc != aoclsparse_status_success)
    {
        (void)hipGetLastError(); // (an optional plan: the pair / lane-per-row kernels serve the handle)
        g.windows.release();
        return aoclsparse_status_success;
    }
    g.win_rows = R;
    g.win      = true;
    return aoclsparse_status_success;
}

aoclsparse_status detect_pairs(const HostCsr &h, SpmvPlan &plan)
{
    MmGroups &g = plan.mm;
    if(g.pairs_tried)
        return aoclsparse_status_success;
    g.pairs_tried = true;
    if(h.m < 2)
        return aoclsparse_status_success;
    std::vector<aoclsparse_int> pf, sg;
    try
    {
        pf.reserve((size_t)h.m / 2 + 1);
        aoclsparse_int i = 0;
        while(i < h.m)
        {
            bool ok = false;
            if(i + 1 < h.m)
            {
                const aoclsparse_int s = h.ptr[i] - h.base, e = h.ptr[i + 1] - h.base, e2 = h.ptr[i + 2] - h.base;
                const aoclsparse_int len = e - s;
                ok = len > 0 && len <= CM_DETOUR_NNZ_PER_ROW && e2 - e == len;
                for(aoclsparse_int k = 0; k < len && ok; k++)
                    ok = h.ind[e + k] == h.ind[s + k] + 1;
            }
            if(ok)
                pf.push_back(i), i += 2;
            else
                sg.push_back(i), i += 1;
        }
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    if((long long)pf.size() * 2 * 10 < (long long)h.m * 8)
        return aoclsparse_status_success;
    hipStream_t       st = Runtime::get().stream();
    aoclsparse_status rc = g.pair_first.upload(pf.data(), sizeof(aoclsparse_int) * pf.size(), st);
    if(rc == aoclsparse_status_success && !sg.empty())
        rc = g.single_rows.upload(sg.data(), sizeof(aoclsparse_int) * sg.size(), st);
    if(rc != aoclsparse_status_success)
        return rc;
    g.npairs = (aoclsparse_int)pf.size(), g.nsingles = (aoclsparse_int)sg.size();
    g.pairs  = true;
    return aoclsparse_status_success;
}

// The reference's argument checks of ?csrmm (csrmm.hpp:429-611), on the operands AS THE CALLER PASSED THEM.  quick_return is set
// when the call is complete without a product (an empty dimension, or alpha == 0 with beta == 1).  Kept apart from csrmm_t so
// that the column-sharded entry points validate the WHOLE operands (n, ldb, ldc, dim * ld range) once, before they split the
// columns: a shard's own width would let an ldb / ldc < n through that the reference rejects.
template <typename T>
static aoclsparse_status csrmm_validate(aoclsparse_operation op, const T alpha, const aoclsparse_matrix A,
                                        const aoclsparse_mat_descr descr, aoclsparse_order order, const T *B, aoclsparse_int n,
                                        aoclsparse_int ldb, const T beta, const T *C, aoclsparse_int ldc, aoclsparse_int kid,
                                        aoclsparse_matrix_data_type vt, bool &quick_return)
{
    quick_return = false;
    if(!A || !B || !C || !descr)
        return aoclsparse_status_invalid_pointer;
    if(A->input_format != aoclsparse_csr_mat)
        return aoclsparse_status_not_implemented;
    if(op != aoclsparse_operation_none && op != aoclsparse_operation_transpose
       && op != aoclsparse_operation_conjugate_transpose)
        return aoclsparse_status_invalid_value;
    if(descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_symmetric
       && descr->type != aoclsparse_matrix_type_hermitian)
        return aoclsparse_status_not_implemented;
    if((descr->type == aoclsparse_matrix_type_symmetric || descr->type == aoclsparse_matrix_type_hermitian)
       && A->m != A->n)
        return aoclsparse_status_invalid_size;
    if(order != aoclsparse_order_row && order != aoclsparse_order_column)
        return aoclsparse_status_invalid_value;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(descr->base != A->base)
        return aoclsparse_status_invalid_value;

    const aoclsparse_int m = A->m, k = A->n;
    if(m < 0 || n < 0 || k < 0)
        return aoclsparse_status_invalid_size;
    quick_return = true;
    if(m == 0 || n == 0 || k == 0)
        return aoclsparse_status_success;
    if(alpha == T(0) && beta == T(1))
        return aoclsparse_status_success;
    quick_return = false;
    if(!A->user.val || !A->user.ptr || !A->user.ind)
        return aoclsparse_status_invalid_pointer;

    const bool           tr     = op != aoclsparse_operation_none;
    const bool           colmaj = order == aoclsparse_order_column;
    const aoclsparse_int b_rows = tr ? m : k; // rows of B
    const aoclsparse_int m_c    = tr ? k : m; // rows of C
    const aoclsparse_int chk_b  = colmaj ? b_rows : n;
    const aoclsparse_int chk_c  = colmaj ? m_c : n;
    if(ldb < (chk_b > 1 ? chk_b : 1))
        return aoclsparse_status_invalid_size;
    if(ldc < (chk_c > 1 ? chk_c : 1))
        return aoclsparse_status_invalid_size;
    // LP64 index range of dim*ld (csrmm.hpp:592-611)
    const long long c_outer = colmaj ? n : m_c, b_outer = colmaj ? n : b_rows;
    if(c_outer * (long long)ldc > 2147483647LL || b_outer * (long long)ldb > 2147483647LL)
        return aoclsparse_status_invalid_size;
    if(kid > 3) // KATs of csrmm.hpp:776-837 hold kernels 0..3
        return aoclsparse_status_invalid_kid;
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status csrmm_t(aoclsparse_operation op, const T alpha, const aoclsparse_matrix A,
                          const aoclsparse_mat_descr descr, aoclsparse_order order, const T *B,
                          aoclsparse_int n, aoclsparse_int ldb, const T beta, T *C, aoclsparse_int ldc,
                          aoclsparse_int kid, aoclsparse_matrix_data_type vt)
{
    bool quick = false;
    if(const aoclsparse_status vs = csrmm_validate<T>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, kid, vt, quick);
       vs != aoclsparse_status_success || quick)
        return vs;
    const aoclsparse_int m = A->m, k = A->n;
    const bool           tr     = op != aoclsparse_operation_none;
    const bool           colmaj = order == aoclsparse_order_column;
    const aoclsparse_int b_rows = tr ? m : k; // rows of B
    const aoclsparse_int m_c    = tr ? k : m; // rows of C
    const long long      c_outer = colmaj ? n : m_c, b_outer = colmaj ? n : b_rows;

    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::recursive_mutex> sl(rt.stage_lock, std::defer_lock);
    if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
        sl.lock();

    const bool bdev = rt.is_device_pointer(B), cdev = rt.is_device_pointer(C);
    void      *dB = const_cast<T *>(B), *dC = C;
    size_t     bbytes = 0, cbytes = 0;
    if(!cdev)
    {
        st = stage_dense<T>(rt, 4, C, (aoclsparse_int)c_outer, colmaj ? m_c : n, ldc, true, &dC, &cbytes);
        if(st != aoclsparse_status_success)
            return st;
    }
    auto finish = [&]() -> aoclsparse_status {
        if(!cdev)
        {
            const aoclsparse_status cs = copy_dense<T>(rt, C, dC, (aoclsparse_int)c_outer, colmaj ? m_c : n, ldc, hipMemcpyDeviceToHost);
            if(cs != aoclsparse_status_success)
                return cs;
            MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        }
        return aoclsparse_status_success;
    };

    if(alpha == T(0))
    {
        // csrmm.hpp:614-618
        st = launch_scale_dense<T>(rt.stream(), order, static_cast<T *>(dC), m_c, n, ldc, beta);
        return st == aoclsparse_status_success ? finish() : st;
    }
    if(descr->type == aoclsparse_matrix_type_hermitian)
        return aoclsparse_status_not_implemented; // real types: hermitian csrmm is not offered

    if(!bdev)
    {
        st = stage_dense<T>(rt, 3, B, (aoclsparse_int)b_outer, colmaj ? b_rows : n, ldb, true, &dB, &bbytes);
        if(st != aoclsparse_status_success)
            return st;
    }
    DeviceCsr *d = nullptr;
    SpmvPlan  *p = nullptr;
    if(descr->type == aoclsparse_matrix_type_symmetric)
    {
        // csrmm.hpp:667-718 runs serial *_sym_ref kernels on one triangle; here the symmetric
        // operator is materialised once (derived.cpp) and the general kernels are used
        Derived *dv = nullptr;
        st = ensure_derived(const_cast<aoclsparse_matrix>(A), descr->type, descr->fill_mode, descr->diag_type,
                            false, dv);
        if(st != aoclsparse_status_success)
            return st;
        d = &dv->dev;
    }
    else
        st = ensure_spmv(const_cast<aoclsparse_matrix>(A), tr, d, p);
    if(st != aoclsparse_status_success)
        return st;
    // Every second product of a plan runs its blocks in descending order (mm_order.hpp): what one product leaves in the Infinity
    // Cache -- the end of B and C -- is where the next one starts.  The word is the calling thread's, for the launches below only.
    struct DirectionScope
    {
        explicit DirectionScope(const SpmvPlan *pl)
        {
            mm_direction_word() = (pl && plan_option(aoclsparse_mi355_option_alternate_sweeps) != 0
                                   && (pl->mm_products.fetch_add(1u, std::memory_order_relaxed) & 1u))
                                      ? MM_DESCENDING
                                      : 0;
        }
        ~DirectionScope() { mm_direction_word() = 0; }
    } direction(p);
    // Column-major with every row shorter than the KT vector width: csrmm_col_kt never forms a full group, its element is the
    // scalar chain from zero followed by fma(beta, C, alpha * cij) (csrmm_kt.cpp:158-191) -- the kid-0 arithmetic to the bit, so
    // the tuned column-major kernels serve it (kid 3 on the 5-point Laplacian: 4.7 -> 1.45 ms at 256 columns).
    const bool kt_is_ref = kid >= 1 && colmaj && p && p->valid && p->max_row_nnz > 0
                           && p->max_row_nnz < (std::is_same<T, double>::value ? 4 : 8) * (kid == 3 ? 2 : 1);
    if(kid >= 1 && descr->type == aoclsparse_matrix_type_general && !kt_is_ref)
    {
        // a pinned kid 1/2/3 asks for the arithmetic of the reference's KT kernels (csrmm.hpp:779-833): reproduced bit for bit
        // (symmetric descriptors have no KT kernel in the reference either: csrmm.hpp:667-718 runs *_sym_ref for every kid)
        const int lanes = (std::is_same<T, double>::value ? 4 : 8) * (kid == 3 ? 2 : 1);
        // Row-major: csrmm_row_kt is "c = c * beta; c = fma(alpha * a_k, b_kj, c)" for the columns its vectors cover and
        // "c = fma(a_k * b_kj, alpha, c)" for the last n mod width ones, which the tuned row-per-wave (n >= 128) and tile
        // (n < 128) kernels compute as a variant of their chain -- 2.8 -> 1.2 ms at 256 columns, 0.62 -> 0.17 at 32 on the
        // 1000^2 Laplacian.  Odd column counts and column-major operands: the plain KT kernels.
        if(!colmaj && p && p->valid)
        {
            std::shared_lock<std::shared_mutex> r(A->guard);
            aoclsparse_status                   ks = aoclsparse_status_not_implemented;
            // (matrices with row groups: the group kernels have no KT form; the per-row kernels below ignore the groups)
            if(p->nblocks > 0 && csrmm_tiled_applies<T>(n, ldb, ldc, static_cast<const T *>(dB), static_cast<const T *>(dC)))
                ks = launch_csrmm_tiled<T>(rt.stream(), d->base, alpha, d->val.as<T>(), d->ind.as<aoclsparse_int>(),
                                           d->ptr.as<aoclsparse_int>(), p->rowblocks.as<aoclsparse_int>(), p->nblocks, p->tile,
                                           p->max_row_nnz, static_cast<const T *>(dB), n, ldb, beta, static_cast<T *>(dC), ldc, lanes);
            else
                ks = launch_csrmm<T>(rt.stream(), order, d->base, alpha, d->m, d->n, d->val.as<T>(), d->ind.as<aoclsparse_int>(),
                                     d->ptr.as<aoclsparse_int>(), static_cast<const T *>(dB), n, ldb, beta, static_cast<T *>(dC),
                                     ldc, nullptr, 0, 0, false, nullptr, lanes);
            if(ks != aoclsparse_status_not_implemented)
                return ks == aoclsparse_status_success ? finish() : ks;
        }
        std::shared_lock<std::shared_mutex> r(A->guard);
        st = launch_csrmm_kt<T>(rt.stream(), order, lanes, d->base, alpha, d->m, d->val.as<T>(), d->ind.as<aoclsparse_int>(),
                                d->ptr.as<aoclsparse_int>(), static_cast<const T *>(dB), n, ldb, beta, static_cast<T *>(dC), ldc);
        return st == aoclsparse_status_success ? finish() : st;
    }
    // Column-major operands with many columns and rows longer than the column kernel's register cache: that kernel
    // would re-read A once per 4 columns (100 ms on the shell-like stand-in).  Detour: B and C are copied to packed
    // row-major scratch, the row-major kernels run, C is copied back -- three streaming passes (~0.5 ms each per GB)
    // instead; per element the arithmetic is the same chain, so the bits do not change.
    // column-major, banded matrix: the LDS-window kernel (tried first: it also serves 9-entry rows, which the detour below
    // would otherwise take)
    if(p && colmaj && !p->mm.win_tried)
    {
        std::unique_lock<std::shared_mutex> w(A->guard);
        st = detect_windows(tr ? *A->trans : A->user, *p, sizeof(T));
        if(st != aoclsparse_status_success)
            return st;
    }
    const bool windowed = colmaj && p && p->mm.win && csrmm_window_applies<T>(n, ldb, static_cast<const T *>(dB));
    // block-dense matrix: the blocked-ELL copy for the MFMA kernels (either layout)
    // (a second copy of the matrix: not under aoclsparse_memory_usage_minimal, which forbids such copies -- analysis.cpp:446)
    if(p && !windowed && !p->bell.tried && A->mem_policy == aoclsparse_memory_usage_unrestricted)
    {
        std::unique_lock<std::shared_mutex> w(A->guard);
        st = build_bell(tr ? *A->trans : A->user, *d, *p, vt);
        if(st != aoclsparse_status_success)
            return st;
    }
    const bool on_mfma_col = colmaj && !windowed && p && p->bell.valid && std::is_same<T, double>::value;
    const bool detour = colmaj && !windowed && !on_mfma_col && n >= 16 && (long long)d->nnz > (long long)CM_DETOUR_NNZ_PER_ROW * d->m;
    if(p && (!colmaj || detour) && !p->bell.valid && !p->mm.tried)
    {
        std::unique_lock<std::shared_mutex> w(A->guard);
        st = build_mm_groups(tr ? *A->trans : A->user, *p);
        if(st != aoclsparse_status_success)
            return st;
    }
    if(p && !colmaj && !p->mm.valid && !p->mm.runs_tried) // (n >= 128: the row-run kernel and the dealt order; narrower: the line blocks)
    {
        std::unique_lock<std::shared_mutex> w(A->guard);
        st = detect_row_runs(tr ? *A->trans : A->user, *p);
        if(st != aoclsparse_status_success)
            return st;
    }
    if(p && colmaj && !detour && !windowed && !on_mfma_col && !p->mm.pairs_tried)
    {
        std::unique_lock<std::shared_mutex> w(A->guard);
        st = detect_pairs(tr ? *A->trans : A->user, *p);
        if(st != aoclsparse_status_success)
            return st;
    }
    if(detour && !sl.owns_lock())
        sl.lock(); // the scratch slots are shared; taken before the handle's guard, as everywhere else
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        const bool                          grouped = p && p->mm.valid;
        const aoclsparse_int               *grp = grouped ? p->mm.first.as<aoclsparse_int>() : nullptr;
        const aoclsparse_int                ngrp = grouped ? p->mm.ngroups : 0;
        if(detour)
        {
            void *bt = nullptr, *ct = nullptr;
            st = rt.staging(5, sizeof(T) * (size_t)b_rows * (size_t)n, &bt);
            // handles with row groups: only B changes layout; the row-group kernel writes (and, beta != 0, reads) the caller's
            // column-major C in place -- two of the detour's three copy passes gone (shell-like, 256 columns: 6.6 -> see DESIGN)
            const bool direct = st == aoclsparse_status_success && grouped
                                && csrmm_groups_ccol_applies<T>(n, n, static_cast<const T *>(bt));
            if(direct)
            {
                st = launch_relayout<T>(rt.stream(), true, static_cast<const T *>(dB), static_cast<T *>(bt), b_rows, n, ldb);
                if(st == aoclsparse_status_success)
                    st = launch_csrmm_groups_ccol<T>(rt.stream(), d->base, alpha, d->val.as<T>(), d->ind.as<aoclsparse_int>(),
                                                     d->ptr.as<aoclsparse_int>(), static_cast<const T *>(bt), n, n, beta,
                                                     static_cast<T *>(dC), ldc, grp, ngrp, p->mm.max_rows, d->m, p->mm.band);
                return st == aoclsparse_status_success ? finish() : st;
            }
            if(st == aoclsparse_status_success)
                st = rt.staging(6, sizeof(T) * (size_t)m_c * (size_t)n, &ct);
            if(st == aoclsparse_status_success)
                st = launch_relayout<T>(rt.stream(), true, static_cast<const T *>(dB), static_cast<T *>(bt), b_rows, n, ldb);
            if(st == aoclsparse_status_success)
                st = launch_relayout<T>(rt.stream(), true, static_cast<const T *>(dC), static_cast<T *>(ct), m_c, n, ldc);
            if(st == aoclsparse_status_success)
                st = launch_csrmm<T>(rt.stream(), aoclsparse_order_row, d->base, alpha, d->m, d->n, d->val.as<T>(),
                                     d->ind.as<aoclsparse_int>(), d->ptr.as<aoclsparse_int>(), static_cast<const T *>(bt),
                                     n, n, beta, static_cast<T *>(ct), n, grp, ngrp, grouped ? p->mm.max_rows : 0, false, nullptr, 0,
                                     grouped ? p->mm.band : 0);
            if(st == aoclsparse_status_success)
                st = launch_relayout<T>(rt.stream(), false, static_cast<const T *>(ct), static_cast<T *>(dC), m_c, n, ldc);
        }
        else if(!colmaj && p && p->bell.valid && std::is_same<T, double>::value)
        {
            // block-dense matrix: the blocked-ELL copy on the matrix cores
            if constexpr(std::is_same<T, double>::value)
                st = launch_csrmm_bell(rt.stream(), alpha, d->m, d->n, p->bell, d->base, d->ptr.as<aoclsparse_int>(),
                                       d->ind.as<aoclsparse_int>(), d->val.as<double>(), static_cast<const double *>(dB), n, ldb,
                                       beta, static_cast<double *>(dC), ldc);
        }
        else if(!colmaj && !grouped && p && p->valid && p->nblocks > 0
                && csrmm_tiled_applies<T>(n, ldb, ldc, static_cast<const T *>(dB), static_cast<const T *>(dC)))
        {
            // narrow row-major operands (a multi-GPU column slab): row blocks of the SpMV plan -- or, on a banded matrix, the blocks that
            // follow its lines -- A staged in LDS
            // (1000^2 Laplacian, 32 columns, same box: 0.163 -> 0.151 ms with C read, 0.1255 -> 0.1172 overwritten; 64 columns unchanged)
            const bool lines = p->mm.row_runs && p->mm.band > 0 && p->mm.slab_nblocks > 0;
            st = launch_csrmm_tiled<T>(rt.stream(), d->base, alpha, d->val.as<T>(), d->ind.as<aoclsparse_int>(),
                                       d->ptr.as<aoclsparse_int>(),
                                       lines ? p->mm.slab_blocks.as<aoclsparse_int>() : p->rowblocks.as<aoclsparse_int>(),
                                       lines ? p->mm.slab_nblocks : p->nblocks, p->tile, p->max_row_nnz, static_cast<const T *>(dB), n, ldb,
                                       beta, static_cast<T *>(dC), ldc, 0, lines);
        }
        else if(on_mfma_col)
        {
            // block-dense matrix, column-major operands: the transposed MFMA product, C stored in 128-byte column segments
            if constexpr(std::is_same<T, double>::value)
                st = launch_csrmm_bell(rt.stream(), alpha, d->m, d->n, p->bell, d->base, d->ptr.as<aoclsparse_int>(),
                                       d->ind.as<aoclsparse_int>(), d->val.as<double>(), static_cast<const double *>(dB), n, ldb,
                                       beta, static_cast<double *>(dC), ldc, /*column_major=*/true);
        }
        else if(windowed)
            // column-major operands, banded matrix: each B column's stretch staged in LDS, rows' entries in registers
            st = launch_csrmm_window<T>(rt.stream(), d->base, alpha, d->m, d->val.as<T>(), d->ind.as<aoclsparse_int>(),
                                        d->ptr.as<aoclsparse_int>(), p->mm.windows.as<aoclsparse_int>(), p->mm.win_rows,
                                        p->max_row_nnz, static_cast<const T *>(dB), n, ldb, beta, static_cast<T *>(dC), ldc);
        else if(colmaj && p && p->mm.pairs && (long long)ldb * (long long)sizeof(T) < (1LL << 32)
                && reinterpret_cast<uintptr_t>(dB) % sizeof(T) == 0)
            // column-major operands, rows paired with a one-column shift: 16-byte loads / stores
            st = launch_csrmm_colpair<T>(rt.stream(), d->base, alpha, p->mm.npairs, p->mm.pair_first.as<aoclsparse_int>(),
                                         p->mm.nsingles, p->mm.single_rows.as<aoclsparse_int>(), d->val.as<T>(),
                                         d->ind.as<aoclsparse_int>(), d->ptr.as<aoclsparse_int>(), p->max_row_nnz,
                                         static_cast<const T *>(dB), n, ldb, beta, static_cast<T *>(dC), ldc);
        else
            st = launch_csrmm<T>(rt.stream(), order, d->base, alpha, d->m, d->n, d->val.as<T>(),
                                 d->ind.as<aoclsparse_int>(), d->ptr.as<aoclsparse_int>(), static_cast<const T *>(dB),
                                 n, ldb, beta, static_cast<T *>(dC), ldc, colmaj ? nullptr : grp, colmaj ? 0 : ngrp,
                                 grouped ? p->mm.max_rows : 0, !colmaj && p && p->mm.row_runs,
                                 !colmaj && p && p->mm.row_runs && p->mm.band > 0 ? p->mm.run_order.as<aoclsparse_int>() : nullptr, 0,
                                 !colmaj && p && (p->mm.row_runs || grouped) ? p->mm.band : 0);
    }
    return st == aoclsparse_status_success ? finish() : st;
}

} // namespace

// Every csrmm plan of the untransposed matrix that a later product (either layout, any column count) could ask for, built NOW
// from the host arrays: what a handle must carry before its device state is shipped to a process that will not have done the
// analysis (aoclsparse_mi355_mm_state_export / aoclsparse_mi355_comm_broadcast_matrix).
aoclsparse_status mi355::prepare_mm_plans(aoclsparse_matrix A)
{
    if(!A || !A->user.ptr || !A->user.ind || !A->user.val)
        return aoclsparse_status_invalid_pointer;
    DeviceCsr        *d = nullptr;
    SpmvPlan         *p = nullptr;
    aoclsparse_status st = ensure_spmv(A, false, d, p);
    if(st != aoclsparse_status_success)
        return st;
    std::unique_lock<std::shared_mutex> w(A->guard);
    const size_t                        elem = val_size(A->val_type);
    if(A->mem_policy == aoclsparse_memory_usage_unrestricted && (st = build_bell(A->user, *d, *p, A->val_type)) != aoclsparse_status_success)
        return st;
    if(!p->bell.valid && (st = build_mm_groups(A->user, *p)) != aoclsparse_status_success)
        return st;
    if(!p->bell.valid && !p->mm.valid && (st = detect_row_runs(A->user, *p)) != aoclsparse_status_success)
        return st;
    if((st = detect_windows(A->user, *p, elem)) != aoclsparse_status_success)
        return st;
    if(!p->mm.win && (st = detect_pairs(A->user, *p)) != aoclsparse_status_success)
        return st;
    // whatever was skipped above is never needed for this matrix: mark it tried, so that a handle that adopts this state (and
    // has done no analysis of its own) never starts one
    p->bell.tried = p->mm.tried = p->mm.runs_tried = p->mm.win_tried = p->mm.pairs_tried = true;
    return aoclsparse_status_success;
}

// in-library multi-device product (defined below, after the column-shard rule it uses)
template <typename T>
static aoclsparse_status csrmm_multi_t(aoclsparse_operation op, const T alpha, const aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, aoclsparse_order order, const T *B,
                                       const T *const *Bs, aoclsparse_int n, aoclsparse_int ldb, const T beta, T *C,
                                       T *const *Cs, aoclsparse_int ldc, aoclsparse_int ndev, const aoclsparse_int *devices,
                                       aoclsparse_matrix_data_type vt);
static int csrmm_env_devices()
{
    static const int v = [] {
        const char *e = std::getenv("AOCLSPARSE_MI355_DEVICES");
        return e ? std::atoi(e) : 0;
    }();
    return v;
}

extern "C" {

aoclsparse_status aoclsparse_dcsrmm(aoclsparse_operation op, const double alpha, const aoclsparse_matrix A,
                                    const aoclsparse_mat_descr descr, aoclsparse_order order, const double *B,
                                    aoclsparse_int n, aoclsparse_int ldb, const double beta, double *C,
                                    aoclsparse_int ldc)
{
    // AOCLSPARSE_MI355_DEVICES=N: an unchanged caller of the reference's API (host operands) gets N GPUs of the node
    if(const int nd = csrmm_env_devices(); nd > 1 && B && C && n >= 4 * nd)
    {
        const aoclsparse_status st
            = csrmm_multi_t<double>(op, alpha, A, descr, order, B, nullptr, n, ldb, beta, C, nullptr, ldc, nd, nullptr, aoclsparse_dmat);
        if(st != aoclsparse_status_not_implemented)
            return st;
    }
    return csrmm_t<double>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, -1, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_scsrmm(aoclsparse_operation op, const float alpha, const aoclsparse_matrix A,
                                    const aoclsparse_mat_descr descr, aoclsparse_order order, const float *B,
                                    aoclsparse_int n, aoclsparse_int ldb, const float beta, float *C,
                                    aoclsparse_int ldc)
{
    return csrmm_t<float>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, -1, aoclsparse_smat);
}

aoclsparse_status aoclsparse_dcsrmm_kid(aoclsparse_operation op, const double alpha, const aoclsparse_matrix A,
                                        const aoclsparse_mat_descr descr, aoclsparse_order order,
                                        const double *B, aoclsparse_int n, aoclsparse_int ldb,
                                        const double beta, double *C, aoclsparse_int ldc,
                                        const aoclsparse_int kid)
{
    return csrmm_t<double>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, kid, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_scsrmm_kid(aoclsparse_operation op, const float alpha, const aoclsparse_matrix A,
                                        const aoclsparse_mat_descr descr, aoclsparse_order order,
                                        const float *B, aoclsparse_int n, aoclsparse_int ldb, const float beta,
                                        float *C, aoclsparse_int ldc, const aoclsparse_int kid)
{
    return csrmm_t<float>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, kid, aoclsparse_smat);
}

} // extern "C"

// ---- column shards (multi-GPU: one process per GPU owns columns [j0, j1) of B and C) --------------------------------
// The reference splits B's columns over its worker threads inside the library (level3/aoclsparse_csrmm_kt.cpp:68-82:
// start = n*t/T rounded up to a multiple of the 4-column block, capped at n).  Here the "thread" is a rank: the same
// rule gives rank `rank` of `world` its slab, and *_shard computes exactly that slab of C from the FULL B / C arrays
// the caller passes (no communication: C[:, J] depends only on A and B[:, J]).
static aoclsparse_int shard_edge(aoclsparse_int n, aoclsparse_int world, aoclsparse_int t)
{
    long long e = (long long)n * t / world;
    if(e % 4)
        e += 4 - e % 4;
    return (aoclsparse_int)std::min<long long>(e, n);
}

template <typename T>
static aoclsparse_status csrmm_shard_t(aoclsparse_operation op, const T alpha, const aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, aoclsparse_order order, const T *B,
                                       aoclsparse_int n, aoclsparse_int ldb, const T beta, T *C, aoclsparse_int ldc,
                                       aoclsparse_int world, aoclsparse_int rank, aoclsparse_matrix_data_type vt)
{
    if(world < 1 || rank < 0 || rank >= world || n < 0)
        return aoclsparse_status_invalid_value;
    if(!B || !C)
        return aoclsparse_status_invalid_pointer;
    if(order != aoclsparse_order_row && order != aoclsparse_order_column)
        return aoclsparse_status_invalid_value;
    // the reference's checks on the WHOLE operands first (its own split happens after them: csrmm.hpp:592-611 precede the
    // threads of csrmm_kt.cpp:68-82): ldb / ldc against the full column count, the LP64 range of n * ld
    bool quick = false;
    if(const aoclsparse_status vs = csrmm_validate<T>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, -1, vt, quick);
       vs != aoclsparse_status_success || quick)
        return vs;
    const aoclsparse_int j0 = shard_edge(n, world, rank), j1 = shard_edge(n, world, rank + 1);
    if(j1 <= j0)
        return aoclsparse_status_success; // more ranks than 4-column blocks: this rank owns nothing
    const size_t ob = order == aoclsparse_order_column ? (size_t)j0 * (size_t)ldb : (size_t)j0;
    const size_t oc = order == aoclsparse_order_column ? (size_t)j0 * (size_t)ldc : (size_t)j0;
    return csrmm_t<T>(op, alpha, A, descr, order, B + ob, j1 - j0, ldb, beta, C + oc, ldc, -1, vt);
}

// ---- in-library multi-device csrmm (round 3, VERDICT r2 missing 4) ----------------------------------------------------
// One process, `ndev` devices: the columns of B and C are split by the same rule, worker thread i runs the ordinary csrmm
// on shard i under runtime slot i (its own device, stream and staging buffers) and on replica i of the handle -- a handle
// over the SAME host arrays whose device copy and plans live on that device.  This is the reference's in-call column split
// (level3/aoclsparse_csrmm_kt.cpp:68-82: one OpenMP thread per column range) with a GPU in the place of a thread; there is
// no collective on the data path, and A reaches every device once, when its replica is built (hints are copied and
// aoclsparse_optimize runs under the slot: the host analysis runs on `ndev` threads at the same time and every device
// fills its copy over its own PCIe link).
// The device state a csrmm needs (CSR arrays, row-block plan, whatever csrmm plans the primary handle has already built),
// copied device to device onto the CURRENT slot's device: what SURVEY 8e calls "A replicated in its device format once".
// Parts the primary has not built stay unbuilt and are made on first use on the replica's device, as on any handle.
static aoclsparse_status clone_mm_state(const _aoclsparse_matrix &A, _aoclsparse_matrix &R, hipStream_t s)
{
    const DeviceCsr &a = A.dev_user;
    DeviceCsr       &r = R.dev_user;
    aoclsparse_status st;
#define MI355_CLONE(dst, src)                                 \
    if((st = (dst).clone_from((src), s)) != aoclsparse_status_success) \
    return st
    MI355_CLONE(r.ptr, a.ptr);
    MI355_CLONE(r.ind, a.ind);
    MI355_CLONE(r.val, a.val);
    r.m = a.m, r.n = a.n, r.nnz = a.nnz, r.base = a.base;
    const SpmvPlan &pa = A.plan_user;
    SpmvPlan       &pr = R.plan_user;
    pr.nblocks = pa.nblocks, pr.long_rows = pa.long_rows, pr.max_row_nnz = pa.max_row_nnz, pr.tile = pa.tile;
    MI355_CLONE(pr.rowblocks, pa.rowblocks);
    MI355_CLONE(pr.rowblocks4, pa.rowblocks4);
    pr.heavy_first = pa.heavy_first;
    // (the SELL-64 and merge-path copies serve ?mv only: not cloned, marked untried, built lazily if the replica ever needs them)
    const MmGroups &ga = pa.mm;
    MmGroups       &gr = pr.mm;
    gr.runs_tried = ga.runs_tried, gr.row_runs = ga.row_runs, gr.band = ga.band;
    MI355_CLONE(gr.run_order, ga.run_order);
    gr.slab_nblocks = ga.slab_nblocks;
    MI355_CLONE(gr.slab_blocks, ga.slab_blocks);
    gr.ngroups = ga.ngroups, gr.max_rows = ga.max_rows, gr.valid = ga.valid, gr.tried = ga.tried;
    MI355_CLONE(gr.first, ga.first);
    gr.pairs_tried = ga.pairs_tried, gr.pairs = ga.pairs, gr.npairs = ga.npairs, gr.nsingles = ga.nsingles;
    MI355_CLONE(gr.pair_first, ga.pair_first);
    MI355_CLONE(gr.single_rows, ga.single_rows);
    gr.win_tried = ga.win_tried, gr.win = ga.win, gr.win_rows = ga.win_rows;
    MI355_CLONE(gr.windows, ga.windows);
    const BellPlan &ba = pa.bell;
    BellPlan       &bl = pr.bell;
    bl.tried = ba.tried, bl.valid = ba.valid, bl.nbr = ba.nbr, bl.width = ba.width, bl.nblocks = ba.nblocks, bl.fill = ba.fill;
    MI355_CLONE(bl.val, ba.val);
    MI355_CLONE(bl.bcol, ba.bcol);
    bl.order_len = ba.order_len, bl.xcd_chunk = ba.xcd_chunk, bl.model_fetches = ba.model_fetches;
    bl.model_fetches_launch_order = ba.model_fetches_launch_order;
    for(int i = 0; i < 3; i++)
        bl.lattice[i] = ba.lattice[i];
    for(int i = 0; i < 2; i++)
        bl.region[i] = ba.region[i], bl.region_cut[i] = ba.region_cut[i];
    MI355_CLONE(bl.order, ba.order);
#undef MI355_CLONE
    MI355_HIP_TRY(hipStreamSynchronize(s));
    r.valid  = true;
    pr.valid = true;
    return aoclsparse_status_success;
}

static aoclsparse_status get_replica(aoclsparse_matrix A, int slot_idx, aoclsparse_matrix &out)
{
    out = nullptr;
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if((int)A->replicas.size() > slot_idx && A->replicas[slot_idx])
        {
            out = A->replicas[slot_idx];
            return aoclsparse_status_success;
        }
    }
    aoclsparse_matrix R  = nullptr;
    aoclsparse_status st = aoclsparse_status_wrong_type;
    if(A->val_type == aoclsparse_dmat)
        st = aoclsparse_create_dcsr(&R, A->base, A->m, A->n, A->nnz, A->user.ptr, A->user.ind, static_cast<double *>(A->user.val));
    else if(A->val_type == aoclsparse_smat)
        st = aoclsparse_create_scsr(&R, A->base, A->m, A->n, A->nnz, A->user.ptr, A->user.ind, static_cast<float *>(A->user.val));
    if(st != aoclsparse_status_success)
        return st;
    // a csrmm replica serves csrmm only: it gets the mm hints (no TRSV level plans, no SELL copies on the other devices)
    try
    {
        for(const Hint &h : A->hints)
            if(h.act == action_mm)
                R->hints.push_back(h), R->hints.back().optimized = false;
        if(R->hints.empty())
        {
            Hint h{};
            h.act = action_mm, h.trans = aoclsparse_operation_none, h.type = aoclsparse_matrix_type_general;
            h.fill = aoclsparse_fill_mode_lower, h.nop = 1, h.kid = -1;
            R->hints.push_back(h);
        }
    }
    catch(const std::bad_alloc &)
    {
        aoclsparse_destroy(&R);
        return aoclsparse_status_memory_error;
    }
    R->mem_policy = A->mem_policy;
    // Fast path: the primary handle already holds its device format (set_mm_hint + optimize, or an earlier product) -> peer copy.
    // Otherwise the replica analyses its own copy (every device at the same time).
    bool cloned = false;
    {
        std::shared_lock<std::shared_mutex> r(A->guard);
        if(A->dev_user.valid && A->plan_user.valid && A->mem_policy == aoclsparse_memory_usage_unrestricted)
        {
            Runtime &rt = Runtime::get(), &pr = Runtime::primary();
            if(rt.init() == aoclsparse_status_success)
            {
                // whatever the primary still has in flight on its stream (uploads of the plan) must have landed
                (void)hipStreamSynchronize(pr.stream());
                if(rt.device != pr.device)
                {
                    int can = 0;
                    if(hipDeviceCanAccessPeer(&can, rt.device, pr.device) == hipSuccess && can)
                        (void)hipDeviceEnablePeerAccess(pr.device, 0); // "already enabled" is fine
                    (void)hipGetLastError();
                }
                cloned = clone_mm_state(*A, *R, rt.stream()) == aoclsparse_status_success;
                if(!cloned)
                {
                    (void)hipGetLastError();
                    R->dev_user.valid = R->plan_user.valid = false;
                }
            }
        }
    }
    if(cloned)
    {
        for(Hint &h : R->hints)
            h.optimized = true;
        st = aoclsparse_status_success;
    }
    else
        st = aoclsparse_optimize(R); // device copy + plans on the CURRENT slot's device
    if(st != aoclsparse_status_success)
    {
        aoclsparse_destroy(&R);
        return st;
    }
    std::unique_lock<std::shared_mutex> w(A->guard);
    if((int)A->replicas.size() <= slot_idx)
        A->replicas.resize((size_t)slot_idx + 1, nullptr);
    if(A->replicas[slot_idx]) // another call built it meanwhile
    {
        w.unlock();
        aoclsparse_destroy(&R);
        std::shared_lock<std::shared_mutex> r(A->guard);
        out = A->replicas[slot_idx];
        return aoclsparse_status_success;
    }
    A->replicas[slot_idx] = R;
    A->replicas_cloned += cloned ? 1 : 0;
    out                   = R;
    return aoclsparse_status_success;
}

// One persistent worker thread per runtime slot (>= 1): a multi-device call posts slot i's share to worker i and runs slot 0's
// itself.  Starting a std::thread per device and call cost ~0.15 ms per call -- as much as the 32-column slab of the 8-device
// job runs on its GPU.  A worker that has just finished a job spins for a few tens of microseconds before it sleeps, so calls
// issued back to back (an iterative caller) hand over without a futex round trip.  Workers and their mailboxes live for the life
// of the process (leaked on purpose: a sleeping detached thread must never see its condition variable destroyed).
namespace
{
    struct SlotWorker
    {
        std::mutex                          m;
        std::condition_variable             cv;
        std::atomic<std::function<void()> *> job{nullptr};
        std::atomic<bool>                   busy{false};
        bool                                started = false;
    };
    std::mutex g_multi_call; // one multi-device call at a time (the workers are a shared resource)
    // wall time each slot spent on its share in the LAST multi-device call (its launches + the wait for its stream): what
    // aoclsparse_mi355_multi_last_ms reports, so that a caller can see the per-device times behind one call's wall clock
    float g_multi_last_ms[64];
    int   g_multi_last_n = 0;

    void spin_pause()
    {
        __builtin_ia32_pause();
    }

    void slot_worker_loop(SlotWorker *w)
    {
        for(;;)
        {
            std::function<void()> *j = nullptr;
            // (~0.1 ms of spinning: enough for a caller that issues its calls back to back, short enough not to hold a core)
            for(int spin = 0; spin < 4000 && !(j = w->job.load(std::memory_order_acquire)); spin++)
                spin_pause();
            if(!j)
            {
                std::unique_lock<std::mutex> l(w->m);
                w->cv.wait(l, [&] { return w->job.load(std::memory_order_acquire) != nullptr; });
                j = w->job.load(std::memory_order_acquire);
            }
            (*j)();
            {
                std::lock_guard<std::mutex> l(w->m);
                w->job.store(nullptr, std::memory_order_release);
                w->busy.store(false, std::memory_order_release);
            }
            w->cv.notify_all();
        }
    }

    // worker of slot idx (1..63), started on first use; nullptr when no thread can be had
    SlotWorker *slot_worker(int idx)
    {
        static std::mutex  m;
        static SlotWorker *workers[64] = {nullptr};
        std::lock_guard<std::mutex> g(m);
        if(!workers[idx])
            workers[idx] = new(std::nothrow) SlotWorker;
        SlotWorker *w = workers[idx];
        if(w && !w->started)
        {
            try
            {
                std::thread(slot_worker_loop, w).detach();
                w->started = true;
            }
            catch(const std::system_error &)
            {
                return nullptr;
            }
        }
        return w;
    }

    void post(SlotWorker *w, std::function<void()> *job)
    {
        {
            std::lock_guard<std::mutex> l(w->m);
            w->busy.store(true, std::memory_order_release);
            w->job.store(job, std::memory_order_release);
        }
        w->cv.notify_all();
    }

    void wait_done(SlotWorker *w)
    {
        for(int spin = 0; spin < 20000 && w->busy.load(std::memory_order_acquire); spin++)
            spin_pause();
        if(w->busy.load(std::memory_order_acquire))
        {
            std::unique_lock<std::mutex> l(w->m);
            w->cv.wait(l, [&] { return !w->busy.load(std::memory_order_acquire); });
        }
    }
} // namespace

// slabs == false: B / C are the FULL operands (host memory, or memory every device can address); device i computes columns
// [j0_i, j1_i).  slabs == true: Bs[i] / Cs[i] are device i's own slabs (device memory there), n_i = its shard width.
template <typename T>
static aoclsparse_status csrmm_multi_t(aoclsparse_operation op, const T alpha, const aoclsparse_matrix A,
                                       const aoclsparse_mat_descr descr, aoclsparse_order order, const T *B,
                                       const T *const *Bs, aoclsparse_int n, aoclsparse_int ldb, const T beta, T *C,
                                       T *const *Cs, aoclsparse_int ldc, aoclsparse_int ndev, const aoclsparse_int *devices,
                                       aoclsparse_matrix_data_type vt)
{
    const bool slabs = Bs != nullptr || Cs != nullptr;
    if(!A || !descr || (slabs ? (!Bs || !Cs) : (!B || !C)))
        return aoclsparse_status_invalid_pointer;
    if(ndev < 1 || ndev > 64 || n < 0)
        return aoclsparse_status_invalid_value;
    if(A->val_type != vt)
        return aoclsparse_status_wrong_type;
    if(!slabs)
    {
        // full operands: the reference's status for the WHOLE call comes first, before any replica is built (a replica's
        // creation error must not mask it, and a shard's width must not let an ldb / ldc < n through)
        bool quick = false;
        if(const aoclsparse_status vs = csrmm_validate<T>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, -1, vt, quick);
           vs != aoclsparse_status_success || quick)
            return vs;
    }
    Runtime          &pr = Runtime::primary();
    aoclsparse_status st = pr.init();
    if(st != aoclsparse_status_success)
        return st;
    int count = 0;
    if(hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return aoclsparse_status_internal_error;
    // (the vectors below allocate: nothing may leave through the C ABI as an exception)
    try
    {
    std::vector<int> dev((size_t)ndev);
    for(int i = 0; i < ndev; i++)
    {
        dev[i] = devices ? (int)devices[i] : (pr.device + i) % count;
        if(dev[i] < 0 || dev[i] >= count)
            return aoclsparse_status_invalid_value;
    }
    if(dev[0] != pr.device)
        return aoclsparse_status_invalid_value; // slot 0 is the library's own device (AOCLSPARSE_MI355_DEVICE / current device)
    if(!slabs)
    {
        // full operands must be reachable from every device: host memory is (each device stages its own slab); memory of
        // ONE device is not -- callers with device-resident operands pass per-device slabs (?csrmm_multi_slabs)
        bool one_device = true;
        for(int i = 1; i < ndev; i++)
            one_device = one_device && dev[i] == dev[0];
        if(!one_device && (pr.is_device_pointer(B) || pr.is_device_pointer(C)))
            return aoclsparse_status_not_implemented;
    }
    std::vector<Runtime *> rts((size_t)ndev, &pr);
    for(int i = 1; i < ndev; i++)
    {
        rts[i] = Runtime::slot(i, dev[i]);
        if(!rts[i])
            return aoclsparse_status_invalid_value; // the slot was used with another device before
    }
    std::vector<aoclsparse_status> res((size_t)ndev, aoclsparse_status_success);
    auto                           work = [&](int i) {
        const auto t_begin = std::chrono::steady_clock::now();
        struct Lap
        {
            int                                   i;
            std::chrono::steady_clock::time_point t0;
            ~Lap()
            {
                g_multi_last_ms[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
        } lap{i, t_begin};
        try
        {
            RuntimeScope sc(rts[i]);
            if(sc.status != aoclsparse_status_success)
            {
                res[i] = sc.status;
                return;
            }
            aoclsparse_matrix Ai = A;
            if(i > 0)
            {
                res[i] = get_replica(A, i, Ai);
                if(res[i] != aoclsparse_status_success)
                    return;
            }
            const aoclsparse_int j0 = shard_edge(n, ndev, i), j1 = shard_edge(n, ndev, i + 1);
            if(j1 <= j0)
                return;
            if(slabs)
            {
                if(!Bs[i] || !Cs[i])
                {
                    res[i] = aoclsparse_status_invalid_pointer;
                    return;
                }
                res[i] = csrmm_t<T>(op, alpha, Ai, descr, order, Bs[i], j1 - j0, ldb, beta, Cs[i], ldc, -1, vt);
            }
            else
                res[i] = csrmm_shard_t<T>(op, alpha, Ai, descr, order, B, n, ldb, beta, C, ldc, ndev, i, vt);
            // the call returns when every device is done: secondary streams are not visible to the caller
            if(res[i] == aoclsparse_status_success && hipStreamSynchronize(Runtime::get().stream()) != hipSuccess)
                res[i] = aoclsparse_status_internal_error;
        }
        catch(const std::bad_alloc &)
        {
            res[i] = aoclsparse_status_memory_error;
        }
        catch(...)
        {
            res[i] = aoclsparse_status_internal_error;
        }
    };
    {
        std::lock_guard<std::mutex>        one_call(g_multi_call);
        g_multi_last_n = ndev;
        std::vector<std::function<void()>> jobs((size_t)ndev);
        std::vector<SlotWorker *>          posted;
        posted.reserve((size_t)ndev);
        for(int i = 1; i < ndev; i++)
        {
            SlotWorker *w = slot_worker(i);
            if(!w)
            {
                work(i); // no thread to be had: this device's share runs from here
                continue;
            }
            jobs[(size_t)i] = [&work, i] { work(i); };
            post(w, &jobs[(size_t)i]);
            posted.push_back(w);
        }
        work(0);
        for(SlotWorker *w : posted)
            wait_done(w);
    }
    for(int i = 0; i < ndev; i++)
        if(res[i] != aoclsparse_status_success)
            return res[i];
    return aoclsparse_status_success;
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
    catch(const std::exception &)
    {
        return aoclsparse_status_internal_error;
    }
}

extern "C" {

aoclsparse_status aoclsparse_mi355_plan_block_row_order(aoclsparse_int nbr, aoclsparse_int width, aoclsparse_int nbc, const aoclsparse_int *bcol,
                                                        aoclsparse_int forced, aoclsparse_int *order, aoclsparse_int order_capacity,
                                                        aoclsparse_int *order_len, aoclsparse_int info[8])
{
    if(!bcol || !order_len || !info || (order_capacity > 0 && !order))
        return aoclsparse_status_invalid_pointer;
    if(nbr < 0 || width < 1 || nbc < 1 || order_capacity < 0 || (long long)nbr * width > (1LL << 31))
        return aoclsparse_status_invalid_size;
    for(long long i = 0; i < (long long)nbr * width; i++)
        if(bcol[i] < -1 || bcol[i] >= nbc)
            return aoclsparse_status_invalid_index_value;
    try
    {
        BellPlan                    bp;
        std::vector<aoclsparse_int> ord;
        choose_bell_order(bcol, (size_t)nbr * (size_t)width, nbr, width, nbc, bp, ord, forced == -2 ? BELL_ORDER_AUTO : (int)forced);
        if((long long)ord.size() > (long long)order_capacity)
            return aoclsparse_status_invalid_size;
        std::copy(ord.begin(), ord.end(), order);
        *order_len = bp.order_len;
        info[0] = bp.xcd_chunk, info[1] = bp.lattice[0], info[2] = bp.lattice[1], info[3] = bp.lattice[2];
        info[4] = bp.region[0], info[5] = bp.region[1];
        info[6] = (aoclsparse_int)(bp.model_fetches * 1000.0 + 0.5), info[7] = (aoclsparse_int)(bp.model_fetches_launch_order * 1000.0 + 0.5);
        return aoclsparse_status_success;
    }
    catch(const std::bad_alloc &)
    {
        return aoclsparse_status_memory_error;
    }
}

aoclsparse_status aoclsparse_mi355_column_shard(aoclsparse_int n, aoclsparse_int world, aoclsparse_int rank,
                                                aoclsparse_int *j0, aoclsparse_int *j1)
{
    if(!j0 || !j1)
        return aoclsparse_status_invalid_pointer;
    if(world < 1 || rank < 0 || rank >= world || n < 0)
        return aoclsparse_status_invalid_value;
    *j0 = shard_edge(n, world, rank);
    *j1 = shard_edge(n, world, rank + 1);
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_dcsrmm_shard(aoclsparse_operation op, const double alpha, const aoclsparse_matrix A,
                                                const aoclsparse_mat_descr descr, aoclsparse_order order,
                                                const double *B, aoclsparse_int n, aoclsparse_int ldb,
                                                const double beta, double *C, aoclsparse_int ldc,
                                                aoclsparse_int world, aoclsparse_int rank)
{
    return csrmm_shard_t<double>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, world, rank, aoclsparse_dmat);
}

aoclsparse_status aoclsparse_mi355_dcsrmm_multi(aoclsparse_operation op, const double alpha, const aoclsparse_matrix A,
                                                const aoclsparse_mat_descr descr, aoclsparse_order order,
                                                const double *B, aoclsparse_int n, aoclsparse_int ldb,
                                                const double beta, double *C, aoclsparse_int ldc, aoclsparse_int ndev,
                                                const aoclsparse_int *devices)
{
    return csrmm_multi_t<double>(op, alpha, A, descr, order, B, nullptr, n, ldb, beta, C, nullptr, ldc, ndev, devices,
                                 aoclsparse_dmat);
}

aoclsparse_status aoclsparse_mi355_scsrmm_multi(aoclsparse_operation op, const float alpha, const aoclsparse_matrix A,
                                                const aoclsparse_mat_descr descr, aoclsparse_order order, const float *B,
                                                aoclsparse_int n, aoclsparse_int ldb, const float beta, float *C,
                                                aoclsparse_int ldc, aoclsparse_int ndev, const aoclsparse_int *devices)
{
    return csrmm_multi_t<float>(op, alpha, A, descr, order, B, nullptr, n, ldb, beta, C, nullptr, ldc, ndev, devices,
                                aoclsparse_smat);
}

aoclsparse_status aoclsparse_mi355_dcsrmm_multi_slabs(aoclsparse_operation op, const double alpha, const aoclsparse_matrix A,
                                                      const aoclsparse_mat_descr descr, aoclsparse_order order,
                                                      const double *const *B_slabs, aoclsparse_int n, aoclsparse_int ldb,
                                                      const double beta, double *const *C_slabs, aoclsparse_int ldc,
                                                      aoclsparse_int ndev, const aoclsparse_int *devices)
{
    if(!B_slabs || !C_slabs)
        return aoclsparse_status_invalid_pointer;
    return csrmm_multi_t<double>(op, alpha, A, descr, order, nullptr, B_slabs, n, ldb, beta, nullptr, C_slabs, ldc, ndev,
                                 devices, aoclsparse_dmat);
}

aoclsparse_int aoclsparse_mi355_multi_last_ms(float *ms, aoclsparse_int capacity)
{
    std::lock_guard<std::mutex> one_call(g_multi_call);
    for(int i = 0; ms && i < g_multi_last_n && i < capacity; i++)
        ms[i] = g_multi_last_ms[i];
    return g_multi_last_n;
}

aoclsparse_int aoclsparse_mi355_replica_count(const aoclsparse_matrix A)
{
    if(!A)
        return -1;
    std::shared_lock<std::shared_mutex> r(A->guard);
    aoclsparse_int                      c = 0;
    for(auto &p : A->replicas)
        c += p != nullptr;
    return c;
}

aoclsparse_int aoclsparse_mi355_replicas_cloned(const aoclsparse_matrix A)
{
    if(!A)
        return -1;
    std::shared_lock<std::shared_mutex> r(A->guard);
    return A->replicas_cloned;
}

aoclsparse_status aoclsparse_mi355_scsrmm_shard(aoclsparse_operation op, const float alpha, const aoclsparse_matrix A,
                                                const aoclsparse_mat_descr descr, aoclsparse_order order,
                                                const float *B, aoclsparse_int n, aoclsparse_int ldb, const float beta,
                                                float *C, aoclsparse_int ldc, aoclsparse_int world, aoclsparse_int rank)
{
    return csrmm_shard_t<float>(op, alpha, A, descr, order, B, n, ldb, beta, C, ldc, world, rank, aoclsparse_smat);
}

} // extern "C"
