// spgemm_kernels.hip -- C = A * B for two CSR matrices (the sp2m / spmm / csr2m path), gfx950.
//
// Reference: two-stage Gustavson with a dense per-thread accumulator of length n
// (level3/aoclsparse_csr2m.cpp:46-302 count, :310-543 finalize).  Row i of C lists its columns in
// FIRST-TOUCH order (walk row i of A left to right, for each entry walk the matching row of B left to
// right) and each value is accumulated in that same order: first product stored, later ones added with
// a contracted multiply-add (:489-498).  Both properties are reproduced exactly here.
//
// Round 4 design (spgemm_hash_kernel): a GROUP of 16 or 64 lanes owns one row of C.  The entries of one B row are taken 16 / 64
// at a time; they are distinct columns when the chunk is strictly increasing, which is checked on the fly -- then the lanes look
// their columns up in a hash table together, the new columns get their list slots by ballot rank (= the order the reference's
// serial walk first touches them) and every column receives its one update.  Chunks that are not strictly increasing (unsorted
// rows, repeated columns: both legal, csr_util.cpp:244) are walked one entry at a time.  The rows are binned by the size of
// their list (count pass: the upper bound sum of the touched B rows, capped at n; fill pass: the exact count): the table, the
// list and the partial sums of a row live in LDS up to 2,048 entries (8,192 in the count pass, which keeps keys only), in a
// global slab of exactly the needed size above that.  Rounds 1-3 walked every product through a wave-wide list search: 7.3 ms
// of kernels for A * A on the 1000^2 Laplacian against 0.5 ms now, 32 ms against 1.6 on a 100,000-row shell mesh.
//
// Stage 1 returns the per-row counts; the host prefix-sums them (64-bit, as the reference) and
// allocates C; stage 2 writes col_ind / val of C at row_ptr_C.  Integer output is bit-exact, fp output
// bit-identical to the reference's single-thread order.
//
// Bound: neither HBM nor MFMA -- irregular, latency/instruction bound; algorithmic traffic is
// 12 B per entry of A and of the touched B rows + 12 B per entry of C.
#include "internal.hpp"

#include <algorithm>

#include <hip/hip_runtime.h>

namespace mi355
{


__device__ __forceinline__ double sp_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float sp_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}
// complex: the reference's std::complex multiply-add (csr2m.cpp:489-498 with T = std::complex); component-wise fma
template <typename R>
__device__ __forceinline__ cplx<R> sp_fma(cplx<R> a, cplx<R> b, cplx<R> c)
{
    c.re = sp_fma(a.re, b.re, c.re);
    c.re = sp_fma(-a.im, b.im, c.re);
    c.im = sp_fma(a.re, b.im, c.im);
    c.im = sp_fma(a.im, b.re, c.im);
    return c;
}
__device__ __forceinline__ double sp_mul(double a, double b)
{
    return a * b;
}
__device__ __forceinline__ float sp_mul(float a, float b)
{
    return a * b;
}
template <typename R>
__device__ __forceinline__ cplx<R> sp_mul(cplx<R> a, cplx<R> b)
{
    return sp_fma(a, b, cplx<R>(R(0), R(0)));
}
__device__ __forceinline__ double sp_conj(double a, bool)
{
    return a;
}
__device__ __forceinline__ float sp_conj(float a, bool)
{
    return a;
}
template <typename R>
__device__ __forceinline__ cplx<R> sp_conj(cplx<R> a, bool on)
{
    return on ? cplx<R>(a.re, -a.im) : a;
}

// value of lane `src` of the caller's group of `width` lanes
__device__ __forceinline__ double spg_shfl(double v, int src, int width)
{
    return __shfl(v, src, width);
}
__device__ __forceinline__ float spg_shfl(float v, int src, int width)
{
    return __shfl(v, src, width);
}
template <typename R>
__device__ __forceinline__ cplx<R> spg_shfl(cplx<R> v, int src, int width)
{
    return cplx<R>(__shfl(v.re, src, width), __shfl(v.im, src, width));
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------------
// LDS mode (GLOBAL = false): per group hkey[H] (open addressing, linear probing, -1 = empty) and, in the fill pass, hpos[H] (list
// slot of the key), list[H / 2] (columns in first-touch order) and acc[H / 2].  The caller guarantees that the row's list has at
// most H / 2 entries.  GLOBAL = true: the same four arrays in a global slab, one record per row (SpgHeavy: row, table size,
// offsets); the table is read with agent-scope loads and every chunk ends with a release + acquire fence, because the inserts are
// atomics executed at the L2 and the CU's vector L1 would otherwise keep answering with the line it loaded before them.
//
// Order: A's row is walked left to right; the entries of one B row are taken G at a time, lane t = t-th entry.  Either way every
// column sees its products in the reference's order: first product stored, the later ones added with a contracted multiply-add
// (csr2m.cpp:489-498).
template <int G>
__device__ __forceinline__ unsigned long long spg_group_bits(unsigned long long wave_mask, int grp)
{
    if constexpr(G == 64)
        return wave_mask;
    else
        return (wave_mask >> (G * grp)) & ((1ull << G) - 1ull);
}

template <bool GLOBAL>
__device__ __forceinline__ int spg_key(const int *p)
{
    if constexpr(GLOBAL)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        return *p;
}

template <bool GLOBAL>
__device__ __forceinline__ void spg_sync()
{
    if constexpr(GLOBAL)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

template <typename T, bool FILL, int G, int LOGH, int NG, bool GLOBAL>
__global__ __launch_bounds__(G *NG) void spgemm_hash_kernel(aoclsparse_int nrows, const aoclsparse_int *__restrict__ rows,
                                                           const SpgHeavy *__restrict__ heavy, int *g_key, int *g_pos, int *g_list,
                                                           T *g_acc, int base_a, const aoclsparse_int *__restrict__ ptr_a,
                                                           const aoclsparse_int *__restrict__ ind_a,
                                                           const T *__restrict__ val_a, int base_b,
                                                           const aoclsparse_int *__restrict__ ptr_b,
                                                           const aoclsparse_int *__restrict__ ind_b,
                                                           const T *__restrict__ val_b,
                                                           const aoclsparse_int *__restrict__ ptr_c,
                                                           aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a, bool conj_b,
                                                           unsigned int *bad)
{
    constexpr int EMPTY = -1;
    constexpr int LH = GLOBAL ? 1 : (1 << LOGH); // (LDS arrays of the global mode: one dummy element)
    using PosT       = typename std::conditional<GLOBAL, int, unsigned short>::type;
    __shared__ int     s_key[NG][LH];
    __shared__ unsigned short s_pos[FILL ? NG : 1][FILL && !GLOBAL ? LH : 1];
    __shared__ int     s_list[FILL ? NG : 1][FILL && !GLOBAL ? LH / 2 : 1];
    __shared__ T       s_acc[FILL ? NG : 1][FILL && !GLOBAL ? LH / 2 : 1];
    const int          grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int          wgrp = (threadIdx.x & 63) / G; // group inside its wavefront (ballots are per wavefront)
    const long long    g   = (long long)blockIdx.x * NG + grp;
    if(g >= nrows)
        return;
    int   i, logh = LOGH;
    int  *hkey, *list;
    PosT *hpos;
    T    *acc;
    if constexpr(GLOBAL)
    {
        const SpgHeavy r = heavy[g];
        i = r.row, logh = r.logh;
        hkey = g_key + r.h_off, hpos = reinterpret_cast<PosT *>(g_pos) + r.h_off, list = g_list + r.c_off, acc = g_acc + r.c_off;
    }
    else
    {
        i    = rows ? rows[g] : (int)g; // (no list: every row of the matrix is in this bin)
        hkey = s_key[grp], hpos = reinterpret_cast<PosT *>(s_pos[FILL ? grp : 0]), list = s_list[FILL ? grp : 0];
        acc  = s_acc[FILL ? grp : 0];
    }
    const int      H     = 1 << logh;
    const unsigned hmask = (unsigned)H - 1u;
    // The list may hold `limit` entries: in the fill pass what the caller's row_ptr gives this row -- a row_ptr that is not this
    // product's must neither run a row into its neighbour's segment nor fill the table (a full table never ends a probe) --, in the
    // count pass half the table (never reached: the bins are chosen by the upper bound).  A row that would exceed it stops and
    // raises *bad; so does a fill that ends short of its segment.
    int limit = H / 2;
    if constexpr(FILL)
        limit = min(limit, ptr_c[i + 1] - ptr_c[i]);
    bool dead = false;
    for(int t = gl; t < H; t += G)
        hkey[t] = EMPTY;
    spg_sync<GLOBAL>();
    const unsigned long long lt = (1ull << gl) - 1ull;
    const int      hshift = 32 - logh;
    auto           hash   = [&](int c) { return (unsigned)c * 2654435761u >> hshift; };
    int  len = 0;
    const int ja = ptr_a[i] - base_a, je = ptr_a[i + 1] - base_a;
    for(int j0 = ja; j0 < je && !dead; j0 += G)
    {
        // this round's entries of A's row, one per lane: value and the extent of the matching B row
        const bool have = j0 + gl < je;
        int        my_kb = 0, my_ke = 0;
        T          my_va = T(0);
        if(have)
        {
            const int ca = ind_a[j0 + gl] - base_a;
            my_kb = ptr_b[ca] - base_b, my_ke = ptr_b[ca + 1] - base_b;
            if constexpr(FILL)
                my_va = sp_conj(val_a[j0 + gl], conj_a);
        }
        const int nj = min(G, je - j0);
        for(int jj = 0; jj < nj && !dead; jj++)
        {
            const int kb = __shfl(my_kb, jj, G), ke = __shfl(my_ke, jj, G);
            T         va = T(0);
            if constexpr(FILL)
                va = spg_shfl(my_va, jj, G);
            int carry = -1;
            for(int k0 = kb; k0 < ke && !dead; k0 += G)
            {
                const int  k     = k0 + gl;
                const bool valid = k < ke;
                const int  c     = valid ? ind_b[k] - base_b : 0x7fffffff;
                T          vb    = T(0);
                if constexpr(FILL)
                    if(valid)
                        vb = sp_conj(val_b[k], conj_b);
                int prev = __shfl_up(c, 1, G);
                if(gl == 0)
                    prev = carry;
                carry = __shfl(c, G - 1, G);
                const unsigned long long bad = spg_group_bits<G>(__ballot(valid && c <= prev), wgrp);
                if(!bad)
                {
                    // distinct columns: look up together, append the new ones in lane order
                    int      pos = -1;
                    unsigned h   = hash(c);
                    if(valid)
                        for(;;)
                        {
                            const int key = spg_key<GLOBAL>(&hkey[h]);
                            if(key == EMPTY)
                                break;
                            if(key == c)
                            {
                                pos = FILL ? (int)hpos[h] : 0;
                                break;
                            }
                            h = (h + 1) & hmask;
                        }
                    const bool               isnew = valid && pos < 0;
                    const unsigned long long nm    = spg_group_bits<G>(__ballot(isnew), wgrp);
                    if(len + __popcll(nm) > limit) // (uniform inside the group)
                    {
                        dead = true;
                        break;
                    }
                    if(isnew)
                    {
                        const int slot = len + __popcll(nm & lt);
                        for(;;)
                        {
                            if(atomicCAS(&hkey[h], EMPTY, c) == EMPTY)
                                break;
                            h = (h + 1) & hmask;
                        }
                        if constexpr(FILL)
                        {
                            hpos[h]    = (PosT)slot;
                            list[slot] = c;
                            acc[slot]  = sp_mul(va, vb); // first touch: the product itself (csr2m.cpp:489-496)
                        }
                    }
                    else if constexpr(FILL)
                    {
                        if(valid)
                            acc[pos] = sp_fma(va, vb, acc[pos]); // csr2m.cpp:498, contracted
                    }
                    len += __popcll(nm);
                }
                else
                {
                    // unsorted or repeated columns inside this chunk: one entry at a time, in order
                    const int nv = min(G, ke - k0);
                    for(int q = 0; q < nv; q++)
                    {
                        const int cq = __shfl(c, q, G);
                        T         vq = T(0);
                        if constexpr(FILL)
                            vq = spg_shfl(vb, q, G);
                        unsigned h   = hash(cq);
                        int      pos = -1;
                        for(;;)
                        {
                            const int key = spg_key<GLOBAL>(&hkey[h]);
                            if(key == EMPTY)
                                break;
                            if(key == cq)
                            {
                                pos = FILL ? (int)hpos[h] : 0;
                                break;
                            }
                            h = (h + 1) & hmask;
                        }
                        if(pos < 0 && len + 1 > limit)
                        {
                            dead = true;
                            break;
                        }
                        if(gl == 0)
                        {
                            if(pos < 0)
                            {
                                hkey[h] = cq;
                                if constexpr(FILL)
                                {
                                    hpos[h]   = (PosT)len;
                                    list[len] = cq;
                                    acc[len]  = sp_mul(va, vq);
                                }
                            }
                            else if constexpr(FILL)
                                acc[pos] = sp_fma(va, vq, acc[pos]);
                        }
                        len += pos < 0;
                        spg_sync<GLOBAL>();
                    }
                }
                spg_sync<GLOBAL>();
            }
        }
    }
    if constexpr(FILL)
        dead |= len != limit;
    if(dead && gl == 0)
        atomicOr(bad, 1u);
    if constexpr(FILL)
    {
        const int dst = ptr_c[i];
        for(int t = gl; t < len; t += G)
        {
            cnt_or_ind_c[dst + t] = list[t];
            val_c[dst + t]        = acc[t];
        }
    }
    else if(gl == 0)
        cnt_or_ind_c[i] = len;
}

// bins of the hash kernel: a row whose list holds at most `cap` entries runs with G lanes, 2^logh table slots and NG rows per workgroup
static constexpr int SPG_CAP[SPGEMM_BINS - 1] = {32, 256, 2048, 8192};

template <typename T>
aoclsparse_status launch_spgemm_bin(hipStream_t s, bool fill, int bin, aoclsparse_int nrows, const aoclsparse_int *rows, int base_a,
                                    const aoclsparse_int *ptr_a, const aoclsparse_int *ind_a, const T *val_a, int base_b,
                                    const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b, const T *val_b,
                                    const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a, bool conj_b,
                                    unsigned int *bad)
{
    if(nrows <= 0)
        return aoclsparse_status_success;
#define MI355_SPG(F, G, LOGH, NG)                                                                                               \
    hipLaunchKernelGGL((spgemm_hash_kernel<T, F, G, LOGH, NG, false>), dim3((unsigned)((nrows + NG - 1) / NG)), dim3(G * NG), 0, s, \
                       nrows, rows, (const SpgHeavy *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr, (T *)nullptr, base_a, \
                       ptr_a, ind_a, val_a, base_b, ptr_b, ind_b, val_b, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b, bad)
    if(fill)
    {
        switch(bin)
        {
        case 0: MI355_SPG(true, 16, 6, 16); break;
        case 1: MI355_SPG(true, 64, 9, 4); break;
        case 2: MI355_SPG(true, 64, 12, 1); break;
        default: return aoclsparse_status_internal_error; // (bin 3 exists in the count pass only)
        }
    }
    else
    {
        switch(bin)
        {
        case 0: MI355_SPG(false, 16, 6, 16); break;
        case 1: MI355_SPG(false, 64, 9, 4); break;
        case 2: MI355_SPG(false, 64, 12, 1); break;
        case 3: MI355_SPG(false, 64, 14, 1); break;
        default: return aoclsparse_status_internal_error;
        }
    }
#undef MI355_SPG
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// the rows above the largest LDS bin: tables in the global slab (g_key / g_pos: table slots, g_list / g_acc: list entries)
template <typename T>
aoclsparse_status launch_spgemm_heavy(hipStream_t s, bool fill, aoclsparse_int nrows, const SpgHeavy *heavy, int *g_key, int *g_pos,
                                      int *g_list, T *g_acc, int base_a, const aoclsparse_int *ptr_a, const aoclsparse_int *ind_a,
                                      const T *val_a, int base_b, const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b,
                                      const T *val_b, const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c,
                                      bool conj_a, bool conj_b, unsigned int *bad)
{
    if(nrows <= 0)
        return aoclsparse_status_success;
    if(fill)
        hipLaunchKernelGGL((spgemm_hash_kernel<T, true, 64, 0, 1, true>), dim3((unsigned)nrows), dim3(64), 0, s, nrows,
                           (const aoclsparse_int *)nullptr, heavy, g_key, g_pos, g_list, g_acc, base_a, ptr_a, ind_a, val_a, base_b,
                           ptr_b, ind_b, val_b, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b, bad);
    else
        hipLaunchKernelGGL((spgemm_hash_kernel<T, false, 64, 0, 1, true>), dim3((unsigned)nrows), dim3(64), 0, s, nrows,
                           (const aoclsparse_int *)nullptr, heavy, g_key, g_pos, g_list, g_acc, base_a, ptr_a, ind_a, val_a, base_b,
                           ptr_b, ind_b, val_b, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b, bad);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// ---- the analysis around the two passes, on the device (round 4) ---------------------------------------------------------------
// Upper bounds, bins, the rows of every bin and the prefix sum of the counts are integer work over arrays that already sit in HBM;
// on the host they cost 3 of the 8 ms of A * A on the 1000^2 Laplacian (thread start-up and memory traffic of 1 M-row loops).
__device__ __forceinline__ int spg_bin_dev(int entries, bool fill)
{
    const int last = fill ? SPGEMM_BINS - 2 : SPGEMM_BINS - 1;
    int       b    = SPGEMM_BINS - 1;
    for(int k = last - 1; k >= 0; k--)
        if(entries <= SPG_CAP[k])
            b = k;
    return b;
}

// cap[i] = min(sum of the lengths of the B rows that row i of A touches, n)
__global__ __launch_bounds__(256) void spg_bound_kernel(aoclsparse_int m, aoclsparse_int n, int base_a,
                                                        const aoclsparse_int *__restrict__ ptr_a,
                                                        const aoclsparse_int *__restrict__ ind_a, const aoclsparse_int *__restrict__ ptr_b,
                                                        int *__restrict__ cap)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= m)
        return;
    long long u = 0;
    for(int p = ptr_a[i] - base_a; p < ptr_a[i + 1] - base_a; p++)
    {
        const int c = ind_a[p] - base_a;
        u += ptr_b[c + 1] - ptr_b[c];
    }
    cap[i] = (int)(u < (long long)n ? u : (long long)n);
}

// key[i] = ptr[i + 1] - ptr[i] (the fill pass of a two-stage call starts from the caller's row_ptr)
__global__ __launch_bounds__(256) void spg_diff_kernel(aoclsparse_int m, const aoclsparse_int *__restrict__ ptr, int *__restrict__ key)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i < m)
        key[i] = ptr[i + 1] - ptr[i];
}

// hist[b] += rows of bin b; hist[SPGEMM_BINS] != 0: some key is negative or above its limit (a row_ptr that is not this product's).
// A fixed grid walks the rows with a stride, every thread counts in registers, one atomic per bin and WORKGROUP at the end (an atomic
// per wavefront on the same five words was 180 us for 1 M rows: rocprofv3, profiles/r4/sp2m_kernel_stats_*.csv).
constexpr int SPG_HIST_BLOCKS = 512;
__global__ __launch_bounds__(256) void spg_hist_kernel(aoclsparse_int m, const int *__restrict__ key, const int *__restrict__ limit,
                                                       bool fill, unsigned int *hist)
{
    __shared__ unsigned int sh[SPGEMM_BINS + 1];
    if(threadIdx.x <= SPGEMM_BINS)
        sh[threadIdx.x] = 0;
    __syncthreads();
    unsigned int mine[SPGEMM_BINS] = {};
    bool         bad               = false;
    for(long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long long)gridDim.x * 256)
    {
        const int k = key[i];
        bad |= k < 0 || (limit && k > limit[i]);
        const int b = spg_bin_dev(k < 0 ? 0 : k, fill);
#pragma unroll
        for(int q = 0; q < SPGEMM_BINS; q++)
            mine[q] += b == q;
    }
#pragma unroll
    for(int q = 0; q < SPGEMM_BINS; q++)
    {
        unsigned int v = mine[q];
        for(int o = 32; o > 0; o >>= 1)
            v += __shfl_down(v, o, 64);
        if((threadIdx.x & 63) == 0 && v)
            atomicAdd(&sh[q], v);
    }
    if(__ballot(bad) && (threadIdx.x & 63) == 0)
        atomicOr(&sh[SPGEMM_BINS], 1u);
    __syncthreads();
    if(threadIdx.x < SPGEMM_BINS && sh[threadIdx.x])
        atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
    if(threadIdx.x == SPGEMM_BINS && sh[SPGEMM_BINS])
        atomicOr(&hist[SPGEMM_BINS], 1u);
}

struct SpgBounds
{
    int at[SPGEMM_BINS + 1];
};

// order[bounds[b] ...] = the rows of bin b; the rows of one workgroup keep their order (ballot rank inside a wavefront, wavefronts in
// order inside the workgroup), workgroups arrive as they come: one atomic per bin and workgroup
__global__ __launch_bounds__(256) void spg_order_kernel(aoclsparse_int m, const int *__restrict__ key, bool fill, SpgBounds bounds,
                                                        unsigned int *cursor, aoclsparse_int *__restrict__ order)
{
    __shared__ unsigned int cnt[4][SPGEMM_BINS], base_of[SPGEMM_BINS];
    const int i    = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b    = i < m ? spg_bin_dev(key[i], fill) : -1;
    unsigned int rank = 0;
#pragma unroll
    for(int k = 0; k < SPGEMM_BINS; k++)
    {
        const unsigned long long mk = __ballot(b == k);
        if(lane == 0)
            cnt[w][k] = (unsigned)__popcll(mk);
        if(b == k)
            rank = (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if(threadIdx.x < SPGEMM_BINS)
    {
        const unsigned total = cnt[0][threadIdx.x] + cnt[1][threadIdx.x] + cnt[2][threadIdx.x] + cnt[3][threadIdx.x];
        base_of[threadIdx.x] = total ? atomicAdd(&cursor[threadIdx.x], total) : 0u;
    }
    __syncthreads();
    if(b >= 0)
    {
        unsigned at = base_of[b] + rank;
        for(int u = 0; u < w; u++)
            at += cnt[u][b];
        order[bounds.at[b] + at] = i;
    }
}

__global__ __launch_bounds__(256) void spg_gather_kernel(aoclsparse_int count, const aoclsparse_int *__restrict__ ids,
                                                         const int *__restrict__ key, int *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if(j < count)
        out[j] = key[ids[j]];
}

// exclusive prefix sum of cnt[0..m) into ptr[0..m], 64-bit block sums (ptr is 32-bit like the reference's row_ptr; the true total
// goes to total[0] so that the caller can refuse a product of more than 2^31 - 1 entries: csr2m.cpp:221-236)
constexpr int SPG_SCAN_BLOCK = 1024;
__global__ __launch_bounds__(256) void spg_scan_sums_kernel(aoclsparse_int m, const int *__restrict__ cnt, long long *__restrict__ sums)
{
    __shared__ long long sh[4];
    const int            i0 = blockIdx.x * SPG_SCAN_BLOCK + threadIdx.x * 4;
    long long            v  = 0;
    for(int q = 0; q < 4; q++)
        if(i0 + q < m)
            v += cnt[i0 + q];
    for(int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o, 64);
    if((threadIdx.x & 63) == 0)
        sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if(threadIdx.x == 0)
        sums[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void spg_scan_blocks_kernel(int nblocks, long long *sums, long long *total)
{
    __shared__ long long sh[256];
    long long            carry = 0;
    for(int b0 = 0; b0 < nblocks; b0 += 256)
    {
        const int       b = b0 + threadIdx.x;
        const long long v = b < nblocks ? sums[b] : 0;
        sh[threadIdx.x]   = v;
        __syncthreads();
        for(int o = 1; o < 256; o <<= 1) // Hillis-Steele, inclusive
        {
            const long long t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if(b < nblocks)
            sums[b] = carry + sh[threadIdx.x] - v; // exclusive
        carry += sh[255];
        __syncthreads();
    }
    if(threadIdx.x == 0)
        total[0] = carry;
}

__global__ __launch_bounds__(256) void spg_scan_apply_kernel(aoclsparse_int m, const int *__restrict__ cnt, const long long *__restrict__ sums,
                                                             const long long *__restrict__ total, aoclsparse_int *__restrict__ ptr)
{
    __shared__ long long sh[256];
    const int            i0 = blockIdx.x * SPG_SCAN_BLOCK + threadIdx.x * 4;
    int                  c[4];
    long long            v = 0;
    for(int q = 0; q < 4; q++)
    {
        c[q] = i0 + q < m ? cnt[i0 + q] : 0;
        v += c[q];
    }
    sh[threadIdx.x] = v;
    __syncthreads();
    for(int o = 1; o < 256; o <<= 1)
    {
        const long long t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    long long run = sums[blockIdx.x] + sh[threadIdx.x] - v;
    for(int q = 0; q < 4; q++)
    {
        if(i0 + q < m)
            ptr[i0 + q] = (aoclsparse_int)run;
        run += c[q];
    }
    if(blockIdx.x == 0 && threadIdx.x == 0)
        ptr[m] = (aoclsparse_int)total[0];
}

aoclsparse_status launch_spg_bounds(hipStream_t s, aoclsparse_int m, aoclsparse_int n, int base_a, const aoclsparse_int *ptr_a,
                                    const aoclsparse_int *ind_a, const aoclsparse_int *ptr_b, int *cap)
{
    if(m > 0)
        hipLaunchKernelGGL(spg_bound_kernel, dim3((m + 255) / 256), dim3(256), 0, s, m, n, base_a, ptr_a, ind_a, ptr_b, cap);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

aoclsparse_status launch_spg_diff(hipStream_t s, aoclsparse_int m, const aoclsparse_int *ptr, int *key)
{
    if(m > 0)
        hipLaunchKernelGGL(spg_diff_kernel, dim3((m + 255) / 256), dim3(256), 0, s, m, ptr, key);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// hist: SPGEMM_BINS + 1 words, zeroed here
aoclsparse_status launch_spg_hist(hipStream_t s, aoclsparse_int m, const int *key, const int *limit, bool fill, unsigned int *hist)
{
    MI355_HIP_TRY(hipMemsetAsync(hist, 0, sizeof(unsigned int) * (SPGEMM_BINS + 1), s));
    if(m > 0)
        hipLaunchKernelGGL(spg_hist_kernel, dim3((unsigned)std::min<long long>(SPG_HIST_BLOCKS, ((long long)m + 255) / 256)), dim3(256), 0, s,
                           m, key, limit, fill, hist);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// cursor: SPGEMM_BINS words, zeroed here; bounds[b] = first position of bin b in order
aoclsparse_status launch_spg_order(hipStream_t s, aoclsparse_int m, const int *key, bool fill, const aoclsparse_int *bounds,
                                   unsigned int *cursor, aoclsparse_int *order)
{
    SpgBounds bd;
    for(int b = 0; b <= SPGEMM_BINS; b++)
        bd.at[b] = bounds[b];
    MI355_HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(unsigned int) * SPGEMM_BINS, s));
    if(m > 0)
        hipLaunchKernelGGL(spg_order_kernel, dim3((m + 255) / 256), dim3(256), 0, s, m, key, fill, bd, cursor, order);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

aoclsparse_status launch_spg_gather(hipStream_t s, aoclsparse_int count, const aoclsparse_int *ids, const int *key, int *out)
{
    if(count > 0)
        hipLaunchKernelGGL(spg_gather_kernel, dim3((count + 255) / 256), dim3(256), 0, s, count, ids, key, out);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// sums: ceil(m / 1024) + 1 long longs of scratch (the last one receives the total)
size_t spg_scan_scratch_bytes(aoclsparse_int m)
{
    return sizeof(long long) * ((size_t)(m + SPG_SCAN_BLOCK - 1) / SPG_SCAN_BLOCK + 2);
}

aoclsparse_status launch_spg_scan(hipStream_t s, aoclsparse_int m, const int *cnt, aoclsparse_int *ptr, long long *scratch,
                                  long long **total_dev)
{
    const int nb = (int)((m + SPG_SCAN_BLOCK - 1) / SPG_SCAN_BLOCK);
    long long *total = scratch + nb;
    *total_dev       = total;
    if(m <= 0)
    {
        MI355_HIP_TRY(hipMemsetAsync(total, 0, sizeof(long long), s));
        MI355_HIP_TRY(hipMemsetAsync(ptr, 0, sizeof(aoclsparse_int), s));
        return aoclsparse_status_success;
    }
    hipLaunchKernelGGL(spg_scan_sums_kernel, dim3(nb), dim3(256), 0, s, m, cnt, scratch);
    hipLaunchKernelGGL(spg_scan_blocks_kernel, dim3(1), dim3(256), 0, s, nb, scratch, total);
    hipLaunchKernelGGL(spg_scan_apply_kernel, dim3(nb), dim3(256), 0, s, m, cnt, scratch, total, ptr);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

int spgemm_bin_of(long long entries, bool fill)
{
    for(int b = 0; b < (fill ? SPGEMM_BINS - 2 : SPGEMM_BINS - 1); b++)
        if(entries <= SPG_CAP[b])
            return b;
    return SPGEMM_BINS - 1; // tables in the global slab
}

#define MI355_SPGEMM_INST(T)                                                                                        \
    template aoclsparse_status launch_spgemm_bin<T>(hipStream_t, bool, int, aoclsparse_int, const aoclsparse_int *, int, \
                                                    const aoclsparse_int *, const aoclsparse_int *, const T *, int, \
                                                    const aoclsparse_int *, const aoclsparse_int *, const T *,      \
                                                    const aoclsparse_int *, aoclsparse_int *, T *, bool, bool,      \
                                                    unsigned int *);                                                \
    template aoclsparse_status launch_spgemm_heavy<T>(hipStream_t, bool, aoclsparse_int, const SpgHeavy *, int *, int *, int *, T *, \
                                                      int, const aoclsparse_int *, const aoclsparse_int *, const T *, int,          \
                                                      const aoclsparse_int *, const aoclsparse_int *, const T *,                    \
                                                      const aoclsparse_int *, aoclsparse_int *, T *, bool, bool, unsigned int *);
MI355_SPGEMM_INST(double)
MI355_SPGEMM_INST(float)
MI355_SPGEMM_INST(cdouble)
MI355_SPGEMM_INST(cfloat)

} // namespace mi355
