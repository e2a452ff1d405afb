// spgemm_kernels.hip -- C = A * B for two CSR matrices (the sp2m / spmm / csr2m path), gfx950.
//
// Reference: two-stage Gustavson with a dense per-thread accumulator of length n
// (level3/aoclsparse_csr2m.cpp:46-302 count, :310-543 finalize).  Row i of C lists its columns in
// FIRST-TOUCH order (walk row i of A left to right, for each entry walk the matching row of B left to
// right) and each value is accumulated in that same order: first product stored, later ones added with
// a contracted multiply-add (:489-498).  Both properties are reproduced exactly here:
//
//   one WAVEFRONT owns one row of C and consumes the products strictly in the reference's order; the 64
//   lanes only parallelise the membership test "is column c already in this row's list?" (one compare
//   per lane, one ballot per 64 list entries).  The list (and, in the fill pass, the partial sums) lives
//   in LDS when the row's upper bound (sum of the B-row lengths it touches) is <= SPGEMM_LDS_CAP, else
//   in a global scratch slab of exactly that upper bound.
//
// Stage 1 returns the per-row counts; the host prefix-sums them (64-bit, as the reference) and
// allocates C; stage 2 writes col_ind / val of C at row_ptr_C.  Integer output is bit-exact, fp output
// bit-identical to the reference's single-thread order.
//
// Bound: neither HBM nor MFMA -- irregular, latency/instruction bound; algorithmic traffic is
// 12 B per entry of A and of the touched B rows + 12 B per entry of C.
//
// Round 4: the rows are binned by the size of their list and served by spgemm_hash_kernel below -- a GROUP of 16 or 64 lanes
// per row, the products of one B row taken 16 / 64 at a time (they are distinct columns when the row is strictly increasing,
// which is checked on the fly: then the lanes look their columns up in an LDS hash table together, new columns get their list
// slots by ballot rank = first-touch order, and every column's multiply-adds still happen in the order of A's row).  The
// one-product-at-a-time kernel above stays for rows whose list does not fit the largest LDS bin.
#include "internal.hpp"

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

// position of c in list[0..len) or -1; wave-uniform result
__device__ __forceinline__ int wave_find(const int *list, int len, int c, int lane)
{
    for(int b0 = 0; b0 < len; b0 += 64)
    {
        const int                idx  = b0 + lane;
        const bool               hit  = idx < len && list[idx] == c;
        const unsigned long long mask = __ballot(hit);
        if(mask)
            return b0 + __ffsll((long long)mask) - 1;
    }
    return -1;
}

template <typename T, bool FILL>
__global__ __launch_bounds__(256) void spgemm_row_kernel(aoclsparse_int nrows, const aoclsparse_int *__restrict__ rows, int base_a,
                                                         const aoclsparse_int *__restrict__ ptr_a,
                                                         const aoclsparse_int *__restrict__ ind_a,
                                                         const T *__restrict__ val_a, int base_b,
                                                         const aoclsparse_int *__restrict__ ptr_b,
                                                         const aoclsparse_int *__restrict__ ind_b,
                                                         const T *__restrict__ val_b,
                                                         const long long *__restrict__ slab_off, int *slab_idx,
                                                         T *slab_val, const aoclsparse_int *__restrict__ ptr_c,
                                                         aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a, bool conj_b)
{
    constexpr int  SPGEMM_LDS_CAP = spgemm_lds_cap<T>();
    __shared__ int s_idx[4][SPGEMM_LDS_CAP];
    __shared__ T   s_val[FILL ? 4 : 1][FILL ? SPGEMM_LDS_CAP : 1];
    const int      w    = threadIdx.x >> 6;
    const int      lane = threadIdx.x & 63;
    const int      g    = blockIdx.x * 4 + w;
    if(g >= nrows)
        return;
    const int       i   = rows ? rows[g] : g; // (round 4: only the rows whose list exceeds the largest bin of the hash kernel come here)
    const long long off = slab_off[i];
    const long long ub  = slab_off[i + 1] - off;
    const bool      in_lds = ub <= SPGEMM_LDS_CAP;
    int            *list = in_lds ? s_idx[w] : slab_idx + off;
    T              *acc  = nullptr;
    if constexpr(FILL)
        acc = in_lds ? s_val[w] : slab_val + off;
    int len = 0;
    for(int j = ptr_a[i] - base_a; j < ptr_a[i + 1] - base_a; j++)
    {
        const int ca = ind_a[j] - base_a;
        T         va = T(0);
        if constexpr(FILL)
            va = sp_conj(val_a[j], conj_a);
        for(int k = ptr_b[ca] - base_b; k < ptr_b[ca + 1] - base_b; k++)
        {
            const int c   = ind_b[k] - base_b;
            const int pos = wave_find(list, len, c, lane);
            if(pos < 0)
            {
                if(lane == 0)
                {
                    list[len] = c; // first touch: new entry of C (csr2m.cpp:489-496)
                    if constexpr(FILL)
                        acc[len] = sp_mul(va, sp_conj(val_b[k], conj_b));
                }
                len++;
            }
            else if constexpr(FILL)
            {
                if(lane == 0)
                    acc[pos] = sp_fma(va, sp_conj(val_b[k], conj_b), acc[pos]); // csr2m.cpp:498, contracted
            }
            if(!in_lds)
                __threadfence_block(); // lane 0's global store must be visible to the next compare
        }
    }
    if constexpr(FILL)
    {
        const int dst = ptr_c[i];
        for(int t = lane; t < len; t += 64)
        {
            cnt_or_ind_c[dst + t] = list[t];
            val_c[dst + t]        = acc[t];
        }
    }
    else if(lane == 0)
        cnt_or_ind_c[i] = len;
}

// ---- round 4: hash-table kernel ------------------------------------------------------------------------------------------------
// A group of G lanes owns row rows[g] of C.  LDS per group: hkey[H] (open addressing, linear probing, -1 = empty) and, in the fill
// pass, hpos[H] (list slot of the key), list[H / 2] (columns in first-touch order) and acc[H / 2].  The caller guarantees that the
// row's list has at most H / 2 entries (count pass: its upper bound; fill pass: its exact count).
//
// Order: A's row is walked left to right; the entries of one B row are taken G at a time, lane t = t-th entry.  When the chunk is
// strictly increasing (with the last column of the previous chunk) its columns are distinct: lookups in parallel, then the new
// columns are appended in lane order -- exactly the order the reference's serial walk first touches them -- and each column's
// value is updated once.  Any other chunk (unsorted rows, repeated columns: both legal, csr_util.cpp:244) is walked one entry at
// a time.  Either way every column sees its products in the reference's order: first product stored, the later ones added with a
// contracted multiply-add (csr2m.cpp:489-498).
template <int G>
__device__ __forceinline__ unsigned long long spg_group_bits(unsigned long long wave_mask, int grp)
{
    if constexpr(G == 64)
        return wave_mask;
    else
        return (wave_mask >> (G * grp)) & ((1ull << G) - 1ull);
}

template <typename T, bool FILL, int G, int LOGH, int NG>
__global__ __launch_bounds__(G *NG) void spgemm_hash_kernel(aoclsparse_int nrows, const aoclsparse_int *__restrict__ rows, int base_a,
                                                           const aoclsparse_int *__restrict__ ptr_a,
                                                           const aoclsparse_int *__restrict__ ind_a,
                                                           const T *__restrict__ val_a, int base_b,
                                                           const aoclsparse_int *__restrict__ ptr_b,
                                                           const aoclsparse_int *__restrict__ ind_b,
                                                           const T *__restrict__ val_b,
                                                           const aoclsparse_int *__restrict__ ptr_c,
                                                           aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a, bool conj_b)
{
    constexpr int      H = 1 << LOGH, CAP = H / 2, EMPTY = -1;
    __shared__ int     s_key[NG][H];
    __shared__ unsigned short s_pos[FILL ? NG : 1][FILL ? H : 1];
    __shared__ int     s_list[FILL ? NG : 1][FILL ? CAP : 1];
    __shared__ T       s_acc[FILL ? NG : 1][FILL ? CAP : 1];
    const int          grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int          wgrp = (threadIdx.x & 63) / G; // group inside its wavefront (ballots are per wavefront)
    const long long    g   = (long long)blockIdx.x * NG + grp;
    if(g >= nrows)
        return;
    const int i = rows ? rows[g] : (int)g; // (no list: every row of the matrix is in this bin)
    int      *hkey = s_key[grp];
    unsigned short *hpos = s_pos[FILL ? grp : 0];
    int      *list = s_list[FILL ? grp : 0];
    T        *acc  = s_acc[FILL ? grp : 0];
    for(int t = gl; t < H; t += G)
        hkey[t] = EMPTY;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const unsigned long long lt = (1ull << gl) - 1ull;
    auto hash = [](int c) { return (unsigned)c * 2654435761u >> (32 - LOGH); };
    int  len = 0;
    const int ja = ptr_a[i] - base_a, je = ptr_a[i + 1] - base_a;
    for(int j0 = ja; j0 < je; j0 += G)
    {
        // this round's entries of A's row, one per lane: column, value, the extent of the matching B row
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
        for(int jj = 0; jj < nj; jj++)
        {
            const int kb = __shfl(my_kb, jj, G), ke = __shfl(my_ke, jj, G);
            T         va = T(0);
            if constexpr(FILL)
                va = spg_shfl(my_va, jj, G);
            int carry = -1;
            for(int k0 = kb; k0 < ke; k0 += G)
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
                            const int key = hkey[h];
                            if(key == EMPTY)
                                break;
                            if(key == c)
                            {
                                pos = FILL ? (int)hpos[h] : 0;
                                break;
                            }
                            h = (h + 1) & (H - 1);
                        }
                    const bool               isnew = valid && pos < 0;
                    const unsigned long long nm    = spg_group_bits<G>(__ballot(isnew), wgrp);
                    if(isnew)
                    {
                        const int slot = len + __popcll(nm & lt);
                        for(;;)
                        {
                            if(atomicCAS(&hkey[h], EMPTY, c) == EMPTY)
                                break;
                            h = (h + 1) & (H - 1);
                        }
                        if constexpr(FILL)
                        {
                            hpos[h]    = (unsigned short)slot;
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
                            const int key = hkey[h];
                            if(key == EMPTY)
                                break;
                            if(key == cq)
                            {
                                pos = FILL ? (int)hpos[h] : 0;
                                break;
                            }
                            h = (h + 1) & (H - 1);
                        }
                        if(gl == 0)
                        {
                            if(pos < 0)
                            {
                                hkey[h] = cq;
                                if constexpr(FILL)
                                {
                                    hpos[h]   = (unsigned short)len;
                                    list[len] = cq;
                                    acc[len]  = sp_mul(va, vq);
                                }
                            }
                            else if constexpr(FILL)
                                acc[pos] = sp_fma(va, vq, acc[pos]);
                        }
                        len += pos < 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
    }
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
                                    const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a, bool conj_b)
{
    if(nrows <= 0)
        return aoclsparse_status_success;
#define MI355_SPG(F, G, LOGH, NG)                                                                                               \
    hipLaunchKernelGGL((spgemm_hash_kernel<T, F, G, LOGH, NG>), dim3((unsigned)((nrows + NG - 1) / NG)), dim3(G * NG), 0, s, nrows, \
                       rows, base_a, ptr_a, ind_a, val_a, base_b, ptr_b, ind_b, val_b, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b)
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

int spgemm_bin_of(long long entries, bool fill)
{
    for(int b = 0; b < (fill ? SPGEMM_BINS - 2 : SPGEMM_BINS - 1); b++)
        if(entries <= SPG_CAP[b])
            return b;
    return SPGEMM_BINS - 1; // the one-product-at-a-time kernel
}

template <typename T>
aoclsparse_status launch_spgemm(hipStream_t s, bool fill, aoclsparse_int nrows, const aoclsparse_int *rows, int base_a,
                                const aoclsparse_int *ptr_a, const aoclsparse_int *ind_a, const T *val_a,
                                int base_b, const aoclsparse_int *ptr_b, const aoclsparse_int *ind_b,
                                const T *val_b, const long long *slab_off, int *slab_idx, T *slab_val,
                                const aoclsparse_int *ptr_c, aoclsparse_int *cnt_or_ind_c, T *val_c, bool conj_a,
                                bool conj_b)
{
    if(nrows <= 0)
        return aoclsparse_status_success;
    dim3 grid((nrows + 3) / 4), block(256);
    if(fill)
        hipLaunchKernelGGL((spgemm_row_kernel<T, true>), grid, block, 0, s, nrows, rows, base_a, ptr_a, ind_a, val_a, base_b,
                           ptr_b, ind_b, val_b, slab_off, slab_idx, slab_val, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b);
    else
        hipLaunchKernelGGL((spgemm_row_kernel<T, false>), grid, block, 0, s, nrows, rows, base_a, ptr_a, ind_a, val_a, base_b,
                           ptr_b, ind_b, val_b, slab_off, slab_idx, slab_val, ptr_c, cnt_or_ind_c, val_c, conj_a, conj_b);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_SPGEMM_INST(T)                                                                                        \
    template aoclsparse_status launch_spgemm<T>(hipStream_t, bool, aoclsparse_int, const aoclsparse_int *, int,     \
                                                const aoclsparse_int *, const aoclsparse_int *, const T *, int,     \
                                                const aoclsparse_int *, const aoclsparse_int *, const T *,          \
                                                const long long *, int *, T *, const aoclsparse_int *,              \
                                                aoclsparse_int *, T *, bool, bool);                                 \
    template aoclsparse_status launch_spgemm_bin<T>(hipStream_t, bool, int, aoclsparse_int, const aoclsparse_int *, int, \
                                                    const aoclsparse_int *, const aoclsparse_int *, const T *, int, \
                                                    const aoclsparse_int *, const aoclsparse_int *, const T *,      \
                                                    const aoclsparse_int *, aoclsparse_int *, T *, bool, bool);
MI355_SPGEMM_INST(double)
MI355_SPGEMM_INST(float)
MI355_SPGEMM_INST(cdouble)
MI355_SPGEMM_INST(cfloat)

} // namespace mi355
