// csrmm_kernels.hip -- C = alpha*A*B + beta*C, A sparse CSR (m x k), B/C dense, for gfx950.
//
// Arithmetic per output element follows the reference's column-major kernel
// (level3/aoclsparse_csrmm.hpp:69-85): sum = fma(a_ik, B_kj, sum) over the row in CSR order, then
// C = fma(beta, C, alpha*sum).  Every reference csrmm kernel reads C even when beta == 0 (SURVEY.md
// Appendix B), which costs a third of the traffic at 256 columns.  Here, for beta == 0, C is read only
// where alpha*sum is an exact zero (the one case where fma(0, C, z) != z for finite C: the sign of the
// zero): bit-identical to the reference for every FINITE C, while a NaN/Inf already sitting in C is
// overwritten instead of propagated.  Round 3: that is the OPT-IN mode (aoclsparse_mi355_set_csrmm_beta0_overwrite(1) or
// AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=1); by default C is read for beta == 0 too, as every reference kernel does
// (csrmm.hpp:83,129; csrmm_kt.cpp:176-191,246), so NaN / Inf in C propagate exactly as there.
//
// HBM-bound (AI ~ 0.6 flop/B at 256 columns, 5 nnz/row): no MFMA -- the dense tiles a 5-point
// stencil would give an MFMA are >90 % zeros, so reshaping to GEMM only adds traffic.
//   row-major  : one lane owns 2 adjacent columns of one C row (16-B loads/stores); the B rows a
//                sparse row touches are contiguous 8*n-byte streams; val/col are wave-uniform loads.
//   column-major: one lane owns one row, keeps its first 8 entries in registers and sweeps 64
//                columns, so A is read n/64 times and every B / C access is coalesced across rows.
// Algorithmic bytes: (m+1+nnz)*4 + nnz*8 + 8*n*(k + m*(1+[beta!=0]))  (BASELINE.md section 2).
#include "internal.hpp"
#include "mm_order.hpp"
#include "kt_order.hpp"

#include <hip/hip_runtime.h>

#include <climits>
#include <cstddef>

#include <type_traits>

namespace mi355
{

int &mm_direction_word()
{
    static thread_local int word = 0;
    return word;
}


template <typename T>
struct vec2;
template <>
struct vec2<double>
{
    using type = double2;
};
template <>
struct vec2<float>
{
    using type = float2;
};

__device__ __forceinline__ double mm_fma(double a, double b, double c)
{
    return fma(a, b, c);
}
__device__ __forceinline__ float mm_fma(float a, float b, float c)
{
    return fmaf(a, b, c);
}

// blockDim = (TX, TY): TX lanes span 2*TX columns, TY rows per workgroup
template <typename T, bool VEC2>
__global__ __launch_bounds__(256) void csrmm_row_kernel(int base, T alpha, aoclsparse_int m,
                                                        const T *__restrict__ val,
                                                        const aoclsparse_int *__restrict__ col,
                                                        const aoclsparse_int *__restrict__ row_ptr,
                                                        const T *__restrict__ B, aoclsparse_int n,
                                                        aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                        aoclsparse_int ldc, bool readc, int xcd_chunk)
{
    using V     = typename vec2<T>::type;
    const int bx = mm_block_index(xcd_chunk);
    const int i  = bx * blockDim.y + threadIdx.y; // rows on grid.x (no 65535 limit)
    if(i >= m)
        return;
    const int s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
    if constexpr(VEC2)
    {
        const int j = 2 * (blockIdx.y * blockDim.x + threadIdx.x);
        if(j >= n)
            return;
        T a0 = T(0), a1 = T(0);
        for(int p = s; p < e; p++)
        {
            const T a = val[p];
            const V b = *reinterpret_cast<const V *>(B + (size_t)(col[p] - base) * ldb + j);
            a0        = mm_fma(a, b.x, a0);
            a1        = mm_fma(a, b.y, a1);
        }
        V      *cp = reinterpret_cast<V *>(C + (size_t)i * ldc + j);
        const T z0 = alpha * a0, z1 = alpha * a1;
        V       c;
        if(readc || z0 == T(0) || z1 == T(0))
        {
            c   = *cp;
            c.x = mm_fma(beta, c.x, z0);
            c.y = mm_fma(beta, c.y, z1);
        }
        else
            c.x = z0, c.y = z1;
        *cp = c;
    }
    else
    {
        const int j = blockIdx.y * blockDim.x + threadIdx.x;
        if(j >= n)
            return;
        T acc = T(0);
        for(int p = s; p < e; p++)
            acc = mm_fma(val[p], B[(size_t)(col[p] - base) * ldb + j], acc);
        T      *cp = C + (size_t)i * ldc + j;
        const T z  = alpha * acc;
        *cp        = (readc || z == T(0)) ? mm_fma(beta, *cp, z) : z;
    }
}

// row-major, n >= 128: one WAVEFRONT per (row, 128-column chunk).  The row index is wave-uniform
// (readfirstlane), so row_ptr / val / col_ind come through the scalar cache and the vector memory
// pipe carries only the B rows (16 B per lane) and the C row.  Four non-zeros are issued per step.
template <typename T>
__global__ __launch_bounds__(256) void csrmm_row_wave_kernel(int base, T alpha, aoclsparse_int m,
                                                             const T *__restrict__ val,
                                                             const aoclsparse_int *__restrict__ col,
                                                             const aoclsparse_int *__restrict__ row_ptr,
                                                             const T *__restrict__ B, aoclsparse_int n,
                                                             aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                             aoclsparse_int ldc, bool readc, int xcd_chunk)
{
    using V       = typename vec2<T>::type;
    const int w   = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-contiguous row order: workgroups with equal blockIdx%8 share an XCD / L2; giving each XCD one
    // contiguous eighth of the rows lets the B rows a row shares with its neighbours (i+-1, i+-g) be L2
    // hits instead of fabric reads by up to five different XCDs
    const int bx  = mm_block_index(xcd_chunk);
    const int i   = bx * 4 + w;
    const int j   = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(i >= m || j >= n)
        return;
    col -= base, val -= base; // the index base, folded into the pointers once
    const int s = row_ptr[i], e = row_ptr[i + 1];
    T         a0 = T(0), a1 = T(0);
    const T  *Bj = B + j - (ptrdiff_t)base * ldb;
    int       p  = s;
    for(; p + 4 <= e; p += 4)
    {
        const T v0 = val[p], v1 = val[p + 1], v2 = val[p + 2], v3 = val[p + 3];
        const V b0 = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)col[p] * ldb);
        const V b1 = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)col[p + 1] * ldb);
        const V b2 = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)col[p + 2] * ldb);
        const V b3 = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)col[p + 3] * ldb);
        a0 = mm_fma(v0, b0.x, a0), a1 = mm_fma(v0, b0.y, a1);
        a0 = mm_fma(v1, b1.x, a0), a1 = mm_fma(v1, b1.y, a1);
        a0 = mm_fma(v2, b2.x, a0), a1 = mm_fma(v2, b2.y, a1);
        a0 = mm_fma(v3, b3.x, a0), a1 = mm_fma(v3, b3.y, a1);
    }
    for(; p < e; p++)
    {
        const T v0 = val[p];
        const V b0 = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)col[p] * ldb);
        a0 = mm_fma(v0, b0.x, a0), a1 = mm_fma(v0, b0.y, a1);
    }
    V      *cp = reinterpret_cast<V *>(C + (size_t)i * ldc + j);
    const T z0 = alpha * a0, z1 = alpha * a1;
    const bool need = readc || z0 == T(0) || z1 == T(0); // (see csrmm_rowgroup2_kernel)
    typedef T  nt2 __attribute__((ext_vector_type(2)));
    nt2        o;
    o.x = z0, o.y = z1;
    if(__builtin_amdgcn_ballot_w64(need) != 0)
    {
        const V c = *cp;
        o.x       = need ? mm_fma(beta, c.x, z0) : z0;
        o.y       = need ? mm_fma(beta, c.y, z1) : z1;
    }
    __builtin_nontemporal_store(o, reinterpret_cast<nt2 *>(cp));
}

// The same with C READ (beta != 0, or the default beta == 0 mode that multiplies C by zero as the reference does), written for
// the wave's lifetime (round 3): the C row is requested FIRST, a row of <= 8 entries is one batch of exactly `len` B-row loads
// (wave-uniform switch), longer rows walk in steps of 8.  The kernel above spends five dependent round trips on a 5-entry row
// (row pointers -> values / columns -> four B rows -> the fifth -> C); this one three.  Same chain per element: same bits.
// KT: the arithmetic of the reference's csrmm_row_kt (csrmm_kt.cpp:244-356) for column counts that are a multiple of its vector
// width: c = c * beta first, then c = fma(alpha * a_k, b_kj, c) entry by entry -- what aoclsparse_?csrmm_kid 1/2/3 asks for.
// kt_tail: the last kt_tail columns (n modulo the vector width) take csrmm_row_kt's scalar statement "C += sv * B * alpha" =
// fma(sv * b, alpha, c) (csrmm_kt.cpp:335-356) instead of fma(alpha * sv, b, c): a per-lane choice between two roundings.
template <typename T, bool KT = false>
__global__ __launch_bounds__(256) void csrmm_row_wave_rc_kernel(int base, T alpha, aoclsparse_int m,
                                                                const T *__restrict__ val,
                                                                const aoclsparse_int *__restrict__ col,
                                                                const aoclsparse_int *__restrict__ row_ptr,
                                                                const T *__restrict__ B, aoclsparse_int n,
                                                                aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                                aoclsparse_int ldc, int xcd_chunk, int kt_tail = 0)
{
    using V      = typename vec2<T>::type;
    const int w  = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int bx = mm_block_index(xcd_chunk);
    const int i  = bx * 4 + w;
    const int j  = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(i >= m || j >= n)
        return;
    col -= base, val -= base;
    V        *cp = reinterpret_cast<V *>(C + (size_t)i * ldc + j);
    const V   c0 = *cp; // no dependency on A: in flight while the row's pointers and entries arrive
    const int s = row_ptr[i], e = row_ptr[i + 1], len = e - s;
    T         a0 = T(0), a1 = T(0);
    bool      first = true;
    const T  *Bj = B + j - (ptrdiff_t)base * ldb;
    auto      batch = [&](int p, auto wtag) {
        constexpr int W = decltype(wtag)::value;
        T             v[W];
        V             b[W];
#pragma unroll
        for(int k = 0; k < W; k++)
            v[k] = KT ? alpha * val[p + k] : val[p + k];
#pragma unroll
        for(int k = 0; k < W; k++)
            b[k] = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)col[p + k] * ldb);
        if(KT && first) // (after the B rows are requested: the chain starts from beta * C)
            a0 = c0.x * beta, a1 = c0.y * beta, first = false;
        if(KT && kt_tail > 0) // (wave-uniform: only launches with tail columns pay for the second form)
        {
            const bool t0 = j >= n - kt_tail, t1 = j + 1 >= n - kt_tail;
#pragma unroll
            for(int k = 0; k < W; k++)
            {
                const T sv = val[p + k];
                a0         = t0 ? mm_fma(sv * b[k].x, alpha, a0) : mm_fma(v[k], b[k].x, a0);
                a1         = t1 ? mm_fma(sv * b[k].y, alpha, a1) : mm_fma(v[k], b[k].y, a1);
            }
        }
        else
        {
#pragma unroll
            for(int k = 0; k < W; k++)
                a0 = mm_fma(v[k], b[k].x, a0), a1 = mm_fma(v[k], b[k].y, a1);
        }
    };
    int p = s;
    for(; p + 8 <= e; p += 8)
        batch(p, std::integral_constant<int, 8>{});
    switch(e - p) // wave-uniform
    {
    case 1: batch(p, std::integral_constant<int, 1>{}); break;
    case 2: batch(p, std::integral_constant<int, 2>{}); break;
    case 3: batch(p, std::integral_constant<int, 3>{}); break;
    case 4: batch(p, std::integral_constant<int, 4>{}); break;
    case 5: batch(p, std::integral_constant<int, 5>{}); break;
    case 6: batch(p, std::integral_constant<int, 6>{}); break;
    case 7: batch(p, std::integral_constant<int, 7>{}); break;
    default: break;
    }
    (void)len;
    typedef T nt2 __attribute__((ext_vector_type(2)));
    nt2       o;
    if constexpr(KT)
    {
        if(first) // an empty row
            a0 = c0.x * beta, a1 = c0.y * beta;
        o.x = a0, o.y = a1;
    }
    else
        o.x = mm_fma(beta, c0.x, alpha * a0), o.y = mm_fma(beta, c0.y, alpha * a1);
    __builtin_nontemporal_store(o, reinterpret_cast<nt2 *>(cp));
}

// float, C read, n a multiple of 4 and >= 256 (round 4): the same kernel with FOUR columns per lane -- a float load instruction
// moves half the bytes of a double one, so the two-column form spends twice the instructions (and reads A's row twice as often)
// per byte of B and C.  Same chain per element (same bits as the two-column form).  1000^2 Laplacian, 256 columns: 0.740 -> 0.580 ms
// (0.53 -> 0.67 of the roofline, the double product's fraction; profiles/r4/float_headline.txt).
__global__ __launch_bounds__(256) void csrmm_row_wave_rc4_kernel(int base, float alpha, aoclsparse_int m, const float *__restrict__ val,
                                                                 const aoclsparse_int *__restrict__ col,
                                                                 const aoclsparse_int *__restrict__ row_ptr,
                                                                 const float *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                                 float beta, float *__restrict__ C, aoclsparse_int ldc, int xcd_chunk)
{
    const int w  = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int bx = mm_block_index(xcd_chunk);
    const int i  = bx * 4 + w;
    const int j  = 4 * (int)(threadIdx.x & 63) + 256 * (int)blockIdx.y;
    if(i >= m || j >= n)
        return;
    col -= base, val -= base;
    float4      *cp = reinterpret_cast<float4 *>(C + (size_t)i * ldc + j);
    const float4 c0 = *cp; // no dependency on A: in flight while the row's pointers and entries arrive
    const int    s = row_ptr[i], e = row_ptr[i + 1];
    float        a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float *Bj = B + j - (ptrdiff_t)base * ldb;
    auto         batch = [&](int p, auto wtag) {
        constexpr int W = decltype(wtag)::value;
        float         v[W];
        float4        b[W];
#pragma unroll
        for(int k = 0; k < W; k++)
            v[k] = val[p + k];
#pragma unroll
        for(int k = 0; k < W; k++)
            b[k] = *reinterpret_cast<const float4 *>(Bj + (ptrdiff_t)col[p + k] * ldb);
#pragma unroll
        for(int k = 0; k < W; k++)
        {
            a0 = mm_fma(v[k], b[k].x, a0), a1 = mm_fma(v[k], b[k].y, a1);
            a2 = mm_fma(v[k], b[k].z, a2), a3 = mm_fma(v[k], b[k].w, a3);
        }
    };
    int p = s;
    for(; p + 8 <= e; p += 8)
        batch(p, std::integral_constant<int, 8>{});
    switch(e - p) // wave-uniform
    {
    case 1: batch(p, std::integral_constant<int, 1>{}); break;
    case 2: batch(p, std::integral_constant<int, 2>{}); break;
    case 3: batch(p, std::integral_constant<int, 3>{}); break;
    case 4: batch(p, std::integral_constant<int, 4>{}); break;
    case 5: batch(p, std::integral_constant<int, 5>{}); break;
    case 6: batch(p, std::integral_constant<int, 6>{}); break;
    case 7: batch(p, std::integral_constant<int, 7>{}); break;
    default: break;
    }
    typedef float nt4 __attribute__((ext_vector_type(4)));
    nt4           o;
    o.x = mm_fma(beta, c0.x, alpha * a0), o.y = mm_fma(beta, c0.y, alpha * a1);
    o.z = mm_fma(beta, c0.z, alpha * a2), o.w = mm_fma(beta, c0.w, alpha * a3);
    __builtin_nontemporal_store(o, reinterpret_cast<nt4 *>(cp));
}

// row-major, n >= 128, ROW RUNS (stencil-like matrices, csrmm_api.cpp: detect_row_runs): a wavefront walks R consecutive rows
// for one 128-column chunk and keeps the previous row's B rows in registers.  A stencil's rows repeat the previous row's
// column list shifted by one (i-1, i, i+1 -> i, i+1, i+2), so entry k of the new row needs exactly the B row that entry
// k + 1 of the previous row loaded: a 5-point row costs 3 new B-row loads instead of 5.  The test is made on the live
// column indices (wave-uniform compares), so a row that does not follow the pattern simply loads everything; rows of more
// than 8 entries take the plain loop.  Per output element the FMA chain is the row in CSR order: same bits as the other
// kernels.  C is stored non-temporally when it is not read.  Measured in tools/history/csrmm_r2.hip ("RR reuse R8 nt", 1000^2
// Laplacian, 256 columns): 0.878 vs 0.967 ms for the row-per-wave kernel on the same box; R = 2 / 4 / 6 0.90, R >= 12 worse.
template <typename T, int R>
__global__ __launch_bounds__(256) void csrmm_row_run_kernel(int base, T alpha, aoclsparse_int m,
                                                            const T *__restrict__ val,
                                                            const aoclsparse_int *__restrict__ col,
                                                            const aoclsparse_int *__restrict__ row_ptr,
                                                            const T *__restrict__ B, aoclsparse_int n,
                                                            aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                            aoclsparse_int ldc, bool readc, int xcd_chunk,
                                                            const aoclsparse_int *__restrict__ order, int ny)
{
    using V      = typename vec2<T>::type;
    const int w  = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // ny > 0: a 1-D grid with the 128-column chunk as the FASTEST index inside an XCD (all chunks of a row block run together)
    int bx, cy = (int)blockIdx.y;
    if(ny > 0)
    {
        const unsigned bi  = mm_linear_index(xcd_chunk);
        const int      idx = (int)(bi >> 3);
        cy                 = idx % ny;
        bx                 = (int)(bi & 7) * (xcd_chunk & ~MM_DESCENDING) + idx / ny;
    }
    else
        bx = mm_block_index(xcd_chunk);
    const int j = 2 * (int)(threadIdx.x & 63) + 128 * cy;
    if((bx * 4 + w) * R >= m || j >= n)
        return;
    // `order` (optional): the R-row blocks in the order the analysis wants them walked (MmGroups::run_order: strips of a
    // banded matrix, so that the B rows a block shares with the blocks one band above / below are still in this XCD's L2)
    const int i0 = order ? order[bx * 4 + w] : (bx * 4 + w) * R;
    // The index base is folded into the pointers once (col / val are indexed with the raw row_ptr values, B rows with the raw
    // column values): with "- base" inside the loop this kernel lost 13 % (0.957 vs 0.842 ms in tools/history/csrmm_r2.hip, RR1 vs RR).
    col -= base, val -= base;
    const T *Bj = B + j - (ptrdiff_t)base * ldb;
    int      pc[8]; // previous row's (raw) columns (INT_MIN = none) and the B rows loaded for them
    V        pb[8];
#pragma unroll
    for(int k = 0; k < 8; k++)
        pc[k] = INT_MIN, pb[k].x = T(0), pb[k].y = T(0);
    // The block's row pointers in one go, and -- when the block is whole and its last row is at least 8 entries away from the
    // end of the arrays -- every row's 8 column indices and values as WIDE scalar loads (entries beyond the row belong to the
    // next rows: read, never used), those of row r + 1 requested before row r's B rows.  Before: two dword loads for the row
    // pointers, then 8 + 8 clamped single loads per row, each row's only after the previous row was stored.
    int        rp[R + 1];
    const int  pend = row_ptr[m];
    const bool whole = i0 + R <= m;
#pragma unroll
    for(int r = 0; r <= R; r++)
        rp[r] = whole ? row_ptr[i0 + r] : row_ptr[min(i0 + r, m)];
    const bool fast = whole && rp[R] + 8 <= pend; // wave-uniform
    int        cn[8]; // (the column indices run one row ahead -- they gate the B-row loads; the values are needed at the
                      // FMAs only and are requested at the start of their own row: both a row ahead spilled 76 SGPRs)
    if(fast)
    {
#pragma unroll
        for(int k = 0; k < 8; k++)
            cn[k] = col[rp[0] + k];
    }
    // (fully unrolled on purpose; as a rolled loop this kernel was SLOWER than the row-per-wave one, 1.07 vs 0.985 ms)
#pragma clang loop unroll(full)
    for(int r = 0; r < R; r++)
    {
        const int  i    = i0 + r;
        const bool live = i < m; // wave-uniform
        const int  s = rp[r], len = live ? rp[r + 1] - s : 0;
        int        c[8];
        T          v[8];
        if(fast)
        {
#pragma unroll
            for(int k = 0; k < 8; k++)
                c[k] = cn[k], v[k] = val[s + k];
            if(r + 1 < R)
            {
#pragma unroll
                for(int k = 0; k < 8; k++)
                    cn[k] = col[rp[r + 1] + k];
            }
        }
        else if(len <= 8)
        {
#pragma unroll
            for(int k = 0; k < 8; k++)
            {
                const int q = s + (k < len ? k : len - 1); // clamped: all eight loads go out together
                c[k]        = len > 0 ? col[q] : INT_MIN + 1;
                v[k]        = len > 0 ? val[q] : T(0);
            }
        }
        T         a0 = T(0), a1 = T(0);
        if(len <= 8)
        {
            V b[8];
            // (Tried: request every entry that cannot reuse a register first, copy the reused ones afterwards, so that a row's
            // loads are all in flight together -- the compiler waits behind each conditional load as written here.  It was
            // SLOWER, 1.156 vs 0.855 ms in tools/history/csrmm_r2.hip: three register sets per row instead of two.)
#pragma unroll
            for(int k = 0; k < 8; k++)
                if(k < len)
                {
                    if(k + 1 < 8 && c[k] == pc[k + 1])
                        b[k] = pb[k + 1];
                    else
                        b[k] = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)c[k] * ldb);
                }
#pragma unroll
            for(int k = 0; k < 8; k++)
                if(k < len)
                    a0 = mm_fma(v[k], b[k].x, a0), a1 = mm_fma(v[k], b[k].y, a1);
#pragma unroll
            for(int k = 0; k < 8; k++)
                pc[k] = k < len ? c[k] : INT_MIN, pb[k] = b[k];
        }
        else
        {
            for(int p = s; p < s + len; p++)
            {
                const T v0 = val[p];
                const V b0 = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)col[p] * ldb);
                a0 = mm_fma(v0, b0.x, a0), a1 = mm_fma(v0, b0.y, a1);
            }
#pragma unroll
            for(int k = 0; k < 8; k++)
                pc[k] = INT_MIN;
        }
        if(!live)
            continue;
        V      *cp = reinterpret_cast<V *>(C + (size_t)i * ldc + j);
        const T z0 = alpha * a0, z1 = alpha * a1;
        // C is read only where the reference's beta * C + z can differ from z (beta != 0, or a zero z whose sign beta * C
        // decides); ONE wave-uniform test, so that the common path is a straight non-temporal store
        const bool need = readc || z0 == T(0) || z1 == T(0);
        typedef T  nt2 __attribute__((ext_vector_type(2)));
        nt2        o;
        o.x = z0, o.y = z1;
        if(__builtin_amdgcn_ballot_w64(need) != 0)
        {
            const V c2 = *cp;
            o.x        = need ? mm_fma(beta, c2.x, z0) : z0;
            o.y        = need ? mm_fma(beta, c2.y, z1) : z1;
        }
        // non-temporal on both paths (with a plain store on one of them the compiler merged the two into ONE plain store)
        __builtin_nontemporal_store(o, reinterpret_cast<nt2 *>(cp));
    }
}

// row-major, n >= 128, ROW GROUPS: consecutive rows with the SAME column pattern (the dof rows of one node of a
// finite-element matrix, the rows of a dense block) are solved by one wavefront per (group, 128-column chunk): every
// B row the group touches is loaded once and used for all its rows, i.e. the L2 -> CU traffic that bounds the
// row-per-wave kernel on matrices with tens of non-zeros per row drops by the group size.  This is the blocked-ELL
// idea without the padding: a group IS a dense r x L block of A, read in place from the CSR arrays (row ii's k-th
// entry sits at row_ptr[ii] + k for every row of the group).  Per output element the FMA chain is still the row in
// CSR order, so the result equals the other kernels' bit for bit.  Groups are found once per handle on the host
// (csrmm_api.cpp: build_mm_groups) and capped at CSRMM_GROUP rows.  (The first version of this kernel, which the one below
// replaced at 2.76 -> 2.15 ms on the shell-like stand-in, was removed in round 3: profiles/r2/csrmm_rowgroup_v2.jsonl.)
// Second version of the row-group kernel (n >= 128), written after reading the first one's ISA: there, every 8-entry step
// was SIX scalar-load round trips one after the other (the column indices, then -- inside a wave-uniform branch per row --
// each row's eight values, waited for before that row's FMAs) and the B-row loads of step k + 1 were not issued before
// the FMAs of step k were done.  Here a step is 4 entries: the values of ALL rows of the group are loaded together
// (GR x 4 doubles through the scalar cache, unconditionally -- rows the group does not have re-read row 0's), the B rows
// of the NEXT step are requested before the FMAs of this one (two register sets, used alternately), and only the FMAs sit
// behind the per-row branch.  GR is the matrix's largest group size exactly (2, 3, 4, 5, 6 or 8), not rounded up to 8.
// RC (round 3): C is read (beta != 0, or the default beta == 0 mode): its rows are requested as soon as the group is known
// instead of one after the other behind the chains (GR dependent round trips at the end of a wave's life).
template <typename T, int GR, bool CCOL = false, bool RC = false>
__global__ __launch_bounds__(256) void csrmm_rowgroup2_kernel(int base, T alpha, aoclsparse_int ngroups,
                                                              const aoclsparse_int *__restrict__ grp,
                                                              const T *__restrict__ val,
                                                              const aoclsparse_int *__restrict__ col,
                                                              const aoclsparse_int *__restrict__ row_ptr,
                                                              const T *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                              T beta, T *__restrict__ C, aoclsparse_int ldc, bool readc,
                                                              int xcd_chunk, const aoclsparse_int *__restrict__ glist)
{
    using V         = typename vec2<T>::type;
    constexpr int U = 4;
    const int     w  = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int     bx = mm_block_index(xcd_chunk);
    const int     gt = bx * 4 + w;
    const int     jr = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    const bool    have = gt < ngroups;
    if constexpr(!CCOL)
    {
        if(!have || jr >= n)
            return;
    }
    else if(bx * 4 >= ngroups) // a workgroup of the XCD padding: all four waves leave together
        return;
    // (CCOL: every wave stays for the workgroup's store phase; a wave without a group has no rows, a lane beyond n computes
    // column 0 again and stores nothing)
    const int j  = (CCOL && jr >= n) ? 0 : jr;
    const int gi = have ? (glist ? glist[gt] : gt) : 0;
    const int i0 = have ? grp[gi] : 0, r = have ? grp[gi + 1] - i0 : 0; // 1 <= r <= GR
    // (the index base is folded into the pointers once: col / val take raw row_ptr values, B rows raw column values --
    // subtracting it per index cost the row-run kernel 13 %)
    col -= base, val -= base;
    const int s0 = row_ptr[i0], len = have ? row_ptr[i0 + 1] - s0 : 0;
    int       so[GR]; // start of every row of the group (wave-uniform); rows beyond the group alias row 0
#pragma unroll
    for(int q = 0; q < GR; q++)
        so[q] = q < r ? row_ptr[i0 + q] : s0;
    T acc0[GR], acc1[GR];
#pragma unroll
    for(int q = 0; q < GR; q++)
        acc0[q] = T(0), acc1[q] = T(0);
    V cpre[RC ? GR : 1];
    if constexpr(RC && !CCOL)
    {
#pragma unroll
        for(int q = 0; q < GR; q++)
            if(q < r)
                cpre[q] = *reinterpret_cast<const V *>(C + (size_t)(i0 + q) * ldc + j);
    }
    const T  *Bj    = B + j - (ptrdiff_t)base * ldb;
    const int nstep = len / U;
    // (no branch on q < r around the FMAs: rows the group does not have accumulate row 0's products into accumulators
    // that are never stored -- with the branch the compiler sinks each row's value loads behind it again)
#define MM_FETCH(b, k)                                                                         \
    {                                                                                          \
        int c_[U];                                                                             \
        _Pragma("unroll") for(int u = 0; u < U; u++) c_[u] = col[s0 + (k) + u];                \
        _Pragma("unroll") for(int u = 0; u < U; u++)                                           \
            b[u] = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)c_[u] * ldb);                  \
    }
#define MM_MAC(b, k)                                                                           \
    {                                                                                          \
        T a_[GR][U];                                                                           \
        _Pragma("unroll") for(int q = 0; q < GR; q++)                                          \
            _Pragma("unroll") for(int u = 0; u < U; u++) a_[q][u] = val[so[q] + (k) + u];      \
        _Pragma("unroll") for(int q = 0; q < GR; q++)                                          \
            _Pragma("unroll") for(int u = 0; u < U; u++)                                       \
            {                                                                                  \
                acc0[q] = mm_fma(a_[q][u], b[u].x, acc0[q]);                                   \
                acc1[q] = mm_fma(a_[q][u], b[u].y, acc1[q]);                                   \
            }                                                                                  \
    }
    V   b0[U], b1[U];
    int st = 0;
    if(nstep > 0)
        MM_FETCH(b0, 0)
    for(; st + 2 <= nstep; st += 2)
    {
        MM_FETCH(b1, (st + 1) * U)
        MM_MAC(b0, st * U)
        if(st + 2 < nstep)
            MM_FETCH(b0, (st + 2) * U)
        MM_MAC(b1, (st + 1) * U)
    }
    if(st < nstep)
        MM_MAC(b0, st * U)
#undef MM_FETCH
#undef MM_MAC
    // the last 1..3 entries as ONE more batch: indices clamped to the row, every load issued (and pinned: the compiler would
    // sink them behind the branches below), then the FMAs of the entries that exist -- entry by entry this tail was three
    // dependent index -> B row -> values chains, as long as the whole main loop of a 35-entry row
    {
        const int k = nstep * U;
        if(k < len)
        {
            int c_[U - 1];
            V   bt[U - 1];
            T   a_[GR][U - 1];
#pragma unroll
            for(int u = 0; u < U - 1; u++)
                c_[u] = col[s0 + min(k + u, len - 1)];
#pragma unroll
            for(int u = 0; u < U - 1; u++)
                bt[u] = *reinterpret_cast<const V *>(Bj + (ptrdiff_t)c_[u] * ldb);
#pragma unroll
            for(int q = 0; q < GR; q++)
#pragma unroll
                for(int u = 0; u < U - 1; u++)
                    a_[q][u] = val[so[q] + min(k + u, len - 1)];
#pragma unroll
            for(int q = 0; q < GR; q++)
#pragma unroll
                for(int u = 0; u < U - 1; u++)
                    asm volatile("" : "+s"(a_[q][u]));
#pragma unroll
            for(int u = 0; u < U - 1; u++)
                asm volatile("" : "+v"(bt[u].x), "+v"(bt[u].y));
#pragma unroll
            for(int u = 0; u < U - 1; u++)
                if(k + u < len)
                {
#pragma unroll
                    for(int q = 0; q < GR; q++)
                    {
                        acc0[q] = mm_fma(a_[q][u], bt[u].x, acc0[q]);
                        acc1[q] = mm_fma(a_[q][u], bt[u].y, acc1[q]);
                    }
                }
        }
    }
    if constexpr(CCOL)
    {
        // C is COLUMN-major (the column-major detour of csrmm_api.cpp: B was copied to row-major scratch, C is written where
        // the caller wants it): element (i, j) at C[i + j * ldc].  The workgroup's four groups are consecutive rows (R = 4..32
        // of them): alpha * acc goes through an LDS tile and leaves column by column, R consecutive rows per run -- lane by
        // lane (r rows x 8 bytes per lane, 64 different lines per store instruction) the kernel took twice as long.
        // beta * C + z is applied at the store (C read only where it can matter, as everywhere).
        __shared__ T tile[4 * GR][130];
        const int    lane = (int)(threadIdx.x & 63);
        const int    g0 = bx * 4, g1 = min(g0 + 4, (int)ngroups);
        const int    ib = grp[g0], R = grp[g1] - ib; // this workgroup's rows [ib, ib + R)
#pragma unroll
        for(int q = 0; q < GR; q++)
            if(q < r)
            {
                tile[i0 - ib + q][2 * lane]     = alpha * acc0[q];
                tile[i0 - ib + q][2 * lane + 1] = alpha * acc1[q];
            }
        __syncthreads();
        const int j0 = 128 * (int)blockIdx.y, nc = min(128, (int)n - j0);
        for(int idx = (int)threadIdx.x; idx < nc * R; idx += 256)
        {
            const int cc = idx / R, q = idx - cc * R;
            T        *cp = C + (size_t)(ib + q) + (size_t)(j0 + cc) * ldc;
            const T   z  = tile[q][cc];
            *cp          = (readc || z == T(0)) ? mm_fma(beta, *cp, z) : z;
        }
        return;
    }
#pragma unroll
    for(int q = 0; q < GR; q++)
        if(q < r)
        {
            V      *cp = reinterpret_cast<V *>(C + (size_t)(i0 + q) * ldc + j);
            const T z0 = alpha * acc0[q], z1 = alpha * acc1[q];
            typedef T  nt2 __attribute__((ext_vector_type(2)));
            nt2        o;
            if constexpr(RC)
            {
                o.x = mm_fma(beta, cpre[q].x, z0), o.y = mm_fma(beta, cpre[q].y, z1);
                __builtin_nontemporal_store(o, reinterpret_cast<nt2 *>(cp));
                continue;
            }
            // C is read only where beta * C + z can differ from z: one wave-uniform test, non-temporal store on both paths
            // (C is written once and never read again: it stays out of the L2 the B rows live in)
            const bool need = readc || z0 == T(0) || z1 == T(0);
            o.x = z0, o.y = z1;
            if(__builtin_amdgcn_ballot_w64(need) != 0)
            {
                const V c = *cp;
                o.x       = need ? mm_fma(beta, c.x, z0) : z0;
                o.y       = need ? mm_fma(beta, c.y, z1) : z1;
            }
            __builtin_nontemporal_store(o, reinterpret_cast<nt2 *>(cp));
        }
}

// narrower B (32 <= n < 128): LANES lanes (2 columns each) per group, 64 / LANES groups per wavefront; the group's
// row_ptr / col / val loads are then per-lane loads of one address per sub-wave instead of scalar loads
template <typename T, int LANES, int GR, bool CCOL = false>
__global__ __launch_bounds__(256) void csrmm_rowgroup_sub_kernel(int base, T alpha, aoclsparse_int ngroups,
                                                             const aoclsparse_int *__restrict__ grp,
                                                             const T *__restrict__ val,
                                                             const aoclsparse_int *__restrict__ col,
                                                             const aoclsparse_int *__restrict__ row_ptr,
                                                             const T *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                             T beta, T *__restrict__ C, aoclsparse_int ldc, bool readc,
                                                             int xcd_chunk)
{
    using V      = typename vec2<T>::type;
    const int bx = mm_block_index(xcd_chunk);
    constexpr int NG = 256 / LANES; // groups per workgroup
    const int     gi = bx * NG + (int)threadIdx.x / LANES;
    const int     jr = 2 * ((int)threadIdx.x % LANES) + 2 * LANES * (int)blockIdx.y;
    const bool    have = gi < ngroups;
    if constexpr(!CCOL)
    {
        if(!have || jr >= n)
            return;
    }
    else if(bx * NG >= ngroups) // a workgroup of the XCD padding
        return;
    // (CCOL: every lane stays for the workgroup's store phase, see csrmm_rowgroup2_kernel)
    const int j  = (CCOL && jr >= n) ? 0 : jr;
    const int i0 = have ? grp[gi] : 0, r = have ? grp[gi + 1] - i0 : 0; // 1 <= r <= GR
    const int s0 = row_ptr[i0] - base, len = have ? row_ptr[i0 + 1] - base - s0 : 0;
    int       so[GR]; // start of every row of the group (wave-uniform)
#pragma unroll
    for(int q = 0; q < GR; q++)
        so[q] = q < r ? row_ptr[i0 + q] - base : s0;
    T acc0[GR], acc1[GR];
#pragma unroll
    for(int q = 0; q < GR; q++)
        acc0[q] = T(0), acc1[q] = T(0);
    const T *Bj = B + j;
    int      k  = 0;
    for(; k + 8 <= len; k += 8) // eight B rows in flight per step
    {
        V b[8];
#pragma unroll
        for(int u = 0; u < 8; u++)
            b[u] = *reinterpret_cast<const V *>(Bj + (size_t)(col[s0 + k + u] - base) * ldb);
#pragma unroll
        for(int q = 0; q < GR; q++)
            if(q < r)
            {
                T a[8];
#pragma unroll
                for(int u = 0; u < 8; u++)
                    a[u] = val[so[q] + k + u];
#pragma unroll
                for(int u = 0; u < 8; u++)
                    acc0[q] = mm_fma(a[u], b[u].x, acc0[q]), acc1[q] = mm_fma(a[u], b[u].y, acc1[q]);
            }
    }
    for(; k + 4 <= len; k += 4) // four B rows in flight per step
    {
        V b[4];
#pragma unroll
        for(int u = 0; u < 4; u++)
            b[u] = *reinterpret_cast<const V *>(Bj + (size_t)(col[s0 + k + u] - base) * ldb);
#pragma unroll
        for(int q = 0; q < GR; q++)
            if(q < r)
            {
                T a[4];
#pragma unroll
                for(int u = 0; u < 4; u++)
                    a[u] = val[so[q] + k + u];
#pragma unroll
                for(int u = 0; u < 4; u++)
                    acc0[q] = mm_fma(a[u], b[u].x, acc0[q]), acc1[q] = mm_fma(a[u], b[u].y, acc1[q]);
            }
    }
    for(; k < len; k++)
    {
        const V b0 = *reinterpret_cast<const V *>(Bj + (size_t)(col[s0 + k] - base) * ldb);
#pragma unroll
        for(int q = 0; q < GR; q++)
            if(q < r)
            {
                const T a0 = val[so[q] + k];
                acc0[q] = mm_fma(a0, b0.x, acc0[q]), acc1[q] = mm_fma(a0, b0.y, acc1[q]);
            }
    }
    if constexpr(CCOL)
    {
        // column-major C through an LDS tile, as in csrmm_rowgroup2_kernel: the workgroup's NG groups are consecutive rows
        __shared__ T tile[NG * GR][2 * LANES + 2];
        const int    lane = (int)threadIdx.x % LANES;
        const int    g0 = bx * NG, g1 = min(g0 + NG, (int)ngroups);
        const int    ib = grp[g0], R = grp[g1] - ib;
#pragma unroll
        for(int q = 0; q < GR; q++)
            if(q < r)
            {
                tile[i0 - ib + q][2 * lane]     = alpha * acc0[q];
                tile[i0 - ib + q][2 * lane + 1] = alpha * acc1[q];
            }
        __syncthreads();
        const int j0 = 2 * LANES * (int)blockIdx.y, nc = min(2 * LANES, (int)n - j0);
        for(int idx = (int)threadIdx.x; idx < nc * R; idx += 256)
        {
            const int cc = idx / R, q = idx - cc * R;
            T        *cp = C + (size_t)(ib + q) + (size_t)(j0 + cc) * ldc;
            const T   z  = tile[q][cc];
            *cp          = (readc || z == T(0)) ? mm_fma(beta, *cp, z) : z;
        }
        return;
    }
#pragma unroll
    for(int q = 0; q < GR; q++)
        if(q < r)
        {
            V      *cp = reinterpret_cast<V *>(C + (size_t)(i0 + q) * ldc + j);
            const T z0 = alpha * acc0[q], z1 = alpha * acc1[q];
            V       c;
            if(readc || z0 == T(0) || z1 == T(0))
            {
                c   = *cp;
                c.x = mm_fma(beta, c.x, z0);
                c.y = mm_fma(beta, c.y, z1);
            }
            else
                c.x = z0, c.y = z1;
            *cp = c;
        }
}


// row-major, NARROW B (n < 128 -- the column slab one of several GPUs owns): one workgroup per CSR-Adaptive row block
// of the handle's SpMV plan (consecutive rows, <= TILE entries).  The block's row_ptr / col_ind / val are fetched with
// coalesced loads into LDS -- two dependent round trips per BLOCK instead of three per row, and a tenth of the vector
// memory instructions the row-per-sub-wave kernel spends on A -- then a sub-wave of LANES lanes (2 columns each) walks
// rows, two rows in flight, eight B-row loads per row and step.  Per output element the FMA chain is the row in CSR
// order, so the bits are those of every other kernel here.  beta == 0 stores are non-temporal (C is written once and
// not read again by this launch).  32 columns of the 1000^2 Laplacian: 0.131 ms against 0.195 ms for csrmm_row_kernel
// (tools/history/csrmm_r2.hip, profiles/r2/csrmm_experiments.txt).
template <typename T, int LANES, int TILE, int UR, int NB, bool RC, bool KT = false, bool TRACE = false>
__global__ __launch_bounds__(256, 4) void csrmm_tile_kernel(int base, T alpha, const T *__restrict__ val,
                                                         const aoclsparse_int *__restrict__ col,
                                                         const aoclsparse_int *__restrict__ row_ptr,
                                                         const aoclsparse_int *__restrict__ blocks, aoclsparse_int nblocks,
                                                         const T *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                         T beta, T *__restrict__ C, aoclsparse_int ldc, bool readc,
                                                         int xcd_chunk, unsigned long long *trace = nullptr, int kt_tail = 0)
{
    // diagnostic (AOCLSPARSE_MI355_MM_TRACE=<file>, tools/mm_trace.py): 100 MHz stamps per workgroup -- start / block table
    // read / tile loads landed / after the barrier / end -- kept in registers and stored by thread 0 at the very end
    unsigned long long t_st[5] = {0, 0, 0, 0, 0};
    if constexpr(TRACE)
        t_st[0] = __builtin_amdgcn_s_memrealtime();
    using V               = typename vec2<T>::type;
    constexpr int MAXR    = 512; // spmv_maxrows(TILE) <= 512
    constexpr int NSUB    = 256 / LANES;
    // UR rows in flight per sub-wave, NB B-row loads per row and step; RC: C is read (beta != 0, or the reference's 0 * C)
    __shared__ int s_ptr[MAXR + 1];
    __shared__ int s_col[TILE];
    __shared__ T   s_val[TILE];
    const int bx = mm_block_index(xcd_chunk);
    if(bx >= nblocks)
        return;
    const int r0 = blocks[2 * bx], s0 = blocks[2 * bx + 1];
    const int nrows = blocks[2 * bx + 2] - r0, cnt = blocks[2 * bx + 3] - s0;
    const int tid = threadIdx.x, sub = tid / LANES, lane = tid % LANES;
    const int j   = 2 * lane + 2 * LANES * (int)blockIdx.y;
    if constexpr(TRACE)
        t_st[1] = __builtin_amdgcn_readfirstlane(cnt) >= 0 ? __builtin_amdgcn_s_memrealtime() : 0;
    auto      put = [&](int row, T a0, T a1) {
        V      *cp = reinterpret_cast<V *>(C + (size_t)row * ldc + j);
        const T z0 = alpha * a0, z1 = alpha * a1;
        V       c;
        if(readc || z0 == T(0) || z1 == T(0))
        {
            c   = *cp;
            c.x = mm_fma(beta, c.x, z0);
            c.y = mm_fma(beta, c.y, z1);
            *cp = c;
        }
        else
        {
            typedef T nt2 __attribute__((ext_vector_type(2))); // native vector: the non-temporal builtin takes no struct
            nt2 o;
            o.x = z0, o.y = z1;
            __builtin_nontemporal_store(o, reinterpret_cast<nt2 *>(cp));
        }
    };
    if(cnt > TILE)
    {
        // a single row longer than a tile: straight from global memory, column chunks spread over the sub-waves
        if(sub == 0 && j < n)
        {
            T        a0 = T(0), a1 = T(0);
            const T *Bj = B + j;
            V       *cp = reinterpret_cast<V *>(C + (size_t)r0 * ldc + j);
            if constexpr(KT) // (csrmm_row_kt: the chain starts from beta * C and carries alpha * a)
            {
                const V c = *cp;
                a0 = c.x * beta, a1 = c.y * beta;
            }
            const bool t0 = KT && j >= n - kt_tail, t1 = KT && j + 1 >= n - kt_tail; // (csrmm_row_kt's scalar tail columns)
            for(int p = s0; p < s0 + cnt; p++)
            {
                const T sv = val[p];
                const T a  = KT ? alpha * sv : sv;
                const V b  = *reinterpret_cast<const V *>(Bj + (size_t)(col[p] - base) * ldb);
                a0 = t0 ? mm_fma(sv * b.x, alpha, a0) : mm_fma(a, b.x, a0);
                a1 = t1 ? mm_fma(sv * b.y, alpha, a1) : mm_fma(a, b.y, a1);
            }
            if constexpr(KT)
            {
                V c;
                c.x = a0, c.y = a1;
                *cp = c;
            }
            else
                put(r0, a0, a1);
        }
        return;
    }
    for(int t = tid; t <= nrows; t += 256)
        s_ptr[t] = row_ptr[r0 + t] - base - s0;
    for(int t = tid; t < cnt; t += 256)
        s_col[t] = col[s0 + t] - base, s_val[t] = val[s0 + t];
    if constexpr(TRACE)
    {
        __builtin_amdgcn_s_waitcnt(0);
        t_st[2] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if constexpr(TRACE)
        t_st[3] = __builtin_amdgcn_s_memrealtime();
    if(j >= n)
        return;
    const T *Bj = B + j;
    for(int r = sub; r < nrows; r += NSUB * UR)
    {
        int p0[UR], p1[UR];
        T   a0[UR], a1[UR];
        V   cin[UR] = {};
#pragma unroll
        for(int q = 0; q < UR; q++)
        {
            const int rr = r + q * NSUB;
            p0[q] = rr < nrows ? s_ptr[rr] : 0, p1[q] = rr < nrows ? s_ptr[rr + 1] : 0;
            a0[q] = T(0), a1[q] = T(0);
            // C is read (beta != 0, or the reference's 0 * C): requested BEFORE the B rows, so that the read-modify-write
            // at the end of the row does not add a dependent round trip (round 3)
            if constexpr(RC)
                if(rr < nrows)
                    cin[q] = *reinterpret_cast<const V *>(C + (size_t)(r0 + rr) * ldc + j);
        }
        bool more = true, first = true;
        while(more)
        {
            V b[UR][NB];
            T v[UR][NB];
#pragma unroll
            for(int q = 0; q < UR; q++)
#pragma unroll
                for(int u = 0; u < NB; u++)
                    if(p0[q] + u < p1[q])
                    {
                        v[q][u] = KT ? alpha * s_val[p0[q] + u] : s_val[p0[q] + u];
                        b[q][u] = *reinterpret_cast<const V *>(Bj + (size_t)s_col[p0[q] + u] * ldb);
                    }
            if(KT && first) // (after the B rows are requested: the chains start from beta * C)
            {
#pragma unroll
                for(int q = 0; q < UR; q++)
                    a0[q] = cin[q].x * beta, a1[q] = cin[q].y * beta;
                first = false;
            }
            more = false;
            if(KT && kt_tail > 0) // (wave-uniform) csrmm_row_kt's scalar tail columns: fma(sv * b, alpha, c)
            {
                const bool t0 = j >= n - kt_tail, t1 = j + 1 >= n - kt_tail;
#pragma unroll
                for(int q = 0; q < UR; q++)
#pragma unroll
                    for(int u = 0; u < NB; u++)
                        if(p0[q] + u < p1[q])
                        {
                            const T sv = s_val[p0[q] + u];
                            a0[q]      = t0 ? mm_fma(sv * b[q][u].x, alpha, a0[q]) : mm_fma(v[q][u], b[q][u].x, a0[q]);
                            a1[q]      = t1 ? mm_fma(sv * b[q][u].y, alpha, a1[q]) : mm_fma(v[q][u], b[q][u].y, a1[q]);
                        }
#pragma unroll
                for(int q = 0; q < UR; q++)
                {
                    p0[q] += NB;
                    more |= p0[q] < p1[q];
                }
            }
            else
#pragma unroll
            for(int q = 0; q < UR; q++)
            {
#pragma unroll
                for(int u = 0; u < NB; u++)
                    if(p0[q] + u < p1[q])
                        a0[q] = mm_fma(v[q][u], b[q][u].x, a0[q]), a1[q] = mm_fma(v[q][u], b[q][u].y, a1[q]);
                p0[q] += NB;
                more |= p0[q] < p1[q];
            }
        }
#pragma unroll
        for(int q = 0; q < UR; q++)
            if(r + q * NSUB < nrows)
            {
                if constexpr(KT)
                {
                    V c;
                    c.x = first ? cin[q].x * beta : a0[q], c.y = first ? cin[q].y * beta : a1[q]; // (first: empty rows only)
                    *reinterpret_cast<V *>(C + (size_t)(r0 + r + q * NSUB) * ldc + j) = c;
                }
                else if constexpr(RC)
                {
                    V c;
                    c.x = mm_fma(beta, cin[q].x, alpha * a0[q]);
                    c.y = mm_fma(beta, cin[q].y, alpha * a1[q]);
                    *reinterpret_cast<V *>(C + (size_t)(r0 + r + q * NSUB) * ldc + j) = c;
                }
                else
                    put(r0 + r + q * NSUB, a0[q], a1[q]);
            }
    }
    if constexpr(TRACE)
        if(tid == 0 && blockIdx.y == 0 && trace)
        {
            unsigned long long *tr = trace + 8 * (size_t)bx;
            tr[0] = t_st[0], tr[1] = t_st[1], tr[2] = t_st[2], tr[3] = t_st[3], tr[4] = __builtin_amdgcn_s_memrealtime();
            tr[5] = ((unsigned long long)nrows << 32) | (unsigned)cnt, tr[6] = (unsigned long long)r0, tr[7] = 0;
        }
}

// column-major: one lane owns one row; the first CM_K entries of the row are kept in registers and the
// lane sweeps CM_COLS columns, so A is read n/CM_COLS times (once for a 32..64-column shard) and every
// B / C access is coalesced across the 64 rows of a wavefront.
constexpr int CM_K    = 8;
constexpr int CM_COLS = 64;
// columns per workgroup = the n columns spread evenly over the grid's y dimension, in whole steps of 4 (the host picks
// gridDim.y = ceil(n / cm_cols()))
__device__ inline int cm_cols_per_block(int n)
{
    return (int)(((unsigned)n + gridDim.y - 1) / gridDim.y + 3u) & ~3;
}
static constexpr int cm_cols()
{
    return CM_COLS; // columns per workgroup of the column-major kernels: the measured optimum (round 2)
}

template <typename T>
__global__ __launch_bounds__(256) void csrmm_col_kernel(int base, T alpha, aoclsparse_int m,
                                                        const T *__restrict__ val,
                                                        const aoclsparse_int *__restrict__ col,
                                                        const aoclsparse_int *__restrict__ row_ptr,
                                                        const T *__restrict__ B, aoclsparse_int n,
                                                        aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                        aoclsparse_int ldc, bool readc, int xcd_chunk,
                                                        const aoclsparse_int *__restrict__ rows)
{
    // rows != nullptr: the m entries of `rows` are the rows to compute (the partner-less rows of the pair kernel)
    const int bx = mm_block_index(xcd_chunk);
    const int t  = bx * blockDim.x + threadIdx.x;
    if(t >= m)
        return;
    const int i  = rows ? rows[t] : t;
    const int cpw = cm_cols_per_block(n);
    const int j0  = blockIdx.y * cpw;
    const int j1  = min(n, j0 + cpw);
    const int s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
    T         v[CM_K];
    int       c[CM_K];
#pragma unroll
    for(int k = 0; k < CM_K; k++)
    {
        v[k] = T(0);
        c[k] = 0;
        if(s + k < e)
        {
            v[k] = val[s + k];
            c[k] = col[s + k] - base;
        }
    }
    const int len = e - s;
    int       j   = j0;
    // four columns per step: four independent chains keep 4x the loads in flight per lane
    for(; j + 4 <= j1; j += 4)
    {
        const T *B0 = B + (size_t)j * ldb, *B1 = B0 + ldb, *B2 = B1 + ldb, *B3 = B2 + ldb;
        T        a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
#pragma unroll
        for(int k = 0; k < CM_K; k++)
            if(k < len)
            {
                a0 = mm_fma(v[k], B0[c[k]], a0);
                a1 = mm_fma(v[k], B1[c[k]], a1);
                a2 = mm_fma(v[k], B2[c[k]], a2);
                a3 = mm_fma(v[k], B3[c[k]], a3);
            }
        for(int p = s + CM_K; p < e; p++)
        {
            const T   av = val[p];
            const int cc = col[p] - base;
            a0 = mm_fma(av, B0[cc], a0);
            a1 = mm_fma(av, B1[cc], a1);
            a2 = mm_fma(av, B2[cc], a2);
            a3 = mm_fma(av, B3[cc], a3);
        }
        T      *cp = C + (size_t)i + (size_t)j * ldc;
        const T z0 = alpha * a0, z1 = alpha * a1, z2 = alpha * a2, z3 = alpha * a3;
        cp[0]                = (readc || z0 == T(0)) ? mm_fma(beta, cp[0], z0) : z0;
        cp[(size_t)ldc]      = (readc || z1 == T(0)) ? mm_fma(beta, cp[(size_t)ldc], z1) : z1;
        cp[2 * (size_t)ldc]  = (readc || z2 == T(0)) ? mm_fma(beta, cp[2 * (size_t)ldc], z2) : z2;
        cp[3 * (size_t)ldc]  = (readc || z3 == T(0)) ? mm_fma(beta, cp[3 * (size_t)ldc], z3) : z3;
    }
    for(; j < j1; j++)
    {
        const T *Bj  = B + (size_t)j * ldb;
        T        acc = T(0);
#pragma unroll
        for(int k = 0; k < CM_K; k++)
            if(k < len)
                acc = mm_fma(v[k], Bj[c[k]], acc);
        for(int p = s + CM_K; p < e; p++)
            acc = mm_fma(val[p], Bj[col[p] - base], acc);
        T      *cp = C + (size_t)i + (size_t)j * ldc;
        const T z  = alpha * acc;
        *cp        = (readc || z == T(0)) ? mm_fma(beta, *cp, z) : z;
    }
}

// column-major, PAIRED rows: a lane owns rows (i, i+1) taken from a list built once per handle (csrmm_api.cpp:
// detect_pairs): row i+1 carries row i's pattern shifted by one column (5-point and other scalar stencils, banded
// matrices) and both fit the register cache.  Entry k of both rows then reads B[c_k] and B[c_k + 1] of a column: ONE
// 16-byte load (8-byte aligned, which gfx950 serves) feeds both rows and the two results leave as one 16-byte store --
// half the vector-memory instructions of csrmm_col_kernel, which is what bounds that kernel (1.02 vs 1.30 ms at 256
// columns of a 1M-row 5-diagonal matrix, tools/history/csrmm_r2.hip).  Per output element the FMA chain is unchanged.  Rows that
// found no partner are served by csrmm_col_kernel through a row list in a second launch.
// K = entries of a row kept in registers (the pairs of detect_pairs have <= CM_K), U = columns per step, RC = C is read (beta != 0 or the
// reference's 0 * C): requested together with the step's B values instead of after its FMAs.
template <typename T, int K, int U, bool RC>
__global__ __launch_bounds__(256) void csrmm_colpair_kernel(int base, T alpha, aoclsparse_int npairs,
                                                            const aoclsparse_int *__restrict__ pair_first,
                                                            const T *__restrict__ val,
                                                            const aoclsparse_int *__restrict__ col,
                                                            const aoclsparse_int *__restrict__ row_ptr,
                                                            const T *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                            T beta, T *__restrict__ C, aoclsparse_int ldc, bool readc,
                                                            bool c_aligned, int xcd_chunk)
{
    typedef T v2 __attribute__((ext_vector_type(2)));
    const int bx = mm_block_index(xcd_chunk);
    const int t  = bx * 256 + (int)threadIdx.x;
    if(t >= npairs)
        return;
    const int i  = pair_first[t];
    const int cpw = cm_cols_per_block(n);
    const int j0 = blockIdx.y * cpw, j1 = min(n, j0 + cpw);
    const int s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
    const int len = e - s; // 1 .. K, and row i+1 has the same length (detect_pairs)
    T         v0[K], v1[K];
    unsigned  off[K]; // byte offset of B[c_k] inside a column (columns < 4 GB: checked by the host)
#pragma unroll
    for(int k = 0; k < K; k++)
    {
        v0[k] = T(0), v1[k] = T(0), off[k] = 0;
        if(k < len)
            v0[k] = val[s + k], v1[k] = val[e + k], off[k] = (unsigned)(col[s + k] - base) * (unsigned)sizeof(T);
    }
    const bool vec_store = c_aligned && (i % 2 == 0);
    auto       getc      = [&](const T *cp) {
        v2 c;
        if(vec_store)
            c = *reinterpret_cast<const v2 *>(cp);
        else
            c.x = cp[0], c.y = cp[1];
        return c;
    };
    // cin: the pair's C values when RC (already loaded), unused otherwise
    auto put = [&](T *cp, T a0, T a1, v2 cin) {
        const T z0 = alpha * a0, z1 = alpha * a1;
        v2      o;
        if constexpr(RC)
        {
            o.x = mm_fma(beta, cin.x, z0), o.y = mm_fma(beta, cin.y, z1);
            if(vec_store)
                *reinterpret_cast<v2 *>(cp) = o;
            else
                cp[0] = o.x, cp[1] = o.y;
            return;
        }
        if(z0 == T(0) || z1 == T(0)) // the sign of an exact zero is beta * C's (the reference computes 0 * C + z)
        {
            const v2 c = getc(cp);
            o.x = mm_fma(beta, c.x, z0), o.y = mm_fma(beta, c.y, z1);
            if(vec_store)
                *reinterpret_cast<v2 *>(cp) = o;
            else
                cp[0] = o.x, cp[1] = o.y;
        }
        else if(vec_store)
        {
            o.x = z0, o.y = z1;
            __builtin_nontemporal_store(o, reinterpret_cast<v2 *>(cp));
        }
        else
            cp[0] = z0, cp[1] = z1;
    };
    int j = j0;
    for(; j + U <= j1; j += U)
    {
        v2 b[U][K], cin[U] = {};
#pragma unroll
        for(int u = 0; u < U; u++)
        {
            if constexpr(RC)
                cin[u] = getc(C + (size_t)i + (size_t)(j + u) * ldc);
            const char *Bu = reinterpret_cast<const char *>(B + (size_t)(j + u) * ldb);
#pragma unroll
            for(int k = 0; k < K; k++)
                if(k < len)
                    __builtin_memcpy(&b[u][k], Bu + off[k], sizeof(v2));
        }
#pragma unroll
        for(int u = 0; u < U; u++)
        {
            T a0 = T(0), a1 = T(0);
#pragma unroll
            for(int k = 0; k < K; k++)
                if(k < len)
                    a0 = mm_fma(v0[k], b[u][k].x, a0), a1 = mm_fma(v1[k], b[u][k].y, a1);
            put(C + (size_t)i + (size_t)(j + u) * ldc, a0, a1, cin[u]);
        }
    }
    for(; j < j1; j++)
    {
        const char *Bu = reinterpret_cast<const char *>(B + (size_t)j * ldb);
        T           a0 = T(0), a1 = T(0);
        v2          cin = {};
        if constexpr(RC)
            cin = getc(C + (size_t)i + (size_t)j * ldc);
#pragma unroll
        for(int k = 0; k < K; k++)
            if(k < len)
            {
                v2 b;
                __builtin_memcpy(&b, Bu + off[k], sizeof(v2));
                a0 = mm_fma(v0[k], b.x, a0), a1 = mm_fma(v1[k], b.y, a1);
            }
        put(C + (size_t)i + (size_t)j * ldc, a0, a1, cin);
    }
}

// ---- the reference's KT kernels, reproduced for aoclsparse_?csrmm_kid(kid = 1, 2, 3) (round 3) ------------------------
// csrmm_col_kt / csrmm_row_kt (level3/aoclsparse_csrmm_kt.cpp:31-363) are what the reference's dispatcher runs for kid 1/2
// (256-bit vectors: PSZ = 4 doubles / 8 floats) and kid 3 (512-bit: 8 / 16), csrmm.hpp:779-833.  Their per-element
// arithmetic differs from the kid-0 kernels (vector lanes + horizontal sum; beta * C first for the row-major one), so a
// caller that pins a kid gets that order here -- bit for bit the FUSED build of the reference (oracle.c header), which
// tests/test_oracle_kt.py pins on the reference's own templates.  Plain kernels: a pinned kid asks for bits, not speed.
// column-major: a lane per (row, column); csrmm_kt.cpp:127-191
template <typename T, int PSZ>
__global__ __launch_bounds__(256) void csrmm_col_kt_kernel(int base, T alpha, aoclsparse_int m, const T *__restrict__ val,
                                                           const aoclsparse_int *__restrict__ col,
                                                           const aoclsparse_int *__restrict__ row_ptr,
                                                           const T *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                           T beta, T *__restrict__ C, aoclsparse_int ldc)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= m)
        return;
    const int s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
    const int nnz = e - s, mul = nnz / PSZ, rem = nnz - PSZ * mul;
    for(int j = blockIdx.y; j < n; j += gridDim.y)
    {
        const T *Bj  = B + (size_t)j * ldb;
        T        cij = T(0);
        if(mul)
        {
            T p[PSZ];
#pragma unroll
            for(int l = 0; l < PSZ; l++)
                p[l] = T(0);
            for(int k = s; k < e - rem; k += PSZ)
#pragma unroll
                for(int l = 0; l < PSZ; l++)
                    p[l] = mm_fma(val[k + l], Bj[col[k + l] - base], p[l]);
            cij = cij + kt_hsum<T, PSZ>(p); // "cij += cdot" with cij = 0 (:153-156)
        }
        for(int k = e - rem; k < e; k++)
            cij = mm_fma(val[k], Bj[col[k] - base], cij);
        cij *= alpha;
        T *cp = C + (size_t)i + (size_t)j * ldc;
        *cp   = mm_fma(beta, *cp, cij);
    }
}

// row-major: a lane per (row, column); csrmm_kt.cpp:244-356.  C * beta first, then the entries in CSR order: columns
// below n - n % PSZ take fma(alpha * a, b, c) (the vector statement), the last n % PSZ columns fma(a * b, alpha, c)
template <typename T, int PSZ>
__global__ __launch_bounds__(256) void csrmm_row_kt_kernel(int base, T alpha, aoclsparse_int m, const T *__restrict__ val,
                                                           const aoclsparse_int *__restrict__ col,
                                                           const aoclsparse_int *__restrict__ row_ptr,
                                                           const T *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                           T beta, T *__restrict__ C, aoclsparse_int ldc)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(i >= m || j >= n)
        return;
    const bool vec = j < n - n % PSZ;
    const int  s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
    T         *cp = C + (size_t)i * ldc + j;
    T          c  = *cp * beta;
    for(int k = s; k < e; k++)
    {
        const T a = val[k], b = B[(size_t)(col[k] - base) * ldb + j];
        c         = vec ? mm_fma(alpha * a, b, c) : mm_fma(a * b, alpha, c);
    }
    *cp = c;
}

template <typename T>
aoclsparse_status launch_csrmm_kt(hipStream_t s, aoclsparse_order order, int lanes, int base, T alpha, aoclsparse_int m,
                                  const T *val, const aoclsparse_int *col, const aoclsparse_int *row_ptr, const T *B,
                                  aoclsparse_int n, aoclsparse_int ldb, T beta, T *C, aoclsparse_int ldc)
{
    if(m <= 0 || n <= 0)
        return aoclsparse_status_success;
    constexpr int P256 = std::is_same<T, double>::value ? 4 : 8, P512 = 2 * P256;
    if(lanes != P256 && lanes != P512)
        return aoclsparse_status_internal_error;
    if(order == aoclsparse_order_column)
    {
        const dim3 grid((m + 255) / 256, std::min<aoclsparse_int>(n, 1024));
        if(lanes == P256)
            hipLaunchKernelGGL((csrmm_col_kt_kernel<T, P256>), grid, dim3(256), 0, s, base, alpha, m, val, col, row_ptr, B, n,
                               ldb, beta, C, ldc);
        else
            hipLaunchKernelGGL((csrmm_col_kt_kernel<T, P512>), grid, dim3(256), 0, s, base, alpha, m, val, col, row_ptr, B, n,
                               ldb, beta, C, ldc);
    }
    else
    {
        // rows on grid.y in chunks (65535 limit): the kernel takes row offsets through the pointers
        const int rows_per_launch = 65535 * 4;
        for(aoclsparse_int r0 = 0; r0 < m; r0 += rows_per_launch)
        {
            const aoclsparse_int mr = std::min<aoclsparse_int>(rows_per_launch, m - r0);
            const dim3           grid((n + 63) / 64, (mr + 3) / 4);
            if(lanes == P256)
                hipLaunchKernelGGL((csrmm_row_kt_kernel<T, P256>), grid, dim3(256), 0, s, base, alpha, mr, val, col, row_ptr + r0,
                                   B, n, ldb, beta, C + (size_t)r0 * ldc, ldc);
            else
                hipLaunchKernelGGL((csrmm_row_kt_kernel<T, P512>), grid, dim3(256), 0, s, base, alpha, mr, val, col, row_ptr + r0,
                                   B, n, ldb, beta, C + (size_t)r0 * ldc, ldc);
        }
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// dense layout change for the column-major detour of csrmm_api.cpp: src is R x N with element (r, c) at
// src[r*rs + c*cs]; dst gets it at dst[r*rd + c*cd].  64 x 64 tiles through LDS so that both sides move whole lines
// (tile rows run along whichever index is contiguous on each side).
template <typename T>
__global__ __launch_bounds__(256) void relayout_kernel(const T *__restrict__ src, long long rs, long long cs, T *__restrict__ dst,
                                                       long long rd, long long cd, aoclsparse_int R, aoclsparse_int N,
                                                       bool src_rows_contig)
{
    __shared__ T tile[64][65];
    const int    r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int    tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
    // read: the contiguous source index varies with tx
    for(int k = ty; k < 64; k += 4)
    {
        const int r = src_rows_contig ? r0 + tx : r0 + k;
        const int c = src_rows_contig ? c0 + k : c0 + tx;
        if(r < R && c < N)
            tile[r - r0][c - c0] = src[(long long)r * rs + (long long)c * cs];
    }
    __syncthreads();
    // write: the contiguous destination index is the other one
    for(int k = ty; k < 64; k += 4)
    {
        const int r = src_rows_contig ? r0 + k : r0 + tx;
        const int c = src_rows_contig ? c0 + tx : c0 + k;
        if(r < R && c < N)
            dst[(long long)r * rd + (long long)c * cd] = tile[r - r0][c - c0];
    }
}

// level3/aoclsparse_csrmm.hpp:361-427: beta == 0 stores exact zeros, otherwise C *= beta
template <typename T>
__global__ void scale_dense_kernel(T *C, aoclsparse_int inner, aoclsparse_int outer, aoclsparse_int ld, T beta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int o = blockIdx.y;
    if(i < inner && o < outer)
    {
        T *p = C + (size_t)o * ld + i;
        *p   = beta == T(0) ? T(0) : *p * beta;
    }
}

static int pow2_at_least(int v)
{
    int p = 1;
    while(p < v)
        p <<= 1;
    return p;
}

// Workgroups (of 4 rows) per XCD turn for a band of `band` rows: a line of W = band / 4 workgroups should be one round of 8 chunks.  The
// part of a chunk whose rows i +- band fall to ANOTHER XCD is |W - 8 c| / c: the candidate with the least of it wins, a multiple of 4
// is preferred while it stays under 0.2 (1000 rows: 32 over 31, measured 1.085 against 1.12 ms), and above 0.4 nothing is dealt (the
// neighbour line on the next XCD is the worst order there is: 1.30 ms against 1.19 for the contiguous eighth).
static int mm_deal_chunk(aoclsparse_int band, double rows_per_workgroup = 4.0)
{
    const double W = (double)band / rows_per_workgroup;
    const int    c0 = (int)(W / 8.0);
    int          best = 0;
    double       bs = 1e30;
    auto         score = [&](int c) { return c >= 1 ? std::abs(W - 8.0 * c) / c : 1e30; };
    for(int c : {c0, c0 + 1})
        if(score(c) < bs)
            bs = score(c), best = c;
    for(int c : {c0 / 4 * 4, c0 / 4 * 4 + 4})
        if(c >= 8 && score(c) <= 0.2)
        {
            best = c, bs = score(c);
            break;
        }
    return bs <= 0.4 ? best : 0;
}

template <typename T>
aoclsparse_status launch_csrmm(hipStream_t s, aoclsparse_order order, int base, T alpha, aoclsparse_int m,
                               aoclsparse_int /*k*/, const T *val, const aoclsparse_int *col,
                               const aoclsparse_int *row_ptr, const T *B, aoclsparse_int n,
                               aoclsparse_int ldb, T beta, T *C, aoclsparse_int ldc, const aoclsparse_int *grp,
                               aoclsparse_int ngroups, int group_rows, bool row_runs, const aoclsparse_int *run_order, int kt_lanes,
                               aoclsparse_int band)
{
    const bool kt = kt_lanes > 0;
    if(m <= 0 || n <= 0)
        return aoclsparse_status_success;
    // kt: the arithmetic of csrmm_row_kt, offered by the row-per-wave kernel only (anything else: not_implemented, the caller
    // runs the plain KT kernel)
    if(kt)
    {
        const bool vec = order == aoclsparse_order_row && (n % 2 == 0) && (ldb % 2 == 0) && (ldc % 2 == 0)
                         && (reinterpret_cast<uintptr_t>(B) % (2 * sizeof(T)) == 0)
                         && (reinterpret_cast<uintptr_t>(C) % (2 * sizeof(T)) == 0);
        if(!vec || n < 128 || (grp && ngroups > 0))
            return aoclsparse_status_not_implemented;
        const int chunk_kt = ((m + 3) / 4 + 7) / 8;
        hipLaunchKernelGGL((csrmm_row_wave_rc_kernel<T, true>), dim3(chunk_kt * 8, (n + 127) / 128), dim3(256), 0, s, base, alpha, m,
                           val, col, row_ptr, B, n, ldb, beta, C, ldc, chunk_kt, (int)(n % kt_lanes));
        MI355_HIP_TRY(hipGetLastError());
        return aoclsparse_status_success;
    }
    const bool readc = csrmm_reads_c(beta != T(0));
    // XCD-contiguous row order (every kernel): each XCD's L2 then serves the B rows its rows share.
    // Row-major n=256 on the 1000^2 Laplacian: 0.96 vs 1.21 ms.
    constexpr bool xcd = true;
    // Banded matrices (detect_row_runs: most rows reach `band` columns right of the diagonal -- a grid numbered line by line), row-per-
    // wavefront kernels: chunks of band / 8 rows are DEALT to the XCDs in turn, so that rows i and i +- band meet in one L2 (a line is one
    // round of the deal) while all eight XCDs stay inside the same line of B and C.  1000^2 Laplacian, C read, 256 / 128 columns, cold:
    // a contiguous eighth per XCD 1.19 / 0.61 ms, chunks of 32 workgroups 1.085 / 0.57; chunks that put row i +- band on the NEXT XCD
    // (10, 12, 18, 20, 24, 40 workgroups) 1.30-1.34 / 0.68-0.69 (profiles/r6/mm_deal_experiments.txt).
    int  deal = 0; // workgroups per XCD turn (0: a contiguous eighth per XCD)
    auto grid_x = [&](int nbx, int &chunk) {
        if(deal > 0)
        {
            chunk = MM_DEAL | deal;
            return (nbx + 8 * deal - 1) / (8 * deal) * (8 * deal);
        }
        chunk = xcd ? (nbx + 7) / 8 : 0;
        return xcd ? chunk * 8 : nbx;
    };
    int chunk = 0;
    if(order == aoclsparse_order_row)
    {
        const bool vec = (n % 2 == 0) && (ldb % 2 == 0) && (ldc % 2 == 0)
                         && (reinterpret_cast<uintptr_t>(B) % (2 * sizeof(T)) == 0)
                         && (reinterpret_cast<uintptr_t>(C) % (2 * sizeof(T)) == 0);
        const int lanes = vec ? n / 2 : n;
        const int tx    = lanes >= 128 ? 128 : pow2_at_least(lanes);
        const int ty    = 256 / tx;
        if(vec && n >= 32 && grp && ngroups > 0)
        {
            // banded matrices with row groups (MmGroups::band of the groups: a mesh of nodes numbered line by line): the deal again -- a
            // workgroup holds 4 / 8 / 16 groups of m / ngroups rows on average
            // (256 columns, C read / overwritten, same box: shell-like 2.38 / 2.07 -> 2.21 / 1.97 ms, flan-like 5.27 / 4.89 -> 4.62 / 4.36; 32
            // columns: flan-like 1.225 -> 1.173, shell-like unchanged; profiles/r6/mm_deal_experiments.txt)
            const double rows_per_group = (double)m / (double)ngroups;
            auto         deal_for = [&](int groups_per_wg) { deal = band >= 256 ? mm_deal_chunk(band, rows_per_group * groups_per_wg) : 0; };
            // accumulators are sized by the largest group the matrix actually has (2, 4 or 8 rows)
            auto go = [&](auto gr_tag) {
                constexpr int GR = decltype(gr_tag)::value;
                if(n >= 64)
                {
                    deal_for(8);
                    const int gx = grid_x((ngroups + 7) / 8, chunk);
                    hipLaunchKernelGGL((csrmm_rowgroup_sub_kernel<T, 32, GR>), dim3(gx, (n + 63) / 64), dim3(256), 0, s, base,
                                       alpha, ngroups, grp, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk);
                }
                else
                {
                    deal_for(16);
                    const int gx = grid_x((ngroups + 15) / 16, chunk);
                    hipLaunchKernelGGL((csrmm_rowgroup_sub_kernel<T, 16, GR>), dim3(gx, (n + 31) / 32), dim3(256), 0, s, base,
                                       alpha, ngroups, grp, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk);
                }
            };
            auto go2 = [&](auto gr_tag) {
                constexpr int GR = decltype(gr_tag)::value;
                deal_for(4);
                const int     gx = grid_x((ngroups + 3) / 4, chunk);
                // C read: its rows are requested up front (shell-like 2.37 -> 2.31 ms, flan-like 5.46 -> 5.19, same box, 256 columns)
                if(readc)
                    hipLaunchKernelGGL((csrmm_rowgroup2_kernel<T, GR, false, true>), dim3(gx, (n + 127) / 128), dim3(256), 0, s, base,
                                       alpha, ngroups, grp, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk,
                                       (const aoclsparse_int *)nullptr);
                else
                    hipLaunchKernelGGL((csrmm_rowgroup2_kernel<T, GR>), dim3(gx, (n + 127) / 128), dim3(256), 0, s, base, alpha,
                                       ngroups, grp, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk,
                                       (const aoclsparse_int *)nullptr);
            };
            if(n >= 128)
            {
                switch(group_rows)
                {
                case 1:
                case 2: go2(std::integral_constant<int, 2>{}); break;
                case 3: go2(std::integral_constant<int, 3>{}); break;
                case 4: go2(std::integral_constant<int, 4>{}); break;
                case 5: go2(std::integral_constant<int, 5>{}); break;
                case 6: go2(std::integral_constant<int, 6>{}); break;
                default: go2(std::integral_constant<int, CSRMM_GROUP>{}); break;
                }
            }
            else if(group_rows <= 2)
                go(std::integral_constant<int, 2>{});
            else if(group_rows <= 4)
                go(std::integral_constant<int, 4>{});
            else
                go(std::integral_constant<int, CSRMM_GROUP>{});
        }
        else if(vec && n >= 128 && row_runs && !readc) // (C read: the row-per-wave kernel is faster -- 1.17 vs 1.18 ms in round 2, and in
                                                       // round 3, both with the C row requested first, 1.10-1.16 vs 1.25-1.28 ms: profiles/r3/csrmm_row_wave_rc.txt)
        {
            constexpr int RUN = 8;
            const int     gx  = grid_x((m + 4 * RUN - 1) / (4 * RUN), chunk);
            // the 128-column chunks of a row block run together (chunk = fastest index inside an XCD) rather than one chunk
            // of every block, then the next: same time at 256 columns, 1.64 vs 1.77 ms at 512 (A is read once, the chunks
            // of a B row come in together).  Only in strip order: in plain row order it doubles the bytes between two
            // touches of a B row (1.08 vs 0.88 ms, FETCH 5.8 vs 3.7 GB).
            const int ny = (n + 127) / 128;
            if(run_order && chunk > 0 && (long long)gx * ny < (1LL << 31))
                hipLaunchKernelGGL((csrmm_row_run_kernel<T, RUN>), dim3(gx * ny, 1), dim3(256), 0, s, base, alpha, m, val, col,
                                   row_ptr, B, n, ldb, beta, C, ldc, readc, mmw(chunk), run_order, ny);
            else
                hipLaunchKernelGGL((csrmm_row_run_kernel<T, RUN>), dim3(gx, ny), dim3(256), 0, s, base, alpha, m, val, col,
                                   row_ptr, B, n, ldb, beta, C, ldc, readc, mmw(chunk), run_order, 0);
        }
        else if(vec && n >= 128)
        {
            if(band >= 256 && readc)
                deal = mm_deal_chunk(band);
            const int gx = grid_x((m + 3) / 4, chunk);
            bool      wide = false;
            if constexpr(std::is_same<T, float>::value)
                wide = readc && n >= 256 && n % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0
                       && reinterpret_cast<uintptr_t>(C) % 16 == 0;
            if(wide)
            {
                if constexpr(std::is_same<T, float>::value)
                    hipLaunchKernelGGL(csrmm_row_wave_rc4_kernel, dim3(gx, (n + 255) / 256), dim3(256), 0, s, base, alpha, m, val, col,
                                       row_ptr, B, n, ldb, beta, C, ldc, mmw(chunk));
            }
            else if(readc)
                hipLaunchKernelGGL((csrmm_row_wave_rc_kernel<T>), dim3(gx, (n + 127) / 128), dim3(256), 0, s, base, alpha,
                                   m, val, col, row_ptr, B, n, ldb, beta, C, ldc, mmw(chunk));
            else
                hipLaunchKernelGGL((csrmm_row_wave_kernel<T>), dim3(gx, (n + 127) / 128), dim3(256), 0, s, base, alpha,
                                   m, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, mmw(chunk));
        }
        else
        {
            const int gx = grid_x((m + ty - 1) / ty, chunk);
            dim3      block(tx, ty), grid(gx, (lanes + tx - 1) / tx);
            if(vec)
                hipLaunchKernelGGL((csrmm_row_kernel<T, true>), grid, block, 0, s, base, alpha, m, val, col, row_ptr,
                                   B, n, ldb, beta, C, ldc, readc, chunk);
            else
                hipLaunchKernelGGL((csrmm_row_kernel<T, false>), grid, block, 0, s, base, alpha, m, val, col,
                                   row_ptr, B, n, ldb, beta, C, ldc, readc, chunk);
        }
    }
    else
    {
        const int gx = grid_x((m + 255) / 256, chunk);
        dim3      block(256), grid(gx, (n + cm_cols() - 1) / cm_cols());
        hipLaunchKernelGGL((csrmm_col_kernel<T>), grid, block, 0, s, base, alpha, m, val, col, row_ptr, B, n, ldb,
                           beta, C, ldc, readc, chunk, (const aoclsparse_int *)nullptr);
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// the row-group kernel with row-major B (packed scratch of the column-major detour) and COLUMN-major C: see CCOL above.
// Applies when the handle has row groups, n >= 32 and even, B 16-byte aligned; false = the caller takes the full detour.
// n < 128: the sub-wave row-group kernels (a group per 32 / 16 lanes), same store phase.
template <typename T>
bool csrmm_groups_ccol_applies(aoclsparse_int n, aoclsparse_int ldb, const T *B)
{
    return n >= 32 && n % 2 == 0 && ldb % 2 == 0 && reinterpret_cast<uintptr_t>(B) % (2 * sizeof(T)) == 0;
}

template <typename T>
aoclsparse_status launch_csrmm_groups_ccol(hipStream_t s, int base, T alpha, const T *val, const aoclsparse_int *col,
                                           const aoclsparse_int *row_ptr, const T *B, aoclsparse_int n, aoclsparse_int ldb,
                                           T beta, T *C, aoclsparse_int ldc, const aoclsparse_int *grp, aoclsparse_int ngroups,
                                           int group_rows, aoclsparse_int m, aoclsparse_int band)
{
    if(ngroups <= 0 || n <= 0)
        return aoclsparse_status_success;
    const bool readc = csrmm_reads_c(beta != T(0));
    // the order word and the grid for nbx workgroups of `groups` row groups each: the band of the groups dealt to the XCDs (launch_csrmm),
    // else a contiguous eighth per XCD
    auto order_of = [&](int nbx, int groups, int &gx) {
        const int deal = band >= 256 && m > 0 ? mm_deal_chunk(band, (double)m / (double)ngroups * groups) : 0;
        if(deal > 0)
        {
            gx = (nbx + 8 * deal - 1) / (8 * deal) * (8 * deal);
            return MM_DEAL | deal;
        }
        const int chunk = (nbx + 7) / 8;
        gx              = chunk * 8;
        return chunk;
    };
    if(n < 128)
    {
        auto gosub = [&](auto gr_tag) {
            constexpr int GR = decltype(gr_tag)::value;
            if(n >= 64)
            {
                int       gx  = 0;
                const int chunk = order_of((int)((ngroups + 7) / 8), 8, gx);
                hipLaunchKernelGGL((csrmm_rowgroup_sub_kernel<T, 32, GR, true>), dim3(gx, (n + 63) / 64), dim3(256), 0, s,
                                   base, alpha, ngroups, grp, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk);
            }
            else
            {
                int       gx  = 0;
                const int chunk = order_of((int)((ngroups + 15) / 16), 16, gx);
                hipLaunchKernelGGL((csrmm_rowgroup_sub_kernel<T, 16, GR, true>), dim3(gx, (n + 31) / 32), dim3(256), 0, s,
                                   base, alpha, ngroups, grp, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk);
            }
        };
        if(group_rows <= 2)
            gosub(std::integral_constant<int, 2>{});
        else if(group_rows <= 4)
            gosub(std::integral_constant<int, 4>{});
        else
            gosub(std::integral_constant<int, CSRMM_GROUP>{});
        MI355_HIP_TRY(hipGetLastError());
        return aoclsparse_status_success;
    }
    int        gx2   = 0;
    const int  chunk = order_of((int)((ngroups + 3) / 4), 4, gx2);
    auto       go  = [&](auto gr_tag) {
        constexpr int GR = decltype(gr_tag)::value;
        hipLaunchKernelGGL((csrmm_rowgroup2_kernel<T, GR, true>), dim3(gx2, (n + 127) / 128), dim3(256), 0, s, base, alpha,
                           ngroups, grp, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk,
                           (const aoclsparse_int *)nullptr);
    };
    switch(group_rows)
    {
    case 1:
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 3: go(std::integral_constant<int, 3>{}); break;
    case 4: go(std::integral_constant<int, 4>{}); break;
    case 5: go(std::integral_constant<int, 5>{}); break;
    case 6: go(std::integral_constant<int, 6>{}); break;
    default: go(std::integral_constant<int, CSRMM_GROUP>{}); break;
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_scale_dense(hipStream_t s, aoclsparse_order order, T *C, aoclsparse_int m,
                                     aoclsparse_int n, aoclsparse_int ld, T beta)
{
    const aoclsparse_int outer = order == aoclsparse_order_column ? n : m;
    const aoclsparse_int inner = order == aoclsparse_order_column ? m : n;
    if(outer <= 0 || inner <= 0)
        return aoclsparse_status_success;
    // gridDim.y is limited to 65535: walk the outer dimension in slabs
    for(aoclsparse_int o0 = 0; o0 < outer; o0 += 65535)
    {
        const aoclsparse_int cnt = outer - o0 < 65535 ? outer - o0 : 65535;
        hipLaunchKernelGGL((scale_dense_kernel<T>), dim3((inner + 255) / 256, cnt), dim3(256), 0, s,
                           C + (size_t)o0 * ld, inner, cnt, ld, beta);
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}


// column-major (ld) <-> packed row-major (ld = N) copies of an R x N dense matrix
template <typename T>
aoclsparse_status launch_relayout(hipStream_t s, bool to_row_major, const T *src, T *dst, aoclsparse_int R, aoclsparse_int N,
                                  aoclsparse_int ld)
{
    if(R <= 0 || N <= 0)
        return aoclsparse_status_success;
    const dim3 grid((R + 63) / 64, (N + 63) / 64), block(256);
    if(to_row_major) // src column-major: rows contiguous
        hipLaunchKernelGGL((relayout_kernel<T>), grid, block, 0, s, src, 1LL, (long long)ld, dst, (long long)N, 1LL, R, N, true);
    else // src packed row-major: columns contiguous
        hipLaunchKernelGGL((relayout_kernel<T>), grid, block, 0, s, src, (long long)N, 1LL, dst, 1LL, (long long)ld, R, N, false);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// Row-major product over the row blocks of an SpMV plan (csrmm_tile_kernel): vec-able operands with n < 128 only.
template <typename T>
bool csrmm_tiled_applies(aoclsparse_int n, aoclsparse_int ldb, aoclsparse_int ldc, const T *B, const T *C)
{
    return n >= 2 && n < 128 && (n % 2 == 0) && (ldb % 2 == 0) && (ldc % 2 == 0)
           && (reinterpret_cast<uintptr_t>(B) % (2 * sizeof(T)) == 0) && (reinterpret_cast<uintptr_t>(C) % (2 * sizeof(T)) == 0);
}

template <typename T>
aoclsparse_status launch_csrmm_tiled(hipStream_t s, int base, T alpha, const T *val, const aoclsparse_int *col,
                                     const aoclsparse_int *row_ptr, const aoclsparse_int *blocks, aoclsparse_int nblocks,
                                     int tile, aoclsparse_int max_row_nnz, const T *B, aoclsparse_int n, aoclsparse_int ldb,
                                     T beta, T *C, aoclsparse_int ldc, int kt_lanes, bool launch_order)
{
    if(nblocks <= 0 || n <= 0)
        return aoclsparse_status_success;
    const bool kt      = kt_lanes > 0;
    const int  kt_tail = kt ? (int)(n % kt_lanes) : 0; // csrmm_row_kt's scalar tail columns
    const bool readc = kt || csrmm_reads_c(beta != T(0)); // (the KT arithmetic always reads C: c = c * beta comes first)
    // XCD-contiguous block order, as the other csrmm kernels -- unless the blocks follow the lines of a band (MmGroups::slab_blocks): then
    // launch order IS the deal of the lines' eighths to the XCDs
    const bool xcd   = !launch_order;
    const int  chunk = xcd ? (nblocks + 7) / 8 : 0;
    // (UR, NB) = rows in flight per 16-lane sub-wave x B-row loads per row and step.  Round-3 sweep on the 32-column slab of
    // the 1000^2 Laplacian (tools/history/exp_r3_slab3.sh, profiles/r3/slab_shapes.txt; beta = 0 overwrite / C read): (2, 8) 0.130 /
    // 0.174 ms, (2, 6) 0.126-0.128 / 0.166-0.168, (1, 8) 0.127-0.133 / 0.165-0.175, (3, 6) 0.133-0.137 / 0.170-0.174, (4, 6) 0.154-0.158 /
    // 0.182-0.184, (4, 8) 0.167 / 0.193-0.198: two rows in flight, and no more load slots than the rows have entries.
    // Occupancy is not the lever (tools/history/exp_r3_slab5.sh, profiles/r3/slab_occupancy.txt): the kernel sits at 4 waves / SIMD
    // (98-110 VGPRs); __launch_bounds__(256, 5) without spills (NB = 5 / 6, overwrite) measures the same 0.123-0.129 ms, with
    // spills (C read) 0.19-0.29 ms, (256, 6) 0.14-0.51 ms.  NB = 5 for rows of <= 5 entries is worth 1-2 % (0.163-0.173 vs
    // 0.167-0.174 ms C read, 0.1225-0.126 vs 0.124-0.1285 overwrite).
    auto       go3   = [&](auto tile_tag, auto nb_tag, auto rc_tag) {
        constexpr int  TILE = decltype(tile_tag)::value, NB = decltype(nb_tag)::value;
        constexpr bool RC   = decltype(rc_tag)::value;
        const dim3     grid(xcd ? chunk * 8 : (nblocks + 7) / 8 * 8, (n + 31) / 32); // (a multiple of 8: the column parts keep the XCDs)
        const size_t   trace_slots = (size_t)grid.x;
        static const char *trace_path = getenv("AOCLSPARSE_MI355_MM_TRACE");
        if constexpr(std::is_same<T, double>::value && TILE == 1024 && NB == 5)
        {
            unsigned long long *trace = nullptr;
            if(trace_path && !kt && hipMalloc(&trace, sizeof(unsigned long long) * 8 * trace_slots) == hipSuccess)
            {
                (void)hipMemsetAsync(trace, 0, sizeof(unsigned long long) * 8 * trace_slots, s);
                hipLaunchKernelGGL((csrmm_tile_kernel<T, 16, TILE, 2, NB, RC, false, true>), grid, dim3(256), 0, s, base, alpha, val,
                                   col, row_ptr, blocks, nblocks, B, n, ldb, beta, C, ldc, readc, mmw(chunk), trace);
                std::vector<unsigned long long> host(8 * (size_t)nblocks);
                if(hipStreamSynchronize(s) == hipSuccess
                   && hipMemcpy(host.data(), trace, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) == hipSuccess)
                    if(FILE *f = fopen(trace_path, "wb"))
                    {
                        fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
                        fclose(f);
                    }
                (void)hipFree(trace);
                return;
            }
        }
        if constexpr(RC)
        {
            if(kt)
            {
                hipLaunchKernelGGL((csrmm_tile_kernel<T, 16, TILE, 2, NB, true, true>), grid, dim3(256), 0, s, base, alpha, val, col,
                                   row_ptr, blocks, nblocks, B, n, ldb, beta, C, ldc, readc, mmw(chunk), (unsigned long long *)nullptr,
                                   kt_tail);
                return;
            }
        }
        hipLaunchKernelGGL((csrmm_tile_kernel<T, 16, TILE, 2, NB, RC>), grid, dim3(256), 0, s, base, alpha, val, col, row_ptr,
                           blocks, nblocks, B, n, ldb, beta, C, ldc, readc, mmw(chunk));
    };
    auto go2 = [&](auto tile_tag, auto nb_tag) {
        if(readc)
            go3(tile_tag, nb_tag, std::true_type{});
        else
            go3(tile_tag, nb_tag, std::false_type{});
    };
    auto go = [&](auto tile_tag) {
        if(max_row_nnz > 0 && max_row_nnz <= 5)
            go2(tile_tag, std::integral_constant<int, 5>{});
        else if(max_row_nnz > 0 && max_row_nnz <= 6)
            go2(tile_tag, std::integral_constant<int, 6>{});
        else
            go2(tile_tag, std::integral_constant<int, 8>{});
    };
    switch(tile & ~1)
    {
    case 512: go(std::integral_constant<int, 512>{}); break;
    case 1024: go(std::integral_constant<int, 1024>{}); break;
    case 2048: go(std::integral_constant<int, 2048>{}); break;
    default: return aoclsparse_status_internal_error;
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template <typename T>
aoclsparse_status launch_csrmm_colpair(hipStream_t s, int base, T alpha, aoclsparse_int npairs,
                                       const aoclsparse_int *pair_first, aoclsparse_int nsingles,
                                       const aoclsparse_int *single_rows, const T *val, const aoclsparse_int *col,
                                       const aoclsparse_int *row_ptr, aoclsparse_int max_row_nnz, const T *B, aoclsparse_int n,
                                       aoclsparse_int ldb, T beta, T *C, aoclsparse_int ldc)
{
    if(n <= 0)
        return aoclsparse_status_success;
    const bool readc     = csrmm_reads_c(beta != T(0));
    const bool c_aligned = reinterpret_cast<uintptr_t>(C) % (2 * sizeof(T)) == 0 && ldc % 2 == 0;
    const dim3 block(256);
    if(npairs > 0)
    {
        const int nbx = (npairs + 255) / 256, chunk = (nbx + 7) / 8;
        // (K, U) = entries cached x columns per step: round-3 sweep on the 32-column slab, beta = 0 overwrite / C read
        // (tools/history/exp_r3_slab4.sh, profiles/r3/slab_colmajor_shapes.txt): (8, 4) 0.190-0.192 / 0.240-0.242 ms, (8, 2) 0.177-0.187 /
        // 0.243, (6, 4) 0.193-0.195 / 0.243-0.245, (6, 2) 0.190 / 0.242-0.244 -- flat; at 256 columns (6, 4) LOSES (1.20 vs 1.09 ms).
        // What did help is requesting C with the step's B values when it is read: 256 columns, beta != 0: 1.58 -> 1.43 ms.
        if(readc)
            hipLaunchKernelGGL((csrmm_colpair_kernel<T, CM_K, 4, true>), dim3(chunk * 8, (n + cm_cols() - 1) / cm_cols()), block, 0, s,
                               base, alpha, npairs, pair_first, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, c_aligned, chunk);
        else
            hipLaunchKernelGGL((csrmm_colpair_kernel<T, CM_K, 4, false>), dim3(chunk * 8, (n + cm_cols() - 1) / cm_cols()), block, 0, s,
                               base, alpha, npairs, pair_first, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, c_aligned, chunk);
    }
    if(nsingles > 0)
    {
        const int nbx = (nsingles + 255) / 256, chunk = (nbx + 7) / 8;
        hipLaunchKernelGGL((csrmm_col_kernel<T>), dim3(chunk * 8, (n + cm_cols() - 1) / cm_cols()), block, 0, s, base, alpha,
                           nsingles, val, col, row_ptr, B, n, ldb, beta, C, ldc, readc, chunk, single_rows);
    }
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

#define MI355_INST_MM(T)                                                                                     \
    template aoclsparse_status launch_csrmm<T>(hipStream_t, aoclsparse_order, int, T, aoclsparse_int,        \
                                               aoclsparse_int, const T *, const aoclsparse_int *,             \
                                               const aoclsparse_int *, const T *, aoclsparse_int,             \
                                               aoclsparse_int, T, T *, aoclsparse_int, const aoclsparse_int *, \
                                               aoclsparse_int, int, bool, const aoclsparse_int *, int,        \
                                               aoclsparse_int);                                               \
    template aoclsparse_status launch_scale_dense<T>(hipStream_t, aoclsparse_order, T *, aoclsparse_int,     \
                                                     aoclsparse_int, aoclsparse_int, T);                     \
    template aoclsparse_status launch_relayout<T>(hipStream_t, bool, const T *, T *, aoclsparse_int,         \
                                                  aoclsparse_int, aoclsparse_int);                            \
    template aoclsparse_status launch_csrmm_colpair<T>(hipStream_t, int, T, aoclsparse_int, const aoclsparse_int *, \
                                                       aoclsparse_int, const aoclsparse_int *, const T *,      \
                                                       const aoclsparse_int *, const aoclsparse_int *, aoclsparse_int, const T *, \
                                                       aoclsparse_int, aoclsparse_int, T, T *, aoclsparse_int); \
    template bool csrmm_tiled_applies<T>(aoclsparse_int, aoclsparse_int, aoclsparse_int, const T *, const T *); \
    template bool csrmm_groups_ccol_applies<T>(aoclsparse_int, aoclsparse_int, const T *);                     \
    template aoclsparse_status launch_csrmm_groups_ccol<T>(hipStream_t, int, T, const T *, const aoclsparse_int *, \
                                                           const aoclsparse_int *, const T *, aoclsparse_int,  \
                                                           aoclsparse_int, T, T *, aoclsparse_int,             \
                                                           const aoclsparse_int *, aoclsparse_int, int,        \
                                                           aoclsparse_int, aoclsparse_int);                    \
    template aoclsparse_status launch_csrmm_tiled<T>(hipStream_t, int, T, const T *, const aoclsparse_int *,   \
                                                     const aoclsparse_int *, const aoclsparse_int *,          \
                                                     aoclsparse_int, int, aoclsparse_int, const T *, aoclsparse_int, \
                                                     aoclsparse_int, T, T *, aoclsparse_int, int, bool);
MI355_INST_MM(double)
MI355_INST_MM(float)
template aoclsparse_status launch_csrmm_kt<double>(hipStream_t, aoclsparse_order, int, int, double, aoclsparse_int, const double *,
                                                   const aoclsparse_int *, const aoclsparse_int *, const double *, aoclsparse_int,
                                                   aoclsparse_int, double, double *, aoclsparse_int);
template aoclsparse_status launch_csrmm_kt<float>(hipStream_t, aoclsparse_order, int, int, float, aoclsparse_int, const float *,
                                                  const aoclsparse_int *, const aoclsparse_int *, const float *, aoclsparse_int,
                                                  aoclsparse_int, float, float *, aoclsparse_int);

} // namespace mi355

extern "C" aoclsparse_status mi355_dcsrmm(void *stream, aoclsparse_int order, aoclsparse_int base, double alpha,
                                          aoclsparse_int m, aoclsparse_int k, const double *val,
                                          const aoclsparse_int *col, const aoclsparse_int *row_ptr,
                                          const double *B, aoclsparse_int n, aoclsparse_int ldb, double beta,
                                          double *C, aoclsparse_int ldc)
{
    if(!val || !col || !row_ptr || !B || !C)
        return aoclsparse_status_invalid_pointer;
    if(m < 0 || n < 0 || k < 0)
        return aoclsparse_status_invalid_size;
    if((order != aoclsparse_order_row && order != aoclsparse_order_column) || (base != 0 && base != 1))
        return aoclsparse_status_invalid_value;
    return mi355::launch_csrmm<double>((hipStream_t)stream, (aoclsparse_order)order, base, alpha, m, k, val, col,
                                       row_ptr, B, n, ldb, beta, C, ldc, nullptr, 0, 0, false, nullptr, 0, 0);
}
