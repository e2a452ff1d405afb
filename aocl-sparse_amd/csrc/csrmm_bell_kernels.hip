// csrmm_bell_kernels.hip -- row-major C = alpha*A*B + beta*C for BLOCK-DENSE A on the matrix cores, gfx950.  Round 4
// (BASELINE north_star: "an ELL/blocked-ELL variant that feeds MFMA only where nnz/row is uniform enough to form dense tiles").
//
// Where it applies: fp64 matrices whose 16 x 16 tiles are at least half full (multi-dof finite-element / block-structured
// matrices: 16 unknowns per node gives dense 16 x 16 couplings).  There csrmm is no longer HBM-bound -- 2 * nnz * n flops
// against ~(8 * nnz + 16 * m * n) bytes is ~12 flop/B at 112 entries per row and 256 columns, the fp64 ridge of this part
// (78.6 TFLOP/s over ~6 TB/s) -- and the lane-per-column kernels reach 10-11 TFLOP/s (issue-bound: one v_fma_f64 per 64
// FMAs plus the operand traffic).  v_mfma_f64_16x16x4_f64 does 1,024 FMAs per instruction at the same peak rate.
//
// Format (built once per handle, csrmm_api.cpp: build_bell): blocked ELL, 16 x 16 blocks, `width` slots per block row, block
// columns ascending, values dense per block in the A-operand order of the instruction.
// Kernel: a wavefront owns one block row and up to four 16-column tiles of C (64 columns).  Per block: 4 coalesced loads of the
// A fragments (512 B each, shared by the four tiles), 4 loads of 128-byte B row segments per tile, 4 MFMAs per tile; the loads
// of block s+1 are issued before the MFMAs of block s.
//
// Arithmetic: the instruction accumulates k = 4t .. 4t+3 in order as one FMA chain per element (bit-identical to fma(): measured,
// tools/history/mfma_f64_probe.hip, profiles/r2/mfma_f64_probe.jsonl), blocks are walked in ascending column order, so every C element
// is the reference's chain over its row in CSR order (csrmm.hpp:69-85) with the tile's explicit zeros interleaved:
// fma(0, b, sum) == sum for every finite b (a sum that starts at +0 is never -0).  An Inf / NaN in B at a position the row does
// not store turns 0 * Inf into NaN inside the tile -- which the reference's CSR kernel never computes.  Round 5: an
// element whose accumulator comes out non-finite is RECOMPUTED from the CSR arrays by its lane (the row's chain without the
// padding, csrmm.hpp:69-85), so the result is the reference's for every B; finite products pay one v_cmp_class per element.
// build_bell requires sorted rows.
// beta == 0 with C read (the reference's 0 * C, the default): the C tile is requested behind the first block's operands and
// reduced to ONE predicate per element -- "is it finite" -- as soon as it lands; a finite C contributes exactly nothing
// (fma(0, c, z) == z for z != 0), so the closing store does not wait for memory.  Elements with a non-finite C, or z == 0 (the
// sign of the zero depends on C), re-read C and take the reference's fma.  Rounds 3-4 read C at the end of the block row: a
// dependent round trip per block row behind a 3 us MFMA chain (1.29 vs 0.93 ms with C overwritten).
#include "internal.hpp"
#include "mm_order.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstdint>

namespace mi355
{

namespace
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));

    // the reference's chain of C element (row, cj) over the CSR row, no padding (csrmm.hpp:69-85): the slow path of an element whose
    // tile accumulator is not finite
    template <bool COLMAJ>
    __device__ __noinline__ double bell_exact_element(int row, int cj, int base, const aoclsparse_int *__restrict__ rp,
                                                      const aoclsparse_int *__restrict__ ci, const double *__restrict__ cv,
                                                      const double *__restrict__ B, aoclsparse_int ldb)
    {
        double sum = 0.0;
        for(int p = rp[row] - base; p < rp[row + 1] - base; p++)
        {
            const size_t kk = (size_t)(ci[p] - base);
            sum             = fma(cv[p], COLMAJ ? B[(size_t)cj * ldb + kk] : B[kk * ldb + cj], sum);
        }
        return sum;
    }
    __device__ __forceinline__ bool bell_finite(double v)
    {
        return __builtin_isfinite(v);
    }

    // Which block row and which column part a wavefront takes.  Without an order list: wavefront g of the launch takes block row g / wpr
    // (wpr wavefronts per block row).  With BellPlan::order (build_bell's model of the eight L2s, csrmm_api.cpp): the dispatcher hands
    // workgroup w to XCD w % 8, so the wavefronts of XCD x form a stream of their own -- wavefront 4 (w / 8) + wv of it takes position
    // p = that / wpr of XCD x's list, order[8 p + x] (-1: the list of this XCD has ended).
    __device__ __forceinline__ bool bell_place(unsigned w, int wv, int wpr, int nbr, const aoclsparse_int *__restrict__ order, int order_len,
                                               int &br, int &cw)
    {
        if(order)
        {
            const long gx = (long)(w >> 3) * 4 + wv, p = gx / wpr;
            cw            = (int)(gx % wpr);
            br            = p < order_len ? __builtin_amdgcn_readfirstlane(order[p * 8 + (w & 7u)]) : -1;
            return br >= 0 && br < nbr;
        }
        const long g = (long)w * 4 + wv;
        br = (int)(g / wpr), cw = (int)(g % wpr);
        return br < nbr;
    }

    // Operand fragments: lane -> (i or j = lane % 16, k = lane / 16), one double per lane.  Where the 4 result registers of a
    // lane sit in the 16 x 16 tile is NOT assumed: every wavefront asks the instruction itself with two extra MFMAs
    // (D = [i] and D = [j]: A = column of row numbers x B = row of ones, and the transpose) -- 2 of ~114 per block row.
    // FULL: every tile column of every wavefront is < n and the column count of A is a multiple of 16: no masks on the B loads
    // RC: 0 C is overwritten (opt-in mode, beta == 0); 1 C is read and used (beta != 0); 2 beta == 0 with C read (the default:
    // the reference's 0 * C) -- the tile of C becomes a finite / not-finite predicate per element early on
    // CW (with WIDE): tiles 2w and 2w+1 hold the even and the odd columns of a 32-column stretch, so a lane's two results are
    // ADJACENT in a row of C: one 16-byte access per pair (16-byte aligned C, even ldc) instead of two 8-byte ones that each use
    // every other double of the lines they touch
    template <int NT, int RC, bool WIDE, bool FULL, bool CW>
    __global__ __launch_bounds__(256) void csrmm_bell_mfma_kernel(double alpha, aoclsparse_int m, aoclsparse_int k, aoclsparse_int nbr,
                                                                  aoclsparse_int width, const double *__restrict__ val,
                                                                  const aoclsparse_int *__restrict__ bcol,
                                                                  const double *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                                  double beta, double *__restrict__ C, aoclsparse_int ldc,
                                                                  int waves_per_row, const aoclsparse_int *__restrict__ order, int order_len, int base, const aoclsparse_int *__restrict__ rp,
                                                                  const aoclsparse_int *__restrict__ ci, const double *__restrict__ cv)
    {
        // (block rows in launch order: giving every XCD a contiguous eighth of them -- the rule of the HBM-bound kernels here --
        // takes a fifth of the fabric traffic away and is SLOWER: 0.93 -> 0.98 ms in round 4, 1.115 -> 1.222 (C read) / 0.922 ->
        // 0.951 ms in round 5, profiles/r4/bell_experiments.txt, profiles/r5/bell_experiments.txt)
        const int wv   = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int  wpr = waves_per_row & ~MM_DESCENDING; // (bit 30 of the word: workgroups in descending order, mm_order.hpp)
        int br, cw;
        if(!bell_place(mm_linear_index(waves_per_row), wv, wpr, nbr, order, order_len, br, cw))
            return;
        const int lane = threadIdx.x & 63, jl = lane & 15, kq = lane >> 4;
        const int j0   = cw * NT * 16;
        v4d       acc[NT];
#pragma unroll
        for(int u = 0; u < NT; u++)
            acc[u] = (v4d){0.0, 0.0, 0.0, 0.0};
        const v4d zero4 = (v4d){0.0, 0.0, 0.0, 0.0};
        const v4d drow  = __builtin_amdgcn_mfma_f64_16x16x4f64(kq == 0 ? (double)jl : 0.0, kq == 0 ? 1.0 : 0.0, zero4, 0, 0, 0);
        const v4d dcol  = __builtin_amdgcn_mfma_f64_16x16x4f64(kq == 0 ? 1.0 : 0.0, kq == 0 ? (double)jl : 0.0, zero4, 0, 0, 0);
        // Column of tile u that operand lane jl feeds (and result column index jl lands in).  WIDE: tiles 2w and 2w+1 take the even
        // and the odd columns of a 32-column stretch, so that ONE 16-byte load per lane feeds both tiles (half the vector-memory
        // instructions: 8 B-loads + 2 A-loads per block instead of 16 + 4); needs 16-byte aligned B rows and an even n.
        auto colof = [&](int u, int jx) { return WIDE ? j0 + 32 * (u / 2) + 2 * jx + (u & 1) : j0 + 16 * u + jx; };
        bool colok[NT]; // (operand side: the B fragment's column of this lane)
#pragma unroll
        for(int u = 0; u < NT; u++)
            colok[u] = colof(u, jl) < n;
        const double         *vb = val + (size_t)br * width * 256;
        const aoclsparse_int *cb = bcol + (size_t)br * width;
        // number of stored blocks of this block row (ascending columns, empty slots last)
        int nblk = 0;
        while(nblk < width && cb[nblk] >= 0)
            nblk++;
        // Operands of block s+1 are requested before the MFMAs of block s.  (Fetching the A fragments -- a pure HBM stream --
        // three blocks ahead gained 3 % without C read and cost 100 VGPRs: profiles/r4/bell_experiments.txt.)
        // A slot past the last block is CLAMPED to it, not skipped: the loads of the look-ahead are then issued on every path, and
        // the waits the compiler places in front of the MFMAs count them exactly.  (With `if(s >= nblk) return;` here the number of
        // younger loads in flight depended on the path, the MFMAs of block s waited for vmcnt(1) / vmcnt(0) -- i.e. for the
        // operands of block s + 1 as well -- and the look-ahead hid nothing: round 5, profiles/r5/bell_experiments.txt.)  The
        // operands of the clamped slot are never multiplied: the loop ends first.
        auto fetch_a = [&](int s, double (&a)[4]) {
            const double *vs = vb + (size_t)(s < nblk ? s : nblk - 1) * 256;
            if constexpr(WIDE)
            {
                // A fragments of t = 2p, 2p+1 sit side by side (build_bell's layout): one 16-byte load per pair
                const v2d a01 = *reinterpret_cast<const v2d *>(vs + 2 * lane), a23 = *reinterpret_cast<const v2d *>(vs + 128 + 2 * lane);
                a[0] = a01.x, a[1] = a01.y, a[2] = a23.x, a[3] = a23.y;
            }
            else
            {
#pragma unroll
                for(int t = 0; t < 4; t++)
                    a[t] = vs[(t / 2) * 128 + 2 * lane + (t & 1)];
            }
        };
        // B operand of (block column bc, fragment t, tile u): B[(16 bc + 4 t + kq) * ldb + colof(u, jl)] = a wave-uniform base
        // (16 bc ldb + j0) + a per-lane 32-bit offset that does not depend on the block + a compile-time tile offset
        unsigned relb[4];
#pragma unroll
        for(int t = 0; t < 4; t++)
            relb[t] = (unsigned)(4 * t + kq) * (unsigned)ldb + (unsigned)(WIDE ? 2 * jl : jl);
        const double *Bj = B + j0;
        auto fetch_b = [&](int s, double (&b)[NT][4]) {
            const int     bc = __builtin_amdgcn_readfirstlane(cb[s < nblk ? s : nblk - 1]);
            const double *bs = Bj + (size_t)bc * 16 * (size_t)ldb;
#pragma unroll
            for(int t = 0; t < 4; t++)
            {
                const bool rok = FULL || bc * 16 + 4 * t + kq < k; // the last block column when k is not a multiple of 16
                const double *bp = bs + (rok ? relb[t] : 0u);
                if constexpr(WIDE)
                {
#pragma unroll
                    for(int w2 = 0; w2 < NT / 2; w2++)
                    {
                        v2d x = (v2d){0.0, 0.0};
                        if(FULL || (rok && colok[2 * w2])) // (n is even: the pair is in or out together)
                            x = *reinterpret_cast<const v2d *>(bp + 32 * w2);
                        b[2 * w2][t] = x.x, b[2 * w2 + 1][t] = x.y;
                    }
                }
                else
                {
#pragma unroll
                    for(int u = 0; u < NT; u++)
                        b[u][t] = (FULL || (rok && colok[u])) ? bp[16 * u] : 0.0;
                }
            }
        };
        auto mac = [&](const double (&a)[4], const double (&b)[NT][4]) {
#pragma unroll
            for(int t = 0; t < 4; t++)
#pragma unroll
                for(int u = 0; u < NT; u++)
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[u][t], acc[u], 0, 0, 0);
        };
        double a0[4], a1[4], b0[NT][4], b1[NT][4];
        if(nblk > 0)
        {
            fetch_a(0, a0);
            fetch_b(0, b0);
        }
        // C is read (beta != 0, or the reference's 0 * C): requested behind the first block's operands, used after the last
        // block's MFMAs -- the closing read-modify-write does not add a dependent round trip per block row
        double   cin[NT][4];
        unsigned cbad = 0; // RC 2: bit 4 u + r set when C element (u, r) is not finite
        if constexpr(RC != 0)
        {
            if constexpr(CW)
            {
#pragma unroll
                for(int w2 = 0; w2 < NT / 2; w2++)
#pragma unroll
                    for(int r = 0; r < 4; r++)
                    {
                        const int row = br * 16 + (int)drow[r], cj = j0 + 32 * w2 + 2 * (int)dcol[r];
                        v2d       c   = (v2d){0.0, 0.0};
                        if(row < m && cj < n)
                            c = *reinterpret_cast<const v2d *>(C + (size_t)row * ldc + cj);
                        cin[2 * w2][r] = c.x, cin[2 * w2 + 1][r] = c.y;
                    }
            }
            else
            {
#pragma unroll
                for(int u = 0; u < NT; u++)
#pragma unroll
                    for(int r = 0; r < 4; r++)
                    {
                        const int row = br * 16 + (int)drow[r], cj = colof(u, (int)dcol[r]);
                        cin[u][r]     = (row < m && cj < n) ? C[(size_t)row * ldc + cj] : 0.0;
                    }
            }
        }
        auto classify = [&]() {
            if constexpr(RC == 2)
            {
#pragma unroll
                for(int u = 0; u < NT; u++)
#pragma unroll
                    for(int r = 0; r < 4; r++)
                        cbad |= bell_finite(cin[u][r]) ? 0u : (1u << (4 * u + r));
            }
        };
        if(nblk == 0)
            classify();
        for(int s = 0; s < nblk; s += 2)
        {
            fetch_a(s + 1, a1);
            fetch_b(s + 1, b1);
            mac(a0, b0);
            if(s == 0)
                classify(); // (behind the first block's MFMAs: the C tile has landed, its registers are free from here on)
            if(s + 1 >= nblk)
                break;
            fetch_a(s + 2, a0);
            fetch_b(s + 2, b0);
            mac(a1, b1);
        }
        // ---- an element whose tile sum is not finite: the reference's chain without the padding (cold path) ---------------------
        {
            unsigned nf = 0;
#pragma unroll
            for(int u = 0; u < NT; u++)
#pragma unroll
                for(int r = 0; r < 4; r++)
                    nf |= bell_finite(acc[u][r]) ? 0u : (1u << (4 * u + r));
            if(nf)
            {
#pragma unroll
                for(int u = 0; u < NT; u++)
#pragma unroll
                    for(int r = 0; r < 4; r++)
                        if((nf >> (4 * u + r)) & 1u)
                        {
                            const int row = br * 16 + (int)drow[r], cj = colof(u, (int)dcol[r]);
                            if(row < m && cj < n)
                                acc[u][r] = bell_exact_element<false>(row, cj, base, rp, ci, cv, B, ldb);
                        }
            }
        }
        // ---- C = beta * C + alpha * acc, the reference's closing fma (csrmm.hpp:83) ----------------------------------------
        if constexpr(CW)
        {
#pragma unroll
            for(int w2 = 0; w2 < NT / 2; w2++)
#pragma unroll
                for(int r = 0; r < 4; r++)
                {
                    const int row = br * 16 + (int)drow[r], cj = j0 + 32 * w2 + 2 * (int)dcol[r];
                    if(row < m && cj < n) // (n is even: the pair is in or out together)
                    {
                        v2d         *cp = reinterpret_cast<v2d *>(C + (size_t)row * ldc + cj);
                        const double z0 = alpha * acc[2 * w2][r], z1 = alpha * acc[2 * w2 + 1][r];
                        v2d          o;
                        if constexpr(RC == 1)
                        {
                            o.x = fma(beta, cin[2 * w2][r], z0), o.y = fma(beta, cin[2 * w2 + 1][r], z1);
                            *cp = o;
                        }
                        else
                        {
                            // RC 2: a finite C contributes nothing unless z == 0 (then the zero's sign depends on it); RC 0 (C
                            // overwritten): the same rule for z == 0.  fma(0, c, z) == z for a finite c and z != 0, so taking the
                            // reference's fma on both elements of a pair is right when either needs it.
                            const bool slow = z0 == 0.0 || z1 == 0.0 || (RC == 2 && ((cbad >> (4 * (2 * w2) + r)) & 0x11u) != 0);
                            if(slow)
                            {
                                const v2d c = *cp;
                                o.x = fma(beta, c.x, z0), o.y = fma(beta, c.y, z1);
                                *cp = o;
                            }
                            else
                            {
                                o.x = z0, o.y = z1;
                                if constexpr(RC == 0)
                                    __builtin_nontemporal_store(o, cp);
                                else
                                    *cp = o;
                            }
                        }
                    }
                }
        }
        else
        {
#pragma unroll
            for(int u = 0; u < NT; u++)
#pragma unroll
                for(int r = 0; r < 4; r++)
                {
                    const int row = br * 16 + (int)drow[r], cj = colof(u, (int)dcol[r]);
                    if(row < m && cj < n)
                    {
                        double      *cp = C + (size_t)row * ldc + cj;
                        const double z  = alpha * acc[u][r];
                        if constexpr(RC == 1)
                            *cp = fma(beta, cin[u][r], z);
                        else if constexpr(RC == 2)
                        {
                            if(((cbad >> (4 * u + r)) & 1u) || z == 0.0)
                                *cp = fma(beta, *cp, z);
                            else
                                *cp = z; // == fma(0, c, z) for a finite c and z != 0
                        }
                        else if(z == 0.0)
                            *cp = fma(beta, *cp, z);
                        else
                            __builtin_nontemporal_store(z, cp);
                    }
                }
        }
    }
    // 4 x 4 transpose between the four 16-lane rows of a wavefront and four registers: lane row a, register b holds M[a][b] on
    // entry and M[b][a] on return.  Two butterfly stages of the gfx950 swap instructions: v_permlane32_swap (lanes 32..63 of the
    // first operand <-> lanes 0..31 of the second) on registers (b, b+2), then v_permlane16_swap (odd rows of the first <-> even
    // rows of the second) on registers (0, 1) and (2, 3).
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    template <bool HALF32>
    __device__ __forceinline__ void swap_rows(double &x, double &y)
    {
        const unsigned long long xb = __builtin_bit_cast(unsigned long long, x), yb = __builtin_bit_cast(unsigned long long, y);
        v2u lo, hi;
        if constexpr(HALF32)
        {
            lo = __builtin_amdgcn_permlane32_swap((unsigned)xb, (unsigned)yb, false, false);
            hi = __builtin_amdgcn_permlane32_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
        }
        else
        {
            lo = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
            hi = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
        }
        x = __builtin_bit_cast(double, ((unsigned long long)hi.x << 32) | lo.x);
        y = __builtin_bit_cast(double, ((unsigned long long)hi.y << 32) | lo.y);
    }
    __device__ __forceinline__ void transpose_rows_regs(double (&m)[4])
    {
        swap_rows<true>(m[0], m[2]);
        swap_rows<true>(m[1], m[3]);
        swap_rows<false>(m[0], m[1]);
        swap_rows<false>(m[2], m[3]);
    }

    // ---- column-major operands (the layout the multi-GPU column shards are contiguous in) ------------------------------------
    // The same blocks, the same fragments, the roles of the two MFMA operands swapped: D' = (B^T tile) x (A^T block), a 16 x 16 tile
    // of C^T whose FIRST index is the column j and whose second is the row i -- so the lanes of a result register hold 16
    // consecutive rows of one column, contiguous in a column-major C (128-byte segments).  Element (k, j) of the B operand is
    // B[j * ldb + k]: fragment t of lane (j, kq) is the double at k = 16 bc + 4 t + kq of column j.  Loading it that way reads 16
    // columns x 32 bytes per instruction (2.70 ms at 256 columns, against 1.39 row-major); WIDE: lane (j, a) loads the 32
    // CONTIGUOUS bytes k = 16 bc + 4 a .. + 3 of its column (two 16-byte loads: whole 128-byte lines per column and instruction)
    // and the four values are transposed between lane rows and registers (transpose_rows_regs).  The chain per element is unchanged
    // (k = 4t .. 4t+3 in order, blocks ascending): the same bits as the row-major kernel and as the reference.
    template <int NT, int RC, bool FULL, bool WIDE>
    __global__ __launch_bounds__(256) void csrmm_bell_mfma_col_kernel(double alpha, aoclsparse_int m, aoclsparse_int k, aoclsparse_int nbr,
                                                                      aoclsparse_int width, const double *__restrict__ val,
                                                                      const aoclsparse_int *__restrict__ bcol,
                                                                      const double *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                                      double beta, double *__restrict__ C, aoclsparse_int ldc,
                                                                      int waves_per_row, const aoclsparse_int *__restrict__ order, int order_len, int base, const aoclsparse_int *__restrict__ rp,
                                                                      const aoclsparse_int *__restrict__ ci, const double *__restrict__ cv)
    {
        const int  wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int  wpr = waves_per_row & ~MM_DESCENDING;
        int br, cw;
        if(!bell_place(mm_linear_index(waves_per_row), wv, wpr, nbr, order, order_len, br, cw))
            return;
        const int lane = threadIdx.x & 63, jl = lane & 15, kq = lane >> 4;
        const int j0   = cw * NT * 16;
        v4d       acc[NT];
#pragma unroll
        for(int u = 0; u < NT; u++)
            acc[u] = (v4d){0.0, 0.0, 0.0, 0.0};
        const v4d zero4 = (v4d){0.0, 0.0, 0.0, 0.0};
        // result register r of this lane: index of the FIRST operand's row (here: the column j) and of the second's column (the row i)
        const v4d dfirst  = __builtin_amdgcn_mfma_f64_16x16x4f64(kq == 0 ? (double)jl : 0.0, kq == 0 ? 1.0 : 0.0, zero4, 0, 0, 0);
        const v4d dsecond = __builtin_amdgcn_mfma_f64_16x16x4f64(kq == 0 ? 1.0 : 0.0, kq == 0 ? (double)jl : 0.0, zero4, 0, 0, 0);
        bool      colok[NT];
#pragma unroll
        for(int u = 0; u < NT; u++)
            colok[u] = j0 + 16 * u + jl < n;
        const double         *vb = val + (size_t)br * width * 256;
        const aoclsparse_int *cb = bcol + (size_t)br * width;
        int                   nblk = 0;
        while(nblk < width && cb[nblk] >= 0)
            nblk++;
        // (a slot past the last block is clamped to it, not skipped: see the row-major kernel)
        auto fetch_a = [&](int s, double (&a)[4]) {
            const double *vs  = vb + (size_t)(s < nblk ? s : nblk - 1) * 256;
            const v2d     a01 = *reinterpret_cast<const v2d *>(vs + 2 * lane), a23 = *reinterpret_cast<const v2d *>(vs + 128 + 2 * lane);
            a[0] = a01.x, a[1] = a01.y, a[2] = a23.x, a[3] = a23.y;
        };
        // this lane's column of every tile: a 64-bit offset per tile (columns are ldb apart), + a wave-uniform row offset per block
        size_t colbase[NT];
#pragma unroll
        for(int u = 0; u < NT; u++)
            colbase[u] = (size_t)(colok[u] ? j0 + 16 * u + jl : 0) * (size_t)ldb + (size_t)(WIDE ? 4 * kq : kq);
        auto fetch_b = [&](int s, double (&b)[NT][4]) {
            const int     bc = __builtin_amdgcn_readfirstlane(cb[s < nblk ? s : nblk - 1]);
            const double *bs = B + (size_t)bc * 16;
            if constexpr(WIDE)
            {
                // (k is a multiple of 4 here -- the launcher's condition -- so a lane's four rows are in or out together)
                const bool rok = FULL || bc * 16 + 4 * kq < k;
#pragma unroll
                for(int u = 0; u < NT; u++)
                {
                    v2d p = (v2d){0.0, 0.0}, q = (v2d){0.0, 0.0};
                    if(FULL || (rok && colok[u]))
                    {
                        p = *reinterpret_cast<const v2d *>(bs + colbase[u]);
                        q = *reinterpret_cast<const v2d *>(bs + colbase[u] + 2);
                    }
                    b[u][0] = p.x, b[u][1] = p.y, b[u][2] = q.x, b[u][3] = q.y;
                }
            }
            else
            {
#pragma unroll
                for(int t = 0; t < 4; t++)
                {
                    const bool rok = FULL || bc * 16 + 4 * t + kq < k;
#pragma unroll
                    for(int u = 0; u < NT; u++)
                        b[u][t] = (FULL || (rok && colok[u])) ? bs[colbase[u] + 4 * t] : 0.0;
                }
            }
        };
        auto mac = [&](const double (&a)[4], double (&b)[NT][4]) {
            if constexpr(WIDE)
            {
#pragma unroll
                for(int u = 0; u < NT; u++)
                    transpose_rows_regs(b[u]); // (lane row a, register r) k = 4a + r  ->  k = 4r + a: the MFMA fragment
            }
#pragma unroll
            for(int t = 0; t < 4; t++)
#pragma unroll
                for(int u = 0; u < NT; u++)
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[u][t], a[t], acc[u], 0, 0, 0);
        };
        double a0[4], a1[4], b0[NT][4], b1[NT][4];
        if(nblk > 0)
        {
            fetch_a(0, a0);
            fetch_b(0, b0);
        }
        // beta == 0 with C read (RC 2): the C tile requested now, reduced to a finite / not-finite bit per element behind the first
        // block's MFMAs (see the row-major kernel)
        double   cin[NT][4];
        unsigned cbad = 0;
        if constexpr(RC == 2)
        {
#pragma unroll
            for(int u = 0; u < NT; u++)
#pragma unroll
                for(int r = 0; r < 4; r++)
                {
                    const int cj = j0 + 16 * u + (int)dfirst[r], row = br * 16 + (int)dsecond[r];
                    cin[u][r]    = (row < m && cj < n) ? C[(size_t)cj * ldc + row] : 0.0;
                }
        }
        auto classify = [&]() {
            if constexpr(RC == 2)
            {
#pragma unroll
                for(int u = 0; u < NT; u++)
#pragma unroll
                    for(int r = 0; r < 4; r++)
                        cbad |= bell_finite(cin[u][r]) ? 0u : (1u << (4 * u + r));
            }
        };
        if(nblk == 0)
            classify();
        for(int s = 0; s < nblk; s += 2)
        {
            fetch_a(s + 1, a1);
            fetch_b(s + 1, b1);
            mac(a0, b0);
            if(s == 0)
                classify();
            if(s + 1 >= nblk)
                break;
            fetch_a(s + 2, a0);
            fetch_b(s + 2, b0);
            mac(a1, b1);
        }
        {
            // an element whose tile sum is not finite: the reference's chain without the padding (cold path, see the row-major kernel)
            unsigned nf = 0;
#pragma unroll
            for(int u = 0; u < NT; u++)
#pragma unroll
                for(int r = 0; r < 4; r++)
                    nf |= bell_finite(acc[u][r]) ? 0u : (1u << (4 * u + r));
            if(nf)
            {
#pragma unroll
                for(int u = 0; u < NT; u++)
#pragma unroll
                    for(int r = 0; r < 4; r++)
                        if((nf >> (4 * u + r)) & 1u)
                        {
                            const int cj = j0 + 16 * u + (int)dfirst[r], row = br * 16 + (int)dsecond[r];
                            if(row < m && cj < n)
                                acc[u][r] = bell_exact_element<true>(row, cj, base, rp, ci, cv, B, ldb);
                        }
            }
        }
#pragma unroll
        for(int u = 0; u < NT; u++)
#pragma unroll
            for(int r = 0; r < 4; r++)
            {
                const int cj = j0 + 16 * u + (int)dfirst[r], row = br * 16 + (int)dsecond[r];
                if(row < m && cj < n)
                {
                    double      *cp = C + (size_t)cj * ldc + row;
                    const double z  = alpha * acc[u][r];
                    if constexpr(RC == 1)
                        *cp = fma(beta, *cp, z);
                    else if constexpr(RC == 2)
                    {
                        if(((cbad >> (4 * u + r)) & 1u) || z == 0.0)
                            *cp = fma(beta, *cp, z);
                        else
                            *cp = z;
                    }
                    else if(z == 0.0)
                        *cp = fma(beta, *cp, z);
                    else
                        __builtin_nontemporal_store(z, cp);
                }
            }
    }
} // namespace

aoclsparse_status launch_csrmm_bell(hipStream_t s, double alpha, aoclsparse_int m, aoclsparse_int k, const BellPlan &bell,
                                    int base, const aoclsparse_int *rp, const aoclsparse_int *ci, const double *cv,
                                    const double *B, aoclsparse_int n, aoclsparse_int ldb, double beta, double *C,
                                    aoclsparse_int ldc, bool column_major)
{
    if(n <= 0 || m <= 0 || !bell.valid)
        return aoclsparse_status_success;
    // 0: C overwritten (opt-in), 1: beta != 0, 2: beta == 0 with C read (the default mode)
    const int rcmode = beta != 0.0 ? 1 : (csrmm_reads_c(false) ? 2 : 0);
    const int  tiles = (n + 15) / 16;
    if(column_major)
    {
        // tiles per wavefront: 2 (32 columns).  A column of B is its own 2 MB page at these sizes, so a wavefront touches one page
        // per column and block; 1 / 2 / 3 / 4 tiles per wavefront at 256 columns: 2.06 / 1.90 / 1.89 / 2.31 ms with C read, 1.69 /
        // 1.58 / 1.53 / 1.69 overwriting (profiles/r4/bell_experiments.txt); the 32-column slab 0.25 / 0.23 with 1 / 2
        const int  nt = std::min(tiles, 2), wpr = (tiles + nt - 1) / nt;
        // (with an order list every XCD has a stream of order_len * wpr wavefronts, 4 per workgroup, workgroups dealt to the XCDs in turn)
        const aoclsparse_int *ord = bell.order_len > 0 ? bell.order.as<aoclsparse_int>() : nullptr;
        const int  ordlen = ord ? bell.order_len : 0;
        const long waves  = (long)bell.nbr * wpr;
        const dim3 grid((unsigned)(ord ? 8 * (((long)ordlen * wpr + 3) / 4) : (waves + 3) / 4)), block(256);
        const bool full = n % (16 * nt) == 0 && k % 16 == 0;
        // 32 contiguous bytes per lane: 16-byte aligned columns (B aligned, ldb even) and a row count of B that is a multiple of 4
        const bool wide = reinterpret_cast<uintptr_t>(B) % 16 == 0 && ldb % 2 == 0 && k % 4 == 0;
#define MI355_BELLC(NT, FULL) MI355_BELLC2(NT, FULL, false)
#define MI355_BELLC2(NT, FULL, WIDE)                                                                                                 \
    do                                                                                                                          \
    {                                                                                                                           \
        if(rcmode == 1)                                                                                                         \
            hipLaunchKernelGGL((csrmm_bell_mfma_col_kernel<NT, 1, FULL, WIDE>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width, \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, ord, ordlen, base, rp, ci, cv); \
        else if(rcmode == 2)                                                                                                    \
            hipLaunchKernelGGL((csrmm_bell_mfma_col_kernel<NT, 2, FULL, WIDE>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width, \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, ord, ordlen, base, rp, ci, cv); \
        else                                                                                                                    \
            hipLaunchKernelGGL((csrmm_bell_mfma_col_kernel<NT, 0, FULL, WIDE>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width, \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, ord, ordlen, base, rp, ci, cv); \
    } while(0)
        if(wide)
        {
            if(nt == 2 && full)
                MI355_BELLC2(2, true, true);
            else if(nt == 2)
                MI355_BELLC2(2, false, true);
            else
                MI355_BELLC2(1, false, true);
        }
        else if(nt == 2)
            MI355_BELLC(2, false);
        else
            MI355_BELLC(1, false);
#undef MI355_BELLC
#undef MI355_BELLC2
        MI355_HIP_TRY(hipGetLastError());
        return aoclsparse_status_success;
    }
    // 16-byte operand loads feed two tiles at once: 16-byte aligned B rows, an even column count
    const bool wide = tiles >= 2 && n % 2 == 0 && ldb % 2 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0;
    // tiles per wavefront: 4 (64 columns) when there are that many, else what the slab has (an even count in wide mode)
    const int nt  = tiles >= 4 ? 4 : (wide ? 2 : tiles);
    const int wpr = (tiles + nt - 1) / nt;
    const aoclsparse_int *ord = bell.order_len > 0 ? bell.order.as<aoclsparse_int>() : nullptr;
    const int  ordlen = ord ? bell.order_len : 0;
    const long waves  = (long)bell.nbr * wpr;
    const dim3 grid((unsigned)(ord ? 8 * (((long)ordlen * wpr + 3) / 4) : (waves + 3) / 4)), block(256);
    // 16-byte accesses to C for the tile pairs of wide mode: 16-byte aligned rows of C
    const bool cw = wide && ldc % 2 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0;
#define MI355_BELL(NT, WIDE) MI355_BELL2(NT, WIDE, false)
#define MI355_BELL2(NT, WIDE, FULL) do { if(cw && WIDE) MI355_BELL3(NT, WIDE, FULL, WIDE); else MI355_BELL3(NT, WIDE, FULL, false); } while(0)
#define MI355_BELL3(NT, WIDE, FULL, CWF)                                                                                                    \
    do                                                                                                                          \
    {                                                                                                                           \
        if(rcmode == 1)                                                                                                         \
            hipLaunchKernelGGL((csrmm_bell_mfma_kernel<NT, 1, WIDE, FULL, CWF>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width,       \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, ord, ordlen, base, rp, ci, cv); \
        else if(rcmode == 2)                                                                                                    \
            hipLaunchKernelGGL((csrmm_bell_mfma_kernel<NT, 2, WIDE, FULL, CWF>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width,       \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, ord, ordlen, base, rp, ci, cv); \
        else                                                                                                                    \
            hipLaunchKernelGGL((csrmm_bell_mfma_kernel<NT, 0, WIDE, FULL, CWF>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width,       \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, ord, ordlen, base, rp, ci, cv); \
    } while(0)
    const bool full = n % (16 * nt) == 0 && k % 16 == 0;
    if(wide && full)
    {
        if(nt == 4)
            MI355_BELL2(4, true, true);
        else
            MI355_BELL2(2, true, true);
    }
    else if(wide)
    {
        if(nt == 4)
            MI355_BELL(4, true);
        else
            MI355_BELL(2, true);
    }
    else
        switch(nt)
        {
        case 1: MI355_BELL(1, false); break;
        case 2: MI355_BELL(2, false); break;
        case 3: MI355_BELL(3, false); break;
        default: MI355_BELL(4, false); break;
        }
#undef MI355_BELL
#undef MI355_BELL2
#undef MI355_BELL3
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

// Values of the blocked-ELL copy, scattered from the CSR arrays in HBM (round 4: the copy used to be assembled on the host -- 0.9 GB
// zeroed, filled and sent for the 1 M-row block-dense stand-in, 380 ms of aoclsparse_optimize).  A thread per row; the row's entries
// ascend, so do the slots of its block row they fall into (bcol: block columns ascending, -1 = empty slot).  `out` is zeroed by
// the caller.  Cell (i, k) of a block: fragment t = k / 4, lane = 16 * (k % 4) + i, stored at 128 * (t / 2) + 2 * lane + t % 2.
__global__ __launch_bounds__(256) void bell_fill_kernel(aoclsparse_int m, int base, const aoclsparse_int *__restrict__ ptr,
                                                        const aoclsparse_int *__restrict__ ind, const double *__restrict__ val,
                                                        aoclsparse_int width, const aoclsparse_int *__restrict__ bcol,
                                                        double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if(i >= m)
        return;
    const long long       b    = i / BELL_BS;
    const int             r    = (int)(i % BELL_BS);
    const aoclsparse_int *slot = bcol + b * width;
    double               *vb   = out + b * width * 256;
    int                   s    = 0;
    for(aoclsparse_int p = ptr[i] - base; p < ptr[i + 1] - base; p++)
    {
        const int c = ind[p] - base, cb = c / BELL_BS, kk = c % BELL_BS;
        while(s < width - 1 && slot[s] != cb)
            s++;
        const int t = kk / 4, ln = 16 * (kk % 4) + r;
        vb[(long long)s * 256 + 128 * (t / 2) + 2 * ln + (t & 1)] = val[p];
    }
}

aoclsparse_status launch_bell_fill(hipStream_t s, aoclsparse_int m, int base, const aoclsparse_int *ptr, const aoclsparse_int *ind,
                                   const double *val, aoclsparse_int nbr, aoclsparse_int width, const aoclsparse_int *bcol, double *out)
{
    MI355_HIP_TRY(hipMemsetAsync(out, 0, sizeof(double) * (size_t)nbr * (size_t)width * 256, s));
    if(m > 0)
        hipLaunchKernelGGL(bell_fill_kernel, dim3((unsigned)(((long long)m + 255) / 256)), dim3(256), 0, s, m, base, ptr, ind, val, width,
                           bcol, out);
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

} // namespace mi355
