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
// tools/mfma_f64_probe.hip, profiles/r2/mfma_f64_probe.jsonl), blocks are walked in ascending column order, so every C element
// is the reference's chain over its row in CSR order (csrmm.hpp:69-85) with the tile's explicit zeros interleaved:
// fma(0, b, sum) == sum for every finite b (a sum that starts at +0 is never -0).  An Inf / NaN in B at a position the row does
// not store would turn 0 * Inf into NaN -- the same caveat as the reference's own padded formats (BLKCSR / br4: SURVEY.md a6);
// build_bell requires sorted rows, and the plan is used for finite alpha / beta classes exactly like the other kernels.
#include "internal.hpp"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace mi355
{

namespace
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    typedef double v2d __attribute__((ext_vector_type(2)));

    // Operand fragments: lane -> (i or j = lane % 16, k = lane / 16), one double per lane.  Where the 4 result registers of a
    // lane sit in the 16 x 16 tile is NOT assumed: every wavefront asks the instruction itself with two extra MFMAs
    // (D = [i] and D = [j]: A = column of row numbers x B = row of ones, and the transpose) -- 2 of ~114 per block row.
    // FULL: every tile column of every wavefront is < n and the column count of A is a multiple of 16: no masks on the B loads
    template <int NT, bool RC, bool WIDE, bool FULL>
    __global__ __launch_bounds__(256) void csrmm_bell_mfma_kernel(double alpha, aoclsparse_int m, aoclsparse_int k, aoclsparse_int nbr,
                                                                  aoclsparse_int width, const double *__restrict__ val,
                                                                  const aoclsparse_int *__restrict__ bcol,
                                                                  const double *__restrict__ B, aoclsparse_int n, aoclsparse_int ldb,
                                                                  double beta, double *__restrict__ C, aoclsparse_int ldc,
                                                                  int waves_per_row, bool readc)
    {
        // (block rows in launch order: giving every XCD a contiguous eighth of them -- the rule of the HBM-bound kernels here --
        // changed nothing, 0.93 -> 0.98 ms: this kernel is bound by the MFMA pipe, profiles/r4/bell_experiments.txt)
        const int wv   = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const long g   = (long)blockIdx.x * 4 + wv;
        const int  br  = (int)(g / waves_per_row), cw = (int)(g % waves_per_row);
        if(br >= nbr)
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
        auto fetch_a = [&](int s, double (&a)[4]) {
            if(s >= nblk)
            {
                a[0] = a[1] = a[2] = a[3] = 0.0;
                return;
            }
            const double *vs = vb + (size_t)s * 256;
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
            if(s >= nblk)
                return; // (never multiplied: the loop ends first)
            const int     bc = __builtin_amdgcn_readfirstlane(cb[s]);
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
        fetch_a(0, a0);
        fetch_b(0, b0);
        for(int s = 0; s < nblk; s += 2)
        {
            fetch_a(s + 1, a1);
            fetch_b(s + 1, b1);
            mac(a0, b0);
            if(s + 1 >= nblk)
                break;
            fetch_a(s + 2, a0);
            fetch_b(s + 2, b0);
            mac(a1, b1);
        }
        // ---- C = beta * C + alpha * acc, the reference's closing fma (csrmm.hpp:83) ----------------------------------------
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
                    if(RC || readc || z == 0.0)
                        *cp = fma(beta, *cp, z);
                    else
                        __builtin_nontemporal_store(z, cp);
                }
            }
    }
} // namespace

aoclsparse_status launch_csrmm_bell(hipStream_t s, double alpha, aoclsparse_int m, aoclsparse_int k, const BellPlan &bell,
                                    const double *B, aoclsparse_int n, aoclsparse_int ldb, double beta, double *C,
                                    aoclsparse_int ldc)
{
    if(n <= 0 || m <= 0 || !bell.valid)
        return aoclsparse_status_success;
    const bool readc = csrmm_reads_c(beta != 0.0);
    const int  tiles = (n + 15) / 16;
    // 16-byte operand loads feed two tiles at once: 16-byte aligned B rows, an even column count
    const bool wide = tiles >= 2 && n % 2 == 0 && ldb % 2 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0;
    // tiles per wavefront: 4 (64 columns) when there are that many, else what the slab has (an even count in wide mode)
    const int nt  = tiles >= 4 ? 4 : (wide ? 2 : tiles);
    const int wpr = (tiles + nt - 1) / nt;
    const long waves  = (long)bell.nbr * wpr;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define MI355_BELL(NT, WIDE) MI355_BELL2(NT, WIDE, false)
#define MI355_BELL2(NT, WIDE, FULL)                                                                                                      \
    do                                                                                                                          \
    {                                                                                                                           \
        if(readc)                                                                                                               \
            hipLaunchKernelGGL((csrmm_bell_mfma_kernel<NT, true, WIDE, FULL>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width,    \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, readc); \
        else                                                                                                                    \
            hipLaunchKernelGGL((csrmm_bell_mfma_kernel<NT, false, WIDE, FULL>), grid, block, 0, s, alpha, m, k, bell.nbr, bell.width,   \
                               bell.val.as<double>(), bell.bcol.as<aoclsparse_int>(), B, n, ldb, beta, C, ldc, wpr, readc); \
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
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

} // namespace mi355
