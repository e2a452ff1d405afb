// csrmm_window_kernels.hip -- column-major C = alpha*A*B + beta*C for matrices whose row blocks touch ONE short stretch of
// columns (banded matrices, 2-D / 3-D stencils in natural order), gfx950.  Round 4.
//
// Why: column-major operands are the layout the multi-GPU column shards are contiguous in, and the
// lane-per-row kernels (csrmm_col_kernel, csrmm_colpair_kernel) gather B with 8 / 16-byte lane loads at unaligned offsets, five
// times per output for a 5-point row: 1.18 ms (C overwritten) / 1.49 ms (C read) at 256 columns of the 1000^2 Laplacian,
// 3.5-4.2 TB/s of algorithmic bytes, issue-bound in the vector-memory path for two rounds.
//
// Window form: a workgroup of 512 lanes owns R = 512 * RPT consecutive rows and a chunk of columns.
//   * the rows' entries stay in REGISTERS for the whole chunk (value + the byte offset of its B operand inside the window,
//     16 bits), so A is read once per chunk;
//   * per column, the B values the block can touch -- B[wmin, wmin + wlen) of that column, one contiguous, 16-byte aligned
//     stretch -- are copied into LDS by LDS-DMA (global_load_lds_dwordx4: whole aligned 1 KiB wave-instructions, no VGPRs),
//     double buffered: column j+1's stretch and column j+1's C values are in flight while column j is computed;
//   * every output is the reference's chain (csrmm.hpp:69-85: sum = fma(a_ik, B_kj, sum) in CSR order, then
//     C = fma(beta, C, alpha * sum)) over ds_read_b64 operands, lanes on consecutive rows (conflict-free for a stencil).
// B is fetched wlen / R times per column in whole lines (1.49x for the 1000^2 Laplacian at R = 4096, the halo out of L2: row
// blocks that share it run side by side on one XCD) instead of 5 unaligned gathers per output.
// Measured (tools/history/csrmm_cm_r4.hip, profiles/r4/cm_window.txt), 1000^2 Laplacian, same box as the kernels it replaces:
//   256 columns: 1.175 -> 0.76 ms (overwrite), 1.490 -> 1.18 ms (C read); 32-column slab: 0.179 -> 0.101, 0.229 -> 0.149 ms.
// Shapes tried there (R = 256 .. 1024 lanes x 2 .. 8 rows, 16-128 columns per workgroup, register staging instead of LDS-DMA,
// stores deferred by a step, three buffers with counted vmcnt): 512 lanes x 8 rows, two buffers is the best; register staging
// loses 20-50 % (VGPRs); a third buffer gains 2.6 % without C read and nothing with it (hipcc drains vmcnt at the first use of
// an ordinary load while an LDS-DMA is pending), not worth its scratch-slot bookkeeping.
//
// The window plan (csrmm_api.cpp: detect_windows) is built once per handle; the kernel applies when B is 16-byte aligned
// with an even leading dimension (every column's stretch then starts on a 16-byte boundary).
// Rows shorter than K are padded with {0.0, offset of a zero cell of LDS}: fma(0, 0, sum) leaves every sum unchanged (a sum
// that starts at +0 can never be -0), so no NaN / Inf of B can enter through padding.  Rows longer than K finish their chain
// from the CSR arrays (same order).
#include "internal.hpp"
#include "mm_order.hpp"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace mi355
{

namespace
{
    typedef __attribute__((address_space(3))) void lds_void;

    __device__ __forceinline__ double cw_fma(double a, double b, double c)
    {
        return fma(a, b, c);
    }
    __device__ __forceinline__ float cw_fma(float a, float b, float c)
    {
        return fmaf(a, b, c);
    }

    constexpr int CW_NT   = 512; // lanes per workgroup
    constexpr int CW_MAXP = 6; // 16-byte pieces per lane and column: a window holds at most 512 * 6 - 1 pieces (48 KB - 16)
    constexpr int CW_ZERO = (CW_NT * CW_MAXP - 1) * 16; // byte offset of the zero cell (the last piece of each buffer)

    // win[2b], win[2b+1] = first column of block b's window (0-based, a multiple of 16 / sizeof(T)) and its 16-byte pieces.
    // The last piece may reach past the matrix's last column but never past the column's ldb elements (ldb is a multiple of
    // the elements per piece: csrmm_window_applies), i.e. never outside the caller's array.
    template <typename T, int RPT, int K, bool RC>
    __global__ __launch_bounds__(CW_NT) void csrmm_colwin_kernel(int base, T alpha, aoclsparse_int m, const T *__restrict__ val,
                                                               const aoclsparse_int *__restrict__ col,
                                                               const aoclsparse_int *__restrict__ row_ptr,
                                                               const aoclsparse_int *__restrict__ win, const T *__restrict__ B,
                                                               aoclsparse_int n, aoclsparse_int ldb, T beta, T *__restrict__ C,
                                                               aoclsparse_int ldc, int cc, int chunk, bool readc)
    {
        extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
        constexpr int KP  = (K + 1) / 2;         // packed offset registers per row
        const int     tid = threadIdx.x;
        const int     bx  = mm_block_index(chunk);
        const int     r0  = bx * CW_NT * RPT;
        if(r0 >= m)
            return;
        const int wmin = win[2 * bx], pieces = win[2 * bx + 1];
        const int j0 = blockIdx.y * cc, j1 = min((int)n, j0 + cc);
        unsigned char *buf0 = lds_raw, *buf1 = lds_raw + CW_NT * CW_MAXP * 16;
        if(tid == 0)
        {
            // the zero cells (never written by a copy: pieces <= CW_NT * CW_MAXP - 1)
            *reinterpret_cast<double *>(buf0 + CW_ZERO) = 0.0, *reinterpret_cast<double *>(buf0 + CW_ZERO + 8) = 0.0;
            *reinterpret_cast<double *>(buf1 + CW_ZERO) = 0.0, *reinterpret_cast<double *>(buf1 + CW_ZERO + 8) = 0.0;
        }
        // ---- this lane's rows: values and packed byte offsets, once per chunk --------------------------------------------
        T        v[RPT][K];
        unsigned offp[RPT][KP];
        bool     longrow = false;
#pragma unroll
        for(int q = 0; q < RPT; q++)
        {
            const int i = r0 + q * CW_NT + tid;
            int       s = 0, e = 0;
            if(i < m)
                s = row_ptr[i] - base, e = row_ptr[i + 1] - base;
#pragma unroll
            for(int k = 0; k < KP; k++)
                offp[q][k] = (unsigned)CW_ZERO | ((unsigned)CW_ZERO << 16);
#pragma unroll
            for(int k = 0; k < K; k++)
            {
                v[q][k] = T(0);
                if(s + k < e)
                {
                    v[q][k]            = val[s + k];
                    const unsigned off = (unsigned)(col[s + k] - base - wmin) * (unsigned)sizeof(T);
                    offp[q][k / 2]     = (k & 1) ? ((offp[q][k / 2] & 0xffffu) | (off << 16)) : ((offp[q][k / 2] & 0xffff0000u) | off);
                }
            }
            longrow = longrow || e - s > K;
        }
        // ---- staging of one column's stretch ---------------------------------------------------------------------------------
        const int wave = tid >> 6, lane = tid & 63;
        auto      stage = [&](int j, unsigned char *dst) {
            const T *src = B + (size_t)j * ldb + wmin;
#pragma unroll
            for(int it = 0; it < CW_MAXP; it++)
            {
                const int p0 = (it * (CW_NT / 64) + wave) * 64; // first piece of this wave-instruction (wave-uniform)
                if(p0 + lane < pieces)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned char *>(src) + (size_t)(p0 + lane) * 16,
                                                     (lds_void *)(dst + (size_t)p0 * 16), 16, 0, 0);
            }
        };
        T    cinA[RPT], cinB[RPT];
        auto load_c = [&](int j, T (&cin)[RPT]) {
            if constexpr(RC)
            {
#pragma unroll
                for(int q = 0; q < RPT; q++)
                {
                    const int i = r0 + q * CW_NT + tid;
                    cin[q]      = i < m ? C[(size_t)i + (size_t)j * ldc] : T(0);
                }
            }
        };
        // One column: wait for this lane's copies (hipcc does not count an LDS-DMA as a pending LDS write: the explicit vmcnt),
        // barrier (column j's stretch is complete, column j-1 is computed), put column j+1's C values and stretch in flight,
        // compute column j.  Column j+1's C values are first used after the next barrier: nothing waits inside a step.
        auto step = [&](int j, const unsigned char *cur, unsigned char *nxt, T (&cin)[RPT], T (&cnx)[RPT]) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if(j + 1 < j1)
            {
                load_c(j + 1, cnx);
                stage(j + 1, nxt);
            }
#pragma unroll
            for(int q = 0; q < RPT; q++)
            {
                const int i = r0 + q * CW_NT + tid;
                T         a = T(0);
#pragma unroll
                for(int k = 0; k < K; k++)
                {
                    const unsigned off = (k & 1) ? (offp[q][k / 2] >> 16) : (offp[q][k / 2] & 0xffffu);
                    a                  = cw_fma(v[q][k], *reinterpret_cast<const T *>(cur + off), a);
                }
                if(longrow && i < m) // (rare: rows longer than the register cache finish their chain from the CSR arrays)
                    for(int p = row_ptr[i] - base + K, e = row_ptr[i + 1] - base; p < e; p++)
                        a = cw_fma(val[p], *reinterpret_cast<const T *>(cur + (size_t)(col[p] - base - wmin) * sizeof(T)), a);
                if(i < m)
                {
                    T      *cp = C + (size_t)i + (size_t)j * ldc;
                    const T z  = alpha * a;
                    if constexpr(RC)
                        *cp = cw_fma(beta, cin[q], z);
                    else if(readc || z == T(0)) // the sign of an exact zero is beta * C's (the reference computes 0 * C + z)
                        *cp = cw_fma(beta, *cp, z);
                    else
                        __builtin_nontemporal_store(z, cp);
                }
            }
        };
        load_c(j0, cinA);
        stage(j0, buf0);
        for(int j = j0; j < j1; j += 2)
        {
            step(j, buf0, buf1, cinA, cinB);
            if(j + 1 < j1)
                step(j + 1, buf1, buf0, cinB, cinA);
        }
    }
} // namespace

int csrmm_window_rows(aoclsparse_int max_row_nnz, size_t elem)
{
    // rows per workgroup: 8 per lane while the register cache holds the rows (<= 5 entries each), else 4 (<= 9 entries each in
    // registers, the rest of a longer row from the CSR arrays): 512 lanes need <= 256 VGPRs each
    (void)elem;
    return max_row_nnz <= 5 ? 8 * CW_NT : 4 * CW_NT;
}

int csrmm_window_max_pieces()
{
    return CW_NT * CW_MAXP - 1;
}

template <typename T>
bool csrmm_window_applies(aoclsparse_int n, aoclsparse_int ldb, const T *B)
{
    // every column's stretch must start on a 16-byte boundary: B aligned, ldb a multiple of the elements per piece
    return n >= 4 && reinterpret_cast<uintptr_t>(B) % 16 == 0 && ldb % (aoclsparse_int)(16 / sizeof(T)) == 0;
}

template <typename T>
aoclsparse_status launch_csrmm_window(hipStream_t s, int base, T alpha, aoclsparse_int m, const T *val, const aoclsparse_int *col,
                                      const aoclsparse_int *row_ptr, const aoclsparse_int *win, int win_rows,
                                      aoclsparse_int max_row_nnz, const T *B, aoclsparse_int n, aoclsparse_int ldb, T beta, T *C,
                                      aoclsparse_int ldc)
{
    if(n <= 0 || m <= 0)
        return aoclsparse_status_success;
    const bool   readc = csrmm_reads_c(beta != T(0));
    const int    nb = (int)((m + win_rows - 1) / win_rows), chunk = (nb + 7) / 8;
    // columns per workgroup: 64 keeps the rows' entries in registers for long enough to amortise their load and still gives
    // 4 x nb workgroups at 256 columns (16 / 32 / 128 measured within 1-4 %); a narrow slab (one of 8 ranks: 32 columns) is one chunk
    const int    cc = n >= 64 ? 64 : (int)n;
    const dim3   grid(chunk * 8, (n + cc - 1) / cc), block(CW_NT);
    const size_t ldsb = (size_t)2 * CW_NT * CW_MAXP * 16; // 96 KB: one workgroup (8 waves) per CU
    // more than 64 KB of dynamic LDS must be allowed per kernel and device (cheap; the multi-device replicas launch on other devices)
#define MI355_CW(RPT, K)                                                                                                          \
    do                                                                                                                            \
    {                                                                                                                             \
        MI355_HIP_TRY(hipFuncSetAttribute(readc ? (const void *)csrmm_colwin_kernel<T, RPT, K, true>                              \
                                                : (const void *)csrmm_colwin_kernel<T, RPT, K, false>,                             \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));                                \
        if(readc)                                                                                                                 \
            hipLaunchKernelGGL((csrmm_colwin_kernel<T, RPT, K, true>), grid, block, ldsb, s, base, alpha, m, val, col, row_ptr, win, B, \
                               n, ldb, beta, C, ldc, cc, chunk, readc);                                                           \
        else                                                                                                                      \
            hipLaunchKernelGGL((csrmm_colwin_kernel<T, RPT, K, false>), grid, block, ldsb, s, base, alpha, m, val, col, row_ptr, win, B, \
                               n, ldb, beta, C, ldc, cc, chunk, readc);                                                           \
    } while(0)
    if(win_rows == 8 * CW_NT)
        MI355_CW(8, 5);
    else
        MI355_CW(4, 9);
#undef MI355_CW
    MI355_HIP_TRY(hipGetLastError());
    return aoclsparse_status_success;
}

template bool              csrmm_window_applies<double>(aoclsparse_int, aoclsparse_int, const double *);
template bool              csrmm_window_applies<float>(aoclsparse_int, aoclsparse_int, const float *);
template aoclsparse_status launch_csrmm_window<double>(hipStream_t, int, double, aoclsparse_int, const double *, const aoclsparse_int *,
                                                       const aoclsparse_int *, const aoclsparse_int *, int, aoclsparse_int,
                                                       const double *, aoclsparse_int, aoclsparse_int, double, double *, aoclsparse_int);
template aoclsparse_status launch_csrmm_window<float>(hipStream_t, int, float, aoclsparse_int, const float *, const aoclsparse_int *,
                                                      const aoclsparse_int *, const aoclsparse_int *, int, aoclsparse_int, const float *,
                                                      aoclsparse_int, aoclsparse_int, float, float *, aoclsparse_int);

} // namespace mi355
