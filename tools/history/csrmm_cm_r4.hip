// csrmm_cm_r4.hip -- diagnostic build (never shipped), round 4: column-major C = alpha*A*B + beta*C for a banded A, the
// "window" form against the shipped kernels (aoclsparse_dcsrmm of the library, same process, device pointers).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/csrmm_cm_r4.hip -Laocl-sparse_amd/lib -laoclsparse_mi355
//         -Wl,-rpath,'$ORIGIN/../../aocl-sparse_amd/lib' -o tools/bin/csrmm_cm_r4
//   csrmm_cm_r4 [g=1000] [n=256] [only=""]
// Window form W<RPT, K, DMA>: a workgroup of 256 lanes owns R = 256*RPT consecutive rows and a chunk of columns.  The rows'
// entries (value + window offset) stay in registers for the whole chunk; per column the B values the block can touch
// -- B[wmin .. wmin + wlen) of that column, ONE contiguous, 16-byte aligned stretch for a banded matrix -- are staged in LDS
// (double buffered, the next column's stretch in flight while this one is computed; DMA = global_load_lds, else through
// registers), and each output is the reference's chain over ds_reads.  Per output: B is fetched ~wlen/R times in whole aligned
// lines (2-3x for the 1000^2 Laplacian, the halo from L2) instead of 5 unaligned 8/16-byte lane gathers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "aoclsparse.h"
#include "aoclsparse_mi355.h"

#define CHECK(x)                                                                  \
    do                                                                            \
    {                                                                             \
        hipError_t e_ = (x);                                                      \
        if(e_ != hipSuccess)                                                      \
        {                                                                         \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while(0)
#define OK(x)                                                               \
    do                                                                      \
    {                                                                       \
        aoclsparse_status s_ = (x);                                         \
        if(s_ != aoclsparse_status_success)                                 \
        {                                                                   \
            printf("aoclsparse status %d at line %d\n", (int)s_, __LINE__); \
            exit(1);                                                        \
        }                                                                   \
    } while(0)

typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

// one column's stretch: wlen2 16-byte pieces from src (16-byte aligned) into dst (LDS), all 256 lanes
template <bool DMA, int MAXP, int NT>
__device__ __forceinline__ void stage_issue(const double *src, double *dst, int pieces, int tid, v2d (&hold)[MAXP])
{
    if constexpr(DMA)
    {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for(int it = 0; it < MAXP; it++)
        {
            const int p0 = (it * (NT / 64) + wave) * 64; // first piece of this wave-instruction (wave-uniform)
            if(p0 + lane < pieces)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const v2d *>(src) + p0 + lane,
                                                 (lds_void *)(reinterpret_cast<v2d *>(dst) + p0), 16, 0, 0);
        }
    }
    else
    {
#pragma unroll
        for(int it = 0; it < MAXP; it++)
        {
            const int p = it * NT + tid;
            if(p < pieces)
                hold[it] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(src) + p);
        }
    }
}
template <bool DMA, int MAXP, int NT>
__device__ __forceinline__ void stage_commit(double *dst, int pieces, int tid, v2d (&hold)[MAXP])
{
    if constexpr(!DMA)
    {
#pragma unroll
        for(int it = 0; it < MAXP; it++)
        {
            const int p = it * NT + tid;
            if(p < pieces)
                reinterpret_cast<v2d *>(dst)[p] = hold[it];
        }
    }
}

// win[2*b] = first column of block b's window (even), win[2*b+1] = 16-byte pieces in it
template <int RPT, int K, bool DMA, bool RC, int MAXP, int NT = 256, bool DEFER = false>
__global__ __launch_bounds__(NT) void wkernel(int m, double alpha, const double *__restrict__ val, const int *__restrict__ col,
                                               const int *__restrict__ row_ptr, const int *__restrict__ win,
                                               const double *__restrict__ B, int n, int ldb, double beta, double *__restrict__ C,
                                               int ldc, int cc, int chunk)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int bx  = chunk > 0 ? (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int r0  = bx * NT * RPT;
    if(r0 >= m)
        return;
    const int wmin = win[2 * bx], pieces = win[2 * bx + 1];
    const int j0 = blockIdx.y * cc, j1 = min(n, j0 + cc);
    double *buf0 = lds, *buf1 = lds + 2 * MAXP * NT;
    double         v[RPT][K];
    unsigned short off[RPT][K];
#pragma unroll
    for(int q = 0; q < RPT; q++)
    {
        const int i = r0 + q * NT + tid;
        int       s = 0, e = 0;
        if(i < m)
            s = row_ptr[i], e = row_ptr[i + 1];
#pragma unroll
        for(int k = 0; k < K; k++)
        {
            v[q][k] = 0.0, off[q][k] = 0;
            if(s + k < e)
                v[q][k] = val[s + k], off[q][k] = (unsigned short)(col[s + k] - wmin);
        }
    }
    v2d    hold[MAXP];
    double cinA[RPT], cinB[RPT];
    auto   load_c = [&](int j, double (&cin)[RPT]) {
        if constexpr(RC)
        {
#pragma unroll
            for(int q = 0; q < RPT; q++)
            {
                const int i = r0 + q * NT + tid;
                cin[q]      = i < m ? C[(size_t)i + (size_t)j * ldc] : 0.0;
            }
        }
    };
    // one column: wait (barrier: every lane's DMA / ds_write of column j retired, column j-1 computed, and -- the compiler drains
    // vmcnt at the barrier -- this lane's C values of column j have arrived), put column j+1's C loads and stretch in flight, then
    // compute column j.  The C values of column j+1 are first USED after the next barrier, so nothing waits inside the step.
    double zprev[RPT];
    auto   step = [&](int j, double *cur, double *nxt, double (&cin)[RPT], double (&cnx)[RPT]) {
        if constexpr(DMA)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // hipcc does not count an LDS-DMA as a pending LDS write
        __syncthreads();
        const bool more = j + 1 < j1;
        if constexpr(DEFER)
            if(j > j0) // column j-1's results leave now: they have the whole step to reach L2 before the next vmcnt(0)
            {
#pragma unroll
                for(int q = 0; q < RPT; q++)
                {
                    const int i = r0 + q * NT + tid;
                    if(i < m)
                    {
                        if constexpr(RC)
                            C[(size_t)i + (size_t)(j - 1) * ldc] = zprev[q];
                        else
                            __builtin_nontemporal_store(zprev[q], C + (size_t)i + (size_t)(j - 1) * ldc);
                    }
                }
            }
        if(more)
        {
            load_c(j + 1, cnx);
            stage_issue<DMA, MAXP, NT>(B + (size_t)(j + 1) * ldb + wmin, nxt, pieces, tid, hold);
        }
#pragma unroll
        for(int q = 0; q < RPT; q++)
        {
            const int i = r0 + q * NT + tid;
            double    a = 0.0;
#pragma unroll
            for(int k = 0; k < K; k++)
                a = fma(v[q][k], cur[off[q][k]], a); // (padding entries: 0 * B[wmin], finite operands only in this harness)
            if constexpr(DEFER)
            {
                const double z = alpha * a;
                zprev[q]       = RC ? fma(beta, cin[q], z) : z;
            }
            else if(i < m)
            {
                const double z = alpha * a;
                if constexpr(RC)
                    C[(size_t)i + (size_t)j * ldc] = fma(beta, cin[q], z);
                else
                    __builtin_nontemporal_store(z, C + (size_t)i + (size_t)j * ldc);
            }
        }
        if(more)
            stage_commit<DMA, MAXP, NT>(nxt, pieces, tid, hold);
    };
    load_c(j0, cinA);
    stage_issue<DMA, MAXP, NT>(B + (size_t)j0 * ldb + wmin, buf0, pieces, tid, hold);
    stage_commit<DMA, MAXP, NT>(buf0, pieces, tid, hold);
    for(int j = j0; j < j1; j += 2)
    {
        step(j, buf0, buf1, cinA, cinB);
        if(j + 1 < j1)
            step(j + 1, buf1, buf0, cinB, cinA);
    }
    if constexpr(DEFER)
    {
#pragma unroll
        for(int q = 0; q < RPT; q++)
        {
            const int i = r0 + q * NT + tid;
            if(i < m)
                C[(size_t)i + (size_t)(j1 - 1) * ldc] = zprev[q];
        }
    }
}

// ---- three LDS buffers, two columns in flight, counted vmcnt (stores and the next column's copies stay in flight across the
// barrier).  Every wave issues exactly MAXP copies per column (whole-wave out-of-range slots copy into a scratch KiB), so the
// number of vector-memory operations behind a column's copies is the same for every wave and fits an s_waitcnt immediate.
template <int RPT, int K, bool RC, int MAXP, int NT>
__global__ __launch_bounds__(NT) void w3kernel(int m, double alpha, const double *__restrict__ val, const int *__restrict__ col,
                                               const int *__restrict__ row_ptr, const int *__restrict__ win,
                                               const double *__restrict__ B, int n, int ldb, double beta, double *__restrict__ C,
                                               int ldc, int cc, int chunk)
{
    extern __shared__ double lds[];
    constexpr int NW  = NT / 64;
    const int     tid = threadIdx.x;
    const int     bx  = chunk > 0 ? (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int     r0  = bx * NT * RPT;
    if(r0 >= m)
        return;
    const bool edge = r0 + NT * RPT > m; // rows past m skip their loads / stores: the counts below do not hold
    const int  wmin = win[2 * bx], pieces = win[2 * bx + 1];
    const int  j0 = blockIdx.y * cc, j1 = min(n, j0 + cc);
    double    *bufs[3] = {lds, lds + 2 * MAXP * NT, lds + 4 * MAXP * NT};
    double    *scratch = lds + 6 * MAXP * NT; // NW KiB
    double         v[RPT][K];
    unsigned short off[RPT][K];
#pragma unroll
    for(int q = 0; q < RPT; q++)
    {
        const int i = r0 + q * NT + tid;
        int       s = 0, e = 0;
        if(i < m)
            s = row_ptr[i], e = row_ptr[i + 1];
#pragma unroll
        for(int k = 0; k < K; k++)
        {
            v[q][k] = 0.0, off[q][k] = 0;
            if(s + k < e)
                v[q][k] = val[s + k], off[q][k] = (unsigned short)(col[s + k] - wmin);
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    auto      stage = [&](int j, double *dst) {
        const v2d *src = reinterpret_cast<const v2d *>(B + (size_t)j * ldb + wmin);
#pragma unroll
        for(int it = 0; it < MAXP; it++)
        {
            const int p0 = (it * NW + wave) * 64;
            if(p0 < pieces)
            {
                if(p0 + lane < pieces)
                    __builtin_amdgcn_global_load_lds(src + p0 + lane, (lds_void *)(reinterpret_cast<v2d *>(dst) + p0), 16, 0, 0);
            }
            else
                __builtin_amdgcn_global_load_lds(src + lane, (lds_void *)(reinterpret_cast<v2d *>(scratch) + wave * 64), 16, 0, 0);
        }
    };
    auto load_c = [&](int j, double (&cin)[RPT]) {
        if constexpr(RC)
        {
#pragma unroll
            for(int q = 0; q < RPT; q++)
            {
                const int i = r0 + q * NT + tid;
                cin[q]      = i < m ? C[(size_t)i + (size_t)j * ldc] : 0.0;
            }
        }
    };
    constexpr int N0 = (RC ? RPT : 0) + MAXP;       // behind column j0's copies at step j0: column j0+1's C loads and copies
    constexpr int N1 = (RC ? RPT : 0) + MAXP + RPT; // steady state: + the stores of the step before
    auto step = [&](int j, int waitkind, double *cur, double *nx2, double (&cin)[RPT], double (&cn2)[RPT]) {
        if(edge || waitkind == 2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if(waitkind == 0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N0) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N1) : "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): this lane's LDS reads of the buffer about to be refilled are done
        __builtin_amdgcn_s_barrier();
        if(j + 2 < j1)
        {
            load_c(j + 2, cn2);
            stage(j + 2, nx2);
        }
#pragma unroll
        for(int q = 0; q < RPT; q++)
        {
            const int i = r0 + q * NT + tid;
            double    a = 0.0;
#pragma unroll
            for(int k = 0; k < K; k++)
                a = fma(v[q][k], cur[off[q][k]], a);
            if(i < m)
            {
                const double z = alpha * a;
                if constexpr(RC)
                    C[(size_t)i + (size_t)j * ldc] = fma(beta, cin[q], z);
                else
                    __builtin_nontemporal_store(z, C + (size_t)i + (size_t)j * ldc);
            }
        }
    };
    double cinA[RPT], cinB[RPT], cinC[RPT];
    load_c(j0, cinA);
    stage(j0, bufs[0]);
    if(j0 + 1 < j1)
    {
        load_c(j0 + 1, cinB);
        stage(j0 + 1, bufs[1]);
    }
    // wait kinds: 0 first step, 1 steady, 2 = nothing new was issued in the step before (the last two columns): drain
    int j = j0;
    auto kind = [&](int jj) { return jj + 1 >= j1 ? 2 : (jj == j0 ? 0 : 1); };
    for(; j < j1; j += 3)
    {
        step(j, kind(j), bufs[0], bufs[2], cinA, cinC);
        if(j + 1 < j1)
            step(j + 1, kind(j + 1), bufs[1], bufs[0], cinB, cinA);
        if(j + 2 < j1)
            step(j + 2, kind(j + 2), bufs[2], bufs[1], cinC, cinB);
    }
}

int main(int argc, char **argv)
{
    const int         g    = argc > 1 ? atoi(argv[1]) : 1000;
    const int         n    = argc > 2 ? atoi(argv[2]) : 256;
    const std::string only = argc > 3 ? argv[3] : "";
    const long        m    = (long)g * g;
    std::vector<int>    rp(m + 1), ci;
    std::vector<double> v;
    rp[0] = 0;
    for(long r = 0; r < m; r++)
    {
        const long i = r / g, jj = r % g;
        if(i > 0) ci.push_back((int)(r - g)), v.push_back(-1.0 - 1e-3 * (r % 7));
        if(jj > 0) ci.push_back((int)(r - 1)), v.push_back(-1.0);
        ci.push_back((int)r), v.push_back(4.0 + 1e-3 * (r % 5));
        if(jj < g - 1) ci.push_back((int)(r + 1)), v.push_back(-1.0);
        if(i < g - 1) ci.push_back((int)(r + g)), v.push_back(-1.0 + 1e-3 * (r % 3));
        rp[r + 1] = (int)ci.size();
    }
    const long          nnz = ci.size();
    std::vector<double> B((size_t)m * n);
    for(size_t q = 0; q < B.size(); q++)
        B[q] = sin(0.001 * (double)(q % 100003)) + 1e-7 * (double)(q % 1013);
    int    *d_rp, *d_ci;
    double *d_v, *d_B, *d_C, *d_R;
    CHECK(hipMalloc(&d_rp, (m + 1) * 4));
    CHECK(hipMalloc(&d_ci, nnz * 4));
    CHECK(hipMalloc(&d_v, nnz * 8));
    CHECK(hipMalloc(&d_B, B.size() * 8));
    CHECK(hipMalloc(&d_C, B.size() * 8));
    CHECK(hipMalloc(&d_R, B.size() * 8));
    CHECK(hipMemcpy(d_rp, rp.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_v, v.data(), nnz * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_B, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    // the library's product as the baseline and as the reference bits
    aoclsparse_matrix    A;
    aoclsparse_mat_descr descr;
    OK(aoclsparse_create_mat_descr(&descr));
    OK(aoclsparse_create_dcsr(&A, aoclsparse_index_base_zero, (aoclsparse_int)m, (aoclsparse_int)m, (aoclsparse_int)nnz, rp.data(),
                              ci.data(), v.data()));
    OK(aoclsparse_set_mm_hint(A, aoclsparse_operation_none, descr, 100));
    OK(aoclsparse_optimize(A));
    OK(aoclsparse_mi355_set_pointer_mode(aoclsparse_mi355_pointer_device));
    hipStream_t st = (hipStream_t)aoclsparse_mi355_get_stream();

    std::vector<double> ref(B.size()), got(B.size());
    auto                timeit = [&](const char *name, std::function<void()> fn, double bytes, bool check) {
        if(!only.empty() && std::string(name).find(only) == std::string::npos)
            return;
        for(int w = 0; w < 3; w++)
            fn();
        CHECK(hipStreamSynchronize(st));
        float best = 1e30f, sum = 0;
        const int reps = 10;
        for(int r = 0; r < reps; r++)
        {
            CHECK(hipEventRecord(e0, st));
            fn();
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms), sum += ms;
        }
        const char *verdict = "";
        if(check)
        {
            CHECK(hipMemcpy(got.data(), d_C, got.size() * 8, hipMemcpyDeviceToHost));
            verdict = memcmp(got.data(), ref.data(), got.size() * 8) == 0 ? "exact" : "DIFFERENT";
        }
        printf("%-44s %-9s min %.4f mean %.4f ms  %.2f TB/s\n", name, verdict, best, sum / reps, bytes / best / 1e9);
        fflush(stdout);
    };
    const double bytes_ow = (double)(m + 1 + nnz) * 4 + (double)nnz * 8 + 8.0 * n * 2.0 * m;
    const double bytes_rc = bytes_ow + 8.0 * n * m;

    for(int rc = 0; rc < 2; rc++)
    {
        // C preset to 0.25 (finite): both modes give the same bits for beta = 0
        OK(aoclsparse_mi355_set_csrmm_beta0_overwrite(rc ? 0 : 1));
        auto lib_call = [&] {
            OK(aoclsparse_dcsrmm(aoclsparse_operation_none, 1.0, A, descr, aoclsparse_order_column, d_B, n, (aoclsparse_int)m, 0.0, d_C,
                                 (aoclsparse_int)m));
        };
        CHECK(hipMemset(d_C, 0, B.size() * 8));
        lib_call();
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(ref.data(), d_C, ref.size() * 8, hipMemcpyDeviceToHost));
        timeit(rc ? "library column-major, C read" : "library column-major, overwrite", lib_call, rc ? bytes_rc : bytes_ow, false);

#define RUNW3(RPT, K, MAXP, CC, NT) RUNWX(RPT, K, true, MAXP, CC, NT, false, 3)
#define RUNW2(RPT, K, DMA, MAXP, CC, NT, DEFER) RUNWX(RPT, K, DMA, MAXP, CC, NT, DEFER, 2)
#define RUNW(RPT, K, DMA, MAXP, CC) RUNW2(RPT, K, DMA, MAXP, CC, 256, false)
#define RUNWX(RPT, K, DMA, MAXP, CC, NT, DEFER, NBUF) \
    {                                                                                                                           \
        const int        R = NT * RPT, nb = (int)((m + R - 1) / R), chunk = (nb + 7) / 8;                                      \
        std::vector<int> win(2 * (size_t)chunk * 8, 0);                                                                         \
        bool             fits = true;                                                                                           \
        for(int b = 0; b < nb; b++)                                                                                             \
        {                                                                                                                       \
            const long ra = (long)b * R, rb = std::min<long>(m, ra + R);                                                        \
            int        lo = INT32_MAX, hi = -1;                                                                                 \
            for(long p = rp[ra]; p < rp[rb]; p++)                                                                               \
                lo = std::min(lo, ci[p]), hi = std::max(hi, ci[p]);                                                             \
            lo &= ~1;                                                                                                           \
            int pcs = (hi - lo + 2) / 2;                                                                                        \
            if((long)lo + 2L * pcs > m)                                                                                         \
                pcs = (int)((m - lo) / 2); /* (m even here; an odd tail would take an 8-byte piece) */                         \
            win[2 * b] = lo, win[2 * b + 1] = pcs;                                                                              \
            fits = fits && pcs <= MAXP * NT && 2 * pcs <= 65535;                                                               \
        }                                                                                                                       \
        char name[128];                                                                                                         \
        snprintf(name, sizeof name, "W%d nt%d rpt%d K%d %s%s maxp%d cc%d %s", NBUF, NT, RPT, K, DMA ? "dma" : "reg", DEFER ? " defer" : "", MAXP, CC, rc ? "C read" : "overwrite"); \
        if(!fits)                                                                                                               \
            printf("%-44s window does not fit\n", name);                                                                        \
        else                                                                                                                    \
        {                                                                                                                       \
            int *d_win;                                                                                                         \
            CHECK(hipMalloc(&d_win, win.size() * 4));                                                                           \
            CHECK(hipMemcpy(d_win, win.data(), win.size() * 4, hipMemcpyHostToDevice));                                         \
            const size_t ldsb = NBUF == 3 ? (size_t)3 * 2 * MAXP * NT * 8 + (NT / 64) * 1024 : (size_t)2 * 2 * MAXP * NT * 8;                                                                 \
            auto         kern = NBUF == 3 ? (rc ? w3kernel<RPT, K, true, MAXP, NT> : w3kernel<RPT, K, false, MAXP, NT>)                  \
                                      : (rc ? wkernel<RPT, K, DMA, true, MAXP, NT, DEFER> : wkernel<RPT, K, DMA, false, MAXP, NT, DEFER>); \
            if(ldsb > 160 * 1024) { printf("%-44s LDS %zu too large\n", name, ldsb); } else {                      \
            CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));              \
            CHECK(hipMemset(d_C, 0, B.size() * 8));                                                                             \
            timeit(name, [&] {                                                                                                  \
                hipLaunchKernelGGL(kern, dim3(chunk * 8, (n + CC - 1) / CC), dim3(NT), ldsb, st, (int)m, 1.0, d_v, d_ci, d_rp, \
                                   d_win, d_B, n, (int)m, 0.0, d_C, (int)m, CC, chunk);                                          \
            }, rc ? bytes_rc : bytes_ow, true);                                                                                  \
            }                                                                                                                   \
            CHECK(hipFree(d_win));                                                                                              \
        }                                                                                                                       \
    }
        // R = 1024 rows: window 3026 elements = 1513 pieces -> MAXP 6 (1536); R = 2048: 4050 -> 2025 pieces -> MAXP 8
        RUNW(8, 5, true, 8, 64)
        RUNW2(8, 5, true, 8, 64, 256, true)
        RUNW2(4, 5, true, 4, 64, 512, false)   // R = 2048 with 8 waves
        RUNW2(4, 5, true, 4, 64, 512, true)
        RUNW2(8, 5, true, 6, 64, 512, false)   // R = 4096: window 6096 elements = 3048 pieces <= 6 * 512, 96 KB of LDS
        RUNW2(8, 5, true, 6, 64, 512, true)
        RUNW2(4, 5, true, 3, 64, 1024, false)  // R = 4096 with 16 waves
        RUNW3(8, 5, 6, 64, 512)
        RUNW3(4, 5, 3, 64, 1024)
        RUNW3(4, 5, 4, 64, 512)
        RUNW3(8, 5, 6, 32, 512)
    }
    OK(aoclsparse_mi355_set_csrmm_beta0_overwrite(0));
    return 0;
}
