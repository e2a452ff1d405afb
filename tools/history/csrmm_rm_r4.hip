// csrmm_rm_r4.hip -- diagnostic build (never shipped), round 4: the ROW-major narrow slab (n = 32 columns: what one of 8 ranks owns)
// C = A * B for the 1000^2 Laplacian, "row-union" form against the library's csrmm_tile_kernel.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Iinclude tools/csrmm_rm_r4.hip -Laocl-sparse_amd/lib -laoclsparse_mi355
//         -Wl,-rpath,'$ORIGIN/../../aocl-sparse_amd/lib' -o tools/bin/csrmm_rm_r4
//   csrmm_rm_r4 [g=1000] [n=32]
// Row-union form U<R, NT>: a persistent workgroup walks row blocks of R rows.  Per block the DISTINCT B rows its entries touch
// (for a stencil: a few contiguous runs) are copied into LDS by LDS-DMA, double buffered across blocks (block b+1's rows in
// flight while block b is computed); every entry carries the 16-bit slot of its B row in that buffer instead of a column index.
// A sub-wave of n/2 lanes owns a row (2 columns per lane, 16-byte LDS reads and C stores), the chain per element is CSR order.
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

constexpr int NRUN = 4; // runs of consecutive B rows per block at most

// plan: per block {first row of run r, rows of run r} x NRUN (rows = 0: unused) ; per entry the slot of its B row in the buffer
template <int R, int NT, int N, bool RC>
__global__ __launch_bounds__(NT) void ukernel(int m, int nblocks, double alpha, const double *__restrict__ val,
                                              const unsigned short *__restrict__ slot, const int *__restrict__ row_ptr,
                                              const int *__restrict__ runs, const double *__restrict__ B, int ldb, double beta,
                                              double *__restrict__ C, int ldc, int umax)
{
    extern __shared__ double lds[];
    constexpr int LPR  = N / 2;       // lanes per row
    constexpr int RPP  = NT / LPR;    // rows per pass of the workgroup
    constexpr int PPR  = N / 2;       // 16-byte pieces per B row
    const int     tid  = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int     sub  = tid / LPR, sl = tid % LPR;
    double       *bufs[2] = {lds, lds + (size_t)umax * N};
    auto stage = [&](int b, double *dst) {
        // all pieces of all runs, in run order: piece p of the block -> (row of the union, 16-byte column piece)
        const int *rb = runs + (size_t)b * 2 * NRUN;
        int        done = 0; // rows staged so far
#pragma unroll
        for(int r = 0; r < NRUN; r++)
        {
            const int r0 = rb[2 * r], nr = rb[2 * r + 1];
            const int pieces = nr * PPR;
            for(int p0 = wave * 64; p0 < pieces; p0 += NT)
            {
                const int p = p0 + lane;
                if(p < pieces)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const v2d *>(B + (size_t)(r0 + p / PPR) * ldb) + p % PPR,
                                                     (lds_void *)(reinterpret_cast<v2d *>(dst) + (size_t)done * PPR + p0), 16, 0, 0);
            }
            done += nr;
        }
    };
    int b = blockIdx.x;
    if(b >= nblocks)
        return;
    stage(b, bufs[0]);
    int cur = 0;
    for(; b < nblocks; b += gridDim.x)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nb = b + gridDim.x;
        if(nb < nblocks)
            stage(nb, bufs[cur ^ 1]);
        const double *buf = bufs[cur];
        const int     r0  = b * R;
#pragma unroll
        for(int pass = 0; pass < R / RPP; pass++)
        {
            const int i = r0 + pass * RPP + sub;
            if(i < m)
            {
                const int s = row_ptr[i], e = row_ptr[i + 1];
                v2d       acc = (v2d){0.0, 0.0};
                v2d       cin = (v2d){0.0, 0.0};
                if constexpr(RC)
                    cin = *reinterpret_cast<const v2d *>(C + (size_t)i * ldc + 2 * sl);
                for(int k = s; k < e; k++)
                {
                    const double a  = val[k];
                    const v2d    bv = *reinterpret_cast<const v2d *>(buf + (size_t)slot[k] * N + 2 * sl);
                    acc.x = fma(a, bv.x, acc.x), acc.y = fma(a, bv.y, acc.y);
                }
                v2d o;
                if constexpr(RC)
                    o.x = fma(beta, cin.x, alpha * acc.x), o.y = fma(beta, cin.y, alpha * acc.y);
                else
                    o.x = alpha * acc.x, o.y = alpha * acc.y;
                if constexpr(RC)
                    *reinterpret_cast<v2d *>(C + (size_t)i * ldc + 2 * sl) = o;
                else
                    __builtin_nontemporal_store(o, reinterpret_cast<v2d *>(C + (size_t)i * ldc + 2 * sl));
            }
        }
        cur ^= 1;
    }
}

int main(int argc, char **argv)
{
    const int  g = argc > 1 ? atoi(argv[1]) : 1000;
    const int  n = 32;
    const long m = (long)g * g;
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
    double *d_v, *d_B, *d_C;
    CHECK(hipMalloc(&d_rp, (m + 1) * 4));
    CHECK(hipMalloc(&d_ci, nnz * 4));
    CHECK(hipMalloc(&d_v, nnz * 8));
    CHECK(hipMalloc(&d_B, B.size() * 8));
    CHECK(hipMalloc(&d_C, B.size() * 8));
    CHECK(hipMemcpy(d_rp, rp.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_v, v.data(), nnz * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_B, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
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
        for(int w = 0; w < 3; w++)
            fn();
        CHECK(hipStreamSynchronize(st));
        float     best = 1e30f, sum = 0;
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
        OK(aoclsparse_mi355_set_csrmm_beta0_overwrite(rc ? 0 : 1));
        auto lib_call = [&] {
            OK(aoclsparse_dcsrmm(aoclsparse_operation_none, 1.0, A, descr, aoclsparse_order_row, d_B, n, n, 0.0, d_C, n));
        };
        CHECK(hipMemset(d_C, 0, B.size() * 8));
        lib_call();
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(ref.data(), d_C, ref.size() * 8, hipMemcpyDeviceToHost));
        timeit(rc ? "library row-major slab, C read" : "library row-major slab, overwrite", lib_call, rc ? bytes_rc : bytes_ow, false);
#define RUNU(R, NT, WGS_PER_CU)                                                                                                   \
    {                                                                                                                           \
        const int                   nb = (int)((m + R - 1) / R);                                                                \
        std::vector<int>            runs((size_t)nb * 2 * NRUN, 0);                                                             \
        std::vector<unsigned short> slot(nnz);                                                                                  \
        int                         umax = 0;                                                                                   \
        bool                        fits = true;                                                                                \
        for(int b = 0; b < nb; b++)                                                                                             \
        {                                                                                                                       \
            const long ra = (long)b * R, rb = std::min<long>(m, ra + R);                                                        \
            std::vector<int> rows(ci.begin() + rp[ra], ci.begin() + rp[rb]);                                                    \
            std::sort(rows.begin(), rows.end());                                                                                \
            rows.erase(std::unique(rows.begin(), rows.end()), rows.end());                                                      \
            int nr = 0;                                                                                                         \
            for(size_t q = 0; q < rows.size();)                                                                                 \
            {                                                                                                                   \
                size_t e = q + 1;                                                                                               \
                while(e < rows.size() && rows[e] == rows[e - 1] + 1)                                                            \
                    e++;                                                                                                        \
                if(nr == NRUN)                                                                                                  \
                {                                                                                                               \
                    fits = false;                                                                                               \
                    break;                                                                                                      \
                }                                                                                                               \
                runs[((size_t)b * NRUN + nr) * 2] = rows[q], runs[((size_t)b * NRUN + nr) * 2 + 1] = (int)(e - q);              \
                nr++, q = e;                                                                                                    \
            }                                                                                                                   \
            umax = std::max(umax, (int)rows.size());                                                                            \
            for(int p = rp[ra]; p < rp[rb]; p++)                                                                                \
                slot[p] = (unsigned short)(std::lower_bound(rows.begin(), rows.end(), ci[p]) - rows.begin());                   \
        }                                                                                                                       \
        char name[128];                                                                                                         \
        snprintf(name, sizeof name, "U R%d nt%d wg/cu%d %s", R, NT, WGS_PER_CU, rc ? "C read" : "overwrite");                   \
        const size_t ldsb = (size_t)2 * umax * n * 8;                                                                           \
        if(!fits || ldsb > 160 * 1024)                                                                                          \
            printf("%-44s does not fit (umax %d)\n", name, umax);                                                               \
        else                                                                                                                    \
        {                                                                                                                       \
            int            *d_runs;                                                                                             \
            unsigned short *d_slot;                                                                                             \
            CHECK(hipMalloc(&d_runs, runs.size() * 4));                                                                         \
            CHECK(hipMalloc(&d_slot, slot.size() * 2));                                                                         \
            CHECK(hipMemcpy(d_runs, runs.data(), runs.size() * 4, hipMemcpyHostToDevice));                                      \
            CHECK(hipMemcpy(d_slot, slot.data(), slot.size() * 2, hipMemcpyHostToDevice));                                      \
            auto kern = rc ? ukernel<R, NT, 32, true> : ukernel<R, NT, 32, false>;                                              \
            CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));              \
            CHECK(hipMemset(d_C, 0, B.size() * 8));                                                                             \
            const int grid = std::min(nb, 256 * WGS_PER_CU);                                                                    \
            timeit(name, [&] {                                                                                                  \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), ldsb, st, (int)m, nb, 1.0, d_v, d_slot, d_rp, d_runs, d_B, n,     \
                                   0.0, d_C, n, umax);                                                                          \
            }, rc ? bytes_rc : bytes_ow, true);                                                                                  \
            CHECK(hipFree(d_runs));                                                                                             \
            CHECK(hipFree(d_slot));                                                                                             \
        }                                                                                                                       \
    }
        RUNU(32, 256, 4)
        RUNU(32, 256, 6)
        RUNU(64, 256, 3)
        RUNU(64, 512, 2)
        RUNU(64, 512, 3)
        RUNU(128, 512, 1)
        RUNU(128, 1024, 1)
    }
    OK(aoclsparse_mi355_set_csrmm_beta0_overwrite(0));
    return 0;
}
