// mfma_f64_probe.hip -- diagnostic build (never shipped): does v_mfma_f64_16x16x4_f64 beat the vector FMA path on the
// dense r x L blocks the csrmm row-group detector finds (shell-like: 5 x 35, flan-like: 3 x 81)?  BASELINE.json's north
// star asks for "an ELL/blocked-ELL variant that feeds MFMA only where nnz/row is uniform enough to form dense tiles";
// this measures that variant against csrmm_rowgroup_kernel's arithmetic on the same data.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/mfma_f64_probe.hip -o tools/bin/mfma_f64_probe
//   mfma_f64_probe [groups=300000] [r=5] [L=35] [n=256]
// Problem: G row groups; group g owns r rows sharing L column indices (7 neighbour nodes x 5 dofs when r = 5), A values
// dense r x L, B row-major k x n.  C[g*r + q, :] = sum_k a[q][k] * B[col[k], :], chain in k order (the csrmm chain).
//   V  = vector path: one wavefront per (group, 128 columns), each B row loaded once (16 B per lane), r FMA chains.
//   M  = MFMA path: the group's r rows padded to the 16-row M tile; per 4 values of k and per 16-column N tile one
//        v_mfma_f64_16x16x4_f64 (A fragment: lane -> (row = lane % 16, k = lane / 16); B fragment: lane -> (column =
//        lane % 16, k = lane / 16), 8 bytes per lane; D: 4 doubles per lane).
// Both must give the same bits (one FMA chain per element in k order); the time per launch is printed with the useful
// TFLOP/s (2 * r * L * n per group) and the fraction of issued MFMA flops that are useful (r / 16).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int RMAX = 8;

// V: wavefront per (group, 128-column chunk); 8 B rows in flight per step like csrmm_rowgroup_kernel
template <int R>
__global__ __launch_bounds__(256) void vec_kernel(int G, int L, const int *__restrict__ col, const double *__restrict__ a,
                                                  const double *__restrict__ B, int n, double *__restrict__ C)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = blockIdx.x * 4 + w;
    const int j = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(g >= G || j >= n)
        return;
    const int    *cg = col + (size_t)g * L;
    const double *ag = a + (size_t)g * R * L; // a[q][k]
    double        acc0[R], acc1[R];
#pragma unroll
    for(int q = 0; q < R; q++)
        acc0[q] = 0, acc1[q] = 0;
    int k = 0;
    for(; k + 8 <= L; k += 8)
    {
        v2d b[8];
#pragma unroll
        for(int u = 0; u < 8; u++)
            b[u] = *reinterpret_cast<const v2d *>(B + (size_t)cg[k + u] * n + j);
#pragma unroll
        for(int q = 0; q < R; q++)
#pragma unroll
            for(int u = 0; u < 8; u++)
            {
                const double av = ag[q * L + k + u];
                acc0[q] = fma(av, b[u].x, acc0[q]), acc1[q] = fma(av, b[u].y, acc1[q]);
            }
    }
    for(; k < L; k++)
    {
        const v2d b = *reinterpret_cast<const v2d *>(B + (size_t)cg[k] * n + j);
#pragma unroll
        for(int q = 0; q < R; q++)
        {
            const double av = ag[q * L + k];
            acc0[q] = fma(av, b.x, acc0[q]), acc1[q] = fma(av, b.y, acc1[q]);
        }
    }
#pragma unroll
    for(int q = 0; q < R; q++)
    {
        v2d o;
        o.x = acc0[q], o.y = acc1[q];
        *reinterpret_cast<v2d *>(C + ((size_t)g * R + q) * n + j) = o;
    }
}

// layout self-test: A[i][k] = (k == 0) * (i + 1), B[k][j] = (k == 0) * 100 * (j + 1)  ->  D[i][j] = 100 (i + 1)(j + 1);
// every lane dumps its four D registers so the host can decode which (i, j) each one holds
__global__ void layout_kernel(double *out)
{
    const int lane = threadIdx.x, row = lane & 15, kk = lane >> 4;
    const double a = kk == 0 ? (double)(row + 1) : 0.0, b = kk == 0 ? 100.0 * (row + 1) : 0.0;
    v4d          acc = (v4d){0, 0, 0, 0};
    acc              = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for(int i = 0; i < 4; i++)
        out[lane * 4 + i] = acc[i];
}

// M: wavefront per (group, 128-column chunk) = 8 N tiles of 16 columns; K walked 4 at a time
template <int R>
__global__ __launch_bounds__(256) void mfma_kernel(int G, int L, const int *__restrict__ col, const double *__restrict__ a,
                                                   const double *__restrict__ B, int n, double *__restrict__ C,
                                                   const int *__restrict__ drow, const int *__restrict__ dcol)
{
    const int w    = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g    = blockIdx.x * 4 + w;
    const int lane = threadIdx.x & 63;
    const int j0   = 128 * (int)blockIdx.y;
    if(g >= G || j0 >= n)
        return;
    const int    *cg = col + (size_t)g * L;
    const double *ag = a + (size_t)g * R * L;
    const int     row = lane & 15, kk = lane >> 4; // A fragment: (row, k); B fragment: (column, k)
    v4d           acc[8];
#pragma unroll
    for(int t = 0; t < 8; t++)
        acc[t] = (v4d){0, 0, 0, 0};
    for(int k = 0; k < L; k += 4)
    {
        const int    kq = k + kk;
        const double av = (row < R && kq < L) ? ag[row * L + kq] : 0.0; // rows R..15 and k >= L are padding
        const int    c  = kq < L ? cg[kq] : cg[0];
        double       bv[8];
#pragma unroll
        for(int t = 0; t < 8; t++)
            bv[t] = kq < L ? B[(size_t)c * n + j0 + 16 * t + row] : 0.0;
#pragma unroll
        for(int t = 0; t < 8; t++)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], acc[t], 0, 0, 0);
    }
    // D layout: (row, column) of register i of this lane, decoded by the self-test at start-up
#pragma unroll
    for(int t = 0; t < 8; t++)
#pragma unroll
        for(int i = 0; i < 4; i++)
        {
            const int q = drow[lane * 4 + i];
            if(q < R)
                C[((size_t)g * R + q) * n + j0 + 16 * t + dcol[lane * 4 + i]] = acc[t][i];
        }
}

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 300000;
    const int r = argc > 2 ? atoi(argv[2]) : 5;
    const int L = argc > 3 ? atoi(argv[3]) : 35;
    const int n = argc > 4 ? atoi(argv[4]) : 256;
    if(r < 1 || r > RMAX || n % 128)
    {
        printf("r in 1..8, n a multiple of 128\n");
        return 1;
    }
    const long          K = (long)G * r; // rows of B (square problem)
    std::vector<int>    col((size_t)G * L);
    std::vector<double> a((size_t)G * r * L), B((size_t)K * n);
    unsigned            s = 12345;
    auto                rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for(int g = 0; g < G; g++)
    {
        // L columns = L / r neighbour nodes x r dofs, neighbours near the diagonal (like a structured mesh)
        const int nb = (L + r - 1) / r;
        for(int k = 0; k < L; k++)
        {
            long node = (long)g + (k / r - nb / 2) * (k / r % 2 ? 600 : 1);
            node      = std::min<long>(std::max<long>(node, 0), G - 1);
            col[(size_t)g * L + k] = (int)(node * r + k % r);
        }
        for(int q = 0; q < r * L; q++)
            a[(size_t)g * r * L + q] = (double)(rnd() % 2001) / 1000.0 - 1.0;
    }
    for(size_t q = 0; q < B.size(); q++)
        B[q] = (double)(rnd() % 20001) / 10000.0 - 1.0;
    int    *d_col;
    double *d_a, *d_B, *d_C1, *d_C2;
    CHECK(hipMalloc(&d_col, col.size() * 4));
    CHECK(hipMalloc(&d_a, a.size() * 8));
    CHECK(hipMalloc(&d_B, B.size() * 8));
    CHECK(hipMalloc(&d_C1, B.size() * 8));
    CHECK(hipMalloc(&d_C2, B.size() * 8));
    CHECK(hipMemcpy(d_col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_a, a.data(), a.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_B, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    // decode the D layout
    double *d_lay;
    int    *d_drow, *d_dcol;
    CHECK(hipMalloc(&d_lay, 256 * 8));
    CHECK(hipMalloc(&d_drow, 256 * 4));
    CHECK(hipMalloc(&d_dcol, 256 * 4));
    layout_kernel<<<1, 64>>>(d_lay);
    std::vector<double> lay(256);
    CHECK(hipMemcpy(lay.data(), d_lay, 256 * 8, hipMemcpyDeviceToHost));
    std::vector<int> drow(256), dcol(256);
    bool             assumed = true;
    for(int q = 0; q < 256; q++)
    {
        const int prod = (int)(lay[q] / 100.0 + 0.5); // (i + 1)(j + 1): not unique, so use the lane's column
        // B fragment column of lane l is l % 16 and the D column of a lane is its own column in every known layout
        const int jx = (q / 4) % 16;
        dcol[q]      = jx;
        drow[q]      = prod / (jx + 1) - 1;
        assumed      = assumed && drow[q] == 4 * ((q / 4) / 16) + q % 4 && prod % (jx + 1) == 0;
    }
    CHECK(hipMemcpy(d_drow, drow.data(), 256 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dcol, dcol.data(), 256 * 4, hipMemcpyHostToDevice));
    const dim3 grid((G + 3) / 4, n / 128);
    auto       runv = [&](double *C) {
        switch(r)
        {
        case 3: vec_kernel<3><<<grid, 256>>>(G, L, d_col, d_a, d_B, n, C); break;
        case 5: vec_kernel<5><<<grid, 256>>>(G, L, d_col, d_a, d_B, n, C); break;
        default: vec_kernel<8><<<grid, 256>>>(G, L, d_col, d_a, d_B, n, C); break;
        }
    };
    auto runm = [&](double *C) {
        switch(r)
        {
        case 3: mfma_kernel<3><<<grid, 256>>>(G, L, d_col, d_a, d_B, n, C, d_drow, d_dcol); break;
        case 5: mfma_kernel<5><<<grid, 256>>>(G, L, d_col, d_a, d_B, n, C, d_drow, d_dcol); break;
        default: mfma_kernel<8><<<grid, 256>>>(G, L, d_col, d_a, d_B, n, C, d_drow, d_dcol); break;
        }
    };
    if(r != 3 && r != 5 && r != 8)
    {
        printf("r must be 3, 5 or 8 in this probe\n");
        return 1;
    }
    CHECK(hipMemset(d_C1, 0xff, B.size() * 8));
    CHECK(hipMemset(d_C2, 0xff, B.size() * 8));
    runv(d_C1);
    runm(d_C2);
    CHECK(hipDeviceSynchronize());
    std::vector<double> c1((size_t)1000 * n), c2((size_t)1000 * n);
    const size_t        off = (size_t)(G / 2) * r * n;
    CHECK(hipMemcpy(c1.data(), d_C1 + off, c1.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(c2.data(), d_C2 + off, c2.size() * 8, hipMemcpyDeviceToHost));
    // host reference of the first sampled row: the k-ordered FMA chain
    const int g0 = G / 2;
    bool      ref_ok = true;
    for(int jx = 0; jx < n; jx++)
    {
        double acc = 0;
        for(int k = 0; k < L; k++)
            acc = fma(a[(size_t)g0 * r * L + k], B[(size_t)col[(size_t)g0 * L + k] * n + jx], acc);
        ref_ok = ref_ok && acc == c1[jx];
    }
    const bool same = !memcmp(c1.data(), c2.data(), c1.size() * 8);
    double     maxdiff = 0, maxabs = 0;
    for(size_t q = 0; q < c1.size(); q++)
    {
        maxdiff = std::max(maxdiff, std::fabs(c1[q] - c2[q]));
        maxabs  = std::max(maxabs, std::fabs(c1[q]));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best[2] = {1e30f, 1e30f};
    for(int rep = 0; rep < 5; rep++)
        for(int v = 0; v < 2; v++)
        {
            CHECK(hipEventRecord(e0));
            for(int q = 0; q < 3; q++)
                v == 0 ? runv(d_C1) : runm(d_C2);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best[v] = std::min(best[v], ms / 3);
        }
    const double flop = 2.0 * G * r * L * n;
    printf("{\"probe\": \"mfma_f64_16x16x4 vs vector FMA on dense r x L row-group blocks\", \"groups\": %d, \"r\": %d, \"L\": %d, "
           "\"n\": %d, \"vector_ms\": %.4f, \"mfma_ms\": %.4f, \"vector_useful_tflops\": %.2f, \"mfma_useful_tflops\": %.2f, "
           "\"mfma_issued_tflops\": %.2f, \"mfma_tile_fill\": %.3f, \"vector_matches_host_chain\": %s, "
           "\"mfma_bits_equal_vector\": %s, \"max_abs_diff\": %.3e, \"max_abs_value\": %.3e, "
           "\"d_layout_is_4x_lane_div16_plus_reg\": %s}\n",
           G, r, L, n, best[0], best[1], flop / best[0] / 1e9, flop / best[1] / 1e9,
           2.0 * G * 16 * ((L + 3) / 4 * 4) * (double)n / best[1] / 1e9, r / 16.0, ref_ok ? "true" : "false",
           same ? "true" : "false", maxdiff, maxabs, assumed ? "true" : "false");
    return 0;
}
