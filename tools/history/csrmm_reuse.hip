// csrmm_reuse.hip -- diagnostic build (never shipped): does re-using the B-row pieces of the PREVIOUS row of A
// (kept in registers, matched by column index) lift row-major csrmm off the L2 -> CU bandwidth?
// PMC says HBM traffic is only 1.21x algorithmic (tools/pmc_sum.py), while every B row crosses L2 -> L1 five times.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/csrmm_reuse.hip -o tools/bin/csrmm_reuse
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
    do                                                                           \
    {                                                                            \
        hipError_t e = (x);                                                      \
        if(e != hipSuccess)                                                      \
        {                                                                        \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                             \
        }                                                                        \
    } while(0)

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int xcd_row(int bx, int chunk)
{
    return (bx & 7) * chunk + (bx >> 3);
}

// shipped shape: wave per (row, 128-col chunk), 4 waves/WG
__global__ __launch_bounds__(256) void k0(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = xcd_row(blockIdx.x, chunk) * 4 + w;
    const int j = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(i >= m || j >= n)
        return;
    const int     s = row_ptr[i], e = row_ptr[i + 1];
    v2d           a  = {0, 0};
    const double *Bj = B + j;
    int           p  = s;
    for(; p + 4 <= e; p += 4)
    {
        const double v0 = val[p], v1 = val[p + 1], v2 = val[p + 2], v3 = val[p + 3];
        const v2d    b0 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p] * n);
        const v2d    b1 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p + 1] * n);
        const v2d    b2 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p + 2] * n);
        const v2d    b3 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p + 3] * n);
        a.x = fma(v0, b0.x, a.x), a.y = fma(v0, b0.y, a.y);
        a.x = fma(v1, b1.x, a.x), a.y = fma(v1, b1.y, a.y);
        a.x = fma(v2, b2.x, a.x), a.y = fma(v2, b2.y, a.y);
        a.x = fma(v3, b3.x, a.x), a.y = fma(v3, b3.y, a.y);
    }
    for(; p < e; p++)
    {
        const double v0 = val[p];
        const v2d    b0 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p] * n);
        a.x = fma(v0, b0.x, a.x), a.y = fma(v0, b0.y, a.y);
    }
    v2d *cp = reinterpret_cast<v2d *>(C + (size_t)i * n + j);
    if(readc)
    {
        const v2d c = *cp;
        a.x = fma(beta, c.x, a.x), a.y = fma(beta, c.y, a.y);
    }
    *cp = a;
}

// kr: a wave walks RW consecutive rows of one 128-column chunk and keeps the <= P pieces of B it loaded for the
// previous row, keyed by column (wave-uniform compare); a row entry whose column is among them costs no load.
// WAVES waves per workgroup.
template <int RW, int P, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void kr(int m, const double *__restrict__ val, const int *__restrict__ col,
                                                 const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                                 double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w  = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i0 = (xcd_row(blockIdx.x, chunk) * WAVES + w) * RW;
    const int j  = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(i0 >= m || j >= n)
        return;
    const double *Bj = B + j;
    int           pc[P];
    v2d           pb[P];
#pragma unroll
    for(int q = 0; q < P; q++)
        pc[q] = -1, pb[q] = v2d{0, 0};
    for(int r = 0; r < RW; r++)
    {
        const int i = i0 + r;
        if(i >= m)
            break;
        const int s = row_ptr[i], e = row_ptr[i + 1], len = e - s;
        v2d       a = {0, 0};
        if(len <= P)
        {
            int    cc[P], hq[P];
            double vv[P];
            v2d    bb[P];
#pragma unroll
            for(int u = 0; u < P; u++)
            {
                cc[u] = u < len ? col[s + u] : -2;
                vv[u] = u < len ? val[s + u] : 0.0;
                hq[u] = -1;
#pragma unroll
                for(int q = 0; q < P; q++)
                    if(cc[u] == pc[q])
                        hq[u] = q;
            }
#pragma unroll
            for(int u = 0; u < P; u++)
                if(u < len && hq[u] < 0)
                    bb[u] = *reinterpret_cast<const v2d *>(Bj + (size_t)cc[u] * n);
#pragma unroll
            for(int u = 0; u < P; u++)
            {
#pragma unroll
                for(int q = 0; q < P; q++)
                    if(hq[u] == q)
                        bb[u] = pb[q];
            }
#pragma unroll
            for(int u = 0; u < P; u++)
                if(u < len)
                    a.x = fma(vv[u], bb[u].x, a.x), a.y = fma(vv[u], bb[u].y, a.y);
#pragma unroll
            for(int u = 0; u < P; u++)
                pc[u] = cc[u], pb[u] = bb[u];
        }
        else
        {
            for(int p = s; p < e; p++)
            {
                const double v0 = val[p];
                const v2d    b0 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p] * n);
                a.x = fma(v0, b0.x, a.x), a.y = fma(v0, b0.y, a.y);
            }
#pragma unroll
            for(int q = 0; q < P; q++)
                pc[q] = -1;
        }
        v2d *cp = reinterpret_cast<v2d *>(C + (size_t)i * n + j);
        if(readc)
        {
            const v2d c = *cp;
            a.x = fma(beta, c.x, a.x), a.y = fma(beta, c.y, a.y);
        }
        *cp = a;
    }
}

// k5p: PROBE for the hypothesis only (5-entry rows whose 2nd/3rd columns equal the previous row's 3rd/4th, i.e. the
// interior of the 5-pt stencil): a sliding window of three pieces in fixed registers, three loads per row.
template <int RW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k5p(int m, const double *__restrict__ val, const int *__restrict__ col,
                                                  const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                                  double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w  = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i0 = (xcd_row(blockIdx.x, chunk) * WAVES + w) * RW;
    const int j  = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(i0 >= m || j >= n)
        return;
    const double *Bj = B + j;
    int           c2 = -1, c3 = -1; // columns of the pieces kept from the previous row
    v2d           b2 = {0, 0}, b3 = {0, 0};
    for(int r = 0; r < RW; r++)
    {
        const int i = i0 + r;
        if(i >= m)
            break;
        const int s = row_ptr[i], e = row_ptr[i + 1];
        v2d       a = {0, 0};
        if(e - s == 5)
        {
            const int    k0 = col[s], k1 = col[s + 1], k2 = col[s + 2], k3 = col[s + 3], k4 = col[s + 4];
            const double v0 = val[s], v1 = val[s + 1], v2 = val[s + 2], v3 = val[s + 3], v4 = val[s + 4];
            const bool   slide = k1 == c2 && k2 == c3;
            const v2d    n0 = *reinterpret_cast<const v2d *>(Bj + (size_t)k0 * n);
            v2d          n1, n2;
            if(!slide)
            {
                n1 = *reinterpret_cast<const v2d *>(Bj + (size_t)k1 * n);
                n2 = *reinterpret_cast<const v2d *>(Bj + (size_t)k2 * n);
            }
            const v2d n3 = *reinterpret_cast<const v2d *>(Bj + (size_t)k3 * n);
            const v2d n4 = *reinterpret_cast<const v2d *>(Bj + (size_t)k4 * n);
            if(slide)
                n1 = b2, n2 = b3;
            a.x = fma(v0, n0.x, a.x), a.y = fma(v0, n0.y, a.y);
            a.x = fma(v1, n1.x, a.x), a.y = fma(v1, n1.y, a.y);
            a.x = fma(v2, n2.x, a.x), a.y = fma(v2, n2.y, a.y);
            a.x = fma(v3, n3.x, a.x), a.y = fma(v3, n3.y, a.y);
            a.x = fma(v4, n4.x, a.x), a.y = fma(v4, n4.y, a.y);
            c2 = k2, c3 = k3, b2 = n2, b3 = n3;
        }
        else
        {
            for(int p = s; p < e; p++)
            {
                const double v0 = val[p];
                const v2d    b0 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p] * n);
                a.x = fma(v0, b0.x, a.x), a.y = fma(v0, b0.y, a.y);
            }
            c2 = c3 = -1;
        }
        v2d *cp = reinterpret_cast<v2d *>(C + (size_t)i * n + j);
        if(readc)
        {
            const v2d c = *cp;
            a.x = fma(beta, c.x, a.x), a.y = fma(beta, c.y, a.y);
        }
        *cp = a;
    }
}

// ks: a workgroup of 4 wavefronts owns R consecutive rows of one 128-column chunk.  The distinct B rows its rows touch
// (<= UCAP) are loaded ONCE into LDS, each by one wavefront, and every row then takes its pieces from LDS: the
// L2 -> CU traffic per row drops from len to (distinct rows)/R pieces.  Generic: the distinct set is found on the
// fly (one wavefront, one column per lane, 64 rounds of readlane + ballot); rows longer than ML entries or a set larger
// than UCAP make the workgroup fall back to direct loads.
template <int R, int ML, int UCAP>
__global__ __launch_bounds__(256) void ks(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n, double beta,
                                          double *__restrict__ C, bool readc, int chunk)
{
    static_assert(R * ML <= 64, "one lane per (row, entry)");
    __shared__ v2d tile[UCAP][64];
    __shared__ int s_slot[R * ML];
    __shared__ int s_ucol[UCAP];
    __shared__ int s_nu;
    const int w    = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int i0   = xcd_row(blockIdx.x, chunk) * R;
    const int j    = 2 * lane + 128 * (int)blockIdx.y;
    if(i0 >= m)
        return;
    const double *Bj = B + j;
    // wave 0: one lane per (row, entry)
    if(w == 0)
    {
        const int rr = lane / ML, kk = lane % ML, i = i0 + rr;
        int       c  = -1;
        bool      too_long = false;
        if(i < m && rr < R)
        {
            const int s = row_ptr[i], e = row_ptr[i + 1];
            if(kk < e - s)
                c = col[s + kk];
            too_long = e - s > ML;
        }
        const bool any_long = __ballot(too_long) != 0ull;
        int        owner    = lane;
        for(int u = 0; u < 64; u++)
        {
            const int cu = __builtin_amdgcn_readlane(c, u);
            if(cu >= 0 && c == cu && owner == lane && lane > u)
                owner = u;
        }
        const unsigned long long uniq = __ballot(c >= 0 && owner == lane);
        const int                nu   = __popcll(uniq);
        const int                myslot = __popcll(uniq & ((1ull << lane) - 1ull));
        if(c >= 0 && owner == lane && myslot < UCAP)
            s_ucol[myslot] = c;
        // slot of my owner
        const int oslot = __shfl(myslot, owner);
        if(lane < R * ML)
            s_slot[lane] = c >= 0 ? oslot : -1;
        if(lane == 0)
            s_nu = (any_long || nu > UCAP) ? -1 : nu;
    }
    __syncthreads();
    const int nu = s_nu;
    if(nu >= 0)
    {
        for(int u = w; u < nu; u += 4)
            tile[u][lane] = *reinterpret_cast<const v2d *>(Bj + (size_t)s_ucol[u] * n);
        __syncthreads();
        for(int rr = w; rr < R; rr += 4)
        {
            const int i = i0 + rr;
            if(i >= m)
                break;
            const int s = row_ptr[i], e = row_ptr[i + 1];
            v2d       a = {0, 0};
            for(int p = s; p < e; p++)
            {
                const double v0 = val[p];
                const v2d    b0 = tile[s_slot[rr * ML + (p - s)]][lane];
                a.x = fma(v0, b0.x, a.x), a.y = fma(v0, b0.y, a.y);
            }
            v2d *cp = reinterpret_cast<v2d *>(C + (size_t)i * n + j);
            if(readc)
            {
                const v2d c = *cp;
                a.x = fma(beta, c.x, a.x), a.y = fma(beta, c.y, a.y);
            }
            *cp = a;
        }
    }
    else
    {
        for(int rr = w; rr < R; rr += 4)
        {
            const int i = i0 + rr;
            if(i >= m)
                break;
            const int s = row_ptr[i], e = row_ptr[i + 1];
            v2d       a = {0, 0};
            for(int p = s; p < e; p++)
            {
                const double v0 = val[p];
                const v2d    b0 = *reinterpret_cast<const v2d *>(Bj + (size_t)col[p] * n);
                a.x = fma(v0, b0.x, a.x), a.y = fma(v0, b0.y, a.y);
            }
            v2d *cp = reinterpret_cast<v2d *>(C + (size_t)i * n + j);
            if(readc)
            {
                const v2d c = *cp;
                a.x = fma(beta, c.x, a.x), a.y = fma(beta, c.y, a.y);
            }
            *cp = a;
        }
    }
}

__global__ void kdiff(size_t n, const double *a, const double *b, unsigned long long *cnt)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if(i < n && !(a[i] == b[i]))
        atomicAdd(cnt, 1ull);
}

int main(int argc, char **argv)
{
    const int  g = argc > 1 ? atoi(argv[1]) : 1000;
    const int  n = argc > 2 ? atoi(argv[2]) : 256;
    const long m = (long)g * g;
    std::vector<int>    rp(m + 1), ci;
    std::vector<double> v;
    rp[0] = 0;
    for(long r = 0; r < m; r++)
    {
        const long i = r / g, j = r % g;
        if(i > 0) ci.push_back(r - g), v.push_back(-1.0 - 1e-3 * (r % 7));
        if(j > 0) ci.push_back(r - 1), v.push_back(-1.0);
        ci.push_back(r), v.push_back(4.0 + 1e-3 * (r % 5));
        if(j < g - 1) ci.push_back(r + 1), v.push_back(-1.0);
        if(i < g - 1) ci.push_back(r + g), v.push_back(-1.0 + 1e-3 * (r % 3));
        rp[r + 1] = (int)ci.size();
    }
    const long nnz = ci.size();
    const size_t nb = (size_t)m * n;
    std::vector<double> B(nb);
    for(size_t q = 0; q < nb; q++)
        B[q] = sin(0.001 * (double)(q % 100003));
    int    *d_rp, *d_ci;
    double *d_v, *d_B, *d_C, *d_R;
    unsigned long long *d_cnt;
    CHECK(hipMalloc(&d_rp, (m + 1) * 4)); CHECK(hipMalloc(&d_ci, nnz * 4)); CHECK(hipMalloc(&d_v, nnz * 8));
    CHECK(hipMalloc(&d_B, nb * 8)); CHECK(hipMalloc(&d_C, nb * 8)); CHECK(hipMalloc(&d_R, nb * 8)); CHECK(hipMalloc(&d_cnt, 8));
    CHECK(hipMemcpy(d_rp, rp.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_v, v.data(), nnz * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_B, B.data(), nb * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    constexpr int NV = 10;
    const char *names[NV] = {"k0 shipped shape", "ks R=4 ML=8 U=16 (LDS)", "k5p RW=2 4 waves", "ks R=8 ML=8 U=32 (LDS)", "ks R=8 ML=8 U=28 (LDS)",
                             "ks R=12 ML=5 U=40", "ks R=4 ML=16 U=16", "k5p RW=8 2 waves", "k5p RW=16 1 wave", "k5p RW=64 4 waves"};
    for(int pass = 0; pass < 2; pass++)
    {
        const bool   readc = pass == 1;
        const double beta  = readc ? -2.0 : 0.0;
        double       best[NV];
        unsigned long long bad[NV] = {0};
        for(int q = 0; q < NV; q++) best[q] = 1e30;
        for(int rep = 0; rep < 6; rep++)
            for(int q = 0; q < NV; q++)
            {
                if(readc) CHECK(hipMemset(d_C, 0, nb * 8));
                auto launch = [&](int rows_per_wg) {
                    int nbx = (int)((m + rows_per_wg - 1) / rows_per_wg), chunk = (nbx + 7) / 8;
                    return dim3(chunk * 8, (n + 127) / 128);
                };
                dim3 gr;
                int  ch;
                CHECK(hipEventRecord(e0));
#define ARGS (int)m, d_v, d_ci, d_rp, d_B, n, beta, d_C, readc, ch
                switch(q)
                {
                case 0: gr = launch(4), ch = gr.x / 8; k0<<<gr, 256>>>(ARGS); break;
                case 1: gr = launch(4), ch = gr.x / 8; ks<4, 8, 16><<<gr, 256>>>(ARGS); break;
                case 2: gr = launch(8), ch = gr.x / 8; k5p<2, 4><<<gr, 256>>>(ARGS); break;
                case 3: gr = launch(8), ch = gr.x / 8; ks<8, 8, 32><<<gr, 256>>>(ARGS); break;
                case 4: gr = launch(8), ch = gr.x / 8; ks<8, 8, 28><<<gr, 256>>>(ARGS); break;
                case 5: gr = launch(12), ch = gr.x / 8; ks<12, 5, 40><<<gr, 256>>>(ARGS); break;
                case 6: gr = launch(4), ch = gr.x / 8; ks<4, 16, 16><<<gr, 256>>>(ARGS); break;
                case 7: gr = launch(16), ch = gr.x / 8; k5p<8, 2><<<gr, 128>>>(ARGS); break;
                case 8: gr = launch(16), ch = gr.x / 8; k5p<16, 1><<<gr, 64>>>(ARGS); break;
                default: gr = launch(256), ch = gr.x / 8; k5p<64, 4><<<gr, 256>>>(ARGS); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best[q] = std::min(best[q], (double)ms);
                if(rep == 0)
                {
                    if(q == 0)
                        CHECK(hipMemcpy(d_R, d_C, nb * 8, hipMemcpyDeviceToDevice));
                    else
                    {
                        CHECK(hipMemset(d_cnt, 0, 8));
                        kdiff<<<(unsigned)((nb + 255) / 256), 256>>>(nb, d_C, d_R, d_cnt);
                        CHECK(hipMemcpy(&bad[q], d_cnt, 8, hipMemcpyDeviceToHost));
                    }
                }
            }
        const double bytes = (double)(m + 1 + nnz) * 4 + nnz * 8.0 + (double)nb * 8 * (readc ? 3 : 2);
        printf("beta=%g  n=%d (algorithmic %.3f GB)\n", beta, n, bytes / 1e9);
        for(int q = 0; q < NV; q++)
            printf("  %-22s %.4f ms  %.2f TB/s  %.1f%%  mismatches %llu\n", names[q], best[q], bytes / best[q] / 1e9,
                   bytes / best[q] / 1e9 / 80, bad[q]);
    }
    return 0;
}
