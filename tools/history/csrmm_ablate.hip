// csrmm_ablate.hip -- diagnostic build (never shipped): row-major csrmm C = A*B (beta = 0 or not) on the
// 5-pt Laplacian, n = 256 columns.  Variants of the wave-per-row kernel, timed interleaved.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/csrmm_ablate.hip -o tools/bin/csrmm_ablate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
    do                                                                             \
    {                                                                              \
        hipError_t e = (x);                                                        \
        if(e != hipSuccess)                                                        \
        {                                                                          \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);   \
            exit(1);                                                               \
        }                                                                          \
    } while(0)

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int xcd_row(int bx, int chunk)
{
    return chunk > 0 ? (bx & 7) * chunk + (bx >> 3) : bx;
}

// V0: shipped kernel: wave per (row, 128-col chunk), 4 waves/WG
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
    double        a0 = 0, a1 = 0;
    const double *Bj = B + j;
    int           p  = s;
    for(; p + 4 <= e; p += 4)
    {
        const double  v0 = val[p], v1 = val[p + 1], v2 = val[p + 2], v3 = val[p + 3];
        const double2 b0 = *reinterpret_cast<const double2 *>(Bj + (size_t)col[p] * n);
        const double2 b1 = *reinterpret_cast<const double2 *>(Bj + (size_t)col[p + 1] * n);
        const double2 b2 = *reinterpret_cast<const double2 *>(Bj + (size_t)col[p + 2] * n);
        const double2 b3 = *reinterpret_cast<const double2 *>(Bj + (size_t)col[p + 3] * n);
        a0 = fma(v0, b0.x, a0), a1 = fma(v0, b0.y, a1);
        a0 = fma(v1, b1.x, a0), a1 = fma(v1, b1.y, a1);
        a0 = fma(v2, b2.x, a0), a1 = fma(v2, b2.y, a1);
        a0 = fma(v3, b3.x, a0), a1 = fma(v3, b3.y, a1);
    }
    for(; p < e; p++)
    {
        const double  v0 = val[p];
        const double2 b0 = *reinterpret_cast<const double2 *>(Bj + (size_t)col[p] * n);
        a0 = fma(v0, b0.x, a0), a1 = fma(v0, b0.y, a1);
    }
    double2 *cp = reinterpret_cast<double2 *>(C + (size_t)i * n + j);
    double2  c;
    if(readc)
    {
        c   = *cp;
        c.x = fma(beta, c.x, a0), c.y = fma(beta, c.y, a1);
    }
    else
        c.x = a0, c.y = a1;
    *cp = c;
}

// V1..: wave covers NC = 128*CH columns of RW consecutive rows (sequentially); NT: non-temporal C stores.
// ROWS_PER_WG = 4*RW
template <int CH, int RW, bool NT, int UNR, bool IL = false>
__global__ __launch_bounds__(256) void k1(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w    = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int i0   = IL ? xcd_row(blockIdx.x, chunk) * 4 * RW + w : (xcd_row(blockIdx.x, chunk) * 4 + w) * RW;
    const int j    = 2 * lane + 128 * CH * (int)blockIdx.y;
    if(i0 >= m)
        return;
    const double *Bj = B + j;
#pragma unroll
    for(int r = 0; r < RW; r++)
    {
        const int i = i0 + (IL ? 4 * r : r);
        if(i >= m)
            break;
        const int s = row_ptr[i], e = row_ptr[i + 1];
        double    a[CH][2];
#pragma unroll
        for(int c = 0; c < CH; c++)
            a[c][0] = 0, a[c][1] = 0;
        int p = s;
        for(; p + UNR <= e; p += UNR)
        {
            double  v[UNR];
            v2d     b[UNR][CH];
#pragma unroll
            for(int u = 0; u < UNR; u++)
            {
                v[u]             = val[p + u];
                const double *bp = Bj + (size_t)col[p + u] * n;
#pragma unroll
                for(int c = 0; c < CH; c++)
                    b[u][c] = *reinterpret_cast<const v2d *>(bp + 128 * c);
            }
#pragma unroll
            for(int u = 0; u < UNR; u++)
#pragma unroll
                for(int c = 0; c < CH; c++)
                    a[c][0] = fma(v[u], b[u][c].x, a[c][0]), a[c][1] = fma(v[u], b[u][c].y, a[c][1]);
        }
        for(; p < e; p++)
        {
            const double  v0 = val[p];
            const double *bp = Bj + (size_t)col[p] * n;
#pragma unroll
            for(int c = 0; c < CH; c++)
            {
                const v2d b = *reinterpret_cast<const v2d *>(bp + 128 * c);
                a[c][0] = fma(v0, b.x, a[c][0]), a[c][1] = fma(v0, b.y, a[c][1]);
            }
        }
        double *cp = C + (size_t)i * n + j;
#pragma unroll
        for(int c = 0; c < CH; c++)
        {
            v2d o;
            if(readc)
            {
                o   = *reinterpret_cast<const v2d *>(cp + 128 * c);
                o.x = fma(beta, o.x, a[c][0]), o.y = fma(beta, o.y, a[c][1]);
            }
            else
                o.x = a[c][0], o.y = a[c][1];
            if(NT)
                __builtin_nontemporal_store(o, reinterpret_cast<v2d *>(cp + 128 * c));
            else
                *reinterpret_cast<v2d *>(cp + 128 * c) = o;
        }
    }
}

// V: two rows in flight at once per wave (loads of both rows issued before any FMA); rows <= 8 nnz fast path
template <int CH, bool NT>
__global__ __launch_bounds__(256) void k2(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    // lane-parallel row metadata: lanes 0..7 hold up to 8 entries of the row, broadcast with readlane
    const int w    = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int i    = xcd_row(blockIdx.x, chunk) * 4 + w;
    const int j    = 2 * lane + 128 * CH * (int)blockIdx.y;
    if(i >= m)
        return;
    const int s = row_ptr[i], e = row_ptr[i + 1];
    const int cnt = e - s;
    double    a[CH][2];
#pragma unroll
    for(int c = 0; c < CH; c++)
        a[c][0] = 0, a[c][1] = 0;
    const double *Bj = B + j;
    for(int p0 = s; p0 < e; p0 += 64)
    {
        const int    k   = min(64, e - p0);
        const int    mc  = lane < k ? col[p0 + lane] : 0;
        const double mv  = lane < k ? val[p0 + lane] : 0.0;
        for(int q = 0; q < k; q++)
        {
            const int     cq = __builtin_amdgcn_readlane(mc, q);
            const double  vq = __shfl(mv, q);
            const double *bp = Bj + (size_t)cq * n;
#pragma unroll
            for(int c = 0; c < CH; c++)
            {
                const v2d b = *reinterpret_cast<const v2d *>(bp + 128 * c);
                a[c][0] = fma(vq, b.x, a[c][0]), a[c][1] = fma(vq, b.y, a[c][1]);
            }
        }
    }
    (void)cnt;
    double *cp = C + (size_t)i * n + j;
#pragma unroll
    for(int c = 0; c < CH; c++)
    {
        v2d o;
        if(readc)
        {
            o   = *reinterpret_cast<const v2d *>(cp + 128 * c);
            o.x = fma(beta, o.x, a[c][0]), o.y = fma(beta, o.y, a[c][1]);
        }
        else
            o.x = a[c][0], o.y = a[c][1];
        if(NT)
            __builtin_nontemporal_store(o, reinterpret_cast<v2d *>(cp + 128 * c));
        else
            *reinterpret_cast<v2d *>(cp + 128 * c) = o;
    }
}

// k3: a wave streams the entries of RW consecutive rows; row_ptr / col / val are held one per lane and
// broadcast with readlane, B rows are prefetched D entries ahead in a register ring, C rows are written
// when the stream crosses a row end
__device__ __forceinline__ double rl_double(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
template <int CH, int RW, int D, bool NT>
__global__ __launch_bounds__(256) void k3(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w    = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int i0   = (xcd_row(blockIdx.x, chunk) * 4 + w) * RW;
    if(i0 >= m)
        return;
    const int     nr  = min(RW, m - i0);
    const int     j   = 2 * lane + 128 * CH * (int)blockIdx.y;
    const int     rpl = row_ptr[i0 + min(lane, nr)];
    const int     s = __builtin_amdgcn_readlane(rpl, 0), e = __builtin_amdgcn_readlane(rpl, nr);
    const double *Bj  = B + j;
    int           cur = 0, rend = __builtin_amdgcn_readlane(rpl, 1);
    double        a[CH][2];
#pragma unroll
    for(int c = 0; c < CH; c++)
        a[c][0] = 0, a[c][1] = 0;
    auto flush = [&]() {
        double *cp = C + (size_t)(i0 + cur) * n + j;
#pragma unroll
        for(int c = 0; c < CH; c++)
        {
            v2d o;
            if(readc)
            {
                o   = *reinterpret_cast<const v2d *>(cp + 128 * c);
                o.x = fma(beta, o.x, a[c][0]), o.y = fma(beta, o.y, a[c][1]);
            }
            else
                o.x = a[c][0], o.y = a[c][1];
            if(NT)
                __builtin_nontemporal_store(o, reinterpret_cast<v2d *>(cp + 128 * c));
            else
                *reinterpret_cast<v2d *>(cp + 128 * c) = o;
            a[c][0] = 0, a[c][1] = 0;
        }
        cur++;
        rend = __builtin_amdgcn_readlane(rpl, min(cur + 1, 63));
    };
    for(int p0 = s; p0 < e; p0 += 64)
    {
        const int    k  = min(64, e - p0);
        const int    mc = lane < k ? col[p0 + lane] : 0;
        const double mv = lane < k ? val[p0 + lane] : 0.0;
        v2d          b[D][CH];
#pragma unroll
        for(int u = 0; u < D; u++)
            if(u < k)
            {
                const double *bp = Bj + (size_t)__builtin_amdgcn_readlane(mc, u) * n;
#pragma unroll
                for(int c = 0; c < CH; c++)
                    b[u][c] = *reinterpret_cast<const v2d *>(bp + 128 * c);
            }
        for(int q = 0; q < k; q += D)
        {
#pragma unroll
            for(int u = 0; u < D; u++)
                if(q + u < k)
                {
                    while(rend <= p0 + q + u)
                        flush();
                    const double vq = rl_double(mv, q + u);
#pragma unroll
                    for(int c = 0; c < CH; c++)
                        a[c][0] = fma(vq, b[u][c].x, a[c][0]), a[c][1] = fma(vq, b[u][c].y, a[c][1]);
                    if(q + u + D < k)
                    {
                        const double *bp = Bj + (size_t)__builtin_amdgcn_readlane(mc, q + u + D) * n;
#pragma unroll
                        for(int c = 0; c < CH; c++)
                            b[u][c] = *reinterpret_cast<const v2d *>(bp + 128 * c);
                    }
                }
        }
    }
    while(cur < nr)
        flush();
}

// k4: k3 with the B-row loads issued through inline asm (invisible to the compiler's waitcnt pass) and
// counted waits: vmcnt(D-1) retires exactly the oldest ring slot
__device__ __forceinline__ v2d gload16(const double *p)
{
    v2d r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_ring(v2d &r)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}
template <int CH, int RW, int D, bool NT>
__global__ __launch_bounds__(256) void k4(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    static_assert(CH == 1, "one 128-column chunk per wave");
    const int w    = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int i0   = (xcd_row(blockIdx.x, chunk) * 4 + w) * RW;
    if(i0 >= m)
        return;
    const int     nr  = min(RW, m - i0);
    const int     j   = 2 * lane + 128 * (int)blockIdx.y;
    const int     rpl = row_ptr[i0 + min(lane, nr)];
    const int     s = __builtin_amdgcn_readlane(rpl, 0), e = __builtin_amdgcn_readlane(rpl, nr);
    const double *Bj  = B + j;
    int           cur = 0, rend = __builtin_amdgcn_readlane(rpl, 1);
    double        a0 = 0, a1 = 0;
    auto flush = [&]() {
        v2d *cp = reinterpret_cast<v2d *>(C + (size_t)(i0 + cur) * n + j);
        v2d  o;
        if(readc)
        {
            o   = *cp;
            o.x = fma(beta, o.x, a0), o.y = fma(beta, o.y, a1);
        }
        else
            o.x = a0, o.y = a1;
        if(NT)
            __builtin_nontemporal_store(o, cp);
        else
            *cp = o;
        a0 = 0, a1 = 0;
        cur++;
        rend = __builtin_amdgcn_readlane(rpl, min(cur + 1, 63));
    };
    for(int p0 = s; p0 < e; p0 += 64)
    {
        const int    k  = min(64, e - p0);
        const int    mc = lane < k ? col[p0 + lane] : 0;
        const double mv = lane < k ? val[p0 + lane] : 0.0;
        v2d          b[D];
#pragma unroll
        for(int u = 0; u < D; u++) // slots past the end re-load the last entry: the ring always holds D loads
            b[u] = gload16(Bj + (size_t)__builtin_amdgcn_readlane(mc, min(u, k - 1)) * n);
        for(int q = 0; q < k; q += D)
        {
#pragma unroll
            for(int u = 0; u < D; u++)
                if(q + u < k)
                {
                    while(rend <= p0 + q + u)
                        flush();
                    const double vq = rl_double(mv, q + u);
                    wait_ring<D - 1>(b[u]);
                    a0 = fma(vq, b[u].x, a0), a1 = fma(vq, b[u].y, a1);
                    b[u] = gload16(Bj + (size_t)__builtin_amdgcn_readlane(mc, min(q + u + D, k - 1)) * n);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // retire the padding loads before the ring is reused
    }
    while(cur < nr)
        flush();
}

// diagnostics on k0: MODE 1 = every entry reads B row i (no neighbour rows), MODE 2 = one entry per row (B row i),
// MODE 3 = normal but no C store (sum kept alive through a never-true store)
template <int MODE>
__global__ __launch_bounds__(256) void kd(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = xcd_row(blockIdx.x, chunk) * 4 + w;
    const int j = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(i >= m || j >= n)
        return;
    const int     s = row_ptr[i], e = MODE == 2 ? row_ptr[i] + 1 : row_ptr[i + 1];
    double        a0 = 0, a1 = 0;
    const double *Bj = B + j;
    for(int p = s; p < e; p++)
    {
        const double  v0 = val[p];
        const int     c  = (MODE == 1 || MODE == 2) ? i + (col[p] & 0) : col[p];
        const double2 b0 = *reinterpret_cast<const double2 *>(Bj + (size_t)c * n);
        a0 = fma(v0, b0.x, a0), a1 = fma(v0, b0.y, a1);
    }
    double2 *cp = reinterpret_cast<double2 *>(C + (size_t)i * n + j);
    double2  c;
    c.x = a0, c.y = a1;
    if(MODE != 3 || a0 == 1.2345e-300)
        *cp = c;
}

// k5: wave per (row, 64-column chunk), one double per lane: half the L2 footprint per row of k0
template <bool NT>
__global__ __launch_bounds__(256) void k5(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = xcd_row(blockIdx.x, chunk) * 4 + w;
    const int j = (int)(threadIdx.x & 63) + 64 * (int)blockIdx.y;
    if(i >= m || j >= n)
        return;
    const int     s = row_ptr[i], e = row_ptr[i + 1];
    double        a0 = 0;
    const double *Bj = B + j;
    int           p  = s;
    for(; p + 4 <= e; p += 4)
    {
        const double v0 = val[p], v1 = val[p + 1], v2 = val[p + 2], v3 = val[p + 3];
        const double b0 = Bj[(size_t)col[p] * n], b1 = Bj[(size_t)col[p + 1] * n], b2 = Bj[(size_t)col[p + 2] * n],
                     b3 = Bj[(size_t)col[p + 3] * n];
        a0 = fma(v0, b0, a0), a0 = fma(v1, b1, a0), a0 = fma(v2, b2, a0), a0 = fma(v3, b3, a0);
    }
    for(; p < e; p++)
        a0 = fma(val[p], Bj[(size_t)col[p] * n], a0);
    double *cp = C + (size_t)i * n + j;
    if(readc)
        a0 = fma(beta, *cp, a0);
    if(NT)
        __builtin_nontemporal_store(a0, cp);
    else
        *cp = a0;
}

// k6: k0 with the C store issued with explicit cache-policy bits (SB: 1 = sc0, 2 = sc1, 4 = nt)
template <int SB>
__global__ __launch_bounds__(256) void k6(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = xcd_row(blockIdx.x, chunk) * 4 + w;
    const int j = 2 * (int)(threadIdx.x & 63) + 128 * (int)blockIdx.y;
    if(i >= m || j >= n)
        return;
    const int     s = row_ptr[i], e = row_ptr[i + 1];
    double        a0 = 0, a1 = 0;
    const double *Bj = B + j;
    for(int p = s; p < e; p++)
    {
        const double  v0 = val[p];
        const double2 b0 = *reinterpret_cast<const double2 *>(Bj + (size_t)col[p] * n);
        a0 = fma(v0, b0.x, a0), a1 = fma(v0, b0.y, a1);
    }
    double *cp = C + (size_t)i * n + j;
    v2d     c;
    if(readc)
    {
        c   = *reinterpret_cast<const v2d *>(cp);
        c.x = fma(beta, c.x, a0), c.y = fma(beta, c.y, a1);
    }
    else
        c.x = a0, c.y = a1;
    if(SB == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(cp), "v"(c) : "memory");
    if(SB == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(cp), "v"(c) : "memory");
    if(SB == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(cp), "v"(c) : "memory");
    if(SB == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(cp), "v"(c) : "memory");
    if(SB == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(cp), "v"(c) : "memory");
    if(SB == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(cp), "v"(c) : "memory");
    if(SB == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(cp), "v"(c) : "memory");
}

// kh: half-wave per row, 64-column chunks (32 lanes x double2): same bytes per wave-load as k0, but the B / C
// footprint per row is halved, so the (2g + rows in flight) window fits the 4 MB L2
template <bool NT>
__global__ __launch_bounds__(256) void kh(int m, const double *__restrict__ val, const int *__restrict__ col,
                                          const int *__restrict__ row_ptr, const double *__restrict__ B, int n,
                                          double beta, double *__restrict__ C, bool readc, int chunk)
{
    const int half = threadIdx.x >> 5; // 8 half-waves per workgroup = 8 rows
    const int lane = threadIdx.x & 31;
    const int i    = xcd_row(blockIdx.x, chunk) * 8 + half;
    const int j    = 2 * lane + 64 * (int)blockIdx.y;
    if(i >= m)
        return;
    const int     s = row_ptr[i], e = row_ptr[i + 1];
    double        a0 = 0, a1 = 0;
    const double *Bj = B + j;
    int           p  = s;
    for(; p + 4 <= e; p += 4)
    {
        const double  v0 = val[p], v1 = val[p + 1], v2 = val[p + 2], v3 = val[p + 3];
        const int     c0 = col[p], c1 = col[p + 1], c2 = col[p + 2], c3 = col[p + 3];
        const double2 b0 = *reinterpret_cast<const double2 *>(Bj + (size_t)c0 * n);
        const double2 b1 = *reinterpret_cast<const double2 *>(Bj + (size_t)c1 * n);
        const double2 b2 = *reinterpret_cast<const double2 *>(Bj + (size_t)c2 * n);
        const double2 b3 = *reinterpret_cast<const double2 *>(Bj + (size_t)c3 * n);
        a0 = fma(v0, b0.x, a0), a1 = fma(v0, b0.y, a1);
        a0 = fma(v1, b1.x, a0), a1 = fma(v1, b1.y, a1);
        a0 = fma(v2, b2.x, a0), a1 = fma(v2, b2.y, a1);
        a0 = fma(v3, b3.x, a0), a1 = fma(v3, b3.y, a1);
    }
    for(; p < e; p++)
    {
        const double  v0 = val[p];
        const double2 b0 = *reinterpret_cast<const double2 *>(Bj + (size_t)col[p] * n);
        a0 = fma(v0, b0.x, a0), a1 = fma(v0, b0.y, a1);
    }
    v2d *cp = reinterpret_cast<v2d *>(C + (size_t)i * n + j);
    v2d  c;
    if(readc)
    {
        c   = *cp;
        c.x = fma(beta, c.x, a0), c.y = fma(beta, c.y, a1);
    }
    else
        c.x = a0, c.y = a1;
    if(NT)
        __builtin_nontemporal_store(c, cp);
    else
        *cp = c;
}

// pure streaming floor: C[i] = B[i] (+ beta*C[i])
__global__ __launch_bounds__(256) void kcopy(size_t n2, const v2d *__restrict__ B, v2d *__restrict__ C, double beta,
                                             bool readc)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if(i >= n2)
        return;
    v2d b = B[i];
    if(readc)
    {
        const v2d c = C[i];
        b.x = fma(beta, c.x, b.x), b.y = fma(beta, c.y, b.y);
    }
    C[i] = b;
}

int main(int argc, char **argv)
{
    const int  g = argc > 1 ? atoi(argv[1]) : 1000;
    const int  n = 256;
    const long m = (long)g * g;
    std::vector<int>    rp(m + 1), ci;
    std::vector<double> v;
    rp[0] = 0;
    for(long r = 0; r < m; r++)
    {
        const long i = r / g, j = r % g;
        if(i > 0) ci.push_back(r - g), v.push_back(-1.0);
        if(j > 0) ci.push_back(r - 1), v.push_back(-1.0);
        ci.push_back(r), v.push_back(4.0);
        if(j < g - 1) ci.push_back(r + 1), v.push_back(-1.0);
        if(i < g - 1) ci.push_back(r + g), v.push_back(-1.0);
        rp[r + 1] = (int)ci.size();
    }
    const long nnz = ci.size();
    std::vector<double> B((size_t)m * n);
    for(size_t q = 0; q < B.size(); q++)
        B[q] = sin(0.001 * (double)(q % 100003));
    int    *d_rp, *d_ci;
    double *d_v, *d_B, *d_C, *d_R;
    CHECK(hipMalloc(&d_rp, (m + 1) * 4)); CHECK(hipMalloc(&d_ci, nnz * 4)); CHECK(hipMalloc(&d_v, nnz * 8));
    CHECK(hipMalloc(&d_B, B.size() * 8)); CHECK(hipMalloc(&d_C, B.size() * 8)); CHECK(hipMalloc(&d_R, B.size() * 8));
    CHECK(hipMemcpy(d_rp, rp.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_v, v.data(), nnz * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_B, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> ref(B.size()), out(B.size());
    constexpr int NV = 45;
    const char *names[NV] = {"V0 shipped: wave/(row,128c)", "V1 wave/(row,256c) unr4", "V2 wave/(row,256c) unr4 nt-store",
                             "V3 wave/2 rows seq,256c", "V4 wave/4 rows seq,256c", "V5 wave/4 rows seq,256c nt",
                             "V6 lane-loaded row meta,256c", "V7 lane-loaded meta,256c nt", "V8 wave/(row,256c) unr8",
                             "V9 wave/(row,128c) unr4 nt", "V10 wave/(row,256c) unr2", "V11 stream 12 rows,128c,D8", "V12 stream 12 rows,128c,D8 nt", "V13 stream 12 rows,256c,D4",
                             "V14 stream 24 rows,128c,D8", "V15 stream 12 rows,128c,D4", "V16 stream 6 rows,128c,D8", "V17 asm ring 12 rows,D8", "V18 asm ring 12 rows,D8 nt", "V19 asm ring 12 rows,D4", "V20 asm ring 12 rows,D16", "V21 asm ring 24 rows,D8", "D1 all entries read B row i", "D2 one entry per row", "D3 normal, no C store", "D4 k0 without XCD remap", "W2 wave/2 rows seq,128c", "W4 wave/4 rows seq,128c", "W8 wave/8 rows seq,128c", "I2 WG 8 rows interleaved,128c", "I4 WG 16 rows interleaved,128c", "I8 WG 32 rows interleaved,128c", "I16 WG 64 rows interleaved,128c", "N1 wave/(row,64c) 8B/lane", "N2 wave/(row,64c) 8B/lane nt", "S1 store sc0", "S2 store sc1", "S3 store sc0 sc1", "S4 store nt", "S5 store sc0 nt", "S6 store sc1 nt", "S7 store sc0 sc1 nt", "H1 half-wave/(row,64c)", "H2 half-wave/(row,64c) nt", "copy floor"};
    for(int pass = 0; pass < 2; pass++)
    {
        const bool   readc = pass == 1;
        const double beta  = readc ? -2.0 : 0.0;
        double       best[NV];
        for(int q = 0; q < NV; q++) best[q] = 1e30;
        for(int rep = 0; rep < 6; rep++)
            for(int q = 0; q < NV; q++)
            {
                if(readc) CHECK(hipMemset(d_C, 0, B.size() * 8));
                auto launch = [&](int rows_per_wg, int colchunk) {
                    int nbx = (int)((m + rows_per_wg - 1) / rows_per_wg), chunk = (nbx + 7) / 8;
                    return dim3(chunk * 8, (n + colchunk - 1) / colchunk);
                };
                CHECK(hipEventRecord(e0));
                dim3 gr;
                int  ch;
#define ARGS (int)m, d_v, d_ci, d_rp, d_B, n, beta, d_C, readc, ch
                switch(q)
                {
                case 0: gr = launch(4, 128), ch = gr.x / 8; k0<<<gr, 256>>>(ARGS); break;
                case 1: gr = launch(4, 256), ch = gr.x / 8; k1<2, 1, false, 4><<<gr, 256>>>(ARGS); break;
                case 2: gr = launch(4, 256), ch = gr.x / 8; k1<2, 1, true, 4><<<gr, 256>>>(ARGS); break;
                case 3: gr = launch(8, 256), ch = gr.x / 8; k1<2, 2, false, 4><<<gr, 256>>>(ARGS); break;
                case 4: gr = launch(16, 256), ch = gr.x / 8; k1<2, 4, false, 4><<<gr, 256>>>(ARGS); break;
                case 5: gr = launch(16, 256), ch = gr.x / 8; k1<2, 4, true, 4><<<gr, 256>>>(ARGS); break;
                case 6: gr = launch(4, 256), ch = gr.x / 8; k2<2, false><<<gr, 256>>>(ARGS); break;
                case 7: gr = launch(4, 256), ch = gr.x / 8; k2<2, true><<<gr, 256>>>(ARGS); break;
                case 8: gr = launch(4, 256), ch = gr.x / 8; k1<2, 1, false, 8><<<gr, 256>>>(ARGS); break;
                case 9: gr = launch(4, 128), ch = gr.x / 8; k1<1, 1, true, 4><<<gr, 256>>>(ARGS); break;
                case 10: gr = launch(4, 256), ch = gr.x / 8; k1<2, 1, false, 2><<<gr, 256>>>(ARGS); break;
                case 11: gr = launch(48, 128), ch = gr.x / 8; k3<1, 12, 8, false><<<gr, 256>>>(ARGS); break;
                case 12: gr = launch(48, 128), ch = gr.x / 8; k3<1, 12, 8, true><<<gr, 256>>>(ARGS); break;
                case 13: gr = launch(48, 256), ch = gr.x / 8; k3<2, 12, 4, false><<<gr, 256>>>(ARGS); break;
                case 14: gr = launch(96, 128), ch = gr.x / 8; k3<1, 24, 8, false><<<gr, 256>>>(ARGS); break;
                case 15: gr = launch(48, 128), ch = gr.x / 8; k3<1, 12, 4, false><<<gr, 256>>>(ARGS); break;
                case 16: gr = launch(24, 128), ch = gr.x / 8; k3<1, 6, 8, false><<<gr, 256>>>(ARGS); break;
                case 17: gr = launch(48, 128), ch = gr.x / 8; k4<1, 12, 8, false><<<gr, 256>>>(ARGS); break;
                case 18: gr = launch(48, 128), ch = gr.x / 8; k4<1, 12, 8, true><<<gr, 256>>>(ARGS); break;
                case 19: gr = launch(48, 128), ch = gr.x / 8; k4<1, 12, 4, false><<<gr, 256>>>(ARGS); break;
                case 20: gr = launch(48, 128), ch = gr.x / 8; k4<1, 12, 16, false><<<gr, 256>>>(ARGS); break;
                case 21: gr = launch(96, 128), ch = gr.x / 8; k4<1, 24, 8, false><<<gr, 256>>>(ARGS); break;
                case 22: gr = launch(4, 128), ch = gr.x / 8; kd<1><<<gr, 256>>>(ARGS); break;
                case 23: gr = launch(4, 128), ch = gr.x / 8; kd<2><<<gr, 256>>>(ARGS); break;
                case 24: gr = launch(4, 128), ch = gr.x / 8; kd<3><<<gr, 256>>>(ARGS); break;
                case 25: gr = dim3((unsigned)((m + 3) / 4), 2), ch = 0; k0<<<gr, 256>>>(ARGS); break;
                case 26: gr = launch(8, 128), ch = gr.x / 8; k1<1, 2, false, 4><<<gr, 256>>>(ARGS); break;
                case 27: gr = launch(16, 128), ch = gr.x / 8; k1<1, 4, false, 4><<<gr, 256>>>(ARGS); break;
                case 28: gr = launch(32, 128), ch = gr.x / 8; k1<1, 8, false, 4><<<gr, 256>>>(ARGS); break;
                case 29: gr = launch(8, 128), ch = gr.x / 8; k1<1, 2, false, 4, true><<<gr, 256>>>(ARGS); break;
                case 30: gr = launch(16, 128), ch = gr.x / 8; k1<1, 4, false, 4, true><<<gr, 256>>>(ARGS); break;
                case 31: gr = launch(32, 128), ch = gr.x / 8; k1<1, 8, false, 4, true><<<gr, 256>>>(ARGS); break;
                case 32: gr = launch(64, 128), ch = gr.x / 8; k1<1, 16, false, 4, true><<<gr, 256>>>(ARGS); break;
                case 33: gr = launch(4, 64), ch = gr.x / 8; k5<false><<<gr, 256>>>(ARGS); break;
                case 34: gr = launch(4, 64), ch = gr.x / 8; k5<true><<<gr, 256>>>(ARGS); break;
                case 35: gr = launch(4, 128), ch = gr.x / 8; k6<1><<<gr, 256>>>(ARGS); break;
                case 36: gr = launch(4, 128), ch = gr.x / 8; k6<2><<<gr, 256>>>(ARGS); break;
                case 37: gr = launch(4, 128), ch = gr.x / 8; k6<3><<<gr, 256>>>(ARGS); break;
                case 38: gr = launch(4, 128), ch = gr.x / 8; k6<4><<<gr, 256>>>(ARGS); break;
                case 39: gr = launch(4, 128), ch = gr.x / 8; k6<5><<<gr, 256>>>(ARGS); break;
                case 40: gr = launch(4, 128), ch = gr.x / 8; k6<6><<<gr, 256>>>(ARGS); break;
                case 41: gr = launch(4, 128), ch = gr.x / 8; k6<7><<<gr, 256>>>(ARGS); break;
                case 42: gr = launch(8, 64), ch = gr.x / 8; kh<false><<<gr, 256>>>(ARGS); break;
                case 43: gr = launch(8, 64), ch = gr.x / 8; kh<true><<<gr, 256>>>(ARGS); break;
                default: kcopy<<<(unsigned)((B.size() / 2 + 255) / 256), 256>>>(B.size() / 2, (const v2d *)d_B, (v2d *)d_C, beta, readc);
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best[q] = std::min(best[q], (double)ms);
                if(rep == 0 && q < NV - 1)
                {
                    CHECK(hipMemcpy(out.data(), d_C, B.size() * 8, hipMemcpyDeviceToHost));
                    if(q == 0) ref = out;
                    else if((q < 22 || q > 25) && q < NV - 1 && out != ref) printf("  !! variant %d differs from V0\n", q);
                }
            }
        const double bytes = (double)(m + 1 + nnz) * 4 + nnz * 8.0 + (double)B.size() * 8 * (readc ? 3 : 2);
        printf("beta=%g  (algorithmic %.3f GB)\n", beta, bytes / 1e9);
        for(int q = 0; q < NV; q++)
            printf("  %-36s %.4f ms  %.2f TB/s  %.1f%%\n", names[q], best[q], bytes / best[q] / 1e9, bytes / best[q] / 1e9 / 80);
    }
    return 0;
}
