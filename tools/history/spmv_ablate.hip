// spmv_ablate.hip -- diagnostic build (never shipped): where does the CSR-Adaptive kernel spend its time?
// Variants of the order-0 / TILE 1024 kernel with one ingredient removed each, timed interleaved in one
// process on the 5-pt Laplacian (cdna_hip_programming.md section 5.4 rules 17, 23, 24).  Outputs of the
// ablated variants are wrong by construction; only the full variant is checked.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/spmv_ablate.hip -o /tmp/spmv_ablate && /tmp/spmv_ablate 4096
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

constexpr int TILE = 1024, BLOCK = 256, MAXROWS = 512;
typedef int    v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// V: 0 full, 1 no gather, 2 no LDS val, 3 no phase 2, 4 stream only, 5 full with product in LDS (8 B/nnz),
//    6 full with non-temporal val/col loads, 7 = 6 + non-temporal y stores and row_ptr loads
template <int V, bool L8 = false>
__global__ __launch_bounds__(BLOCK) void k(const int2 *__restrict__ blocks, const int *__restrict__ row_ptr,
                                           const int *__restrict__ col, const double *__restrict__ val,
                                           const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double s_val[TILE + 4];
    __shared__ double s_x[TILE + 4];
    __shared__ int    s_row[MAXROWS + 1];
    const int         tid = threadIdx.x;
    const int         b   = blockIdx.x;
    const int2        e0 = blocks[b], e1 = blocks[b + 1];
    const int         r0 = e0.x, p0 = e0.y, nrows = e1.x - r0, cnt = e1.y - p0;
    const int         w0 = p0 & ~3, cntw = cnt + (p0 - w0);
    for(int i = tid; i <= nrows; i += BLOCK)
        s_row[i] = (V == 7 ? __builtin_nontemporal_load(row_ptr + r0 + i) : row_ptr[r0 + i]) - w0;
    double sink = 0;
    const int i = 4 * tid;
    if(i + 3 < cntw)
    {
        int4    c;
        double2 va, vb;
        if(V == 6 || V == 7)
        {
            const v4i cc = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(col + w0 + i));
            const v2d a0 = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(val + w0 + i));
            const v2d a1 = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(val + w0 + i + 2));
            c.x = cc.x, c.y = cc.y, c.z = cc.z, c.w = cc.w;
            va.x = a0.x, va.y = a0.y, vb.x = a1.x, vb.y = a1.y;
        }
        else
        {
            c  = *reinterpret_cast<const int4 *>(col + w0 + i);
            va = *reinterpret_cast<const double2 *>(val + w0 + i);
            vb = *reinterpret_cast<const double2 *>(val + w0 + i + 2);
        }
        double        x0, x1, x2, x3;
        if(V == 1 || V == 4)
        {
            x0 = c.x * 1e-9, x1 = c.y * 1e-9, x2 = c.z * 1e-9, x3 = c.w * 1e-9;
        }
        else
        {
            x0 = x[c.x], x1 = x[c.y], x2 = x[c.z], x3 = x[c.w];
        }
        if(V == 4)
            sink = va.x * x0 + va.y * x1 + vb.x * x2 + vb.y * x3;
        else if(V == 5)
        {
            s_x[i] = va.x * x0, s_x[i + 1] = va.y * x1, s_x[i + 2] = vb.x * x2, s_x[i + 3] = vb.y * x3;
        }
        else
        {
            if(V != 2)
            {
                s_val[i] = va.x, s_val[i + 1] = va.y, s_val[i + 2] = vb.x, s_val[i + 3] = vb.y;
            }
            else
                sink = va.x + va.y + vb.x + vb.y;
            s_x[i] = x0, s_x[i + 1] = x1, s_x[i + 2] = x2, s_x[i + 3] = x3;
        }
    }
    else
        for(int q = i; q < cntw && q < i + 4; q++)
        {
            s_val[q] = val[w0 + q];
            s_x[q]   = V == 5 ? val[w0 + q] * x[col[w0 + q]] : x[col[w0 + q]];
        }
    if(tid < 3 && TILE + tid < cntw)
    {
        s_val[TILE + tid] = val[w0 + TILE + tid];
        s_x[TILE + tid]   = V == 5 ? val[w0 + TILE + tid] * x[col[w0 + TILE + tid]] : x[col[w0 + TILE + tid]];
    }
    if(V == 4)
    {
        if(tid < nrows)
            y[r0 + tid] = sink;
        if(tid + BLOCK < nrows)
            y[r0 + tid + BLOCK] = sink;
        return;
    }
    __syncthreads();
    if(V == 3)
    {
        for(int rr = tid; rr < nrows; rr += BLOCK)
            y[r0 + rr] = s_x[rr] + sink;
        return;
    }
    if(L8)
    {
        const int grp = tid >> 3, lane = tid & 7;
        for(int rr = grp; rr < nrows; rr += BLOCK / 8)
        {
            const int s = s_row[rr], e = s_row[rr + 1];
            const int nfull = (e - s) & ~7;
            double    acc = 0;
            for(int j = s + lane; j < s + nfull; j += 8)
                acc = fma(s_val[j], s_x[j], acc);
            double vq = acc + __shfl_down(acc, 4, 8);
            double t = vq + __shfl_down(vq, 1, 8);
            double res = t + __shfl_down(t, 2, 8);
            if(lane == 0)
            {
                if(nfull == 0) res = 0;
                for(int j = s + nfull; j < e; j++)
                    res = fma(s_val[j], s_x[j], res);
                y[r0 + rr] = res;
            }
        }
        return;
    }
    for(int rr = tid; rr < nrows; rr += BLOCK)
    {
        const int s = s_row[rr], e = s_row[rr + 1];
        double    acc = sink;
        if(V == 2 || V == 5)
            for(int j = s; j < e; j++)
                acc += s_x[j];
        else
            for(int j = s; j < e; j++)
                acc = fma(s_val[j], s_x[j], acc);
        if(V == 7 || V == 8)
            __builtin_nontemporal_store(acc, y + r0 + rr);
        else
            y[r0 + rr] = acc;
    }
}

// V9: cols go through LDS so that the 64 lanes of one gather instruction cover 64 CONSECUTIVE non-zeros
template <int NT>
__global__ __launch_bounds__(BLOCK) void k2(const int2 *__restrict__ blocks, const int *__restrict__ row_ptr,
                                            const int *__restrict__ col, const double *__restrict__ val,
                                            const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double s_val[TILE + 4];
    __shared__ double s_x[TILE + 4];
    __shared__ int    s_col[TILE + 4];
    __shared__ int    s_row[MAXROWS + 1];
    const int         tid = threadIdx.x;
    const int         b   = blockIdx.x;
    const int2        e0 = blocks[b], e1 = blocks[b + 1];
    const int         r0 = e0.x, p0 = e0.y, nrows = e1.x - r0, cnt = e1.y - p0;
    const int         w0 = p0 & ~3, cntw = cnt + (p0 - w0);
    for(int i = tid; i <= nrows; i += BLOCK)
        s_row[i] = row_ptr[r0 + i] - w0;
    const int i = 4 * tid;
    if(i + 3 < cntw)
    {
        const int4    c  = *reinterpret_cast<const int4 *>(col + w0 + i);
        const double2 va = *reinterpret_cast<const double2 *>(val + w0 + i);
        const double2 vb = *reinterpret_cast<const double2 *>(val + w0 + i + 2);
        *reinterpret_cast<int4 *>(&s_col[i]) = c;
        s_val[i] = va.x, s_val[i + 1] = va.y, s_val[i + 2] = vb.x, s_val[i + 3] = vb.y;
    }
    else
        for(int q = i; q < cntw && q < i + 4; q++)
        {
            s_val[q] = val[w0 + q];
            s_col[q] = col[w0 + q];
        }
    if(tid < 3 && TILE + tid < cntw)
    {
        s_val[TILE + tid] = val[w0 + TILE + tid];
        s_col[TILE + tid] = col[w0 + TILE + tid];
    }
    __syncthreads();
#pragma unroll
    for(int kk = 0; kk < TILE / BLOCK; kk++)
    {
        const int q = tid + kk * BLOCK;
        if(q < cntw)
            s_x[q] = x[s_col[q]];
    }
    if(tid < 3 && TILE + tid < cntw)
        s_x[TILE + tid] = x[s_col[TILE + tid]];
    __syncthreads();
    for(int rr = tid; rr < nrows; rr += BLOCK)
    {
        const int s = s_row[rr], e = s_row[rr + 1];
        double    acc = 0;
        for(int j = s; j < e; j++)
            acc = fma(s_val[j], s_x[j], acc);
        if(NT)
            __builtin_nontemporal_store(acc, y + r0 + rr);
        else
            y[r0 + rr] = acc;
    }
}

// V11: persistent workgroups, grid-stride over row blocks (no per-block workgroup launch / drain)
__global__ __launch_bounds__(BLOCK) void k3(const int2 *__restrict__ blocks, int nblocks, const int *__restrict__ row_ptr,
                                            const int *__restrict__ col, const double *__restrict__ val,
                                            const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double s_val[TILE + 4];
    __shared__ double s_x[TILE + 4];
    __shared__ int    s_row[MAXROWS + 1];
    const int         tid = threadIdx.x;
    for(int b = blockIdx.x; b < nblocks; b += gridDim.x)
    {
        const int2 e0 = blocks[b], e1 = blocks[b + 1];
        const int  r0 = e0.x, p0 = e0.y, nrows = e1.x - r0, cnt = e1.y - p0;
        const int  w0 = p0 & ~3, cntw = cnt + (p0 - w0);
        for(int i = tid; i <= nrows; i += BLOCK)
            s_row[i] = row_ptr[r0 + i] - w0;
        const int i = 4 * tid;
        if(i + 3 < cntw)
        {
            const int4    c  = *reinterpret_cast<const int4 *>(col + w0 + i);
            const double2 va = *reinterpret_cast<const double2 *>(val + w0 + i);
            const double2 vb = *reinterpret_cast<const double2 *>(val + w0 + i + 2);
            s_val[i] = va.x, s_val[i + 1] = va.y, s_val[i + 2] = vb.x, s_val[i + 3] = vb.y;
            s_x[i] = x[c.x], s_x[i + 1] = x[c.y], s_x[i + 2] = x[c.z], s_x[i + 3] = x[c.w];
        }
        else
            for(int q = i; q < cntw && q < i + 4; q++)
            {
                s_val[q] = val[w0 + q];
                s_x[q]   = x[col[w0 + q]];
            }
        if(tid < 3 && TILE + tid < cntw)
        {
            s_val[TILE + tid] = val[w0 + TILE + tid];
            s_x[TILE + tid]   = x[col[w0 + TILE + tid]];
        }
        __syncthreads();
        for(int rr = tid; rr < nrows; rr += BLOCK)
        {
            const int s = s_row[rr], e = s_row[rr + 1];
            double    acc = 0;
            for(int j = s; j < e; j++)
                acc = fma(s_val[j], s_x[j], acc);
            __builtin_nontemporal_store(acc, y + r0 + rr);
        }
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const int g = argc > 1 ? atoi(argv[1]) : 4096;
    const int K = argc > 2 ? atoi(argv[2]) : 5; // K > 5: uniform rows of K entries in groups of 5 columns
    const long long m = K > 5 ? (long long)g * 400 : (long long)g * g;
    std::vector<int> rp(m + 1), ci;
    std::vector<double> v;
    ci.reserve((size_t)K * m), v.reserve((size_t)K * m);
    rp[0] = 0;
    if(K > 5)
    {
        const int ngrp = K / 5;
        for(long long r = 0; r < m; r++)
        {
            for(int gq = 0; gq < ngrp; gq++)
            {
                const long long basec = r - r % 5 + (long long)(gq - ngrp / 2) * 3005;
                for(int q = 0; q < 5; q++)
                {
                    const long long c = basec + q;
                    if(c >= 0 && c < m) ci.push_back((int)c), v.push_back(0.37 + 1e-3 * q);
                }
            }
            rp[r + 1] = (int)ci.size();
        }
    }
    else
    for(long long r = 0; r < m; r++)
    {
        const long long i = r / g, j = r % g;
        if(i > 0) ci.push_back(r - g), v.push_back(-1.0);
        if(j > 0) ci.push_back(r - 1), v.push_back(-1.0);
        ci.push_back(r), v.push_back(4.0);
        if(j < g - 1) ci.push_back(r + 1), v.push_back(-1.0);
        if(i < g - 1) ci.push_back(r + g), v.push_back(-1.0);
        rp[r + 1] = (int)ci.size();
    }
    const long long nnz = ci.size();
    std::vector<int> blk;
    for(long long i = 0; i < m;)
    {
        long long j = i;
        while(j < m && j - i < MAXROWS && rp[j + 1] - rp[i] <= TILE) j++;
        blk.push_back((int)i), blk.push_back(rp[i]);
        i = j;
    }
    const int nb = (int)blk.size() / 2;
    blk.push_back((int)m), blk.push_back((int)nnz);
    std::vector<double> x(m), yref(m);
    for(long long r = 0; r < m; r++) x[r] = sin(0.01 * r);
    for(long long r = 0; r < m; r++)
    {
        double a = 0;
        if(K > 10)
        {
            double l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int n = rp[r + 1] - rp[r], nf = n & ~7;
            for(int p = 0; p < nf; p++) l[p & 7] = fma(v[rp[r] + p], x[ci[rp[r] + p]], l[p & 7]);
            if(nf) a = ((l[0] + l[4]) + (l[1] + l[5])) + ((l[2] + l[6]) + (l[3] + l[7]));
            for(int p = rp[r] + nf; p < rp[r + 1]; p++) a = fma(v[p], x[ci[p]], a);
        }
        else
            for(int p = rp[r]; p < rp[r + 1]; p++) a = fma(v[p], x[ci[p]], a);
        yref[r] = a;
    }
    int *d_rp, *d_ci, *d_blk;
    double *d_v, *d_x, *d_y;
    CHECK(hipMalloc(&d_rp, (m + 1) * 4)); CHECK(hipMalloc(&d_ci, nnz * 4)); CHECK(hipMalloc(&d_blk, blk.size() * 4));
    CHECK(hipMalloc(&d_v, nnz * 8)); CHECK(hipMalloc(&d_x, m * 8)); CHECK(hipMalloc(&d_y, m * 8));
    CHECK(hipMemcpy(d_rp, rp.data(), (m + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_blk, blk.data(), blk.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_v, v.data(), nnz * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_x, x.data(), m * 8, hipMemcpyHostToDevice));
    const double abytes = (m + 1 + nnz) * 4.0 + (2 * m + nnz) * 8.0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[13] = {"full (val+x in LDS, fma chain)", "no x gather", "no val in LDS", "no phase 2",
                            "stream only (no LDS, no gather)", "product in LDS (not bit-exact)", "full + nt val/col loads",
                            "full + nt loads + nt y store", "full + nt y store only", "cols via LDS, compact gathers", "compact gathers + nt y store", "persistent 2048 WGs + nt store", "persistent 4096 WGs + nt store"};
    double best[13];
    for(int q = 0; q < 13; q++) best[q] = 1e30;
    auto run = [&](int variant) {
        const int2 *B = (const int2 *)d_blk;
        switch(variant)
        {
        case 0: if(K > 10) hipLaunchKernelGGL((k<0, true>), dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y);
                else hipLaunchKernelGGL(k<0>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 9: hipLaunchKernelGGL(k2<0>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 10: hipLaunchKernelGGL(k2<1>, dim3(nb), dim3(BLOCK), 0, 0, B, d_rp, d_ci, d_v, d_x, d_y); break;
        case 11: hipLaunchKernelGGL(k3, dim3(2048), dim3(BLOCK), 0, 0, B, nb, d_rp, d_ci, d_v, d_x, d_y); break;
        case 12: hipLaunchKernelGGL(k3, dim3(4096), dim3(BLOCK), 0, 0, B, nb, d_rp, d_ci, d_v, d_x, d_y); break;
        }
    };
    for(int round = 0; round < 5; round++)
        for(int q = 0; q < 13; q++)
        {
            for(int w = 0; w < 3; w++) run(q);
            CHECK(hipEventRecord(e0, 0));
            for(int it = 0; it < 20; it++) run(q);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best[q] = std::min(best[q], (double)ms / 20);
        }
    run(0);
    std::vector<double> y(m);
    CHECK(hipMemcpy(y.data(), d_y, m * 8, hipMemcpyDeviceToHost));
    long long bad = 0;
    for(long long r = 0; r < m; r++) bad += y[r] != yref[r];
    printf("grid %d K %d: m=%lld nnz=%lld blocks=%d, full variant mismatches vs host fma chain: %lld\n", g, K, m, nnz, nb, bad);
    for(int q = 0; q < 13; q++)
        printf("V%d %-34s %.4f ms  %.0f GB/s (algorithmic bytes)\n", q, names[q], best[q], abytes / best[q] / 1e6);
    return 0;
}
