// exp_gather.hip -- diagnostic build (never shipped): how fast can this part serve the x[col] gathers of a short irregular SpMV?
// The web-like stand-in (3.1 M entries over 1 M columns, 8 MB of x) spends a workgroup's life waiting for its gathers (round 5
// trace: 7.3 us median from block table to "tile in LDS" of a 27 us kernel).  Each variant streams an index array with coalesced
// 16-byte loads (four indices per lane, 128-lane workgroups of 512 entries: the CSR-Adaptive tile), gathers x[idx] with one load
// flavour, and stores one sum per lane.
//   hipcc -O3 --offload-arch=gfx950 tools/exp_gather.hip -o tools/bin/exp_gather
//   exp_gather [n_columns=1000005] [entries=3100840] [zipf_permille=500]
// Prints one JSON line per variant: us per launch (back to back, 200 launches), gathers per ns.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
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

// load flavours: 0 plain, 1 non-temporal builtin, 2 relaxed agent-scope atomic load (sc1: served by L2, bypasses L1),
// 3 asm "sc0 sc1", 4 asm "nt", 5 asm "sc1 nt"
template <int F>
__device__ __forceinline__ double gload(const double *p)
{
    if constexpr(F == 0)
        return *p;
    else if constexpr(F == 1)
        return __builtin_nontemporal_load(p);
    else if constexpr(F == 2)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
    {
        double v;
        if constexpr(F == 3)
            asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
        else if constexpr(F == 4)
            asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
        else
            asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
}

template <int F, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gather_kernel(const int *__restrict__ idx, const double *__restrict__ x, int entries,
                                                       double *__restrict__ out)
{
    const int i = 4 * (blockIdx.x * BLOCK + threadIdx.x);
    if(i + 3 >= entries)
        return;
    const int4 c = *reinterpret_cast<const int4 *>(idx + i);
    double     a = gload<F>(x + c.x), b = gload<F>(x + c.y), d = gload<F>(x + c.z), e = gload<F>(x + c.w);
    if constexpr(F >= 3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * BLOCK + threadIdx.x] = (a + b) + (d + e);
}

// the same gathers as FLOATS (4-byte elements): is the cost per request or per byte?
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void gather_f32_kernel(const int *__restrict__ idx, const float *__restrict__ x, int entries,
                                                           float *__restrict__ out)
{
    const int i = 4 * (blockIdx.x * BLOCK + threadIdx.x);
    if(i + 3 >= entries)
        return;
    const int4 c                          = *reinterpret_cast<const int4 *>(idx + i);
    out[blockIdx.x * BLOCK + threadIdx.x] = (x[c.x] + x[c.y]) + (x[c.z] + x[c.w]);
}

// no gather at all: the index stream + the store (what the rest of the kernel costs)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void stream_kernel(const int *__restrict__ idx, int entries, double *__restrict__ out)
{
    const int i = 4 * (blockIdx.x * BLOCK + threadIdx.x);
    if(i + 3 >= entries)
        return;
    const int4 c                          = *reinterpret_cast<const int4 *>(idx + i);
    out[blockIdx.x * BLOCK + threadIdx.x] = (double)(c.x + c.y + c.z + c.w);
}

// the full stream of an SpMV tile (columns + values in, one result out) around the plain gathers, with the STREAMS marked
// non-temporal or not: do the once-read col / val lines push x out of the XCD's 4 MiB L2?
// S: 0 plain streams, 1 nt loads of col / val, 2 nt loads + nt store
template <int S, int BLOCK>
__global__ __launch_bounds__(BLOCK) void spmv_like_kernel(const int *__restrict__ idx, const double *__restrict__ val,
                                                          const double *__restrict__ x, int entries, double *__restrict__ out)
{
    const int i = 4 * (blockIdx.x * BLOCK + threadIdx.x);
    if(i + 3 >= entries)
        return;
    int4    c;
    double2 va, vb;
    if constexpr(S == 0)
    {
        c  = *reinterpret_cast<const int4 *>(idx + i);
        va = *reinterpret_cast<const double2 *>(val + i);
        vb = *reinterpret_cast<const double2 *>(val + i + 2);
    }
    else
    {
        typedef int    v4i __attribute__((ext_vector_type(4)));
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v4i cc = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(idx + i));
        const v2d a2 = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(val + i));
        const v2d b2 = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(val + i + 2));
        c = make_int4(cc.x, cc.y, cc.z, cc.w), va = make_double2(a2.x, a2.y), vb = make_double2(b2.x, b2.y);
    }
    const double r = fma(va.x, x[c.x], fma(va.y, x[c.y], fma(vb.x, x[c.z], vb.y * x[c.w])));
    if constexpr(S == 2)
        __builtin_nontemporal_store(r, out + blockIdx.x * BLOCK + threadIdx.x);
    else
        out[blockIdx.x * BLOCK + threadIdx.x] = r;
}

// gathers sorted inside each wavefront's 256 indices (same multiset per wavefront): how much is line sharing inside a wavefront worth?
template <typename K>
static double time_us(K launch, int reps = 200)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for(int i = 0; i < 20; i++)
        launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for(int i = 0; i < reps; i++)
        launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    const int n       = argc > 1 ? atoi(argv[1]) : 1000005;
    const int entries = (argc > 2 ? atoi(argv[2]) : 3100840) & ~3;
    const int zipf    = argc > 3 ? atoi(argv[3]) : 500;
    std::vector<int> idx(entries);
    unsigned long long s = 88172645463325252ULL;
    auto               rnd = [&]() {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        return (double)(s >> 11) / 9007199254740992.0;
    };
    for(int i = 0; i < entries; i++)
    {
        if((int)(rnd() * 1000) < zipf)
        {
            // Zipf-like with exponent 1.3: inverse-CDF of a Pareto tail, as numpy's zipf(1.3) - 1 mod n behaves for our purpose
            const double u = rnd();
            const double v = std::pow(1.0 - u, -1.0 / 0.3);
            idx[i]         = (int)std::fmod(v - 1.0, (double)n);
        }
        else
            idx[i] = (int)(rnd() * n) % n;
    }
    int    *d_idx;
    double *d_x, *d_out;
    float  *d_xf, *d_outf;
    CHECK(hipMalloc(&d_idx, sizeof(int) * entries));
    CHECK(hipMalloc(&d_x, sizeof(double) * n));
    CHECK(hipMalloc(&d_xf, sizeof(float) * n));
    CHECK(hipMalloc(&d_out, sizeof(double) * (entries / 4 + 1024)));
    CHECK(hipMalloc(&d_outf, sizeof(float) * (entries / 4 + 1024)));
    CHECK(hipMemcpy(d_idx, idx.data(), sizeof(int) * entries, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_x, 0, sizeof(double) * n));
    CHECK(hipMemset(d_xf, 0, sizeof(float) * n));
    auto report = [&](const char *name, int block, double us) {
        printf("{\"variant\": \"%s\", \"block\": %d, \"n_columns\": %d, \"entries\": %d, \"zipf_permille\": %d, \"us\": %.2f, "
               "\"gathers_per_ns\": %.1f}\n",
               name, block, n, entries, zipf, us, entries / us / 1e3);
    };
#define RUN(F, NAME)                                                                                                             \
    report(NAME, 128, time_us([&] { hipLaunchKernelGGL((gather_kernel<F, 128>), dim3((entries / 4 + 127) / 128), dim3(128), 0, 0, \
                                                       d_idx, d_x, entries, d_out); }));                                          \
    report(NAME, 256, time_us([&] { hipLaunchKernelGGL((gather_kernel<F, 256>), dim3((entries / 4 + 255) / 256), dim3(256), 0, 0, \
                                                       d_idx, d_x, entries, d_out); }));
    report("index stream only", 128, time_us([&] {
               hipLaunchKernelGGL((stream_kernel<128>), dim3((entries / 4 + 127) / 128), dim3(128), 0, 0, d_idx, entries, d_out);
           }));
    RUN(0, "plain")
    RUN(1, "nontemporal builtin")
    RUN(2, "atomic relaxed agent (sc1)")
    RUN(3, "asm sc0 sc1")
    RUN(4, "asm nt")
    RUN(5, "asm sc1 nt")
    double *d_val;
    CHECK(hipMalloc(&d_val, sizeof(double) * entries));
    CHECK(hipMemset(d_val, 0, sizeof(double) * entries));
#define RUNS(S, NAME)                                                                                                              \
    report(NAME, 128, time_us([&] { hipLaunchKernelGGL((spmv_like_kernel<S, 128>), dim3((entries / 4 + 127) / 128), dim3(128), 0, \
                                                       0, d_idx, d_val, d_x, entries, d_out); }));
    RUNS(0, "col + val + gather + store, plain streams")
    RUNS(1, "col + val + gather + store, nt col / val")
    RUNS(2, "col + val + gather + store, nt col / val / store")
    // column-range split: the same entries partitioned by x range into K groups, one launch per group (each launch gathers from
    // 1 / K of x only): is the sum of the K launches shorter than the one launch over all of x?
    for(int K : {2, 4, 8})
    {
        std::vector<std::vector<int>> part(K);
        for(int i = 0; i < entries; i++)
            part[(long long)idx[i] * K / n].push_back(idx[i]);
        std::vector<int *> d_part(K);
        std::vector<int>   cnt(K);
        for(int k = 0; k < K; k++)
        {
            while(part[k].size() % 4)
                part[k].push_back(part[k].back());
            cnt[k] = (int)part[k].size();
            CHECK(hipMalloc(&d_part[k], sizeof(int) * (cnt[k] + 4)));
            CHECK(hipMemcpy(d_part[k], part[k].data(), sizeof(int) * cnt[k], hipMemcpyHostToDevice));
        }
        char name[96];
        snprintf(name, sizeof(name), "plain, %d launches, one x range each", K);
        report(name, 128, time_us([&] {
                   for(int k = 0; k < K; k++)
                       hipLaunchKernelGGL((gather_kernel<0, 128>), dim3((cnt[k] / 4 + 127) / 128), dim3(128), 0, 0, d_part[k], d_x,
                                          cnt[k], d_out);
               }));
        for(int k = 0; k < K; k++)
            CHECK(hipFree(d_part[k]));
    }
    report("plain, float x", 128, time_us([&] {
               hipLaunchKernelGGL((gather_f32_kernel<128>), dim3((entries / 4 + 127) / 128), dim3(128), 0, 0, d_idx, d_xf, entries,
                                  d_outf);
           }));
    return 0;
}
