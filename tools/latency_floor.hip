// latency_floor.hip -- diagnostic build (never shipped): what does a SHORT sparse kernel cost on this box before it
// moves a single useful byte?  The irregular SuiteSparse stand-ins of BASELINE config 3 (circuit-like: 11 MB of
// algorithmic bytes, web-like: 57 MB) run for 10-30 us, where the launch and the chain of DEPENDENT memory round trips
// every workgroup must make (block table -> row_ptr / col_ind / val -> x[col]) weigh as much as the bytes.
//   hipcc -O3 --offload-arch=gfx950 tools/latency_floor.hip -o tools/bin/latency_floor
//   latency_floor [workgroups=2000] [footprint_MB=64] [stream_entries=3100840] [lanes_per_workgroup=256]
// Prints one JSON line: back-to-back time per launch of (a) an empty kernel, (b..d) a kernel whose every wavefront
// makes 1, 2, 3 DEPENDENT loads from the footprint (two coalesced ones whose address comes out of the load before -- block table
// -> column indices -- then a per-lane gather, x[col]), (e) the three followed by a 12-byte-per-entry coalesced stream.
#include <hip/hip_runtime.h>

#include <algorithm>
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

__global__ void empty_kernel(int *sink)
{
    if(sink && threadIdx.x == 4096)
        *sink = 1;
}

// The round trips of a CSR SpMV workgroup, in their cheapest form: hop 1 and hop 2 are COALESCED loads whose address depends on the
// previous load (block table -> column indices / values: one line stretch per wavefront), hop 3 is a per-lane gather (x[col]).
// HOPS = 1: the first only, 2: both coalesced ones, 3: all three.
template <int HOPS>
__global__ __launch_bounds__(256) void chase_kernel(const int *__restrict__ idx, int n, int *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int p = (int)((wave * 2654435761u) % (unsigned)(n - 128)) & ~63;
    p     = idx[p + lane];
    if(HOPS >= 2)
        p = idx[(__builtin_amdgcn_readfirstlane(p) % (n - 128) & ~63) + lane];
    if(HOPS >= 3)
        p = idx[p];
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
}

__global__ __launch_bounds__(256) void chase_stream_kernel(const int *__restrict__ idx, int n, const double *__restrict__ v,
                                                           const int *__restrict__ c, long per_wg, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int p = (int)((wave * 2654435761u) % (unsigned)(n - 128)) & ~63;
    p     = idx[p + lane];
    p     = idx[(__builtin_amdgcn_readfirstlane(p) % (n - 128) & ~63) + lane];
    p     = idx[p];
    double     acc = (double)(p & 1);
    const long s   = (long)blockIdx.x * per_wg;
    for(long q = s + threadIdx.x; q < s + per_wg; q += blockDim.x)
        acc += v[q] * (double)c[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int  wgs  = argc > 1 ? atoi(argv[1]) : 2000;
    const long fmb  = argc > 2 ? atol(argv[2]) : 64;
    const int  lanes = argc > 4 && atoi(argv[4]) == 128 ? 128 : 256;
    const int  n    = (int)(fmb * 1024 * 1024 / 4);
    std::vector<int> idx(n);
    unsigned         s = 777;
    for(int i = 0; i < n; i++)
    {
        s      = s * 1664525u + 1013904223u;
        idx[i] = (int)((s >> 4) % (unsigned)n) & ~15; // line-aligned targets: one fresh 64-byte line per hop
    }
    int    *d_idx, *d_out, *d_c;
    double *d_v, *d_o2;
    const long stream_entries = argc > 3 ? std::max(atol(argv[3]), (long)wgs) : 3100840; // default: web-like, 3.1 M non-zeros
    CHECK(hipMalloc(&d_idx, (size_t)n * 4));
    CHECK(hipMalloc(&d_out, (size_t)wgs * 256 * 4));
    CHECK(hipMalloc(&d_v, stream_entries * 8));
    CHECK(hipMalloc(&d_c, stream_entries * 4));
    CHECK(hipMalloc(&d_o2, (size_t)wgs * 256 * 8));
    CHECK(hipMemcpy(d_idx, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_v, 0, stream_entries * 8));
    CHECK(hipMemset(d_c, 0, stream_entries * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        for(int i = 0; i < 20; i++)
            launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for(int rep = 0; rep < 5; rep++)
        {
            CHECK(hipEventRecord(e0));
            for(int i = 0; i < 200; i++)
                launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 200 * 1000.0f);
        }
        return best;
    };
    const float t_empty = timeit([&] { empty_kernel<<<wgs, lanes>>>(nullptr); });
    const float t1      = timeit([&] { chase_kernel<1><<<wgs, lanes>>>(d_idx, n, d_out); });
    const float t2      = timeit([&] { chase_kernel<2><<<wgs, lanes>>>(d_idx, n, d_out); });
    const float t3      = timeit([&] { chase_kernel<3><<<wgs, lanes>>>(d_idx, n, d_out); });
    const long  per_wg  = stream_entries / wgs;
    const float t4      = timeit([&] { chase_stream_kernel<<<wgs, lanes>>>(d_idx, n, d_v, d_c, per_wg, d_o2); });
    printf("{\"probe\": \"launch + dependent-round-trip floor\", \"workgroups\": %d, \"lanes_per_workgroup\": %d, \"footprint_mb\": %ld, "
           "\"empty_kernel_us\": %.3f, \"one_hop_us\": %.3f, \"two_hops_us\": %.3f, \"three_hops_us\": %.3f, "
           "\"three_hops_then_stream_us\": %.3f, \"stream_entries\": %ld, \"per_hop_us\": %.3f}\n",
           wgs, lanes, fmb, t_empty, t1, t2, t3, t4, stream_entries, (t3 - t1) / 2);
    return 0;
}
