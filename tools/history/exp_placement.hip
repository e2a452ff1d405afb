// round 5 experiment: does the streaming-read rate of a buffer depend on WHERE hipMalloc put it?
// K buffers of SIZE MB each; each is read end to end (16-byte loads, grid-stride, 2,048 workgroups) REPS times and timed with
// events; then the slowest and the fastest buffer are read segment by segment (SEG MB) to see whether the difference is uniform.
// hipcc --offload-arch=gfx950 -O3 tools/exp_placement.hip -o tools/bin/exp_placement && tools/bin/exp_placement [K=8] [SIZE_MB=448]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(e)                                                                                    \
    do                                                                                              \
    {                                                                                               \
        hipError_t s_ = (e);                                                                        \
        if(s_ != hipSuccess)                                                                        \
        {                                                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(s_));               \
            exit(1);                                                                                \
        }                                                                                           \
    } while(0)

__global__ __launch_bounds__(256) void read_kernel(const double2 *p, size_t n16, double *sink)
{
    double acc = 0;
    for(size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if(acc == 123.456)
        *sink = acc;
}

static double time_read(const void *p, size_t bytes, double *sink, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const int grid = (int)std::min<size_t>(2048, (bytes / 16 + 255) / 256);
    for(int r = 0; r < 3; r++)
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, (const double2 *)p, bytes / 16, sink);
    CHECK(hipEventRecord(a));
    for(int r = 0; r < reps; r++)
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, (const double2 *)p, bytes / 16, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return (double)bytes * reps / (ms * 1e-3) / 1e12; // TB/s
}

int main(int argc, char **argv)
{
    const int    K    = argc > 1 ? atoi(argv[1]) : 8;
    const size_t size = (size_t)(argc > 2 ? atoi(argv[2]) : 448) << 20;
    const size_t seg  = (size_t)32 << 20;
    double      *sink;
    CHECK(hipMalloc(&sink, 8));
    std::vector<void *> buf(K);
    std::vector<double> rate(K);
    for(int k = 0; k < K; k++)
    {
        CHECK(hipMalloc(&buf[k], size));
        CHECK(hipMemset(buf[k], 0, size));
    }
    CHECK(hipDeviceSynchronize());
    for(int round = 0; round < 2; round++)
    {
        printf("round %d, TB/s per buffer (allocation order):", round);
        for(int k = 0; k < K; k++)
        {
            rate[k] = time_read(buf[k], size, sink, 20);
            printf(" %.3f", rate[k]);
        }
        printf("\n");
    }
    for(int k = 0; k < K; k++)
        printf("buffer %d at %p\n", k, buf[k]);
    const int lo = (int)(std::min_element(rate.begin(), rate.end()) - rate.begin());
    const int hi = (int)(std::max_element(rate.begin(), rate.end()) - rate.begin());
    for(int which : {lo, hi})
    {
        printf("buffer %d (%s), TB/s per %zu MB segment:", which, which == lo ? "slowest" : "fastest", seg >> 20);
        for(size_t off = 0; off + seg <= size; off += seg)
            printf(" %.2f", time_read((const char *)buf[which] + off, seg, sink, 40));
        printf("\n");
    }
    // two buffers read by ONE launch each half (as a kernel with two streams of operands does)
    return 0;
}
