// mfma_f64_peak.hip -- diagnostic build (never shipped), round 4: what does v_mfma_f64_16x16x4_f64 sustain on this part when nothing
// else is going on?  Every wavefront issues ITERS x ACCS back-to-back MFMAs on ACCS independent accumulators, operands in registers.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_peak.hip -o tools/bin/mfma_f64_peak ;  mfma_f64_peak [waves_per_simd=2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while(0)
template <int ACCS>
__global__ __launch_bounds__(256) void peak(double *out, int iters, double a0, double b0)
{
    v4d acc[ACCS];
#pragma unroll
    for(int u = 0; u < ACCS; u++)
        acc[u] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for(int it = 0; it < iters; it++)
    {
#pragma unroll
        for(int u = 0; u < ACCS; u++)
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for(int u = 0; u < ACCS; u++)
        s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACCS>
static void run(int wps, int iters)
{
    int cus = 256;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    cus = p.multiProcessorCount;
    const int blocks = cus * wps; // 4 waves per block = one per SIMD; wps blocks per CU
    double   *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    peak<ACCS><<<blocks, 256>>>(out, 100, 1.0, 1e-3);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for(int r = 0; r < 5; r++)
    {
        CHECK(hipEventRecord(e0));
        peak<ACCS><<<blocks, 256>>>(out, iters, 1.0, 1e-3);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if(ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * ACCS * 2048.0;
    printf("{\"probe\": \"v_mfma_f64_16x16x4_f64 back to back\", \"accumulators\": %d, \"waves_per_simd\": %d, \"cus\": %d, \"ms\": %.3f, "
           "\"tflops\": %.2f, \"clocks_per_mfma_at_2.4GHz\": %.1f}\n", ACCS, wps, cus, best, flops / best / 1e9,
           best * 1e-3 * 2.4e9 / ((double)iters * ACCS * wps));
    CHECK(hipFree(out));
}
int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    run<1>(wps, 20000);
    run<2>(wps, 10000);
    run<4>(wps, 5000);
    run<8>(wps, 2500);
    return 0;
}
