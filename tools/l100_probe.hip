// l100_probe.hip -- diagnostic build (never shipped): what does ONE aoclsparse_dmv call on BASELINE configs[1] literal
// (10k x 10k 5-point Laplacian, nnz = 49,600) cost a C caller, next to the launch floor of the box?
//   hipcc -O3 --offload-arch=gfx950 -Iinclude tools/l100_probe.hip -Laocl-sparse_amd/lib -laoclsparse_mi355
//         -Wl,-rpath,'$ORIGIN/../../aocl-sparse_amd/lib' -o tools/bin/l100_probe
//   l100_probe [calls=5000]
// Prints one JSON line.  For each of {empty kernel, aoclsparse_dmv device pointers, aoclsparse_dmv in pointer-auto mode}:
//   host_us  = wall clock of the calling loop / calls, WITHOUT waiting for the device (what the caller's thread pays per call;
//              when the device is the slower side the queue fills and this converges to the device figure)
//   total_us = the same loop + one stream synchronisation at the end (throughput per call)
// and for dmv the time of a single call + synchronisation (latency of one product, host to host).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
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
#define OK(x)                                                            \
    do                                                                   \
    {                                                                    \
        aoclsparse_status s_ = (x);                                      \
        if(s_ != aoclsparse_status_success)                              \
        {                                                                \
            printf("aoclsparse status %d at line %d\n", (int)s_, __LINE__); \
            exit(1);                                                     \
        }                                                                \
    } while(0)

__global__ void empty_kernel(int *sink)
{
    if(sink && threadIdx.x == 4096)
        *sink = 1;
}

// the argument list of sell_mv_kernel (13 arguments, 96 bytes), no work
__global__ void fat_kernel(int a, int b, const long long *c, const double *d, const int *e, const int *f, double g, const double *h,
                           double i, double *j, bool k, const long long *l, const unsigned short *m)
{
    if(threadIdx.x == 4096 && j)
        *j = a + b + g + i + k + (c != nullptr) + (d != nullptr) + (e != nullptr) + (f != nullptr) + (h != nullptr) + (l != nullptr) + (m != nullptr);
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename F>
static void loop(const char *name, int calls, F &&f, bool last)
{
    for(int i = 0; i < 200; i++)
        f();
    CHECK(hipDeviceSynchronize());
    double best_host = 1e30, best_total = 1e30;
    for(int rep = 0; rep < 5; rep++)
    {
        const double t0 = now_us();
        for(int i = 0; i < calls; i++)
            f();
        const double t1 = now_us();
        CHECK(hipDeviceSynchronize());
        const double t2 = now_us();
        best_host       = std::min(best_host, (t1 - t0) / calls);
        best_total      = std::min(best_total, (t2 - t0) / calls);
    }
    // short bursts: the queue never fills, so this is the pure host cost of a call
    double burst = 1e30;
    for(int rep = 0; rep < 50; rep++)
    {
        const double t0 = now_us();
        for(int i = 0; i < 16; i++)
            f();
        const double t1 = now_us();
        CHECK(hipDeviceSynchronize());
        burst = std::min(burst, (t1 - t0) / 16);
    }
    double single = 1e30;
    for(int rep = 0; rep < 200; rep++)
    {
        const double t0 = now_us();
        f();
        CHECK(hipDeviceSynchronize());
        single = std::min(single, now_us() - t0);
    }
    printf("\"%s\": {\"host_us\": %.3f, \"host_us_burst_of_16\": %.3f, \"total_us\": %.3f, \"single_call_and_sync_us\": %.3f}%s", name,
           best_host, burst, best_total, single, last ? "" : ", ");
}

int main(int argc, char **argv)
{
    const int calls = argc > 1 ? atoi(argv[1]) : 5000;
    const int g     = 100, m = g * g;
    std::vector<aoclsparse_int> rp(m + 1), ci;
    std::vector<double>         v;
    rp[0] = 0;
    for(int i = 0; i < g; i++)
        for(int j = 0; j < g; j++)
        {
            const int r = i * g + j;
            if(i > 0)
                ci.push_back(r - g), v.push_back(-1.0);
            if(j > 0)
                ci.push_back(r - 1), v.push_back(-1.0);
            ci.push_back(r), v.push_back(4.0);
            if(j < g - 1)
                ci.push_back(r + 1), v.push_back(-1.0);
            if(i < g - 1)
                ci.push_back(r + g), v.push_back(-1.0);
            rp[r + 1] = (aoclsparse_int)ci.size();
        }
    aoclsparse_matrix    A;
    aoclsparse_mat_descr d;
    OK(aoclsparse_create_dcsr(&A, aoclsparse_index_base_zero, m, m, (aoclsparse_int)v.size(), rp.data(), ci.data(), v.data()));
    OK(aoclsparse_create_mat_descr(&d));
    OK(aoclsparse_set_mv_hint(A, aoclsparse_operation_none, d, 1000));
    OK(aoclsparse_optimize(A));
    double *x, *y;
    int    *sink;
    CHECK(hipMalloc(&x, m * sizeof(double)));
    CHECK(hipMalloc(&y, m * sizeof(double)));
    CHECK(hipMalloc(&sink, sizeof(int)));
    std::vector<double> hx(m, 1.0);
    CHECK(hipMemcpy(x, hx.data(), m * sizeof(double), hipMemcpyHostToDevice));
    const double alpha = 1.0, beta = 0.0;
    printf("{\"probe\": \"per-call cost of aoclsparse_dmv on the 10k x 10k Laplacian, C caller\", \"calls\": %d, ", calls);
    loop("empty_kernel_launch", calls, [&] { hipLaunchKernelGGL(empty_kernel, dim3(157), dim3(64), 0, 0, sink); }, false);
    loop("empty_kernel_13_arguments", calls,
         [&] { hipLaunchKernelGGL(fat_kernel, dim3(157), dim3(64), 0, 0, 1, 2, nullptr, x, nullptr, nullptr, 1.0, x, 0.0, y, false, nullptr, nullptr); },
         false);
    OK(aoclsparse_mi355_set_pointer_mode(aoclsparse_mi355_pointer_device));
    loop("dmv_pointer_mode_device", calls, [&] { aoclsparse_dmv(aoclsparse_operation_none, &alpha, A, d, x, &beta, y); }, false);
    OK(aoclsparse_mi355_set_pointer_mode(aoclsparse_mi355_pointer_auto));
    loop("dmv_pointer_mode_auto", calls, [&] { aoclsparse_dmv(aoclsparse_operation_none, &alpha, A, d, x, &beta, y); }, true);
    printf("}\n");
    std::vector<double> hy(m);
    CHECK(hipMemcpy(hy.data(), y, m * sizeof(double), hipMemcpyDeviceToHost));
    if(hy[0] != 2.0 || hy[m / 2 + g / 2] != 0.0)
    {
        printf("wrong result %g %g\n", hy[0], hy[m / 2 + g / 2]);
        return 1;
    }
    aoclsparse_destroy_mat_descr(d);
    aoclsparse_destroy(&A);
    return 0;
}
