// h2d_probe.hip -- diagnostic build (never shipped): what limits a host -> device copy of PAGEABLE memory on this box?
//   hipcc -O3 --offload-arch=gfx950 tools/h2d_probe.hip -o tools/bin/h2d_probe -lpthread
// Measures, for 1 GiB: (a) hipMemcpy from pageable memory, (b) hipMemcpy from pinned memory, (c) memcpy pageable ->
// pinned with 1 / 2 / 4 / 8 / 16 threads, (d) the pipelined scheme of csrc/runtime.cpp (threads fill pinned ring slots
// while the DMA engine drains the previous ones) over chunk sizes and thread counts, (e) hipHostRegister + copy.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
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

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_copy(char *dst, const char *src, size_t bytes, int nthr)
{
    if(nthr <= 1)
    {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t             per = (bytes + nthr - 1) / nthr;
    for(int t = 0; t < nthr; t++)
    {
        const size_t lo = per * t, hi = lo + per < bytes ? lo + per : bytes;
        if(lo < hi)
            th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for(auto &t : th)
        t.join();
}

int main()
{
    const size_t N = 1ull << 30;
    char        *page = (char *)malloc(N);
    memset(page, 1, N);
    char *dev, *pin;
    CHECK(hipMalloc(&dev, N));
    CHECK(hipHostMalloc(&pin, N, hipHostMallocDefault));
    memset(pin, 2, N);
    CHECK(hipMemcpy(dev, page, 1 << 20, hipMemcpyHostToDevice));
    double t;
    for(int rep = 0; rep < 2; rep++)
    {
        t = now();
        CHECK(hipMemcpy(dev, page, N, hipMemcpyHostToDevice));
        printf("{\"what\": \"hipMemcpy pageable\", \"GBs\": %.1f}\n", N / (now() - t) / 1e9);
        t = now();
        CHECK(hipMemcpy(dev, pin, N, hipMemcpyHostToDevice));
        printf("{\"what\": \"hipMemcpy pinned\", \"GBs\": %.1f}\n", N / (now() - t) / 1e9);
    }
    for(int nthr : {1, 2, 4, 8, 16})
    {
        t = now();
        par_copy(pin, page, N, nthr);
        printf("{\"what\": \"memcpy pageable -> pinned\", \"threads\": %d, \"GBs\": %.1f}\n", nthr, N / (now() - t) / 1e9);
    }
    hipEvent_t ev[4];
    for(auto &e : ev)
        CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    for(size_t chunk : {4ull << 20, 16ull << 20, 64ull << 20})
        for(int nthr : {1, 4, 8})
            for(hipStream_t st : {(hipStream_t) nullptr, s})
            {
                CHECK(hipDeviceSynchronize());
                t = now();
                int i = 0;
                for(size_t off = 0; off < N; off += chunk, i++)
                {
                    const int k = i % 3;
                    if(i >= 3)
                        CHECK(hipEventSynchronize(ev[k]));
                    par_copy(pin + (size_t)k * chunk, page + off, chunk, nthr);
                    CHECK(hipMemcpyAsync(dev + off, pin + (size_t)k * chunk, chunk, hipMemcpyHostToDevice, st));
                    CHECK(hipEventRecord(ev[k], st));
                }
                CHECK(hipStreamSynchronize(st));
                printf("{\"what\": \"pipelined ring of 3\", \"chunk_mb\": %zu, \"threads\": %d, \"stream\": \"%s\", \"GBs\": %.1f}\n",
                       chunk >> 20, nthr, st ? "own" : "null", N / (now() - t) / 1e9);
            }
    t = now();
    CHECK(hipHostRegister(page, N, hipHostRegisterDefault));
    const double treg = now() - t;
    t                 = now();
    CHECK(hipMemcpy(dev, page, N, hipMemcpyHostToDevice));
    const double tcp = now() - t;
    CHECK(hipHostUnregister(page));
    printf("{\"what\": \"hipHostRegister + hipMemcpy\", \"register_ms\": %.1f, \"copy_GBs\": %.1f, \"total_GBs\": %.1f}\n", treg * 1e3,
           N / tcp / 1e9, N / (treg + tcp) / 1e9);
    return 0;
}
