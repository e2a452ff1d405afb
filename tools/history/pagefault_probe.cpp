// pagefault_probe.cpp -- diagnostic: what does first-touching a fresh 156 MB host array cost on this box (the result arrays of an
// sp2m: the copy out of HBM faults them in), by allocation and by the number of touching threads?
//   g++ -O2 -pthread tools/pagefault_probe.cpp -o tools/bin/pagefault_probe
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double touch(char *p, size_t bytes, int nt, size_t step)
{
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for(int t = 0; t < nt; t++)
        th.emplace_back([=] {
            const size_t a = bytes * t / nt, b = bytes * (t + 1) / nt;
            for(size_t o = a / step * step; o < b; o += step)
                if(o >= a)
                    ((volatile char *)p)[o] = 0;
        });
    for(auto &x : th)
        x.join();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main()
{
    const size_t bytes = 156u << 20;
    printf("{\"hardware_threads\": %u", std::thread::hardware_concurrency());
    for(int nt : {1, 4, 16})
    {
        char *p = new char[bytes];
        printf(", \"new_%dthreads_ms\": %.2f", nt, touch(p, bytes, nt, 4096));
        delete[] p;
    }
    for(int nt : {1, 4, 16})
    {
        void *q = nullptr;
        if(posix_memalign(&q, 2u << 20, bytes))
            return 1;
        madvise(q, bytes, MADV_HUGEPAGE);
        printf(", \"thp_%dthreads_ms\": %.2f", nt, touch((char *)q, bytes, nt, 4096));
        free(q);
    }
    {
        char *p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        auto  t0 = std::chrono::steady_clock::now();
        char *r = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        printf(", \"mmap_populate_ms\": %.2f", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        munmap(p, bytes), munmap(r, bytes);
    }
    FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char  buf[128] = {0};
    if(f)
    {
        if(fgets(buf, 127, f))
            buf[strcspn(buf, "\n")] = 0;
        fclose(f);
    }
    printf(", \"thp_enabled\": \"%s\"}\n", buf);
    return 0;
}
