// xcd_hop_probe.hip -- diagnostic build (never shipped): what does ONE dependent hand-off between two wavefronts cost on
// this part, as the sync-free TRSV makes one per level (the solved x[k] is the flag: a consumer polls it, then computes and
// publishes its own)?  A chain of single-wavefront workgroups: workgroup i polls slot i-1 until it holds a value, adds one,
// stores slot i.  Placement and the scope of the loads / stores are varied:
//   spread   : workgroup i runs wherever the dispatcher puts it (consecutive workgroups on consecutive XCDs)
//   one XCD  : 8 x the workgroups are launched, those that do not find themselves on XCD `pick` (HW_REG_XCC_ID) leave at once,
//              the others take their chain position from a counter -- the whole chain lives behind ONE L2
//   scope    : agent (what the product uses: sc1 loads / stores), workgroup-style sc0 (L1 bypassed, the XCD's L2 answers),
//              system (sc0 sc1)
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_hop_probe.hip -o tools/bin/xcd_hop_probe
// Prints one JSON line per case: ns per hop (chain of `links` workgroups walked `rounds` times inside one launch).
#include <hip/hip_runtime.h>

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

typedef unsigned long long u64;
static constexpr u64 EMPTY = ~0ull;

template <int SCOPE>
__device__ __forceinline__ u64 poll_load(const u64 *p)
{
    u64 v;
    if constexpr(SCOPE == 0)
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr(SCOPE == 1)
    {
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    }
    else if constexpr(SCOPE == 2)
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else // 3, 4: a returning read-modify-write that changes nothing -- executed AT the L2
        v = __hip_atomic_fetch_or(const_cast<u64 *>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

template <int SCOPE>
__device__ __forceinline__ void publish(u64 *p, u64 v)
{
    if constexpr(SCOPE == 0)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr(SCOPE == 1)
    {
        asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
    }
    else if constexpr(SCOPE == 2)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if constexpr(SCOPE == 3)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else // 4: the store is a read-modify-write too
        (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// slots: rounds * links + 1 words, slot 0 preset to 0, the others EMPTY.  pick < 0: every workgroup takes part.
template <int SCOPE>
__global__ __launch_bounds__(64) void chain_kernel(u64 *slots, int links, int rounds, int pick, unsigned *ticket, unsigned *xcc_of,
                                                   int stride)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15u;
    if(threadIdx.x == 0 && xcc_of)
        xcc_of[blockIdx.x] = xcc;
    int me;
    if(pick >= 0)
    {
        if((int)xcc != pick)
            return;
        unsigned t = 0;
        if(threadIdx.x == 0)
            t = atomicAdd(ticket, 1u);
        me = __builtin_amdgcn_readfirstlane((int)t);
        if(me >= links)
            return;
    }
    else
        me = blockIdx.x;
    if(threadIdx.x != 0)
        return;
    for(int r = 0; r < rounds; r++)
    {
        const long at = ((long)r * links + me) * stride;
        u64        v;
        long       spins = 0;
        while((v = poll_load<SCOPE>(slots + at)) == EMPTY)
        {
            __builtin_amdgcn_s_sleep(1);
            if(++spins > 3000000L)
                return; // (a lost chain must not hang the box)
        }
        publish<SCOPE>(slots + at + stride, v + 1);
    }
}

template <int SCOPE>
static void run_case(const char *scope_name, int links, int rounds, int pick, int stride)
{
    const long   nslots = ((long)links * rounds + 1) * stride;
    u64         *d_slots;
    unsigned    *d_ticket, *d_xcc;
    const int    grid = pick >= 0 ? links * 8 : links;
    CHECK(hipMalloc(&d_slots, nslots * 8));
    CHECK(hipMalloc(&d_ticket, 4));
    CHECK(hipMalloc(&d_xcc, (size_t)grid * 4));
    std::vector<u64> h(nslots, EMPTY);
    h[0] = 0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    u64   last = 0;
    for(int rep = 0; rep < 4; rep++)
    {
        CHECK(hipMemcpy(d_slots, h.data(), nslots * 8, hipMemcpyHostToDevice));
        CHECK(hipMemset(d_ticket, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        chain_kernel<SCOPE><<<grid, 64>>>(d_slots, links, rounds, pick, d_ticket, d_xcc, stride);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if(rep > 0 && ms < best)
            best = ms;
        CHECK(hipMemcpy(&last, d_slots + (nslots - stride), 8, hipMemcpyDeviceToHost));
    }
    std::vector<unsigned> xcc(grid);
    CHECK(hipMemcpy(xcc.data(), d_xcc, (size_t)grid * 4, hipMemcpyDeviceToHost));
    int rr = 0; // does workgroup i sit on XCD i % 8?
    for(int i = 0; i < grid; i++)
        rr += (int)xcc[i] == i % 8;
    const long hops = (long)links * rounds;
    printf("{\"placement\": \"%s\", \"scope\": \"%s\", \"links\": %d, \"rounds\": %d, \"slot_stride_bytes\": %d, \"ns_per_hop\": %.1f, "
           "\"chain_complete\": %s, \"workgroups_on_xcd_i_mod_8\": \"%d of %d\"}\n",
           pick >= 0 ? "one XCD" : "spread", scope_name, links, rounds, stride * 8, best * 1e6 / hops,
           last == (u64)hops ? "true" : "false", rr, grid);
    CHECK(hipFree(d_slots));
    CHECK(hipFree(d_ticket));
    CHECK(hipFree(d_xcc));
}

int main(int argc, char **argv)
{
    const int links  = argc > 1 ? atoi(argv[1]) : 256;
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    for(int stride : {1, 16})
    {
        run_case<0>("agent", links, rounds, -1, stride);
        if(argc > 3)
            run_case<1>("sc0", links, rounds, -1, stride);
        run_case<2>("system", links, rounds, -1, stride);
        run_case<3>("agent, polled with an L2 atomic", links, rounds, -1, stride);
        run_case<4>("agent, L2 atomics both ways", links, rounds, -1, stride);
        run_case<0>("agent", links, rounds, 0, stride);
        if(argc > 3) // (neither placement completes the chain with sc0: the load may be served by the CU's own L1)
            run_case<1>("sc0", links, rounds, 0, stride);
        run_case<2>("system", links, rounds, 0, stride);
        run_case<3>("agent, polled with an L2 atomic", links, rounds, 0, stride);
        run_case<4>("agent, L2 atomics both ways", links, rounds, 0, stride);
    }
    return 0;
}
