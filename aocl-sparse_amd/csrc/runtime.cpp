// runtime.cpp -- device context, pointer classification, staging buffers, misc. entry points.
#include "internal.hpp"

#include <dlfcn.h>
#include <sys/mman.h>

#include <cctype>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <sched.h>

namespace mi355
{

aoclsparse_status map_hip_error(hipError_t e)
{
    // hipError_t -> memory_error / internal_error; never throw across the ABI
    if(e == hipSuccess)
        return aoclsparse_status_success;
    (void)hipGetLastError();
    if(e == hipErrorOutOfMemory)
        return aoclsparse_status_memory_error;
    return aoclsparse_status_internal_error;
}

DeviceBuffer::~DeviceBuffer()
{
    release();
}

void DeviceBuffer::release()
{
    if(ptr)
        (void)hipFree(ptr);
    ptr   = nullptr;
    bytes = 0;
}

aoclsparse_status DeviceBuffer::alloc(size_t nbytes)
{
    if(nbytes <= bytes && ptr)
        return aoclsparse_status_success;
    release();
    size_t want = nbytes ? nbytes : 8;
    MI355_HIP_TRY(hipMalloc(&ptr, want));
    bytes = want;
    return aoclsparse_status_success;
}

aoclsparse_status DeviceBuffer::upload(const void *host, size_t nbytes, hipStream_t s)
{
    aoclsparse_status st = alloc(nbytes);
    if(st != aoclsparse_status_success)
        return st;
    if(nbytes)
    {
        if(s == Runtime::get().stream())
        {
            st = Runtime::get().h2d(ptr, host, nbytes);
            if(st != aoclsparse_status_success)
                return st;
        }
        else
            MI355_HIP_TRY(hipMemcpyAsync(ptr, host, nbytes, hipMemcpyHostToDevice, s));
        MI355_HIP_TRY(hipStreamSynchronize(s)); // host arrays are pageable: complete before return
    }
    return aoclsparse_status_success;
}

aoclsparse_status DeviceBuffer::clone_from(const DeviceBuffer &src, hipStream_t s)
{
    if(!src.ptr || !src.bytes)
    {
        release();
        return aoclsparse_status_success;
    }
    aoclsparse_status st = alloc(src.bytes);
    if(st != aoclsparse_status_success)
        return st;
    MI355_HIP_TRY(hipMemcpyAsync(ptr, src.ptr, src.bytes, hipMemcpyDeviceToDevice, s));
    return aoclsparse_status_success;
}

HostCsr::~HostCsr()
{
    if(owned)
    {
        delete[] ptr;
        if(result_arrays)
        {
            std::free(ind);
            std::free(val);
        }
        else
        {
            delete[] ind;
            ::operator delete(val);
        }
    }
    delete[] idiag;
    delete[] iurow;
}

constexpr size_t HOST_RESULT_BIG = (size_t)4 << 20, HOST_RESULT_PAGE = (size_t)2 << 20;

void *host_result_alloc(size_t bytes)
{
    if(bytes < HOST_RESULT_BIG)
        return std::malloc(bytes ? bytes : 1);
    void *p = nullptr;
    if(posix_memalign(&p, HOST_RESULT_PAGE, (bytes + HOST_RESULT_PAGE - 1) / HOST_RESULT_PAGE * HOST_RESULT_PAGE) != 0)
        return nullptr;
    (void)madvise(p, bytes, MADV_HUGEPAGE); // (advice: without transparent huge pages the array is simply 4 KB pages)
    return p;
}

void host_result_touch(void *p, size_t bytes)
{
    if(!p || bytes < HOST_RESULT_BIG)
        return;
    const long long pages = (long long)((bytes + HOST_RESULT_PAGE - 1) / HOST_RESULT_PAGE);
    parallel_for(pages, 4, [&](long long p0, long long p1) {
        volatile char *b = static_cast<volatile char *>(p);
        for(long long pg = p0; pg < p1; pg++)
            for(size_t o = (size_t)pg * HOST_RESULT_PAGE; o < std::min(bytes, (size_t)(pg + 1) * HOST_RESULT_PAGE); o += 4096)
                b[o] = 0;
    });
}

thread_local Runtime *tl_runtime = nullptr;

Runtime &Runtime::primary()
{
    static Runtime r;
    return r;
}

Runtime &Runtime::get()
{
    return tl_runtime ? *tl_runtime : primary();
}

namespace
{
    // secondary runtimes live for the life of the process (like the primary one)
    std::mutex &slot_mutex()
    {
        static std::mutex m;
        return m;
    }
    std::vector<Runtime *> &slot_table()
    {
        static std::vector<Runtime *> slots;
        return slots;
    }
}

std::vector<Runtime *> Runtime::secondary()
{
    std::lock_guard<std::mutex> g(slot_mutex());
    std::vector<Runtime *>      out;
    for(Runtime *r : slot_table())
        if(r)
            out.push_back(r);
    return out;
}

Runtime *Runtime::slot(int idx, int dev)
{
    std::vector<Runtime *> &slots = slot_table();
    if(idx < 1 || idx > 63 || dev < 0)
        return nullptr;
    std::lock_guard<std::mutex> g(slot_mutex());
    if((int)slots.size() <= idx)
        slots.resize((size_t)idx + 1, nullptr);
    if(!slots[idx])
    {
        slots[idx] = new(std::nothrow) Runtime;
        if(slots[idx])
            slots[idx]->forced_device = dev;
    }
    else if(slots[idx]->forced_device != dev)
        return nullptr; // a slot keeps its device: its staging buffers and the replicas built under it live there
    return slots[idx];
}

RuntimeScope::RuntimeScope(Runtime *r) : prev(tl_runtime), status(aoclsparse_status_internal_error)
{
    if(!r)
        return;
    Runtime &pr = Runtime::primary();
    if(r != &pr)
    {
        r->pointer_mode = pr.pointer_mode;
    }
    tl_runtime = r;
    status     = r->init();
}

RuntimeScope::~RuntimeScope()
{
    tl_runtime = prev;
    Runtime &back = Runtime::get();
    if(back.device >= 0)
        back.bind_thread();
}

thread_local int tl_bound_device = -1;

aoclsparse_status Runtime::init()
{
    // double-checked: the flag is an atomic stored with release order after everything it guards
    if(inited_.load(std::memory_order_acquire))
    {
        bind_thread();
        return init_status_;
    }
    std::lock_guard<std::mutex> g(lock);
    if(inited_.load(std::memory_order_relaxed))
    {
        // lost the first-init race: this thread still has to be bound to the library's device
        bind_thread();
        return init_status_;
    }
    int        count = 0;
    hipError_t e     = hipGetDeviceCount(&count);
    if(e != hipSuccess || count <= 0)
    {
        (void)hipGetLastError();
        // the HIP path is the product: fail loudly, there is no CPU fallback
        std::fprintf(stderr,
                     "aoclsparse(mi355): no HIP device available (%s); this library has no CPU "
                     "fallback\n",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        init_status_ = aoclsparse_status_internal_error;
        inited_.store(true, std::memory_order_release);
        return init_status_;
    }
    if(forced_device >= 0)
    {
        // a secondary slot of a multi-device call: its own device and its own BLOCKING stream -- a blocking stream is ordered
        // against that device's null stream, where a caller that has just filled B_slabs[i] / zeroed C_slabs[i] (torch's
        // default stream) has its work in flight; a non-blocking stream would race with it
        if(forced_device >= count || hipSetDevice(forced_device) != hipSuccess)
        {
            (void)hipGetLastError();
            init_status_ = aoclsparse_status_invalid_value;
            inited_.store(true, std::memory_order_release);
            return init_status_;
        }
        if(hipStreamCreateWithFlags(&stream_, hipStreamDefault) != hipSuccess)
        {
            (void)hipGetLastError();
            stream_ = nullptr;
        }
    }
    // one process per GPU: honour an explicit ordinal, else keep the caller's current device
    else if(const char *env = std::getenv("AOCLSPARSE_MI355_DEVICE"))
    {
        int d = std::atoi(env);
        if(d >= 0 && d < count)
            (void)hipSetDevice(d);
    }
    (void)hipGetDevice(&device);
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, device) == hipSuccess)
    {
        cus = prop.multiProcessorCount;
        std::snprintf(name, sizeof(name), "%s (%s)", prop.name, prop.gcnArchName);
    }
    if(hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess)
        init_status_ = aoclsparse_status_internal_error;
    void *tw = nullptr, *twd = nullptr;
    if(hipHostMalloc(&tw, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&twd, tw, 0) == hipSuccess)
    {
        trsv_timeout_host  = static_cast<volatile unsigned int *>(tw);
        *trsv_timeout_host = 0;
        trsv_timeout_dev   = static_cast<unsigned int *>(twd);
        plan_stale_host    = trsv_timeout_host + 1;
        *plan_stale_host   = 0;
        plan_stale_dev     = trsv_timeout_dev + 1;
    }
    else
        (void)hipGetLastError(); // the solves then keep their device-side word and a blocking read
    tl_bound_device = device;
    inited_.store(true, std::memory_order_release);
    return init_status_;
}

// The HIP current device is per THREAD: a thread other than the initialising one would launch on device 0 while
// every buffer of the library lives on `device`.  Every ABI entry point passes through init(), which binds the calling
// thread once.
void Runtime::bind_thread()
{
    if(tl_bound_device != device && device >= 0)
    {
        (void)hipSetDevice(device);
        tl_bound_device = device;
    }
}

thread_local int tl_device_scope = 0;
DeviceScope::DeviceScope()
{
    ++tl_device_scope;
}
DeviceScope::~DeviceScope()
{
    --tl_device_scope;
}

// Process-wide mode word.  AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE is read exactly once, when the word is first touched -- by
// the setter or by the first product, whichever comes first -- so an explicit aoclsparse_mi355_set_csrmm_beta0_overwrite() made
// before the first product is never overridden by the environment afterwards.
std::atomic<bool> &csrmm_beta0_overwrite_flag()
{
    static std::atomic<bool> flag{[] {
        const char *e = std::getenv("AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE");
        return e && std::atoi(e) != 0;
    }()};
    return flag;
}

namespace
{
    std::atomic<int> g_plan_options[aoclsparse_mi355_option_count] = {{0}, {-1}, {0}, {1}, {-1}};
}
unsigned long long stream_uid(hipStream_t s)
{
    using get_id_t = hipError_t (*)(hipStream_t, unsigned long long *);
    // looked up in THE runtime this library is bound to (the one hipStreamCreate resolves to), not in whatever the process holds:
    // a process can hold two copies of libamdhip64 (aoclsparse_mi355_hip_runtime_path), and a stream is an object of one of them
    static const get_id_t get_id = [] {
        Dl_info info;
        if(!dladdr(reinterpret_cast<const void *>(&hipStreamCreate), &info) || !info.dli_fname)
            return static_cast<get_id_t>(nullptr);
        void *h = dlopen(info.dli_fname, RTLD_LAZY | RTLD_NOLOAD);
        return h ? reinterpret_cast<get_id_t>(dlsym(h, "hipStreamGetId")) : static_cast<get_id_t>(nullptr);
    }();
    unsigned long long id = 0;
    if(!get_id)
        return 0;
    if(get_id(s, &id) != hipSuccess)
    {
        (void)hipGetLastError();
        id = 0;
    }
    return id;
}

int plan_option(aoclsparse_mi355_option option)
{
    return option >= 0 && option < aoclsparse_mi355_option_count ? g_plan_options[option].load(std::memory_order_relaxed) : 0;
}

bool csrmm_reads_c(bool beta_nonzero)
{
    return beta_nonzero || !csrmm_beta0_overwrite_flag().load(std::memory_order_relaxed);
}

bool Runtime::is_device_pointer(const void *p)
{
    if(tl_device_scope > 0) // a composite routine calling the executors on its own device buffers
        return true;
    if(pointer_mode == aoclsparse_mi355_pointer_device)
        return true;
    if(pointer_mode == aoclsparse_mi355_pointer_host)
        return false;
    hipPointerAttribute_t attr;
    hipError_t            e = hipPointerGetAttributes(&attr, p);
    if(e != hipSuccess)
    {
        (void)hipGetLastError(); // plain malloc'ed memory is "invalid value": host
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged
           || attr.type == hipMemoryTypeUnified;
}

aoclsparse_status Runtime::staging(int slot, size_t bytes, void **out)
{
    aoclsparse_status st = stage_[slot].alloc(bytes);
    *out                 = stage_[slot].ptr;
    return st;
}

size_t Runtime::release_staging()
{
    std::lock_guard<std::recursive_mutex> sl(stage_lock);
    if(stream_)
        (void)hipStreamSynchronize(stream_); // (a kernel of the last call may still read a slot)
    size_t freed = 0;
    for(DeviceBuffer &b : stage_)
    {
        freed += b.ptr ? b.bytes : 0;
        b.release();
    }
    return freed;
}

// ---- pageable <-> device copies -----------------------------------------------------------------------------------------
// Plain stream-ordered copies.  A hand-made pipeline through a ring of pinned slots (round 2) measured SLOWER than the ROCm
// runtime's own staging of pageable memory (51-54 vs 56 GB/s, profiles/r2/h2d_probe.jsonl) and was removed in round 3.
aoclsparse_status Runtime::h2d(void *dev, const void *host, size_t bytes)
{
    if(bytes)
        MI355_HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream()));
    return aoclsparse_status_success;
}

aoclsparse_status Runtime::d2h(void *host, const void *dev, size_t bytes)
{
    if(bytes)
        MI355_HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream()));
    return aoclsparse_status_success;
}

size_t val_size(aoclsparse_matrix_data_type t)
{
    return t == aoclsparse_smat ? sizeof(float) : (t == aoclsparse_zmat ? 2 * sizeof(double) : sizeof(double)); // cmat: 2 floats
}

} // namespace mi355

using namespace mi355;

extern "C" {

aoclsparse_status aoclsparse_mi355_set_csrmm_beta0_overwrite(int overwrite)
{
    csrmm_beta0_overwrite_flag().store(overwrite != 0, std::memory_order_relaxed); // process-wide: every runtime slot reads it
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_release_staging(size_t *bytes_freed)
{
    // process-wide: the primary runtime (every host thread stages through it, under its stage_lock) and the secondary slots of
    // the multi-device calls, each synchronised and freed with the thread bound to its own device
    size_t f = Runtime::primary().release_staging();
    for(Runtime *r : Runtime::secondary())
    {
        RuntimeScope scope(r);
        if(scope.status == aoclsparse_status_success)
            f += r->release_staging();
    }
    if(bytes_freed)
        *bytes_freed = f;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_set_option(aoclsparse_mi355_option option, aoclsparse_int value)
{
    if(option < 0 || option >= aoclsparse_mi355_option_count)
        return aoclsparse_status_invalid_value;
    if(option == aoclsparse_mi355_option_spmv_kernel && (value < 0 || value > 2))
        return aoclsparse_status_invalid_value;
    if(option == aoclsparse_mi355_option_sell && (value < -1 || value > 1))
        return aoclsparse_status_invalid_value;
    if(option == aoclsparse_mi355_option_spmv_strict && (value < 0 || value > 1))
        return aoclsparse_status_invalid_value;
    if(option == aoclsparse_mi355_option_alternate_sweeps && (value < 0 || value > 1))
        return aoclsparse_status_invalid_value;
    if(option == aoclsparse_mi355_option_trsv_chunks && (value < -1 || value > 1))
        return aoclsparse_status_invalid_value;
    g_plan_options[option].store((int)value, std::memory_order_relaxed);
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_set_pointer_mode(aoclsparse_mi355_pointer_mode mode)
{
    if(mode != aoclsparse_mi355_pointer_auto && mode != aoclsparse_mi355_pointer_host
       && mode != aoclsparse_mi355_pointer_device)
        return aoclsparse_status_invalid_value;
    Runtime::get().pointer_mode = mode;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_set_stream(void *hip_stream)
{
    aoclsparse_status st = Runtime::get().init();
    if(st != aoclsparse_status_success)
        return st;
    Runtime::get().set_stream(static_cast<hipStream_t>(hip_stream));
    return aoclsparse_status_success;
}

void *aoclsparse_mi355_get_stream(void)
{
    return Runtime::get().stream();
}

aoclsparse_status aoclsparse_mi355_hip_runtime_path(char *path, size_t capacity)
{
    if(!path)
        return aoclsparse_status_invalid_pointer;
    if(capacity == 0)
        return aoclsparse_status_invalid_size;
    Dl_info info;
    if(!dladdr(reinterpret_cast<const void *>(&hipStreamCreate), &info) || !info.dli_fname)
        return aoclsparse_status_internal_error;
    std::snprintf(path, capacity, "%s", info.dli_fname);
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_synchronize(void)
{
    aoclsparse_status st = Runtime::get().init();
    if(st != aoclsparse_status_success)
        return st;
    MI355_HIP_TRY(hipStreamSynchronize(Runtime::get().stream()));
    return aoclsparse_status_success;
}

aoclsparse_status
    aoclsparse_mi355_device_info(aoclsparse_int *device, aoclsparse_int *compute_units, char name[256])
{
    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    if(device)
        *device = rt.device;
    if(compute_units)
        *compute_units = rt.cus;
    if(name)
        std::memcpy(name, rt.name, 256);
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_timer_start(void)
{
    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    MI355_HIP_TRY(hipEventRecord(rt.ev0, rt.stream()));
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_timer_stop(float *elapsed_ms)
{
    Runtime &rt = Runtime::get();
    if(!elapsed_ms)
        return aoclsparse_status_invalid_pointer;
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    MI355_HIP_TRY(hipEventRecord(rt.ev1, rt.stream()));
    MI355_HIP_TRY(hipEventSynchronize(rt.ev1));
    MI355_HIP_TRY(hipEventElapsedTime(elapsed_ms, rt.ev0, rt.ev1));
    return aoclsparse_status_success;
}

// Per-iteration timing (the reference harness reports min / quartiles / max of its iterations,
// tests/include/aoclsparse_stats.hpp:41-129): mark() records one event on the library's stream; laps() waits for
// the last one and returns the elapsed time between consecutive marks.
aoclsparse_status aoclsparse_mi355_timer_mark(void)
{
    Runtime          &rt = Runtime::get();
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    if(rt.marks_used == rt.marks.size())
    {
        if(rt.marks.size() >= 65536)
            return aoclsparse_status_invalid_size;
        hipEvent_t e;
        MI355_HIP_TRY(hipEventCreate(&e));
        rt.marks.push_back(e);
    }
    MI355_HIP_TRY(hipEventRecord(rt.marks[rt.marks_used], rt.stream()));
    rt.marks_used++;
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_mi355_timer_laps(float *laps_ms, aoclsparse_int capacity, aoclsparse_int *count)
{
    Runtime &rt = Runtime::get();
    if(!count || (capacity > 0 && !laps_ms))
        return aoclsparse_status_invalid_pointer;
    aoclsparse_status st = rt.init();
    if(st != aoclsparse_status_success)
        return st;
    const size_t used = rt.marks_used;
    rt.marks_used     = 0; // the ring is reset whatever happens below
    *count            = used > 0 ? (aoclsparse_int)(used - 1) : 0;
    if(used < 2)
        return aoclsparse_status_success;
    MI355_HIP_TRY(hipEventSynchronize(rt.marks[used - 1]));
    for(size_t i = 0; i + 1 < used && (aoclsparse_int)i < capacity; i++)
        MI355_HIP_TRY(hipEventElapsedTime(&laps_ms[i], rt.marks[i], rt.marks[i + 1]));
    return aoclsparse_status_success;
}

// ---- version / context shims (library/src/extra/aoclsparse_auxiliary.cpp:56-173, 1433) --------
const char *aoclsparse_get_version(void)
{
    return "AOCL-Sparse 5.3.2 compatible; MI355X-native engine r2 (gfx950)";
}

// The ISA preference selects CPU kernels in the reference; here it is accepted and recorded so
// callers that set it keep working.  Valid tokens: context.hpp:182-213.
static char g_isa[16] = "ENV";

aoclsparse_status aoclsparse_enable_instructions(const char isa_preference[])
{
    if(!isa_preference)
        return aoclsparse_status_invalid_pointer;
    static const char *ok[] = {"ENV", "GENERIC", "AVX2", "AVX512", ""};
    char               up[16];
    size_t             n = std::strlen(isa_preference);
    if(n >= sizeof(up))
        return aoclsparse_status_invalid_value;
    for(size_t i = 0; i <= n; i++)
        up[i] = (char)std::toupper((unsigned char)isa_preference[i]);
    for(const char *t : ok)
        if(std::strcmp(t, up) == 0)
        {
            std::strcpy(g_isa, up[0] ? up : "ENV");
            return aoclsparse_status_success;
        }
    return aoclsparse_status_invalid_value;
}

aoclsparse_status aoclsparse_debug_get(char            isa_preference[],
                                       aoclsparse_int *num_threads,
                                       char            tl_isa_preference[],
                                       bool           *is_isa_updated,
                                       char            arch[])
{
    if(isa_preference)
        std::strcpy(isa_preference, g_isa);
    if(tl_isa_preference)
        std::strcpy(tl_isa_preference, g_isa);
    if(num_threads)
        *num_threads = 1;
    if(is_isa_updated)
        *is_isa_updated = false;
    if(arch)
        std::strcpy(arch, "GFX950");
    return aoclsparse_status_success;
}

aoclsparse_int aoclsparse_is_avx512_build(void)
{
    return 0;
}

} // extern "C"
