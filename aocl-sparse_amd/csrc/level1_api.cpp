// level1_api.cpp -- aoclsparse_?axpyi, ?doti / ?dotci / ?dotui, ?gthr / ?gthrz / ?gthrs, ?sctr / ?sctrs, ?roti and
// their _kid twins.
//
// Checks follow the reference's dispatchers in their order: level1/aoclsparse_axpyi.hpp:56-85, aoclsparse_dot.hpp:64-86,
// aoclsparse_gthr.hpp:69-105, aoclsparse_sctr.hpp:59-90, aoclsparse_roti.hpp:61-84; kid selects among four CPU
// kernels there (0 reference, 1-2 AVX2, 3 AVX-512) -- here it is validated (kid > 3: invalid_kid) and otherwise
// ignored, one GPU kernel serves all.  Vectors may be host or device memory.  For host vectors the dense vector's
// extent is not an argument of these routines, so it is taken from the indices (max + 1) and that prefix is staged;
// the reference kernel's negative-index check (invalid_index_value, axpyi.hpp:44, gthr.hpp:50, sctr.hpp:47,
// roti.hpp:47) is made on the host indices BEFORE anything is modified.  Device-resident indices are not inspected
// (as the reference's AVX kernels do not).
#include "internal.hpp"

#include <algorithm>
#include <cstring>
#include <type_traits>

using namespace mi355;

namespace
{

// one call's staging state: slot k of the runtime's scratch holds array k when it came from the host
struct Stage
{
    Runtime                               &rt;
    std::unique_lock<std::recursive_mutex> sl;
    hipStream_t                            s = nullptr;
    aoclsparse_status                      st;
    struct Back
    {
        void  *host;
        void  *dev;
        size_t bytes;
    } back[3];
    int nback = 0;

    Stage()
        : rt(Runtime::get())
        , sl(rt.stage_lock, std::defer_lock)
    {
        st = rt.init();
        if(st == aoclsparse_status_success)
        {
            if(rt.pointer_mode != aoclsparse_mi355_pointer_device)
                sl.lock();
            s = rt.stream();
        }
    }
    // device view of `p` (bytes long); host arrays are copied in when `in`, copied back at finish() when `out`
    void *view(int slot, const void *p, size_t bytes, bool in, bool out)
    {
        if(st != aoclsparse_status_success)
            return nullptr;
        if(rt.is_device_pointer(p))
            return const_cast<void *>(p);
        void *d = nullptr;
        st      = rt.staging(slot, bytes ? bytes : 1, &d);
        if(st != aoclsparse_status_success)
            return nullptr;
        if(in && bytes && hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s) != hipSuccess)
            st = aoclsparse_status_internal_error;
        if(out)
            back[nback++] = {const_cast<void *>(p), d, bytes};
        return d;
    }
    aoclsparse_status finish()
    {
        if(st != aoclsparse_status_success)
            return st;
        for(int i = 0; i < nback; i++)
            if(back[i].bytes && hipMemcpyAsync(back[i].host, back[i].dev, back[i].bytes, hipMemcpyDeviceToHost, s) != hipSuccess)
                return aoclsparse_status_internal_error;
        if(nback && hipStreamSynchronize(s) != hipSuccess)
            return aoclsparse_status_internal_error;
        return aoclsparse_status_success;
    }
};

// number of elements of y the entries reach; -1: a negative index; host indices only (device: 0 = "not needed")
long long extent_of(Runtime &rt, const aoclsparse_int *indx, aoclsparse_int nnz)
{
    if(rt.is_device_pointer(indx))
        return 0;
    aoclsparse_int mx = -1;
    for(aoclsparse_int i = 0; i < nnz; i++)
    {
        if(indx[i] < 0)
            return -1;
        mx = std::max(mx, indx[i]);
    }
    return (long long)mx + 1;
}

template <typename T>
aoclsparse_status axpyi_t(aoclsparse_int nnz, T a, const T *x, const aoclsparse_int *indx, T *y, aoclsparse_int kid)
{
    if(!x || !indx || !y)
        return aoclsparse_status_invalid_pointer;
    if(nnz == 0)
        return aoclsparse_status_success;
    if(nnz < 0)
        return aoclsparse_status_invalid_size;
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    Stage g;
    if(g.st != aoclsparse_status_success)
        return g.st;
    const long long ext = extent_of(g.rt, indx, nnz);
    if(ext < 0)
        return aoclsparse_status_invalid_index_value;
    if(ext == 0 && !g.rt.is_device_pointer(y))
        return aoclsparse_status_invalid_value; // host y with device indices: its extent is unknown
    const T              *dx = static_cast<const T *>(g.view(0, x, sizeof(T) * (size_t)nnz, true, false));
    const aoclsparse_int *di = static_cast<const aoclsparse_int *>(g.view(1, indx, sizeof(aoclsparse_int) * (size_t)nnz, true, false));
    T                    *dy = static_cast<T *>(g.view(2, y, sizeof(T) * (size_t)ext, true, true));
    if(g.st == aoclsparse_status_success)
        g.st = launch_axpyi<T>(g.s, nnz, a, dx, di, dy);
    return g.finish();
}

template <typename T>
aoclsparse_status dot_t(aoclsparse_int nnz, const T *x, const aoclsparse_int *indx, const T *y, T *dot, bool conj,
                        aoclsparse_int kid)
{
    if(!dot)
        return aoclsparse_status_invalid_pointer;
    if(nnz <= 0)
    {
        std::memset(dot, 0, sizeof(T)); // dot.hpp:78-82
        return aoclsparse_status_invalid_size;
    }
    if(!x || !indx || !y)
        return aoclsparse_status_invalid_pointer;
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    Stage g;
    if(g.st != aoclsparse_status_success)
        return g.st;
    const long long ext = extent_of(g.rt, indx, nnz);
    if(ext < 0)
        return aoclsparse_status_invalid_index_value;
    if(ext == 0 && !g.rt.is_device_pointer(y))
        return aoclsparse_status_invalid_value;
    const T              *dx = static_cast<const T *>(g.view(0, x, sizeof(T) * (size_t)nnz, true, false));
    const aoclsparse_int *di = static_cast<const aoclsparse_int *>(g.view(1, indx, sizeof(aoclsparse_int) * (size_t)nnz, true, false));
    const T              *dy = static_cast<const T *>(g.view(2, y, sizeof(T) * (size_t)ext, true, false));
    void                 *scratch = nullptr;
    if(g.st == aoclsparse_status_success)
        g.st = g.rt.staging(3, sizeof(T) * (size_t)(L1_DOT_PARTIALS + 1), &scratch);
    if(g.st != aoclsparse_status_success)
        return g.st;
    T *part = static_cast<T *>(scratch), *out = part + L1_DOT_PARTIALS;
    g.st    = launch_doti<T>(g.s, nnz, dx, di, dy, conj, part, out);
    if(g.st != aoclsparse_status_success)
        return g.st;
    // the result is a host scalar in the reference's interface (returned by value for the real types)
    MI355_HIP_TRY(hipMemcpyAsync(dot, out, sizeof(T), hipMemcpyDeviceToHost, g.s));
    MI355_HIP_TRY(hipStreamSynchronize(g.s));
    return aoclsparse_status_success;
}

// mode 0 gthr, 1 gthrz, 2 sctr; indexed (indx) or strided (indx == nullptr, stride)
template <typename T>
aoclsparse_status move_t(int mode, aoclsparse_int nnz, T *y, T *x, const aoclsparse_int *indx, bool strided,
                         aoclsparse_int stride, aoclsparse_int kid)
{
    if(mode == 2)
    {
        // sctr.hpp:65-88
        if(!x || !y)
            return aoclsparse_status_invalid_pointer;
        if(nnz == 0)
            return aoclsparse_status_success;
        if(nnz < 0)
            return aoclsparse_status_invalid_size;
        if(strided ? stride <= 0 : false)
            return aoclsparse_status_invalid_size;
        if(!strided && !indx)
            return aoclsparse_status_invalid_pointer;
    }
    else
    {
        // gthr.hpp:73-104
        if(nnz < 0)
            return aoclsparse_status_invalid_size;
        if(nnz == 0)
            return aoclsparse_status_success;
        if(!y || !x)
            return aoclsparse_status_invalid_pointer;
        if(strided && stride < 0)
            return aoclsparse_status_invalid_size;
        if(!strided && !indx)
            return aoclsparse_status_invalid_pointer;
    }
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    Stage g;
    if(g.st != aoclsparse_status_success)
        return g.st;
    long long ext = strided ? (long long)stride * (nnz - 1) + 1 : extent_of(g.rt, indx, nnz);
    if(ext < 0)
        return aoclsparse_status_invalid_index_value;
    if(ext == 0 && !g.rt.is_device_pointer(y))
        return aoclsparse_status_invalid_value;
    const bool            y_in = true; // scatter keeps the untouched elements, gather reads
    T                    *dx = static_cast<T *>(g.view(0, x, sizeof(T) * (size_t)nnz, mode == 2, mode != 2));
    const aoclsparse_int *di = strided ? nullptr
                                       : static_cast<const aoclsparse_int *>(
                                           g.view(1, indx, sizeof(aoclsparse_int) * (size_t)nnz, true, false));
    T                    *dy = static_cast<T *>(g.view(2, y, sizeof(T) * (size_t)ext, y_in, mode != 0));
    if(g.st == aoclsparse_status_success)
        g.st = launch_gather_scatter<T>(g.s, nnz, dx, di, (long long)stride, dy, mode);
    return g.finish();
}

template <typename T>
aoclsparse_status roti_t(aoclsparse_int nnz, T *x, const aoclsparse_int *indx, T *y, T c, T s, aoclsparse_int kid)
{
    if(!x || !indx || !y)
        return aoclsparse_status_invalid_pointer;
    if(nnz == 0)
        return aoclsparse_status_success;
    if(nnz < 0)
        return aoclsparse_status_invalid_size;
    if(kid > 3)
        return aoclsparse_status_invalid_kid;
    Stage g;
    if(g.st != aoclsparse_status_success)
        return g.st;
    const long long ext = extent_of(g.rt, indx, nnz);
    if(ext < 0)
        return aoclsparse_status_invalid_index_value;
    if(ext == 0 && !g.rt.is_device_pointer(y))
        return aoclsparse_status_invalid_value;
    T                    *dx = static_cast<T *>(g.view(0, x, sizeof(T) * (size_t)nnz, true, true));
    const aoclsparse_int *di = static_cast<const aoclsparse_int *>(g.view(1, indx, sizeof(aoclsparse_int) * (size_t)nnz, true, false));
    T                    *dy = static_cast<T *>(g.view(2, y, sizeof(T) * (size_t)ext, true, true));
    if(g.st == aoclsparse_status_success)
        g.st = launch_roti<T>(g.s, nnz, dx, di, dy, c, s);
    return g.finish();
}

template <typename T>
T *vp(void *p)
{
    return static_cast<T *>(p);
}
template <typename T>
const T *vp(const void *p)
{
    return static_cast<const T *>(p);
}

} // namespace

extern "C" {

// ---- real types --------------------------------------------------------------------------------------------------
#define MI355_L1_REAL(P, T)                                                                                              \
    aoclsparse_status aoclsparse_##P##axpyi(const aoclsparse_int nnz, const T a, const T *x, const aoclsparse_int *indx, \
                                            T *y)                                                                        \
    {                                                                                                                    \
        return axpyi_t<T>(nnz, a, x, indx, y, -1);                                                                       \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##axpyi_kid(const aoclsparse_int nnz, const T a, const T *x,                         \
                                                const aoclsparse_int *indx, T *y, aoclsparse_int kid)                    \
    {                                                                                                                    \
        return axpyi_t<T>(nnz, a, x, indx, y, kid);                                                                      \
    }                                                                                                                    \
    T aoclsparse_##P##doti(const aoclsparse_int nnz, const T *x, const aoclsparse_int *indx, const T *y)                 \
    {                                                                                                                    \
        T dot = 0;                                                                                                       \
        dot_t<T>(nnz, x, indx, y, &dot, false, -1);                                                                      \
        return dot;                                                                                                      \
    }                                                                                                                    \
    T aoclsparse_##P##doti_kid(const aoclsparse_int nnz, const T *x, const aoclsparse_int *indx, const T *y,             \
                               aoclsparse_int kid)                                                                       \
    {                                                                                                                    \
        T dot = 0;                                                                                                       \
        dot_t<T>(nnz, x, indx, y, &dot, false, kid);                                                                     \
        return dot;                                                                                                      \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthr(aoclsparse_int nnz, const T *y, T *x, const aoclsparse_int *indx)             \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(y), x, indx, false, 0, -1);                                             \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthr_kid(aoclsparse_int nnz, const T *y, T *x, const aoclsparse_int *indx,         \
                                               aoclsparse_int kid)                                                       \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(y), x, indx, false, 0, kid);                                            \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrz(aoclsparse_int nnz, T *y, T *x, const aoclsparse_int *indx)                  \
    {                                                                                                                    \
        return move_t<T>(1, nnz, y, x, indx, false, 0, -1);                                                              \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrz_kid(aoclsparse_int nnz, T *y, T *x, const aoclsparse_int *indx,              \
                                                aoclsparse_int kid)                                                      \
    {                                                                                                                    \
        return move_t<T>(1, nnz, y, x, indx, false, 0, kid);                                                             \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrs(aoclsparse_int nnz, const T *y, T *x, aoclsparse_int stride)                 \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(y), x, nullptr, true, stride, -1);                                      \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrs_kid(aoclsparse_int nnz, const T *y, T *x, aoclsparse_int stride,             \
                                                aoclsparse_int kid)                                                      \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(y), x, nullptr, true, stride, kid);                                     \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctr(const aoclsparse_int nnz, const T *x, const aoclsparse_int *indx, T *y)       \
    {                                                                                                                    \
        return move_t<T>(2, nnz, y, const_cast<T *>(x), indx, false, 0, -1);                                             \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctr_kid(const aoclsparse_int nnz, const T *x, const aoclsparse_int *indx, T *y,   \
                                               aoclsparse_int kid)                                                       \
    {                                                                                                                    \
        return move_t<T>(2, nnz, y, const_cast<T *>(x), indx, false, 0, kid);                                            \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctrs(const aoclsparse_int nnz, const T *x, aoclsparse_int stride, T *y)           \
    {                                                                                                                    \
        return move_t<T>(2, nnz, y, const_cast<T *>(x), nullptr, true, stride, -1);                                      \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctrs_kid(const aoclsparse_int nnz, const T *x, aoclsparse_int stride, T *y,       \
                                                aoclsparse_int kid)                                                      \
    {                                                                                                                    \
        return move_t<T>(2, nnz, y, const_cast<T *>(x), nullptr, true, stride, kid);                                     \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##roti(const aoclsparse_int nnz, T *x, const aoclsparse_int *indx, T *y, const T c,  \
                                           const T s)                                                                    \
    {                                                                                                                    \
        return roti_t<T>(nnz, x, indx, y, c, s, -1);                                                                     \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##roti_kid(const aoclsparse_int nnz, T *x, const aoclsparse_int *indx, T *y,         \
                                               const T c, const T s, aoclsparse_int kid)                                 \
    {                                                                                                                    \
        return roti_t<T>(nnz, x, indx, y, c, s, kid);                                                                    \
    }
MI355_L1_REAL(d, double)
MI355_L1_REAL(s, float)

// ---- complex types: void pointers, the scalar of axpyi by pointer (aoclsparse_functions.h:83-88) --------------------
#define MI355_L1_CPLX(P, T)                                                                                              \
    aoclsparse_status aoclsparse_##P##axpyi(const aoclsparse_int nnz, const void *a, const void *x,                      \
                                            const aoclsparse_int *indx, void *y)                                         \
    {                                                                                                                    \
        if(!a)                                                                                                           \
            return aoclsparse_status_invalid_pointer;                                                                    \
        return axpyi_t<T>(nnz, *vp<T>(a), vp<T>(x), indx, vp<T>(y), -1);                                                 \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##axpyi_kid(const aoclsparse_int nnz, const void *a, const void *x,                  \
                                                const aoclsparse_int *indx, void *y, aoclsparse_int kid)                 \
    {                                                                                                                    \
        if(!a)                                                                                                           \
            return aoclsparse_status_invalid_pointer;                                                                    \
        return axpyi_t<T>(nnz, *vp<T>(a), vp<T>(x), indx, vp<T>(y), kid);                                                \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##dotci(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx,         \
                                            const void *y, void *dot)                                                    \
    {                                                                                                                    \
        return dot_t<T>(nnz, vp<T>(x), indx, vp<T>(y), vp<T>(dot), true, -1);                                            \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##dotci_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx,     \
                                                const void *y, void *dot, aoclsparse_int kid)                            \
    {                                                                                                                    \
        return dot_t<T>(nnz, vp<T>(x), indx, vp<T>(y), vp<T>(dot), true, kid);                                           \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##dotui(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx,         \
                                            const void *y, void *dot)                                                    \
    {                                                                                                                    \
        return dot_t<T>(nnz, vp<T>(x), indx, vp<T>(y), vp<T>(dot), false, -1);                                           \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##dotui_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx,     \
                                                const void *y, void *dot, aoclsparse_int kid)                            \
    {                                                                                                                    \
        return dot_t<T>(nnz, vp<T>(x), indx, vp<T>(y), vp<T>(dot), false, kid);                                          \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthr(aoclsparse_int nnz, const void *y, void *x, const aoclsparse_int *indx)       \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(vp<T>(y)), vp<T>(x), indx, false, 0, -1);                               \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthr_kid(aoclsparse_int nnz, const void *y, void *x, const aoclsparse_int *indx,   \
                                               aoclsparse_int kid)                                                       \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(vp<T>(y)), vp<T>(x), indx, false, 0, kid);                              \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrz(aoclsparse_int nnz, void *y, void *x, const aoclsparse_int *indx)            \
    {                                                                                                                    \
        return move_t<T>(1, nnz, vp<T>(y), vp<T>(x), indx, false, 0, -1);                                                \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrz_kid(aoclsparse_int nnz, void *y, void *x, const aoclsparse_int *indx,        \
                                                aoclsparse_int kid)                                                      \
    {                                                                                                                    \
        return move_t<T>(1, nnz, vp<T>(y), vp<T>(x), indx, false, 0, kid);                                               \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrs(aoclsparse_int nnz, const void *y, void *x, aoclsparse_int stride)           \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(vp<T>(y)), vp<T>(x), nullptr, true, stride, -1);                        \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##gthrs_kid(aoclsparse_int nnz, const void *y, void *x, aoclsparse_int stride,       \
                                                aoclsparse_int kid)                                                      \
    {                                                                                                                    \
        return move_t<T>(0, nnz, const_cast<T *>(vp<T>(y)), vp<T>(x), nullptr, true, stride, kid);                       \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctr(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx, void *y) \
    {                                                                                                                    \
        return move_t<T>(2, nnz, vp<T>(y), const_cast<T *>(vp<T>(x)), indx, false, 0, -1);                               \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctr_kid(const aoclsparse_int nnz, const void *x, const aoclsparse_int *indx,      \
                                               void *y, aoclsparse_int kid)                                              \
    {                                                                                                                    \
        return move_t<T>(2, nnz, vp<T>(y), const_cast<T *>(vp<T>(x)), indx, false, 0, kid);                              \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctrs(const aoclsparse_int nnz, const void *x, aoclsparse_int stride, void *y)     \
    {                                                                                                                    \
        return move_t<T>(2, nnz, vp<T>(y), const_cast<T *>(vp<T>(x)), nullptr, true, stride, -1);                        \
    }                                                                                                                    \
    aoclsparse_status aoclsparse_##P##sctrs_kid(const aoclsparse_int nnz, const void *x, aoclsparse_int stride, void *y, \
                                                aoclsparse_int kid)                                                      \
    {                                                                                                                    \
        return move_t<T>(2, nnz, vp<T>(y), const_cast<T *>(vp<T>(x)), nullptr, true, stride, kid);                       \
    }
MI355_L1_CPLX(z, cdouble)
MI355_L1_CPLX(c, cfloat)

} // extern "C"
