// itsol_api.cpp -- aoclsparse_itsol_*: Conjugate Gradient and restarted GMRES with the iterates resident
// on the GPU.
//
// Reference: solvers/aoclsparse_itsol_functions.hpp (state machines :632-875 CG, :910-1367 GMRES, drivers
// :1369-1619, option parsing :1621-1716), solvers/aoclsparse_itsol_functions.cpp (C wrappers),
// solvers/aoclsparse_itsol_list_options.hpp (option names, bounds, defaults).  The reference runs the vector
// steps with AOCL-BLAS level-1 routines and libFLAME's lartg (neither vendored, no version pinned,
// cmake/Dependencies.cmake:93-136); here they are the kernels of itsol_kernels.hip and a restatement of
// LAPACK 3.10's dlartg.  Summation orders of dot / nrm2 therefore differ from the CPU library's: parity is
// on iteration counts, exit codes, rinfo and the solution within the solver tolerances.
//
// Two interfaces, as in the reference:
//  * direct (aoclsparse_itsol_?_solve): b and x are staged to HBM once (or used in place when they are device
//    pointers), every vector of the method lives in HBM, SpMV / SymGS / ILU(0) steps are this library's
//    executors; one scalar read-back per reduction.  User callbacks (precond, monit) are host functions: their
//    arguments are copied to host memory for the call.
//  * reverse communication (aoclsparse_itsol_?_rci_input / _rci_solve): the caller performs v = A u and the
//    preconditioner on the pointers handed out.  With a device-pointer b the workspaces are HBM and u / v are
//    device pointers; with a host b they are pinned host allocations the GPU addresses over PCIe, so that a
//    host caller can dereference them while the vector arithmetic still runs in HIP kernels.
#include "internal.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <complex>
#include <string>
#include <vector>

using namespace mi355;

namespace
{

#define MI355_TRY(expr)                       \
    do                                        \
    {                                         \
        aoclsparse_status st__ = (expr);      \
        if(st__ != aoclsparse_status_success) \
            return st__;                      \
    } while(0)

constexpr int RINFO_RES_NORM = 0, RINFO_RHS_NORM = 1, RINFO_ITER = 30; // itsol_functions.hpp:36-38
enum
{
    solver_cg = 1,
    solver_gmres
};
enum cg_stage
{
    CG_ENTRY = 0,
    CG_RESIDUAL,
    CG_CHECK,
    CG_PRECOND_Z,
    CG_DIRECTION,
    CG_UPDATE_X
};
enum gmres_stage
{
    GM_ENTRY = 0,
    GM_RESIDUAL,
    GM_PRECOND_R0,
    GM_ARNOLDI_STEP,
    GM_AFTER_PRECOND,
    GM_UPDATE_X,
    GM_RESTART,
    GM_CHECK
};

// ---- options (include/aoclsparse_itsol_options.hpp; names / bounds / defaults: list_options.hpp:92-237) ----
std::string prepare(std::string s) // trim, squeeze blanks, lower case (options.hpp:104-114)
{
    std::string o;
    bool        blank = true;
    for(char ch : s)
    {
        if(std::isspace((unsigned char)ch))
        {
            if(!blank)
                o.push_back(' ');
            blank = true;
        }
        else
        {
            o.push_back((char)std::tolower((unsigned char)ch));
            blank = false;
        }
    }
    while(!o.empty() && o.back() == ' ')
        o.pop_back();
    return o;
}

struct Option
{
    int                        kind = 0; // 1 integer, 2 real, 3 string
    long long                  ival = 0;
    double                     rval = 0;
    std::string                sval;
    int                        key = 0; // label value of a string option
    std::map<std::string, int> labels;
    std::string                desc;
    bool                       user = false;
};

struct Options
{
    std::map<std::string, Option> reg;
    bool                          locked = false;
    void add_int(const char *n, const char *d, long long v)
    {
        Option o;
        o.kind = 1, o.ival = v, o.desc = d;
        reg[n] = o;
    }
    void add_real(const char *n, const char *d, double v)
    {
        Option o;
        o.kind = 2, o.rval = v, o.desc = d;
        reg[n] = o;
    }
    void add_str(const char *n, const char *d, std::map<std::string, int> labels, const char *v)
    {
        Option o;
        o.kind = 3, o.labels = std::move(labels), o.sval = prepare(v), o.desc = d;
        o.key  = o.labels[o.sval];
        reg[n] = o;
    }
    // handle_parse_option + SetOption: unknown name, out of range, bad label, locked -> invalid_value
    aoclsparse_status set(const char *name, const char *value)
    {
        if(!name || !value)
            return aoclsparse_status_invalid_pointer;
        auto it = reg.find(prepare(name));
        if(it == reg.end())
            return aoclsparse_status_invalid_value;
        Option &o = it->second;
        try
        {
            if(o.kind == 1)
            {
                const int v = std::stoi(value);
                if(locked || v < 1) // every integer option: 1 <= v
                    return aoclsparse_status_invalid_value;
                o.ival = v;
            }
            else if(o.kind == 2)
            {
                const double v = std::stod(value);
                if(locked || !(v >= 0.0)) // every real option: 0 <= v
                    return aoclsparse_status_invalid_value;
                o.rval = v;
            }
            else
            {
                const std::string v = prepare(value);
                auto              l = o.labels.find(v);
                if(locked || l == o.labels.end())
                    return aoclsparse_status_invalid_value;
                o.sval = v, o.key = l->second;
            }
        }
        catch(const std::exception &)
        {
            return aoclsparse_status_invalid_value; // not a number (the reference lets std::stoi throw)
        }
        o.user = true;
        return aoclsparse_status_success;
    }
    void print() const
    {
        std::printf("Begin Options\n");
        for(const auto &kv : reg)
        {
            const Option &o = kv.second;
            if(o.kind == 1)
                std::printf("   %s = %lld\n", kv.first.c_str(), o.ival);
            else if(o.kind == 2)
                std::printf("   %s = %.6e\n", kv.first.c_str(), o.rval);
            else
                std::printf("   %s = %s\n", kv.first.c_str(), o.sval.c_str());
        }
        std::printf("End Options\n");
    }
};

template <typename T>
void register_options(Options &o)
{
    // expected_precision(scale) = scale * safeguard * sqrt(2 eps), safeguard 1 (double) / 2 (float):
    // extra/aoclsparse_utils.hpp:557-580
    const double guard = sizeof(T) == 8 ? 1.0 : 2.0;
    const double prec  = guard * std::sqrt(2.0 * (double)std::numeric_limits<T>::epsilon());
    o.add_str("iterative method", "Choose solver to use",
              {{"cg", solver_cg}, {"pcg", solver_cg}, {"gmres", solver_gmres}, {"gm res", solver_gmres}}, "CG");
    o.add_int("cg iteration limit", "Set CG iteration limit", 500);
    o.add_real("cg rel tolerance", "Set relative convergence tolerance for cg method", 2.0 * prec);
    o.add_real("cg abs tolerance", "Set absolute convergence tolerance for cg method", prec);
    o.add_str("cg preconditioner", "Choose preconditioner to use with cg method",
              {{"none", 0}, {"user", 1}, {"gs", 3}, {"symgs", 3}, {"sgs", 3}}, "None");
    o.add_int("gmres iteration limit", "Set GMRES iteration limit", 150);
    o.add_real("gmres rel tolerance", "Set relative convergence tolerance for gmres method", 2.0 * prec);
    o.add_real("gmres abs tolerance", "Set absolute convergence tolerance for gmres method", prec);
    o.add_str("gmres preconditioner", "Choose preconditioner to use with gmres method",
              {{"none", 0}, {"user", 1}, {"ilu0", 2}}, "None");
    o.add_int("gmres restart iterations", "Set GMRES restart iterations", 20);
}

// ---- workspaces ----------------------------------------------------------------------------------------
struct VBuf
{
    void  *ptr   = nullptr;
    size_t bytes = 0;
    bool   pinned = false;
    VBuf()        = default;
    VBuf(const VBuf &)            = delete;
    VBuf &operator=(const VBuf &) = delete;
    ~VBuf()
    {
        release();
    }
    void release()
    {
        if(ptr)
            (void)(pinned ? hipHostFree(ptr) : hipFree(ptr));
        ptr = nullptr, bytes = 0;
    }
    aoclsparse_status alloc(size_t nbytes, bool pin)
    {
        if(ptr && nbytes <= bytes && pin == pinned)
            return aoclsparse_status_success;
        release();
        const size_t want = nbytes ? nbytes : 8;
        pinned            = pin;
        if(pin)
            MI355_HIP_TRY(hipHostMalloc(&ptr, want, hipHostMallocDefault));
        else
            MI355_HIP_TRY(hipMalloc(&ptr, want));
        bytes = want;
        return aoclsparse_status_success;
    }
    template <typename T>
    T *as() const
    {
        return static_cast<T *>(ptr);
    }
};

template <typename T>
bool near_zero(T v)
{
    return std::fabs(v) <= T(1e-2) * T(2) * std::numeric_limits<T>::epsilon(); // utils.hpp:598-613
}
template <typename T>
bool negative_or_near_zero(T v)
{
    return v <= T(1e-2) * T(2) * std::numeric_limits<T>::epsilon(); // utils.hpp:626-638
}

// LAPACK 3.10 dlartg / slartg (la_lartg.f90), the routine behind libflame::lartg (:1148)
template <typename T>
void lartg(T f, T g, T &c, T &s, T &r)
{
    const T safmin = std::numeric_limits<T>::min(), safmax = T(1) / safmin;
    const T rtmin = std::sqrt(safmin), rtmax = std::sqrt(safmax / 2);
    const T f1 = std::fabs(f), g1 = std::fabs(g);
    if(g == T(0))
        c = T(1), s = T(0), r = f;
    else if(f == T(0))
        c = T(0), s = std::copysign(T(1), g), r = g1;
    else if(f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax)
    {
        const T d = std::sqrt(f * f + g * g);
        c = f1 / d, r = std::copysign(d, f), s = g / r;
    }
    else
    {
        const T u = std::min(safmax, std::max(safmin, std::max(f1, g1)));
        const T fs = f / u, gs = g / u, d = std::sqrt(fs * fs + gs * gs);
        c = std::fabs(fs) / d, r = std::copysign(d, f), s = gs / r, r = r * u;
    }
}

template <typename T>
struct Solver
{
    aoclsparse_int n = 0;
    bool           have_b = false, pinned = false, solving = false;
    int            method = solver_cg;
    Options        opts;
    VBuf           b, xshadow, red_partial, red_out, coef;
    // CG (cg_data, aoclsparse_itsol_data.hpp:96-111)
    VBuf           r, z, p, q, y;
    T              alpha = 0, rz = 0, beta = 0, rnorm2 = 0, bnorm2 = 0, brtol = 0, rtol = 0, atol = 0;
    T              rr_last = 0; // r.r of the current residual (what z.r is when there is no preconditioner)
    int            stage = CG_ENTRY;
    aoclsparse_int niter = 0, maxit = 0;
    int            precond = 0;
    // GMRES (gmres_data, :127-143)
    VBuf              v, zz;
    std::vector<T>    h, g, s, c;
    aoclsparse_int    j = 0, restart = 0;
    bool              x_dirty = false;

    void free_solver_data() // aoclsparse_itsol_data_free(itsol, true)
    {
        r.release(), z.release(), p.release(), q.release(), y.release(), v.release(), zz.release();
        h.clear(), g.clear(), s.clear(), c.clear();
        stage = CG_ENTRY, niter = 0, j = 0;
    }

    // ---- reductions: result read back through a pinned-free plain copy ----
    aoclsparse_status dots(Runtime &rt, int k, const T *V, long long ld, const T *w, T *out_host)
    {
        MI355_TRY(red_partial.alloc(sizeof(T) * (size_t)vec_reduce_scratch_elems(k), false));
        MI355_TRY(red_out.alloc(sizeof(T) * (size_t)std::max(k, 1), false));
        MI355_TRY(launch_multidot<T>(rt.stream(), n, k, V, ld, w, red_partial.as<T>(), red_out.as<T>()));
        MI355_HIP_TRY(hipMemcpyAsync(out_host, red_out.ptr, sizeof(T) * (size_t)k, hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        return aoclsparse_status_success;
    }
    aoclsparse_status nrm2(Runtime &rt, const T *a, T &out)
    {
        T d = 0;
        MI355_TRY(dots(rt, 1, a, 0, a, &d));
        out = std::sqrt(d);
        return aoclsparse_status_success;
    }

    aoclsparse_status init() // aoclsparse_itsol_solver_init (:335-384)
    {
        method = opts.reg["iterative method"].key;
        if(method == solver_cg)
        {
            const size_t nb = sizeof(T) * (size_t)n;
            MI355_TRY(r.alloc(nb, pinned));
            MI355_TRY(z.alloc(nb, pinned));
            MI355_TRY(p.alloc(nb, pinned));
            MI355_TRY(q.alloc(nb, pinned));
            stage    = CG_ENTRY;
            precond = opts.reg["cg preconditioner"].key;
            rtol    = (T)opts.reg["cg rel tolerance"].rval;
            atol    = (T)opts.reg["cg abs tolerance"].rval;
            maxit   = (aoclsparse_int)opts.reg["cg iteration limit"].ival;
        }
        else
        {
            if(v.ptr == nullptr)
            {
                restart           = (aoclsparse_int)opts.reg["gmres restart iterations"].ival;
                const long long m = restart;
                if((m + 1) * (long long)n > std::numeric_limits<aoclsparse_int>::max()
                   || m * m > std::numeric_limits<aoclsparse_int>::max())
                    return aoclsparse_status_invalid_size; // :106-112
                const size_t kb = sizeof(T) * (size_t)(m + 1) * (size_t)n;
                MI355_TRY(v.alloc(kb, pinned));
                MI355_TRY(zz.alloc(kb, pinned));
                MI355_HIP_TRY(hipMemset(v.ptr, 0, kb));
                MI355_HIP_TRY(hipMemset(zz.ptr, 0, kb));
                try
                {
                    h.assign((size_t)(m * m), T(0)), g.assign((size_t)m + 1, T(0));
                    c.assign((size_t)m, T(0)), s.assign((size_t)m, T(0));
                }
                catch(const std::bad_alloc &)
                {
                    return aoclsparse_status_memory_error;
                }
                MI355_TRY(coef.alloc(sizeof(T) * (size_t)(m + 1), false));
                niter = 0, j = 0;
            }
            stage    = GM_ENTRY;
            precond = opts.reg["gmres preconditioner"].key;
            rtol    = (T)opts.reg["gmres rel tolerance"].rval;
            atol    = (T)opts.reg["gmres abs tolerance"].rval;
            maxit   = (aoclsparse_int)opts.reg["gmres iteration limit"].ival;
        }
        return aoclsparse_status_success;
    }

    aoclsparse_status upload_coef(Runtime &rt, const T *host, int k)
    {
        MI355_HIP_TRY(hipMemcpyAsync(coef.ptr, host, sizeof(T) * (size_t)k, hipMemcpyHostToDevice, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream())); // host is a stack / vector temporary
        return aoclsparse_status_success;
    }

    // ---- CG, aoclsparse_cg_rci_solve (:632-875) ------------------------------------------------------
    aoclsparse_status cg_step(Runtime &rt, aoclsparse_itsol_rci_job *ircomm, T **u, T **vv, T *x, T *rinfo)
    {
        aoclsparse_status exit_status = aoclsparse_status_success;
        hipStream_t       st          = rt.stream();
        if(stage != CG_ENTRY && *ircomm == aoclsparse_rci_interrupt)
        {
            *ircomm = aoclsparse_rci_stop;
            return aoclsparse_status_user_stop;
        }
        bool loop;
        do
        {
            loop = false;
            switch(stage)
            {
            case CG_ENTRY:
                for(int i = 0; i < 100; i++)
                    rinfo[i] = T(0);
                niter = 0;
                MI355_TRY(launch_cg_init<T>(st, n, b.as<T>(), x, r.as<T>(), p.as<T>()));
                MI355_TRY(nrm2(rt, b.as<T>(), bnorm2));
                if(bnorm2 != bnorm2)
                    return aoclsparse_status_invalid_value;
                rinfo[RINFO_RHS_NORM] = bnorm2;
                brtol                 = rtol * bnorm2;
                *ircomm               = aoclsparse_rci_mv;
                stage                  = CG_RESIDUAL;
                *u = p.as<T>(), *vv = q.as<T>();
                break;
            case CG_RESIDUAL:
                MI355_TRY(launch_vec_add<T>(st, n, q.as<T>(), r.as<T>()));
                MI355_TRY(dots(rt, 1, r.as<T>(), 0, r.as<T>(), &rr_last));
                rnorm2 = std::sqrt(rr_last);
                if(rnorm2 != rnorm2)
                {
                    exit_status = aoclsparse_status_numerical_error;
                    break;
                }
                rinfo[RINFO_RES_NORM] = rnorm2;
                MI355_TRY(launch_vec_fill<T>(st, n, p.as<T>(), T(0)));
                rz   = T(1);
                stage = CG_CHECK;
                [[fallthrough]];
            case CG_CHECK:
                *u = r.as<T>(), *vv = nullptr;
                if(T(0) < atol && rnorm2 <= atol)
                {
                    *ircomm = aoclsparse_rci_stop;
                    break;
                }
                if(T(0) < rtol && rnorm2 <= brtol)
                {
                    *ircomm = aoclsparse_rci_stop;
                    break;
                }
                if(maxit > 0 && niter > maxit)
                {
                    *ircomm     = aoclsparse_rci_stop;
                    exit_status = aoclsparse_status_maxit;
                    break;
                }
                stage    = CG_PRECOND_Z;
                *ircomm = aoclsparse_rci_stopping_criterion;
                break;
            case CG_PRECOND_Z:
                niter++;
                rinfo[RINFO_ITER] = (T)niter;
                stage              = CG_DIRECTION;
                if(precond) // (without one z = r: the copy of :790-797 is skipped, r itself is used)
                {
                    *ircomm = aoclsparse_rci_precond;
                    *u = r.as<T>(), *vv = z.as<T>();
                    break;
                }
                [[fallthrough]];
            case CG_DIRECTION:
            {
                T rz_new = rr_last; // z = r: z.r is the r.r the previous step already reduced (same summation order)
                if(precond)
                    MI355_TRY(dots(rt, 1, z.as<T>(), 0, r.as<T>(), &rz_new));
                if(negative_or_near_zero(rz))
                    return aoclsparse_status_numerical_error;
                beta = rz_new / rz;
                rz   = rz_new;
                MI355_TRY(launch_cg_direction<T>(st, n, beta, p.as<T>(), precond ? z.as<T>() : r.as<T>()));
                *ircomm = aoclsparse_rci_mv;
                stage    = CG_UPDATE_X;
                *u = p.as<T>(), *vv = q.as<T>();
                break;
            }
            case CG_UPDATE_X:
            {
                // p.q, alpha = rz / (p.q) and the step are queued back to back; the host waits once, for r.r and p.q
                MI355_TRY(red_partial.alloc(sizeof(T) * (size_t)vec_reduce_scratch_elems(2), false));
                MI355_TRY(red_out.alloc(sizeof(T) * 2, false));
                MI355_TRY(launch_cg_step_dev<T>(st, n, rz, T(1e-2) * T(2) * std::numeric_limits<T>::epsilon(), p.as<T>(),
                                                q.as<T>(), x, r.as<T>(), red_partial.as<T>(), red_out.as<T>()));
                T two[2] = {0, 0};
                MI355_HIP_TRY(hipMemcpyAsync(two, red_out.ptr, sizeof(T) * 2, hipMemcpyDeviceToHost, st));
                MI355_HIP_TRY(hipStreamSynchronize(st));
                const T rr = two[0], pq = two[1];
                if(negative_or_near_zero(pq) || pq == T(0))
                    return aoclsparse_status_numerical_error; // A is not positive definite (x, r left untouched)
                alpha   = rz / pq;
                rr_last = rr;
                x_dirty = true;
                rnorm2  = std::sqrt(rr);
                if(rnorm2 != rnorm2)
                {
                    exit_status = aoclsparse_status_numerical_error;
                    break;
                }
                rinfo[RINFO_RES_NORM] = rnorm2;
                loop                  = true;
                stage                  = CG_CHECK;
                break;
            }
            default:
                *ircomm = aoclsparse_rci_stop;
                return aoclsparse_status_internal_error;
            }
        } while(loop);
        return exit_status;
    }

    // ---- GMRES, aoclsparse_gmres_rci_solve (:910-1367) ---------------------------------------------------
    aoclsparse_status gmres_step(Runtime &rt, aoclsparse_itsol_rci_job *ircomm, T **io1, T **io2, T *x, T *rinfo)
    {
        aoclsparse_status    exit_status = aoclsparse_status_success;
        hipStream_t          st          = rt.stream();
        const aoclsparse_int m           = restart;
        const long long      ld          = n;
        T                   *V = v.as<T>(), *Z = zz.as<T>();
        if(stage != GM_ENTRY && *ircomm == aoclsparse_rci_interrupt)
        {
            *ircomm = aoclsparse_rci_stop;
            return aoclsparse_status_user_stop;
        }
        bool loop;
        do
        {
            loop = false;
            switch(stage)
            {
            case GM_ENTRY:
                *io1 = x, *io2 = V;
                *ircomm = aoclsparse_rci_mv;
                stage    = GM_RESIDUAL;
                break;
            case GM_RESIDUAL:
            {
                MI355_TRY(nrm2(rt, b.as<T>(), bnorm2));
                if(std::isnan(bnorm2))
                    return aoclsparse_status_invalid_value;
                brtol                 = rtol * bnorm2;
                rinfo[RINFO_RHS_NORM] = brtol; // sic (:1004)
                if(near_zero(atol) && near_zero(brtol))
                {
                    exit_status = aoclsparse_status_invalid_value;
                    *ircomm     = aoclsparse_rci_stop;
                    break;
                }
                MI355_TRY(launch_waxpby<T>(st, n, T(1), b.as<T>(), T(-1), V, V)); // v = b - A x
                T g0 = 0;
                MI355_TRY(nrm2(rt, V, g0));
                g[0]                  = g0;
                rnorm2                = g0;
                rinfo[RINFO_RES_NORM] = rnorm2;
                if((T(0) < rnorm2 && rnorm2 <= atol) || (T(0) < rnorm2 && rnorm2 <= brtol) || rnorm2 == T(0))
                {
                    // (rnorm2 == 0: an exact initial guess; the reference would scale by 1/0)
                    *ircomm           = aoclsparse_rci_stop;
                    rinfo[RINFO_ITER] = (T)niter;
                    break;
                }
                MI355_TRY(launch_scale<T>(st, V, n, T(1) / rnorm2));
                stage = GM_PRECOND_R0;
                if(!precond)
                    loop = true;
                else
                {
                    *ircomm = aoclsparse_rci_precond;
                    *io1 = V + (long long)j * ld, *io2 = Z + (long long)j * ld;
                }
                break;
            }
            case GM_PRECOND_R0:
            case GM_AFTER_PRECOND:
                *io1    = (precond ? Z : V) + (long long)j * ld;
                *io2    = V + (long long)(j + 1) * ld;
                *ircomm = aoclsparse_rci_mv;
                stage    = GM_ARNOLDI_STEP;
                break;
            case GM_ARNOLDI_STEP:
            {
                T *w = V + (long long)(j + 1) * ld;
                // classical Gram-Schmidt: all h(i,j) from the unmodified w, then one update (:1087-1113)
                std::vector<T> col((size_t)j + 1);
                MI355_TRY(dots(rt, (int)j + 1, V, ld, w, col.data()));
                for(aoclsparse_int i = 0; i <= j; i++)
                    h[(size_t)i * m + j] = col[i];
                MI355_TRY(upload_coef(rt, col.data(), (int)j + 1));
                MI355_TRY(launch_lincomb<T>(st, -1, n, (int)j + 1, coef.as<T>(), V, ld, w));
                T hh = 0;
                MI355_TRY(nrm2(rt, w, hh));
                if(hh < atol || hh < brtol)
                {
                    // the new direction is (numerically) inside the Krylov space already (:1121-1136)
                    niter += j + 1;
                    rinfo[RINFO_ITER]     = (T)niter;
                    rinfo[RINFO_RES_NORM] = hh;
                    *ircomm               = aoclsparse_rci_stop;
                    break;
                }
                MI355_TRY(launch_scale<T>(st, w, n, T(1) / hh));
                for(aoclsparse_int i = 0; i < j; i++) // previous plane rotations on the new column
                {
                    const T r1 = h[(size_t)i * m + j], r2 = h[(size_t)(i + 1) * m + j];
                    h[(size_t)i * m + j]       = c[i] * r1 - s[i] * r2;
                    h[(size_t)(i + 1) * m + j] = s[i] * r1 + c[i] * r2;
                }
                T rr = h[(size_t)j * m + j];
                hh   = -hh;
                lartg<T>(rr, hh, c[j], s[j], h[(size_t)j * m + j]);
                const T g0 = g[j];
                g[j]       = c[j] * g0;
                g[j + 1]   = s[j] * g0;
                rnorm2     = std::fabs(g[j]); // sic (:1173)
                rinfo[RINFO_ITER]     = (T)niter;
                rinfo[RINFO_RES_NORM] = rnorm2;
                j++;
                if(j >= m)
                {
                    stage = GM_UPDATE_X;
                    loop = true;
                    break;
                }
                stage = GM_AFTER_PRECOND;
                if(!precond)
                    loop = true;
                else
                {
                    *ircomm = aoclsparse_rci_precond;
                    *io1 = V + (long long)j * ld, *io2 = Z + (long long)j * ld;
                }
                break;
            }
            case GM_UPDATE_X:
            {
                // back substitution with the rotated Hessenberg matrix (:885-908); the result shares the
                // array of the rotation sines, as in the reference
                for(aoclsparse_int jj = j - 1; jj >= 0; jj--)
                {
                    T yj = g[jj];
                    for(aoclsparse_int i = jj + 1; i < j; i++)
                        yj -= h[(size_t)jj * m + i] * s[i];
                    const T diag = h[(size_t)jj * m + jj];
                    if(near_zero(diag))
                    {
                        exit_status = aoclsparse_status_numerical_error;
                        break;
                    }
                    s[jj] = yj / diag;
                }
                if(exit_status != aoclsparse_status_success)
                    break;
                MI355_TRY(upload_coef(rt, s.data(), (int)m));
                MI355_TRY(launch_lincomb<T>(st, 1, n, (int)m, coef.as<T>(), precond ? Z : V, ld, x));
                x_dirty = true;
                rnorm2  = std::fabs(g[j]);
                niter += j;
                rinfo[RINFO_ITER]     = (T)niter;
                rinfo[RINFO_RES_NORM] = rnorm2;
                const bool below_abs = T(0) < atol && rnorm2 <= atol;
                const bool below_rel = T(0) < rnorm2 && rnorm2 <= brtol;
                const bool at_max    = maxit > 0 && niter >= maxit;
                if(j >= m)
                    j = 0;
                *ircomm = aoclsparse_rci_stopping_criterion;
                stage    = (below_abs || below_rel || at_max) ? GM_CHECK : GM_RESTART;
                break;
            }
            case GM_RESTART:
                *io1 = x, *io2 = V;
                *ircomm = aoclsparse_rci_mv;
                stage    = GM_RESIDUAL;
                break;
            case GM_CHECK:
                if((T(0) < atol && rnorm2 <= atol) || (T(0) < rnorm2 && rnorm2 <= brtol))
                    *ircomm = aoclsparse_rci_stop;
                else if(maxit > 0 && niter >= maxit)
                {
                    exit_status = aoclsparse_status_maxit;
                    *ircomm     = aoclsparse_rci_stop;
                }
                break;
            default:
                *ircomm = aoclsparse_rci_stop;
                return aoclsparse_status_internal_error;
            }
        } while(loop);
        return exit_status;
    }

    // aoclsparse_itsol_rci_solve (:481-553); x must be addressable by the GPU
    aoclsparse_status rci(Runtime &rt, aoclsparse_itsol_rci_job *ircomm, T **u, T **vv, T *x, T *rinfo)
    {
        aoclsparse_status st;
        if(!solving)
        {
            st = init();
            if(st != aoclsparse_status_success)
            {
                *ircomm = aoclsparse_rci_stop;
                return st;
            }
            solving     = true;
            opts.locked = true;
        }
        st = method == solver_cg ? cg_step(rt, ircomm, u, vv, x, rinfo) : gmres_step(rt, ircomm, u, vv, x, rinfo);
        if(st != aoclsparse_status_success)
            *ircomm = aoclsparse_rci_stop;
        if(*ircomm == aoclsparse_rci_stop)
        {
            solving     = false;
            opts.locked = false;
        }
        return st;
    }
};

// aoclsparse_itsol_rci_input (:294-330)
template <typename T>
aoclsparse_status set_rhs(Solver<T> &S, aoclsparse_int n, const T *b, bool force_device)
{
    if(n < 0)
        return aoclsparse_status_invalid_value;
    if(!b)
        return aoclsparse_status_invalid_pointer;
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    S.free_solver_data();
    const bool bdev = rt.is_device_pointer(b);
    S.pinned        = !force_device && !bdev;
    S.n             = n;
    MI355_TRY(S.b.alloc(sizeof(T) * (size_t)n, S.pinned));
    if(n > 0)
        MI355_HIP_TRY(hipMemcpy(S.b.ptr, b, sizeof(T) * (size_t)n, hipMemcpyDefault));
    S.have_b  = true;
    S.solving = false;
    return aoclsparse_status_success;
}

// public RCI step: with host workspaces x is shadowed in pinned memory and synchronised around the step
template <typename T>
aoclsparse_status rci_public(Solver<T> *S, aoclsparse_itsol_rci_job *ircomm, T **u, T **v, T *x, T *rinfo)
{
    if(!ircomm)
        return aoclsparse_status_invalid_pointer;
    if(!S)
    {
        *ircomm = aoclsparse_rci_stop;
        return aoclsparse_status_internal_error;
    }
    if(!u || !v || !x || !rinfo)
    {
        *ircomm = aoclsparse_rci_stop;
        return aoclsparse_status_invalid_pointer;
    }
    if(!S->have_b)
    {
        *ircomm = aoclsparse_rci_stop;
        return aoclsparse_status_invalid_pointer; // rci_input was never called (the reference dereferences b)
    }
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    T *xd = x;
    if(S->pinned)
    {
        const bool starting = !S->solving;
        MI355_TRY(S->xshadow.alloc(sizeof(T) * (size_t)S->n, true));
        xd = S->xshadow.template as<T>();
        if(starting)
            std::memcpy(xd, x, sizeof(T) * (size_t)S->n);
    }
    S->x_dirty                 = false;
    const aoclsparse_status st = S->rci(rt, ircomm, u, v, xd, rinfo);
    if(S->pinned)
    {
        (void)hipStreamSynchronize(rt.stream()); // the caller reads u / v / x from the CPU next
        if(S->x_dirty)
            std::memcpy(x, xd, sizeof(T) * (size_t)S->n);
        if(u && *u == xd)
            *u = x; // GMRES hands x itself out as the mv operand
    }
    return st;
}

// typed access to the executors
inline aoclsparse_status exec_mv(aoclsparse_matrix A, const aoclsparse_mat_descr d, const double *x, double *y)
{
    const double one = 1.0, zero = 0.0;
    return aoclsparse_dmv(aoclsparse_operation_none, &one, A, d, x, &zero, y);
}
inline aoclsparse_status exec_mv(aoclsparse_matrix A, const aoclsparse_mat_descr d, const float *x, float *y)
{
    const float one = 1.0f, zero = 0.0f;
    return aoclsparse_smv(aoclsparse_operation_none, &one, A, d, x, &zero, y);
}
inline aoclsparse_status exec_trsv(aoclsparse_operation op, aoclsparse_matrix A, const aoclsparse_mat_descr d,
                                   const double *b, double *x)
{
    return aoclsparse_dtrsv(op, 1.0, A, d, b, x);
}
inline aoclsparse_status exec_trsv(aoclsparse_operation op, aoclsparse_matrix A, const aoclsparse_mat_descr d,
                                   const float *b, float *x)
{
    return aoclsparse_strsv(op, 1.0f, A, d, b, x);
}
inline aoclsparse_status exec_ilu(aoclsparse_matrix A, const aoclsparse_mat_descr d, double *x, const double *b)
{
    double *f = nullptr;
    return aoclsparse_dilu_smoother(aoclsparse_operation_none, A, d, &f, nullptr, x, b);
}
inline aoclsparse_status exec_ilu(aoclsparse_matrix A, const aoclsparse_mat_descr d, float *x, const float *b)
{
    float *f = nullptr;
    return aoclsparse_silu_smoother(aoclsparse_operation_none, A, d, &f, nullptr, x, b);
}

// the built-in SymGS preconditioner of CG (aoclsparse_itsol_symgs, :390-479): (L+D) y = r, y := D y, (U+D) z = y
template <typename T>
aoclsparse_status precond_symgs(Runtime &rt, aoclsparse_matrix A, const aoclsparse_mat_descr descr, const T *r, T *y,
                                T *z)
{
    if(descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_symmetric)
        return aoclsparse_status_invalid_value;
    if(descr->diag_type == aoclsparse_diag_type_zero)
        return aoclsparse_status_invalid_value;
    _aoclsparse_mat_descr d = *descr;
    d.type                  = aoclsparse_matrix_type_triangular;
    const bool lower_direct = descr->type == aoclsparse_matrix_type_general || descr->fill_mode == aoclsparse_fill_mode_lower;
    d.fill_mode             = lower_direct ? aoclsparse_fill_mode_lower : aoclsparse_fill_mode_upper;
    MI355_TRY(exec_trsv(lower_direct ? aoclsparse_operation_none : aoclsparse_operation_transpose, A, &d, r, y));
    if(descr->diag_type == aoclsparse_diag_type_non_unit)
        MI355_TRY(launch_vec_mul<T>(rt.stream(), A->m, A->dev_diag.as<T>(), y)); // diagonal of the clean CSR
    const bool upper_direct = descr->type == aoclsparse_matrix_type_general || descr->fill_mode == aoclsparse_fill_mode_upper;
    d.fill_mode             = upper_direct ? aoclsparse_fill_mode_upper : aoclsparse_fill_mode_lower;
    MI355_TRY(exec_trsv(upper_direct ? aoclsparse_operation_none : aoclsparse_operation_transpose, A, &d, y, z));
    return aoclsparse_status_success;
}

// aoclsparse_itsol_solve + aoclsparse_cg_solve / aoclsparse_gmres_solve (:556-630, :1369-1619)
template <typename T>
aoclsparse_status solve_direct(Solver<T> *S, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
                               const T *b, T *x, T *rinfo,
                               aoclsparse_int precond(aoclsparse_int, aoclsparse_int, const T *, T *, void *),
                               aoclsparse_int monit(aoclsparse_int, const T *, const T *, T *, void *), void *udata,
                               aoclsparse_matrix_data_type vt)
{
    if(!S)
        return aoclsparse_status_internal_error;
    if(!x || !rinfo)
        return aoclsparse_status_invalid_pointer;
    for(int i = 0; i < 100; i++)
        rinfo[i] = T(0);
    MI355_TRY(set_rhs(*S, n, b, true));
    MI355_TRY(S->init());
    if(!mat || !descr)
        return aoclsparse_status_invalid_pointer;
    if(mat->val_type != vt)
        return aoclsparse_status_wrong_type;
    MI355_TRY(csr_optimize(mat));
    Runtime &rt = Runtime::get();

    if(mat->m != n || mat->n != n)
        return aoclsparse_status_invalid_size;
    if(S->method == solver_cg)
    {
        if(descr->type != aoclsparse_matrix_type_symmetric || descr->fill_mode != aoclsparse_fill_mode_lower)
            return aoclsparse_status_invalid_value;
        if(S->precond == 1 && !precond)
            return aoclsparse_status_invalid_pointer;
        if(S->precond == 3)
        {
            if((!mat->opt_csr_full_diag && descr->diag_type != aoclsparse_diag_type_unit)
               || descr->diag_type == aoclsparse_diag_type_zero)
                return aoclsparse_status_invalid_value;
            MI355_TRY(S->y.alloc(sizeof(T) * (size_t)n, false));
        }
    }
    else if(S->precond == 1 && !precond)
        return aoclsparse_status_invalid_pointer;

    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    // x in HBM for the whole solve
    const bool xdev = rt.is_device_pointer(x);
    T         *xd   = x;
    if(!xdev)
    {
        MI355_TRY(S->xshadow.alloc(sizeof(T) * (size_t)n, false));
        xd = S->xshadow.template as<T>();
        MI355_HIP_TRY(hipMemcpy(xd, x, sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
    }
    std::vector<T> hu, hv; // host copies for user callbacks
    if(precond || monit)
    {
        try
        {
            hu.resize((size_t)n), hv.resize((size_t)n);
        }
        catch(const std::bad_alloc &)
        {
            return aoclsparse_status_memory_error;
        }
    }
    auto to_host = [&](std::vector<T> &h, const T *d) -> aoclsparse_status {
        MI355_HIP_TRY(hipMemcpyAsync(h.data(), d, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        return aoclsparse_status_success;
    };

    S->solving     = true;
    S->opts.locked = true;
    aoclsparse_itsol_rci_job ircomm = aoclsparse_rci_start;
    T                       *u = nullptr, *v = nullptr;
    aoclsparse_status        exit_status = aoclsparse_status_success, st;
    auto finish = [&](aoclsparse_status code) {
        S->solving     = false;
        S->opts.locked = false;
        if(!xdev)
            (void)hipMemcpy(x, xd, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost);
        return code;
    };
    DeviceScope scope;
    while(ircomm != aoclsparse_rci_stop)
    {
        exit_status = S->rci(rt, &ircomm, &u, &v, xd, rinfo);
        if(exit_status != aoclsparse_status_success && ircomm != aoclsparse_rci_stop)
            return finish(exit_status);
        switch(ircomm)
        {
        case aoclsparse_rci_mv:
            if(exec_mv(mat, descr, u, v) != aoclsparse_status_success)
                return finish(aoclsparse_status_internal_error);
            break;
        case aoclsparse_rci_precond:
            if(S->precond == 1)
            {
                st = to_host(hu, u);
                if(st != aoclsparse_status_success)
                    return finish(st);
                if(precond(0, n, hu.data(), hv.data(), udata) != 0)
                    ircomm = aoclsparse_rci_interrupt;
                if(hipMemcpy(v, hv.data(), sizeof(T) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)
                    return finish(aoclsparse_status_internal_error);
            }
            else if(S->method == solver_cg && S->precond == 3)
            {
                if(precond_symgs<T>(rt, mat, descr, u, S->y.template as<T>(), v) != aoclsparse_status_success)
                    return finish(aoclsparse_status_internal_error);
            }
            else if(S->method == solver_gmres && S->precond == 2)
                (void)exec_ilu(mat, descr, v, u); // status ignored by the reference as well (:1589-1594)
            else if(launch_vec_copy<T>(rt.stream(), n, u, v) != aoclsparse_status_success)
                return finish(aoclsparse_status_internal_error);
            break;
        case aoclsparse_rci_stopping_criterion:
            if(monit)
            {
                // CG hands the residual out as u (v = nullptr); GMRES leaves the last operands.  The callback
                // gets host copies: (x, r) here, which is what its documented signature promises.
                st = to_host(hu, xd);
                if(st == aoclsparse_status_success && S->method == solver_cg)
                    st = to_host(hv, S->r.template as<T>());
                if(st != aoclsparse_status_success)
                    return finish(st);
                if(monit(n, hu.data(), S->method == solver_cg ? hv.data() : nullptr, rinfo, udata) != 0)
                    ircomm = aoclsparse_rci_interrupt;
            }
            break;
        default:
            break;
        }
    }
    return finish(exit_status);
}

// ==== complex handles (aoclsparse_itsol_{c,z}_*) ==========================================================
// The same two state machines with complex vectors (itsol_functions.hpp is one template over T).  CG: the products
// r.z and p.q are the UNCONJUGATED sums the reference forms (:786-812), i.e. the conjugate-orthogonal CG for complex
// SYMMETRIC matrices; norms and tolerances are real (tolerance_t<T>).  GMRES deviates on purpose: the reference stores
// h(i,j) = sum conj(w_k) v_i[k] (the conjugate of the Hessenberg entry, :1090-1099), rotates with a complex sine WITHOUT
// the conjugation a unitary rotation needs (:1140-1146) and updates x with conj(V) (:1255-1262), which cancels only
// when the Krylov data are real multiples of one complex number -- as in its own sample (A = (1 + 0.1i) A_real with
// ILU(0)); restated literally it diverges on a general complex matrix.  Here h(i,j) = v_i^H w, the rotation is
// [c s; -conj(s) c] from ?lartg on (h(j,j), |w|) and x += sum y_i v_i, which reproduces the reference wherever it
// converges.  Vector steps are the generic complex kernels of complex_kernels.hip, one reduction read back per
// scalar: correctness path, not tuned.  Preconditioners: none, user, ILU(0) for GMRES; the built-in
// SymGS of CG is not offered for complex handles (not_implemented).
template <typename R>
void clartg(std::complex<R> f, std::complex<R> g, R &c, std::complex<R> &s, std::complex<R> &r)
{
    // LAPACK 3.10 zlartg (la_lartg.f90), unscaled branch; operands outside [rtmin, rtmax] are scaled by their
    // largest component first
    using Cs = std::complex<R>;
    if(g == Cs(0))
    {
        c = R(1), s = Cs(0), r = f;
        return;
    }
    const R g1 = std::max(std::fabs(g.real()), std::fabs(g.imag()));
    if(f == Cs(0))
    {
        const Cs gs = g / g1;
        const R  d  = std::sqrt(std::norm(gs));
        c = R(0), s = std::conj(gs) / d, r = Cs(d * g1);
        return;
    }
    const R  f1 = std::max(std::fabs(f.real()), std::fabs(f.imag()));
    const R  safmin = std::numeric_limits<R>::min(), safmax = R(1) / safmin;
    const R  rtmin = std::sqrt(safmin), rtmax = std::sqrt(safmax / 4);
    R        u = R(1);
    Cs       fs = f, gs = g;
    if(!(f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax))
    {
        u  = std::min(safmax, std::max(safmin, std::max(f1, g1)));
        fs = f / u, gs = g / u;
    }
    const R f2 = std::norm(fs), g2 = std::norm(gs), h2 = f2 + g2;
    c          = std::sqrt(f2 / h2);
    r          = (fs / c) * u;
    s          = std::conj(gs) * (fs * (R(1) / std::sqrt(f2 * h2)));
}

template <typename R>
struct CSolver
{
    using C  = cplx<R>;
    using Cs = std::complex<R>;
    aoclsparse_int n = 0;
    bool           have_b = false, pinned = false, solving = false, x_dirty = false;
    int            method = solver_cg, stage = CG_ENTRY, precond = 0;
    Options        opts;
    VBuf           b, xshadow, red_partial, red_out, r, z, p, q, v, zz, symgs_y;
    Cs             alpha = 0, rz = 0, beta = 0;
    R              rnorm2 = 0, bnorm2 = 0, brtol = 0, rtol = 0, atol = 0;
    aoclsparse_int niter = 0, maxit = 0, j = 0, restart = 0;
    std::vector<Cs> h, g, s;
    std::vector<R>  c;

    static C dev(Cs v)
    {
        return C(v.real(), v.imag());
    }
    void free_solver_data()
    {
        r.release(), z.release(), p.release(), q.release(), v.release(), zz.release();
        h.clear(), g.clear(), s.clear(), c.clear();
        stage = CG_ENTRY, niter = 0, j = 0;
    }
    aoclsparse_status dot(Runtime &rt, const C *x, const C *y, bool conj_x, Cs &out)
    {
        MI355_TRY(red_partial.alloc(sizeof(C) * 1024, false));
        MI355_TRY(red_out.alloc(sizeof(C), false));
        MI355_TRY(launch_cdot<R>(rt.stream(), n, x, y, red_partial.as<C>(), red_out.as<C>(), conj_x));
        C hv(R(0), R(0));
        MI355_HIP_TRY(hipMemcpyAsync(&hv, red_out.ptr, sizeof(C), hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        out = Cs(hv.re, hv.im);
        return aoclsparse_status_success;
    }
    aoclsparse_status nrm2(Runtime &rt, const C *x, R &out)
    {
        Cs d;
        MI355_TRY(dot(rt, x, x, true, d));
        out = std::sqrt(d.real());
        return aoclsparse_status_success;
    }
    aoclsparse_status axpby(Runtime &rt, Cs a, const C *x, Cs bb, const C *y, C *w)
    {
        return launch_caxpby<R>(rt.stream(), n, dev(a), x, dev(bb), y, w);
    }
    aoclsparse_status init()
    {
        method = opts.reg["iterative method"].key;
        const size_t nb = sizeof(C) * (size_t)n;
        if(method == solver_cg)
        {
            MI355_TRY(r.alloc(nb, pinned));
            MI355_TRY(z.alloc(nb, pinned));
            MI355_TRY(p.alloc(nb, pinned));
            MI355_TRY(q.alloc(nb, pinned));
            stage    = CG_ENTRY;
            precond = opts.reg["cg preconditioner"].key;
            rtol = (R)opts.reg["cg rel tolerance"].rval, atol = (R)opts.reg["cg abs tolerance"].rval;
            maxit = (aoclsparse_int)opts.reg["cg iteration limit"].ival;
        }
        else
        {
            if(v.ptr == nullptr)
            {
                restart           = (aoclsparse_int)opts.reg["gmres restart iterations"].ival;
                const long long m = restart;
                if((m + 1) * (long long)n > std::numeric_limits<aoclsparse_int>::max()
                   || m * m > std::numeric_limits<aoclsparse_int>::max())
                    return aoclsparse_status_invalid_size;
                const size_t kb = nb * (size_t)(m + 1);
                MI355_TRY(v.alloc(kb, pinned));
                MI355_TRY(zz.alloc(kb, pinned));
                MI355_HIP_TRY(hipMemset(v.ptr, 0, kb));
                MI355_HIP_TRY(hipMemset(zz.ptr, 0, kb));
                try
                {
                    h.assign((size_t)(m * m), Cs(0)), g.assign((size_t)m + 1, Cs(0));
                    s.assign((size_t)m, Cs(0)), c.assign((size_t)m, R(0));
                }
                catch(const std::bad_alloc &)
                {
                    return aoclsparse_status_memory_error;
                }
                niter = 0, j = 0;
            }
            stage    = GM_ENTRY;
            precond = opts.reg["gmres preconditioner"].key;
            rtol = (R)opts.reg["gmres rel tolerance"].rval, atol = (R)opts.reg["gmres abs tolerance"].rval;
            maxit = (aoclsparse_int)opts.reg["gmres iteration limit"].ival;
        }
        return aoclsparse_status_success;
    }
    static bool tiny(Cs v)
    {
        return std::abs(v) <= R(1e-2) * R(2) * std::numeric_limits<R>::epsilon();
    }

    aoclsparse_status cg_step(Runtime &rt, aoclsparse_itsol_rci_job *ircomm, C **u, C **vv, C *x, R *rinfo)
    {
        aoclsparse_status exit_status = aoclsparse_status_success;
        C *rp = r.as<C>(), *zp = z.as<C>(), *pp = p.as<C>(), *qp = q.as<C>();
        if(stage != CG_ENTRY && *ircomm == aoclsparse_rci_interrupt)
        {
            *ircomm = aoclsparse_rci_stop;
            return aoclsparse_status_user_stop;
        }
        bool loop;
        do
        {
            loop = false;
            switch(stage)
            {
            case CG_ENTRY:
                for(int i = 0; i < 100; i++)
                    rinfo[i] = R(0);
                niter = 0;
                // r = -b, p = x (:676-686)
                MI355_TRY(axpby(rt, Cs(-1), b.as<C>(), Cs(0), nullptr, rp));
                MI355_TRY(axpby(rt, Cs(1), x, Cs(0), nullptr, pp));
                MI355_TRY(nrm2(rt, b.as<C>(), bnorm2));
                if(bnorm2 != bnorm2)
                    return aoclsparse_status_invalid_value;
                rinfo[RINFO_RHS_NORM] = bnorm2;
                brtol                 = rtol * bnorm2;
                *ircomm = aoclsparse_rci_mv, stage = CG_RESIDUAL;
                *u = pp, *vv = qp;
                break;
            case CG_RESIDUAL:
                MI355_TRY(axpby(rt, Cs(1), rp, Cs(1), qp, rp));
                MI355_TRY(nrm2(rt, rp, rnorm2));
                if(rnorm2 != rnorm2)
                {
                    exit_status = aoclsparse_status_numerical_error;
                    break;
                }
                rinfo[RINFO_RES_NORM] = rnorm2;
                MI355_TRY(axpby(rt, Cs(0), nullptr, Cs(0), nullptr, pp));
                rz   = Cs(1, 1); // sic (:723)
                stage = CG_CHECK;
                [[fallthrough]];
            case CG_CHECK:
                *u = rp, *vv = nullptr;
                if((R(0) < atol && rnorm2 <= atol) || (R(0) < rtol && rnorm2 <= brtol))
                {
                    *ircomm = aoclsparse_rci_stop;
                    break;
                }
                if(maxit > 0 && niter > maxit)
                {
                    *ircomm = aoclsparse_rci_stop, exit_status = aoclsparse_status_maxit;
                    break;
                }
                stage = CG_PRECOND_Z, *ircomm = aoclsparse_rci_stopping_criterion;
                break;
            case CG_PRECOND_Z:
                niter++;
                rinfo[RINFO_ITER] = (R)niter;
                stage              = CG_DIRECTION;
                if(precond)
                {
                    *ircomm = aoclsparse_rci_precond;
                    *u = rp, *vv = zp;
                    break;
                }
                MI355_TRY(axpby(rt, Cs(1), rp, Cs(0), nullptr, zp));
                [[fallthrough]];
            case CG_DIRECTION:
            {
                Cs rz_new;
                MI355_TRY(dot(rt, rp, zp, false, rz_new));
                if(tiny(rz))
                    return aoclsparse_status_numerical_error;
                beta = rz_new / rz, rz = rz_new;
                MI355_TRY(axpby(rt, beta, pp, Cs(-1), zp, pp));
                *ircomm = aoclsparse_rci_mv, stage = CG_UPDATE_X;
                *u = pp, *vv = qp;
                break;
            }
            case CG_UPDATE_X:
            {
                Cs pq;
                MI355_TRY(dot(rt, pp, qp, false, pq));
                if(tiny(pq) || pq == Cs(0))
                    return aoclsparse_status_numerical_error;
                alpha = rz / pq;
                MI355_TRY(axpby(rt, alpha, pp, Cs(1), x, x));
                MI355_TRY(axpby(rt, alpha, qp, Cs(1), rp, rp));
                x_dirty = true;
                MI355_TRY(nrm2(rt, rp, rnorm2));
                if(rnorm2 != rnorm2)
                {
                    exit_status = aoclsparse_status_numerical_error;
                    break;
                }
                rinfo[RINFO_RES_NORM] = rnorm2;
                loop = true, stage = CG_CHECK;
                break;
            }
            default:
                *ircomm = aoclsparse_rci_stop;
                return aoclsparse_status_internal_error;
            }
        } while(loop);
        return exit_status;
    }

    aoclsparse_status gmres_step(Runtime &rt, aoclsparse_itsol_rci_job *ircomm, C **io1, C **io2, C *x, R *rinfo)
    {
        aoclsparse_status    exit_status = aoclsparse_status_success;
        const aoclsparse_int m           = restart;
        const long long      ld          = n;
        C                   *V = v.as<C>(), *Z = zz.as<C>();
        if(stage != GM_ENTRY && *ircomm == aoclsparse_rci_interrupt)
        {
            *ircomm = aoclsparse_rci_stop;
            return aoclsparse_status_user_stop;
        }
        bool loop;
        do
        {
            loop = false;
            switch(stage)
            {
            case GM_ENTRY:
                *io1 = x, *io2 = V;
                *ircomm = aoclsparse_rci_mv, stage = GM_RESIDUAL;
                break;
            case GM_RESIDUAL:
            {
                MI355_TRY(nrm2(rt, b.as<C>(), bnorm2));
                if(std::isnan(bnorm2))
                    return aoclsparse_status_invalid_value;
                brtol                 = rtol * bnorm2;
                rinfo[RINFO_RHS_NORM] = brtol; // sic (:1004)
                if(near_zero(atol) && near_zero(brtol))
                {
                    exit_status = aoclsparse_status_invalid_value, *ircomm = aoclsparse_rci_stop;
                    break;
                }
                MI355_TRY(axpby(rt, Cs(1), b.as<C>(), Cs(-1), V, V)); // v0 = b - A x
                R g0 = 0;
                MI355_TRY(nrm2(rt, V, g0));
                g[0] = Cs(g0, 0), rnorm2 = g0;
                rinfo[RINFO_RES_NORM] = rnorm2;
                if((R(0) < rnorm2 && rnorm2 <= atol) || (R(0) < rnorm2 && rnorm2 <= brtol) || rnorm2 == R(0))
                {
                    *ircomm = aoclsparse_rci_stop, rinfo[RINFO_ITER] = (R)niter;
                    break;
                }
                MI355_TRY(axpby(rt, Cs(R(1) / rnorm2), V, Cs(0), nullptr, V));
                stage = GM_PRECOND_R0;
                if(!precond)
                    loop = true;
                else
                {
                    *ircomm = aoclsparse_rci_precond;
                    *io1 = V + (long long)j * ld, *io2 = Z + (long long)j * ld;
                }
                break;
            }
            case GM_PRECOND_R0:
            case GM_AFTER_PRECOND:
                *io1 = (precond ? Z : V) + (long long)j * ld, *io2 = V + (long long)(j + 1) * ld;
                *ircomm = aoclsparse_rci_mv, stage = GM_ARNOLDI_STEP;
                break;
            case GM_ARNOLDI_STEP:
            {
                C *w = V + (long long)(j + 1) * ld;
                for(aoclsparse_int i = 0; i <= j; i++) // all h(i,j) = v_i^H w from the unmodified w, then the update
                    MI355_TRY(dot(rt, V + (long long)i * ld, w, true, h[(size_t)i * m + j]));
                for(aoclsparse_int i = 0; i <= j; i++)
                    MI355_TRY(axpby(rt, -h[(size_t)i * m + j], V + (long long)i * ld, Cs(1), w, w));
                R hh = 0;
                MI355_TRY(nrm2(rt, w, hh));
                if(hh < atol || hh < brtol)
                {
                    niter += j + 1;
                    rinfo[RINFO_ITER] = (R)niter, rinfo[RINFO_RES_NORM] = hh;
                    *ircomm = aoclsparse_rci_stop;
                    break;
                }
                MI355_TRY(axpby(rt, Cs(R(1) / hh), w, Cs(0), nullptr, w));
                for(aoclsparse_int i = 0; i < j; i++)
                {
                    const Cs r1 = h[(size_t)i * m + j], r2 = h[(size_t)(i + 1) * m + j];
                    h[(size_t)i * m + j]       = c[i] * r1 + s[i] * r2;
                    h[(size_t)(i + 1) * m + j] = -std::conj(s[i]) * r1 + c[i] * r2;
                }
                const Cs rr = h[(size_t)j * m + j];
                clartg<R>(rr, Cs(hh, 0), c[j], s[j], h[(size_t)j * m + j]);
                const Cs g0 = g[j];
                g[j] = c[j] * g0, g[j + 1] = -std::conj(s[j]) * g0;
                rnorm2 = std::abs(g[j]); // sic (:1173)
                rinfo[RINFO_ITER] = (R)niter, rinfo[RINFO_RES_NORM] = rnorm2;
                j++;
                if(j >= m)
                {
                    stage = GM_UPDATE_X, loop = true;
                    break;
                }
                stage = GM_AFTER_PRECOND;
                if(!precond)
                    loop = true;
                else
                {
                    *ircomm = aoclsparse_rci_precond;
                    *io1 = V + (long long)j * ld, *io2 = Z + (long long)j * ld;
                }
                break;
            }
            case GM_UPDATE_X:
            {
                for(aoclsparse_int jj = j - 1; jj >= 0; jj--)
                {
                    Cs yj = g[jj];
                    for(aoclsparse_int i = jj + 1; i < j; i++)
                        yj -= h[(size_t)jj * m + i] * s[i];
                    const Cs diag = h[(size_t)jj * m + jj];
                    if(tiny(diag))
                    {
                        exit_status = aoclsparse_status_numerical_error;
                        break;
                    }
                    s[jj] = yj / diag;
                }
                if(exit_status != aoclsparse_status_success)
                    break;
                for(aoclsparse_int i = 0; i < m; i++) // x += sum y_i v_i (z_i with a preconditioner), :1213-1232
                    MI355_TRY(axpby(rt, s[i], (precond ? Z : V) + (long long)i * ld, Cs(1), x, x));
                x_dirty = true;
                rnorm2  = std::abs(g[j]);
                niter += j;
                rinfo[RINFO_ITER] = (R)niter, rinfo[RINFO_RES_NORM] = rnorm2;
                const bool below_abs = R(0) < atol && rnorm2 <= atol, below_rel = R(0) < rnorm2 && rnorm2 <= brtol;
                const bool at_max    = maxit > 0 && niter >= maxit;
                if(j >= m)
                    j = 0;
                *ircomm = aoclsparse_rci_stopping_criterion;
                stage    = (below_abs || below_rel || at_max) ? GM_CHECK : GM_RESTART;
                break;
            }
            case GM_RESTART:
                *io1 = x, *io2 = V;
                *ircomm = aoclsparse_rci_mv, stage = GM_RESIDUAL;
                break;
            case GM_CHECK:
                if((R(0) < atol && rnorm2 <= atol) || (R(0) < rnorm2 && rnorm2 <= brtol))
                    *ircomm = aoclsparse_rci_stop;
                else if(maxit > 0 && niter >= maxit)
                    exit_status = aoclsparse_status_maxit, *ircomm = aoclsparse_rci_stop;
                break;
            default:
                *ircomm = aoclsparse_rci_stop;
                return aoclsparse_status_internal_error;
            }
        } while(loop);
        return exit_status;
    }

    aoclsparse_status rci(Runtime &rt, aoclsparse_itsol_rci_job *ircomm, C **u, C **vv, C *x, R *rinfo)
    {
        aoclsparse_status st;
        if(!solving)
        {
            st = init();
            if(st != aoclsparse_status_success)
            {
                *ircomm = aoclsparse_rci_stop;
                return st;
            }
            solving = true, opts.locked = true;
        }
        st = method == solver_cg ? cg_step(rt, ircomm, u, vv, x, rinfo) : gmres_step(rt, ircomm, u, vv, x, rinfo);
        if(st != aoclsparse_status_success)
            *ircomm = aoclsparse_rci_stop;
        if(*ircomm == aoclsparse_rci_stop)
            solving = false, opts.locked = false;
        return st;
    }
};

template <typename R>
aoclsparse_status cset_rhs(CSolver<R> &S, aoclsparse_int n, const cplx<R> *b, bool force_device)
{
    if(n < 0)
        return aoclsparse_status_invalid_value;
    if(!b)
        return aoclsparse_status_invalid_pointer;
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    S.free_solver_data();
    S.pinned = !force_device && !rt.is_device_pointer(b);
    S.n      = n;
    MI355_TRY(S.b.alloc(sizeof(cplx<R>) * (size_t)n, S.pinned));
    if(n > 0)
        MI355_HIP_TRY(hipMemcpy(S.b.ptr, b, sizeof(cplx<R>) * (size_t)n, hipMemcpyDefault));
    S.have_b = true, S.solving = false;
    return aoclsparse_status_success;
}

template <typename R>
aoclsparse_status crci_public(CSolver<R> *S, aoclsparse_itsol_rci_job *ircomm, cplx<R> **u, cplx<R> **v, cplx<R> *x,
                              R *rinfo)
{
    using C = cplx<R>;
    if(!ircomm)
        return aoclsparse_status_invalid_pointer;
    if(!S)
    {
        *ircomm = aoclsparse_rci_stop;
        return aoclsparse_status_internal_error;
    }
    if(!u || !v || !x || !rinfo)
    {
        *ircomm = aoclsparse_rci_stop;
        return aoclsparse_status_invalid_pointer;
    }
    if(!S->have_b)
    {
        *ircomm = aoclsparse_rci_stop;
        return aoclsparse_status_invalid_pointer;
    }
    Runtime &rt = Runtime::get();
    MI355_TRY(rt.init());
    C *xd = x;
    if(S->pinned)
    {
        const bool starting = !S->solving;
        MI355_TRY(S->xshadow.alloc(sizeof(C) * (size_t)S->n, true));
        xd = S->xshadow.template as<C>();
        if(starting)
            std::memcpy(xd, x, sizeof(C) * (size_t)S->n);
    }
    S->x_dirty                 = false;
    const aoclsparse_status st = S->rci(rt, ircomm, u, v, xd, rinfo);
    if(S->pinned)
    {
        (void)hipStreamSynchronize(rt.stream());
        if(S->x_dirty)
            std::memcpy(x, xd, sizeof(C) * (size_t)S->n);
        if(*u == xd)
            *u = x;
    }
    return st;
}

template <typename R>
aoclsparse_status cexec_mv(aoclsparse_matrix A, const aoclsparse_mat_descr d, const cplx<R> *x, cplx<R> *y);
template <>
aoclsparse_status cexec_mv<double>(aoclsparse_matrix A, const aoclsparse_mat_descr d, const cdouble *x, cdouble *y)
{
    const aoclsparse_double_complex one{1.0, 0.0}, zero{0.0, 0.0};
    return aoclsparse_zmv(aoclsparse_operation_none, &one, A, d, reinterpret_cast<const aoclsparse_double_complex *>(x), &zero,
                          reinterpret_cast<aoclsparse_double_complex *>(y));
}
template <>
aoclsparse_status cexec_mv<float>(aoclsparse_matrix A, const aoclsparse_mat_descr d, const cfloat *x, cfloat *y)
{
    const aoclsparse_float_complex one{1.0f, 0.0f}, zero{0.0f, 0.0f};
    return aoclsparse_cmv(aoclsparse_operation_none, &one, A, d, reinterpret_cast<const aoclsparse_float_complex *>(x), &zero,
                          reinterpret_cast<aoclsparse_float_complex *>(y));
}
inline aoclsparse_status cexec_ilu(aoclsparse_matrix A, const aoclsparse_mat_descr d, cdouble *x, const cdouble *b)
{
    aoclsparse_double_complex *f = nullptr;
    return aoclsparse_zilu_smoother(aoclsparse_operation_none, A, d, &f, nullptr, reinterpret_cast<aoclsparse_double_complex *>(x),
                                    reinterpret_cast<const aoclsparse_double_complex *>(b));
}
inline aoclsparse_status cexec_ilu(aoclsparse_matrix A, const aoclsparse_mat_descr d, cfloat *x, const cfloat *b)
{
    aoclsparse_float_complex *f = nullptr;
    return aoclsparse_cilu_smoother(aoclsparse_operation_none, A, d, &f, nullptr, reinterpret_cast<aoclsparse_float_complex *>(x),
                                    reinterpret_cast<const aoclsparse_float_complex *>(b));
}

inline aoclsparse_status cexec_trsv(aoclsparse_operation op, aoclsparse_matrix A, const aoclsparse_mat_descr d,
                                    const cdouble *b, cdouble *x)
{
    return aoclsparse_ztrsv(op, aoclsparse_double_complex{1.0, 0.0}, A, d, reinterpret_cast<const aoclsparse_double_complex *>(b),
                            reinterpret_cast<aoclsparse_double_complex *>(x));
}
inline aoclsparse_status cexec_trsv(aoclsparse_operation op, aoclsparse_matrix A, const aoclsparse_mat_descr d,
                                    const cfloat *b, cfloat *x)
{
    return aoclsparse_ctrsv(op, aoclsparse_float_complex{1.0f, 0.0f}, A, d, reinterpret_cast<const aoclsparse_float_complex *>(b),
                            reinterpret_cast<aoclsparse_float_complex *>(x));
}
// the built-in SymGS preconditioner of CG for complex handles: precond_symgs with complex solves and scale
template <typename R>
aoclsparse_status cprecond_symgs(Runtime &rt, aoclsparse_matrix A, const aoclsparse_mat_descr descr, const cplx<R> *r,
                                 cplx<R> *y, cplx<R> *z)
{
    if(descr->type != aoclsparse_matrix_type_general && descr->type != aoclsparse_matrix_type_symmetric)
        return aoclsparse_status_invalid_value;
    if(descr->diag_type == aoclsparse_diag_type_zero)
        return aoclsparse_status_invalid_value;
    _aoclsparse_mat_descr d = *descr;
    d.type                  = aoclsparse_matrix_type_triangular;
    const bool lower_direct = descr->type == aoclsparse_matrix_type_general || descr->fill_mode == aoclsparse_fill_mode_lower;
    d.fill_mode             = lower_direct ? aoclsparse_fill_mode_lower : aoclsparse_fill_mode_upper;
    MI355_TRY(cexec_trsv(lower_direct ? aoclsparse_operation_none : aoclsparse_operation_transpose, A, &d, r, y));
    if(descr->diag_type == aoclsparse_diag_type_non_unit)
        MI355_TRY(launch_cvec_mul<R>(rt.stream(), A->m, A->dev_diag.as<cplx<R>>(), y));
    const bool upper_direct = descr->type == aoclsparse_matrix_type_general || descr->fill_mode == aoclsparse_fill_mode_upper;
    d.fill_mode             = upper_direct ? aoclsparse_fill_mode_upper : aoclsparse_fill_mode_lower;
    MI355_TRY(cexec_trsv(upper_direct ? aoclsparse_operation_none : aoclsparse_operation_transpose, A, &d, y, z));
    return aoclsparse_status_success;
}

// aoclsparse_itsol_solve for complex handles: the loop of solve_direct with complex operands.  PT is the public complex
// struct of the callbacks (layout-identical to cplx<R>).
template <typename R, typename PT>
aoclsparse_status csolve_direct(CSolver<R> *S, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
                                const cplx<R> *b, cplx<R> *x, R *rinfo,
                                aoclsparse_int precond(aoclsparse_int, aoclsparse_int, const PT *, PT *, void *),
                                aoclsparse_int monit(aoclsparse_int, const PT *, const PT *, R *, void *), void *udata,
                                aoclsparse_matrix_data_type vt)
{
    using C = cplx<R>;
    if(!S)
        return aoclsparse_status_internal_error;
    if(!x || !rinfo)
        return aoclsparse_status_invalid_pointer;
    for(int i = 0; i < 100; i++)
        rinfo[i] = R(0);
    MI355_TRY(cset_rhs(*S, n, b, true));
    MI355_TRY(S->init());
    if(!mat || !descr)
        return aoclsparse_status_invalid_pointer;
    if(mat->val_type != vt)
        return aoclsparse_status_wrong_type;
    MI355_TRY(csr_optimize(mat));
    Runtime &rt = Runtime::get();
    if(mat->m != n || mat->n != n)
        return aoclsparse_status_invalid_size;
    if(S->method == solver_cg)
    {
        if(descr->type != aoclsparse_matrix_type_symmetric || descr->fill_mode != aoclsparse_fill_mode_lower)
            return aoclsparse_status_invalid_value;
        if(S->precond == 3)
        {
            if((!mat->opt_csr_full_diag && descr->diag_type != aoclsparse_diag_type_unit)
               || descr->diag_type == aoclsparse_diag_type_zero)
                return aoclsparse_status_invalid_value;
            MI355_TRY(S->symgs_y.alloc(sizeof(C) * (size_t)n, false));
        }
    }
    if(S->precond == 1 && !precond)
        return aoclsparse_status_invalid_pointer;
    std::lock_guard<std::recursive_mutex> sl(rt.stage_lock);
    const bool xdev = rt.is_device_pointer(x);
    C         *xd   = x;
    if(!xdev)
    {
        MI355_TRY(S->xshadow.alloc(sizeof(C) * (size_t)n, false));
        xd = S->xshadow.template as<C>();
        MI355_HIP_TRY(hipMemcpy(xd, x, sizeof(C) * (size_t)n, hipMemcpyHostToDevice));
    }
    std::vector<C> hu, hv;
    if(precond || monit)
    {
        try
        {
            hu.resize((size_t)n), hv.resize((size_t)n);
        }
        catch(const std::bad_alloc &)
        {
            return aoclsparse_status_memory_error;
        }
    }
    auto to_host = [&](std::vector<C> &hh, const C *d) -> aoclsparse_status {
        MI355_HIP_TRY(hipMemcpyAsync(hh.data(), d, sizeof(C) * (size_t)n, hipMemcpyDeviceToHost, rt.stream()));
        MI355_HIP_TRY(hipStreamSynchronize(rt.stream()));
        return aoclsparse_status_success;
    };
    S->solving = true, S->opts.locked = true;
    aoclsparse_itsol_rci_job ircomm = aoclsparse_rci_start;
    C                       *u = nullptr, *v = nullptr;
    aoclsparse_status        exit_status = aoclsparse_status_success, st;
    auto finish = [&](aoclsparse_status code) {
        S->solving = false, S->opts.locked = false;
        if(!xdev)
            (void)hipMemcpy(x, xd, sizeof(C) * (size_t)n, hipMemcpyDeviceToHost);
        return code;
    };
    DeviceScope scope;
    while(ircomm != aoclsparse_rci_stop)
    {
        exit_status = S->rci(rt, &ircomm, &u, &v, xd, rinfo);
        if(exit_status != aoclsparse_status_success && ircomm != aoclsparse_rci_stop)
            return finish(exit_status);
        switch(ircomm)
        {
        case aoclsparse_rci_mv:
            if(cexec_mv<R>(mat, descr, u, v) != aoclsparse_status_success)
                return finish(aoclsparse_status_internal_error);
            break;
        case aoclsparse_rci_precond:
            if(S->precond == 1)
            {
                st = to_host(hu, u);
                if(st != aoclsparse_status_success)
                    return finish(st);
                if(precond(0, n, reinterpret_cast<const PT *>(hu.data()), reinterpret_cast<PT *>(hv.data()), udata) != 0)
                    ircomm = aoclsparse_rci_interrupt;
                if(hipMemcpy(v, hv.data(), sizeof(C) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)
                    return finish(aoclsparse_status_internal_error);
            }
            else if(S->method == solver_cg && S->precond == 3)
            {
                if(cprecond_symgs<R>(rt, mat, descr, u, S->symgs_y.template as<C>(), v) != aoclsparse_status_success)
                    return finish(aoclsparse_status_internal_error);
            }
            else if(S->method == solver_gmres && S->precond == 2)
                (void)cexec_ilu(mat, descr, v, u);
            else if(S->axpby(rt, std::complex<R>(1), u, std::complex<R>(0), nullptr, v) != aoclsparse_status_success)
                return finish(aoclsparse_status_internal_error);
            break;
        case aoclsparse_rci_stopping_criterion:
            if(monit)
            {
                st = to_host(hu, xd);
                if(st == aoclsparse_status_success && S->method == solver_cg)
                    st = to_host(hv, S->r.template as<C>());
                if(st != aoclsparse_status_success)
                    return finish(st);
                if(monit(n, reinterpret_cast<const PT *>(hu.data()),
                         S->method == solver_cg ? reinterpret_cast<const PT *>(hv.data()) : nullptr, rinfo, udata)
                   != 0)
                    ircomm = aoclsparse_rci_interrupt;
            }
            break;
        default:
            break;
        }
    }
    return finish(exit_status);
}

} // namespace

struct _aoclsparse_itsol_handle
{
    aoclsparse_matrix_data_type type = aoclsparse_dmat;
    Solver<float>              *s    = nullptr;
    Solver<double>             *d    = nullptr;
    CSolver<float>             *c    = nullptr;
    CSolver<double>            *z    = nullptr;
};

extern "C" {

void aoclsparse_itsol_handle_prn_options(aoclsparse_itsol_handle handle)
{
    if(!handle)
        return;
    if(handle->type == aoclsparse_dmat && handle->d)
        handle->d->opts.print();
    else if(handle->type == aoclsparse_smat && handle->s)
        handle->s->opts.print();
    else if(handle->type == aoclsparse_zmat && handle->z)
        handle->z->opts.print();
    else if(handle->type == aoclsparse_cmat && handle->c)
        handle->c->opts.print();
}

aoclsparse_status aoclsparse_itsol_option_set(aoclsparse_itsol_handle handle, const char *option, const char *value)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type == aoclsparse_dmat)
        return handle->d ? handle->d->opts.set(option, value) : aoclsparse_status_internal_error;
    if(handle->type == aoclsparse_smat)
        return handle->s ? handle->s->opts.set(option, value) : aoclsparse_status_internal_error;
    if(handle->type == aoclsparse_zmat)
        return handle->z ? handle->z->opts.set(option, value) : aoclsparse_status_internal_error;
    if(handle->type == aoclsparse_cmat)
        return handle->c ? handle->c->opts.set(option, value) : aoclsparse_status_internal_error;
    return aoclsparse_status_invalid_value;
}

void aoclsparse_itsol_destroy(aoclsparse_itsol_handle *handle)
{
    if(handle && *handle)
    {
        delete(*handle)->s;
        delete(*handle)->d;
        delete(*handle)->c;
        delete(*handle)->z;
        delete *handle;
        *handle = nullptr;
    }
}

aoclsparse_status aoclsparse_itsol_d_init(aoclsparse_itsol_handle *handle)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    try
    {
        *handle       = new _aoclsparse_itsol_handle;
        (*handle)->type = aoclsparse_dmat;
        (*handle)->d    = new Solver<double>;
        register_options<double>((*handle)->d->opts);
    }
    catch(const std::bad_alloc &)
    {
        aoclsparse_itsol_destroy(handle);
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_itsol_s_init(aoclsparse_itsol_handle *handle)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    try
    {
        *handle         = new _aoclsparse_itsol_handle;
        (*handle)->type = aoclsparse_smat;
        (*handle)->s    = new Solver<float>;
        register_options<float>((*handle)->s->opts);
    }
    catch(const std::bad_alloc &)
    {
        aoclsparse_itsol_destroy(handle);
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}

aoclsparse_status aoclsparse_itsol_d_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n, const double *b)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_dmat)
        return aoclsparse_status_wrong_type;
    return set_rhs(*handle->d, n, b, false);
}
aoclsparse_status aoclsparse_itsol_s_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n, const float *b)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_smat)
        return aoclsparse_status_wrong_type;
    return set_rhs(*handle->s, n, b, false);
}

aoclsparse_status aoclsparse_itsol_d_rci_solve(aoclsparse_itsol_handle handle, aoclsparse_itsol_rci_job *ircomm,
                                               double **u, double **v, double *x, double rinfo[100])
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_dmat)
        return aoclsparse_status_wrong_type;
    return rci_public(handle->d, ircomm, u, v, x, rinfo);
}
aoclsparse_status aoclsparse_itsol_s_rci_solve(aoclsparse_itsol_handle handle, aoclsparse_itsol_rci_job *ircomm,
                                               float **u, float **v, float *x, float rinfo[100])
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_smat)
        return aoclsparse_status_wrong_type;
    return rci_public(handle->s, ircomm, u, v, x, rinfo);
}

aoclsparse_status aoclsparse_itsol_d_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const double *b, double *x, double rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const double *u, double *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const double *x, const double *r, double rinfo[100], void *udata),
    void *udata)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_dmat)
        return aoclsparse_status_wrong_type;
    return solve_direct<double>(handle->d, n, mat, descr, b, x, rinfo, precond, monit, udata, aoclsparse_dmat);
}
aoclsparse_status aoclsparse_itsol_s_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const float *b, float *x, float rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const float *u, float *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const float *x, const float *r, float rinfo[100], void *udata),
    void *udata)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_smat)
        return aoclsparse_status_wrong_type;
    return solve_direct<float>(handle->s, n, mat, descr, b, x, rinfo, precond, monit, udata, aoclsparse_smat);
}

aoclsparse_status aoclsparse_itsol_z_init(aoclsparse_itsol_handle *handle)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    try
    {
        *handle         = new _aoclsparse_itsol_handle;
        (*handle)->type = aoclsparse_zmat;
        (*handle)->z    = new CSolver<double>;
        register_options<double>((*handle)->z->opts);
    }
    catch(const std::bad_alloc &)
    {
        aoclsparse_itsol_destroy(handle);
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}
aoclsparse_status aoclsparse_itsol_z_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n, const aoclsparse_double_complex *b)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_zmat)
        return aoclsparse_status_wrong_type;
    return cset_rhs(*handle->z, n, reinterpret_cast<const cplx<double> *>(b), false);
}
aoclsparse_status aoclsparse_itsol_z_rci_solve(aoclsparse_itsol_handle handle, aoclsparse_itsol_rci_job *ircomm,
                                               aoclsparse_double_complex **u, aoclsparse_double_complex **v, aoclsparse_double_complex *x, double rinfo[100])
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_zmat)
        return aoclsparse_status_wrong_type;
    return crci_public(handle->z, ircomm, reinterpret_cast<cplx<double> **>(u), reinterpret_cast<cplx<double> **>(v),
                       reinterpret_cast<cplx<double> *>(x), rinfo);
}
aoclsparse_status aoclsparse_itsol_z_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const aoclsparse_double_complex *b, aoclsparse_double_complex *x, double rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const aoclsparse_double_complex *u, aoclsparse_double_complex *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const aoclsparse_double_complex *x, const aoclsparse_double_complex *r, double rinfo[100], void *udata), void *udata)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_zmat)
        return aoclsparse_status_wrong_type;
    return csolve_direct<double, aoclsparse_double_complex>(handle->z, n, mat, descr, reinterpret_cast<const cplx<double> *>(b),
                                    reinterpret_cast<cplx<double> *>(x), rinfo, precond, monit, udata, aoclsparse_zmat);
}

aoclsparse_status aoclsparse_itsol_c_init(aoclsparse_itsol_handle *handle)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    try
    {
        *handle         = new _aoclsparse_itsol_handle;
        (*handle)->type = aoclsparse_cmat;
        (*handle)->c    = new CSolver<float>;
        register_options<float>((*handle)->c->opts);
    }
    catch(const std::bad_alloc &)
    {
        aoclsparse_itsol_destroy(handle);
        return aoclsparse_status_memory_error;
    }
    return aoclsparse_status_success;
}
aoclsparse_status aoclsparse_itsol_c_rci_input(aoclsparse_itsol_handle handle, aoclsparse_int n, const aoclsparse_float_complex *b)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_cmat)
        return aoclsparse_status_wrong_type;
    return cset_rhs(*handle->c, n, reinterpret_cast<const cplx<float> *>(b), false);
}
aoclsparse_status aoclsparse_itsol_c_rci_solve(aoclsparse_itsol_handle handle, aoclsparse_itsol_rci_job *ircomm,
                                               aoclsparse_float_complex **u, aoclsparse_float_complex **v, aoclsparse_float_complex *x, float rinfo[100])
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_cmat)
        return aoclsparse_status_wrong_type;
    return crci_public(handle->c, ircomm, reinterpret_cast<cplx<float> **>(u), reinterpret_cast<cplx<float> **>(v),
                       reinterpret_cast<cplx<float> *>(x), rinfo);
}
aoclsparse_status aoclsparse_itsol_c_solve(
    aoclsparse_itsol_handle handle, aoclsparse_int n, aoclsparse_matrix mat, const aoclsparse_mat_descr descr,
    const aoclsparse_float_complex *b, aoclsparse_float_complex *x, float rinfo[100],
    aoclsparse_int precond(aoclsparse_int flag, aoclsparse_int n, const aoclsparse_float_complex *u, aoclsparse_float_complex *v, void *udata),
    aoclsparse_int monit(aoclsparse_int n, const aoclsparse_float_complex *x, const aoclsparse_float_complex *r, float rinfo[100], void *udata), void *udata)
{
    if(!handle)
        return aoclsparse_status_invalid_pointer;
    if(handle->type != aoclsparse_cmat)
        return aoclsparse_status_wrong_type;
    return csolve_direct<float, aoclsparse_float_complex>(handle->c, n, mat, descr, reinterpret_cast<const cplx<float> *>(b),
                                    reinterpret_cast<cplx<float> *>(x), rinfo, precond, monit, udata, aoclsparse_cmat);
}

} // extern "C"
