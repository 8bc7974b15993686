// Shared helpers for libgda_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/gda_hip.h"

#define GDA_HIP_TRY(expr)                              \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

// kernel launches do not return a status: pick up launch-configuration errors here
// Measurement aid (tools/whatif_cfgA.sh): with the library built with -DGDA_MEASUREMENT_AIDS (PYGDA_AMD_MEASUREMENT_AIDS=1 for
// pygda_amd/_build.py), PYGDA_AMD_DBG_SKIP="k_slab_sum,k_transpose" leaves the named launches out -- the results are then
// WRONG; what is read off is what the step would cost without those launches, before anything is built to remove them.
// The release build (the default) compiles none of it: the variable has no effect on the product library.
#ifdef GDA_MEASUREMENT_AIDS
#include <cstdio>
#include <cstdlib>
#include <cstring>
static inline bool gda_dbg_skip(const char* name) {
    static const char* env = [] {
        const char* e = std::getenv("PYGDA_AMD_DBG_SKIP");
        if (e && *e) std::fprintf(stderr, "libgda_hip.so: PYGDA_AMD_DBG_SKIP=%s -- named launches are LEFT OUT, results are wrong\n", e);
        return e;
    }();
    if (!env || !*env) return false;
    const size_t n = std::strlen(name);
    for (const char* p = std::strstr(env, name); p; p = std::strstr(p + 1, name))
        if ((p == env || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}
#else
static inline constexpr bool gda_dbg_skip(const char*) { return false; }
#endif
#define GDA_UNLESS_SKIPPED(name) if (!gda_dbg_skip(name))

#define GDA_LAUNCH_CHECK()                             \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

// Opt a kernel into more than 64 KB of dynamic LDS.  The attribute is per DEVICE: the flag word holds one bit per device
// ordinal (a process that drives a second GPU configures the kernel there too); racing first calls set the same value.
#include <atomic>
#define GDA_LDS_ATTR_ONCE(func, bytes)                                                                         \
    do {                                                                                                       \
        static std::atomic<uint64_t> _done{0};                                                                 \
        int _dev = 0;                                                                                          \
        GDA_HIP_TRY(hipGetDevice(&_dev));                                                                      \
        const uint64_t _bit = 1ull << (_dev & 63);                                                             \
        if (!(_done.load(std::memory_order_relaxed) & _bit)) {                                                 \
            GDA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(func),                               \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));        \
            _done.fetch_or(_bit, std::memory_order_relaxed);                                                   \
        }                                                                                                      \
    } while (0)

static inline size_t gda_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline int64_t gda_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int GDA_WAVE = 64;   // CDNA4 wavefront
