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
#define GDA_LAUNCH_CHECK()                             \
    do {                                               \
        hipError_t _e = hipGetLastError();             \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

static inline size_t gda_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline int64_t gda_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int GDA_WAVE = 64;   // CDNA4 wavefront
