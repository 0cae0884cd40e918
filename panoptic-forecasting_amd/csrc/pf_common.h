// Internal helpers shared by the translation units of libpfhip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/pfhip.h"

namespace pf {

// thread-local last-error text (pf_last_error)
char *error_buffer();
int fail(int code, const char *fmt, ...);

#define PF_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return pf::fail(PF_EHIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define PF_LAUNCH_CHECK(what)                                                                \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) return pf::fail(PF_EHIP, "%s: %s", what, hipGetErrorString(_e)); \
    } while (0)

// A/B and debugging switches read from the environment (PF_STEM_V3, PF_SYNC_OPS, PF_PROBE, ...) exist only in libraries
// built with -DPF_AB=1 (the probe / A-B targets of the Makefile); the shipped libpfhip.so reads no environment variable
#ifndef PF_AB
#define PF_AB 0
#endif
static inline const char *ab_env(const char *name) { return PF_AB ? getenv(name) : nullptr; }

// pf_fill.hip: zero fill / device-to-device copy as kernels (a captured call of the library holds kernel nodes only - see there)
int launch_zero_fill(void *p, size_t bytes, hipStream_t s);
int launch_copy(void *dst, const void *src, size_t bytes, hipStream_t s);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace pf
