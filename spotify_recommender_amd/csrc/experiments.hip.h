// experiments.hip.h — every hook that exists for MEASURING the engine, in one place.
//
// The product build defines none of the switches below: all macros here then expand to nothing and no tuning knob is
// read from the environment.  The tools/ scripts build instrumented copies of the library under gpurun_out/ (never
// the product library) with
//   -DMI355REC_PHASE_CLOCK   100 MHz wall-clock stamps per workgroup and phase (tools/phase_clock.py, hm_clock.py, ...)
//   -DMI355REC_EXPERIMENTS   environment knobs for A/B runs (MI355REC_EXP_*) and the routes that only exist for A/B
//                            (the single-query scan over the fp16 replica, the 8-bit front end of the multi-query pass)
//   -DMI355_HM_EXP=n         timing experiments of the multi-query pass (wrong results on purpose, tools/hm_exp.sh)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace mi355 {

#ifdef MI355REC_PHASE_CLOCK
__device__ unsigned long long g_phase_clock[1024 * 8];   // [workgroup][phase]
// phases of every workgroup of a scan (row = blockIdx.x)
#define MI355REC_PHASE(i)                                                                                 \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 1023) g_phase_clock[blockIdx.x * 8 + (i)] = wall_clock64();  \
    } while (0)
// a duration (or a count) accumulated into a slot by the calling thread: MI355REC_PHASE_T0(t) ... MI355REC_PHASE_ADD(i, t)
#define MI355REC_PHASE_ZERO(i)                                                            \
    do {                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 1023) g_phase_clock[blockIdx.x * 8 + (i)] = 0; \
    } while (0)
#define MI355REC_PHASE_T0(t) const unsigned long long t = wall_clock64()
// by lane 0 of EVERY wave of the workgroup: the workgroup's slot takes the LARGEST duration any of its waves has
// accumulated (the wave passes its own running total), resp. the sum of the waves' counts, resp. the latest stamp
#define MI355REC_PHASE_WAVE_MAX(i, total)                                                                    \
    do {                                                                                                     \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 1023) atomicMax(&g_phase_clock[blockIdx.x * 8 + (i)], (total)); \
    } while (0)
#define MI355REC_PHASE_WAVE_COUNT(i)                                                                    \
    do {                                                                                                \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 1023) atomicAdd(&g_phase_clock[blockIdx.x * 8 + (i)], 1ull); \
    } while (0)
#define MI355REC_PHASE_WAVE_LAST(i)                                                                              \
    do {                                                                                                         \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 1023) atomicMax(&g_phase_clock[blockIdx.x * 8 + (i)], wall_clock64()); \
    } while (0)
// phases of ONE merge (merge_body in the workgroup with blockIdx.x == 0: merge_notify_kernel), kept in row 1023
#define MI355REC_MPHASE(i)                                                                       \
    do {                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_phase_clock[1023 * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MI355REC_PHASE(i) \
    do {                  \
    } while (0)
#define MI355REC_PHASE_ZERO(i) \
    do {                       \
    } while (0)
#define MI355REC_PHASE_T0(t) \
    do {                     \
    } while (0)
#define MI355REC_PHASE_WAVE_MAX(i, total) \
    do {                                  \
    } while (0)
#define MI355REC_PHASE_WAVE_COUNT(i) \
    do {                             \
    } while (0)
#define MI355REC_PHASE_WAVE_LAST(i) \
    do {                            \
    } while (0)
#define MI355REC_MPHASE(i) \
    do {                   \
    } while (0)
#endif
#define MI355REC_KPHASE(i) MI355REC_PHASE(i)

#ifndef MI355_HM_EXP
#define MI355_HM_EXP 0
#endif

// Host side: `MI355REC_EXP_INT(var, "NAME", lo, hi)` overrides `var` from the environment variable NAME when it
// parses to an integer in [lo, hi] — in MI355REC_EXPERIMENTS builds only.
#ifdef MI355REC_EXPERIMENTS
#define MI355REC_EXP_INT(var, name, lo, hi)                              \
    do {                                                                 \
        if (const char* e_ = std::getenv(name)) {                        \
            const long v_ = std::atol(e_);                               \
            if (v_ >= (lo) && v_ <= (hi)) (var) = static_cast<decltype(var)>(v_); \
        }                                                                \
    } while (0)
#define MI355REC_EXP_FLAG(name) (std::getenv(name) != nullptr)
#else
#define MI355REC_EXP_INT(var, name, lo, hi) \
    do {                                    \
    } while (0)
#define MI355REC_EXP_FLAG(name) false
#endif

}  // namespace mi355
