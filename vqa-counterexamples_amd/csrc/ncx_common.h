// ncx_common.h -- shared definitions of the NeuralCX HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/neuralcx.h"

namespace ncx {

#define NCX_HIP_TRY(expr)                                 \
    do {                                                  \
        hipError_t e__ = (expr);                          \
        if (e__ != hipSuccess) return (int)e__;           \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Counter-based dropout generator.  Restated on the host in oracle/ncx_oracle.py:dropout_keep_mask
// (test infrastructure) -- keep the two in sync.  keep <=> u >= p, u = top 24 bits / 2^24.
__host__ __device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ bool dropout_keep(unsigned seed_lo, unsigned seed_hi, unsigned layer,
                                                      unsigned long long idx, float p) {
    unsigned x = mix32((unsigned)(idx & 0xFFFFFFFFull) ^ seed_lo);
    x = mix32(x ^ (unsigned)(idx >> 32) ^ seed_hi ^ (layer * 0x9E3779B9u));
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= p;
}

}  // namespace ncx

// The planner's experiment hooks (NCX_CFG_<id>, NCX_SPLIT_<id>, NCX_NO_FAST, NCX_PERSISTENT, NCX_BF16_NT_CFG, ...) are
// only consulted when NCX_EXPERIMENT=1 is set: the steady-state path then costs one getenv per call instead of ~25.
#include <stdlib.h>
namespace ncx {
static inline bool experiment_hooks_on() { const char* e = getenv("NCX_EXPERIMENT"); return e && e[0] == '1'; }
static inline const char* hook_env(const char* name) { return experiment_hooks_on() ? getenv(name) : nullptr; }
}  // namespace ncx

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: one bit per device (hipGetDevice) in an
// atomic mask per kernel instantiation, so a host that drives several GPUs from one process (or several host threads)
// opts every device in exactly once.
#include <atomic>
namespace ncx {
typedef std::atomic<unsigned long long> DevMask;
static inline hipError_t set_max_lds_once(DevMask& done, const void* fn, int lds_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}
}  // namespace ncx
