// Shared device/host helpers for libmi355seg (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/mi355seg.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

int mi_set_error(int code, const char* fmt, ...);

#define MI_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) return mi_set_error(MI_EINVAL, __VA_ARGS__); \
    } while (0)

#define MI_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) return mi_set_error(MI_EHIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// Kernels that need more than 64 KiB of dynamic LDS must be allowed to, once per (kernel, device): the attribute is
// per device, so a process-wide "done" flag would leave the second device of a process launching without it.  One bit
// per device ordinal; two racing threads both set the attribute (idempotent).
static inline void mi_allow_dynamic_lds(const void* kern, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}
constexpr int MI_LDS_MAX = 160 * 1024;

static inline bool mi_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// XCD-aware bijective remap of a linear workgroup id: consecutive logical ids land on the same XCD
// (blocks b and b+8 share an XCD under round-robin dispatch), so tiles sharing an operand panel share an L2.
__device__ __forceinline__ int mi_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ float mi_bf16_to_f32(__bf16 v) { return (float)v; }
