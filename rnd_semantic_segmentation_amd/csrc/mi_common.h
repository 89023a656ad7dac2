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

// Every tuning / measurement switch of the library (MI_* environment variables), read ONCE per process under std::call_once
// (csrc/elementwise.hip): no getenv on a launch path, no unsynchronised static caches - mi355seg.h promises re-entrancy.
// Defaults are the product configuration; the switches exist for tools/ (A/B measurements) and for tests that force a kernel.
struct MiSwitches {
    int igemm_staged;      // MI_IGEMM_STAGED     1: residual / output tiles through LDS with row-contiguous lanes (igemm_nt)
    int igemm_pp;          // MI_IGEMM_PP         1: long contractions take the wide-tile ping-pong loop (igemm_pp)
    int igemm_pp_mink;     // MI_IGEMM_PP_MINK    512: smallest 1x1 contraction routed there
    int igemm_mt;          // MI_IGEMM_MT         0: tile height chosen by the cost model (else 4..6 x 32 rows)
    int igemm_bn;          // MI_IGEMM_BN         0: 128-wide tiles (256: the experiment tile, builds with -DMI_EXPERIMENTS only)
    int igemm_pref;        // MI_IGEMM_PREF       1: epilogue operands prefetched before the main loop
    int pp_korder;         // MI_IGEMM_PP_KORDER  1: channel-chunk-major contraction of a 3x3 (0: tap-major, bit-equal to igemm_nt)
    int pp_loop;           // MI_IGEMM_PP_LOOP    0: two-group ping-pong main loop, 1: rolling fragment ring (both waves of a SIMD stream MFMAs)
    int igemm_pw;          // MI_IGEMM_PW         0: shared-window 3x3 kernel off (-DMI_EXPERIMENTS builds only)
    int wgrad_q3_slots;    // MI_WGRAD_Q3_SLOTS   512: workgroup slots the fused-row 3x3 weight gradient's split fills when the launch runs alone (mi_conv_wgrad)
    int wgrad_q3_slots_beside;   // MI_WGRAD_Q3_SLOTS_BESIDE   448: ... when it runs beside a data-gradient chain (mi_conv_wgrad_partial)
    int wgrad_s4_slots;    // MI_WGRAD_S4_SLOTS   512: workgroup slots the 1x1 weight-gradient split picker plans for
    int wgrad_ti256;       // MI_WGRAD_TI256      -1: 128 x 256 weight-gradient tile by rule (0 never, 1 always)
    int wgrad_p3;          // MI_WGRAD_P3         0: 8-wave fused-row 3x3 weight gradient off (-DMI_EXPERIMENTS builds only)
    int wgrad_q3;          // MI_WGRAD_Q3         1: 4-wave fused-row 3x3 weight gradient by rule (0 never, 2 whenever possible)
    int wgrad_s4;          // MI_WGRAD_S4         1: deep-stream 1x1 weight gradient
    int gconv_bn128;       // MI_GCONV_BN128      0: general conv keeps 64-wide tiles (1: 128-wide where they fit)
    int gconv_kc;          // MI_GCONV_KC         0: K chunk of the general conv by rule (32 | 64: forced)
    int gconv_remap;       // MI_GCONV_REMAP      1: general conv / weight gradient walk their tiles XCD-contiguous (0: plain grid order)
    int gconv_ks2_wgs;     // MI_GCONV_KS2_WGS    320: general-conv launches of at most this many workgroups run two wave groups over the K chunks (0: never)
    int gconv_kc32_wgs;    // MI_GCONV_KC32_WGS   1536: launches of at least this many workgroups take 32-channel K chunks
    int gconv_bn32_wgs;    // MI_GCONV_BN32_WGS   256: launches of fewer 64-wide workgroups take 32-wide tiles
    int gconv_bn_any;      // MI_GCONV_BN_ANY     1: tile width per launch from {16, 32, 64, 80, 112} by the cost model (0: 32 / 64 only)
    int gconv_bn_c;        // MI_GCONV_BN_C       64: the cost model's fixed cost per column tile, in columns
    int gconv_bn_force;    // MI_GCONV_BN_FORCE   0: (measurement) one tile width for every launch
    int gconv3_wgs;        // MI_GCONV3_WGS       512: three-column stride-1 convs of at least this many 64-wide workgroups take the kernel-row window kernel (0 never, 1 all: tests)
    int gconv_dbg;         // MI_GC_DBG           0: (-DMI_EXPERIMENTS builds only) bit 0 skip the main loop, bit 1 the statistics, bit 2 the stores
    int gw_dbg;            // MI_GW_DBG           0: (-DMI_EXPERIMENTS builds only) bit 0: weight gradient without its main loop
    int gwm_steps;         // MI_GWM_STEPS        48: batched weight gradients: K steps of 64 pixels per workgroup
    int gwm_fused3;        // MI_GWM_FUSED3       1: batched weight gradients: fused kernel rows wherever the geometry allows (0: the one-conv rule)
    int gwgrad3;           // MI_GWGRAD3          1: general weight gradient of three-column kernels as one fused kernel row per workgroup from 65 536 pixels up (0: per tap, 2: always)
    int p3_dbg;            // MI_P3_DBG           0 (-DMI_EXPERIMENTS builds only)
    int pp_trace_wg;       // MI_PP_TRACE_WG      0 (-DMI_PP_TRACE builds only)
};
const MiSwitches& mi_sw();

// Epilogue statistics of a conv launch (MI_EPI_STATS): where the per-row-tile partial sums go, the pilot they are taken against, and (out) how
// many partial rows the launch wrote - the tile height is the launcher's choice.  Internal linkage between igemm_nt.hip, igemm_pp.hip, batchnorm.hip.
struct MiConvStats {
    float* partial;        // [rows][2][N] fp32, rows <= mi_conv_gemm_stats_workspace's bound
    const float* pilot;    // [N]
    int nparts;            // out
};
int mi_conv_gemm_impl(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride, int pad,
                      int dil, int gather_mode, const float* scale, const float* bias, const void* res, const void* msk, void* mask_out, int flags,
                      int zgw, float alpha, void* stream, MiConvStats* st);
int mi_conv_gemm_pp_impl(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride, int pad,
                         int dil, int gather_mode, const float* scale, const float* bias, const void* res, const void* msk, void* mask_out, int flags,
                         int zgw, float alpha, int mtg, void* stream, MiConvStats* st);
// BatchNorm finalize arguments (mi_bn_finalize's), for launches that finalize in the same kernel as their last reduction level
struct MiBnFinal {
    const float* pilot;
    double count;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    long long* num_batches_tracked;
    float momentum, eps;
    float* out4;
};
// fixed-order sum of `nparts` partial rows [part][2][C] into s1[C], s2[C] in two launches (batchnorm.hip); tmp: mi_bn_reduce_tmp_floats(nparts, C)
// floats; fin != NULL: the second launch also finalizes the statistics
size_t mi_bn_reduce_tmp_floats(int nparts, int C);
int mi_bn_reduce_partials(const float* partial, int nparts, int C, float* tmp, float* s1, float* s2, const MiBnFinal* fin, void* stream);

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

// ---- BatchNorm finalize arithmetic shared by the separate finalize kernel (gnet.hip) and the conv's in-launch finalize (gconv.hip): every multiply / add an
// explicitly rounded operation, so that the two give the same bits whatever fused-multiply-add contraction the compiler applies around them (the in-launch
// version moved into a device function in round 4 and its running_var went one ulp off the kernel's).
struct MiBnFin {
    float mean, invstd, scale, shift;
    double var;
};
__device__ __forceinline__ MiBnFin mi_bn_finalize_channel(double s1, double s2, double count, float eps, float gamma, float beta) {
    MiBnFin r;
    const double mean = s1 / count;
    double var = __dsub_rn(s2 / count, __dmul_rn(mean, mean));
    if (var < 0.0) var = 0.0;
    r.var = var;
    r.mean = (float)mean;
    r.invstd = (float)(1.0 / sqrt(__dadd_rn(var, (double)eps)));
    r.scale = __fmul_rn(gamma, r.invstd);
    r.shift = __fsub_rn(beta, __fmul_rn(r.mean, r.scale));
    return r;
}
__device__ __forceinline__ float mi_bn_running(float old, float momentum, float value) {      // (1 - momentum) * old + momentum * value
    return __fadd_rn(__fmul_rn(1.f - momentum, old), __fmul_rn(momentum, value));
}
__device__ __forceinline__ float mi_bn_unbiased(double var, double count) { return (float)(count > 1.0 ? __dmul_rn(var, count) / (count - 1.0) : var); }

// ---- in-launch second-level reductions (the last-arriving workgroup of a launch combines the other workgroups' small partial results) ----
// The sc1 form of the split-K recipe of cdna_hip_programming.md section 5 / section 6 Guideline 16: the XCDs' L2s are not coherent with each
// other, so every partial value is stored WRITE-THROUGH (agent-scope relaxed atomic store = global_store ... sc1: no release fence, which would
// write back every dirty line of the XCD's L2 - the launch's own output tiles), every storing wave drains its stores, ONE lane draws a ticket
// with an agent-scope fetch_add, and the workgroup that draws the last ticket reads every partial value with sc1 loads (which bypass its L1).
// The ticket word must be zero before the first launch (the host allocates it zeroed); the last arriver resets it, so the buffer is reusable by
// the next launch on the same stream and under HIP-graph replay.  Only worth it while the partials per reducer are a few tens of KB: the
// callers fall back to a second launch above that.
typedef __attribute__((address_space(1))) unsigned mi_gu32;
typedef __attribute__((address_space(1))) unsigned long long mi_gu64;
__device__ __forceinline__ void mi_st_sc1(float* p, float v) {
    __hip_atomic_store((mi_gu32*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float mi_ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load((mi_gu32*)const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void mi_st2_sc1(float* p, float a, float b) {          // p 8-byte aligned
    __hip_atomic_store((mi_gu64*)p, ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mi_ld2_sc1(const float* p, float& a, float& b) {
    const unsigned long long x = __hip_atomic_load((mi_gu64*)const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __uint_as_float((unsigned)x);
    b = __uint_as_float((unsigned)(x >> 32));
}
// Call with every thread of the workgroup, after the workgroup's sc1 stores.  `lds_flag`: one int of LDS nobody else uses around the call.
// Returns true in every thread of the workgroup that arrived last among `total`.
__device__ __forceinline__ bool mi_last_arriver(unsigned* ticket, unsigned total, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add((mi_gu32*)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == total - 1u;
        if (last) __hip_atomic_store((mi_gu32*)ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_flag = last ? 1 : 0;
    }
    __syncthreads();
    return *lds_flag != 0;
}
// In the last-arriving workgroup, before it reads the other workgroups' write-through partials with PLAIN loads (which the compiler may issue
// back to back - a chain of relaxed atomic loads is issued one round trip at a time: measured +16 us on a 61-tile finalize): one agent-scope
// acquire (buffer_inv sc1) drops this CU's stale lines, then every wave proceeds.
__device__ __forceinline__ void mi_acquire_partials() {
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
constexpr int MI_INLAUNCH_MAX_PARTS = 64;      // partial rows one reducer thread adds per channel
