// Small HBM-bound kernels: fused SGD step on flat buffers, ReLU-backward mask, FrozenBN fold, error plumbing.
#include "mi_common.h"
#include <mutex>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

int mi_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const MiSwitches& mi_sw() {
    static MiSwitches sw;
    static std::once_flag once;
    std::call_once(once, [] {
        auto env = [](const char* name, int dflt) {
            const char* e = getenv(name);
            return e ? atoi(e) : dflt;
        };
        sw.igemm_staged = env("MI_IGEMM_STAGED", 1);
        sw.igemm_pp = env("MI_IGEMM_PP", 1);
        sw.igemm_pp_mink = env("MI_IGEMM_PP_MINK", 512);
        sw.igemm_mt = env("MI_IGEMM_MT", 0);
        sw.igemm_bn = env("MI_IGEMM_BN", 0);
        sw.igemm_pref = env("MI_IGEMM_PREF", 1);
        sw.pp_korder = env("MI_IGEMM_PP_KORDER", 1);
#ifdef MI_EXPERIMENTS
        sw.pp_loop = env("MI_IGEMM_PP_LOOP", 0);
#else
        sw.pp_loop = 0;
#endif
        sw.igemm_pw = env("MI_IGEMM_PW", 0);
        sw.wgrad_q3_slots = env("MI_WGRAD_Q3_SLOTS", 512);
        sw.wgrad_q3_slots_beside = env("MI_WGRAD_Q3_SLOTS_BESIDE", 448);
        sw.wgrad_s4_slots = env("MI_WGRAD_S4_SLOTS", 512);
        if (sw.wgrad_s4_slots < 64) sw.wgrad_s4_slots = 512;
        sw.wgrad_ti256 = env("MI_WGRAD_TI256", -1);
        sw.wgrad_p3 = env("MI_WGRAD_P3", 0);
        sw.wgrad_q3 = env("MI_WGRAD_Q3", 1);
        sw.wgrad_s4 = env("MI_WGRAD_S4", 1);
        sw.gconv_bn128 = env("MI_GCONV_BN128", 0);
        sw.gconv_kc = env("MI_GCONV_KC", 0);
        sw.gconv_remap = env("MI_GCONV_REMAP", 1);
        sw.gconv_ks2_wgs = env("MI_GCONV_KS2_WGS", 320);
        sw.gconv_kc32_wgs = env("MI_GCONV_KC32_WGS", 1536);
        sw.gconv_bn32_wgs = env("MI_GCONV_BN32_WGS", 256);
        sw.gconv_bn_any = env("MI_GCONV_BN_ANY", 1);
        sw.gconv_bn_c = env("MI_GCONV_BN_C", 64);
        sw.gconv_bn_force = env("MI_GCONV_BN_FORCE", 0);
        sw.gconv3_wgs = env("MI_GCONV3_WGS", 512);
#ifdef MI_EXPERIMENTS          // measurement switches: experiment builds only (the product library ignores the variables)
        sw.gconv_dbg = env("MI_GC_DBG", 0);
        sw.gw_dbg = env("MI_GW_DBG", 0);
#else
        sw.gconv_dbg = sw.gw_dbg = 0;
#endif
        sw.gwm_steps = env("MI_GWM_STEPS", 48);
        if (sw.gwm_steps < 1) sw.gwm_steps = 48;
        sw.gwm_fused3 = env("MI_GWM_FUSED3", 1);
        sw.gwgrad3 = env("MI_GWGRAD3", 1);
#ifdef MI_EXPERIMENTS
        sw.p3_dbg = env("MI_P3_DBG", 0);
#else
        sw.p3_dbg = 0;
#endif
        sw.pp_trace_wg = env("MI_PP_TRACE_WG", 0);
    });
    return sw;
}

extern "C" int mi_version(void) { return MI355SEG_VERSION; }
extern "C" const char* mi_last_error(void) { return g_err; }

namespace {

// torch.optim.SGD (reference core/trainers/aspp_trainer.py:25-26,94-95): g' = g + wd*p; buf = mu*buf + g'; p -= lr*buf
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n4, size_t n, float lr,
                           float mu, float wd) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 bv = reinterpret_cast<f32x4*>(buf)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] + wd * pv[e];
            bv[e] = mu * bv[e] + gg;
            pv[e] = pv[e] - lr * bv[e];
        }
        reinterpret_cast<f32x4*>(buf)[i] = bv;
        reinterpret_cast<f32x4*>(p)[i] = pv;
    }
    // tail (n % 4 elements) by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float gg = g[i] + wd * p[i];
        const float b = mu * buf[i] + gg;
        buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}

// the same update with the hyper-parameters read from device memory: a captured HIP graph replays with whatever learning rate
// the host wrote into `hyper` before the launch (a by-value kernel argument would be frozen into the graph)
__global__ void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n4, size_t n,
                               const float* __restrict__ hyper) {
    const float lr = hyper[0], mu = hyper[1], wd = hyper[2];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 bv = reinterpret_cast<f32x4*>(buf)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] + wd * pv[e];
            bv[e] = mu * bv[e] + gg;
            pv[e] = pv[e] - lr * bv[e];
        }
        reinterpret_cast<f32x4*>(buf)[i] = bv;
        reinterpret_cast<f32x4*>(p)[i] = pv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float gg = g[i] + wd * p[i];
        const float b = mu * buf[i] + gg;
        buf[i] = b;
        p[i] = p[i] - lr * b;
    }
}

__global__ void relu_mask_kernel(const bf16x8* __restrict__ x, const bf16x8* __restrict__ m, bf16x8* __restrict__ y, size_t n8) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const bf16x8 xv = x[i], mv = m[i];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ((float)mv[e] > 0.f) ? xv[e] : (__bf16)0.f;
        y[i] = o;
    }
}

// reference core/components/layers.py:18-20 (no eps)
__global__ void bn_fold_kernel(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = w[i] * (1.0f / sqrtf(var[i]));
    scale[i] = s;
    shift[i] = b[i] - mean[i] * s;
}


extern "C" int mi_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay, void* stream) {
    MI_REQUIRE(p && g && buf && n > 0, "mi_sgd_step: bad argument");
    MI_REQUIRE(mi_aligned16(p) && mi_aligned16(g) && mi_aligned16(buf), "mi_sgd_step: flat buffers must be 16-byte aligned");
    const size_t n4 = n >> 2;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, n4, n, lr, momentum, weight_decay);
    MI_CHECK_LAUNCH("mi_sgd_step");
    return MI_OK;
}

extern "C" int mi_sgd_step_dev(float* p, const float* g, float* buf, size_t n, const float* hyper, void* stream) {
    MI_REQUIRE(p && g && buf && hyper && n > 0, "mi_sgd_step_dev: bad argument");
    MI_REQUIRE(mi_aligned16(p) && mi_aligned16(g) && mi_aligned16(buf), "mi_sgd_step_dev: flat buffers must be 16-byte aligned");
    const size_t n4 = n >> 2;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sgd_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, n4, n, hyper);
    MI_CHECK_LAUNCH("mi_sgd_step_dev");
    return MI_OK;
}

__global__ void relu_mask_bits_kernel(const bf16x8* __restrict__ x, const uint8_t* __restrict__ m, bf16x8* __restrict__ y, size_t n8) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const bf16x8 xv = x[i];
        const unsigned bits = m[i];              // 8 elements <-> one byte of the little-endian uint16 words
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ((bits >> e) & 1u) ? xv[e] : (__bf16)0.f;
        y[i] = o;
    }
}

}  // namespace

extern "C" int mi_relu_mask(const void* x, const void* msk, void* y, size_t n, int bits, void* stream) {
    MI_REQUIRE(x && msk && y && n > 0 && n % 8 == 0, "mi_relu_mask: bad argument (n %% 8 == 0)");
    MI_REQUIRE(mi_aligned16(x) && mi_aligned16(y) && (bits || mi_aligned16(msk)), "mi_relu_mask: alignment");
    const size_t n8 = n >> 3;
    size_t blocks = (n8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (bits) {
        MI_REQUIRE(n % 16 == 0, "mi_relu_mask: packed mask needs n %% 16 == 0");
        hipLaunchKernelGGL(relu_mask_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x,
                           (const uint8_t*)msk, (bf16x8*)y, n8);
        MI_CHECK_LAUNCH("mi_relu_mask (bits)");
        return MI_OK;
    }
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, (const bf16x8*)msk,
                       (bf16x8*)y, n8);
    MI_CHECK_LAUNCH("mi_relu_mask");
    return MI_OK;
}

extern "C" int mi_frozen_bn_fold(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift, int n,
                                 void* stream) {
    MI_REQUIRE(w && b && mean && var && scale && shift && n > 0, "mi_frozen_bn_fold: bad argument");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, b, mean, var, scale, shift, n);
    MI_CHECK_LAUNCH("mi_frozen_bn_fold");
    return MI_OK;
}
