// Stem tail of the dilated ResNet: FrozenBN + ReLU + 3x3 / stride 2 / pad 1 max-pool, fused (forward and backward).
// Reference: core/components/resnet.py:138-141 (bn1, relu, maxpool after the 7x7 conv), core/components/layers.py:18-23.
// HBM-bound: forward reads the conv output once (152 MB at B=8, 769x769) and writes the pooled map (38 MB) plus one
// argmax byte per output; backward reads dpool + the bytes and writes d(conv out).  The eager form is ~10 passes.
#include "mi_common.h"

namespace {

// thread = (b, ho, wo, 8 channels).  idx byte: winning tap 0..8 (first maximum in (ky,kx) scan order, like ATen), or 9 when
// the maximum of relu(bn(y)) is 0, i.e. no gradient flows (ReLU backward is x > 0).
__global__ void stem_pool_fwd_kernel(const bf16x8* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                     bf16x8* __restrict__ pool, uint8_t* __restrict__ idx, int B, int Hc, int Wc, int C8, int Hp, int Wp) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * Hp * Wp * C8) return;
    const int c8 = (int)(id % C8);
    const int wo = (int)((id / C8) % Wp), ho = (int)((id / ((long)C8 * Wp)) % Hp), b = (int)(id / ((long)C8 * Wp * Hp));
    float sc[8], sh[8], best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = scale[c8 * 8 + e];
        sh[e] = shift[c8 * 8 + e];
        best[e] = -3.0e38f;
        bi[e] = 9;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int h = 2 * ho - 1 + ky;
        if ((unsigned)h >= (unsigned)Hc) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int w = 2 * wo - 1 + kx;
            if ((unsigned)w >= (unsigned)Wc) continue;
            const bf16x8 v = y[(((long)b * Hc + h) * Wc + w) * C8 + c8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = (float)v[e] * sc[e] + sh[e];
                f = (float)(__bf16)(f > 0.f ? f : 0.f);      // the eager path stores relu(bn(y)) as bf16 before pooling
                if (f > best[e]) {
                    best[e] = f;
                    bi[e] = ky * 3 + kx;
                }
            }
        }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = (__bf16)best[e];
        idx[id * 8 + e] = (uint8_t)(best[e] > 0.f ? bi[e] : 9);
    }
    pool[id] = o;
}

// thread = (b, h, w, 8 channels) of the conv output: sums dpool over the (<= 4) windows that elected this pixel.
__global__ void stem_pool_bwd_kernel(const bf16x8* __restrict__ dpool, const uint8_t* __restrict__ idx, const float* __restrict__ scale,
                                     bf16x8* __restrict__ dy, int B, int Hc, int Wc, int C8, int Hp, int Wp) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * Hc * Wc * C8) return;
    const int c8 = (int)(id % C8);
    const int w = (int)((id / C8) % Wc), h = (int)((id / ((long)C8 * Wc)) % Hc), b = (int)(id / ((long)C8 * Wc * Hc));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ho_lo = h >> 1, ho_hi = (h + 1) >> 1, wo_lo = w >> 1, wo_hi = (w + 1) >> 1;   // windows covering h: ho_lo..ho_hi
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        if (ho >= Hp) continue;
        const int ky = h - (2 * ho - 1);
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            if (wo >= Wp) continue;
            const int tap = ky * 3 + (w - (2 * wo - 1));
            const long p = (((long)b * Hp + ho) * Wp + wo) * C8 + c8;
            const bf16x8 g = dpool[p];
            const uint2 ib = *reinterpret_cast<const uint2*>(idx + p * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned byte = ((e < 4 ? ib.x : ib.y) >> (8 * (e & 3))) & 0xffu;
                if ((int)byte == tap) acc[e] += (float)g[e];
            }
        }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)(acc[e] * scale[c8 * 8 + e]);
    dy[id] = o;
}

}  // namespace

extern "C" int mi_stem_pool_fwd(const void* y, const float* scale, const float* shift, void* pool, uint8_t* idx, int B, int Hc, int Wc, int C,
                                int Hp, int Wp, void* stream) {
    MI_REQUIRE(y && scale && shift && pool && idx && B > 0 && Hc > 0 && Wc > 0 && C > 0 && C % 8 == 0, "mi_stem_pool_fwd: bad argument");
    MI_REQUIRE(Hp == (Hc + 2 - 3) / 2 + 1 && Wp == (Wc + 2 - 3) / 2 + 1, "mi_stem_pool_fwd: pooled size must be that of a 3x3/2/1 max-pool");
    MI_REQUIRE(mi_aligned16(y) && mi_aligned16(pool) && ((uintptr_t)idx & 7) == 0, "mi_stem_pool_fwd: alignment");
    const long n = (long)B * Hp * Wp * (C / 8);
    hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)y, scale, shift,
                       (bf16x8*)pool, idx, B, Hc, Wc, C / 8, Hp, Wp);
    MI_CHECK_LAUNCH("mi_stem_pool_fwd");
    return MI_OK;
}

extern "C" int mi_stem_pool_bwd(const void* dpool, const uint8_t* idx, const float* scale, void* dy, int B, int Hc, int Wc, int C, int Hp, int Wp,
                                void* stream) {
    MI_REQUIRE(dpool && idx && scale && dy && B > 0 && Hc > 0 && Wc > 0 && C > 0 && C % 8 == 0, "mi_stem_pool_bwd: bad argument");
    MI_REQUIRE(Hp == (Hc + 2 - 3) / 2 + 1 && Wp == (Wc + 2 - 3) / 2 + 1, "mi_stem_pool_bwd: pooled size must be that of a 3x3/2/1 max-pool");
    MI_REQUIRE(mi_aligned16(dpool) && mi_aligned16(dy) && ((uintptr_t)idx & 7) == 0, "mi_stem_pool_bwd: alignment");
    const long n = (long)B * Hc * Wc * (C / 8);
    hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)dpool, idx, scale,
                       (bf16x8*)dy, B, Hc, Wc, C / 8, Hp, Wp);
    MI_CHECK_LAUNCH("mi_stem_pool_bwd");
    return MI_OK;
}

// ---- stem conv weight gradient without the library: patch matrix for the 7x7 / stride 2 / pad 3 conv ----------------------------------
// MIOpen's weight gradient of the stem conv (torch convolution_backward) is not run-to-run reproducible - its atomics are the only
// source of nondeterminism in the training step (tools/determinism_check.py: with the stem frozen, or with this path, 160 steps end in
// bit-identical parameters).  col[m][k] = x[b][2*ho - 3 + ky][2*wo - 3 + kx][c] for k = (c * 7 + ky) * 7 + kx < 147 (the order of the
// OIHW weight tensor), zero for 147 <= k < ncols (160 for the weight gradient alone, 192 when the same matrix also feeds the forward GEMM, whose K is a multiple of 64); dW[o][k] = sum_m dy[m][o] * col[m][k] is then a 1x1 weight gradient (mi_conv_wgrad,
// fixed-order slabs).  x is the bf16 channels_last input [B][H][W][3]; a thread writes 8 consecutive k of one output pixel.
namespace {
__global__ __launch_bounds__(256) void stem_im2col_kernel(const __bf16* __restrict__ x, bf16x8* __restrict__ col, int B, int H, int W, int Ho, int Wo,
                                                          int groups) {
    // grid: x = 256-thread chunks of one output row's (wo, column group) pairs, y = (b, ho): no runtime divisions per thread
    const int in_row = blockIdx.x * 256 + threadIdx.x;
    if (in_row >= Wo * groups) return;
    const int wo = in_row / groups, kg = in_row - wo * groups;
    const int ho = blockIdx.y % Ho, b = blockIdx.y / Ho;
    // first column of this thread's eight, then (c, ky, kx) advance like an odometer
    int k = kg * 8;
    int c = k / 49, r = k - c * 49, ky = r / 7, kx = r - ky * 7;
    const __bf16* img = x + (long)b * H * W * 3;
    // (eight unconditional loads with the out-of-image taps zeroed afterwards measured slower: 244 vs 203 us)
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 val = (__bf16)0.f;
        if (k + e < 147) {
            const int h = 2 * ho - 3 + ky, w = 2 * wo - 3 + kx;
            if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) val = img[((long)h * W + w) * 3 + c];
        }
        v[e] = val;
        if (++kx == 7) {
            kx = 0;
            if (++ky == 7) {
                ky = 0;
                ++c;
            }
        }
    }
    col[((long)blockIdx.y * Wo + wo) * groups + kg] = v;
}
}  // namespace

extern "C" int mi_stem_im2col(const void* x_bf16_nhwc, void* col_bf16, int B, int H, int W, int Ho, int Wo, int ncols, void* stream) {
    MI_REQUIRE(x_bf16_nhwc && col_bf16 && B > 0 && H > 0 && W > 0, "mi_stem_im2col: bad argument");
    MI_REQUIRE(ncols >= 152 && ncols % 8 == 0 && ncols <= 256, "mi_stem_im2col: ncols=%d (a multiple of 8 in [152, 256])", ncols);
    MI_REQUIRE(Ho == (H + 6 - 7) / 2 + 1 && Wo == (W + 6 - 7) / 2 + 1, "mi_stem_im2col: output size must be that of a 7x7/2/3 conv");
    MI_REQUIRE(mi_aligned16(col_bf16), "mi_stem_im2col: alignment");
    MI_REQUIRE((long)B * Ho < 65536, "mi_stem_im2col: B * Ho = %ld exceeds the grid", (long)B * Ho);
    hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)((Wo * (ncols / 8) + 255) / 256), (unsigned)(B * Ho)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x_bf16_nhwc, (bf16x8*)col_bf16, B, H, W, Ho, Wo, ncols / 8);
    MI_CHECK_LAUNCH("mi_stem_im2col");
    return MI_OK;
}
