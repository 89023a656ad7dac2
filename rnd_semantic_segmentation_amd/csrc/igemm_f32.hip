// Exact-fp32 evaluation path for gfx950: implicit-GEMM convolution on the f32-input MFMA, the 7x7/2 stem, the max-pool.
//
// Why it exists: the reference computes in fp32, and BASELINE.json asks for logits within 1e-3 relative,
// argmax masks identical and mIoU equal.  The bf16 training engine cannot give that through 33 bottlenecks (1e-2 of the logit
// range on a random net), so test.py / ASPPTester run THIS path: fp32 NHWC activations, fp32 weights, fp32 accumulate.
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain (MI355X_MICROARCH.md, Matrix cores): one rounding per
// product, no reduced-precision operand, 157 TFLOP/s peak - 1/16 of the bf16 rate, which an evaluation pass can afford
// (732 GFLOP for one 512x1024 image).
//
//   out[m][n] = epi( sum_t sum_c A[src(m,t)][c] * Wp[t][n][c] )      replaces nn.Conv2d + FrozenBatchNorm2d + ReLU + residual of
//   reference core/components/resnet.py:93-113 and the four biased dilated convs of classifiers/aspp/classifier.py:26-29.
//
// Structure: no LDS.  A wave owns 64 pixels x 64 channels (2 x 2 MFMA tiles of 32 x 32); a lane (r = lane & 31, h = lane >> 5)
// loads, per 16-channel K-step, the 8 contiguous channels 8h..8h+7 of its pixel / weight row (two 16-B loads per tile: the two
// halves of a wave cover 64 contiguous bytes of a row) and feeds element kk of both fragments to MFMA step kk - the k <-> (h, kk)
// map is the same on both operands, so the sum runs over all 16 channels.  Four waves (2 x 2) of a workgroup share the rows of
// a 128 x 128 block through the vector L1.  Fragments for K-step s+1 are in flight while step s computes.
// D rows = pixels (MFMA "A" operand), D columns = channels: one accumulator register across lanes 0-31 is 32 contiguous floats
// of an NHWC output row (128-B stores).
#include "mi_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

struct F32Params {
    const float* A;
    const float* Wp;
    float* out;
    const float* scale;
    const float* bias;
    const float* res;
    int M, N, Ca, T;
    int Ho, Wo, Ha, Wa;
    int ksz, stride, pad, dil;
    int flags;
};

struct Frag {
    f32x4 lo, hi;      // channels 8h .. 8h+3, 8h+4 .. 8h+7 of the lane's row
};

__device__ __forceinline__ Frag load_frag(const float* p, bool ok) {
    Frag f;
    if (ok) {
        f.lo = *reinterpret_cast<const f32x4*>(p);
        f.hi = *reinterpret_cast<const f32x4*>(p + 4);
    } else {
        f.lo = f32x4{0.f, 0.f, 0.f, 0.f};
        f.hi = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return f;
}

__device__ __forceinline__ float frag_at(const Frag& f, int kk) { return kk < 4 ? f.lo[kk] : f.hi[kk - 4]; }

__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(F32Params p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int n_tiles = (p.N + 127) / 128;
    const int tile = mi_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
    const int m0 = mt * 128 + wm * 64, n0 = nt * 128 + wn * 64;
    if (n0 >= p.N) return;                       // a wave whose 64 columns are all padding (N = 19: three of the four waves' columns)

    // the lane's two pixel rows and two weight rows
    int a_b[2], a_ho[2], a_wo[2];
    bool a_ok[2];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 32 + r;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        a_b[i] = mm / HoWo;
        const int rem = mm - a_b[i] * HoWo;
        a_ho[i] = rem / p.Wo;
        a_wo[i] = rem - a_ho[i] * p.Wo;
    }
    bool w_ok[2];
    const float* w_row[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 32 + r;
        w_ok[j] = n < p.N;
        w_row[j] = p.Wp + (long)(w_ok[j] ? n : 0) * p.Ca + h * 8;
    }
    const bool j1_live = n0 + 32 < p.N;          // wave-uniform: skip the second column tile when it is all padding

    const int cpt = p.Ca >> 4;                   // K-steps per tap
    const int nk = p.T * cpt;
    const long w_tap = (long)p.N * p.Ca;

    auto load_step = [&](int ks, Frag (&af)[2], Frag (&bf)[2]) {
        const int t = ks / cpt, c0 = (ks - t * cpt) << 4;
        const int ky = t / p.ksz, kx = t - ky * p.ksz;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hs = a_ho[i] * p.stride + ky * p.dil - p.pad, ws = a_wo[i] * p.stride + kx * p.dil - p.pad;
            const bool ok = a_ok[i] && (unsigned)hs < (unsigned)p.Ha && (unsigned)ws < (unsigned)p.Wa;
            const long off = (((long)a_b[i] * p.Ha + (ok ? hs : 0)) * p.Wa + (ok ? ws : 0)) * p.Ca + c0 + h * 8;
            af[i] = load_frag(p.A + off, ok);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = load_frag(w_row[j] + (long)t * w_tap + c0, w_ok[j] && (j == 0 || j1_live));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](const Frag (&af)[2], const Frag (&bf)[2]) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag_at(af[i], kk), frag_at(bf[0], kk), acc[i][0], 0, 0, 0);
                if (j1_live) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag_at(af[i], kk), frag_at(bf[1], kk), acc[i][1], 0, 0, 0);
            }
        }
    };

    Frag a0[2], b0[2], a1[2], b1[2];
    load_step(0, a0, b0);
    int ks = 0;
    for (; ks + 2 <= nk; ks += 2) {              // two steps per iteration: named register sets, no runtime-indexed arrays
        load_step(ks + 1, a1, b1);
        compute(a0, b0);
        if (ks + 2 < nk) load_step(ks + 2, a0, b0);
        compute(a1, b1);
    }
    if (ks < nk) compute(a0, b0);

    // ---- epilogue: register e of tile (i, j): pixel m0 + 32 i + (e & 3) + 8 (e >> 2) + 4 h, channel n0 + 32 j + r -------------
    const int flags = p.flags;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 32 + r;
        if (n >= p.N) continue;
        float sc = 1.f, bi = 0.f;
        if (flags & MI_EPI_SCALE_BIAS) {
            if (p.scale) sc = p.scale[n];
            bi = p.bias[n];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = acc[i][j][e];
                // torch evaluates x * scale + bias as two rounded operations (layers.py:21-23): no fused multiply-add here
                if (flags & MI_EPI_SCALE_BIAS) v = __fadd_rn(p.scale ? __fmul_rn(v, sc) : v, bi);
                const long o = (long)m * p.N + n;
                if (flags & MI_EPI_RESIDUAL) v = __fadd_rn(v, p.res[o]);
                if (flags & MI_EPI_RELU) v = v > 0.f ? v : 0.f;
                p.out[o] = v;
            }
        }
    }
}

// wp[t][o][i] = w[o][i][t] (fp32, no rounding)
__global__ void pack_f32_kernel(const float* __restrict__ w, float* __restrict__ wp, int O, int I, int T) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)O * I) return;
    const long plane = (long)O * I;
    for (int t = 0; t < T; ++t) wp[t * plane + idx] = w[idx * T + t];
}

// Stem: 7x7 / stride 2 / pad 3 conv of a 3-channel NCHW image + FrozenBN + ReLU (reference resnet.py:137-139), fp32, direct.
// A workgroup computes 64 consecutive output pixels of one row for all 64 channels: thread = (pixel, group of 16 channels);
// the 64 x 147 weights sit in LDS as [c*49 + ky*7 + kx][o], so the 64 lanes of a wave (same channel group) read one
// broadcast address.  Summation order: (c, ky, kx) ascending, one accumulator per output.
__global__ __launch_bounds__(256) void stem_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ y, int B, int H, int W, int Hc,
                                                       int Wc) {
    __shared__ __attribute__((aligned(16))) float wl[147 * 64];
    for (int e = threadIdx.x; e < 147 * 64; e += 256) {
        const int o = e & 63, k = e >> 6;
        wl[e] = w[o * 147 + k];
    }
    __syncthreads();
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int wo = blockIdx.x * 64 + px, ho = blockIdx.y, b = blockIdx.z;
    if (wo >= Wc) return;
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int c = 0; c < 3; ++c) {
        const float* xc = x + ((long)b * 3 + c) * H * W;
        for (int ky = 0; ky < 7; ++ky) {
            const int hs = ho * 2 - 3 + ky;
            if ((unsigned)hs >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int ws = wo * 2 - 3 + kx;
                const float v = ((unsigned)ws < (unsigned)W) ? xc[(long)hs * W + ws] : 0.f;
                const float* wk = wl + (c * 49 + ky * 7 + kx) * 64 + cg * 16;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = fmaf(v, wk[e], acc[e]);
            }
        }
    }
    float* dst = y + (((long)b * Hc + ho) * Wc + wo) * 64 + cg * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = __fadd_rn(__fmul_rn(acc[e], scale[cg * 16 + e]), shift[cg * 16 + e]);
        dst[e] = v > 0.f ? v : 0.f;
    }
}

// 3x3 / stride 2 / pad 1 max-pool on fp32 NHWC (resnet.py:141); exact (a selection, no arithmetic)
__global__ void maxpool_f32_kernel(const f32x4* __restrict__ y, f32x4* __restrict__ pool, int B, int Hc, int Wc, int C4, int Hp, int Wp) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)B * Hp * Wp * C4) return;
    const int c4 = (int)(id % C4);
    const int wo = (int)((id / C4) % Wp), ho = (int)((id / ((long)C4 * Wp)) % Hp), b = (int)(id / ((long)C4 * Wp * Hp));
    f32x4 best = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hh = 2 * ho - 1 + ky;
        if ((unsigned)hh >= (unsigned)Hc) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ww = 2 * wo - 1 + kx;
            if ((unsigned)ww >= (unsigned)Wc) continue;
            const f32x4 v = y[(((long)b * Hc + hh) * Wc + ww) * C4 + c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) best[e] = v[e] > best[e] ? v[e] : best[e];
        }
    }
    pool[id] = best;
}

}  // namespace

extern "C" int mi_pack_weight_f32(const float* w, float* wp, int O, int I, int ksize, void* stream) {
    MI_REQUIRE(w && wp && O > 0 && I > 0 && ksize > 0, "mi_pack_weight_f32: bad argument");
    hipLaunchKernelGGL(pack_f32_kernel, dim3((unsigned)(((long)O * I + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wp, O, I, ksize * ksize);
    MI_CHECK_LAUNCH("mi_pack_weight_f32");
    return MI_OK;
}

extern "C" int mi_conv_f32(const float* a, const float* wp, float* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize,
                           int stride, int pad, int dil, const float* scale, const float* bias, const float* res, int flags, void* stream) {
    MI_REQUIRE(a && wp && out, "mi_conv_f32: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0 && N > 0, "mi_conv_f32: non-positive dimension");
    MI_REQUIRE(Ca > 0 && Ca % 16 == 0, "mi_conv_f32: Ca=%d must be a multiple of 16", Ca);
    MI_REQUIRE(ksize >= 1 && ksize <= 7 && stride >= 1 && dil >= 1 && pad >= 0, "mi_conv_f32: bad ksize/stride/dil/pad");
    MI_REQUIRE(mi_aligned16(a) && mi_aligned16(wp), "mi_conv_f32: a and wp must be 16-byte aligned");
    MI_REQUIRE(!(flags & ~(MI_EPI_SCALE_BIAS | MI_EPI_RESIDUAL | MI_EPI_RELU)), "mi_conv_f32: flags 0x%x (scale/bias, residual, relu only)", flags);
    MI_REQUIRE(!(flags & MI_EPI_SCALE_BIAS) || bias, "mi_conv_f32: MI_EPI_SCALE_BIAS needs bias (scale may be NULL: bias only)");
    MI_REQUIRE(!(flags & MI_EPI_RESIDUAL) || res, "mi_conv_f32: residual");
    MI_REQUIRE((Ho - 1) * stride - pad < Ha && (Wo - 1) * stride - pad < Wa, "mi_conv_f32: output larger than the input supports");
    const long M = (long)B * Ho * Wo;
    MI_REQUIRE(M < (1L << 31) && (long)B * Ha * Wa < (1L << 31), "mi_conv_f32: pixel count overflows int32");
    F32Params p;
    p.A = a;
    p.Wp = wp;
    p.out = out;
    p.scale = scale;
    p.bias = bias;
    p.res = res;
    p.M = (int)M;
    p.N = N;
    p.Ca = Ca;
    p.T = ksize * ksize;
    p.Ho = Ho;
    p.Wo = Wo;
    p.Ha = Ha;
    p.Wa = Wa;
    p.ksz = ksize;
    p.stride = stride;
    p.pad = pad;
    p.dil = dil;
    p.flags = flags;
    const long tiles = ((M + 127) / 128) * ((N + 127) / 128);
    hipLaunchKernelGGL(igemm_f32_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_f32");
    return MI_OK;
}

extern "C" int mi_stem_f32(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int H, int W, void* stream) {
    MI_REQUIRE(x && w && scale && shift && y && B > 0 && H > 0 && W > 0, "mi_stem_f32: bad argument");
    const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;
    MI_REQUIRE(Hc <= 65535 && B <= 65535, "mi_stem_f32: grid dimension overflow");
    hipLaunchKernelGGL(stem_f32_kernel, dim3((Wc + 63) / 64, Hc, B), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, B, H, W, Hc, Wc);
    MI_CHECK_LAUNCH("mi_stem_f32");
    return MI_OK;
}

extern "C" int mi_maxpool_f32(const float* y, float* pool, int B, int Hc, int Wc, int C, void* stream) {
    MI_REQUIRE(y && pool && B > 0 && Hc > 0 && Wc > 0 && C > 0 && C % 4 == 0, "mi_maxpool_f32: bad argument (C %% 4 == 0)");
    MI_REQUIRE(mi_aligned16(y) && mi_aligned16(pool), "mi_maxpool_f32: alignment");
    const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
    const long n = (long)B * Hp * Wp * (C / 4);
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)y, (f32x4*)pool, B, Hc, Wc,
                       C / 4, Hp, Wp);
    MI_CHECK_LAUNCH("mi_maxpool_f32");
    return MI_OK;
}
