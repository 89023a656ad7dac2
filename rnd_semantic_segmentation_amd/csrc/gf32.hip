// The general, view-based kernel family in the REFERENCE'S precision (fp32 operands, fp32 accumulation on v_mfma_f32_16x16x4_f32): the
// evaluation forward of the PraNet and GALD networks (core/testers/pranet_tester.py:25-53, core/testers/gald_tester.py:47-90), so that the
// masks / IoU the testers report are the reference's own (BASELINE: logits within 1e-3 relative, argmax masks identical) and not the bf16
// training regime's.  Same operand convention as gconv.hip / gnet.hip: every tensor is a channel-slice VIEW of an NHWC tensor (pointer to
// channel 0 + elements per pixel row), now of floats; weights are read straight from the fp32 OIHW masters (no pack).
//
// Forward only, written for exactness and robustness, not for speed (an evaluation pass is 26 GFLOP per 352 x 352 image for PraNet): scalar
// global loads with zero fill (any alignment, any channel count), one LDS stage per 16-channel chunk.  BatchNorm2d in eval() is the per-channel
// affine ATen applies (alpha = weight * rsqrt(running_var + eps), beta = bias - running_mean * alpha: mi_gbn_fold) in the conv's epilogue,
// followed by the residual add and the activation, in the reference's order (Res2Net_v1b.py:86-92: bn3 -> += residual -> relu).
#include "mi_common.h"

namespace {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2 };

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// ------------------------------------------------------------------------------------------------ convolution
struct CF32P {
    const float* A;
    const float* W;        // [N][Ca][kh][kw]
    const float* bias;     // [N] or null (added first: nn.Conv2d's own bias)
    const float* scale;    // [N] or null: v = v * scale + shift (BatchNorm2d in eval())
    const float* shift;
    const float* add;      // residual view or null
    float* out;
    long lda, ldadd, ldo;
    int M, N, Ca, T;
    int Ha, Wa, Ho, Wo;
    int kw, sh, sw, ph, pw, dh, dw, act;
};

constexpr int FBM = 64, FBN = 64, FKC = 16, FLD = FKC + 1;      // 64 pixels x 64 channels per workgroup, 16-channel K chunks, padded LDS rows

__global__ __launch_bounds__(256) void gconv_f32_kernel(CF32P p) {
    __shared__ float As[FBM * FLD];
    __shared__ float Bs[FBN * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * FBM, n0 = blockIdx.y * FBN;
    // loader: thread -> (row, 4 consecutive channels of the chunk) for both tiles
    const int lrow = tid >> 2, lc = (tid & 3) * 4;
    const int am = m0 + lrow;
    const bool am_ok = am < p.M;
    int ab = 0, aoh = 0, aow = 0;
    if (am_ok) {
        const int hw = p.Ho * p.Wo;
        ab = am / hw;
        const int rem = am - ab * hw;
        aoh = rem / p.Wo;
        aow = rem - aoh * p.Wo;
    }
    const int bn = n0 + lrow;
    const bool bn_ok = bn < p.N;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const int nchunks = (p.Ca + FKC - 1) / FKC;
    for (int t = 0; t < p.T; ++t) {
        const int ky = t / p.kw, kx = t - ky * p.kw;
        const int ih = aoh * p.sh + ky * p.dh - p.ph, iw = aow * p.sw + kx * p.dw - p.pw;
        const bool ok = am_ok && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
        const float* arow = p.A + (((long)ab * p.Ha + (ok ? ih : 0)) * p.Wa + (ok ? iw : 0)) * p.lda;
        for (int kc = 0; kc < nchunks; ++kc) {
            const int c0 = kc * FKC + lc;
            float av[4], bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                av[j] = (ok && c < p.Ca) ? arow[c] : 0.f;
                bv[j] = (bn_ok && c < p.Ca) ? p.W[((long)bn * p.Ca + c) * p.T + t] : 0.f;
            }
            __syncthreads();                      // the previous chunk's fragments have been read
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                As[lrow * FLD + lc + j] = av[j];
                Bs[lrow * FLD + lc + j] = bv[j];
            }
            __syncthreads();
#pragma unroll
            for (int k4 = 0; k4 < FKC / 4; ++k4) {
                const float a = As[(wave * 16 + frow) * FLD + k4 * 4 + fk];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float b = Bs[(j * 16 + frow) * FLD + k4 * 4 + fk];
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // D[row][col]: the lane holds rows 4 * (lane / 16) + r, column lane % 16 of each 16 x 16 block
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + frow;
        if (n >= p.N) continue;
        const float bi = p.bias ? p.bias[n] : 0.f;
        const float sc = p.scale ? p.scale[n] : 1.f, sf = p.scale ? p.shift[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + fk * 4 + r;
            if (m >= p.M) continue;
            float v = acc[j][r] + bi;
            if (p.scale) v = v * sc + sf;
            if (p.add) v += p.add[(long)m * p.ldadd + n];
            p.out[(long)m * p.ldo + n] = act_f(v, p.act);
        }
    }
}

// ------------------------------------------------------------------------------------------------ pools
struct PF32P {
    int B, H, W, C, Ho, Wo, k, s, p, mode;      // mode 0: average, divisor k * k; 1: average over the in-image part of the window; 2: max
    long ldx, ldo;
};
__global__ __launch_bounds__(256) void gpool_f32_kernel(const float* x, float* out, PF32P q) {
    const long n = (long)q.B * q.Ho * q.Wo * q.C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % q.C);
        const long m = e / q.C;
        const int ow = (int)(m % q.Wo);
        const long t = m / q.Wo;
        const int oh = (int)(t % q.Ho), b = (int)(t / q.Ho);
        float s = q.mode == 2 ? -INFINITY : 0.f;
        int cnt = 0;
        for (int ky = 0; ky < q.k; ++ky) {
            const int ih = oh * q.s - q.p + ky;
            if ((unsigned)ih >= (unsigned)q.H) continue;
            for (int kx = 0; kx < q.k; ++kx) {
                const int iw = ow * q.s - q.p + kx;
                if ((unsigned)iw >= (unsigned)q.W) continue;
                const float v = x[(((long)b * q.H + ih) * q.W + iw) * q.ldx + c];
                s = q.mode == 2 ? fmaxf(s, v) : s + v;
                ++cnt;
            }
        }
        if (q.mode == 0) s /= (float)(q.k * q.k);
        else if (q.mode == 1) s /= (float)(cnt > 0 ? cnt : 1);
        out[m * q.ldo + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3 (+ bias) (+ affine) (+ activation)
struct DF32P {
    int B, H, W, C, Ho, Wo, s, p, act;
    long ldx, ldo;
};
__global__ __launch_bounds__(256) void gdwconv_f32_kernel(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* out,
                                                          DF32P q) {
    const long n = (long)q.B * q.Ho * q.Wo * q.C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % q.C);
        const long m = e / q.C;
        const int ow = (int)(m % q.Wo);
        const long t = m / q.Wo;
        const int oh = (int)(t % q.Ho), b = (int)(t / q.Ho);
        float s = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int ih = oh * q.s - q.p + ky;
            if ((unsigned)ih >= (unsigned)q.H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int iw = ow * q.s - q.p + kx;
                if ((unsigned)iw >= (unsigned)q.W) continue;
                s = fmaf(x[(((long)b * q.H + ih) * q.W + iw) * q.ldx + c], w[c * 9 + ky * 3 + kx], s);
            }
        }
        if (bias) s += bias[c];
        if (scale) s = s * scale[c] + shift[c];
        out[m * q.ldo + c] = act_f(s, q.act);
    }
}

// ------------------------------------------------------------------------------------------------ criss-cross attention core (ccnet.py:56-127)
// One workgroup per pixel (b, h, w): affinities q . k with its column (own position masked with -inf) and its row, ONE softmax over the H + W
// candidates, values aggregated with those weights.
struct CCP {
    int B, H, W, Cq, C;
    long ldq, ldk, ldv, ldo;
};
__global__ __launch_bounds__(256) void gcca_f32_kernel(const float* q, const float* k, const float* v, float* out, CCP p) {
    extern __shared__ float sm[];      // [H + W] weights, [Cq] the query
    float* att = sm;
    float* qv = sm + p.H + p.W;
    const int pix = blockIdx.x;
    const int w = pix % p.W, h = (pix / p.W) % p.H, b = pix / (p.W * p.H);
    const int tid = threadIdx.x, L = p.H + p.W;
    for (int c = tid; c < p.Cq; c += 256) qv[c] = q[(long)pix * p.ldq + c];
    __syncthreads();
    for (int i = tid; i < L; i += 256) {
        const long kp = i < p.H ? ((long)b * p.H + i) * p.W + w : ((long)b * p.H + h) * p.W + (i - p.H);
        float e = 0.f;
        for (int c = 0; c < p.Cq; ++c) e = fmaf(qv[c], k[kp * p.ldk + c], e);
        att[i] = (i < p.H && i == h) ? -INFINITY : e;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = 0; i < L; ++i) mx = fmaxf(mx, att[i]);
    float den = 0.f;
    for (int i = 0; i < L; ++i) den += __expf(att[i] - mx);      // (every thread computes the same two scalars: L <= a few dozen)
    __syncthreads();
    for (int i = tid; i < L; i += 256) att[i] = __expf(att[i] - mx) / den;
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        float s = 0.f;
        for (int g = 0; g < p.H; ++g) s = fmaf(att[g], v[(((long)b * p.H + g) * p.W + w) * p.ldv + c], s);
        for (int u = 0; u < p.W; ++u) s = fmaf(att[p.H + u], v[(((long)b * p.H + h) * p.W + u) * p.ldv + c], s);
        out[(long)pix * p.ldo + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------ pointwise
enum { PW_AFFINE = 0 /* act(a * scale[c] + shift[c] (+ b)) */, PW_REVERSE = 1 /* (1 - sigmoid(gate[m])) * a, gate = b with one value per pixel */,
       PW_GATE = 2 /* a + a * sigmoid(b) */, PW_MULRELU = 3 /* relu(a * b) */ };
__global__ __launch_bounds__(256) void gpoint_f32_kernel(int op, const float* a, long lda, const float* b, long ldb, const float* scale, const float* shift, int act,
                                                         float* out, long ldo, long M, int C) {
    const long n = M * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / C;
        const int c = (int)(e - m * C);
        const float av = a[m * lda + c];
        float r;
        if (op == PW_AFFINE) {
            r = av * scale[c] + shift[c];
            if (b) r += b[m * ldb + c];
            r = act_f(r, act);
        } else if (op == PW_REVERSE) {
            r = (1.f - 1.f / (1.f + expf(-b[m]))) * av;
        } else if (op == PW_GATE) {
            r = av + av * (1.f / (1.f + expf(-b[m * ldb + c])));
        } else {
            r = fmaxf(av * b[m * ldb + c], 0.f);
        }
        out[m * ldo + c] = r;
    }
}

inline int grid_f32(long n) {
    const long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g));
}

}  // namespace

extern "C" {

int mi_gconv_f32(const float* a, long lda, const float* w_oihw, const float* bias, const float* scale, const float* shift, const float* add, long ldadd, int act,
                 float* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                 void* stream) {
    MI_REQUIRE(a && w_oihw && out, "mi_gconv_f32: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0 && Ca > 0 && N > 0, "mi_gconv_f32: empty shape");
    MI_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0, "mi_gconv_f32: bad conv geometry");
    MI_REQUIRE((Ha + 2 * ph - dh * (kh - 1) - 1) / sh + 1 == Ho && (Wa + 2 * pw - dw * (kw - 1) - 1) / sw + 1 == Wo,
               "mi_gconv_f32: output %dx%d does not follow from input %dx%d", Ho, Wo, Ha, Wa);
    MI_REQUIRE(lda >= Ca && ldo >= N && (!add || ldadd >= N), "mi_gconv_f32: a view's row stride is smaller than its channel count");
    MI_REQUIRE((scale == nullptr) == (shift == nullptr), "mi_gconv_f32: scale and shift come together");
    MI_REQUIRE(act >= 0 && act <= 2, "mi_gconv_f32: activation %d", act);
    MI_REQUIRE((long)B * Ho * Wo < (1L << 31) && (long)B * Ha * Wa < (1L << 31), "mi_gconv_f32: more than 2^31 pixels");
    CF32P p;
    p.A = a; p.W = w_oihw; p.bias = bias; p.scale = scale; p.shift = shift; p.add = add; p.out = out;
    p.lda = lda; p.ldadd = ldadd; p.ldo = ldo;
    p.M = B * Ho * Wo; p.N = N; p.Ca = Ca; p.T = kh * kw;
    p.Ha = Ha; p.Wa = Wa; p.Ho = Ho; p.Wo = Wo;
    p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw; p.act = act;
    const dim3 grid((p.M + FBM - 1) / FBM, (N + FBN - 1) / FBN);
    hipLaunchKernelGGL(gconv_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("gconv_f32_kernel");
    return MI_OK;
}

int mi_gpool_f32(const float* x, long ldx, float* out, long ldo, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int mode, void* stream) {
    MI_REQUIRE(x && out, "mi_gpool_f32: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && k > 0 && stride > 0 && pad >= 0 && ldx >= C && ldo >= C, "mi_gpool_f32: bad shape");
    MI_REQUIRE(mode >= 0 && mode <= 2, "mi_gpool_f32: mode %d", mode);
    MI_REQUIRE((Ho - 1) * stride - pad < H && (Wo - 1) * stride - pad < W, "mi_gpool_f32: the last window starts outside the input");
    PF32P q{B, H, W, C, Ho, Wo, k, stride, pad, mode, ldx, ldo};
    hipLaunchKernelGGL(gpool_f32_kernel, dim3(grid_f32((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, out, q);
    MI_CHECK_LAUNCH("gpool_f32_kernel");
    return MI_OK;
}

int mi_gdwconv_f32(const float* x, long ldx, const float* w, const float* bias, const float* scale, const float* shift, int act, float* out, long ldo, int B, int H,
                   int W, int C, int Ho, int Wo, int stride, int pad, void* stream) {
    MI_REQUIRE(x && w && out, "mi_gdwconv_f32: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && stride > 0 && pad >= 0 && ldx >= C && ldo >= C, "mi_gdwconv_f32: bad shape");
    MI_REQUIRE((H + 2 * pad - 3) / stride + 1 == Ho && (W + 2 * pad - 3) / stride + 1 == Wo && Ho > 0 && Wo > 0, "mi_gdwconv_f32: output %dx%d does not follow from input %dx%d", Ho, Wo, H, W);
    MI_REQUIRE((scale == nullptr) == (shift == nullptr) && act >= 0 && act <= 2, "mi_gdwconv_f32: bad epilogue");
    DF32P q{B, H, W, C, Ho, Wo, stride, pad, act, ldx, ldo};
    hipLaunchKernelGGL(gdwconv_f32_kernel, dim3(grid_f32((long)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, w, bias, scale, shift, out, q);
    MI_CHECK_LAUNCH("gdwconv_f32_kernel");
    return MI_OK;
}

int mi_gcca_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* out, long ldo, int B, int H, int W, int Cq, int C, void* stream) {
    MI_REQUIRE(q && k && v && out, "mi_gcca_f32: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cq > 0 && C > 0 && ldq >= Cq && ldk >= Cq && ldv >= C && ldo >= C, "mi_gcca_f32: bad shape");
    MI_REQUIRE((long)(H + W + Cq) * 4 <= 60000, "mi_gcca_f32: H + W + Cq = %d does not fit the LDS plan", H + W + Cq);
    CCP p{B, H, W, Cq, C, ldq, ldk, ldv, ldo};
    hipLaunchKernelGGL(gcca_f32_kernel, dim3(B * H * W), dim3(256), (size_t)(H + W + Cq) * 4, (hipStream_t)stream, q, k, v, out, p);
    MI_CHECK_LAUNCH("gcca_f32_kernel");
    return MI_OK;
}

int mi_gpoint_f32(int op, const float* a, long lda, const float* b, long ldb, const float* scale, const float* shift, int act, float* out, long ldo, long M, int C,
                  void* stream) {
    MI_REQUIRE(a && out && M > 0 && C > 0 && lda >= C && ldo >= C, "mi_gpoint_f32: bad operand");
    MI_REQUIRE(op >= 0 && op <= 3 && act >= 0 && act <= 2, "mi_gpoint_f32: op %d / act %d", op, act);
    MI_REQUIRE(op == PW_AFFINE ? (scale && shift) : b != nullptr, "mi_gpoint_f32: missing operand for op %d", op);
    MI_REQUIRE(!b || op == PW_REVERSE || ldb >= C, "mi_gpoint_f32: second operand's row stride");
    hipLaunchKernelGGL(gpoint_f32_kernel, dim3(grid_f32(M * C)), dim3(256), 0, (hipStream_t)stream, op, a, lda, b, ldb, scale, shift, act, out, ldo, M, C);
    MI_CHECK_LAUNCH("gpoint_f32_kernel");
    return MI_OK;
}

}  // extern "C"
