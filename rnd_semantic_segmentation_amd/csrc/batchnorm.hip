// Trainable BatchNorm2d over NHWC bf16 activations [M][C] (M = B*H*W pixels): the kernels behind MODEL.FREEZE_BN=False
// (reference core/models/feature_extractor.py:37-39 builds the backbone on torch.nn.BatchNorm2d then; core/components/resnet.py:84-113
// applies it after every conv).  Per-channel reductions are deterministic two-level sums (up to 1024 workgroups each sum a
// contiguous range of rows in a fixed lane order, one thread per channel then adds the partials in ascending order) and return RAW
// sums: the host divides by the pixel count - after an all-reduce over ranks when the statistics are synchronised.
//   mi_bn_colsum        s[c] = sum_m y[m][c]                         (mean == NULL)
//                       s[c] = sum_m (y[m][c] - mean[c])^2           (two-pass variance: no cancellation)
//   mi_bn_colsum2       s1[c] = sum_m (y - pilot[c]), s2[c] = sum_m (y - pilot[c])^2 in ONE pass: mean = pilot + s1/N,
//                       var = s2/N - (s1/N)^2; with the pilot near the mean (the running mean) the subtraction loses 2-3 of 24 bits
//   mi_bn_finalize      (s1, s2, pilot, count) -> mean, invstd, gamma * invstd, beta - mean * gamma * invstd; running statistics as torch updates them
//   mi_bn_apply         out = relu?((y - mean) * scale + beta (+ res)), optional packed sign bits   scale = gamma * invstd
//   mi_bn_bwd_colsums   dbeta[c] = sum_m g[m][c],   dgamma[c] = sum_m g[m][c] * (y[m][c] - mean[c]) * invstd[c]   (g optionally masked by the
//                       packed ReLU sign bits of the layer's output: no separate mask pass)
//   mi_bn_bwd_apply     dy = gamma * invstd * (g - dbeta * inv_count - xhat * dgamma * inv_count)
// All arithmetic in fp32; bf16 only in memory.  Bound: HBM (each kernel is one or two streaming passes).
#include "mi_common.h"

namespace {

constexpr int BN_MAX_BLOCKS = 256;

// MODE 0: sum y   1: sum (y - mean)^2   2: sum g and sum g * xhat (g masked by the packed ReLU sign bits when given)
// 3: sum (y - pilot) and sum (y - pilot)^2 in ONE pass (`mean` carries the pilot).  A thread owns 8 consecutive channels (one 16-byte
// load per row).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const bf16x8* __restrict__ y, const bf16x8* __restrict__ g, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const uint8_t* __restrict__ bits,
                                                         float* __restrict__ partial, long M, int slots, int cp, long rows_per_block) {
    constexpr int NP = MODE >= 2 ? 2 : 1;
    __shared__ float red[256 * 8];
    const int slot = threadIdx.x % cp, rl = threadIdx.x / cp, lanes = 256 / cp;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float a0[8], a1[8], mu[8], is[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = mu[e] = 0.f, is[e] = 1.f;
    if (slot < slots) {
        if (MODE >= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) mu[e] = mean[slot * 8 + e];
        if (MODE == 2)
#pragma unroll
            for (int e = 0; e < 8; ++e) is[e] = invstd[slot * 8 + e];
        auto row = [&](const bf16x8 v, const bf16x8 gv, unsigned keep) {
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a0[e] += (float)v[e];
            } else if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)v[e] - mu[e];
                    a0[e] += d * d;
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)v[e] - mu[e];
                    a0[e] += d;
                    a1[e] += d * d;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ge = ((keep >> e) & 1u) ? (float)gv[e] : 0.f;
                    a0[e] += ge;
                    a1[e] += ge * (((float)v[e] - mu[e]) * is[e]);
                }
            }
        };
        // four rows per trip: the loads are issued together (a lone 16-byte load per trip leaves the kernel latency-bound), the
        // additions keep the row order, so the sums do not depend on the unrolling
        long r = r0 + rl;
        for (; r + 3 * lanes < r1; r += 4 * lanes) {
            bf16x8 v[4], gv[4];
            unsigned keep[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long at = (r + (long)u * lanes) * slots + slot;
                v[u] = y[at];
                if (MODE == 2) gv[u] = g[at];
                keep[u] = (MODE == 2 && bits) ? bits[at] : 0xffu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) row(v[u], gv[u], keep[u]);
        }
        for (; r < r1; r += lanes) {
            const long at = r * slots + slot;
            bf16x8 gv;
            if (MODE == 2) gv = g[at];
            row(y[at], gv, (MODE == 2 && bits) ? bits[at] : 0xffu);
        }
    }
    const int C = slots * 8;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p) __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = p ? a1[e] : a0[e];
        __syncthreads();
        if (rl == 0 && slot < slots) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = 0.f;
                for (int q = 0; q < lanes; ++q) s += red[(q * cp + slot) * 8 + e];
                partial[((long)blockIdx.x * NP + p) * C + slot * 8 + e] = s;
            }
        }
    }
}

// 8 channels per workgroup, 32 lanes per channel: lane q adds partials q, q+32, ... ascending; lane 0 adds the 32 lane sums ascending.
// (64 channels x 4 lanes with coalesced rows was tried: 3x slower - 256 dependent additions per lane; the chain length, not the
// access pattern, sets this kernel's time)
__global__ __launch_bounds__(256) void bn_final_kernel(const float* __restrict__ partial, int nblocks, int C, int nplanes, float* __restrict__ out0,
                                                       float* __restrict__ out1) {
    __shared__ float red[32][8];
    const int c = threadIdx.x & 7, q = threadIdx.x >> 3;
    const int n = blockIdx.x * 8 + c;
    for (int p = 0; p < nplanes; ++p) {
        float s = 0.f;
        if (n < C)
            for (int b = q; b < nblocks; b += 32) s += partial[((long)b * nplanes + p) * C + n];
        if (p) __syncthreads();
        red[q][c] = s;
        __syncthreads();
        if (q == 0 && n < C) {
            float t = 0.f;
            for (int k = 0; k < 32; ++k) t += red[k][c];
            (p ? out1 : out0)[n] = t;
        }
    }
}

template <bool RELU, bool RES, bool BITS>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16x8* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ scale,
                                                       const float* __restrict__ beta, const bf16x8* __restrict__ res, bf16x8* __restrict__ out,
                                                       uint8_t* __restrict__ bits, long n8, int slots) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    const int slot = (int)(idx % slots);
    const bf16x8 v = y[idx];
    bf16x8 r;
    if (RES) r = res[idx];
    float pm[8], ps[8], pb[8];                     // this thread's 8 channels of each parameter: two 16-byte loads each
    *reinterpret_cast<f32x4*>(pm) = reinterpret_cast<const f32x4*>(mean)[slot * 2];
    *reinterpret_cast<f32x4*>(pm + 4) = reinterpret_cast<const f32x4*>(mean)[slot * 2 + 1];
    *reinterpret_cast<f32x4*>(ps) = reinterpret_cast<const f32x4*>(scale)[slot * 2];
    *reinterpret_cast<f32x4*>(ps + 4) = reinterpret_cast<const f32x4*>(scale)[slot * 2 + 1];
    *reinterpret_cast<f32x4*>(pb) = reinterpret_cast<const f32x4*>(beta)[slot * 2];
    *reinterpret_cast<f32x4*>(pb + 4) = reinterpret_cast<const f32x4*>(beta)[slot * 2 + 1];
    bf16x8 o;
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = ((float)v[e] - pm[e]) * ps[e] + pb[e];
        if (RES) t += (float)r[e];
        if (RELU) t = fmaxf(t, 0.f);
        o[e] = (__bf16)t;
        m |= (unsigned)((float)o[e] > 0.f) << e;
    }
    out[idx] = o;
    if (BITS) bits[idx] = (uint8_t)m;          // byte idx = channels 8*slot .. 8*slot+7 of the row: the uint16-per-16-channels layout of the convs
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16x8* __restrict__ g, const bf16x8* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_count,
                                                           const uint8_t* __restrict__ bits, bf16x8* __restrict__ dy, long n8, int slots) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n8) return;
    const int slot = (int)(idx % slots);
    const bf16x8 gv = g[idx], v = y[idx];
    const unsigned keep = bits ? bits[idx] : 0xffu;
    float pm[8], pi[8], pg[8], pdb[8], pdg[8];
#define BN_LOAD8(dst, src)                                                                \
    *reinterpret_cast<f32x4*>(dst) = reinterpret_cast<const f32x4*>(src)[slot * 2];       \
    *reinterpret_cast<f32x4*>(dst + 4) = reinterpret_cast<const f32x4*>(src)[slot * 2 + 1];
    BN_LOAD8(pm, mean)
    BN_LOAD8(pi, invstd)
    BN_LOAD8(pg, gamma)
    BN_LOAD8(pdb, dbeta)
    BN_LOAD8(pdg, dgamma)
#undef BN_LOAD8
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xhat = ((float)v[e] - pm[e]) * pi[e];
        const float ge = ((keep >> e) & 1u) ? (float)gv[e] : 0.f;
        o[e] = (__bf16)(pg[e] * pi[e] * (ge - pdb[e] * inv_count - xhat * pdg[e] * inv_count));
    }
    dy[idx] = o;
}

// The pilot-form sums of one channel (after the all-reduce over ranks, when the layer is synchronised) -> batch mean, invstd, the folded affine
// (scale = gamma * invstd, shift = beta - mean * scale) and torch's running-statistics update (biased variance for the normalisation, unbiased
// for running_var), in double: no host arithmetic between the conv and the normalise pass.
struct BnFinalArgs {
    const float* pilot;
    double count;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    long long* nbt;
    float momentum, eps;
    float* out;          // [4][C]: mean | invstd | gamma * invstd | beta - mean * gamma * invstd
};

__device__ __forceinline__ void bn_finalize_channel(const BnFinalArgs& a, float s1, float s2, int c, int C) {
    const double d = (double)s1 / a.count;
    const double mean = (double)a.pilot[c] + d;
    double var = (double)s2 / a.count - d * d;
    var = var > 0.0 ? var : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float sc = a.gamma[c] * invstd;
    a.out[c] = (float)mean;
    a.out[C + c] = invstd;
    a.out[2 * C + c] = sc;
    a.out[3 * C + c] = a.beta[c] - (float)mean * sc;
    if (a.running_mean) {
        const double unb = a.count > 1.0 ? a.count / (a.count - 1.0) : 1.0;
        a.running_mean[c] = (float)((1.0 - (double)a.momentum) * (double)a.running_mean[c] + (double)a.momentum * mean);
        a.running_var[c] = (float)((1.0 - (double)a.momentum) * (double)a.running_var[c] + (double)a.momentum * var * unb);
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ s1, const float* __restrict__ s2, BnFinalArgs a, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && a.nbt) a.nbt[0] += 1;
    if (c < C) bn_finalize_channel(a, s1[c], s2[c], c, C);
}

// Reduction of the conv epilogue's partial rows [nparts][2][C] (thousands of rows at M = 300 000: one per 64..160 output rows).  Stage 1: workgroup
// (column group of 32 channels, row split) - thread (q, l) adds rows l, l+32, ... of its split for channels 4q..4q+3 of both planes (16-byte loads,
// a 128-byte line per row and plane), then lane order 0..31 in LDS -> tmp[split][2][C].  Stage 2: one thread per channel adds the splits in
// ascending order and, when asked, finalizes the BatchNorm statistics in the same launch.  Every order is fixed: bitwise reproducible.
constexpr int BN_RED_ROWS = 256;       // partial rows per stage-1 workgroup

__global__ __launch_bounds__(256) void bn_reduce_stage1_kernel(const float* __restrict__ partial, int nparts, int C, float* __restrict__ tmp) {
    __shared__ f32x4 red[2][32][8];
    const int q = threadIdx.x & 7, l = threadIdx.x >> 3;
    const int c = blockIdx.x * 32 + 4 * q;
    const int r0 = blockIdx.y * BN_RED_ROWS, r1 = min(nparts, r0 + BN_RED_ROWS);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (c < C) {
        for (int r = r0 + l; r < r1; r += 32) {
            a0 += *reinterpret_cast<const f32x4*>(partial + ((long)r * 2) * C + c);
            a1 += *reinterpret_cast<const f32x4*>(partial + ((long)r * 2 + 1) * C + c);
        }
    }
    red[0][l][q] = a0;
    red[1][l][q] = a1;
    __syncthreads();
    if (threadIdx.x < 16 && blockIdx.x * 32 + 4 * (threadIdx.x & 7) < C) {
        const int pl = threadIdx.x >> 3, qq = threadIdx.x & 7;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 32; ++k) t += red[pl][k][qq];
        *reinterpret_cast<f32x4*>(tmp + ((long)blockIdx.y * 2 + pl) * C + blockIdx.x * 32 + 4 * qq) = t;
    }
}

template <bool FINALIZE>
__global__ __launch_bounds__(256) void bn_reduce_stage2_kernel(const float* __restrict__ tmp, int nsplit, int C, float* __restrict__ s1,
                                                               float* __restrict__ s2, BnFinalArgs a) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (FINALIZE && c == 0 && a.nbt) a.nbt[0] += 1;
    if (c >= C) return;
    float t0 = 0.f, t1 = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        t0 += tmp[((long)s * 2) * C + c];
        t1 += tmp[((long)s * 2 + 1) * C + c];
    }
    s1[c] = t0;
    s2[c] = t1;
    if (FINALIZE) bn_finalize_channel(a, t0, t1, c, C);
}

inline int pow2_at_least(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct Plan { int slots, cp, nblocks; long rows_per_block; };

inline Plan bn_plan(long M, int C) {
    Plan p;
    p.slots = C / 8;
    p.cp = pow2_at_least(p.slots);
    const int lanes = 256 / p.cp;
    long rpb = (M + BN_MAX_BLOCKS - 1) / BN_MAX_BLOCKS;
    rpb = ((rpb + lanes - 1) / lanes) * lanes;
    if (rpb < lanes) rpb = lanes;
    p.rows_per_block = rpb;
    p.nblocks = (int)((M + rpb - 1) / rpb);
    return p;
}

}  // namespace

extern "C" size_t mi_bn_workspace(long M, int C) {
    (void)M;
    return (size_t)BN_MAX_BLOCKS * 2 * (size_t)C * sizeof(float);
}

#define BN_COMMON_CHECKS(who)                                                                                      \
    MI_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, who ": M=%ld, C=%d (C a multiple of 8, at most 2048)", M, C); \
    MI_REQUIRE(workspace && workspace_bytes >= mi_bn_workspace(M, C), who ": workspace too small");

extern "C" int mi_bn_colsum(const void* y_bf16, const float* mean, long M, int C, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(y_bf16 && out && mi_aligned16(y_bf16), "mi_bn_colsum: null or unaligned operand");
    BN_COMMON_CHECKS("mi_bn_colsum")
    const Plan p = bn_plan(M, C);
    float* partial = (float*)workspace;
    if (mean)
        hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(p.nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)y_bf16, (const bf16x8*)nullptr, mean,
                           (const float*)nullptr, (const uint8_t*)nullptr, partial, M, p.slots, p.cp, p.rows_per_block);
    else
        hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(p.nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)y_bf16, (const bf16x8*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const uint8_t*)nullptr, partial, M, p.slots, p.cp, p.rows_per_block);
    hipLaunchKernelGGL(bn_final_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partial, p.nblocks, C, 1, out, (float*)nullptr);
    MI_CHECK_LAUNCH("mi_bn_colsum");
    return MI_OK;
}

extern "C" int mi_bn_colsum2(const void* y_bf16, const float* pilot, long M, int C, float* s1, float* s2, void* workspace, size_t workspace_bytes,
                             void* stream) {
    MI_REQUIRE(y_bf16 && pilot && s1 && s2 && mi_aligned16(y_bf16), "mi_bn_colsum2: null or unaligned operand");
    BN_COMMON_CHECKS("mi_bn_colsum2")
    const Plan p = bn_plan(M, C);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(bn_partial_kernel<3>, dim3(p.nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)y_bf16, (const bf16x8*)nullptr, pilot,
                       (const float*)nullptr, (const uint8_t*)nullptr, partial, M, p.slots, p.cp, p.rows_per_block);
    hipLaunchKernelGGL(bn_final_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partial, p.nblocks, C, 2, s1, s2);
    MI_CHECK_LAUNCH("mi_bn_colsum2");
    return MI_OK;
}

static int bn_final_args_ok(const MiBnFinal* f) {
    return f->pilot && f->gamma && f->beta && f->out4 && f->count >= 1.0 && ((f->running_mean == nullptr) == (f->running_var == nullptr));
}

static BnFinalArgs bn_final_args(const MiBnFinal* f) {
    BnFinalArgs a;
    a.pilot = f->pilot, a.count = f->count, a.gamma = f->gamma, a.beta = f->beta, a.running_mean = f->running_mean, a.running_var = f->running_var;
    a.nbt = f->num_batches_tracked, a.momentum = f->momentum, a.eps = f->eps, a.out = f->out4;
    return a;
}

size_t mi_bn_reduce_tmp_floats(int nparts, int C) { return (size_t)((nparts + BN_RED_ROWS - 1) / BN_RED_ROWS) * 2 * (size_t)C; }

int mi_bn_reduce_partials(const float* partial, int nparts, int C, float* tmp, float* s1, float* s2, const MiBnFinal* fin, void* stream) {
    MI_REQUIRE(partial && tmp && s1 && s2 && nparts > 0 && C > 0 && C % 4 == 0, "mi_bn_reduce_partials: null operand or C=%d not a multiple of 4", C);
    MI_REQUIRE(!fin || bn_final_args_ok(fin), "mi_bn_reduce_partials: incomplete finalize arguments");
    const int nsplit = (nparts + BN_RED_ROWS - 1) / BN_RED_ROWS;
    hipLaunchKernelGGL(bn_reduce_stage1_kernel, dim3((C + 31) / 32, nsplit), dim3(256), 0, (hipStream_t)stream, partial, nparts, C, tmp);
    if (fin)
        hipLaunchKernelGGL(bn_reduce_stage2_kernel<true>, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, nsplit, C, s1, s2,
                           bn_final_args(fin));
    else
        hipLaunchKernelGGL(bn_reduce_stage2_kernel<false>, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, nsplit, C, s1, s2,
                           BnFinalArgs{});
    MI_CHECK_LAUNCH("mi_bn_reduce_partials");
    return MI_OK;
}

extern "C" int mi_bn_finalize(const float* s1, const float* s2, const float* pilot, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* out4,
                              int C, void* stream) {
    const MiBnFinal f{pilot, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, out4};
    MI_REQUIRE(s1 && s2 && C > 0 && bn_final_args_ok(&f), "mi_bn_finalize: null operand, C=%d, count=%g", C, count);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, s1, s2, bn_final_args(&f), C);
    MI_CHECK_LAUNCH("mi_bn_finalize");
    return MI_OK;
}

extern "C" int mi_bn_bwd_colsums(const void* g_bf16, const void* y_bf16, const float* mean, const float* invstd, const void* relu_bits, long M, int C,
                                 float* dbeta, float* dgamma, void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(g_bf16 && y_bf16 && mean && invstd && dbeta && dgamma && mi_aligned16(g_bf16) && mi_aligned16(y_bf16), "mi_bn_bwd_colsums: null or unaligned operand");
    BN_COMMON_CHECKS("mi_bn_bwd_colsums")
    const Plan p = bn_plan(M, C);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(bn_partial_kernel<2>, dim3(p.nblocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)y_bf16, (const bf16x8*)g_bf16, mean, invstd,
                       (const uint8_t*)relu_bits, partial, M, p.slots, p.cp, p.rows_per_block);
    hipLaunchKernelGGL(bn_final_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partial, p.nblocks, C, 2, dbeta, dgamma);
    MI_CHECK_LAUNCH("mi_bn_bwd_colsums");
    return MI_OK;
}

extern "C" int mi_bn_apply(const void* y_bf16, const float* mean, const float* scale, const float* beta, const void* res_bf16, void* out_bf16,
                           void* mask_out, int relu, long M, int C, void* stream) {
    MI_REQUIRE(y_bf16 && mean && scale && beta && out_bf16 && mi_aligned16(y_bf16) && mi_aligned16(out_bf16), "mi_bn_apply: null or unaligned operand");
    MI_REQUIRE(M > 0 && C > 0 && C % 8 == 0, "mi_bn_apply: M=%ld, C=%d (C a multiple of 8)", M, C);
    MI_REQUIRE(!res_bf16 || mi_aligned16(res_bf16), "mi_bn_apply: residual alignment");
    MI_REQUIRE(mi_aligned16(mean) && mi_aligned16(scale) && mi_aligned16(beta), "mi_bn_apply: per-channel vectors must be 16-byte aligned");
    MI_REQUIRE(!mask_out || C % 16 == 0, "mi_bn_apply: sign bits need C %% 16 == 0");
    const long n8 = M * (C / 8);
    const dim3 grid((unsigned)((n8 + 255) / 256));
#define BN_GO(R, S, B)                                                                                                                         \
    hipLaunchKernelGGL((bn_apply_kernel<R, S, B>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16x8*)y_bf16, mean, scale, beta, (const bf16x8*)res_bf16, \
                       (bf16x8*)out_bf16, (uint8_t*)mask_out, n8, C / 8)
    const int sel = (relu ? 4 : 0) | (res_bf16 ? 2 : 0) | (mask_out ? 1 : 0);
    switch (sel) {
        case 0: BN_GO(false, false, false); break;
        case 1: BN_GO(false, false, true); break;
        case 2: BN_GO(false, true, false); break;
        case 3: BN_GO(false, true, true); break;
        case 4: BN_GO(true, false, false); break;
        case 5: BN_GO(true, false, true); break;
        case 6: BN_GO(true, true, false); break;
        default: BN_GO(true, true, true); break;
    }
#undef BN_GO
    MI_CHECK_LAUNCH("mi_bn_apply");
    return MI_OK;
}

extern "C" int mi_bn_bwd_apply(const void* g_bf16, const void* y_bf16, const float* mean, const float* invstd, const float* gamma, const float* dbeta,
                               const float* dgamma, float inv_count, const void* relu_bits, void* dy_bf16, long M, int C, void* stream) {
    MI_REQUIRE(g_bf16 && y_bf16 && mean && invstd && gamma && dbeta && dgamma && dy_bf16, "mi_bn_bwd_apply: null operand");
    MI_REQUIRE(mi_aligned16(g_bf16) && mi_aligned16(y_bf16) && mi_aligned16(dy_bf16), "mi_bn_bwd_apply: alignment");
    MI_REQUIRE(mi_aligned16(mean) && mi_aligned16(invstd) && mi_aligned16(gamma) && mi_aligned16(dbeta) && mi_aligned16(dgamma),
               "mi_bn_bwd_apply: per-channel vectors must be 16-byte aligned");
    MI_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && (!relu_bits || C % 16 == 0), "mi_bn_bwd_apply: M=%ld, C=%d (C a multiple of 8; 16 with sign bits)", M, C);
    const long n8 = M * (C / 8);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)g_bf16,
                       (const bf16x8*)y_bf16, mean, invstd, gamma, dbeta, dgamma, inv_count, (const uint8_t*)relu_bits, (bf16x8*)dy_bf16, n8, C / 8);
    MI_CHECK_LAUNCH("mi_bn_bwd_apply");
    return MI_OK;
}
