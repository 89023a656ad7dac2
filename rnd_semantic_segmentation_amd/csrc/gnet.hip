// The non-GEMM kernels of the PraNet path on channel-slice VIEWS of NHWC tensors (pointer + elements per pixel
// row, any channel count): trainable BatchNorm2d on batch statistics split into finalize / apply / backward sums / backward apply
// (the first-level statistics come out of the conv epilogue, gconv.hip), average pools, bilinear resizing with either
// align_corners convention, the elementwise products / sums of the partial decoder and the reverse-attention gate.
//   reference: core/models/classifiers/pranet/PraNet_Res2Net.py:7-179, Res2Net_v1b.py:15-170.
// Every reduction has a fixed order (no float atomics): two runs give the same bits.
#include "mi_common.h"
#include <initializer_list>
#include <utility>

namespace {

__device__ __attribute__((aligned(256))) uint32_t g_nzero[64];          // what an absent optional operand (mask, y) is read from: loads stay unconditional

__device__ __forceinline__ float ldf(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ void stf(__bf16* p, float v) { *p = (__bf16)v; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }


// VEC consecutive channels of one pixel as floats (VEC = 8: one 16-byte access, 2: one 4-byte access, 1: scalar); fp32 sources are scalar
template <int VEC>
__device__ __forceinline__ void ldv(const __bf16* p, float (&v)[VEC]) {
    if constexpr (VEC == 8) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    } else if constexpr (VEC == 2) {
        const bf16x2 x = *reinterpret_cast<const bf16x2*>(p);
        v[0] = (float)x[0];
        v[1] = (float)x[1];
    } else {
        v[0] = (float)*p;
    }
}
template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = p[j];
}
template <int VEC>
__device__ __forceinline__ void stv(__bf16* p, const float (&v)[VEC]) {
    if constexpr (VEC == 8) {
        bf16x8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (__bf16)v[j];
        *reinterpret_cast<bf16x8*>(p) = x;
    } else if constexpr (VEC == 2) {
        bf16x2 x;
        x[0] = (__bf16)v[0];
        x[1] = (__bf16)v[1];
        *reinterpret_cast<bf16x2*>(p) = x;
    } else {
        *p = (__bf16)v[0];
    }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) p[j] = v[j];
}
// VEC per-channel fp32 parameters starting at channel c0 (c0 % VEC == 0; VEC == 8: two 16-byte loads - the parameter vectors of this path are
// 32-byte aligned rows of 256-byte aligned buffers)
template <int VEC>
__device__ __forceinline__ void ldp(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = a[j];
            v[4 + j] = b[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = p[j];
    }
}
// widest access every one of the given bf16 views allows: 8 channels (16-byte aligned, ld % 8 == 0), 2, or 1
inline int common_vec(int C, std::initializer_list<std::pair<const void*, long>> views) {
    int v = 8;
    if (C % 8) v = C % 2 ? 1 : 2;
    for (const auto& pv : views) {
        if (!pv.first) continue;
        const uintptr_t a = reinterpret_cast<uintptr_t>(pv.first);
        if (v == 8 && ((a & 15) || pv.second % 8)) v = 2;
        if (v == 2 && ((a & 3) || pv.second % 2)) v = 1;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------ BatchNorm: finalize
// partials[tile][2][C] (sum, sum of squares per 128-pixel tile, written by gconv_kernel) -> batch mean / biased variance in double,
// invstd, the folded affine (scale = gamma * invstd, shift = beta - mean * scale) and torch's running-statistics update
// (momentum m: running = (1 - m) * running + m * batch, unbiased variance for running_var; nn.BatchNorm2d defaults).
// Two channels x 128 tile lanes per workgroup: a lane adds the tiles t = lane, lane + 128, ... (its loads independent), the 128 lane sums are added in
// lane order by one thread per channel (double).  (Round 3 used 8 channels x 32 lanes: 30 dependent round trips per thread on the 968-tile maps, 6 - 12 us per
// launch for a few KB of data - 153 launches per PraNet step.)  The in-launch finalize of gconv.hip replays this order (tiles <= 64: plain ascending sum).
constexpr int FIN_LANES = 128;
__global__ __launch_bounds__(256) void gbn_finalize_kernel(const float* partials, int tiles, int C, double count, const float* gamma, const float* beta,
                                                           float* running_mean, float* running_var, float momentum, float eps, float* mean_out,
                                                           float* invstd_out, float* scale_out, float* shift_out) {
    __shared__ double red[2][2][FIN_LANES];
    const int cx = threadIdx.x / FIN_LANES, ry = threadIdx.x % FIN_LANES;
    const int c = blockIdx.x * 2 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int t = ry; t < tiles; t += FIN_LANES) {
            s1 += (double)partials[(long)t * 2 * C + c];
            s2 += (double)partials[(long)t * 2 * C + C + c];
        }
    red[0][cx][ry] = s1;
    red[1][cx][ry] = s2;
    __syncthreads();
    if (ry == 0 && c < C) {
        s1 = s2 = 0.0;
        const int live = tiles < FIN_LANES ? tiles : FIN_LANES;
        for (int k = 0; k < live; ++k) {
            s1 += red[0][cx][k];
            s2 += red[1][cx][k];
        }
        const MiBnFin f = mi_bn_finalize_channel(s1, s2, count, eps, gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f);
        mean_out[c] = f.mean;
        invstd_out[c] = f.invstd;
        scale_out[c] = f.scale;
        shift_out[c] = f.shift;
        if (running_mean) {
            running_mean[c] = mi_bn_running(running_mean[c], momentum, f.mean);
            running_var[c] = mi_bn_running(running_var[c], momentum, mi_bn_unbiased(f.var, count));
        }
    }
}

// eval(): scale = gamma * rsqrt(running_var + eps), shift = beta - running_mean * scale
__global__ void gbn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] * (1.f / sqrtf(rv[c] + eps));
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm: apply
// out = relu?(y * scale[c] + shift[c] (+ add)), one thread per VEC channels of a pixel
template <int VEC, typename TO>
__global__ __launch_bounds__(256) void gbn_apply_kernel(const __bf16* y, long ldy, const float* scale, const float* shift, const __bf16* add, long ldadd,
                                                        TO* out, long ldo, long M, int C, int relu) {
    const int cv = C / VEC;
    const long n = M * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / cv;
        const int c0 = (int)(e - m * cv) * VEC;
        float v[VEC], a[VEC], sc[VEC], sh[VEC];
        ldv<VEC>(y + m * ldy + c0, v);
        ldp<VEC>(scale + c0, sc);
        ldp<VEC>(shift + c0, sh);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = v[j] * sc[j] + sh[j];
        ldv<VEC>(add ? add + m * ldadd + c0 : reinterpret_cast<const __bf16*>(g_nzero), a);       // (unconditional: an absent residual reads zeros)
        if (add) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] += a[j];
        }
        if (relu) {                                   // 1: ReLU, 2: ReLU6 (hardnet_68.py:78)
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = relu == 2 ? fminf(fmaxf(v[j], 0.f), 6.f) : fmaxf(v[j], 0.f);
        }
        stv<VEC>(out + m * ldo + c0, v);
    }
}

// The same pass with up to four EXTRA destinations (round 5): channel range [c0, c1) of the result, as it is stored (rounded to bf16), optionally plus a second
// operand, also goes to another view - what a stand-alone copy (HarDBlock gathers: hardnet_68.py:137-160; the pass-through group of a Res2Net bottleneck:
// Res2Net_v1b.py:79-84) or a stand-alone add (sp = sp + spx[i], Res2Net_v1b.py:72-74) would produce from the stored tensor, bit for bit, without its launch
// and without reading the tensor back.
constexpr int APPLY_MAX_EXTRA = 4;
struct GApplyExtras {
    int n;
    int c0[APPLY_MAX_EXTRA], c1[APPLY_MAX_EXTRA];
    __bf16* dst[APPLY_MAX_EXTRA];
    long ldd[APPLY_MAX_EXTRA];
    const __bf16* add[APPLY_MAX_EXTRA];
    long lda[APPLY_MAX_EXTRA];
};
template <int VEC>
__global__ __launch_bounds__(256) void gbn_apply_multi_kernel(const __bf16* y, long ldy, const float* scale, const float* shift, const __bf16* add, long ldadd,
                                                              __bf16* out, long ldo, long M, int C, int relu, GApplyExtras ex) {
    const int cv = C / VEC;
    const long n = M * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / cv;
        const int c0 = (int)(e - m * cv) * VEC;
        float v[VEC], a[VEC], sc[VEC], sh[VEC];
        ldv<VEC>(y + m * ldy + c0, v);
        ldp<VEC>(scale + c0, sc);
        ldp<VEC>(shift + c0, sh);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = v[j] * sc[j] + sh[j];
        ldv<VEC>(add ? add + m * ldadd + c0 : reinterpret_cast<const __bf16*>(g_nzero), a);
        if (add) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] += a[j];
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = relu == 2 ? fminf(fmaxf(v[j], 0.f), 6.f) : fmaxf(v[j], 0.f);
        }
        if (out) stv<VEC>(out + m * ldo + c0, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = (float)(__bf16)v[j];            // the value a later pass would read back
#pragma unroll
        for (int k = 0; k < APPLY_MAX_EXTRA; ++k) {
            if (k < ex.n && c0 >= ex.c0[k] && c0 < ex.c1[k]) {
                const int cc = c0 - ex.c0[k];
                float r[VEC];
                if (ex.add[k]) {
                    float b[VEC];
                    ldv<VEC>(ex.add[k] + m * ex.lda[k] + cc, b);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) r[j] = v[j] + b[j];
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) r[j] = v[j];
                }
                stv<VEC>(ex.dst[k] + m * ex.ldd[k] + cc, r);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ column sums (BatchNorm backward, bias gradients)
// partial[blk][0][c] = sum_m g'[m][c], partial[blk][1][c] = sum_m g'[m][c] * (y[m][c] - mean[c]) * invstd[c] over the block's rows;
// g' = g where mask[m][c] > 0 (mask = the layer's ReLU output) or g itself when mask is NULL; y NULL: the first sum only.
// Thread layout: tx = chunk of VEC channels, ty = row lane (TX x TY = 256, TX = chunks per row rounded up to a power of two); a block
// sums `rows_per_block` rows in registers, the row lanes are combined through LDS in a fixed order.
struct ColsumPlan {
    int vec, tx, ty, rows_per_block, blocks;
};
inline ColsumPlan colsum_plan(long M, int C, int vec) {
    ColsumPlan q;
    q.vec = vec;
    const int cv = (C + vec - 1) / vec;
    int tx = 1;
    while (tx < cv && tx < 256) tx <<= 1;
    q.tx = tx;
    q.ty = 256 / tx;
    long rpb = (M + 383) / 384;                       // ~384 blocks at most (1.5 per CU)
    const long lo = (long)q.ty * 4;
    if (rpb < lo) rpb = lo;
    rpb = (rpb + q.ty - 1) / q.ty * q.ty;
    q.rows_per_block = (int)rpb;
    q.blocks = (int)((M + rpb - 1) / rpb);
    return q;
}
template <int VEC, typename TG, typename TM>
__global__ __launch_bounds__(256) void gcolsum_partial_kernel(const TG* g, long ldg, const __bf16* y, long ldy, const TM* mask, long ldm, const float* mean,
                                                              const float* invstd, long M, int C, float* partial, int tx_n, int rows_per_block, float hi) {
    __shared__ float red[2][256][VEC];
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n, ty_n = 256 / tx_n;
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
    const int c0 = (blockIdx.y * tx_n + tx) * VEC;      // (blockIdx.y > 0 only when a row has more than 256 chunks)
    if (c0 < C) {
        float mu[VEC], is[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            mu[j] = y ? mean[c0 + j] : 0.f;
            is[j] = y ? invstd[c0 + j] : 0.f;
        }
        const long r0 = (long)blockIdx.x * rows_per_block;
        const long rend = min(M, r0 + rows_per_block);
        // four rows per trip: their loads are issued together (one dependent 16-byte load per trip left these launches latency-bound: 20 trips of
        // ~0.7 us on the 100-channel maps of PraNet), the additions keep the row order, so the sums do not depend on the unrolling
        long m = r0 + ty;
        for (; m + 3L * ty_n < rend; m += 4L * ty_n) {
            float gv[4][VEC], yv[4][VEC], mv[4][VEC];
#pragma unroll
            // (the twelve loads of a trip are issued back to back: an absent mask / y is read from a zero page instead of being branched around - with
            //  `if (mask) load` the compiler waited for every optional load inside its own block, eight dependent round trips per trip)
            for (int u = 0; u < 4; ++u) {
                const long mm = m + (long)u * ty_n;
                ldv<VEC>(g + mm * ldg + c0, gv[u]);
                ldv<VEC>(mask ? mask + mm * ldm + c0 : reinterpret_cast<const TM*>(g_nzero), mv[u]);
                ldv<VEC>(y ? y + mm * ldy + c0 : reinterpret_cast<const __bf16*>(g_nzero), yv[u]);
            }
            const bool has_mask = mask != nullptr;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) gv[u][j] = (has_mask && !(mv[u][j] > 0.f && mv[u][j] < hi)) ? 0.f : gv[u][j];
#pragma unroll
                for (int j = 0; j < VEC; ++j) s1[j] += gv[u][j];
                if (y) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) s2[j] += gv[u][j] * ((yv[u][j] - mu[j]) * is[j]);
                }
            }
        }
        for (; m < rend; m += ty_n) {
            float gv[VEC], yv[VEC], mv[VEC];
            ldv<VEC>(g + m * ldg + c0, gv);
            ldv<VEC>(mask ? mask + m * ldm + c0 : reinterpret_cast<const TM*>(g_nzero), mv);
            ldv<VEC>(y ? y + m * ldy + c0 : reinterpret_cast<const __bf16*>(g_nzero), yv);
            const bool has_mask = mask != nullptr;
#pragma unroll
            for (int j = 0; j < VEC; ++j) gv[j] = (has_mask && !(mv[j] > 0.f && mv[j] < hi)) ? 0.f : gv[j];
#pragma unroll
            for (int j = 0; j < VEC; ++j) s1[j] += gv[j];
            if (y) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) s2[j] += gv[j] * ((yv[j] - mu[j]) * is[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        red[0][threadIdx.x][j] = s1[j];
        red[1][threadIdx.x][j] = s2[j];
    }
    __syncthreads();
    // the row lanes meet in a fixed tree (lane ty takes lane ty + stride, stride = ty_n / 2 ... 1): log2(ty_n) parallel steps.  The first version had the
    // tx_n threads of row lane 0 add all ty_n lanes one after the other - with 32 channels that is 4 threads x 1024 LDS reads, 4 of the launch's 13 us.
    for (int stride = ty_n >> 1; stride > 0; stride >>= 1) {
        if (ty < stride) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                red[0][threadIdx.x][j] += red[0][threadIdx.x + stride * tx_n][j];
                red[1][threadIdx.x][j] += red[1][threadIdx.x + stride * tx_n][j];
            }
        }
        __syncthreads();
    }
    if (ty == 0 && c0 < C) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (c0 + j < C) {
                partial[((long)blockIdx.x * 2 + 0) * C + c0 + j] = red[0][tx][j];
                partial[((long)blockIdx.x * 2 + 1) * C + c0 + j] = red[1][tx][j];
            }
        }
    }
}

// out1[c] (+)= sum over blocks of partial[blk][0][c], out2 likewise: two channels x 128 block lanes per workgroup (lane l adds the blocks l, l + 128, ...; the
// lane sums are added in lane order; double) - see gbn_finalize_kernel for why not 8 x 32
__global__ __launch_bounds__(256) void gcolsum_final_kernel(const float* partial, int blocks, int C, float* out1, float* out2, int accumulate) {
    __shared__ double red[2][2][FIN_LANES];
    const int cx = threadIdx.x / FIN_LANES, ry = threadIdx.x % FIN_LANES;
    const int c = blockIdx.x * 2 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int b = ry; b < blocks; b += FIN_LANES) {
            s1 += (double)partial[((long)b * 2 + 0) * C + c];
            s2 += (double)partial[((long)b * 2 + 1) * C + c];
        }
    red[0][cx][ry] = s1;
    red[1][cx][ry] = s2;
    __syncthreads();
    if (ry == 0 && c < C) {
        s1 = s2 = 0.0;
        const int live = blocks < FIN_LANES ? blocks : FIN_LANES;
        for (int k = 0; k < live; ++k) {
            s1 += red[0][cx][k];
            s2 += red[1][cx][k];
        }
        if (out1) out1[c] = accumulate ? out1[c] + (float)s1 : (float)s1;
        if (out2) out2[c] = accumulate ? out2[c] + (float)s2 : (float)s2;
    }
}

// dy = gamma * invstd * (g' - dbeta / n - xhat * dgamma / n)
template <int VEC, typename TG, typename TM>
__global__ __launch_bounds__(256) void gbn_bwd_apply_kernel(const TG* g, long ldg, const __bf16* y, long ldy, const TM* mask, long ldm, const float* mean,
                                                            const float* invstd, const float* gamma, const float* dbeta, const float* dgamma,
                                                            float inv_count, __bf16* dy, long lddy, long M, int C, float hi) {
    const int cv = C / VEC;
    const long n = M * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / cv;
        const int c0 = (int)(e - m * cv) * VEC;
        float gv[VEC], yv[VEC], mv[VEC], o[VEC];
        ldv<VEC>(g + m * ldg + c0, gv);
        ldv<VEC>(y + m * ldy + c0, yv);
        ldv<VEC>(mask ? mask + m * ldm + c0 : reinterpret_cast<const TM*>(g_nzero), mv);       // (unconditional: see gcolsum_partial_kernel)
        {
            const bool has_mask = mask != nullptr;
#pragma unroll
            for (int j = 0; j < VEC; ++j) gv[j] = (has_mask && !(mv[j] > 0.f && mv[j] < hi)) ? 0.f : gv[j];
        }
        float is[VEC], mu[VEC], ga[VEC], db[VEC], dg[VEC];
        ldp<VEC>(invstd + c0, is);
        ldp<VEC>(mean + c0, mu);
        ldp<VEC>(dbeta + c0, db);
        ldp<VEC>(dgamma + c0, dg);
        ldp<VEC>(gamma ? gamma + c0 : reinterpret_cast<const float*>(g_nzero), ga);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xhat = (yv[j] - mu[j]) * is[j];
            o[j] = (gamma ? ga[j] : 1.f) * is[j] * (gv[j] - db[j] * inv_count - xhat * dg[j] * inv_count);
        }
        stv<VEC>(dy + m * lddy + c0, o);
    }
}

// ------------------------------------------------------------------------------------------------ elementwise on views
enum { OP_ADD = 0, OP_MUL = 1, OP_COPY = 2, OP_RELU_MASK = 3 /* a where b > 0 else 0 */, OP_MULRELU = 4 /* relu(a * b) */ };
template <int OP, int VEC, typename TA, typename TB, typename TO>
__global__ __launch_bounds__(256) void gbinary_kernel(const TA* a, long lda, const TB* b, long ldb, TO* out, long ldo, long M, int C) {
    const int cv = C / VEC;
    const long n = M * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / cv;
        const int c0 = (int)(e - m * cv) * VEC;
        float av[VEC], bv[VEC], r[VEC];
        ldv<VEC>(a + m * lda + c0, av);
        if constexpr (OP != OP_COPY) ldv<VEC>(b + m * ldb + c0, bv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if constexpr (OP == OP_COPY) r[j] = av[j];
            else r[j] = OP == OP_ADD ? av[j] + bv[j] : (OP == OP_MUL ? av[j] * bv[j] : (OP == OP_MULRELU ? fmaxf(av[j] * bv[j], 0.f) : (bv[j] > 0.f ? av[j] : 0.f)));
        }
        stv<VEC>(out + m * ldo + c0, r);
    }
}

// ------------------------------------------------------------------------------------------------ average pools (NHWC bf16 views)
// include_pad != 0: AvgPool2d(k, s, p) with torch's default count_include_pad=True (Res2Net_v1b.py:40: divisor k*k);
// include_pad == 0: AvgPool2d(s, s, ceil_mode=True, count_include_pad=False) of the downsample path (Res2Net_v1b.py:122-123).
struct PoolP {
    int B, H, W, C, Ho, Wo, k, s, p, include_pad;
    long ldx, ldo;
};
// thread = (output pixel, VEC channels): the pixel coordinates are divided out once per VEC channels and every tap is one vector access
// (the per-element first version: 58 us for the 26-channel stage pools, 150 us for the 256-channel downsample pool of layer2.0)
template <int VEC>
__global__ __launch_bounds__(256) void gavgpool_fwd_kernel(const __bf16* x, __bf16* out, PoolP q) {
    const int cv = q.C / VEC;
    const long n = (long)q.B * q.Ho * q.Wo * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % cv) * VEC;
        long m = e / cv;
        const int ow = (int)(m % q.Wo);
        const long t = m / q.Wo;
        const int oh = (int)(t % q.Ho), b = (int)(t / q.Ho);
        float s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = 0.f;
        int cnt = 0;
        for (int ky = 0; ky < q.k; ++ky) {
            const int ih = oh * q.s - q.p + ky;
            if ((unsigned)ih >= (unsigned)q.H) continue;
            for (int kx = 0; kx < q.k; ++kx) {
                const int iw = ow * q.s - q.p + kx;
                if ((unsigned)iw >= (unsigned)q.W) continue;
                float v[VEC];
                ldv<VEC>(x + (((long)b * q.H + ih) * q.W + iw) * q.ldx + c, v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) s[j] += v[j];
                ++cnt;
            }
        }
        const float div = q.include_pad ? (float)(q.k * q.k) : (float)(cnt > 0 ? cnt : 1);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] /= div;
        stv<VEC>(out + m * q.ldo + c, s);
    }
}
// dx[b][ih][iw][c] = sum over the windows that contain (ih, iw) of dout / divisor(window); q.ldx / q.ldo are the strides of dx / dout
template <int VEC>
__global__ __launch_bounds__(256) void gavgpool_bwd_kernel(const __bf16* dout, __bf16* dx, PoolP q) {
    const int cv = q.C / VEC;
    const long n = (long)q.B * q.H * q.W * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % cv) * VEC;
        long m = e / cv;
        const int iw = (int)(m % q.W);
        const long t = m / q.W;
        const int ih = (int)(t % q.H), b = (int)(t / q.H);
        float s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = 0.f;
        for (int oh = (ih + q.p) / q.s; oh >= 0 && oh * q.s - q.p + q.k - 1 >= ih; --oh) {
            if (oh >= q.Ho) continue;
            for (int ow = (iw + q.p) / q.s; ow >= 0 && ow * q.s - q.p + q.k - 1 >= iw; --ow) {
                if (ow >= q.Wo) continue;
                float div = (float)(q.k * q.k);
                if (!q.include_pad) {
                    const int h0 = max(oh * q.s - q.p, 0), h1 = min(oh * q.s - q.p + q.k, q.H);
                    const int w0 = max(ow * q.s - q.p, 0), w1 = min(ow * q.s - q.p + q.k, q.W);
                    div = (float)((h1 - h0) * (w1 - w0));
                }
                float v[VEC];
                ldv<VEC>(dout + (((long)b * q.Ho + oh) * q.Wo + ow) * q.ldo + c, v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) s[j] += v[j] / div;
            }
        }
        stv<VEC>(dx + m * q.ldx + c, s);
    }
}

// ------------------------------------------------------------------------------------------------ max pools (NHWC bf16 views)
// MaxPool2d(k, s, p) of HarDNet (hardnet_68.py:213,233: 3/2/1 after the stem, 2/2 after the transitions): out + the winning tap (first
// maximum in scan order, like ATen) as one byte per element; backward routes dout to that tap (gather form, fixed order).
// thread = (pixel, VEC channels); every tap is read unconditionally at a clamped position and selected afterwards (a branch around a load makes hipcc
// wait for each load before the next: the per-element first version took 273 us for the backward of HarDNet's 2 x 2 pools at 6 x 180 x 320 x 128)
template <int VEC>
__global__ __launch_bounds__(256) void gmaxpool_fwd_kernel(const __bf16* x, __bf16* out, uint8_t* idx, PoolP q) {
    const int cv = q.C / VEC;
    const long n = (long)q.B * q.Ho * q.Wo * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % cv) * VEC;
        long m = e / cv;
        const int ow = (int)(m % q.Wo);
        const long t = m / q.Wo;
        const int oh = (int)(t % q.Ho), b = (int)(t / q.Ho);
        float best[VEC];
        int arg[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) best[j] = -INFINITY, arg[j] = 255;
        for (int ky = 0; ky < q.k; ++ky) {
            const int ih = oh * q.s - q.p + ky;
            const bool okh = (unsigned)ih < (unsigned)q.H;
            for (int kx = 0; kx < q.k; ++kx) {
                const int iw = ow * q.s - q.p + kx;
                const bool ok = okh && (unsigned)iw < (unsigned)q.W;
                float v[VEC];
                ldv<VEC>(x + (((long)b * q.H + (okh ? ih : 0)) * q.W + ((unsigned)iw < (unsigned)q.W ? iw : 0)) * q.ldx + c, v);
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (ok && (v[j] > best[j] || arg[j] == 255)) best[j] = v[j], arg[j] = ky * q.k + kx;
            }
        }
        stv<VEC>(out + m * q.ldo + c, best);
#pragma unroll
        for (int j = 0; j < VEC; ++j) idx[m * q.C + c + j] = (uint8_t)arg[j];
    }
}
template <int VEC>
__global__ __launch_bounds__(256) void gmaxpool_bwd_kernel(const __bf16* dout, const uint8_t* idx, __bf16* dx, PoolP q) {
    const int cv = q.C / VEC;
    const long n = (long)q.B * q.H * q.W * cv;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % cv) * VEC;
        long m = e / cv;
        const int iw = (int)(m % q.W);
        const long t = m / q.W;
        const int ih = (int)(t % q.H), b = (int)(t / q.H);
        float s[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] = 0.f;
        for (int ky = 0; ky < q.k; ++ky) {
            const int nh = ih + q.p - ky;
            const int oh = nh / q.s;
            const bool okh = nh >= 0 && oh * q.s == nh && oh < q.Ho;
            for (int kx = 0; kx < q.k; ++kx) {
                const int nw = iw + q.p - kx;
                const int ow = nw / q.s;
                const bool ok = okh && nw >= 0 && ow * q.s == nw && ow < q.Wo;
                const long mo = ((long)b * q.Ho + (okh ? oh : 0)) * q.Wo + ((nw >= 0 && ow < q.Wo) ? ow : 0);
                float g[VEC];
                ldv<VEC>(dout + mo * q.ldo + c, g);
                const uint8_t* ip = idx + mo * q.C + c;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (ok && ip[j] == ky * q.k + kx) s[j] += g[j];
            }
        }
        stv<VEC>(dx + m * q.ldx + c, s);
    }
}

// ------------------------------------------------------------------------------------------------ cross-entropy on NHWC fp32 logits
// CrossEntropyLoss(ignore_index) (gald_trainer.py:66-84, one per deep-supervision head) on logits [M][K] (view, K <= 32), labels int64 [M].
// Two levels, fixed order: per-block (loss, valid, out-of-range) partials, then one block sums them into loss_out[0..2].
constexpr int CE_ROWS = 1024;
__global__ __launch_bounds__(256) void gce_partial_kernel(const float* logits, long ld, const long* labels, int ignore, long M, int K, float* partial) {
    __shared__ float red[3][256];
    float loss = 0.f, cnt = 0.f, bad = 0.f;
    const long r0 = (long)blockIdx.x * CE_ROWS;
    for (int r = threadIdx.x; r < CE_ROWS; r += 256) {
        const long m = r0 + r;
        if (m >= M) break;
        const long y = labels[m];
        if (y == ignore) continue;
        if (y < 0 || y >= K) {
            bad += 1.f;
            continue;
        }
        const float* z = logits + m * ld;
        float mx = z[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
        float se = 0.f;
        for (int k = 0; k < K; ++k) se += __expf(z[k] - mx);
        loss += __logf(se) - (z[y] - mx);
        cnt += 1.f;
    }
    red[0][threadIdx.x] = loss;
    red[1][threadIdx.x] = cnt;
    red[2][threadIdx.x] = bad;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int a = 0; a < 3; ++a) red[a][threadIdx.x] += red[a][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[(long)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ __launch_bounds__(256) void gce_final_kernel(const float* partial, int blocks, float* loss_out) {
    __shared__ double red[3][256];
    double a[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < blocks; b += 256)
        for (int k = 0; k < 3; ++k) a[k] += (double)partial[(long)b * 3 + k];
    for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = a[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_out[0] = red[1][0] > 0.0 ? (float)(red[0][0] / red[1][0]) : NAN;       // mean over the valid pixels (nan if none, like torch)
        loss_out[1] = (float)red[1][0];
        loss_out[2] = (float)red[2][0];
        loss_out[3] = 0.f;
    }
}
// dlogits = (softmax - onehot) * valid / n_valid * grad_scale, n_valid read from loss_out[1] on the device.  A block owns 256 pixels: each
// thread reduces its pixel's K logits to (max, 1 / sum), then the block walks the 256 x K elements linearly (coalesced reads and writes when
// the rows are dense).
__global__ __launch_bounds__(256) void gce_bwd_kernel(const float* logits, long ld, const long* labels, int ignore, long M, int K, const float* loss_out, float grad_scale,
                                                      float* dlogits, long ldd) {
    __shared__ float smx[256], sinv[256];
    __shared__ int slab[256];
    const float inv = loss_out[1] > 0.f ? grad_scale / loss_out[1] : 0.f;
    const long m0 = (long)blockIdx.x * 256;
    {
        const long m = m0 + threadIdx.x;
        float mx = 0.f, r = 0.f;
        int y = -1;
        if (m < M) {
            const long yy = labels[m];
            if (yy != ignore && yy >= 0 && yy < K) {
                y = (int)yy;
                const float* z = logits + m * ld;
                mx = z[0];
                for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
                float se = 0.f;
                for (int k = 0; k < K; ++k) se += __expf(z[k] - mx);
                r = inv / se;
            }
        }
        smx[threadIdx.x] = mx;
        sinv[threadIdx.x] = r;
        slab[threadIdx.x] = y;
    }
    __syncthreads();
    const int n = 256 * K;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int p = e / K, k = e - p * K;
        const long m = m0 + p;
        if (m >= M) break;
        const int y = slab[p];
        float d = 0.f;
        if (y >= 0) d = __expf(logits[m * ld + k] - smx[p]) * sinv[p] - (k == y ? inv : 0.f);
        dlogits[m * ldd + k] = d;
    }
}

// ------------------------------------------------------------------------------------------------ bilinear resize (both conventions)
// ATen's upsample_bilinear2d: src = align ? scale * dst : max(scale * (dst + 0.5) - 0.5, 0); i0 = (int)src; i1 = i0 + (i0 < in - 1);
// l1 = src - i0.  scale is computed on the host the way torch does (align: (in - 1) / (out - 1); otherwise 1 / scale_factor when the
// caller gave a scale_factor - F.interpolate(..., scale_factor=s) of PraNet_Res2Net.py:127-177 - or in / out when it gave a size).
struct ResizeP {
    int B, H, W, C, Ho, Wo, align;
    float sh, sw;
    long ldx, ldo;
};
__device__ __forceinline__ void rs_src(int d, float scale, int align, int in, int& i0, int& p, float& l1) {
    float s = align ? scale * (float)d : fmaxf(scale * ((float)d + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    p = i0 < in - 1 ? 1 : 0;
    l1 = s - (float)i0;
}
template <typename T>
__global__ __launch_bounds__(256) void gresize_fwd_kernel(const T* x, T* out, ResizeP q) {
    const long n = (long)q.B * q.Ho * q.Wo * q.C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % q.C);
        long m = e / q.C;
        const int ow = (int)(m % q.Wo);
        const long t = m / q.Wo;
        const int oh = (int)(t % q.Ho), b = (int)(t / q.Ho);
        int h0, hp, w0, wp;
        float hl, wl;
        rs_src(oh, q.sh, q.align, q.H, h0, hp, hl);
        rs_src(ow, q.sw, q.align, q.W, w0, wp, wl);
        const T* r0 = x + (((long)b * q.H + h0) * q.W) * q.ldx + c;
        const T* r1 = r0 + (long)hp * q.W * q.ldx;
        const float v00 = ldf(r0 + (long)w0 * q.ldx), v01 = ldf(r0 + (long)(w0 + wp) * q.ldx);
        const float v10 = ldf(r1 + (long)w0 * q.ldx), v11 = ldf(r1 + (long)(w0 + wp) * q.ldx);
        const float v = (1.f - hl) * ((1.f - wl) * v00 + wl * v01) + hl * ((1.f - wl) * v10 + wl * v11);
        stf(out + m * q.ldo + c, v);
    }
}
// gather form of the backward: one thread per SOURCE element sums, in ascending destination order, the contributions of every
// destination pixel whose two-tap footprint touches it (the candidate range comes from inverting the monotone source map, widened
// by two on each side; each candidate is tested with the forward formula itself).  q.ldx / q.ldo: strides of dx / dout.
template <typename T>
__global__ __launch_bounds__(256) void gresize_bwd_kernel(const T* dout, T* dx, ResizeP q) {
    const long n = (long)q.B * q.H * q.W * q.C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % q.C);
        long m = e / q.C;
        const int iw = (int)(m % q.W);
        const long t = m / q.W;
        const int ih = (int)(t % q.H), b = (int)(t / q.H);
        auto range = [&](int i, float scale, int out, int& lo, int& hi) {
            if (scale <= 0.f) {
                lo = 0;
                hi = out - 1;
                return;
            }
            const float off = q.align ? 0.f : 0.5f;
            lo = (int)floorf(((float)(i - 1) + off) / scale - off) - 2;
            hi = (int)ceilf(((float)(i + 1) + off) / scale - off) + 2;
            if (lo < 0) lo = 0;
            if (hi > out - 1) hi = out - 1;
        };
        int hlo, hhi, wlo, whi;
        range(ih, q.sh, q.Ho, hlo, hhi);
        range(iw, q.sw, q.Wo, wlo, whi);
        float s = 0.f;
        for (int oh = hlo; oh <= hhi; ++oh) {
            int h0, hp;
            float hl;
            rs_src(oh, q.sh, q.align, q.H, h0, hp, hl);
            float wh = 0.f;
            if (h0 == ih) wh += 1.f - hl;
            if (h0 + hp == ih) wh += hl;          // (a clamped second tap lands on the same row)
            if (wh == 0.f) continue;
            float rs = 0.f;
            const T* drow = dout + (((long)b * q.Ho + oh) * q.Wo) * q.ldo + c;
            for (int ow = wlo; ow <= whi; ++ow) {
                int w0, wp;
                float wl;
                rs_src(ow, q.sw, q.align, q.W, w0, wp, wl);
                float ww = 0.f;
                if (w0 == iw) ww += 1.f - wl;
                if (w0 + wp == iw) ww += wl;
                if (ww != 0.f) rs += ww * ldf(drow + (long)ow * q.ldo);
            }
            s += wh * rs;
        }
        stf(dx + m * q.ldx + c, s);
    }
}

// The same sum with one WAVE per source element, for large magnifications (x8 .. x32: 20^2 .. 70^2 candidates and only B * h * w * C sources):
// lane l takes the candidate rows l, l + 64, ... of the rectangle (each row summed left to right), the 64 partial sums are combined by a fixed
// butterfly.  Deterministic, a different (equally valid) association than the thread-per-element kernel.
template <typename T>
__global__ __launch_bounds__(256) void gresize_bwd_wave_kernel(const T* dout, T* dx, ResizeP q) {
    const int lane = threadIdx.x & 63;
    const long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long n = (long)q.B * q.H * q.W * q.C;
    if (e >= n) return;
    const int c = (int)(e % q.C);
    long m = e / q.C;
    const int iw = (int)(m % q.W);
    const long t = m / q.W;
    const int ih = (int)(t % q.H), b = (int)(t / q.H);
    auto range = [&](int i, float scale, int out, int& lo, int& hi) {
        if (scale <= 0.f) {
            lo = 0;
            hi = out - 1;
            return;
        }
        const float off = q.align ? 0.f : 0.5f;
        lo = (int)floorf(((float)(i - 1) + off) / scale - off) - 2;
        hi = (int)ceilf(((float)(i + 1) + off) / scale - off) + 2;
        if (lo < 0) lo = 0;
        if (hi > out - 1) hi = out - 1;
    };
    int hlo, hhi, wlo, whi;
    range(ih, q.sh, q.Ho, hlo, hhi);
    range(iw, q.sw, q.Wo, wlo, whi);
    float s = 0.f;
    for (int oh = hlo + lane; oh <= hhi; oh += 64) {
        int h0, hp;
        float hl;
        rs_src(oh, q.sh, q.align, q.H, h0, hp, hl);
        float wh = 0.f;
        if (h0 == ih) wh += 1.f - hl;
        if (h0 + hp == ih) wh += hl;
        if (wh == 0.f) continue;
        float rs = 0.f;
        const T* drow = dout + (((long)b * q.Ho + oh) * q.Wo) * q.ldo + c;
        for (int ow = wlo; ow <= whi; ++ow) {
            int w0, wp;
            float wl;
            rs_src(ow, q.sw, q.align, q.W, w0, wp, wl);
            float ww = 0.f;
            if (w0 == iw) ww += 1.f - wl;
            if (w0 + wp == iw) ww += wl;
            if (ww != 0.f) rs += ww * ldf(drow + (long)ow * q.ldo);
        }
        s += wh * rs;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) stf(dx + m * q.ldx + c, s);
}

// The same sum for feature maps (C >= 8): one wave per source PIXEL, lanes over its channels (coalesced reads of the destination pixels'
// channel vectors), every lane walks the candidate rectangle in ascending order - no cross-lane reduction at all.
template <typename T>
__global__ __launch_bounds__(256) void gresize_bwd_pix_kernel(const T* dout, T* dx, ResizeP q) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)q.B * q.H * q.W) return;
    const int iw = (int)(m % q.W);
    const long t = m / q.W;
    const int ih = (int)(t % q.H), b = (int)(t / q.H);
    auto range = [&](int i, float scale, int out, int& lo, int& hi) {
        if (scale <= 0.f) {
            lo = 0;
            hi = out - 1;
            return;
        }
        const float off = q.align ? 0.f : 0.5f;
        lo = (int)floorf(((float)(i - 1) + off) / scale - off) - 2;
        hi = (int)ceilf(((float)(i + 1) + off) / scale - off) + 2;
        if (lo < 0) lo = 0;
        if (hi > out - 1) hi = out - 1;
    };
    int hlo, hhi, wlo, whi;
    range(ih, q.sh, q.Ho, hlo, hhi);
    range(iw, q.sw, q.Wo, wlo, whi);
    {   // the range is padded by two on both sides: trim the columns that do not touch iw, so that the unconditional loads below fetch contributors only
        auto touches = [&](int ow) {
            int w0, wp;
            float wl;
            rs_src(ow, q.sw, q.align, q.W, w0, wp, wl);
            return w0 == iw || w0 + wp == iw;
        };
        while (wlo < whi && !touches(wlo)) ++wlo;
        while (whi > wlo && !touches(whi)) --whi;
    }
    for (int c = lane; c < q.C; c += 64) {
        float s = 0.f;
        for (int oh = hlo; oh <= hhi; ++oh) {
            int h0, hp;
            float hl;
            rs_src(oh, q.sh, q.align, q.H, h0, hp, hl);
            float wh = 0.f;
            if (h0 == ih) wh += 1.f - hl;
            if (h0 + hp == ih) wh += hl;
            if (wh == 0.f) continue;
            float rsum = 0.f;
            const T* drow = dout + (((long)b * q.Ho + oh) * q.Wo) * q.ldo + c;
            // eight candidates per trip: weights first, then eight UNCONDITIONAL loads (a candidate past the range re-reads the last one with weight 0),
            // then the adds in ascending order - with the load inside `if (ww != 0)` every contributing pixel was its own memory round trip (a 1/32 -> 1/4
            // gradient of 256 channels: 1 024 dependent loads per wave, 650 us)
            for (int ow = wlo; ow <= whi; ow += 8) {
                float ww[8], v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int oc = min(ow + u, whi);
                    int w0, wp;
                    float wl;
                    rs_src(oc, q.sw, q.align, q.W, w0, wp, wl);
                    float w_ = 0.f;
                    w_ += w0 == iw ? 1.f - wl : 0.f;
                    w_ += w0 + wp == iw ? wl : 0.f;
                    ww[u] = ow + u <= whi ? w_ : 0.f;
                    v[u] = ldf(drow + (long)oc * q.ldo);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) rsum = ww[u] != 0.f ? __builtin_fmaf(ww[u], v[u], rsum) : rsum;
            }
            s += wh * rsum;
        }
        stf(dx + m * q.ldx + c, s);
    }
}

// 8-channel forms for bf16 feature maps whose views allow 16-byte accesses (C % 8 == 0, 16-byte aligned rows): thread item = (pixel, 8-channel
// chunk), four 16-byte loads and one store forward; backward the same candidate walk as gresize_bwd_pix_kernel (same order of additions per
// channel: identical results) with one 16-byte load per contributing destination pixel instead of 2-byte loads per lane.
__global__ __launch_bounds__(256) void gresize_fwd8_kernel(const __bf16* x, __bf16* out, ResizeP q) {
    const int c8n = q.C >> 3;
    const long n = (long)q.B * q.Ho * q.Wo * c8n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % c8n) * 8;
        long m = e / c8n;
        const int ow = (int)(m % q.Wo);
        const long t = m / q.Wo;
        const int oh = (int)(t % q.Ho), b = (int)(t / q.Ho);
        int h0, hp, w0, wp;
        float hl, wl;
        rs_src(oh, q.sh, q.align, q.H, h0, hp, hl);
        rs_src(ow, q.sw, q.align, q.W, w0, wp, wl);
        const __bf16* r0 = x + (((long)b * q.H + h0) * q.W) * q.ldx + c;
        const __bf16* r1 = r0 + (long)hp * q.W * q.ldx;
        const bf16x8 v00 = *reinterpret_cast<const bf16x8*>(r0 + (long)w0 * q.ldx), v01 = *reinterpret_cast<const bf16x8*>(r0 + (long)(w0 + wp) * q.ldx);
        const bf16x8 v10 = *reinterpret_cast<const bf16x8*>(r1 + (long)w0 * q.ldx), v11 = *reinterpret_cast<const bf16x8*>(r1 + (long)(w0 + wp) * q.ldx);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            o[k] = (__bf16)((1.f - hl) * ((1.f - wl) * (float)v00[k] + wl * (float)v01[k]) + hl * ((1.f - wl) * (float)v10[k] + wl * (float)v11[k]));
        *reinterpret_cast<bf16x8*>(out + m * q.ldo + c) = o;
    }
}

__global__ __launch_bounds__(256) void gresize_bwd8_kernel(const __bf16* dout, __bf16* dx, ResizeP q) {
    const int c8n = q.C >> 3;
    const long n = (long)q.B * q.H * q.W * c8n;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e % c8n) * 8;
    const long m = e / c8n;
    const int iw = (int)(m % q.W);
    const long t = m / q.W;
    const int ih = (int)(t % q.H), b = (int)(t / q.H);
    auto range = [&](int i, float scale, int out, int& lo, int& hi) {
        if (scale <= 0.f) {
            lo = 0;
            hi = out - 1;
            return;
        }
        const float off = q.align ? 0.f : 0.5f;
        lo = (int)floorf(((float)(i - 1) + off) / scale - off) - 2;
        hi = (int)ceilf(((float)(i + 1) + off) / scale - off) + 2;
        if (lo < 0) lo = 0;
        if (hi > out - 1) hi = out - 1;
    };
    int hlo, hhi, wlo, whi;
    range(ih, q.sh, q.Ho, hlo, hhi);
    range(iw, q.sw, q.Wo, wlo, whi);
    {   // the range is padded by two on both sides: trim the columns that do not touch iw, so that the unconditional loads below fetch contributors only
        auto touches = [&](int ow) {
            int w0, wp;
            float wl;
            rs_src(ow, q.sw, q.align, q.W, w0, wp, wl);
            return w0 == iw || w0 + wp == iw;
        };
        while (wlo < whi && !touches(wlo)) ++wlo;
        while (whi > wlo && !touches(whi)) --whi;
    }
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    for (int oh = hlo; oh <= hhi; ++oh) {
        int h0, hp;
        float hl;
        rs_src(oh, q.sh, q.align, q.H, h0, hp, hl);
        float wh = 0.f;
        if (h0 == ih) wh += 1.f - hl;
        if (h0 + hp == ih) wh += hl;
        if (wh == 0.f) continue;
        float rsum[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rsum[k] = 0.f;
        const __bf16* drow = dout + (((long)b * q.Ho + oh) * q.Wo) * q.ldo + c;
        for (int ow = wlo; ow <= whi; ow += 4) {                 // (four candidates per trip, loads unconditional: see gresize_bwd_pix_kernel)
            float ww[4];
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int oc = min(ow + u, whi);
                int w0, wp;
                float wl;
                rs_src(oc, q.sw, q.align, q.W, w0, wp, wl);
                float w_ = 0.f;
                w_ += w0 == iw ? 1.f - wl : 0.f;
                w_ += w0 + wp == iw ? wl : 0.f;
                ww[u] = ow + u <= whi ? w_ : 0.f;
                v[u] = *reinterpret_cast<const bf16x8*>(drow + (long)oc * q.ldo);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) rsum[k] = ww[u] != 0.f ? __builtin_fmaf(ww[u], (float)v[u][k], rsum[k]) : rsum[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += wh * rsum[k];
    }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (__bf16)s[k];
    *reinterpret_cast<bf16x8*>(dx + m * q.ldx + c) = o;
}

// ------------------------------------------------------------------------------------------------ reverse attention (PraNet_Res2Net.py:131-133)
// out[m][c] = (1 - sigmoid(gate[m])) * feat[m][c]   ( `-1*(torch.sigmoid(crop)) + 1` expanded over the channels, times the feature )
__global__ __launch_bounds__(256) void gra_fwd_kernel(const float* gate, const __bf16* feat, long ldf_, __bf16* out, long ldo, long M, int C) {
    const long n = M * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / C;
        const int c = (int)(e - m * C);
        const float a = -1.f * (1.f / (1.f + __expf(-gate[m]))) + 1.f;
        out[m * ldo + c] = (__bf16)(a * (float)feat[m * ldf_ + c]);
    }
}
// dfeat[m][c] = (1 - s) * dy[m][c];  dgate[m] = -s (1 - s) * sum_c feat[m][c] * dy[m][c]     one wave per pixel, fixed-order butterfly
__global__ __launch_bounds__(256) void gra_bwd_kernel(const float* gate, const __bf16* feat, long ldf_, const __bf16* dy, long lddy, __bf16* dfeat, long lddf,
                                                      float* dgate, long M, int C) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float s = 1.f / (1.f + __expf(-gate[m]));
    const float a = 1.f - s;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = (float)dy[m * lddy + c];
        dot += (float)feat[m * ldf_ + c] * d;
        dfeat[m * lddf + c] = (__bf16)(a * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if (lane == 0) dgate[m] = -s * a * dot;
}

inline int grid_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" {

int mi_gbn_finalize(const float* partials, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, float* mean_out, float* invstd_out, float* scale_out, float* shift_out, void* stream) {
    MI_REQUIRE(partials && mean_out && invstd_out && scale_out && shift_out, "mi_gbn_finalize: null operand");
    MI_REQUIRE(tiles > 0 && C > 0 && count > 0, "mi_gbn_finalize: empty shape");
    MI_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mi_gbn_finalize: running_mean and running_var come together");
    hipLaunchKernelGGL(gbn_finalize_kernel, dim3((C + 1) / 2), dim3(256), 0, (hipStream_t)stream, partials, tiles, C, (double)count, gamma, beta,
                       running_mean, running_var, momentum, eps, mean_out, invstd_out, scale_out, shift_out);
    MI_CHECK_LAUNCH("gbn_finalize_kernel");
    return MI_OK;
}

int mi_gbn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, float* scale, float* shift, int C,
                void* stream) {
    MI_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "mi_gbn_fold: null operand");
    hipLaunchKernelGGL(gbn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, scale, shift, C);
    MI_CHECK_LAUNCH("gbn_fold_kernel");
    return MI_OK;
}

int mi_gbn_apply(const void* y, long ldy, const float* scale, const float* shift, const void* add, long ldadd, void* out, long ldo, int out_f32, long M, int C,
                 int relu, void* stream) {
    MI_REQUIRE(y && scale && shift && out, "mi_gbn_apply: null operand");
    MI_REQUIRE(M > 0 && C > 0 && ldy >= C && ldo >= C && (!add || ldadd >= C), "mi_gbn_apply: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const __bf16* yy = (const __bf16*)y;
    const __bf16* aa = (const __bf16*)add;
    int vec = out_f32 ? 1 : common_vec(C, {{y, ldy}, {out, ldo}, {add, ldadd}});
    if (vec == 8 && ((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15)) vec = 2;
    if (out_f32) hipLaunchKernelGGL((gbn_apply_kernel<1, float>), dim3(grid_for(M * C)), dim3(256), 0, s, yy, ldy, scale, shift, aa, ldadd, (float*)out, ldo, M, C, relu);
    else if (vec == 8) hipLaunchKernelGGL((gbn_apply_kernel<8, __bf16>), dim3(grid_for(M * C / 8)), dim3(256), 0, s, yy, ldy, scale, shift, aa, ldadd, (__bf16*)out, ldo, M, C, relu);
    else if (vec == 2) hipLaunchKernelGGL((gbn_apply_kernel<2, __bf16>), dim3(grid_for(M * C / 2)), dim3(256), 0, s, yy, ldy, scale, shift, aa, ldadd, (__bf16*)out, ldo, M, C, relu);
    else hipLaunchKernelGGL((gbn_apply_kernel<1, __bf16>), dim3(grid_for(M * C)), dim3(256), 0, s, yy, ldy, scale, shift, aa, ldadd, (__bf16*)out, ldo, M, C, relu);
    MI_CHECK_LAUNCH("gbn_apply_kernel");
    return MI_OK;
}

int mi_gbn_apply_multi(const void* y, long ldy, const float* scale, const float* shift, const void* add, long ldadd, void* out, long ldo, long M, int C, int relu,
                       int n_extra, const int* c0, const int* c1, void* const* dst, const long* ldd, const void* const* add2, const long* lda2, void* stream) {
    MI_REQUIRE(y && scale && shift, "mi_gbn_apply_multi: null operand");
    MI_REQUIRE(M > 0 && C > 0 && ldy >= C && (!out || ldo >= C) && (!add || ldadd >= C), "mi_gbn_apply_multi: bad shape");
    MI_REQUIRE(n_extra >= 0 && n_extra <= APPLY_MAX_EXTRA && (n_extra == 0 || (c0 && c1 && dst && ldd && add2 && lda2)), "mi_gbn_apply_multi: 0 .. %d extra destinations", APPLY_MAX_EXTRA);
    MI_REQUIRE(out || n_extra > 0, "mi_gbn_apply_multi: nothing to write");
    GApplyExtras ex;
    ex.n = n_extra;
    int vec = common_vec(C, {{y, ldy}, {out, ldo}, {add, ldadd}});
    if (vec == 8 && ((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15)) vec = 2;
    for (int k = 0; k < APPLY_MAX_EXTRA; ++k) {
        const bool live = k < n_extra;
        ex.c0[k] = live ? c0[k] : 0;
        ex.c1[k] = live ? c1[k] : 0;
        ex.dst[k] = live ? (__bf16*)dst[k] : nullptr;
        ex.ldd[k] = live ? ldd[k] : 0;
        ex.add[k] = live ? (const __bf16*)add2[k] : nullptr;
        ex.lda[k] = live ? lda2[k] : 0;
        if (!live) continue;
        const int w = c1[k] - c0[k];
        MI_REQUIRE(dst[k] && c0[k] >= 0 && w > 0 && c1[k] <= C && ldd[k] >= w && (!add2[k] || lda2[k] >= w), "mi_gbn_apply_multi: extra destination %d: bad range or view", k);
        int vk = common_vec(w, {{dst[k], ldd[k]}, {add2[k], lda2[k]}});          // the widest access this range and its views allow ...
        while (vk > 1 && c0[k] % vk) vk = vk == 8 ? 2 : 1;                       // ... at a start channel that is a multiple of it
        if (vk < vec) vec = vk;
    }
    hipStream_t s = (hipStream_t)stream;
    const __bf16* yy = (const __bf16*)y;
    const __bf16* aa = (const __bf16*)add;
    if (vec == 8) hipLaunchKernelGGL((gbn_apply_multi_kernel<8>), dim3(grid_for(M * C / 8)), dim3(256), 0, s, yy, ldy, scale, shift, aa, ldadd, (__bf16*)out, ldo, M, C, relu, ex);
    else if (vec == 2) hipLaunchKernelGGL((gbn_apply_multi_kernel<2>), dim3(grid_for(M * C / 2)), dim3(256), 0, s, yy, ldy, scale, shift, aa, ldadd, (__bf16*)out, ldo, M, C, relu, ex);
    else hipLaunchKernelGGL((gbn_apply_multi_kernel<1>), dim3(grid_for(M * C)), dim3(256), 0, s, yy, ldy, scale, shift, aa, ldadd, (__bf16*)out, ldo, M, C, relu, ex);
    MI_CHECK_LAUNCH("gbn_apply_multi_kernel");
    return MI_OK;
}

size_t mi_gcolsum_workspace(long M, int C) { (void)M; return (size_t)770 * 2 * C * sizeof(float); }      // colsum_plan never uses more than 769 blocks

int mi_gbn_bwd_sums(const void* g, long ldg, int g_f32, const void* y, long ldy, const void* mask, long ldm, int mask_f32, const float* mean,
                    const float* invstd, long M, int C, float* dbeta, float* dgamma, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(g && workspace && (dbeta || dgamma), "mi_gbn_bwd_sums: null operand");
    MI_REQUIRE(M > 0 && C > 0 && ldg >= C, "mi_gbn_bwd_sums: bad shape");
    MI_REQUIRE(!y || (mean && invstd && ldy >= C), "mi_gbn_bwd_sums: y needs mean / invstd");
    const float hi = (mask_f32 & 2) ? 6.f : INFINITY;      // bit 1 of the mask flags: the mask is a ReLU6 output (gradient only where 0 < out < 6)
    mask_f32 &= 1;
    MI_REQUIRE(g_f32 || !mask_f32, "mi_gbn_bwd_sums: an fp32 mask goes with an fp32 gradient");
    if (workspace_bytes < mi_gcolsum_workspace(M, C)) return mi_set_error(MI_ENOMEM, "mi_gbn_bwd_sums: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)workspace;
    const __bf16* yy = (const __bf16*)y;
    const int vec = (g_f32 || mask_f32) ? 1 : common_vec(C, {{g, ldg}, {y, ldy}, {mask, ldm}});
    const ColsumPlan q = colsum_plan(M, C, vec);
    const int blocks = q.blocks;
    const dim3 grid(blocks, ((C + vec - 1) / vec + q.tx - 1) / q.tx);
#define CSK(V, TG, TM) hipLaunchKernelGGL((gcolsum_partial_kernel<V, TG, TM>), grid, dim3(256), 0, s, (const TG*)g, ldg, yy, ldy, (const TM*)mask, ldm, mean, invstd, M, C, part, q.tx, q.rows_per_block, hi)
    if (g_f32 && mask_f32) CSK(1, float, float);
    else if (g_f32) CSK(1, float, __bf16);
    else if (vec == 8) CSK(8, __bf16, __bf16);
    else if (vec == 2) CSK(2, __bf16, __bf16);
    else CSK(1, __bf16, __bf16);
#undef CSK
    MI_CHECK_LAUNCH("gcolsum_partial_kernel");
    hipLaunchKernelGGL(gcolsum_final_kernel, dim3((C + 1) / 2), dim3(256), 0, s, part, blocks, C, dbeta, dgamma, accumulate);
    MI_CHECK_LAUNCH("gcolsum_final_kernel");
    return MI_OK;
}

int mi_gbn_bwd_apply(const void* g, long ldg, int g_f32, const void* y, long ldy, const void* mask, long ldm, int mask_f32, const float* mean,
                     const float* invstd, const float* gamma, const float* dbeta, const float* dgamma, float inv_count, void* dy, long lddy, long M, int C,
                     void* stream) {
    MI_REQUIRE(g && y && mean && invstd && dbeta && dgamma && dy, "mi_gbn_bwd_apply: null operand");
    MI_REQUIRE(M > 0 && C > 0 && ldg >= C && ldy >= C && lddy >= C, "mi_gbn_bwd_apply: bad shape");
    const float hi = (mask_f32 & 2) ? 6.f : INFINITY;
    mask_f32 &= 1;
    MI_REQUIRE(g_f32 || !mask_f32, "mi_gbn_bwd_apply: an fp32 mask goes with an fp32 gradient");
    hipStream_t s = (hipStream_t)stream;
    const __bf16* yy = (const __bf16*)y;
    int vec = (g_f32 || mask_f32) ? 1 : common_vec(C, {{g, ldg}, {y, ldy}, {mask, ldm}, {dy, lddy}});
    if (vec == 8 && ((reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(invstd) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dbeta) |
                      reinterpret_cast<uintptr_t>(dgamma)) & 15)) vec = 2;
    const dim3 grid(grid_for(M * C / vec));
#define BAK(V, TG, TM) hipLaunchKernelGGL((gbn_bwd_apply_kernel<V, TG, TM>), grid, dim3(256), 0, s, (const TG*)g, ldg, yy, ldy, (const TM*)mask, ldm, mean, invstd, gamma, dbeta, dgamma, inv_count, (__bf16*)dy, lddy, M, C, hi)
    if (g_f32 && mask_f32) BAK(1, float, float);
    else if (g_f32) BAK(1, float, __bf16);
    else if (vec == 8) BAK(8, __bf16, __bf16);
    else if (vec == 2) BAK(2, __bf16, __bf16);
    else BAK(1, __bf16, __bf16);
#undef BAK
    MI_CHECK_LAUNCH("gbn_bwd_apply_kernel");
    return MI_OK;
}

/* op: 0 add, 1 mul, 2 copy (b ignored), 3 relu mask (a where b > 0).  dtype: 0 = all bf16, 1 = all fp32, 2 = fp32 a -> bf16 out (copy: the cast),
 * 3 = bf16 a -> fp32 out (copy) */
int mi_gbinary(int op, int dtype, const void* a, long lda, const void* b, long ldb, void* out, long ldo, long M, int C, void* stream) {
    MI_REQUIRE(a && out && (b || op == OP_COPY), "mi_gbinary: null operand");
    MI_REQUIRE(M > 0 && C > 0 && lda >= C && ldo >= C && (op == OP_COPY || ldb >= C), "mi_gbinary: bad shape");
    MI_REQUIRE(op >= 0 && op <= 4 && dtype >= 0 && dtype <= 3 && (dtype < 2 || op == OP_COPY) && (op != OP_MULRELU || dtype == 0), "mi_gbinary: op %d / dtype %d", op, dtype);
    hipStream_t s = (hipStream_t)stream;
    const int vec = dtype == 0 ? common_vec(C, {{a, lda}, {out, ldo}, {op == OP_COPY ? nullptr : b, ldb}}) : 1;
    const dim3 grid(grid_for(M * C / vec));
#define GB(OP, V, TA, TB, TO) hipLaunchKernelGGL((gbinary_kernel<OP, V, TA, TB, TO>), grid, dim3(256), 0, s, (const TA*)a, lda, (const TB*)b, ldb, (TO*)out, ldo, M, C)
#define GBV(OP)                                            \
    do {                                                   \
        if (vec == 8) GB(OP, 8, __bf16, __bf16, __bf16);   \
        else if (vec == 2) GB(OP, 2, __bf16, __bf16, __bf16); \
        else GB(OP, 1, __bf16, __bf16, __bf16);            \
    } while (0)
    if (dtype == 0) {
        if (op == OP_ADD) GBV(OP_ADD);
        else if (op == OP_MUL) GBV(OP_MUL);
        else if (op == OP_COPY) GBV(OP_COPY);
        else if (op == OP_MULRELU) GBV(OP_MULRELU);
        else GBV(OP_RELU_MASK);
    } else if (dtype == 1) {
        if (op == OP_ADD) GB(OP_ADD, 1, float, float, float);
        else if (op == OP_MUL) GB(OP_MUL, 1, float, float, float);
        else if (op == OP_COPY) GB(OP_COPY, 1, float, float, float);
        else GB(OP_RELU_MASK, 1, float, float, float);
    } else if (dtype == 2) GB(OP_COPY, 1, float, float, __bf16);
    else GB(OP_COPY, 1, __bf16, __bf16, float);
#undef GBV
#undef GB
    MI_CHECK_LAUNCH("gbinary_kernel");
    return MI_OK;
}

int mi_gavgpool(const void* x, long ldx, void* out, long ldo, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int include_pad,
                int backward, void* stream) {
    MI_REQUIRE(x && out, "mi_gavgpool: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && k > 0 && stride > 0 && pad >= 0 && ldx >= C && ldo >= C, "mi_gavgpool: bad shape");
    MI_REQUIRE((Ho - 1) * stride - pad < H && (Wo - 1) * stride - pad < W, "mi_gavgpool: the last window starts outside the input");
    PoolP q{B, H, W, C, Ho, Wo, k, stride, pad, include_pad, ldx, ldo};
    hipStream_t s = (hipStream_t)stream;
    const int vec = common_vec(C, {{x, ldx}, {out, ldo}});
#define AP(V)                                                                                                                                              \
    do {                                                                                                                                                   \
        if (!backward) hipLaunchKernelGGL((gavgpool_fwd_kernel<V>), dim3(grid_for((long)B * Ho * Wo * (C / V))), dim3(256), 0, s, (const __bf16*)x, (__bf16*)out, q); \
        else hipLaunchKernelGGL((gavgpool_bwd_kernel<V>), dim3(grid_for((long)B * H * W * (C / V))), dim3(256), 0, s, (const __bf16*)out, (__bf16*)const_cast<void*>(x), q); \
    } while (0)
    if (vec == 8) AP(8);
    else if (vec == 2) AP(2);
    else AP(1);
#undef AP
    MI_CHECK_LAUNCH("gavgpool_kernel");
    return MI_OK;
}

int mi_gmaxpool(const void* x, long ldx, void* out, long ldo, uint8_t* idx, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int backward,
                void* stream) {
    MI_REQUIRE(x && out && idx, "mi_gmaxpool: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && k <= 15 && stride > 0 && pad >= 0 && ldx >= C && ldo >= C, "mi_gmaxpool: bad shape");
    MI_REQUIRE((H + 2 * pad - k) / stride + 1 == Ho && (W + 2 * pad - k) / stride + 1 == Wo && Ho > 0 && Wo > 0, "mi_gmaxpool: output %dx%d does not follow from input %dx%d", Ho, Wo, H, W);
    PoolP q{B, H, W, C, Ho, Wo, k, stride, pad, 0, ldx, ldo};
    hipStream_t s = (hipStream_t)stream;
    const int vec = common_vec(C, {{x, ldx}, {out, ldo}});
#define MP(V)                                                                                                                                                       \
    do {                                                                                                                                                            \
        if (!backward) hipLaunchKernelGGL((gmaxpool_fwd_kernel<V>), dim3(grid_for((long)B * Ho * Wo * (C / V))), dim3(256), 0, s, (const __bf16*)x, (__bf16*)out, idx, q); \
        else hipLaunchKernelGGL((gmaxpool_bwd_kernel<V>), dim3(grid_for((long)B * H * W * (C / V))), dim3(256), 0, s, (const __bf16*)out, (const uint8_t*)idx, (__bf16*)const_cast<void*>(x), q); \
    } while (0)
    if (vec == 8) MP(8);
    else if (vec == 2) MP(2);
    else MP(1);
#undef MP
    MI_CHECK_LAUNCH("gmaxpool_kernel");
    return MI_OK;
}

size_t mi_gce_workspace(long M) { return (size_t)((M + CE_ROWS - 1) / CE_ROWS) * 3 * sizeof(float); }

int mi_gce(const float* logits, long ld, const int64_t* labels, long M, int K, int ignore_index, float* loss_out, float* dlogits, long ldd, float grad_scale,
           void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(logits && labels && loss_out && workspace, "mi_gce: null operand");
    MI_REQUIRE(M > 0 && K > 0 && K <= 32 && ld >= K && (!dlogits || ldd >= K), "mi_gce: bad shape (K <= 32)");
    if (workspace_bytes < mi_gce_workspace(M)) return mi_set_error(MI_ENOMEM, "mi_gce: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((M + CE_ROWS - 1) / CE_ROWS);
    hipLaunchKernelGGL(gce_partial_kernel, dim3(blocks), dim3(256), 0, s, logits, ld, (const long*)labels, ignore_index, M, K, (float*)workspace);
    MI_CHECK_LAUNCH("gce_partial_kernel");
    hipLaunchKernelGGL(gce_final_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, blocks, loss_out);
    MI_CHECK_LAUNCH("gce_final_kernel");
    if (dlogits) {
        hipLaunchKernelGGL(gce_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, logits, ld, (const long*)labels, ignore_index, M, K, (const float*)loss_out, grad_scale, dlogits, ldd);
        MI_CHECK_LAUNCH("gce_bwd_kernel");
    }
    return MI_OK;
}

int mi_gresize(const void* x, long ldx, void* out, long ldo, int f32, int B, int H, int W, int C, int Ho, int Wo, int align_corners, float scale_h, float scale_w,
               int backward, void* stream) {
    MI_REQUIRE(x && out, "mi_gresize: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && ldx >= C && ldo >= C, "mi_gresize: bad shape");
    ResizeP q{B, H, W, C, Ho, Wo, align_corners, scale_h, scale_w, ldx, ldo};
    hipStream_t s = (hipStream_t)stream;
    const bool vec8 = !f32 && C % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (!backward) {
        if (vec8) {
            hipLaunchKernelGGL(gresize_fwd8_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 8))), dim3(256), 0, s, (const __bf16*)x, (__bf16*)out, q);
            MI_CHECK_LAUNCH("gresize_fwd8_kernel");
            return MI_OK;
        }
        const dim3 grid(grid_for((long)B * Ho * Wo * C));
        if (f32) hipLaunchKernelGGL((gresize_fwd_kernel<float>), grid, dim3(256), 0, s, (const float*)x, (float*)out, q);
        else hipLaunchKernelGGL((gresize_fwd_kernel<__bf16>), grid, dim3(256), 0, s, (const __bf16*)x, (__bf16*)out, q);
    } else {       // x = dx (written), out = dout (read)
        const long nsrc = (long)B * H * W * C;
        const float mag = (scale_h > 0.f ? 1.f / scale_h : (float)Ho) * (scale_w > 0.f ? 1.f / scale_w : (float)Wo);
        if (vec8 && mag <= 256.f) {                      // bf16 feature maps with 16-byte views: thread per (source pixel, 8 channels)
            const long items = (long)B * H * W * (C / 8);
            hipLaunchKernelGGL(gresize_bwd8_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, (const __bf16*)out, (__bf16*)const_cast<void*>(x), q);
        } else if (C >= 8 && mag >= 4.f) {               // feature maps / class logits: one wave per source pixel, lanes over the channels
            const dim3 grid((unsigned)(((long)B * H * W + 3) / 4));
            if (f32) hipLaunchKernelGGL((gresize_bwd_pix_kernel<float>), grid, dim3(256), 0, s, (const float*)out, (float*)const_cast<void*>(x), q);
            else hipLaunchKernelGGL((gresize_bwd_pix_kernel<__bf16>), grid, dim3(256), 0, s, (const __bf16*)out, (__bf16*)const_cast<void*>(x), q);
        } else if (mag >= 16.f && nsrc <= (1L << 22)) {  // many candidates per source element, few source elements: one wave each
            const dim3 grid((unsigned)((nsrc + 3) / 4));
            if (f32) hipLaunchKernelGGL((gresize_bwd_wave_kernel<float>), grid, dim3(256), 0, s, (const float*)out, (float*)const_cast<void*>(x), q);
            else hipLaunchKernelGGL((gresize_bwd_wave_kernel<__bf16>), grid, dim3(256), 0, s, (const __bf16*)out, (__bf16*)const_cast<void*>(x), q);
        } else {
            const dim3 grid(grid_for(nsrc));
            if (f32) hipLaunchKernelGGL((gresize_bwd_kernel<float>), grid, dim3(256), 0, s, (const float*)out, (float*)const_cast<void*>(x), q);
            else hipLaunchKernelGGL((gresize_bwd_kernel<__bf16>), grid, dim3(256), 0, s, (const __bf16*)out, (__bf16*)const_cast<void*>(x), q);
        }
    }
    MI_CHECK_LAUNCH("gresize_kernel");
    return MI_OK;
}

int mi_gra_fwd(const float* gate, const void* feat, long ldfeat, void* out, long ldo, long M, int C, void* stream) {
    MI_REQUIRE(gate && feat && out && M > 0 && C > 0 && ldfeat >= C && ldo >= C, "mi_gra_fwd: bad operand");
    hipLaunchKernelGGL(gra_fwd_kernel, dim3(grid_for(M * C)), dim3(256), 0, (hipStream_t)stream, gate, (const __bf16*)feat, ldfeat, (__bf16*)out, ldo, M, C);
    MI_CHECK_LAUNCH("gra_fwd_kernel");
    return MI_OK;
}

int mi_gra_bwd(const float* gate, const void* feat, long ldfeat, const void* dy, long lddy, void* dfeat, long lddf, float* dgate, long M, int C, void* stream) {
    MI_REQUIRE(gate && feat && dy && dfeat && dgate && M > 0 && C > 0 && ldfeat >= C && lddy >= C && lddf >= C, "mi_gra_bwd: bad operand");
    hipLaunchKernelGGL(gra_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gate, (const __bf16*)feat, ldfeat, (const __bf16*)dy, lddy,
                       (__bf16*)dfeat, lddf, dgate, M, C);
    MI_CHECK_LAUNCH("gra_bwd_kernel");
    return MI_OK;
}

}  // extern "C"
