// First kernels of the GALD / GCPA path (reference core/trainers/gald_trainer.py, core/models/classifiers/gcpacc/**):
//   * depthwise 3x3 convolution with bias (LocalAttenModule, contextagg/GALDNet.py:124-141: Conv2d(C, C, 3, groups=C, stride=2), no padding)
//     forward (+ BatchNorm tile statistics in the layout of gconv.hip), data gradient, weight / bias gradient;
//   * criss-cross attention (contextagg/ccnet.py:37-127): affinity of every pixel with its column and its row, -inf on the column's own
//     position, ONE softmax over the H + W candidates, aggregation of the values - forward and backward, without the reference's six
//     permute / contiguous / bmm round trips: one workgroup per pixel, the candidates' keys and values are read in place;
//   * the sigmoid gate x + x * sigmoid(g) of the local attention module (GALDNet.py:150-157).
// Operands are channel-slice views of NHWC bf16 tensors (pointer + elements per pixel row), like the rest of the g* family.
// Fixed summation orders everywhere: bitwise reproducible.
#include "mi_common.h"

namespace {

constexpr int DW_TILE = 128;       // pixels per block = rows of one BatchNorm statistics tile (the tile size of gconv.hip)

struct DwP {
    int B, H, W, C, Ho, Wo, stride, pad;
    long ldx, ldo;
};

// out[m][c] = bias[c] + sum_t x[src(m, t)][c] * w[c][t];  block = 128 pixels x 64 channels, thread = (channel, 4 pixel lanes)
__global__ __launch_bounds__(256) void gdw_fwd_kernel(const __bf16* x, const float* w, const float* bias, __bf16* out, float* stats, DwP q) {
    __shared__ float red[2][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    const long M = (long)q.B * q.Ho * q.Wo;
    float s1 = 0.f, s2 = 0.f;
    if (c < q.C) {
        float wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = w[c * 9 + t];
        const float bv = bias ? bias[c] : 0.f;
        // Two pixels per trip, their 18 taps loaded UNCONDITIONALLY at clamped coordinates and selected afterwards: with the loads inside the padding
        // branches every tap was its own memory round trip - 288 dependent loads per thread, 90 us for the 8 - 36 workgroups of a local-attention
        // module (the taps are added in the same order, so the sums keep their bits).
        for (int r0 = ry; r0 < DW_TILE; r0 += 8) {
            float xv[2][9];
            bool ok[2][9];
            long ms[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long m = (long)blockIdx.x * DW_TILE + r0 + 4 * u;
                ms[u] = m;
                const long mm = m < M ? m : M - 1;
                const int ow = (int)(mm % q.Wo);
                const long tt = mm / q.Wo;
                const int oh = (int)(tt % q.Ho), b = (int)(tt / q.Ho);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int ih = oh * q.stride - q.pad + ky;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iw = ow * q.stride - q.pad + kx;
                        ok[u][ky * 3 + kx] = ((unsigned)ih < (unsigned)q.H) & ((unsigned)iw < (unsigned)q.W);
                        xv[u][ky * 3 + kx] = (float)x[(((long)b * q.H + min(max(ih, 0), q.H - 1)) * q.W + min(max(iw, 0), q.W - 1)) * q.ldx + c];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (r0 + 4 * u >= DW_TILE || ms[u] >= M) break;
                float acc = bv;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = ok[u][t] ? __builtin_fmaf(xv[u][t], wt[t], acc) : acc;
                const __bf16 o = (__bf16)acc;
                out[ms[u] * q.ldo + c] = o;
                const float v = (float)o;                  // statistics of the ROUNDED output, as gconv.hip
                s1 += v;
                s2 += v * v;
            }
        }
    }
    if (stats) {
        red[0][ry][cx] = s1;
        red[1][ry][cx] = s2;
        __syncthreads();
        if (ry == 0 && c < q.C) {
            float* st = stats + (long)blockIdx.x * 2 * q.C;
            st[c] = red[0][0][cx] + red[0][1][cx] + red[0][2][cx] + red[0][3][cx];
            st[q.C + c] = red[1][0][cx] + red[1][1][cx] + red[1][2][cx] + red[1][3][cx];
        }
    }
}

// dx[b][ih][iw][c] = sum over the outputs whose window holds (ih, iw) of dy * w
__global__ __launch_bounds__(256) void gdw_dgrad_kernel(const __bf16* dy, const float* w, __bf16* dx, DwP q) {
    const long n = (long)q.B * q.H * q.W * q.C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % q.C);
        long m = e / q.C;
        const int iw = (int)(m % q.W);
        const long tt = m / q.W;
        const int ih = (int)(tt % q.H), b = (int)(tt / q.H);
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int nh = ih + q.pad - ky;
            if (nh < 0 || nh % q.stride) continue;
            const int oh = nh / q.stride;
            if (oh >= q.Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int nw = iw + q.pad - kx;
                if (nw < 0 || nw % q.stride) continue;
                const int ow = nw / q.stride;
                if (ow >= q.Wo) continue;
                acc += (float)dy[(((long)b * q.Ho + oh) * q.Wo + ow) * q.ldo + c] * w[c * 9 + ky * 3 + kx];
            }
        }
        dx[m * q.ldx + c] = (__bf16)acc;
    }
}

// partial[blk][t][c] (t = 0..8 taps, 9 = bias) over the block's 256 output pixels; fixed order
constexpr int DWG_ROWS = 32;       // (256 rows per block left the 6 x 11 x 19 maps of the local attention modules with 20 workgroups: 197 us per launch)
__global__ __launch_bounds__(256) void gdw_wgrad_partial_kernel(const __bf16* dy, const __bf16* x, float* partial, DwP q) {
    __shared__ float red[10][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cx;
    const long M = (long)q.B * q.Ho * q.Wo;
    float s[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) s[t] = 0.f;
    if (c < q.C)
        for (int r = ry; r < DWG_ROWS; r += 4) {
            const long m = (long)blockIdx.x * DWG_ROWS + r;
            if (m >= M) break;
            const int ow = (int)(m % q.Wo);
            const long tt = m / q.Wo;
            const int oh = (int)(tt % q.Ho), b = (int)(tt / q.Ho);
            const float g = (float)dy[m * q.ldo + c];
            s[9] += g;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ih = oh * q.stride - q.pad + ky;
                const bool okh = (unsigned)ih < (unsigned)q.H;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {                       // unconditional loads at a clamped position: no wait between them
                    const int iw = ow * q.stride - q.pad + kx;
                    const bool ok = okh && (unsigned)iw < (unsigned)q.W;
                    const float xv = (float)x[(((long)b * q.H + (okh ? ih : 0)) * q.W + ((unsigned)iw < (unsigned)q.W ? iw : 0)) * q.ldx + c];
                    s[ky * 3 + kx] += ok ? g * xv : 0.f;
                }
            }
        }
#pragma unroll
    for (int t = 0; t < 10; ++t) red[t][ry][cx] = s[t];
    __syncthreads();
    if (ry == 0 && c < q.C)
#pragma unroll
        for (int t = 0; t < 10; ++t)
            partial[((long)blockIdx.x * 10 + t) * q.C + c] = red[t][0][cx] + red[t][1][cx] + red[t][2][cx] + red[t][3][cx];
}
__global__ void gdw_wgrad_final_kernel(const float* partial, int blocks, int C, float* dw, float* dbias, int accumulate) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // (c, t)
    if (e >= C * 10) return;
    const int c = e / 10, t = e - c * 10;
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += (double)partial[((long)b * 10 + t) * C + c];
    float* d = t < 9 ? dw + c * 9 + t : (dbias ? dbias + c : nullptr);
    if (d) *d = accumulate ? *d + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------------------ criss-cross attention
// One workgroup (256 threads) per pixel (b, h, w).  Candidates j = 0 .. H-1: the column (b, j, w); j = H .. H+W-1: the row (b, h, j-H).
// The column's own position j == h is masked (-inf): the row holds it (ccnet.py:23-28,93-97).  att: fp32 [B][H][W][H+W].
struct CcaP {
    int B, H, W, Cq, C;
    long ldq, ldk, ldv, ldo;
};
constexpr int CCA_MAXJ = 512;          // H + W of the maps this path attends over (11 + 11 at 352 x 352; 97 + 97 would fit)

__device__ __forceinline__ long cca_cand(const CcaP& p, int b, int h, int w, int j) {
    return j < p.H ? ((long)b * p.H + j) * p.W + w : ((long)b * p.H + h) * p.W + (j - p.H);
}

__global__ __launch_bounds__(256) void gcca_fwd_kernel(const __bf16* qv, const __bf16* kv, const __bf16* vv, float* att, __bf16* agg, CcaP p) {
    __shared__ float e[CCA_MAXJ];
    __shared__ float redv[256];
    const int tid = threadIdx.x;
    const long pix = blockIdx.x;
    const int w = (int)(pix % p.W);
    const long tt = pix / p.W;
    const int h = (int)(tt % p.H), b = (int)(tt / p.H);
    const int J = p.H + p.W;
    const __bf16* qp = qv + pix * p.ldq;
    for (int j = tid; j < J; j += 256) {
        float s = -INFINITY;
        if (j != h) {
            const __bf16* kp = kv + cca_cand(p, b, h, w, j) * p.ldk;
            s = 0.f;
            for (int c = 0; c < p.Cq; ++c) s += (float)qp[c] * (float)kp[c];
        }
        e[j] = s;
    }
    __syncthreads();
    // softmax over the J candidates: max and sum by a fixed tree over the 256 lanes
    float mx = -INFINITY;
    for (int j = tid; j < J; j += 256) mx = fmaxf(mx, e[j]);
    redv[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) redv[tid] = fmaxf(redv[tid], redv[tid + s]);
        __syncthreads();
    }
    mx = redv[0];
    __syncthreads();
    float sm = 0.f;
    for (int j = tid; j < J; j += 256) {
        const float ex = __expf(e[j] - mx);
        e[j] = ex;
        sm += ex;
    }
    redv[tid] = sm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) redv[tid] += redv[tid + s];
        __syncthreads();
    }
    const float inv = 1.f / redv[0];
    for (int j = tid; j < J; j += 256) {
        const float a = e[j] * inv;
        e[j] = a;
        att[pix * J + j] = a;
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        float acc = 0.f;
        for (int j = 0; j < J; ++j) acc += e[j] * (float)vv[cca_cand(p, b, h, w, j) * p.ldv + c];
        agg[pix * p.ldo + c] = (__bf16)acc;
    }
}

// backward, part A (per pixel): datt[j] = dagg . v[cand j];  de = att * (datt - sum_j att datt);  dq = sum_j de[j] k[cand j];  de is kept
__global__ __launch_bounds__(256) void gcca_bwd_a_kernel(const __bf16* kv, const __bf16* vv, const float* att, const __bf16* dagg, long lddagg, float* de,
                                                         __bf16* dq, long lddq, CcaP p) {
    __shared__ float d[CCA_MAXJ];
    __shared__ float redv[256];
    const int tid = threadIdx.x;
    const long pix = blockIdx.x;
    const int w = (int)(pix % p.W);
    const long tt = pix / p.W;
    const int h = (int)(tt % p.H), b = (int)(tt / p.H);
    const int J = p.H + p.W;
    const __bf16* gp = dagg + pix * lddagg;
    // datt: four lanes per candidate, each a quarter of the channels in 8-channel pieces (16-byte loads where the views allow), the four partial sums
    // added in lane order.  (The first version gave a candidate to ONE thread - 62 of 256 busy, 2 x 256 dependent 2-byte loads each: 306 us per launch.)
    {
        const bool vec = (p.C & 7) == 0 && (p.ldv & 7) == 0 && (lddagg & 7) == 0 && ((reinterpret_cast<uintptr_t>(vv) | reinterpret_cast<uintptr_t>(dagg)) & 15) == 0;
        const int sub = tid & 3;
        const int cq = ((p.C + 3) / 4 + 7) & ~7;                    // channels per lane, a multiple of 8
        const int c_lo = sub * cq, c_hi = min(p.C, c_lo + cq);
        for (int j0 = 0; j0 < J; j0 += 64) {
            const int j = j0 + (tid >> 2);
            float s = 0.f;
            if (j < J) {
                const __bf16* vp = vv + cca_cand(p, b, h, w, j) * p.ldv;
                if (vec) {
                    for (int c = c_lo; c < c_hi; c += 8) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(gp + c), bb = *reinterpret_cast<const bf16x8*>(vp + c);
#pragma unroll
                        for (int k = 0; k < 8; ++k) s += (float)a[k] * (float)bb[k];
                    }
                } else {
                    for (int c = c_lo; c < c_hi; ++c) s += (float)gp[c] * (float)vp[c];
                }
            }
            const float s1 = __shfl_xor(s, 1, 64);
            const float pair = (sub & 1) ? s1 + s : s + s1;         // (lane 0 + lane 1), (lane 2 + lane 3): the same operand order on both lanes
            const float other = __shfl_xor(pair, 2, 64);
            if (sub == 0 && j < J) d[j] = pair + other;
        }
    }
    __syncthreads();
    float dot = 0.f;
    for (int j = tid; j < J; j += 256) dot += att[pix * J + j] * d[j];
    redv[tid] = dot;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) redv[tid] += redv[tid + s];
        __syncthreads();
    }
    dot = redv[0];
    __syncthreads();
    for (int j = tid; j < J; j += 256) {
        const float v = att[pix * J + j] * (d[j] - dot);
        d[j] = v;
        de[pix * J + j] = v;
    }
    __syncthreads();
    // dq: the candidates are dealt to 256 / Cq' thread groups (Cq' = Cq rounded up to a power of two <= 256), each thread adds its group's candidates in
    // ascending order with unconditional loads, the groups meet in LDS in group order (Cq = 32: 8 groups instead of 32 busy threads and 62 serial loads)
    {
        int cw = 1;
        while (cw < p.Cq && cw < 256) cw <<= 1;
        const int groups = 256 / cw, g = tid / cw, c = tid % cw;
        for (int c0 = 0; c0 < p.Cq; c0 += cw) {
            float acc = 0.f;
            const int cc = c0 + c;
            if (cc < p.Cq) {
                for (int j = g; j < J; j += groups) {
                    const float kvv = (float)kv[cca_cand(p, b, h, w, j) * p.ldk + cc];
                    acc += j != h ? d[j] * kvv : 0.f;
                }
            }
            __syncthreads();
            redv[tid] = acc;
            __syncthreads();
            if (g == 0 && cc < p.Cq) {
                float tot = 0.f;
                for (int gg = 0; gg < groups; ++gg) tot += redv[gg * cw + c];
                dq[pix * lddq + cc] = (__bf16)tot;
            }
        }
    }
}
// part B (per pixel, gather form): this pixel (b, h', w') is candidate h' of every pixel of its column and candidate H + w' of every pixel of
// its row:  dk = sum_col de_p[h'] q_p + sum_row de_p[H + w'] q_p;  dv = sum_col att_p[h'] dagg_p + sum_row att_p[H + w'] dagg_p
__global__ __launch_bounds__(256) void gcca_bwd_b_kernel(const __bf16* qv, const float* att, const float* de, const __bf16* dagg, long lddagg, __bf16* dk, long lddk,
                                                         __bf16* dv, long lddv, CcaP p) {
    const int tid = threadIdx.x;
    const long pix = blockIdx.x;
    const int w = (int)(pix % p.W);
    const long tt = pix / p.W;
    const int h = (int)(tt % p.H), b = (int)(tt / p.H);
    const int J = p.H + p.W;
    for (int c = tid; c < p.C + p.Cq; c += 256) {
        const bool isv = c < p.C;
        float acc = 0.f;
        for (int hh = 0; hh < p.H; ++hh) {                     // pixels of my column: (b, hh, w); I am their candidate j = h (masked when hh == h)
            if (hh == h) continue;
            const long pp = ((long)b * p.H + hh) * p.W + w;
            acc += isv ? att[pp * J + h] * (float)dagg[pp * lddagg + c] : de[pp * J + h] * (float)qv[pp * p.ldq + (c - p.C)];
        }
        for (int ww = 0; ww < p.W; ++ww) {                     // pixels of my row: (b, h, ww); I am their candidate j = H + w
            const long pp = ((long)b * p.H + h) * p.W + ww;
            acc += isv ? att[pp * J + p.H + w] * (float)dagg[pp * lddagg + c] : de[pp * J + p.H + w] * (float)qv[pp * p.ldq + (c - p.C)];
        }
        if (isv) dv[pix * lddv + c] = (__bf16)acc;
        else dk[pix * lddk + (c - p.C)] = (__bf16)acc;
    }
}

// ------------------------------------------------------------------------------------------------ sigmoid gate of the local attention module
// forward: out = x + x * sigmoid(g);  backward: dx = dout * (1 + s), dg = dout * x * s * (1 - s)
__global__ __launch_bounds__(256) void ggate_kernel(const __bf16* x, long ldx, const __bf16* g, long ldg, const __bf16* dout, long lddo, __bf16* o1, long ld1, __bf16* o2,
                                                    long ld2, long M, int C) {
    const long n = M * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / C;
        const int c = (int)(e - m * C);
        const float xv = (float)x[m * ldx + c];
        const float s = 1.f / (1.f + __expf(-(float)g[m * ldg + c]));
        if (!dout) {
            o1[m * ld1 + c] = (__bf16)(xv + xv * s);
        } else {
            const float d = (float)dout[m * lddo + c];
            o1[m * ld1 + c] = (__bf16)(d * (1.f + s));
            o2[m * ld2 + c] = (__bf16)(d * xv * s * (1.f - s));
        }
    }
}

inline int grid_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" {

size_t mi_gdwconv_stats_elems(int B, int Ho, int Wo, int C) { return (size_t)(((long)B * Ho * Wo + DW_TILE - 1) / DW_TILE) * 2 * C; }

int mi_gdwconv(const void* x, long ldx, const float* w, const float* bias, void* out, long ldo, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad,
               float* stats, void* stream) {
    MI_REQUIRE(x && w && out, "mi_gdwconv: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && stride > 0 && pad >= 0 && ldx >= C && ldo >= C, "mi_gdwconv: bad shape");
    MI_REQUIRE((H + 2 * pad - 3) / stride + 1 == Ho && (W + 2 * pad - 3) / stride + 1 == Wo && Ho > 0 && Wo > 0, "mi_gdwconv: output %dx%d does not follow from input %dx%d", Ho, Wo, H, W);
    DwP q{B, H, W, C, Ho, Wo, stride, pad, ldx, ldo};
    const long M = (long)B * Ho * Wo;
    hipLaunchKernelGGL(gdw_fwd_kernel, dim3((unsigned)((M + DW_TILE - 1) / DW_TILE), (unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, w, bias,
                       (__bf16*)out, stats, q);
    MI_CHECK_LAUNCH("gdw_fwd_kernel");
    return MI_OK;
}

int mi_gdwconv_dgrad(const void* dy, long ldy, const float* w, void* dx, long lddx, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad, void* stream) {
    MI_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && lddx >= C, "mi_gdwconv_dgrad: bad operand");
    MI_REQUIRE((H + 2 * pad - 3) / stride + 1 == Ho && (W + 2 * pad - 3) / stride + 1 == Wo, "mi_gdwconv_dgrad: geometry");
    DwP q{B, H, W, C, Ho, Wo, stride, pad, lddx, ldy};
    hipLaunchKernelGGL(gdw_dgrad_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, w, (__bf16*)dx, q);
    MI_CHECK_LAUNCH("gdw_dgrad_kernel");
    return MI_OK;
}

size_t mi_gdwconv_wgrad_workspace(int B, int Ho, int Wo, int C) { return (size_t)(((long)B * Ho * Wo + DWG_ROWS - 1) / DWG_ROWS) * 10 * C * sizeof(float); }

int mi_gdwconv_wgrad(const void* dy, long ldy, const void* x, long ldx, float* dw, float* dbias, int B, int H, int W, int C, int Ho, int Wo, int stride, int pad,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(dy && x && dw && workspace && B > 0 && H > 0 && W > 0 && C > 0 && ldy >= C && ldx >= C, "mi_gdwconv_wgrad: bad operand");
    MI_REQUIRE((H + 2 * pad - 3) / stride + 1 == Ho && (W + 2 * pad - 3) / stride + 1 == Wo, "mi_gdwconv_wgrad: geometry");
    if (workspace_bytes < mi_gdwconv_wgrad_workspace(B, Ho, Wo, C)) return mi_set_error(MI_ENOMEM, "mi_gdwconv_wgrad: workspace too small");
    DwP q{B, H, W, C, Ho, Wo, stride, pad, ldx, ldy};
    const long M = (long)B * Ho * Wo;
    const int blocks = (int)((M + DWG_ROWS - 1) / DWG_ROWS);
    hipLaunchKernelGGL(gdw_wgrad_partial_kernel, dim3(blocks, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, (const __bf16*)x, (float*)workspace, q);
    MI_CHECK_LAUNCH("gdw_wgrad_partial_kernel");
    hipLaunchKernelGGL(gdw_wgrad_final_kernel, dim3((C * 10 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, blocks, C, dw, dbias, accumulate);
    MI_CHECK_LAUNCH("gdw_wgrad_final_kernel");
    return MI_OK;
}

int mi_gcca_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, float* att, void* agg, long ldo, int B, int H, int W, int Cq, int C,
                void* stream) {
    MI_REQUIRE(q && k && v && att && agg, "mi_gcca_fwd: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cq > 0 && C > 0 && H + W <= CCA_MAXJ && ldq >= Cq && ldk >= Cq && ldv >= C && ldo >= C, "mi_gcca_fwd: bad shape (H + W <= %d)", CCA_MAXJ);
    CcaP p{B, H, W, Cq, C, ldq, ldk, ldv, ldo};
    hipLaunchKernelGGL(gcca_fwd_kernel, dim3((unsigned)((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)q, (const __bf16*)k, (const __bf16*)v, att,
                       (__bf16*)agg, p);
    MI_CHECK_LAUNCH("gcca_fwd_kernel");
    return MI_OK;
}

int mi_gcca_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const float* att, const void* dagg, long lddagg, float* de_ws, void* dq,
                long lddq, void* dk, long lddk, void* dv, long lddv, int B, int H, int W, int Cq, int C, void* stream) {
    MI_REQUIRE(q && k && v && att && dagg && de_ws && dq && dk && dv, "mi_gcca_bwd: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cq > 0 && C > 0 && H + W <= CCA_MAXJ, "mi_gcca_bwd: bad shape");
    MI_REQUIRE(ldq >= Cq && ldk >= Cq && ldv >= C && lddagg >= C && lddq >= Cq && lddk >= Cq && lddv >= C, "mi_gcca_bwd: a view's row stride is smaller than its channel count");
    CcaP p{B, H, W, Cq, C, ldq, ldk, ldv, 0};
    const dim3 grid((unsigned)((long)B * H * W));
    hipLaunchKernelGGL(gcca_bwd_a_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)k, (const __bf16*)v, att, (const __bf16*)dagg, lddagg, de_ws, (__bf16*)dq,
                       lddq, p);
    MI_CHECK_LAUNCH("gcca_bwd_a_kernel");
    hipLaunchKernelGGL(gcca_bwd_b_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)q, att, (const float*)de_ws, (const __bf16*)dagg, lddagg, (__bf16*)dk, lddk,
                       (__bf16*)dv, lddv, p);
    MI_CHECK_LAUNCH("gcca_bwd_b_kernel");
    return MI_OK;
}

int mi_ggate(const void* x, long ldx, const void* g, long ldg, const void* dout, long lddo, void* o1, long ld1, void* o2, long ld2, long M, int C, void* stream) {
    MI_REQUIRE(x && g && o1 && (!dout || o2), "mi_ggate: null operand");
    MI_REQUIRE(M > 0 && C > 0 && ldx >= C && ldg >= C && ld1 >= C && (!dout || (lddo >= C && ld2 >= C)), "mi_ggate: bad shape");
    hipLaunchKernelGGL(ggate_kernel, dim3(grid_for(M * C)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, ldx, (const __bf16*)g, ldg, (const __bf16*)dout, lddo,
                       (__bf16*)o1, ld1, (__bf16*)o2, ld2, M, C);
    MI_CHECK_LAUNCH("ggate_kernel");
    return MI_OK;
}

}  // extern "C"
