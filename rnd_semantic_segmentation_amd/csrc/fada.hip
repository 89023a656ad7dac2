// Kernels of the FADA adversarial step: conv bias gradient, fused bilinear upsample + soft-label
// cross-entropy with on-the-fly soft labels, Adam.  References: core/combos/aspp_fada.py:80-127,
// core/utils/utility.py:172-177 (soft_label_cross_entropy), core/adapters/fada_adapter.py:24 (Adam betas .9/.99).
// fp32 arithmetic, fixed summation orders (bitwise reproducible), HBM-bound.
#include "mi_common.h"

namespace {

constexpr int KMAX = 32;       // segmentation classes held in registers
constexpr int JT = 32;         // low-res columns per workgroup (same tiling as upce_pass1)

struct Axis {                  // align_corners source index as ATen computes it in fp32 (same as upsample_ce.hip)
    float scale;
    int n_in, n_out;
    __device__ __forceinline__ void src(int dst, int& i0, int& i1, float& lam) const {
        const float f = scale * (float)dst;
        i0 = (int)f;
        if (i0 > n_in - 1) i0 = n_in - 1;
        i1 = (i0 < n_in - 1) ? i0 + 1 : i0;
        lam = f - (float)i0;
    }
    __device__ __forceinline__ int first_with_i0_ge(int c) const {
        if (c <= 0) return 0;
        if (scale <= 0.f) return n_out;
        if (c > n_in - 1) return n_out;
        int d = (int)((float)c / scale) - 2;
        if (d < 0) d = 0;
        if (d > n_out) d = n_out;
        while (d < n_out) {
            int i0 = (int)(scale * (float)d);
            if (i0 > n_in - 1) i0 = n_in - 1;
            if (i0 >= c) break;
            ++d;
        }
        return d;
    }
};

inline Axis make_axis(int n_in, int n_out) {
    Axis a;
    a.n_in = n_in;
    a.n_out = n_out;
    a.scale = (n_out > 1) ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    return a;
}

__device__ __forceinline__ void block_sum(float& a, float* red) {
    const int t = threadIdx.x;
    red[t] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    a = red[0];
    __syncthreads();
}

// pass 1: workgroup = (b, y, tile of JT low-res columns).  Per high-res pixel (once): soft labels from the segmentation logits,
// log-softmax of the 2K discriminator logits, loss term, d = S*softmax(z) - placed(soft) into LDS; then gather along x.
// KT > 0: compile-time class count (19: exact-length unrolled loops, soft labels indexed statically); KT == 0: runtime K.
template <int KT>
__global__ __launch_bounds__(256) void softce_pass1_kernel(const float* __restrict__ seg, float inv_t, float clip, const float* __restrict__ dl,
                                                           int ldD, int domain, float* __restrict__ partial, float* __restrict__ tmp, int B,
                                                           int Krt, Axis ay, Axis ax, int npx_max) {
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int K = KT > 0 ? KT : Krt;
    constexpr int KR = KT > 0 ? KT : KMAX;
    const int K2 = 2 * K;
    float* dbuf = sh;                                   // [npx_max][K2]
    float* lam = dbuf + (long)npx_max * K2;             // [npx_max]
    int* x0s = reinterpret_cast<int*>(lam + npx_max);   // [npx_max]
    float* red = reinterpret_cast<float*>(x0s + npx_max);   // [256]
    int* pstart = reinterpret_cast<int*>(red + 256);    // [JT+3] first pixel (relative to xa) whose x0 >= j0 - 1 + q
    float* srow = red + 256 + JT + 4;                   // [JT+2][K]   seg logits interpolated along y
    float* drow = srow + (JT + 2) * K;                  // [JT+2][K2]  discriminator logits interpolated along y
    const int H = ay.n_out, h = ay.n_in, w = ax.n_in;
    const int jt = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int j0 = jt * JT, j1 = min(w, j0 + JT);
    const int xa = ax.first_with_i0_ge(j0 - 1), xb = ax.first_with_i0_ge(j1);
    const int npx = xb - xa;
    int y0, y1;
    float ly;
    ay.src(y, y0, y1, ly);
    const int cbase = max(j0 - 1, 0), ncol = min(j1, w - 1) - cbase + 1;
    if (threadIdx.x < j1 - j0 + 2) pstart[threadIdx.x] = ax.first_with_i0_ge(j0 - 1 + (int)threadIdx.x) - xa;
    for (int e = threadIdx.x; e < ncol * K; e += 256) {
        const int c = e / K, k = e - c * K;
        const long o0 = (((long)b * h + y0) * w + cbase + c) * K + k, o1 = (((long)b * h + y1) * w + cbase + c) * K + k;
        srow[e] = (1.f - ly) * seg[o0] + ly * seg[o1];
    }
    for (int e = threadIdx.x; e < ncol * K2; e += 256) {
        const int c = e / K2, k = e - c * K2;
        const long o0 = (((long)b * h + y0) * w + cbase + c) * ldD + k, o1 = (((long)b * h + y1) * w + cbase + c) * ldD + k;
        drow[e] = (1.f - ly) * dl[o0] + ly * dl[o1];
    }
    __syncthreads();
    float loss = 0.f;
    for (int px = threadIdx.x; px < npx; px += 256) {
        const int x = xa + px;
        int x0, x1;
        float lx;
        ax.src(x, x0, x1, lx);
        lam[px] = lx;
        const float* s0 = srow + (x0 - cbase) * K;
        const float* s1 = srow + (x1 - cbase) * K;
        const float* z0 = drow + (x0 - cbase) * K2;
        const float* z1 = drow + (x1 - cbase) * K2;
        // soft labels: softmax(seg / T), clipped (aspp_fada.py:91-103)
        float soft[KR];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < KR; ++k)
            if (k < K) {
                soft[k] = ((1.f - lx) * s0[k] + lx * s1[k]) * inv_t;
                mx = fmaxf(mx, soft[k]);
            }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < KR; ++k)
            if (k < K) {
                soft[k] = __expf(soft[k] - mx);
                se += soft[k];
            }
        float S = 0.f;
        const float rse = 1.f / se;
#pragma unroll
        for (int k = 0; k < KR; ++k)
            if (k < K) {
                soft[k] = fminf(soft[k] * rse, clip);
                S += soft[k];
            }
        // discriminator log-softmax over 2K channels (two sweeps over LDS instead of 2K registers)
        float zm = -3.0e38f;
        for (int k = 0; k < K2; ++k) zm = fmaxf(zm, (1.f - lx) * z0[k] + lx * z1[k]);
        float ze = 0.f;
        for (int k = 0; k < K2; ++k) ze += __expf((1.f - lx) * z0[k] + lx * z1[k] - zm);
        const float lse = zm + __logf(ze), rze = 1.f / ze;
        float* d = dbuf + (long)px * K2;
        float l = 0.f;
        if (KT > 0) {                     // channel k = half*K + q carries soft[q] when half == domain, else 0: static indices
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int q = 0; q < KR; ++q) {
                    const int k = half * KR + q;
                    const float z = (1.f - lx) * z0[k] + lx * z1[k];
                    const float placed = (half == domain) ? soft[q] : 0.f;
                    l -= placed * (z - lse);
                    d[k] = S * __expf(z - zm) * rze - placed;
                }
        } else {
            for (int k = 0; k < K2; ++k) {
                const float z = (1.f - lx) * z0[k] + lx * z1[k];
                float placed = 0.f;
                const int c = k - domain * K;
#pragma unroll
                for (int q = 0; q < KMAX; ++q)
                    if (q == c) placed = soft[q];
                l -= placed * (z - lse);
                d[k] = S * __expf(z - zm) * rze - placed;
            }
        }
        if (x0 >= j0) loss += l;          // the tile that owns x0 accounts for the pixel's loss
    }
    __syncthreads();
    if (tmp) {
        const int nj = j1 - j0;
        for (int item = threadIdx.x; item < nj * K2; item += 256) {
            const int jj = item / K2, k = item - jj * K2;
            const int j = j0 + jj;
            float s = 0.f;
            // pixels with x0 == j-1 contribute lam (as x1), then pixels with x0 == j contribute 1-lam (+ lam at the clamped right edge)
            const int p0 = max(pstart[jj], 0), p1 = min(max(pstart[jj + 1], 0), npx), p2 = min(pstart[jj + 2], npx);
            for (int px = p0; px < p1; ++px) s += (0.f + lam[px]) * dbuf[(long)px * K2 + k];
            const bool edge = j == w - 1;
            for (int px = p1; px < p2; ++px) s += ((1.f - lam[px]) + (edge ? lam[px] : 0.f)) * dbuf[(long)px * K2 + k];
            tmp[(((long)b * H + y) * w + j) * K2 + k] = s;
        }
    }
    block_sum(loss, red);
    if (threadIdx.x == 0) partial[((long)b * H + y) * gridDim.x + jt] = loss;
}

__global__ void softce_finalize_kernel(const float* __restrict__ partial, int n, float count, float* __restrict__ loss_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    block_sum(s, red);
    if (threadIdx.x == 0) {
        loss_out[0] = s / count;
        loss_out[1] = count;
    }
}

// pass 2: dd_low[b][i][j][k] = grad_scale / count * sum_y wy(y,i) tmp[b][y][j][k]
__global__ void softce_pass2_kernel(const float* __restrict__ tmp, float* __restrict__ dd, int B, int K2, int ldD, Axis ay, int w,
                                    float scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H = ay.n_out, h = ay.n_in;
    const long per_row = (long)w * K2;
    if (idx >= (long)B * h * per_row) return;
    const long jk = idx % per_row;
    const int i = (int)((idx / per_row) % h), b = (int)(idx / (per_row * h));
    const int ya = ay.first_with_i0_ge(i - 1), yb = ay.first_with_i0_ge(i + 1);
    float s = 0.f;
    for (int y = ya; y < yb; ++y) {
        int y0, y1;
        float ly;
        ay.src(y, y0, y1, ly);
        const float wy = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
        s += wy * tmp[((long)b * H + y) * per_row + jk];
    }
    const int j = (int)(jk / K2), k = (int)(jk - (long)j * K2);
    dd[(((long)b * h + i) * w + j) * ldD + k] = s * scale;
}

// torch.optim.Adam single-tensor update; CLAMP: the gradient is first clamped to [-clamp, clamp] IN PLACE (core/utils/utils.py:6-16
// clip_gradient, called before optimizer.step() at pranet_trainer.py:59)
template <bool CLAMP>
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float step_size, float beta1, float beta2, float inv_sqrt_bc2, float eps, float clamp) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i];
        if (CLAMP) {
            gi = fminf(fmaxf(gi, -clamp), clamp);
            g[i] = gi;
        }
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

// The same update with every hyper-parameter read from DEVICE memory (hyper = {lr, beta1, beta2, eps, grad_clamp (<= 0: none), step}): a
// captured HIP graph freezes by-value kernel arguments, and Adam's bias corrections change with the step count on every replay.
__global__ void adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                const float* __restrict__ hyper) {
    const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], clamp = hyper[4], step = hyper[5];
    const float step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i];
        if (clamp > 0.f) {
            gi = fminf(fmaxf(gi, -clamp), clamp);
            g[i] = gi;
        }
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

inline int npx_bound(const Axis& ax) {
    if (ax.scale <= 0.f) return ax.n_out;
    const long n = (long)((float)(JT + 1) / ax.scale) + 4;
    return (int)(n < ax.n_out ? n : ax.n_out);
}

}  // namespace

extern "C" size_t mi_upsample_softce_workspace(int B, int h, int w, int K, int H, int W) {
    const size_t tiles = (size_t)((w + JT - 1) / JT);
    const size_t partial = (((size_t)B * H * tiles * sizeof(float)) + 255) & ~(size_t)255;
    return partial + (size_t)B * H * w * 2 * K * sizeof(float);
}

extern "C" int mi_upsample_softce(const float* seg_low, float inv_temperature, float clip, const float* d_low, int ldD, int domain,
                                  float* loss_out, float* dd_low, int B, int h, int w, int K, int H, int W, float grad_scale,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(seg_low && d_low && loss_out && workspace, "mi_upsample_softce: null operand");
    MI_REQUIRE(B > 0 && h > 0 && w > 0 && H >= h && W >= w && K > 0 && K <= KMAX && ldD >= 2 * K, "mi_upsample_softce: bad dimension");
    MI_REQUIRE(domain == 0 || domain == 1, "mi_upsample_softce: domain is 0 (source half) or 1 (target half)");
    MI_REQUIRE(H <= 65535 && B <= 65535, "mi_upsample_softce: grid dimension overflow");
    if (workspace_bytes < mi_upsample_softce_workspace(B, h, w, K, H, W)) return mi_set_error(MI_ENOMEM, "mi_upsample_softce: workspace too small");
    const Axis ay = make_axis(h, H), ax = make_axis(w, W);
    const int tiles = (w + JT - 1) / JT, K2 = 2 * K;
    float* partial = (float*)workspace;
    const size_t poff = (((size_t)B * H * tiles * sizeof(float)) + 255) & ~(size_t)255;
    float* tmp = dd_low ? (float*)((char*)workspace + poff) : nullptr;
    const int npx_max = npx_bound(ax);
    const size_t lds = ((size_t)npx_max * K2 + (size_t)npx_max * 2 + 256 + (JT + 4) + (size_t)(JT + 2) * (K + K2)) * 4;
    MI_REQUIRE(lds <= 160 * 1024, "mi_upsample_softce: upsample factor too large for one LDS tile (%zu B)", lds);
    static std::atomic<uint64_t> lds_set[2];
    mi_allow_dynamic_lds((const void*)softce_pass1_kernel<19>, MI_LDS_MAX, lds_set[0]);
    mi_allow_dynamic_lds((const void*)softce_pass1_kernel<0>, MI_LDS_MAX, lds_set[1]);
    if (K == 19)
        hipLaunchKernelGGL(softce_pass1_kernel<19>, dim3(tiles, H, B), dim3(256), lds, (hipStream_t)stream, seg_low, inv_temperature, clip, d_low,
                           ldD, domain, partial, tmp, B, K, ay, ax, npx_max);
    else
        hipLaunchKernelGGL(softce_pass1_kernel<0>, dim3(tiles, H, B), dim3(256), lds, (hipStream_t)stream, seg_low, inv_temperature, clip, d_low,
                           ldD, domain, partial, tmp, B, K, ay, ax, npx_max);
    MI_CHECK_LAUNCH("mi_upsample_softce pass1");
    const float count = (float)B * (float)H * (float)W;
    hipLaunchKernelGGL(softce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, B * H * tiles, count, loss_out);
    MI_CHECK_LAUNCH("mi_upsample_softce finalize");
    if (dd_low) {
        if (ldD > K2) {
            hipError_t e = hipMemsetAsync(dd_low, 0, (size_t)B * h * w * ldD * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) return mi_set_error(MI_EHIP, "mi_upsample_softce: memset: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(softce_pass2_kernel, dim3(nblk((long)B * h * w * K2, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, dd_low,
                           B, K2, ldD, ay, w, grad_scale / count);
        MI_CHECK_LAUNCH("mi_upsample_softce pass2");
    }
    return MI_OK;
}

extern "C" int mi_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                            float eps, int step, void* stream) {
    MI_REQUIRE(p && g && exp_avg && exp_avg_sq && n > 0 && step >= 1, "mi_adam_step: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, const_cast<float*>(g), exp_avg, exp_avg_sq, n,
                       (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, 0.f);
    MI_CHECK_LAUNCH("mi_adam_step");
    return MI_OK;
}

extern "C" int mi_adam_step_dev(float* p, float* g, float* exp_avg, float* exp_avg_sq, size_t n, const float* hyper, void* stream) {
    MI_REQUIRE(p && g && exp_avg && exp_avg_sq && hyper && n > 0, "mi_adam_step_dev: bad argument");
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, n, hyper);
    MI_CHECK_LAUNCH("mi_adam_step_dev");
    return MI_OK;
}

extern "C" int mi_adam_step_clamped(float* p, float* g, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                                    float eps, int step, float grad_clamp, void* stream) {
    MI_REQUIRE(p && g && exp_avg && exp_avg_sq && n > 0 && step >= 1 && grad_clamp > 0.f, "mi_adam_step_clamped: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, n, (float)(lr / bc1),
                       beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, grad_clamp);
    MI_CHECK_LAUNCH("mi_adam_step_clamped");
    return MI_OK;
}
