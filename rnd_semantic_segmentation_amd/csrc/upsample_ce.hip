// Bilinear upsample (align_corners=True), per-pixel softmax cross-entropy with ignore_index, and their fusion.
//   F.interpolate(..., mode='bilinear', align_corners=True)   reference core/models/classifiers/aspp/classifier.py:31,
//                                                              core/utils/utility.py:185
//   torch.nn.CrossEntropyLoss(ignore_index=255)               reference core/trainers/aspp_trainer.py:61,91
//   softmax over classes for inference                        reference core/utils/utility.py:186
// All arithmetic fp32 (kept fp32 in bf16 mode too).  HBM-bound: the fused training path reads the
// 1/8-resolution logits (5.7 MB at B=8, 769x769) and labels and never writes the 360 MB [B,19,769,769] tensor.
// Every reduction has a fixed summation order (no float atomics) so results are bitwise reproducible.
#include "mi_common.h"

namespace {

constexpr int KMAX = 32;   // classes held in registers

struct Axis {              // source index exactly as ATen computes it in fp32: align_corners (off = 0): scale * dst; otherwise (off = 0.5):
    float scale, off;      // max(scale * (dst + 0.5) - 0.5, 0)  (adding / subtracting 0.0f is exact: the align_corners bits are unchanged)
    int n_in, n_out;
    __device__ __forceinline__ float srcf(int dst) const {
        const float f = scale * ((float)dst + off) - off;
        return f < 0.f ? 0.f : f;
    }
    __device__ __forceinline__ void src(int dst, int& i0, int& i1, float& lam) const {
        const float f = srcf(dst);
        i0 = (int)f;
        if (i0 > n_in - 1) i0 = n_in - 1;
        i1 = (i0 < n_in - 1) ? i0 + 1 : i0;
        lam = f - (float)i0;
    }
    // first dst index whose i0 >= c  (n_out if none)
    __device__ __forceinline__ int first_with_i0_ge(int c) const {
        if (c <= 0) return 0;
        if (scale <= 0.f) return n_out;
        if (c > n_in - 1) return n_out;
        int d = (int)((float)c / scale) - 2;
        if (d < 0) d = 0;
        if (d > n_out) d = n_out;
        while (d < n_out) {
            int i0 = (int)srcf(d);
            if (i0 > n_in - 1) i0 = n_in - 1;
            if (i0 >= c) break;
            ++d;
        }
        return d;
    }
};

inline Axis make_axis(int n_in, int n_out, int align_corners = 1) {
    Axis a;
    a.n_in = n_in;
    a.n_out = n_out;
    a.off = align_corners ? 0.f : 0.5f;
    a.scale = align_corners ? ((n_out > 1) ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f) : (float)n_in / (float)n_out;      // (a size was given: in / out)
    return a;
}

__device__ __forceinline__ float lerp2(float v00, float v01, float v10, float v11, float lx, float ly) {
    return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// ------------------------------------------------------------------------------------------------ unfused
__global__ void upsample_fwd_kernel(const float* __restrict__ low, float* __restrict__ up, int B, int K, Axis ay, Axis ax) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H = ay.n_out, W = ax.n_out, h = ay.n_in, w = ax.n_in;
    if (idx >= (long)B * H * W) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((long)W * H));
    int y0, y1, x0, x1;
    float ly, lx;
    ay.src(y, y0, y1, ly);
    ax.src(x, x0, x1, lx);
    const float* p00 = low + (((long)b * h + y0) * w + x0) * K;
    const float* p01 = low + (((long)b * h + y0) * w + x1) * K;
    const float* p10 = low + (((long)b * h + y1) * w + x0) * K;
    const float* p11 = low + (((long)b * h + y1) * w + x1) * K;
    float* o = up + ((long)b * K * H + y) * W + x;
    for (int k = 0; k < K; ++k) o[(long)k * H * W] = lerp2(p00[k], p01[k], p10[k], p11[k], lx, ly);
}

// gather form: thread per (b,k,i,j), j fastest; sums contributions in (y,x) ascending order
__global__ void upsample_bwd_kernel(const float* __restrict__ dup, float* __restrict__ dlow, int B, int K, Axis ay, Axis ax) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H = ay.n_out, W = ax.n_out, h = ay.n_in, w = ax.n_in;
    if (idx >= (long)B * K * h * w) return;
    const int j = (int)(idx % w), i = (int)((idx / w) % h), k = (int)((idx / ((long)w * h)) % K), b = (int)(idx / ((long)w * h * K));
    const int ya = ay.first_with_i0_ge(i - 1), yb = ay.first_with_i0_ge(i + 1);
    const int xa = ax.first_with_i0_ge(j - 1), xb = ax.first_with_i0_ge(j + 1);
    const float* src = dup + ((long)b * K + k) * H * W;
    float s = 0.f;
    for (int y = ya; y < yb; ++y) {
        int y0, y1;
        float ly;
        ay.src(y, y0, y1, ly);
        const float wy = (y0 == i ? 1.f - ly : 0.f) + (y1 == i ? ly : 0.f);
        if (wy == 0.f && y0 != i && y1 != i) continue;
        float r = 0.f;
        for (int x = xa; x < xb; ++x) {
            int x0, x1;
            float lx;
            ax.src(x, x0, x1, lx);
            const float wx = (x0 == j ? 1.f - lx : 0.f) + (x1 == j ? lx : 0.f);
            r += wx * src[(long)y * W + x];
        }
        s += wy * r;
    }
    dlow[(((long)b * h + i) * w + j) * K + k] = s;
}

__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    // fixed-order tree over 256 threads
    const int t = threadIdx.x;
    red[t] = a;
    red[256 + t] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
            red[t] += red[t + w];
            red[256 + t] += red[256 + t + w];
        }
        __syncthreads();
    }
    a = red[0];
    b = red[256];
    __syncthreads();
}

// Labels outside [0, K) that are not ignore_index: torch.nn.CrossEntropyLoss raises a device assert; here they are excluded
// from the loss AND counted (integer atomic: order-independent) so that the host can refuse the batch (loss_out[2]).
__device__ __forceinline__ void count_bad_label(long lab, int K, int ignore_index, unsigned* bad) {
    if (lab != ignore_index && (lab < 0 || lab >= K)) atomicAdd(bad, 1u);
}

__global__ void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ partial,
                              int B, int K, long HW, int ignore_index, unsigned* __restrict__ bad) {
    __shared__ float red[512];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f, cnt = 0.f;
    if (idx < (long)B * HW) {
        const long b = idx / HW, pix = idx - b * HW;
        const long lab = labels[idx];
        count_bad_label(lab, K, ignore_index, bad);
        if (lab != ignore_index && lab >= 0 && lab < K) {
            const float* p = logits + b * K * HW + pix;
            float mx = p[0];
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, p[k * HW]);
            float se = 0.f;
            for (int k = 0; k < K; ++k) se += __expf(p[k * HW] - mx);
            loss = (mx + __logf(se)) - p[lab * HW];
            cnt = 1.f;
        }
    }
    block_sum2(loss, cnt, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = loss;
        partial[2 * blockIdx.x + 1] = cnt;
    }
}

__global__ void ce_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ loss_out) {
    __shared__ float red[512];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        s += partial[2 * i];
        c += partial[2 * i + 1];
    }
    block_sum2(s, c, red);
    if (threadIdx.x == 0) {
        loss_out[0] = s / c;   // 0/0 = nan when every pixel is ignored, like torch
        loss_out[1] = c;
        loss_out[2] = (float)*reinterpret_cast<const unsigned*>(loss_out + 3);      // out-of-range labels seen by pass 1
    }
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ loss_out,
                              float* __restrict__ dlogits, int B, int K, long HW, int ignore_index, float grad_scale) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * HW) return;
    const long b = idx / HW, pix = idx - b * HW;
    const long lab = labels[idx];
    const float* p = logits + b * K * HW + pix;
    float* d = dlogits + b * K * HW + pix;
    if (lab == ignore_index || lab < 0 || lab >= K) {
        for (int k = 0; k < K; ++k) d[k * HW] = 0.f;
        return;
    }
    const float inv = grad_scale / loss_out[1];
    float mx = p[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, p[k * HW]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(p[k * HW] - mx);
    const float rse = 1.f / se;
    for (int k = 0; k < K; ++k) d[k * HW] = (__expf(p[k * HW] - mx) * rse - (k == lab ? 1.f : 0.f)) * inv;
}

// ------------------------------------------------------------------------------------------------ fused
// pass 1: one workgroup per (b, y, tile of JT low-res columns).  Threads compute, ONCE per high-res pixel, the
// interpolated logits, the loss term and d = softmax - onehot into LDS; then (j,k) items gather the pixels of
// their column support in ascending x:  tmp[b][y][j][k] = sum_x wx(x,j) d[x][k].
constexpr int JT = 32;              // the largest tile; the launcher narrows it for large upsample factors (pick_jt)

// KT > 0: the class count is a compile-time constant (19 for Cityscapes: exact-length register loops instead of 32 predicated
// iterations); KT == 0: K is read from the arguments.
template <int KT>
__global__ __launch_bounds__(256) void upce_pass1_kernel(const float* __restrict__ low, const int64_t* __restrict__ labels,
                                                         float* __restrict__ partial, float* __restrict__ tmp, int B, int Krt, Axis ay,
                                                         Axis ax, int ignore_index, int npx_max, unsigned* __restrict__ bad, int jt_cols) {
    const int K = KT > 0 ? KT : Krt;
    constexpr int KR = KT > 0 ? KT : KMAX;          // register array length
    extern __shared__ __attribute__((aligned(16))) float sh[];
    float* dbuf = sh;                               // [npx_max][K]
    float* lam = sh + (long)npx_max * K;            // [npx_max]  lambda_x
    int* x0s = reinterpret_cast<int*>(lam + npx_max);  // [npx_max]  x0
    float* red = reinterpret_cast<float*>(x0s + npx_max);  // [512]
    int* pstart = reinterpret_cast<int*>(red + 512);   // [JT+3] first pixel (relative to xa) whose x0 >= j0 - 1 + q
    float* vrow = red + 512 + JT + 4;                // [JT+2][K] low-res row already interpolated along y
    const int H = ay.n_out, W = ax.n_out, h = ay.n_in, w = ax.n_in;
    const int jt = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int j0 = jt * jt_cols, j1 = min(w, j0 + jt_cols);
    const int xa = ax.first_with_i0_ge(j0 - 1), xb = ax.first_with_i0_ge(j1);   // pixels with x0 in [j0-1, j1-1]
    const int npx = xb - xa;
    int y0, y1;
    float ly;
    ay.src(y, y0, y1, ly);
    const float* row0 = low + ((long)b * h + y0) * w * K;
    const float* row1 = low + ((long)b * h + y1) * w * K;
    const int cbase = max(j0 - 1, 0), ncol = min(j1, w - 1) - cbase + 1;          // source columns this tile touches
    if (threadIdx.x < j1 - j0 + 2) pstart[threadIdx.x] = ax.first_with_i0_ge(j0 - 1 + (int)threadIdx.x) - xa;
    for (int e = threadIdx.x; e < ncol * K; e += 256) {
        const long o = (long)cbase * K + e;
        vrow[e] = (1.f - ly) * row0[o] + ly * row1[o];
    }
    __syncthreads();
    float loss = 0.f, cnt = 0.f;
    for (int px = threadIdx.x; px < npx; px += 256) {
        const int x = xa + px;
        int x0, x1;
        float lx;
        ax.src(x, x0, x1, lx);
        lam[px] = lx;
        const long lab = labels[((long)b * H + y) * W + x];
        float* d = dbuf + (long)px * K;
        if (x0 >= j0) count_bad_label(lab, K, ignore_index, bad);       // once per pixel: by the tile that owns it
        if (lab == ignore_index || lab < 0 || lab >= K) {
            for (int k = 0; k < K; ++k) d[k] = 0.f;
            continue;
        }
        const float* c0 = vrow + (x0 - cbase) * K;
        const float* c1 = vrow + (x1 - cbase) * K;
        float v[KR];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            if (k < K) {
                v[k] = (1.f - lx) * c0[k] + lx * c1[k];
                mx = fmaxf(mx, v[k]);
            }
        }
        float se = 0.f, picked = 0.f;          // picked = x[label] - max BEFORE the exponential: exp() of it underflows to 0 for
#pragma unroll                                // a confidently wrong pixel (|logit| gap > 87) and log(0) would make the loss inf
        for (int k = 0; k < KR; ++k) {
            if (k < K) {
                if (k == lab) picked = v[k] - mx;
                v[k] = __expf(v[k] - mx);
                se += v[k];
            }
        }
        const float rse = 1.f / se;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            if (k < K) d[k] = v[k] * rse - (k == lab ? 1.f : 0.f);
        }
        if (x0 >= j0) {   // the tile that owns x0 accounts for the loss (x0 == j0-1 pixels belong to the previous tile)
            loss += __logf(se) - picked;          // = logsumexp(x) - x[label], like ATen's log_softmax + nll_loss
            cnt += 1.f;
        }
    }
    __syncthreads();
    if (tmp) {
        const int nj = j1 - j0;
        for (int item = threadIdx.x; item < nj * K; item += 256) {
            const int jj = item / K, k = item - jj * K;
            const int j = j0 + jj;
            float s = 0.f;
            // pixels with x0 == j-1 contribute lam to j (as x1), then pixels with x0 == j contribute 1-lam (and lam too when x1 is
            // clamped onto j at the right edge); same weights and the same ascending-x order as a per-pixel test of x0 / x1
            const int p0 = max(pstart[jj], 0), p1 = min(max(pstart[jj + 1], 0), npx), p2 = min(pstart[jj + 2], npx);
            for (int px = p0; px < p1; ++px) s += (0.f + lam[px]) * dbuf[(long)px * K + k];
            const bool edge = j == w - 1;
            for (int px = p1; px < p2; ++px) s += ((1.f - lam[px]) + (edge ? lam[px] : 0.f)) * dbuf[(long)px * K + k];
            tmp[(((long)b * H + y) * w + j) * K + k] = s;
        }
    }
    block_sum2(loss, cnt, red);
    if (threadIdx.x == 0) {
        const long pidx = ((long)b * H + y) * gridDim.x + jt;
        partial[2 * pidx] = loss;
        partial[2 * pidx + 1] = cnt;
    }
}

// pass 2: dlow[b][i][j][k] = grad_scale / n_valid * sum_y wy(y,i) tmp[b][y][j][k]   (ascending y)
__global__ void upce_pass2_kernel(const float* __restrict__ tmp, const float* __restrict__ loss_out, float* __restrict__ dlow, int B, int K,
                                  Axis ay, int w, float grad_scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H = ay.n_out, h = ay.n_in;
    const long per_row = (long)w * K;
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
    dlow[idx] = s * (grad_scale / loss_out[1]);
}

// inference tail: probs NCHW + optional argmax
__global__ void upsample_softmax_kernel(const float* __restrict__ low, float* __restrict__ probs, uint8_t* __restrict__ pred, int B, int K,
                                        Axis ay, Axis ax) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H = ay.n_out, W = ax.n_out, h = ay.n_in, w = ax.n_in;
    if (idx >= (long)B * H * W) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((long)W * H));
    int y0, y1, x0, x1;
    float ly, lx;
    ay.src(y, y0, y1, ly);
    ax.src(x, x0, x1, lx);
    const float* p00 = low + (((long)b * h + y0) * w + x0) * K;
    const float* p01 = low + (((long)b * h + y0) * w + x1) * K;
    const float* p10 = low + (((long)b * h + y1) * w + x0) * K;
    const float* p11 = low + (((long)b * h + y1) * w + x1) * K;
    float v[KMAX];
    float mx = -3.0e38f;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            v[k] = lerp2(p00[k], p01[k], p10[k], p11[k], lx, ly);
            if (v[k] > mx) {
                mx = v[k];
                arg = k;
            }
        }
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            v[k] = __expf(v[k] - mx);
            se += v[k];
        }
    }
    const float rse = 1.f / se;
    float* o = probs + ((long)b * K * H + y) * W + x;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) o[(long)k * H * W] = v[k] * rse;
    }
    if (pred) pred[idx] = (uint8_t)arg;
}

inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

inline int pass1_npx_max(const Axis& ax, int jt_cols) {
    // upper bound of pixels whose x0 falls in jt_cols+1 consecutive source columns
    if (ax.scale <= 0.f) return ax.n_out;
    const long n = (long)((float)(jt_cols + 1) / ax.scale) + 4;
    return (int)(n < ax.n_out ? n : ax.n_out);
}

// Low-res columns per workgroup: 32 up to an 8x upsample, fewer above (a tile of 32 columns at 32x is 1 056 pixels x 19 classes = 80 KB of LDS: one
// workgroup per CU - the 1/32 head of GALD took 788 us against 204 us for the 1/4 head with the same 5.5 M pixels); about 256 pixels per tile.
inline int pick_jt(int w, int W) {
    const int f = w > 0 ? (W + w - 1) / w : 1;
    int jt = JT;
    while (jt > 4 && jt * f > 256) jt >>= 1;
    return jt;
}

}  // namespace

extern "C" int mi_upsample_ac_fwd(const float* low, float* up, int B, int h, int w, int K, int H, int W, void* stream) {
    MI_REQUIRE(low && up && B > 0 && h > 0 && w > 0 && K > 0 && H > 0 && W > 0, "mi_upsample_ac_fwd: bad argument");
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(nblk((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, low, up, B, K, make_axis(h, H),
                       make_axis(w, W));
    MI_CHECK_LAUNCH("mi_upsample_ac_fwd");
    return MI_OK;
}

extern "C" int mi_upsample_ac_bwd(const float* dup, float* dlow, int B, int h, int w, int K, int H, int W, void* stream) {
    MI_REQUIRE(dup && dlow && B > 0 && h > 0 && w > 0 && K > 0 && H > 0 && W > 0, "mi_upsample_ac_bwd: bad argument");
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(nblk((long)B * K * h * w, 256)), dim3(256), 0, (hipStream_t)stream, dup, dlow, B, K,
                       make_axis(h, H), make_axis(w, W));
    MI_CHECK_LAUNCH("mi_upsample_ac_bwd");
    return MI_OK;
}

extern "C" size_t mi_ce_workspace(int B, int H, int W) { return (size_t)nblk((long)B * H * W, 256) * 2 * sizeof(float); }

extern "C" int mi_softmax_ce_fwd(const float* logits, const int64_t* labels, float* loss_out, int B, int K, int H, int W, int ignore_index,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(logits && labels && loss_out && workspace && B > 0 && K > 0 && H > 0 && W > 0, "mi_softmax_ce_fwd: bad argument");
    if (workspace_bytes < mi_ce_workspace(B, H, W)) return mi_set_error(MI_ENOMEM, "mi_softmax_ce_fwd: workspace too small");
    const unsigned nb = nblk((long)B * H * W, 256);
    unsigned* bad = reinterpret_cast<unsigned*>(loss_out + 3);
    if (hipMemsetAsync(bad, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return mi_set_error(MI_EHIP, "mi_softmax_ce_fwd: memset");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits, labels, (float*)workspace, B, K, (long)H * W, ignore_index, bad);
    MI_CHECK_LAUNCH("mi_softmax_ce_fwd");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, (int)nb, loss_out);
    MI_CHECK_LAUNCH("mi_softmax_ce_fwd finalize");
    return MI_OK;
}

extern "C" int mi_softmax_ce_bwd(const float* logits, const int64_t* labels, const float* loss_out, float* dlogits, int B, int K, int H, int W,
                                 int ignore_index, float grad_scale, void* stream) {
    MI_REQUIRE(logits && labels && loss_out && dlogits && B > 0 && K > 0 && H > 0 && W > 0, "mi_softmax_ce_bwd: bad argument");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(nblk((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, loss_out, dlogits, B, K,
                       (long)H * W, ignore_index, grad_scale);
    MI_CHECK_LAUNCH("mi_softmax_ce_bwd");
    return MI_OK;
}

extern "C" size_t mi_upsample_ce_workspace(int B, int h, int w, int K, int H, int W) {
    const int jt_cols = pick_jt(w, W);
    const size_t tiles = (size_t)((w + jt_cols - 1) / jt_cols);
    const size_t partial = (size_t)B * H * tiles * 2 * sizeof(float);
    const size_t tmp = (size_t)B * H * w * K * sizeof(float);
    return ((partial + 255) & ~(size_t)255) + tmp;
}

extern "C" int mi_upsample_ce_ex(const float* low, const int64_t* labels, float* loss_out, float* dlow, int B, int h, int w, int K, int H, int W,
                                 int ignore_index, float grad_scale, int align_corners, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int mi_upsample_ce(const float* low, const int64_t* labels, float* loss_out, float* dlow, int B, int h, int w, int K, int H, int W,
                              int ignore_index, float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
    return mi_upsample_ce_ex(low, labels, loss_out, dlow, B, h, w, K, H, W, ignore_index, grad_scale, 1, workspace, workspace_bytes, stream);
}

extern "C" int mi_upsample_ce_ex(const float* low, const int64_t* labels, float* loss_out, float* dlow, int B, int h, int w, int K, int H, int W,
                                 int ignore_index, float grad_scale, int align_corners, void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(low && labels && loss_out && workspace, "mi_upsample_ce: null operand");
    MI_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && K > 0 && K <= KMAX, "mi_upsample_ce: bad dimension (K <= 32)");
    MI_REQUIRE(H >= h && W >= w, "mi_upsample_ce: only upsampling (H >= h, W >= w) is supported");
    MI_REQUIRE(H <= 65535 && B <= 65535, "mi_upsample_ce: grid dimension overflow");
    if (workspace_bytes < mi_upsample_ce_workspace(B, h, w, K, H, W)) return mi_set_error(MI_ENOMEM, "mi_upsample_ce: workspace too small");
    const Axis ay = make_axis(h, H, align_corners), ax = make_axis(w, W, align_corners);
    const int jt_cols = pick_jt(w, W);
    const int tiles = (w + jt_cols - 1) / jt_cols;
    float* partial = (float*)workspace;
    const size_t poff = (((size_t)B * H * tiles * 2 * sizeof(float)) + 255) & ~(size_t)255;
    float* tmp = dlow ? (float*)((char*)workspace + poff) : nullptr;
    const int npx_max = pass1_npx_max(ax, jt_cols);
    const size_t lds = (size_t)npx_max * K * 4 + (size_t)npx_max * 8 + 512 * 4 + (JT + 4) * 4 + (size_t)(JT + 2) * K * 4;
    MI_REQUIRE(lds <= 160 * 1024, "mi_upsample_ce: upsample factor too large for one LDS tile (%zu B)", lds);
    static std::atomic<uint64_t> lds_set[2];           // the launch size varies with the upsample factor: allow the maximum once per device
    mi_allow_dynamic_lds((const void*)upce_pass1_kernel<19>, MI_LDS_MAX, lds_set[0]);
    mi_allow_dynamic_lds((const void*)upce_pass1_kernel<0>, MI_LDS_MAX, lds_set[1]);
    unsigned* bad = reinterpret_cast<unsigned*>(loss_out + 3);
    if (hipMemsetAsync(bad, 0, sizeof(unsigned), (hipStream_t)stream) != hipSuccess) return mi_set_error(MI_EHIP, "mi_upsample_ce: memset");
    if (K == 19)
        hipLaunchKernelGGL(upce_pass1_kernel<19>, dim3(tiles, H, B), dim3(256), lds, (hipStream_t)stream, low, labels, partial, tmp, B, K, ay,
                           ax, ignore_index, npx_max, bad, jt_cols);
    else
        hipLaunchKernelGGL(upce_pass1_kernel<0>, dim3(tiles, H, B), dim3(256), lds, (hipStream_t)stream, low, labels, partial, tmp, B, K, ay,
                           ax, ignore_index, npx_max, bad, jt_cols);
    MI_CHECK_LAUNCH("mi_upsample_ce pass1");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, B * H * tiles, loss_out);
    MI_CHECK_LAUNCH("mi_upsample_ce finalize");
    if (dlow) {
        hipLaunchKernelGGL(upce_pass2_kernel, dim3(nblk((long)B * h * w * K, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, loss_out,
                           dlow, B, K, ay, w, grad_scale);
        MI_CHECK_LAUNCH("mi_upsample_ce pass2");
    }
    return MI_OK;
}

extern "C" int mi_upsample_softmax(const float* low, float* probs, uint8_t* pred, int B, int h, int w, int K, int H, int W, void* stream) {
    MI_REQUIRE(low && probs && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && K > 0 && K <= KMAX, "mi_upsample_softmax: bad argument (K <= 32)");
    hipLaunchKernelGGL(upsample_softmax_kernel, dim3(nblk((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, low, probs, pred, B, K,
                       make_axis(h, H), make_axis(w, W));
    MI_CHECK_LAUNCH("mi_upsample_softmax");
    return MI_OK;
}
