// PraNet structure loss (first kernel of that path): reference core/trainers/pranet_trainer.py:22-31
//   weit = 1 + 5 * |avg_pool2d(mask, 31, stride 1, pad 15) - mask|               (zero padding counted: every window divides by 961)
//   wbce = binary_cross_entropy_with_logits(pred, mask, reduce='none')           -> torch reads the string as the legacy reduce=True:
//          the MEAN over the whole batch, a scalar; (weit * wbce).sum / weit.sum is then that scalar again for every image
//   p = sigmoid(pred); inter = sum(p * mask * weit), union = sum((p + mask) * weit) per image; wiou = 1 - (inter + 1) / (union - inter + 1)
//   loss = mean_i(wbce + wiou_i)
// and its gradient with respect to pred:
//   d loss / d x = (p - z) / (B*H*W)  -  (1/B) * w * (z * D_i - N_i * (1 - z)) / D_i^2 * p * (1 - p),   N_i = inter_i + 1, D_i = union_i - inter_i + 1
// pred / mask fp32 [B][H][W] (the maps have one channel).  Sums are two-level with a fixed order (bitwise reproducible); the box filter is
// separable (31 + 31 taps).  Bound: launch latency (the maps are 352 x 352 x B).
#include "mi_common.h"

namespace {


__global__ __launch_bounds__(256) void sl_hbox_kernel(const float* __restrict__ mask, float* __restrict__ tmp, int B, int H, int W) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * H * W) return;
    const int w = (int)(idx % W);
    const float* row = mask + (idx - w);
    float s = 0.f;
    // (unconditional loads at a clamped index + a select: a branch around a load makes hipcc wait for every load before the next one -
    //  31 dependent round trips per pixel; the first version of these two kernels took 23 + 89 us for 16 maps of 352 x 352)
#pragma unroll
    for (int dx = -15; dx <= 15; ++dx) {
        const int ww = w + dx;
        const float v = row[min(max(ww, 0), W - 1)];
        s += (unsigned)ww < (unsigned)W ? v : 0.f;
    }
    tmp[idx] = s;
}

// weit, and per-workgroup partial sums {bce, inter, union} over a 32 x 64 pixel tile of ONE image.  The tile's 62 x 64 window of row sums goes through
// LDS once (the first version read its 31 rows per pixel from global memory: 212 MB of HBM traffic and 97 us per launch for 16 maps of 352 x 352 -
// counters of profiles/pmc_pranet.json); every pixel then adds its 31 taps from LDS in the same order as before.
constexpr int SLT_H = 32, SLT_W = 64;
__global__ __launch_bounds__(256) void sl_terms_kernel(const float* __restrict__ pred, const float* __restrict__ mask, const float* __restrict__ tmp,
                                                       float* __restrict__ weit, float* __restrict__ partial, int H, int W) {
    __shared__ float win[(SLT_H + 30) * SLT_W];
    __shared__ float red[3][256];
    const int b = blockIdx.z;
    const int h0 = blockIdx.y * SLT_H, w0 = blockIdx.x * SLT_W;
    const long base = (long)b * H * W;
    for (int e = threadIdx.x; e < (SLT_H + 30) * SLT_W; e += 256) {
        const int r = e / SLT_W, c = e - r * SLT_W;
        const int hh = h0 - 15 + r, ww = w0 + c;
        const float v = tmp[base + (long)min(max(hh, 0), H - 1) * W + min(ww, W - 1)];          // unconditional load, then select
        win[e] = ((unsigned)hh < (unsigned)H && ww < W) ? v : 0.f;
    }
    __syncthreads();
    const int cx = threadIdx.x & (SLT_W - 1), ry = threadIdx.x / SLT_W;
    float a_bce = 0.f, a_int = 0.f, a_uni = 0.f;
    for (int r = ry; r < SLT_H; r += 256 / SLT_W) {
        const int h = h0 + r, w = w0 + cx;
        if (h >= H || w >= W) continue;
        float box = 0.f;
#pragma unroll
        for (int dy = 0; dy <= 30; ++dy) box += win[(r + dy) * SLT_W + cx];       // rows h - 15 .. h + 15 (zeros outside the image)
        box *= (1.0f / 961.0f);
        const long p = base + (long)h * W + w;
        const float z = mask[p], x = pred[p];
        const float wt = 1.0f + 5.0f * fabsf(box - z);
        weit[p] = wt;
        // max(x, 0) - x z + log(1 + exp(-|x|)): torch's stable form of the logistic loss
        a_bce += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
        const float s = 1.0f / (1.0f + expf(-x));
        a_int += s * z * wt;
        a_uni += (s + z) * wt;
    }
    red[0][threadIdx.x] = a_bce;
    red[1][threadIdx.x] = a_int;
    red[2][threadIdx.x] = a_uni;
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.f;
        for (int k = 0; k < 256; ++k) s += red[threadIdx.x][k];              // fixed order
        partial[((long)b * gridDim.x * gridDim.y + blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = s;
    }
}

// one workgroup: per image the partials in ascending order, then the images in ascending order.  out = {loss, N_0, D_0, N_1, D_1, ...}.
// Thread (image, term) adds its image's partials (the first version did all B * nblk * 3 dependent loads on one thread: 79 us for 16 images);
// thread 0 then combines the images in ascending order - the same additions in the same order as before.
__global__ __launch_bounds__(256) void sl_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int B, int nblk, float inv_pixels) {
    __shared__ float sums[256];
    float bce = 0.f, iou = 0.f;
    for (int b0 = 0; b0 < B; b0 += 85) {                     // 85 images (x 3 terms) per pass
        const int bl = threadIdx.x / 3, term = threadIdx.x - bl * 3, b = b0 + bl;
        if (threadIdx.x < 255 && b < B) {
            float s = 0.f;
            for (int k = 0; k < nblk; ++k) s += partial[((long)b * nblk + k) * 3 + term];
            sums[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 0; i < 85 && b0 + i < B; ++i) {
                const float sb = sums[3 * i], si = sums[3 * i + 1], su = sums[3 * i + 2];
                bce += sb;
                const float N = si + 1.0f, D = su - si + 1.0f;
                out[1 + 2 * (b0 + i)] = N;
                out[2 + 2 * (b0 + i)] = D;
                iou += 1.0f - N / D;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = bce * inv_pixels + iou / (float)B;
}

__global__ __launch_bounds__(256) void sl_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ mask, const float* __restrict__ weit,
                                                     const float* __restrict__ nd, float* __restrict__ grad, int B, int HW, float inv_pixels,
                                                     float inv_b, float gscale) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * HW) return;
    const int b = (int)(idx / HW);
    const float N = nd[1 + 2 * b], D = nd[2 + 2 * b];
    const float x = pred[idx], z = mask[idx], w = weit[idx];
    const float s = 1.0f / (1.0f + expf(-x));
    const float diou = -w * (z * D - N * (1.0f - z)) / (D * D);
    grad[idx] = gscale * ((s - z) * inv_pixels + inv_b * diou * s * (1.0f - s));
}

}  // namespace

static inline int sl_tiles(int H, int W) { return ((H + SLT_H - 1) / SLT_H) * ((W + SLT_W - 1) / SLT_W); }

extern "C" size_t mi_structure_loss_workspace(int B, int H, int W) {
    return ((size_t)2 * B * H * W + (size_t)B * sl_tiles(H, W) * 3) * sizeof(float);
}

extern "C" int mi_structure_loss(const float* pred, const float* mask, int B, int H, int W, float* out, float* grad, float grad_scale,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(pred && mask && out && workspace, "mi_structure_loss: null operand");
    MI_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "mi_structure_loss: bad shape");
    MI_REQUIRE(workspace_bytes >= mi_structure_loss_workspace(B, H, W), "mi_structure_loss: workspace too small");
    const long n = (long)B * H * W;
    float* tmp = (float*)workspace;
    float* weit = tmp + n;
    float* partial = weit + n;
    const int HW = H * W;
    const int nblk = sl_tiles(H, W);
    const float inv_pixels = 1.0f / (float)n;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sl_hbox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mask, tmp, B, H, W);
    hipLaunchKernelGGL(sl_terms_kernel, dim3((W + SLT_W - 1) / SLT_W, (H + SLT_H - 1) / SLT_H, B), dim3(256), 0, st, pred, mask, tmp, weit, partial, H, W);
    hipLaunchKernelGGL(sl_final_kernel, dim3(1), dim3(256), 0, st, partial, out, B, nblk, inv_pixels);
    if (grad)
        hipLaunchKernelGGL(sl_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pred, mask, weit, out, grad, B, HW, inv_pixels,
                           1.0f / (float)B, grad_scale);
    MI_CHECK_LAUNCH("mi_structure_loss");
    return MI_OK;
}
