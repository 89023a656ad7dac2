// Weight gradient as a pixel-contraction GEMM for gfx950 (bf16 operands, fp32 MFMA accumulate).
//
//   dw[o][i][t] = scale[o] * sum_m dy[m][o] * x[src(m,t)][i]          m = (b,ho,wo) pixel
//
// Both operands have the contraction index (pixel) as their memory ROW (NHWC), so MFMA fragments need the
// K index across rows: tiles are staged row-major into LDS and read back with ds_read_b64_tr_b16 (hardware
// transpose).  K = B*Ho*Wo (75 272 at the headline config) is split across S workgroups per output tile;
// every split writes an fp32 slab and a second kernel sums the slabs in a fixed order (bitwise
// reproducible; no float atomics), applies the folded FrozenBN scale and scatters to torch's OIHW layout.
// Replaces the weight half of convolution_backward for reference core/components/resnet.py:22-30 convs and
// (out_map 1) for the four ASPP convs of core/models/classifiers/aspp/classifier.py:12-20.
#include "mi_common.h"

namespace {

constexpr int TO = 128, TI = 128, KP = 64;      // output tile 128(o) x 128(i), 64 pixels per K step
constexpr int ROWB = 288;                       // 256 B of data + 32 B pad: tr reads are bank-conflict free
constexpr int TILE_BYTES = KP * ROWB;           // 18 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;      // 72 KiB

struct WgradParams {
    const __bf16* dY;
    const __bf16* X;
    float* slab;
    int M, O, I, T;
    int Ho, Wo, Ha, Wa;
    int ksz, stride, pad, dil;
    int S, rows_per_split;
    int o_tiles, i_tiles;
};

__device__ __forceinline__ s16x4 tr_read(const char* addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
}

__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ot = blockIdx.x / p.i_tiles, it = blockIdx.x - ot * p.i_tiles;
    const int o0 = ot * TO, i0 = it * TI;
    const int t = blockIdx.y, split = blockIdx.z;
    const int ky = t / p.ksz, kx = t - ky * p.ksz;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nk = (m_end - m_begin + KP - 1) / KP;

    // staging: each tile is 64 rows x 16 chunks of 16 B -> 4 chunks per thread: rows r0 + 16*j, chunk c
    const int c = tid & 15, r0 = tid >> 4;
    const bool dy_col_ok = (o0 + c * 8) < p.O;
    const bool x_col_ok = (i0 + c * 8) < p.I;
    const int HoWo = p.Ho * p.Wo;

    u32x4 rdy[4], rx[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m_begin + kt * KP + r0 + 16 * j;
            u32x4 vdy = {0, 0, 0, 0}, vx = {0, 0, 0, 0};
            if (m < m_end) {
                if (dy_col_ok) vdy = *reinterpret_cast<const u32x4*>(p.dY + (long)m * p.O + o0 + c * 8);
                if (x_col_ok) {
                    const int b = m / HoWo, rem = m - b * HoWo;
                    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    const int ha = ho * p.stride + ky * p.dil - p.pad, wa = wo * p.stride + kx * p.dil - p.pad;
                    if ((unsigned)ha < (unsigned)p.Ha && (unsigned)wa < (unsigned)p.Wa)
                        vx = *reinterpret_cast<const u32x4*>(p.X + ((long)(b * p.Ha + ha) * p.Wa + wa) * p.I + i0 + c * 8);
                }
            }
            rdy[j] = vdy;
            rx[j] = vx;
        }
    };
    auto store_tile = [&](int stage) {
        char* sy = smem + stage * STAGE_BYTES;
        char* sx = sy + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (r0 + 16 * j) * ROWB + c * 16;
            *reinterpret_cast<u32x4*>(sy + off) = rdy[j];
            *reinterpret_cast<u32x4*>(sx + off) = rx[j];
        }
    };

    // MFMA orientation: D rows = i (A operand = X^T), D cols = o (B operand = dY^T); wave owns 64(i) x 64(o).
    const int wi = wave & 1, wo_ = wave >> 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: 16-lane group g reads pixel rows ks*32 + half*16 + g*4 + q (q = (lane&15)>>2),
    // 4 columns starting at 4*(lane&3) of the 16-column subtile; the same pixel<->k mapping on both operands.
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int lane_off = (g * 4 + q) * ROWB + pc * 8;
    auto compute = [&](int stage) {
        const char* sy = smem + stage * STAGE_BYTES;
        const char* sx = sy + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[4], yf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const char* base = sx + lane_off + (ks * 32) * ROWB + (wi * 64 + a * 16) * 2;
                const s16x4 lo = tr_read(base), hi = tr_read(base + 16 * ROWB);
                union { bf16x8 v; s16x4 h[2]; } u;
                u.h[0] = lo;
                u.h[1] = hi;
                xf[a] = u.v;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const char* base = sy + lane_off + (ks * 32) * ROWB + (wo_ * 64 + b * 16) * 2;
                const s16x4 lo = tr_read(base), hi = tr_read(base + 16 * ROWB);
                union { bf16x8 v; s16x4 h[2]; } u;
                u.h[0] = lo;
                u.h[1] = hi;
                yf[b] = u.v;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a], yf[b], acc[a][b], 0, 0, 0);
        }
    };

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = (kt + 1) < nk;
            if (more) load_tile(kt + 1);
            compute(cur);
            if (more) store_tile(cur ^ 1);
            __syncthreads();
        }
    }

    // slab[split][t][o][i], i fastest: lane owns o = col, i..i+3 = rows
    float* slab = p.slab + ((long)(split * p.T + t) * p.O) * p.I;
    const int fcol = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int o = o0 + wo_ * 64 + b * 16 + fcol;
        if (o >= p.O) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = i0 + wi * 64 + a * 16 + fq * 4;
            if (i >= p.I) continue;
            *reinterpret_cast<f32x4*>(slab + (long)o * p.I + i) = acc[a][b];
        }
    }
}

// dw[o][i][t] = scale[o] * sum_s slab[s][t][o][i]   (fixed summation order -> reproducible)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, const float* __restrict__ scale,
                                    int S, int T, int O, int I, int accumulate, int out_map, int o_real, int ncls) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)o_real * I) return;
    const int o = (int)(idx / I), i = (int)(idx - (long)o * I);
    const long plane = (long)O * I;
    const float sc = scale ? scale[o] : 1.f;
    for (int t = 0; t < T; ++t) {
        float s = 0.f;
        const float* src = slab + (long)t * plane + (long)o * I + i;
        for (int k = 0; k < S; ++k) s += src[(long)k * T * plane];
        s *= sc;
        long dst;
        if (out_map == 0) {
            dst = ((long)o * I + i) * T + t;
        } else {  // ASPP: o = (r*9+tap)*ncls + cls  ->  dw[r][cls][i][tap]
            const int grp = o / ncls, cls = o - grp * ncls;
            const int r = grp / 9, tap = grp - r * 9;
            dst = (((long)r * ncls + cls) * I + i) * 9 + tap;
        }
        dw[dst] = accumulate ? dw[dst] + s : s;
    }
}

int pick_splits(long M, int tiles) {
    // aim for ~1024 workgroups (4 per CU at 2 resident) but keep >= 8 K-steps of 64 pixels per split
    int s = (int)((1024 + tiles - 1) / tiles);
    const long max_s = (M + 8 * KP - 1) / (8 * KP);
    if (s > max_s) s = (int)max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

}  // namespace

extern "C" size_t mi_conv_wgrad_workspace(int B, int Ho, int Wo, int O, int I, int ksize) {
    const long M = (long)B * Ho * Wo;
    const int T = ksize * ksize;
    const int tiles = ((O + TO - 1) / TO) * ((I + TI - 1) / TI) * T;
    const int S = pick_splits(M, tiles);
    return (size_t)S * T * O * I * sizeof(float);
}

extern "C" int mi_conv_wgrad(const void* dy, const void* x, float* dw, int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                             int ksize, int stride, int pad, int dil, const float* scale_o, int accumulate, int out_map,
                             void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(dy && x && dw && workspace, "mi_conv_wgrad: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0, "mi_conv_wgrad: non-positive dimension");
    MI_REQUIRE(O % 8 == 0 && I % 8 == 0, "mi_conv_wgrad: O=%d, I=%d must be multiples of 8", O, I);
    MI_REQUIRE(ksize == 1 || ksize == 3, "mi_conv_wgrad: ksize");
    MI_REQUIRE(out_map == 0 || (out_map == 1 && ksize == 1), "mi_conv_wgrad: out_map 1 needs ksize 1");
    MI_REQUIRE(mi_aligned16(dy) && mi_aligned16(x) && mi_aligned16(workspace), "mi_conv_wgrad: alignment");
    const long M = (long)B * Ho * Wo;
    MI_REQUIRE(M < (1L << 31) && (long)B * Ha * Wa < (1L << 31), "mi_conv_wgrad: pixel count overflows int32");
    const size_t need = mi_conv_wgrad_workspace(B, Ho, Wo, O, I, ksize);
    if (workspace_bytes < need) return mi_set_error(MI_ENOMEM, "mi_conv_wgrad: workspace %zu < %zu", workspace_bytes, need);
    WgradParams p;
    p.dY = (const __bf16*)dy;
    p.X = (const __bf16*)x;
    p.slab = (float*)workspace;
    p.M = (int)M;
    p.O = O;
    p.I = I;
    p.T = ksize * ksize;
    p.Ho = Ho;
    p.Wo = Wo;
    p.Ha = Ha;
    p.Wa = Wa;
    p.ksz = ksize;
    p.stride = stride;
    p.pad = pad;
    p.dil = dil;
    p.o_tiles = (O + TO - 1) / TO;
    p.i_tiles = (I + TI - 1) / TI;
    p.S = pick_splits(M, p.o_tiles * p.i_tiles * p.T);
    const long steps = (M + KP - 1) / KP;
    p.rows_per_split = (int)(((steps + p.S - 1) / p.S) * KP);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3(p.o_tiles * p.i_tiles, p.T, p.S), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_wgrad");
    int o_real = O, ncls = 1;
    if (out_map == 1) {
        ncls = 19;
        o_real = 36 * ncls;
        MI_REQUIRE(O >= o_real, "mi_conv_wgrad: out_map 1 needs O >= 684");
    }
    const long n = (long)o_real * I;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p.slab, dw, scale_o,
                       p.S, p.T, O, I, accumulate, out_map, o_real, ncls);
    MI_CHECK_LAUNCH("mi_conv_wgrad reduce");
    return MI_OK;
}
