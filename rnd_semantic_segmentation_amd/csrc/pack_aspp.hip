// Weight packing (fp32 OIHW masters -> bf16 GEMM operands) and the ASPP head's data-movement kernels.
// ASPP: reference core/models/classifiers/aspp/classifier.py:6-32 (4 dilated 3x3 convs, summed).
#include "mi_common.h"

namespace {

// one thread per (o,i): reads k*k contiguous floats, writes one element into each tap plane
__global__ void pack_fwd_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int O, int I, int T) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)O * I) return;
    const long plane = (long)O * I;
    for (int t = 0; t < T; ++t) wp[t * plane + idx] = (__bf16)w[idx * T + t];
}

// wp[t][i][o] = bf16(w[o][i][t] * scale[o]); thread per (i,o) so writes are coalesced over o
__global__ void pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale, __bf16* __restrict__ wp, int O, int I, int T) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)O * I) return;
    const int i = (int)(idx / O), o = (int)(idx - (long)i * O);
    const float s = scale ? scale[o] : 1.f;
    const long plane = (long)O * I;
    const float* src = w + ((long)o * I + i) * T;
    for (int t = 0; t < T; ++t) wp[t * plane + idx] = (__bf16)(src[t] * s);
}

// wall[(g*20+n)][c], g = r*9+tap
__global__ void aspp_pack_fwd_kernel(const float* __restrict__ w4, __bf16* __restrict__ wall, int C, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = 36L * MI_ASPP_ZGW * C;
    if (idx >= total) return;
    const int row = (int)(idx / C), c = (int)(idx - (long)row * C);
    const int g = row / MI_ASPP_ZGW, n = row - g * MI_ASPP_ZGW;
    float v = 0.f;
    if (n < K) {
        const int r = g / 9, tap = g - r * 9;
        v = w4[(((long)r * K + n) * C + c) * 9 + tap];
    }
    wall[idx] = (__bf16)v;
}

// wallT[c][k], k = g*K+n (k >= 36*K zero)
__global__ void aspp_pack_dgrad_kernel(const float* __restrict__ w4, __bf16* __restrict__ wallT, int C, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)C * MI_ASPP_KPAD;
    if (idx >= total) return;
    const int c = (int)(idx / MI_ASPP_KPAD), k = (int)(idx - (long)c * MI_ASPP_KPAD);
    float v = 0.f;
    if (k < 36 * K) {
        const int g = k / K, n = k - g * K;
        const int r = g / 9, tap = g - r * 9;
        v = w4[(((long)r * K + n) * C + c) * 9 + tap];
    }
    wallT[idx] = (__bf16)v;
}

// All conv weights of a module in ONE launch: table row = {w_off, scale_off(-1: none), wp_off, wpt_off(-1: skip), O, I, T,
// first_block}; a block handles up to four 32 (o) x 32 (i) tiles (consecutive along i) of one tensor: the 32 * T floats of an o row are contiguous in OIHW and are
// read as such into LDS; both packs then leave as 64-byte runs (32 consecutive i of the forward operand [t][o][i], 32 consecutive o
// of the data-gradient operand [t][i][o]).  A thread per (o, i) pair writing its T taps scattered the data-gradient pack as
// single bf16 stores O * 2 bytes apart: 303 us per step for 42 M weights, 4.4x the time of the bytes it moves.
constexpr int PACK_TILES = 4;      // consecutive 32 x 32 tiles (along i) per block: one table search for 4-36 K weights
__global__ __launch_bounds__(256) void pack_multi_kernel(const float* __restrict__ wflat, const float* __restrict__ sflat, __bf16* __restrict__ wp,
                                                         __bf16* __restrict__ wpt, const long* __restrict__ table, int n_desc) {
    __shared__ float tile[32][32 * 9 + 1];           // odd row stride: the transposed read (lanes over o) is conflict-free
    int lo = 0, hi = n_desc - 1;                     // last row whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 7] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long* d = table + lo * 8;
    const int O = (int)d[4], I = (int)d[5], T = (int)d[6];
    const int tiles_i = (I + 31) >> 5, groups_i = (tiles_i + PACK_TILES - 1) / PACK_TILES;
    const int local = (int)((long)blockIdx.x - d[7]);
    const int ot = local / groups_i;
    const int o0 = ot * 32, it0 = (local - ot * groups_i) * PACK_TILES;
    if (o0 >= O) return;
    const int no = min(32, O - o0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long plane = (long)O * I;
    const int c = tid & 31, r = tid >> 5;
    const float sc = (d[3] >= 0 && d[1] >= 0 && c < no) ? sflat[d[1] + o0 + c] : 1.f;
    for (int it = it0; it < it0 + PACK_TILES && it < tiles_i; ++it) {
        const int i0 = it * 32;
        const int ni = min(32, I - i0);
        const int rowlen = ni * T;
        if (it != it0) __syncthreads();
        for (int o = wave; o < no; o += 4) {
            const float* src = wflat + d[0] + ((long)(o0 + o) * I + i0) * T;
            for (int e = lane; e < rowlen; e += 64) tile[o][e] = src[e];
        }
        __syncthreads();
        // forward operand wp[t][o][i]: lanes over i
        if (c < ni) {
            for (int k = 0; k < 4; ++k) {
                const int o = r + 8 * k;
                if (o >= no) break;
                __bf16* f = wp + d[2] + (long)(o0 + o) * I + i0 + c;
                for (int t = 0; t < T; ++t) f[t * plane] = (__bf16)tile[o][c * T + t];
            }
        }
        // data-gradient operand wpt[t][i][o] with the FrozenBN scale of output channel o folded in: lanes over o
        if (d[3] >= 0 && c < no) {
            for (int k = 0; k < 4; ++k) {
                const int i = r + 8 * k;
                if (i >= ni) break;
                __bf16* b = wpt + d[3] + (long)(i0 + i) * O + o0 + c;
                for (int t = 0; t < T; ++t) b[t * plane] = (__bf16)(tile[c][i * T + t] * sc);
            }
        }
    }
}

struct Rates { int d[4]; };

// low[m][n] = sum_r bias[r][n] + sum_g Z[g][m + shift_g][n]; thread per (m, n) with n fastest (20 lanes per pixel)
__global__ void aspp_col2im_kernel(const float* __restrict__ z, const float* __restrict__ bias4, float* __restrict__ low,
                                   int B, int H, int W, int K, Rates rates) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long M = (long)B * H * W;
    if (idx >= M * MI_ASPP_ZGW) return;
    const long m = idx / MI_ASPP_ZGW;
    const int n = (int)(idx - m * MI_ASPP_ZGW);
    if (n >= K) return;
    const int hw = (int)(m % ((long)H * W));
    const int h = hw / W, w = hw - h * W;
    // same association as the reference: (((conv0 + conv1) + conv2) + conv3), each conv = bias + taps
    float total = 0.f;
    for (int r = 0; r < 4; ++r) {
        const int d = rates.d[r];
        float s = bias4[r * K + n];
        for (int ky = 0; ky < 3; ++ky) {
            const int hh = h + (ky - 1) * d;
            if ((unsigned)hh >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ww = w + (kx - 1) * d;
                if ((unsigned)ww >= (unsigned)W) continue;
                const int g = r * 9 + ky * 3 + kx;
                const long ms = m + (long)(ky - 1) * d * W + (kx - 1) * d;
                s += z[((long)g * M + ms) * MI_ASPP_ZGW + n];
            }
        }
        total = (r == 0) ? s : total + s;
    }
    low[m * K + n] = total;
}

// G[m][k] = dlow[m - shift_g][n], k = g*K+n; thread per (m, k8) writing 8 bf16 (16 B)
__global__ void aspp_im2col_kernel(const float* __restrict__ dlow, __bf16* __restrict__ gmat, int B, int H, int W, int K, Rates rates) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long M = (long)B * H * W;
    constexpr int CH = MI_ASPP_KPAD / 8;
    if (idx >= M * CH) return;
    const long m = idx / CH;
    const int k0 = (int)(idx - m * CH) * 8;
    const int hw = (int)(m % ((long)H * W));
    const int h = hw / W, w = hw - h * W;
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        float v = 0.f;
        if (k < 36 * K) {
            const int g = k / K, n = k - g * K;
            const int r = g / 9, tap = g - r * 9;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int d = rates.d[r];
            // out[p] += W_tap x[p + s]  =>  dx[q] += W_tap^T dout[q - s]
            const int hh = h - (ky - 1) * d, ww = w - (kx - 1) * d;
            if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)
                v = dlow[(m - (long)(ky - 1) * d * W - (kx - 1) * d) * K + n];
        }
        out[e] = (__bf16)v;
    }
    *reinterpret_cast<bf16x8*>(gmat + m * MI_ASPP_KPAD + k0) = out;
}

// Row-blocked forms of the two kernels above (one workgroup per output image row): every access to the big tensor is a 16-byte access to a
// contiguous row segment, and the small one (dlow) is staged in LDS once per source row instead of being re-fetched 4 bytes at a time.
//
// col2im: thread item = (pixel w, 4-class group j); a tap plane contributes the contiguous segment z[g][(b, h + (ky-1)d, *)][20 floats] shifted by
// (kx-1)d pixels: consecutive items read consecutive float4s.  Same association as the reference: (((conv0 + conv1) + conv2) + conv3), conv = bias + taps.
__global__ __launch_bounds__(256) void aspp_col2im_rows_kernel(const float* __restrict__ z, const float* __restrict__ bias4, float* __restrict__ low,
                                                               int B, int H, int W, int K, Rates rates) {
    constexpr int Q = MI_ASPP_ZGW / 4;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const long M = (long)B * H * W, mrow = ((long)b * H + h) * W;
    for (int item = threadIdx.x; item < W * Q; item += 256) {
        const int w = item / Q, j = item - w * Q;
        f32x4 total = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = rates.d[r];
            f32x4 s;
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = (4 * j + e < K) ? bias4[r * K + 4 * j + e] : 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int hh = h + (ky - 1) * d;
                if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ww = w + (kx - 1) * d;
                    if ((unsigned)ww >= (unsigned)W) continue;
                    const int g = r * 9 + ky * 3 + kx;
                    const long ms = mrow + (long)(ky - 1) * d * W + ww;
                    s += *reinterpret_cast<const f32x4*>(z + ((long)g * M + ms) * MI_ASPP_ZGW + 4 * j);
                }
            }
            total = (r == 0) ? s : total + s;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * j + e < K) low[(mrow + w) * K + 4 * j + e] = total[e];
    }
}

// im2col: the 12 source rows (rate r, kernel row ky) of dlow that an output row draws from are copied to LDS as bf16 in their own layout ([W][K]
// contiguous: a straight float4 copy, zero rows where the source row lies outside the image); an LDS table maps every column k = g*K + n of the
// patch matrix to (LDS offset of its source at pixel 0, pixel shift).  No division on the per-element path.
constexpr int IM2COL_ROW = 2496;                        // bf16 elements per source-row slot: W * K <= 2496 (W <= 131 at 19 classes); 12 slots + tables = 62.6 KiB
__global__ __launch_bounds__(256) void aspp_im2col_rows_kernel(const float* __restrict__ dlow, __bf16* __restrict__ gmat, int B, int H, int W, int K,
                                                               Rates rates) {
    __shared__ __attribute__((aligned(16))) __bf16 rows[12 * IM2COL_ROW];
    __shared__ __attribute__((aligned(16))) int kbase[MI_ASPP_KPAD];       // slot * IM2COL_ROW + shift * K + n; padding column: any valid offset
    __shared__ __attribute__((aligned(16))) short kshift[MI_ASPP_KPAD];    // pixel shift; padding column: -30000 (always out of range)
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const long mrow = ((long)b * H + h) * W;
    for (int k = threadIdx.x; k < MI_ASPP_KPAD; k += 256) {
        int base = 0, sh = -30000;
        if (k < 36 * K) {
            const int g = k / K, n = k - g * K;
            const int r = g / 9, tap = g - r * 9;
            const int ky = tap / 3, kx = tap - ky * 3;
            sh = -(kx - 1) * rates.d[r];                // out[p] += W_tap x[p + s]  =>  dx[q] += W_tap^T dout[q - s]
            base = (r * 3 + ky) * IM2COL_ROW + sh * K + n;
        }
        kbase[k] = base;
        kshift[k] = (short)sh;
    }
    typedef f32x4 __attribute__((aligned(4))) f32x4_a4;      // a source row starts at a multiple of W * K floats: 4-byte aligned, 16-byte loads all the same
    const int rowlen = W * K, row4 = rowlen >> 2;
    for (int slot = 0; slot < 12; ++slot) {
        const int r = slot / 3, ky = slot - r * 3;
        const int hh = h - (ky - 1) * rates.d[r];
        __bf16* dst = rows + slot * IM2COL_ROW;
        const bool inside = (unsigned)hh < (unsigned)H;
        const float* src = dlow + (((long)b * H + (inside ? hh : h)) * W) * K;
        for (int i = threadIdx.x; i < row4; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4_a4*>(src + 4 * i);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(inside ? v[e] : 0.f);
            *reinterpret_cast<bf16x4*>(dst + 4 * i) = o;
        }
        for (int i = 4 * row4 + threadIdx.x; i < rowlen; i += 256) dst[i] = (__bf16)(inside ? src[i] : 0.f);
    }
    __syncthreads();
    constexpr int CH = MI_ASPP_KPAD / 8;
    typedef int __attribute__((ext_vector_type(4))) i32x4;
    typedef short __attribute__((ext_vector_type(8))) s16x8;
    for (int item = threadIdx.x; item < W * CH; item += 256) {
        const int w = item / CH, k0 = (item - w * CH) * 8;
        const int wk = w * K;
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(kbase + k0), b1 = *reinterpret_cast<const i32x4*>(kbase + k0 + 4);
        const s16x8 sh = *reinterpret_cast<const s16x8*>(kshift + k0);
        bf16x8 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = (unsigned)(w + sh[e]) < (unsigned)W;
            const __bf16 x = rows[in ? (e < 4 ? b0[e] : b1[e - 4]) + wk : 0];
            out[e] = in ? x : (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(gmat + (mrow + w) * MI_ASPP_KPAD + k0) = out;
    }
}

inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

extern "C" int mi_pack_weight_fwd(const float* w, void* wp, int O, int I, int ksize, void* stream) {
    MI_REQUIRE(w && wp && O > 0 && I > 0 && ksize > 0, "mi_pack_weight_fwd: bad argument");
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(nblk((long)O * I, 256)), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wp, O, I, ksize * ksize);
    MI_CHECK_LAUNCH("mi_pack_weight_fwd");
    return MI_OK;
}

extern "C" int mi_pack_weight_dgrad(const float* w, const float* scale_o, void* wp, int O, int I, int ksize, void* stream) {
    MI_REQUIRE(w && wp && O > 0 && I > 0 && ksize > 0, "mi_pack_weight_dgrad: bad argument");
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(nblk((long)O * I, 256)), dim3(256), 0, (hipStream_t)stream, w, scale_o, (__bf16*)wp, O, I,
                       ksize * ksize);
    MI_CHECK_LAUNCH("mi_pack_weight_dgrad");
    return MI_OK;
}

extern "C" int mi_pack_weights_multi(const float* wflat, const float* sflat, void* wp, void* wpt, const int64_t* table_dev, int n_desc,
                                     int total_blocks, void* stream) {
    MI_REQUIRE(wflat && wp && table_dev && n_desc > 0 && total_blocks > 0, "mi_pack_weights_multi: bad argument");
    hipLaunchKernelGGL(pack_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, wflat, sflat, (__bf16*)wp, (__bf16*)wpt,
                       (const long*)table_dev, n_desc);
    MI_CHECK_LAUNCH("mi_pack_weights_multi");
    return MI_OK;
}

extern "C" int mi_aspp_pack_fwd(const float* w4, void* wall, int C, int K, void* stream) {
    MI_REQUIRE(w4 && wall && C > 0 && K > 0 && K < MI_ASPP_ZGW, "mi_aspp_pack_fwd: bad argument (K=%d must be < 20)", K);
    hipLaunchKernelGGL(aspp_pack_fwd_kernel, dim3(nblk(36L * MI_ASPP_ZGW * C, 256)), dim3(256), 0, (hipStream_t)stream, w4, (__bf16*)wall, C, K);
    MI_CHECK_LAUNCH("mi_aspp_pack_fwd");
    return MI_OK;
}

extern "C" int mi_aspp_pack_dgrad(const float* w4, void* wallT, int C, int K, void* stream) {
    MI_REQUIRE(w4 && wallT && C > 0 && K > 0 && 36 * K <= MI_ASPP_KPAD, "mi_aspp_pack_dgrad: bad argument");
    hipLaunchKernelGGL(aspp_pack_dgrad_kernel, dim3(nblk((long)C * MI_ASPP_KPAD, 256)), dim3(256), 0, (hipStream_t)stream, w4, (__bf16*)wallT, C, K);
    MI_CHECK_LAUNCH("mi_aspp_pack_dgrad");
    return MI_OK;
}

extern "C" int mi_aspp_col2im(const float* z, const float* bias4, float* low, int B, int H, int W, int K, const int* rates4, void* stream) {
    MI_REQUIRE(z && bias4 && low && rates4 && B > 0 && H > 0 && W > 0 && K > 0 && K < MI_ASPP_ZGW, "mi_aspp_col2im: bad argument");
    Rates r;
    for (int i = 0; i < 4; ++i) {
        MI_REQUIRE(rates4[i] >= 1, "mi_aspp_col2im: rate");
        r.d[i] = rates4[i];
    }
    if (mi_aligned16(z) && (long)B * H < (1L << 31)) {
        hipLaunchKernelGGL(aspp_col2im_rows_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, z, bias4, low, B, H, W, K, r);
    } else {
        const long n = (long)B * H * W * MI_ASPP_ZGW;
        hipLaunchKernelGGL(aspp_col2im_kernel, dim3(nblk(n, 320)), dim3(320), 0, (hipStream_t)stream, z, bias4, low, B, H, W, K, r);
    }
    MI_CHECK_LAUNCH("mi_aspp_col2im");
    return MI_OK;
}

extern "C" int mi_aspp_im2col(const float* dlow, void* g, int B, int H, int W, int K, const int* rates4, void* stream) {
    MI_REQUIRE(dlow && g && rates4 && B > 0 && H > 0 && W > 0 && K > 0 && 36 * K <= MI_ASPP_KPAD, "mi_aspp_im2col: bad argument");
    MI_REQUIRE(mi_aligned16(g), "mi_aspp_im2col: alignment");
    Rates r;
    for (int i = 0; i < 4; ++i) r.d[i] = rates4[i];
    bool small_shift = (long)W * K <= IM2COL_ROW;
    for (int i = 0; i < 4; ++i) small_shift = small_shift && r.d[i] >= 1 && r.d[i] < 30000;
    if (small_shift) {         // the row-blocked kernel: a source row fits its LDS slot, a shift fits the table's byte
        hipLaunchKernelGGL(aspp_im2col_rows_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, dlow, (__bf16*)g, B, H, W, K, r);
    } else {
        const long n = (long)B * H * W * (MI_ASPP_KPAD / 8);
        hipLaunchKernelGGL(aspp_im2col_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, dlow, (__bf16*)g, B, H, W, K, r);
    }
    MI_CHECK_LAUNCH("mi_aspp_im2col");
    return MI_OK;
}
