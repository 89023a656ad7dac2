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
constexpr int ROWB = 256;                       // one pixel row of a tile: 128 channels bf16, unpadded
constexpr int TILE_BYTES = KP * ROWB;           // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;      // 64 KiB -> 2 workgroups per CU

__device__ __attribute__((aligned(256))) uint32_t g_zero_page_tn[64];

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

__device__ __forceinline__ void glds16_tn(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Staging: tiles go global -> LDS directly (global_load_lds, 1 KiB = 4 pixel rows x 256 B per wave-instruction).
// The LDS image is lane-linear per DMA; 16-B chunk c of pixel row r sits at physical chunk c ^ ((r & 7) << 1), applied to
// the per-lane SOURCE address and to the transposed-read address.  With that XOR the 32 lanes of a
// ds_read_b64_tr_b16 half-wave (8 pixel rows x 32 B) hit 64 distinct banks.
// MODE 0: general (any stride)   1: unit stride, k x k taps   2: 1x1, stride 1 (no pixel coordinates needed at all)
template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradParams p) {
    constexpr bool UNIT_STRIDE = MODE >= 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware flattening: consecutive logical ids (= the output tiles of ONE K-split and tap, which all stream the same
    // pixel rows) land on the same XCD, so the rows are fetched into one L2 instead of eight.
    const int tiles = p.o_tiles * p.i_tiles;
    const int logical = mi_xcd_remap(blockIdx.x, tiles * p.T * p.S);
    const int tile = logical % tiles, rest = logical / tiles;
    const int t = rest % p.T, split = rest / p.T;
    const int ot = tile / p.i_tiles, it = tile - ot * p.i_tiles;
    const int o0 = ot * TO, i0 = it * TI;
    const int ky = t / p.ksz, kx = t - ky * p.ksz;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nk = (m_end - m_begin + KP - 1) / KP;
    const char* zero = reinterpret_cast<const char*>(g_zero_page_tn);

    // DMA assignment: wave w moves pieces 4w..4w+3 (4 rows each) of both tiles; lane -> (row, physical chunk)
    const int prow = lane >> 4, pch = lane & 15;
    const int HoWo = p.Ho * p.Wo;
    const int dys = ky * p.dil - p.pad, dxs = kx * p.dil - p.pad;
    int r_m[4], r_b[4], r_ho[4], r_wo[4], r_lch[4];
    const char* y_ptr[4];
    const char* x_ptr[4];
    bool y_col[4], x_col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 4 + prow;
        const int m = m_begin + row;
        r_m[j] = m;
        r_b[j] = m / HoWo;
        const int rem = m - r_b[j] * HoWo;
        r_ho[j] = rem / p.Wo;
        r_wo[j] = rem - r_ho[j] * p.Wo;
        r_lch[j] = pch ^ ((row & 7) << 1);            // logical chunk this lane fetches
        const int oc = o0 + r_lch[j] * 8, ic = i0 + r_lch[j] * 8;
        y_col[j] = oc < p.O;
        x_col[j] = ic < p.I;
        // with stride 1 the source pixel of tap (ky,kx) is m + dys*Wa + dxs: linear in m, only its validity is not
        y_ptr[j] = reinterpret_cast<const char*>(p.dY + (long)m * p.O + oc);
        x_ptr[j] = reinterpret_cast<const char*>(p.X + ((long)m + (long)dys * p.Wa + dxs) * p.I + ic);
    }
    const int step_q = KP / p.Wo, step_r = KP - step_q * p.Wo;
    const long y_step = (long)KP * p.O * 2, x_step = (long)KP * p.I * 2;

    auto stage = [&](int buf) {
        char* sy = smem + buf * STAGE_BYTES + wave * 4096;
        char* sx = sy + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool row_ok = r_m[j] < m_end;
            const char* gy = (row_ok && y_col[j]) ? y_ptr[j] : zero;
            const char* gx = zero;
            if (MODE == 2) {
                gx = (row_ok && x_col[j]) ? x_ptr[j] : zero;
            } else if (MODE == 1) {
                const int hs = r_ho[j] + dys, ws = r_wo[j] + dxs;
                if (row_ok && x_col[j] && (unsigned)hs < (unsigned)p.Ha && (unsigned)ws < (unsigned)p.Wa) gx = x_ptr[j];
            } else {
                const int ha = r_ho[j] * p.stride + dys, wa = r_wo[j] * p.stride + dxs;
                const bool ok = row_ok && x_col[j] && (unsigned)ha < (unsigned)p.Ha && (unsigned)wa < (unsigned)p.Wa;
                const long xoff = ((long)(r_b[j] * p.Ha + (ok ? ha : 0)) * p.Wa + (ok ? wa : 0)) * p.I + i0 + r_lch[j] * 8;
                if (ok) gx = reinterpret_cast<const char*>(p.X + xoff);
            }
            glds16_tn(gy, sy + j * 1024);
            glds16_tn(gx, sx + j * 1024);
            // advance this row by KP pixels
            r_m[j] += KP;
            y_ptr[j] += y_step;
            x_ptr[j] += x_step;
            if (MODE != 2) {
                r_wo[j] += step_r;
                r_ho[j] += step_q;
                if (r_wo[j] >= p.Wo) {
                    r_wo[j] -= p.Wo;
                    ++r_ho[j];
                }
                while (r_ho[j] >= p.Ho) {
                    r_ho[j] -= p.Ho;
                    ++r_b[j];
                }
            }
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
    // 4 columns (8 B) starting at column 4*(lane&3) of the 16-column subtile; same pixel<->k mapping on both operands.
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int rsw = (((g & 1) * 4 + q) << 1);                 // ((row & 7) << 1): ks*32 and half*16 are multiples of 8
    const int row_off = (g * 4 + q) * ROWB + (pc & 1) * 8;
    auto compute = [&](int buf) {
        const char* sy = smem + buf * STAGE_BYTES;
        const char* sx = sy + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[4], yf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int ch = ((wi * 8 + a * 2 + (pc >> 1)) ^ rsw) << 4;
                const char* base = sx + row_off + (ks * 32) * ROWB + ch;
                const s16x4 lo = tr_read(base), hi = tr_read(base + 16 * ROWB);
                union { bf16x8 v; s16x4 h[2]; } u;
                u.h[0] = lo;
                u.h[1] = hi;
                xf[a] = u.v;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ch = ((wo_ * 8 + b * 2 + (pc >> 1)) ^ rsw) << 4;
                const char* base = sy + row_off + (ks * 32) * ROWB + ch;
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
        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// dw[o][i][t] = scale[o] * sum_s slab[s][t][o][i]   (fixed summation order -> reproducible).
// One workgroup per (o, 64 consecutive i): threads (t-major) read slab rows coalesced over i, the T x 64 block is
// transposed through LDS and written out as 64*T contiguous floats (OIHW keeps t fastest).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           const float* __restrict__ scale, int S, int T, int O, int I,
                                                           int accumulate, int out_map, int ncls) {
    __shared__ float tile[9 * 64];
    const int IB = (T == 1) ? 256 : 64;            // i-columns per workgroup: T * IB <= 576 items
    const int o = blockIdx.y, ib = blockIdx.x * IB;
    const long plane = (long)O * I;
    const float sc = scale ? scale[o] : 1.f;
    const int ni = min(IB, I - ib);
    for (int item = threadIdx.x; item < T * IB; item += 256) {
        const int t = item / IB, ii = item - t * IB;
        float s = 0.f;
        if (ii < ni) {
            const float* src = slab + (long)t * plane + (long)o * I + ib + ii;
            for (int k = 0; k < S; ++k) s += src[(long)k * T * plane];
        }
        tile[ii * T + t] = s * sc;
    }
    __syncthreads();
    if (out_map == 0) {
        float* dst = dw + ((long)o * I + ib) * T;
        for (int e = threadIdx.x; e < ni * T; e += 256) dst[e] = accumulate ? dst[e] + tile[e] : tile[e];
    } else {  // ASPP: o = (r*9+tap)*ncls + cls  ->  dw[r][cls][i][tap]   (T == 1)
        const int grp = o / ncls, cls = o - grp * ncls;
        const int r = grp / 9, tap = grp - r * 9;
        for (int e = threadIdx.x; e < ni; e += 256) {
            const long d = (((long)r * ncls + cls) * I + ib + e) * 9 + tap;
            dw[d] = accumulate ? dw[d] + tile[e] : tile[e];
        }
    }
}

int pick_splits(long M, int tiles) {
    // 512 workgroup slots (256 CUs x 2 resident).  Minimise rounds x (K-steps per split + ~6 steps of fixed cost per
    // workgroup: prologue, pipeline fill, slab write); never fewer than 8 K-steps per split.
    const long steps = (M + KP - 1) / KP;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 64; ++s) {
        const long per = (steps + s - 1) / s;
        if (s > 1 && per < 8) break;
        const long blocks = (long)tiles * s;
        const long rounds = (blocks + 511) / 512;
        const double cost = (double)rounds * (double)(per + 6);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
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
                             int ksize, int stride, int pad, int dil, const float* scale_o, int accumulate, int out_map, int ncls,
                             size_t dw_elems, void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(dy && x && dw && workspace, "mi_conv_wgrad: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0, "mi_conv_wgrad: non-positive dimension");
    MI_REQUIRE(O % 8 == 0 && I % 8 == 0, "mi_conv_wgrad: O=%d, I=%d must be multiples of 8", O, I);
    MI_REQUIRE(ksize == 1 || ksize == 3, "mi_conv_wgrad: ksize");
    MI_REQUIRE(out_map == 0 || (out_map == 1 && ksize == 1), "mi_conv_wgrad: out_map 1 needs ksize 1");
    // the reducer scatters into dw: bound it here (out_map 1 writes 4 stacked [ncls][I][3][3] tensors)
    if (out_map == 1) {
        MI_REQUIRE(ncls > 0 && 36 * ncls <= MI_ASPP_KPAD && O >= 36 * ncls, "mi_conv_wgrad: out_map 1 needs 0 < 36*ncls=%d <= min(O=%d, %d)", 36 * ncls, O, MI_ASPP_KPAD);
        MI_REQUIRE(dw_elems >= (size_t)36 * ncls * I, "mi_conv_wgrad: dw holds %zu floats, the ASPP gradient needs %zu", dw_elems, (size_t)36 * ncls * I);
    } else {
        MI_REQUIRE(dw_elems >= (size_t)O * I * ksize * ksize, "mi_conv_wgrad: dw holds %zu floats, the gradient needs %zu", dw_elems, (size_t)O * I * ksize * ksize);
    }
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
    static std::atomic<uint64_t> attr_set[3];
    mi_allow_dynamic_lds((const void*)wgrad_tn_kernel<0>, LDS_BYTES, attr_set[0]);
    mi_allow_dynamic_lds((const void*)wgrad_tn_kernel<1>, LDS_BYTES, attr_set[1]);
    mi_allow_dynamic_lds((const void*)wgrad_tn_kernel<2>, LDS_BYTES, attr_set[2]);
    const unsigned nblocks = (unsigned)(p.o_tiles * p.i_tiles * p.T * p.S);
    const bool unit = stride == 1 && Ha == Ho && Wa == Wo;
    if (unit && ksize == 1 && pad == 0)
        hipLaunchKernelGGL(wgrad_tn_kernel<2>, dim3(nblocks), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    else if (unit)
        hipLaunchKernelGGL(wgrad_tn_kernel<1>, dim3(nblocks), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(wgrad_tn_kernel<0>, dim3(nblocks), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_wgrad");
    int o_real = O;
    if (out_map == 1) o_real = 36 * ncls;
    else ncls = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((I + (p.T == 1 ? 255 : 63)) / (p.T == 1 ? 256 : 64)), (unsigned)o_real), dim3(256), 0, (hipStream_t)stream, p.slab, dw,
                       scale_o, p.S, p.T, O, I, accumulate, out_map, ncls);
    MI_CHECK_LAUNCH("mi_conv_wgrad reduce");
    return MI_OK;
}
