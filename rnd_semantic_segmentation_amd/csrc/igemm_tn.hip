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
#include <stdlib.h>

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

// Transposed LDS read as INLINE ASM on a raw LDS byte address.  Through the builtin (or any load the compiler can see) hipcc
// orders every LDS read behind ALL pending global_load_lds of the wave - it cannot prove that the DMA in flight targets the
// other buffer - and emits s_waitcnt vmcnt(0) in front of the reads: the next tile's DMA was drained before the current
// tile was even read, i.e. no DMA / MFMA overlap inside a workgroup (found in the ISA of both kernels of this file, round 2).
// The asm form is invisible to that pass; the callers order reads against DMA themselves (counted vmcnt + barrier) and wait
// with an explicit s_waitcnt lgkmcnt before the MFMAs (+ sched_barrier: cdna_hip_programming.md 5.4 rule 18).
template <int OFF>
__device__ __forceinline__ s16x4 tr_read_lds(unsigned lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
    return v;
}

__device__ __forceinline__ unsigned lds_address(const char* generic_ptr_into_lds) {
    return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)generic_ptr_into_lds);
}

__device__ __forceinline__ void glds16_tn(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Buffer form: per-lane 32-bit byte offset against a buffer resource; an offset at or past num_records (a row past the end of
// this K-split, or bit 31 set for a channel chunk past O / I) writes zeros - no per-piece selects or 64-bit pointer updates.
__device__ __forceinline__ void blds16_tn(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, 0, 0, 0);
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
    // MODE 2 (1x1, stride 1, no padding: source pixel = output pixel): one buffer resource per operand that covers exactly this
    // split's pixel rows, so the rows past m_end read as zeros by range checking; one 32-bit add per piece and K step
    unsigned vy[4], vx[4];
    __amdgpu_buffer_rsrc_t rs_y, rs_x;
    if (MODE == 2) {
        const long rows = m_end > m_begin ? m_end - m_begin : 0;
        rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.dY + (long)m_begin * p.O), 0, (int)(rows * p.O * 2), 0x00020000);
        rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.X + (long)m_begin * p.I), 0, (int)(rows * p.I * 2), 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 4 + prow;
            vy[j] = (unsigned)(((long)row * p.O + o0 + r_lch[j] * 8) * 2) | (y_col[j] ? 0u : 0x80000000u);
            vx[j] = (unsigned)(((long)row * p.I + i0 + r_lch[j] * 8) * 2) | (x_col[j] ? 0u : 0x80000000u);
        }
    }

    auto stage = [&](int buf) {
        char* sy = smem + buf * STAGE_BYTES + wave * 4096;
        char* sx = sy + TILE_BYTES;
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                blds16_tn(rs_y, vy[j], sy + j * 1024);
                blds16_tn(rs_x, vx[j], sx + j * 1024);
                vy[j] += (unsigned)y_step;
                vx[j] += (unsigned)x_step;
            }
            return;
        }
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
    const unsigned lds0 = lds_address(smem);
    auto compute = [&](int buf) {
        const unsigned sy = lds0 + buf * STAGE_BYTES + row_off;
        const unsigned sx = sy + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { bf16x8 v; s16x4 h[2]; } xf[4], yf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const unsigned base = sx + (ks * 32) * ROWB + (((wi * 8 + a * 2 + (pc >> 1)) ^ rsw) << 4);
                xf[a].h[0] = tr_read_lds<0>(base);
                xf[a].h[1] = tr_read_lds<16 * ROWB>(base);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned base = sy + (ks * 32) * ROWB + (((wo_ * 8 + b * 2 + (pc >> 1)) ^ rsw) << 4);
                yf[b].h[0] = tr_read_lds<0>(base);
                yf[b].h[1] = tr_read_lds<16 * ROWB>(base);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a].v, yf[b].v, acc[a][b], 0, 0, 0);
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

// ---- 1x1 / stride 1 weight gradient as a deeper stream: 32-pixel stages, 4-slot ring, three stages in flight ------------------
// wgrad_tn_kernel<2> keeps ONE 64-pixel stage in flight per workgroup and drains it (vmcnt(0) + barrier) every step.  Same tile
// (128 x 128, four waves, two workgroups per CU) and the same LDS image per 32 pixel rows here, but stages of 32 pixels (16 KiB)
// through a 4-slot ring with three stages in flight (counted vmcnt, one barrier per stage): 96 KiB in flight per CU.
// Buffer addressing as in wgrad_tn_kernel<2>: the rows past the split end (and the dummy stages past the last step, which
// keep the vmcnt arithmetic uniform) are out of range for the resource and cost no memory traffic.
// Measured (B = 8, 97 x 97, 256 <-> 1024): equal with Infinity-Cache-warm operands (50-51 us incl. reducer), 5 % faster with cold
// ones (94 vs 100 us, `COLD=1 tools/wgexp.py`; a 5-slot ring with four stages in flight: 91 vs 95, no better), the training step
// +1.5 % (the 70 launches of this class: 6.57 -> 6.27 ms).  The class stays far from the HBM read rate (2.8 TB/s in the step):
// more bytes in flight are not what it lacks.
constexpr int KP4 = 32, SLOT4 = 2 * KP4 * ROWB, NSLOT4 = 4, LDS4_BYTES = NSLOT4 * SLOT4;      // 16 KiB x 4 = 64 KiB

__global__ __launch_bounds__(256, 2) void wgrad_s4_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.o_tiles * p.i_tiles;
    const int logical = mi_xcd_remap(blockIdx.x, tiles * p.S);
    const int tile = logical % tiles, split = logical / tiles;
    const int ot = tile / p.i_tiles, it = tile - ot * p.i_tiles;
    const int o0 = ot * TO, i0 = it * TI;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nk = (m_end - m_begin + KP4 - 1) / KP4;

    // DMA: a stage is 8 + 8 pieces of 4 pixel rows x 256 B; wave w moves pieces 2w, 2w+1 of both tiles
    const int prow = lane >> 4, pch = lane & 15;
    const long rows = m_end > m_begin ? m_end - m_begin : 0;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.dY + (long)m_begin * p.O), 0, (int)(rows * p.O * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.X + (long)m_begin * p.I), 0, (int)(rows * p.I * 2), 0x00020000);
    unsigned vy[2], vx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 4 + prow;
        const int lch = pch ^ ((row & 7) << 1);                // logical chunk this lane fetches (swizzle of wgrad_tn_kernel)
        const int oc = o0 + lch * 8, ic = i0 + lch * 8;
        vy[j] = (unsigned)(((long)row * p.O + oc) * 2) | (oc < p.O ? 0u : 0x80000000u);
        vx[j] = (unsigned)(((long)row * p.I + ic) * 2) | (ic < p.I ? 0u : 0x80000000u);
    }
    const unsigned y_step = (unsigned)KP4 * p.O * 2, x_step = (unsigned)KP4 * p.I * 2;
    auto stage = [&](int slot) {
        char* sy = smem + slot * SLOT4 + wave * 2048;
        char* sx = sy + KP4 * ROWB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            blds16_tn(rs_y, vy[j], sy + j * 1024);
            blds16_tn(rs_x, vx[j], sx + j * 1024);
            vy[j] += y_step;
            vx[j] += x_step;
        }
    };

    const int wi = wave & 1, wo_ = wave >> 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int rsw = (((g & 1) * 4 + q) << 1);
    const int row_off = (g * 4 + q) * ROWB + (pc & 1) * 8;
    const unsigned lds0 = lds_address(smem);
    auto compute = [&](int slot) {
        const unsigned sy = lds0 + slot * SLOT4 + row_off;
        const unsigned sx = sy + KP4 * ROWB;
        union { bf16x8 v; s16x4 h[2]; } xf[4], yf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const unsigned base = sx + (((wi * 8 + a * 2 + (pc >> 1)) ^ rsw) << 4);
            xf[a].h[0] = tr_read_lds<0>(base);
            xf[a].h[1] = tr_read_lds<16 * ROWB>(base);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned base = sy + (((wo_ * 8 + b * 2 + (pc >> 1)) ^ rsw) << 4);
            yf[b].h[0] = tr_read_lds<0>(base);
            yf[b].h[1] = tr_read_lds<16 * ROWB>(base);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a].v, yf[b].v, acc[a][b], 0, 0, 0);
    };

    // stages 0..2 in flight; step s: my pieces of stage s landed (8 younger instructions may be pending), barrier (everyone's have,
    // and everyone is done with stage s-1), stage s+3 into the slot stage s-1 has left, compute stage s
    stage(0);
    stage(1);
    stage(2);
    for (int s = 0; s < nk; ++s) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage((s + 3) & (NSLOT4 - 1));
        compute(s & (NSLOT4 - 1));
    }
    float* slab = p.slab + ((long)split * p.O) * p.I;
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

// ---- 128 (o) x 256 (i) tile, 32 pixels per K step --------------------------------------------------------------------------
// The 128 x 128 kernel above was read as bound by the bytes the L2 hands to the LDS (15.6 per kFLOP at ~43 GB/s per CU; the later
// microbenchmark tools/micro/l2lds.hip puts that path at 90-125 GB/s per CU for L2 hits and 28-30 for misses).  Widening the tile over the INPUT channels to 256 with a 32-pixel step keeps the MFMA
// work per step (32 per wave) and the two-workgroups-per-CU structure (48 KiB of LDS, 128 accumulator registers) and moves
// 24 KiB instead of 32 KiB per step: 11.4 bytes per kFLOP.  Used for stride-1 convs with I >= 256 (every layer3 / layer4 / ASPP
// weight gradient).  Same slab layout, same reducer.
constexpr int TI2 = 256, KP2 = 32;
constexpr int XROWB = 512;                                  // one pixel row of the x tile: 256 channels bf16
constexpr int Y2_BYTES = KP2 * ROWB, X2_BYTES = KP2 * XROWB;  // 8 KiB + 16 KiB
constexpr int STAGE2_BYTES = Y2_BYTES + X2_BYTES, NSTAGE2 = 3, LDS2_BYTES = NSTAGE2 * STAGE2_BYTES;   // 72 KiB: two workgroups per CU

// MODE 1: unit stride, k x k taps   2: 1x1, stride 1
template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_tn256_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.o_tiles * p.i_tiles;
    const int logical = mi_xcd_remap(blockIdx.x, tiles * p.T * p.S);
    const int tile = logical % tiles, rest = logical / tiles;
    const int t = rest % p.T, split = rest / p.T;
    const int ot = tile / p.i_tiles, it = tile - ot * p.i_tiles;
    const int o0 = ot * TO, i0 = it * TI2;
    const int ky = t / p.ksz, kx = t - ky * p.ksz;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nk = (m_end - m_begin + KP2 - 1) / KP2;
    const char* zero = reinterpret_cast<const char*>(g_zero_page_tn);

    // DMA: wave w moves dy pieces 2w, 2w+1 (4 rows x 256 B each) and x pieces 4w .. 4w+3 (2 rows x 512 B each)
    const int yrow_l = lane >> 4, ych = lane & 15, xrow_l = lane >> 5, xch = lane & 31;
    const int dys = ky * p.dil - p.pad, dxs = kx * p.dil - p.pad;
    int y_m[2], x_m[4], x_ho[4], x_wo[4];
    const char* y_ptr[2];
    const char* x_ptr[4];
    bool y_col[2], x_col[4];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 4 + yrow_l;
        const int lch = ych ^ ((row & 7) << 1);
        y_m[j] = m_begin + row;
        y_col[j] = o0 + lch * 8 < p.O;
        y_ptr[j] = reinterpret_cast<const char*>(p.dY + (long)y_m[j] * p.O + o0 + lch * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 2 + xrow_l;
        const int lch = xch ^ ((row & 7) << 1);
        const int m = m_begin + row;
        x_m[j] = m;
        const int b = m / HoWo, rem = m - b * HoWo;
        x_ho[j] = rem / p.Wo;
        x_wo[j] = rem - x_ho[j] * p.Wo;
        x_col[j] = i0 + lch * 8 < p.I;
        x_ptr[j] = reinterpret_cast<const char*>(p.X + ((long)m + (long)dys * p.Wa + dxs) * p.I + i0 + lch * 8);
    }
    const int step_q = KP2 / p.Wo, step_r = KP2 - step_q * p.Wo;
    const long y_step = (long)KP2 * p.O * 2, x_step = (long)KP2 * p.I * 2;

    auto stage = [&](int buf) {
        char* sy = smem + buf * STAGE2_BYTES + wave * 2048;
        char* sx = smem + buf * STAGE2_BYTES + Y2_BYTES + wave * 4096;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            glds16_tn((y_m[j] < m_end && y_col[j]) ? y_ptr[j] : zero, sy + j * 1024);
            y_m[j] += KP2;
            y_ptr[j] += y_step;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = x_m[j] < m_end && x_col[j];
            if (MODE == 1) {
                const int hs = x_ho[j] + dys, ws = x_wo[j] + dxs;
                ok = ok && (unsigned)hs < (unsigned)p.Ha && (unsigned)ws < (unsigned)p.Wa;
            }
            glds16_tn(ok ? x_ptr[j] : zero, sx + j * 1024);
            x_m[j] += KP2;
            x_ptr[j] += x_step;
            if (MODE == 1) {
                x_wo[j] += step_r;
                x_ho[j] += step_q;
                if (x_wo[j] >= p.Wo) {
                    x_wo[j] -= p.Wo;
                    ++x_ho[j];
                }
                while (x_ho[j] >= p.Ho) x_ho[j] -= p.Ho;
            }
        }
    };

    // MFMA orientation: D rows = i (A operand = X^T), D cols = o (B operand = dY^T); wave owns 128 (i) x 64 (o)
    const int wi = wave & 1, wo_ = wave >> 1;
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int rsw = (((g & 1) * 4 + q) << 1);
    const unsigned lds0 = lds_address(smem);
    const unsigned y_off = (g * 4 + q) * ROWB + (pc & 1) * 8, x_off = Y2_BYTES + (g * 4 + q) * XROWB + (pc & 1) * 8;
    auto compute = [&](int buf) {
        const unsigned sb = lds0 + buf * STAGE2_BYTES;
        union { bf16x8 v; s16x4 h[2]; } xf[8], yf[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned base = sb + y_off + (((wo_ * 8 + b * 2 + (pc >> 1)) ^ rsw) << 4);
            yf[b].h[0] = tr_read_lds<0>(base);
            yf[b].h[1] = tr_read_lds<16 * ROWB>(base);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const unsigned base = sb + x_off + (((wi * 16 + a * 2 + (pc >> 1)) ^ rsw) << 4);
            xf[a].h[0] = tr_read_lds<0>(base);
            xf[a].h[1] = tr_read_lds<16 * XROWB>(base);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a].v, yf[b].v, acc[a][b], 0, 0, 0);
    };

    // Three-stage ring, two stages in flight: a K step of this kernel is bound by the round trip of its DMA (the per-step
    // drain + barrier of the 128 x 128 kernel leaves the MFMA pipe idle ~70 % of a step), so stage kt+2 is issued before stage kt
    // is computed.  One raw barrier per step: it publishes stage kt (every wave waited for its own six pieces: counted vmcnt, one
    // younger stage stays in flight) and proves that every wave has finished reading stage kt-1, whose slot stage kt+2 then takes.
    if (nk > 0) {
        stage(0);
        if (nk > 1) stage(1);
        int slot = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage(slot == 0 ? 2 : slot - 1);          // slot of stage kt-1 == (kt+2) % 3
            compute(slot);
            slot = slot == 2 ? 0 : slot + 1;
        }
    }

    float* slab = p.slab + ((long)(split * p.T + t) * p.O) * p.I;
    const int fcol = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int o = o0 + wo_ * 64 + b * 16 + fcol;
        if (o >= p.O) continue;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int i = i0 + wi * 128 + a * 16 + fq * 4;
            if (i >= p.I) continue;
            *reinterpret_cast<f32x4*>(slab + (long)o * p.I + i) = acc[a][b];
        }
    }
}

int pick_splits256(long M, int tiles) {
    // as pick_splits, for 32-pixel steps: 512 slots, ~12 steps of fixed cost per workgroup, never fewer than 16 steps per split
    const long steps = (M + KP2 - 1) / KP2;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 128; ++s) {
        const long per = (steps + s - 1) / s;
        if (s > 1 && per < 16) break;
        const long rounds = ((long)tiles * s + 511) / 512;
        const double cost = (double)rounds * (double)(per + 12);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

// Splits of the deep-stream 1x1 kernel.  MI_WGRAD_S4_SLOTS (default 512 = two workgroups per CU): workgroup slots the cost model fills.
// With 256 (one workgroup per CU, twice the K steps, half the partial planes) the launches themselves are 7 % faster when timed alone
// (256 <-> 1024: 64.7 vs 70.0 us; the class 3.72 vs 4.02 ms single-stream), but the two-stream step is not (278.9 vs 280.0 images/s): a
// launch that fills every CU's slot leaves the data-gradient chain beside it nothing to overlap with.
int pick_splits_s4(long M, int tiles) {
    const int slots = mi_sw().wgrad_s4_slots;
    const long steps = (M + 31) / 32;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 128; ++s) {
        const long per = (steps + s - 1) / s;
        if (s > 1 && per < 16) break;
        const long rounds = ((long)tiles * s + slots - 1) / slots;
        const double cost = (double)rounds * (double)(per + 12);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

inline bool use_tn256(int O, int I, int ksize, int pad, int stride, int Ha, int Ho, int Wa, int Wo) {
    // MI_WGRAD_TI256: unset = by rule, 0 = never, 1 = every stride-1 conv with I >= 256 (tests).
    // Warm (operands in the Infinity Cache from the previous iteration of a timing loop) this kernel is SLOWER than the 128 x 128
    // ones (B = 8, 97 x 97: 1x1 256 <-> 1024 71 us vs 58 us, 3x3 256 144 vs 129, 3x3 512 397 vs 406).  In the training step, with
    // cold operands, the big 1x1 launches are HBM-latency-bound: the L2 -> LDS fill rate is (bytes in flight) / (HBM round trip),
    // equal for both kernels, and this tile turns it into 1.33x the unique bytes (dy is read once instead of twice).  In-step:
    // 512 -> 2048 172 vs 199 us, 2048 -> 512 162 vs 197, 1024 -> 2048 264 vs 374, 512 <-> 1024 98 vs 105; 256 <-> 1024 72 vs 70
    // (the slab traffic of the larger split count eats the gain there), hence the size rule.
    const int mode = mi_sw().wgrad_ti256;
    if (mode == 0 || I < 256 || stride != 1 || Ha != Ho || Wa != Wo) return false;
    if (mode > 0) return true;
    return ksize == 1 && pad == 0 && (long)O * I >= 512L * 1024;
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
            const long sstride = (long)T * plane;
            // eight partial planes per batch: the loads are issued together (this kernel lives on bytes in flight), the additions
            // keep the ascending-split order, so the sum is the same fixed-order sum as a plain loop
            int k = 0;
            for (; k + 8 <= S; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + u) * sstride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < S; ++k) s += src[(long)k * sstride];
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

// The reducers of several weight gradients in ONE launch (mi_conv_wgrad_reduce: the three convs of a bottleneck, round 5).  Alone each reducer is a 10-us
// launch of 33 MB that, on the weight-gradient stream beside the data-gradient chain, waits ~35 us for its turn on the CUs - three times per block, with the
// next weight gradient queued behind it.  The jobs travel in the kernel arguments; a workgroup finds its job by the prefix sums of their grids and then runs
// wgrad_reduce_kernel's body unchanged (same fixed order, same bits).
constexpr int MI_REDUCE_MAX_JOBS = 8;
struct WgradReduceJob {
    const float* slab;
    float* dw;
    const float* scale;
    int S, T, O, I, accumulate, out_map, ncls, gx, gy;       // gx * gy workgroups: (i blocks, output rows)
};
struct WgradReduceTable {
    WgradReduceJob job[MI_REDUCE_MAX_JOBS];
    int first[MI_REDUCE_MAX_JOBS + 1];
    int n;
};
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slab, float* __restrict__ dw, const float* __restrict__ scale, int S, int T, int O, int I,
                                                  int accumulate, int out_map, int ncls, int bx, int o, float* tile) {
    const int IB = (T == 1) ? 256 : 64;
    const int ib = bx * IB;
    const long plane = (long)O * I;
    const float sc = scale ? scale[o] : 1.f;
    const int ni = min(IB, I - ib);
    for (int item = threadIdx.x; item < T * IB; item += 256) {
        const int t = item / IB, ii = item - t * IB;
        float s = 0.f;
        if (ii < ni) {
            const float* src = slab + (long)t * plane + (long)o * I + ib + ii;
            const long sstride = (long)T * plane;
            int k = 0;
            for (; k + 8 <= S; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + u) * sstride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < S; ++k) s += src[(long)k * sstride];
        }
        tile[ii * T + t] = s * sc;
    }
    __syncthreads();
    if (out_map == 0) {
        float* dst = dw + ((long)o * I + ib) * T;
        for (int e = threadIdx.x; e < ni * T; e += 256) dst[e] = accumulate ? dst[e] + tile[e] : tile[e];
    } else {
        const int grp = o / ncls, cls = o - grp * ncls;
        const int r = grp / 9, tap = grp - r * 9;
        for (int e = threadIdx.x; e < ni; e += 256) {
            const long d = (((long)r * ncls + cls) * I + ib + e) * 9 + tap;
            dw[d] = accumulate ? dw[d] + tile[e] : tile[e];
        }
    }
}
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WgradReduceTable tab) {
    __shared__ float tile[9 * 64];
    int j = 0;
#pragma unroll
    for (int k = 1; k < MI_REDUCE_MAX_JOBS; ++k) j += (k < tab.n && (int)blockIdx.x >= tab.first[k]) ? 1 : 0;
    const WgradReduceJob q = tab.job[j];
    const int local = (int)blockIdx.x - tab.first[j];
    const int o = local / q.gx, bx = local - o * q.gx;
    wgrad_reduce_body(q.slab, q.dw, q.scale, q.S, q.T, q.O, q.I, q.accumulate, q.out_map, q.ncls, bx, o, tile);
}

// =====================================================================================================================
// 3x3 weight gradient with the three taps of a kernel row FUSED, wide-K ping-pong main loop (the dilated 3x3 family of
// layer3 / layer4: reference core/components/resnet.py:22-25,100).
//
// Why: one workgroup per tap re-streams dy and x nine times through the L2 (15.6 L2 bytes per kFLOP at 128 x 128 tiles),
// and the launch is bound by exactly that traffic.  The three taps (ky, 0..2) of one kernel row read x at pixel offsets
// -d, 0, +d of the SAME image row, so one LDS window of 64 + 2d pixel rows feeds three MFMA groups against one dy tile:
// 5.2 bytes per kFLOP.
// Validity without masks: the contraction index is a PADDED pixel coordinate q = (b*H + h) * (W + 2d) + wp, every image
// row carrying d zero slots on either side (staged from the zero page, no memory traffic).  For tap kx the x row paired
// with dy row q is window row q + kx*d - d - q0, i.e. padded column wp + (kx-1)d: inside the image it is the right source
// pixel, outside it is a pad slot = 0 - exactly the zero padding of the convolution.  Rows h + (ky-1)d outside the image are
// zero-paged when the window is staged.  Cost: 2d / W more (all-zero) K rows: 4 % at d = 2, W = 97.
// Main loop: as igemm_pp.hip - 8 waves in two groups that alternate between an MFMA segment (48 MFMAs per wave and 64-row
// slab) and a read segment (40 ds_read_b64_tr_b16 + this wave's DMA pieces of slab s+3), 4-slot LDS ring, counted vmcnt.
// Group 0 stages the dy rows, group 1 the x window.  Wave (wi = wave & 3, wo = group): 32 (i) x 64 (o) x 3 taps.
// Output: fp32 partial planes slab[split][ky*3+kx][o][i], summed in fixed order by wgrad_reduce_kernel (deterministic).
constexpr int P3_KS = 64, P3_XROWS = 80;                       // d <= 8
constexpr int P3_DY_BYTES = P3_KS * ROWB, P3_X_BYTES = P3_XROWS * ROWB, P3_SLAB = P3_DY_BYTES + P3_X_BYTES;   // 16 + 20 KiB
constexpr int P3_LDS = 4 * P3_SLAB;                            // 144 KiB: one workgroup per CU
constexpr int P3_NPY = 4, P3_NPX = 5;                          // DMA pieces (4 rows x 256 B) per wave and slab: dy (group 0), x (group 1)

struct WgradP3Params {
    const __bf16* dY;
    const __bf16* X;
    float* slab;
    int O, I, H, W, d, WP, BH;
    long Q;
    int S, slabs_per_split;
    int o_tiles, i_tiles;
    int dbg;             // experiment toggles (MI_P3_DBG): 1 no DMA in the loop, 2 no LDS reads, 4 no MFMAs, 8 no partial-plane stores
};

#ifdef MI_EXPERIMENTS   // opt-in kernel that does not win: built by tools/experiments/build.sh only, not part of libmi355seg.so
__global__ __launch_bounds__(512, 1) void wgrad_p3_kernel(WgradP3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int tiles = p.o_tiles * p.i_tiles;
    const int logical = mi_xcd_remap(blockIdx.x, tiles * 3 * p.S);
    const int tile = logical % tiles, rest = logical / tiles;
    const int ky = rest % 3, split = rest / 3;
    const int ot = tile / p.i_tiles, it = tile - ot * p.i_tiles;
    const int o0 = ot * TO, i0 = it * TI;
    const long q_begin = (p.dbg & 16) ? 0 : (long)split * p.slabs_per_split * P3_KS;      // dbg 16: every split streams the same (L2-hot) rows
    long left = (p.Q - q_begin + P3_KS - 1) / P3_KS;
    const int ns = (int)(left < 0 ? 0 : (left < p.slabs_per_split ? left : p.slabs_per_split));
    // ---- DMA roles.  lane -> (row within the piece, physical 16-B chunk); it fetches logical chunk physical ^ ((row & 7) << 1).
    const int prow = lane >> 4, pch = lane & 15;
    // zero rows (pad slots, rows outside the image, tile tails): 16 lanes read the 256-B zero page as one contiguous row
    const char* zero = reinterpret_cast<const char*>(g_zero_page_tn) + pch * 16;
    // A wave stages a SPAN of consecutive padded coordinates per slab (16 dy rows or 20 window rows): piece j = rows 4j .. 4j+3.
    // The span's first row is wave-uniform, so its coordinates (q, padded column wp, image row h, pixel index pix) live in
    // scalar registers and advance with scalar arithmetic: 64 coordinates = n0 whole image rows + r0 columns (+ one carry).
    // Fast path (3 of 4 slabs at W = 97): the whole span lies inside the real columns of one valid image row -> the rows are
    // consecutive pixels, address = scalar base + a per-lane constant (2 vector ops per piece).  Otherwise every lane derives
    // its row's coordinates from the scalars (one carry at most: the plan requires span <= WP).
    constexpr int NPMAX = P3_NPX;
    const int np = grp == 0 ? P3_NPY : P3_NPX;
    const int span = np * 4;
    const int dh = grp == 0 ? 0 : (ky - 1) * p.d;
    const int n0 = P3_KS / p.WP, r0 = P3_KS - n0 * p.WP;
    const int pix_step = n0 * p.W + r0;                 // minus 2d when the column wraps into the next image row
    const int span0 = grp == 0 ? wq * (P3_NPY * 4) : wq * (P3_NPX * 4);
    int s_q = (int)(q_begin + span0 - (grp == 0 ? 0 : p.d));                 // may be < 0 (first window rows of split 0)
    int s_wp, s_h, s_pix;
    {
        const long qs = (long)s_q + p.WP;                                     // >= 0 (d <= WP)
        const int bh = (int)(qs / p.WP) - 1;
        s_wp = (int)(qs % p.WP);
        s_h = bh < 0 ? p.H - 1 : bh % p.H;                                    // row -1 precedes row 0 of image 0
        s_pix = (bh + dh) * p.W + s_wp - p.d;
    }
    // the lane's channel chunk depends on (row & 7), i.e. on the parity of the piece: two per-lane constants
    const __bf16* opnd = grp == 0 ? p.dY : p.X;
    const int c0 = grp == 0 ? o0 : i0, cmax = grp == 0 ? p.O : p.I;
    const unsigned row_bytes = (unsigned)cmax * 2u;
    const int lch_e = pch ^ (((span0 + prow) & 7) << 1), lch_o = pch ^ (((span0 + prow + 4) & 7) << 1);
    const bool ok_e = c0 + lch_e * 8 < cmax, ok_o = c0 + lch_o * 8 < cmax;
    const unsigned off_e = prow * row_bytes + (c0 + lch_e * 8) * 2, off_o = prow * row_bytes + (c0 + lch_o * 8) * 2;
    const char* opnd_b = reinterpret_cast<const char*>(opnd);
    const int Qi = (int)p.Q;
    int ld_s = 0;
    auto stage_next = [&]() {
        char* dst = smem + (ld_s & 3) * P3_SLAB + (grp == 0 ? wq * (P3_NPY * 1024) : P3_DY_BYTES + wq * (P3_NPX * 1024));
        const bool fast = s_q >= 0 && s_q + span <= Qi && s_wp >= p.d && s_wp + span <= p.W + p.d && (unsigned)(s_h + dh) < (unsigned)p.H;
        if (fast) {
            const char* P = opnd_b + (unsigned long)(unsigned)s_pix * row_bytes;              // scalar
#pragma unroll
            for (int j = 0; j < NPMAX; ++j)
                if (j < np) {
                    const char* src = P + (j * 4 * row_bytes + ((j & 1) ? off_o : off_e));
                    glds16_tn(((j & 1) ? ok_o : ok_e) ? src : zero, dst + j * 1024);
                }
        } else {
#pragma unroll
            for (int j = 0; j < NPMAX; ++j)
                if (j < np) {
                    const int r = j * 4 + prow;
                    int wp = s_wp + r;
                    const bool c = wp >= p.WP;
                    wp -= c ? p.WP : 0;
                    int h = s_h + (c ? 1 : 0);
                    h = h >= p.H ? h - p.H : h;
                    const int pix = s_pix + r - (c ? 2 * p.d : 0);
                    const bool ok = ((j & 1) ? ok_o : ok_e) && (unsigned)(s_q + r) < (unsigned)Qi && (unsigned)(wp - p.d) < (unsigned)p.W &&
                                    (unsigned)(h + dh) < (unsigned)p.H;
                    const char* src = opnd_b + (unsigned long)(unsigned)(ok ? pix : 0) * row_bytes + (((j & 1) ? off_o : off_e) - prow * row_bytes);
                    glds16_tn(ok ? src : zero, dst + j * 1024);
                }
        }
        // the span's first row in the next slab (scalar)
        s_q += P3_KS;
        s_wp += r0;
        const int c = s_wp >= p.WP ? 1 : 0;
        s_wp -= c ? p.WP : 0;
        s_h += n0 + c;
        s_h = s_h >= p.H ? s_h - p.H : s_h;
        s_pix += pix_step - (c ? 2 * p.d : 0);
        ++ld_s;
    };

    // ---- compute roles: wave (wi, wo) owns i in [wi*32, +32) x o in [wo*64, +64) for the three taps kx
    const int wi = wq, wo = grp;
    f32x4 acc[3][2][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[kx][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pc = lane & 3;
    const int rk0 = g * 4 + q4;
    const int y_off = rk0 * ROWB + (pc & 1) * 8, y_sw = (rk0 & 7) << 1;
    int x_off[3], x_sw[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int rk = rk0 + kx * p.d;
        x_off[kx] = P3_DY_BYTES + rk * ROWB + (pc & 1) * 8;
        x_sw[kx] = (rk & 7) << 1;
    }
    union Frag { bf16x8 v; s16x4 h[2]; };
    Frag yf[2][4], xf[2][3][2];
    const unsigned lds0 = lds_address(smem);
    auto read_frags = [&](int s) {
        unsigned sb = lds0 + (s & 3) * P3_SLAB;
        asm volatile("" : "+v"(sb));          // opaque: keeps the 10 fragment offsets, not 40 precomputed per-slot addresses, in registers
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned base = sb + y_off + ((((wo * 4 + b) * 2 + (pc >> 1)) ^ y_sw) << 4);
            yf[0][b].h[0] = tr_read_lds<0>(base);
            yf[0][b].h[1] = tr_read_lds<16 * ROWB>(base);
            yf[1][b].h[0] = tr_read_lds<32 * ROWB>(base);
            yf[1][b].h[1] = tr_read_lds<48 * ROWB>(base);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const unsigned base = sb + x_off[kx] + ((((wi * 2 + a) * 2 + (pc >> 1)) ^ x_sw[kx]) << 4);
                xf[0][kx][a].h[0] = tr_read_lds<0>(base);
                xf[0][kx][a].h[1] = tr_read_lds<16 * ROWB>(base);
                xf[1][kx][a].h[0] = tr_read_lds<32 * ROWB>(base);
                xf[1][kx][a].h[1] = tr_read_lds<48 * ROWB>(base);
            }
    };
    auto mfma_half = [&](int ks) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[kx][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks][kx][a].v, yf[ks][b].v, acc[kx][a][b], 0, 0, 0);
    };

    // ---- main loop (interval scheme of igemm_pp.hip) ----------------------------------------------------------------------------
    for (int s = 0; s < 3 && s < ns; ++s) stage_next();
    if (ns >= 3) {
        if (grp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P3_NPY) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P3_NPX) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    const int dbg = p.dbg;
    for (int s = 0; s < ns; ++s) {
        if (s + 3 < ns && !(dbg & 1)) stage_next();
        if (!(dbg & 2) || s == 0) read_frags(s);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) {
            if (s + 3 < ns) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P3_NPX) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
        if (!(dbg & 4)) {
            mfma_half(0);
            mfma_half(1);
        }
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) {
            if (s + 3 < ns) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P3_NPY) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---- partial planes: slab[split][ky*3+kx][o][i], i fastest; lane owns o = column, i .. i+3 = rows
    if (dbg & 8) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(acc[kx][a][b]));
        return;
    }
    const int fcol = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        float* plane = p.slab + ((long)(split * 9 + ky * 3 + kx) * p.O) * p.I;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int o = o0 + wo * 64 + b * 16 + fcol;
            if (o >= p.O) continue;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int i = i0 + wi * 32 + a * 16 + fq * 4;
                if (i >= p.I) continue;
                *reinterpret_cast<f32x4*>(plane + (long)o * p.I + i) = acc[kx][a][b];
            }
        }
    }
}
#endif  // MI_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------------------------
// wgrad_q3_kernel: the fused-row idea in the structure that the per-tap kernel has proven - 4 waves, TWO workgroups per CU that
// cover each other's DMA waits, double-buffered 64-coordinate slabs with a drain + barrier per step - and a tile that keeps the
// partial planes at 96 KiB per workgroup so that 504 workgroups (two per CU) still cost the same 49.5 MB of slabs:
// 64 (o) x 128 (i) x the three taps of a kernel row.  Per step a workgroup moves 8 KiB of dy + 17.4 KiB of x window for
// 3.1 MFLOP: 8.1 L2 bytes per kFLOP (15.6 per tap, 5.2 in wgrad_p3_kernel) and 48 MFMAs per wave and DMA round trip instead of 32,
// with a third of the steps.  Padded contraction coordinate, scalar span addressing and x-window reads as in wgrad_p3_kernel.
// dy rows are 128 B (64 channels): DMA pieces of 8 rows, 16-B chunk c of row r at chunk c ^ (((r >> 1) & 3) << 1), which makes
// the 8 rows x 32 B of a transposed-read half-wave cover all 64 banks.
constexpr int Q3_TO = 64;
constexpr int Q3_YROWB = 128;
constexpr int Q3_DY_BYTES = P3_KS * Q3_YROWB, Q3_STAGE = Q3_DY_BYTES + P3_X_BYTES;      // 8 KiB + 20 KiB
constexpr int Q3_LDS = 2 * Q3_STAGE;                                                    // 56 KiB: two workgroups per CU

#ifdef MI_EXPERIMENTS      // phase toggles of MI_P3_DBG (tools/wgexp.py toggles): experiment builds only, the constant 0 in the product library
#define Q3_DBG(b) (p.dbg & (b))
#else
#define Q3_DBG(b) 0
#endif
__global__ __launch_bounds__(256, 2) void wgrad_q3_kernel(WgradP3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.o_tiles * p.i_tiles;
    const int logical = mi_xcd_remap(blockIdx.x, tiles * 3 * p.S);
    const int tile = logical % tiles, rest = logical / tiles;
    const int ky = rest % 3, split = rest / 3;
    const int ot = tile / p.i_tiles, it = tile - ot * p.i_tiles;
    const int o0 = ot * Q3_TO, i0 = it * TI;
    const long q_begin = (long)split * p.slabs_per_split * P3_KS;
    long left = (p.Q - q_begin + P3_KS - 1) / P3_KS;
    const int ns = (int)(left < 0 ? 0 : (left < p.slabs_per_split ? left : p.slabs_per_split));
    const int Qi = (int)p.Q;
    const int n0 = P3_KS / p.WP, r0 = P3_KS - n0 * p.WP;
    const int pix_step = n0 * p.W + r0;

    // ---- dy span: rows 16 wq .. 16 wq + 15 (two pieces of 8 rows x 128 B) ------------------------------------------------------
    const int yprow = lane >> 3, ypch = lane & 7;
    const char* yzero = reinterpret_cast<const char*>(g_zero_page_tn) + ypch * 16;
    const int yspan0 = wq * 16;
    const int ylch = ypch ^ ((((yspan0 + yprow) >> 1) & 3) << 1);
    const bool yok = o0 + ylch * 8 < p.O;
    const unsigned yrow_bytes = (unsigned)p.O * 2u;
    const unsigned yoff = yprow * yrow_bytes + (o0 + ylch * 8) * 2;
    const char* dy_b = reinterpret_cast<const char*>(p.dY);
    int y_q = (int)(q_begin + yspan0), y_wp, y_pix;
    {
        const int bh = y_q / p.WP;
        y_wp = y_q - bh * p.WP;
        y_pix = bh * p.W + y_wp - p.d;
    }
    // ---- x span: window rows 20 wq .. 20 wq + 19 (five pieces of 4 rows x 256 B) ---------------------------------------------------
    const int xprow = lane >> 4, xpch = lane & 15;
    const char* xzero = reinterpret_cast<const char*>(g_zero_page_tn) + xpch * 16;
    const int dh = (ky - 1) * p.d;
    const int xspan0 = wq * 20;
    const int xlch_e = xpch ^ (((xspan0 + xprow) & 7) << 1), xlch_o = xpch ^ (((xspan0 + xprow + 4) & 7) << 1);
    const bool xok_e = i0 + xlch_e * 8 < p.I, xok_o = i0 + xlch_o * 8 < p.I;
    const unsigned xrow_bytes = (unsigned)p.I * 2u;
    const unsigned xoff_e = xprow * xrow_bytes + (i0 + xlch_e * 8) * 2, xoff_o = xprow * xrow_bytes + (i0 + xlch_o * 8) * 2;
    const char* x_b = reinterpret_cast<const char*>(p.X);
    int x_q = (int)(q_begin + xspan0 - p.d), x_wp, x_h, x_pix;
    {
        const long qs = (long)x_q + p.WP;
        const int bh = (int)(qs / p.WP) - 1;
        x_wp = (int)(qs % p.WP);
        x_h = bh < 0 ? p.H - 1 : bh % p.H;
        x_pix = (bh + dh) * p.W + x_wp - p.d;
    }

    auto stage = [&](int buf) {
        char* ydst = smem + buf * Q3_STAGE + wq * 2048;
        char* xdst = smem + buf * Q3_STAGE + Q3_DY_BYTES + wq * 5120;
        // dy: real iff the padded column is an image column (the image row always exists for dy)
        if (y_q + 16 <= Qi && y_wp >= p.d && y_wp + 16 <= p.W + p.d) {
            const char* P = dy_b + (unsigned long)(unsigned)y_pix * yrow_bytes;
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16_tn(yok ? P + (j * 8 * yrow_bytes + yoff) : yzero, ydst + j * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = j * 8 + yprow;
                int wp = y_wp + r;
                const bool c = wp >= p.WP;
                wp -= c ? p.WP : 0;
                const int pix = y_pix + r - (c ? 2 * p.d : 0);
                const bool ok = yok && (unsigned)(y_q + r) < (unsigned)Qi && (unsigned)(wp - p.d) < (unsigned)p.W;
                glds16_tn(ok ? dy_b + (unsigned long)(unsigned)pix * yrow_bytes + (yoff - yprow * yrow_bytes) : yzero, ydst + j * 1024);
            }
        }
        if (x_q >= 0 && x_q + 20 <= Qi && x_wp >= p.d && x_wp + 20 <= p.W + p.d && (unsigned)(x_h + dh) < (unsigned)p.H) {
            const char* P = x_b + (unsigned long)(unsigned)x_pix * xrow_bytes;
#pragma unroll
            for (int j = 0; j < 5; ++j) glds16_tn(((j & 1) ? xok_o : xok_e) ? P + (j * 4 * xrow_bytes + ((j & 1) ? xoff_o : xoff_e)) : xzero, xdst + j * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int r = j * 4 + xprow;
                int wp = x_wp + r;
                const bool c = wp >= p.WP;
                wp -= c ? p.WP : 0;
                int h = x_h + (c ? 1 : 0);
                h = h >= p.H ? h - p.H : h;
                const int pix = x_pix + r - (c ? 2 * p.d : 0);
                const bool ok = ((j & 1) ? xok_o : xok_e) && (unsigned)(x_q + r) < (unsigned)Qi && (unsigned)(wp - p.d) < (unsigned)p.W &&
                                (unsigned)(h + dh) < (unsigned)p.H;
                glds16_tn(ok ? x_b + (unsigned long)(unsigned)pix * xrow_bytes + (((j & 1) ? xoff_o : xoff_e) - xprow * xrow_bytes) : xzero, xdst + j * 1024);
            }
        }
        // both spans advance by 64 coordinates (scalar)
        y_q += P3_KS;
        y_wp += r0;
        const int cy = y_wp >= p.WP ? 1 : 0;
        y_wp -= cy ? p.WP : 0;
        y_pix += pix_step - (cy ? 2 * p.d : 0);
        x_q += P3_KS;
        x_wp += r0;
        const int cx = x_wp >= p.WP ? 1 : 0;
        x_wp -= cx ? p.WP : 0;
        x_h += n0 + cx;
        x_h = x_h >= p.H ? x_h - p.H : x_h;
        x_pix += pix_step - (cx ? 2 * p.d : 0);
    };

    // ---- compute: wave owns i in [32 wq, +32) x all 64 o x three taps -----------------------------------------------------------
    f32x4 acc[3][2][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[kx][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, q4 = (lane & 15) >> 2, pc = lane & 3;
    const int rk0 = g * 4 + q4;
    const unsigned y_off = rk0 * Q3_YROWB + (pc >> 1) * 16 + (pc & 1) * 8;
    const int y_sw = (rk0 >> 1) & 3;
    unsigned x_off[3];
    int x_sw[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int rk = rk0 + kx * p.d;
        x_off[kx] = Q3_DY_BYTES + rk * ROWB + (pc & 1) * 8;
        x_sw[kx] = (rk & 7) << 1;
    }
    const unsigned lds0 = lds_address(smem);
    auto compute = [&](int buf) {
        unsigned sb = lds0 + buf * Q3_STAGE;
        asm volatile("" : "+v"(sb));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            union { bf16x8 v; s16x4 h[2]; } yf[4], xf[3][2];
            if (Q3_DBG(2)) {
#pragma unroll
                for (int b = 0; b < 4; ++b) yf[b].h[0] = yf[b].h[1] = s16x4{(short)lane, 1, 2, 3};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int a = 0; a < 2; ++a) xf[kx][a].h[0] = xf[kx][a].h[1] = s16x4{(short)lane, 3, 2, 1};
            }
            if (!Q3_DBG(2)) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned base = sb + y_off + ((b ^ y_sw) << 5);
                if (ks == 0) {
                    yf[b].h[0] = tr_read_lds<0>(base);
                    yf[b].h[1] = tr_read_lds<16 * Q3_YROWB>(base);
                } else {
                    yf[b].h[0] = tr_read_lds<32 * Q3_YROWB>(base);
                    yf[b].h[1] = tr_read_lds<48 * Q3_YROWB>(base);
                }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const unsigned base = sb + x_off[kx] + ((((wq * 2 + a) * 2 + (pc >> 1)) ^ x_sw[kx]) << 4);
                    if (ks == 0) {
                        xf[kx][a].h[0] = tr_read_lds<0>(base);
                        xf[kx][a].h[1] = tr_read_lds<16 * ROWB>(base);
                    } else {
                        xf[kx][a].h[0] = tr_read_lds<32 * ROWB>(base);
                        xf[kx][a].h[1] = tr_read_lds<48 * ROWB>(base);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!Q3_DBG(4)) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[kx][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kx][a].v, yf[b].v, acc[kx][a][b], 0, 0, 0);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(yf[b].v));
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int a = 0; a < 2; ++a) asm volatile("" ::"v"(xf[kx][a].v));
            }
        }
    };

    if (ns > 0) {
        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int s = 0; s < ns; ++s) {
            const int cur = s & 1;
            if (s + 1 < ns && !Q3_DBG(1)) stage(cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    const int fcol = lane & 15, fq = lane >> 4;
    if (Q3_DBG(8)) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(acc[kx][a][b]));
        return;
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        float* plane = p.slab + ((long)(split * 9 + ky * 3 + kx) * p.O) * p.I;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int o = o0 + b * 16 + fcol;
            if (o >= p.O) continue;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int i = i0 + wq * 32 + a * 16 + fq * 4;
                if (i >= p.I) continue;
                *reinterpret_cast<f32x4*>(plane + (long)o * p.I + i) = acc[kx][a][b];
            }
        }
    }
}

// Split / grid of the fused 3x3 kernel; returns false when the shape should stay on the per-tap kernel.
struct P3Plan { int S, slabs_per_split; long Q; int WP, o_tiles, i_tiles; };
// q3: the 4-wave / two-per-CU kernel (64-channel o tiles, 512 workgroup slots) instead of the 8-wave ping-pong kernel (128, 256)
bool p3_plan(int B, int H, int W, int O, int I, int dil, bool force, P3Plan& pl, bool q3 = false, bool beside = false) {
    if (dil < 1 || dil > 8) return false;
    pl.WP = W + 2 * dil;
    pl.Q = (long)B * H * pl.WP;
    pl.o_tiles = q3 ? (O + Q3_TO - 1) / Q3_TO : (O + TO - 1) / TO;
    pl.i_tiles = (I + TI - 1) / TI;
    const long ns = (pl.Q + P3_KS - 1) / P3_KS;
    const int groups = pl.o_tiles * pl.i_tiles * 3;
    // Workgroup slots the split fills.  A launch that runs alone wants both residents of every CU (512: 127 us at 256 -> 256 against 134 / 160 with 384 / 256
    // slots).  A launch that runs BESIDE a data-gradient chain (the deferred-reducer form, mi_conv_wgrad_partial: the DeepLab engines' side stream) is
    // better off leaving an eighth of the slots to its neighbour and writing 14 % fewer partial planes: the training step 288.9 vs 286.4 images/s with
    // 448, 288.5 with 384, 283.8 with 320 (three interleaved rounds, profiles/r05_wgrad_slots_ab.txt); FADA +1.0 %, trainable-BatchNorm step +2.0 %.
    long S = (q3 ? (beside ? mi_sw().wgrad_q3_slots_beside : mi_sw().wgrad_q3_slots) : 256) / groups;
    if (S < 1) S = 1;
    if (S > ns) S = ns;
    pl.slabs_per_split = (int)((ns + S - 1) / S);
    pl.S = (int)((ns + pl.slabs_per_split - 1) / pl.slabs_per_split);
    // one image-row carry per slab and per 20-row span; 32-bit coordinates
    if (P3_KS / pl.WP + 1 > H || pl.WP < 20 || pl.Q >= (1L << 31) - 4096) return false;
    if (force) return true;
    // worth it when the tiles are mostly real channels and every workgroup has a real loop behind its prologue and its
    // 192-KiB partial planes
    if (q3) return (long)O * I >= 256L * 256 && pl.slabs_per_split >= 8;       // smaller convs: the partial planes would outweigh the operands
    return O >= 96 && I >= 96 && pl.slabs_per_split >= 8;
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
    size_t need = (size_t)S * T * O * I * sizeof(float);
    if (I >= 256) {           // the 128 x 256 tile kernel splits differently
        const int S2 = pick_splits256(M, ((O + TO - 1) / TO) * ((I + TI2 - 1) / TI2) * T);
        const size_t n2 = (size_t)S2 * T * O * I * sizeof(float);
        if (n2 > need) need = n2;
    }
    if (ksize == 1) {          // the deep-stream 1x1 kernel: 32-pixel steps on the 128 x 128 tile
        const size_t n4 = (size_t)(pick_splits256(M, tiles) > pick_splits_s4(M, tiles) ? pick_splits256(M, tiles) : pick_splits_s4(M, tiles)) * O * I * sizeof(float);
        if (n4 > need) need = n4;
    }
    if (ksize == 3) {          // the fused-row kernel may be chosen for any dilation <= 8: budget for its largest split count
        P3Plan pl;
        for (int d = 1; d <= 8; d *= 2)
            for (int q3 = 0; q3 < 2; ++q3)
                if (p3_plan(B, Ho, Wo, O, I, d, true, pl, q3 != 0)) {
                    const size_t n3 = (size_t)pl.S * 9 * O * I * sizeof(float);
                    if (n3 > need) need = n3;
                }
    }
    return need;
}

static int launch_p3(const void* dy, const void* x, float* ws, int H, int W, int O, int I, int dil, const P3Plan& pl, int BH, hipStream_t st, bool q3 = false) {
    WgradP3Params q;
    q.dY = (const __bf16*)dy;
    q.X = (const __bf16*)x;
    q.slab = ws;
    q.O = O;
    q.I = I;
    q.H = H;
    q.W = W;
    q.d = dil;
    q.WP = pl.WP;
    q.BH = BH;
    q.Q = pl.Q;
    q.S = pl.S;
    q.slabs_per_split = pl.slabs_per_split;
    q.o_tiles = pl.o_tiles;
    q.i_tiles = pl.i_tiles;
    q.dbg = mi_sw().p3_dbg;
    if (q3) {
        static std::atomic<uint64_t> attrq{0};
        mi_allow_dynamic_lds((const void*)wgrad_q3_kernel, Q3_LDS, attrq);
        hipLaunchKernelGGL(wgrad_q3_kernel, dim3((unsigned)(pl.o_tiles * pl.i_tiles * 3 * pl.S)), dim3(256), Q3_LDS, st, q);
        return 0;
    }
#ifdef MI_EXPERIMENTS
    static std::atomic<uint64_t> attr{0};
    mi_allow_dynamic_lds((const void*)wgrad_p3_kernel, P3_LDS, attr);
    hipLaunchKernelGGL(wgrad_p3_kernel, dim3((unsigned)(pl.o_tiles * pl.i_tiles * 3 * pl.S)), dim3(512), P3_LDS, st, q);
    return 0;
#else
    return -1;
#endif
}

// which kernel mi_conv_wgrad launches for a shape: 0 wgrad_tn_kernel (128 x 128 per tap), 1 wgrad_tn256_kernel, 2 wgrad_p3_kernel,
// 3 wgrad_q3_kernel, 4 wgrad_s4_kernel (measurement tools; same rules as the dispatch below)
extern "C" int mi_conv_wgrad_route(int B, int Ha, int Wa, int I, int Ho, int Wo, int O, int ksize, int stride, int pad, int dil, int out_map) {
#ifdef MI_EXPERIMENTS
    const int p3_mode = mi_sw().wgrad_p3;
#else
    const int p3_mode = 0;                       // the 8-wave fused-row kernel is not in the product library
#endif
    const int q3_mode = mi_sw().wgrad_q3;
    const bool fused_ok = out_map == 0 && ksize == 3 && stride == 1 && Ha == Ho && Wa == Wo && pad == dil;
    P3Plan pl;
    if (fused_ok && q3_mode && !p3_mode && p3_plan(B, Ho, Wo, O, I, dil, q3_mode == 2, pl, true)) return 3;
    if (fused_ok && p3_mode && p3_plan(B, Ho, Wo, O, I, dil, p3_mode == 2, pl)) return 2;
    if (use_tn256(O, I, ksize, pad, stride, Ha, Ho, Wa, Wo)) return 1;
    const long M = (long)B * Ho * Wo;
    if (mi_sw().wgrad_s4 && ksize == 1 && pad == 0 && stride == 1 && Ha == Ho && Wa == Wo && M * (O > I ? O : I) * 2 < (1L << 31)) return 4;
    return 0;
}

// One reducer launch (defer == nullptr: right behind the main kernel, as always) or a job for mi_conv_wgrad_reduce (defer != nullptr)
static int wgrad_finish(WgradReduceJob* defer, hipStream_t stream, const float* slab, float* dw, const float* scale, int S, int T, int O, int I, int accumulate,
                        int out_map, int ncls, int o_real) {
    const int gx = (I + (T == 1 ? 255 : 63)) / (T == 1 ? 256 : 64);
    if (defer) {
        *defer = WgradReduceJob{slab, dw, scale, S, T, O, I, accumulate, out_map, ncls, gx, o_real};
        return MI_OK;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gx, (unsigned)o_real), dim3(256), 0, stream, slab, dw, scale, S, T, O, I, accumulate, out_map, ncls);
    MI_CHECK_LAUNCH("mi_conv_wgrad reduce");
    return MI_OK;
}

static int mi_conv_wgrad_impl(const void* dy, const void* x, float* dw, int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                              int ksize, int stride, int pad, int dil, const float* scale_o, int accumulate, int out_map, int ncls,
                              size_t dw_elems, void* workspace, size_t workspace_bytes, void* stream, WgradReduceJob* defer) {
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
    // 3x3, stride 1, pad == dilation <= 8: the fused-row kernel (three taps per x window, ping-pong main loop).
    // MI_WGRAD_P3: 0 = never (default), 1 = by the plan's own rule, 2 = whenever the geometry allows (tests: tiny shapes).
    // Opt-in because it does not win yet: at 256 -> 256, d = 2, M = 75 272 it takes 127 us against 131 us for the per-tap
    // kernel (512 -> 512: 404 vs 405).  Its parts, measured with the MI_P3_DBG toggles (tools/wgexp.py): MFMAs alone 42 us,
    // LDS->register reads 25 us, DMA alone 49 us - the same with L2-hot rows (~43 GB/s per CU; a bare stream of the same shape
    // reaches 120 GB/s from the L2, tools/micro/l2lds.hip, so the DMA issue beside the partner's MFMAs is what was measured) -,
    // partial-plane stores 14 us, slab reducer 23 us.  With the DMA at 115 % of the MFMA time the ping-pong's read segments
    // (which carry the DMA issue) outlast the MFMA segments and the two hardly overlap (main loop 80 us).
#ifdef MI_EXPERIMENTS
    const int p3_mode = mi_sw().wgrad_p3;
#else
    const int p3_mode = 0;
#endif
    // the 4-wave fused-row kernel: MI_WGRAD_Q3 0 = never, 1 = by the plan's rule (default), 2 = whenever the geometry allows (tests)
    const int q3_mode = mi_sw().wgrad_q3;
    if (q3_mode && !p3_mode && out_map == 0 && ksize == 3 && stride == 1 && Ha == Ho && Wa == Wo && pad == dil) {
        P3Plan pl;
        if (p3_plan(B, Ho, Wo, O, I, dil, q3_mode == 2, pl, true, defer != nullptr) && (size_t)pl.S * 9 * O * I * sizeof(float) <= workspace_bytes) {
            launch_p3(dy, x, (float*)workspace, Ho, Wo, O, I, dil, pl, B * Ho, (hipStream_t)stream, true);
            MI_CHECK_LAUNCH("mi_conv_wgrad (fused 3x3 rows, 4 waves)");
            return wgrad_finish(defer, (hipStream_t)stream, (const float*)workspace, dw, scale_o, pl.S, 9, O, I, accumulate, 0, 1, O);
        }
    }
    if (p3_mode && out_map == 0 && ksize == 3 && stride == 1 && Ha == Ho && Wa == Wo && pad == dil) {
        P3Plan pl;
        if (p3_plan(B, Ho, Wo, O, I, dil, p3_mode == 2, pl) && (size_t)pl.S * 9 * O * I * sizeof(float) <= workspace_bytes) {
            launch_p3(dy, x, (float*)workspace, Ho, Wo, O, I, dil, pl, B * Ho, (hipStream_t)stream);
            MI_CHECK_LAUNCH("mi_conv_wgrad (fused 3x3 rows)");
            return wgrad_finish(defer, (hipStream_t)stream, (const float*)workspace, dw, scale_o, pl.S, 9, O, I, accumulate, 0, 1, O);
        }
    }
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
    const bool wide = use_tn256(O, I, ksize, pad, stride, Ha, Ho, Wa, Wo);
    if (wide) {
        p.i_tiles = (I + TI2 - 1) / TI2;
        p.S = pick_splits256(M, p.o_tiles * p.i_tiles * p.T);
        const long steps2 = (M + KP2 - 1) / KP2;
        p.rows_per_split = (int)(((steps2 + p.S - 1) / p.S) * KP2);
        static std::atomic<uint64_t> attr2[2];
        mi_allow_dynamic_lds((const void*)wgrad_tn256_kernel<1>, LDS2_BYTES, attr2[0]);
        mi_allow_dynamic_lds((const void*)wgrad_tn256_kernel<2>, LDS2_BYTES, attr2[1]);
        const unsigned nb = (unsigned)(p.o_tiles * p.i_tiles * p.T * p.S);
        if (ksize == 1 && pad == 0)
            hipLaunchKernelGGL(wgrad_tn256_kernel<2>, dim3(nb), dim3(256), LDS2_BYTES, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(wgrad_tn256_kernel<1>, dim3(nb), dim3(256), LDS2_BYTES, (hipStream_t)stream, p);
        MI_CHECK_LAUNCH("mi_conv_wgrad (128 x 256 tile)");
    }
    const int s4_mode = mi_sw().wgrad_s4;                 // MI_WGRAD_S4=0: the 64-pixel double-buffer kernel for the 1x1 / stride-1 weight gradients too
    const bool deep = !wide && s4_mode && ksize == 1 && pad == 0 && stride == 1 && Ha == Ho && Wa == Wo && (long)M * (O > I ? O : I) * 2 < (1L << 31);
    if (deep) {
        p.i_tiles = (I + TI - 1) / TI;
        p.S = pick_splits_s4(M, p.o_tiles * p.i_tiles);
        const long steps4 = (M + KP4 - 1) / KP4;
        p.rows_per_split = (int)(((steps4 + p.S - 1) / p.S) * KP4);
        static std::atomic<uint64_t> attr4;
        mi_allow_dynamic_lds((const void*)wgrad_s4_kernel, LDS4_BYTES, attr4);
        hipLaunchKernelGGL(wgrad_s4_kernel, dim3((unsigned)(p.o_tiles * p.i_tiles * p.S)), dim3(256), LDS4_BYTES, (hipStream_t)stream, p);
        MI_CHECK_LAUNCH("mi_conv_wgrad (1x1 deep stream)");
    }
    if (!wide && !deep) {
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
    // MODE 2 addresses a split's rows through 32-bit buffer offsets: operands of 2 GiB or more take the pointer-arithmetic kernel
    const bool small32 = (long)M * (O > I ? O : I) * 2 < (1L << 31);
    if (unit && ksize == 1 && pad == 0 && small32)
        hipLaunchKernelGGL(wgrad_tn_kernel<2>, dim3(nblocks), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    else if (unit)
        hipLaunchKernelGGL(wgrad_tn_kernel<1>, dim3(nblocks), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(wgrad_tn_kernel<0>, dim3(nblocks), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_wgrad");
    }
    int o_real = O;
    if (out_map == 1) o_real = 36 * ncls;
    else ncls = 1;
    return wgrad_finish(defer, (hipStream_t)stream, p.slab, dw, scale_o, p.S, p.T, O, I, accumulate, out_map, ncls, o_real);
}

extern "C" int mi_conv_wgrad(const void* dy, const void* x, float* dw, int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                             int ksize, int stride, int pad, int dil, const float* scale_o, int accumulate, int out_map, int ncls,
                             size_t dw_elems, void* workspace, size_t workspace_bytes, void* stream) {
    return mi_conv_wgrad_impl(dy, x, dw, B, Ha, Wa, I, Ho, Wo, O, ksize, stride, pad, dil, scale_o, accumulate, out_map, ncls, dw_elems, workspace, workspace_bytes,
                              stream, nullptr);
}

// The same without the slab reducer: the split-K slabs stay in `workspace` (which must then outlive the call until mi_conv_wgrad_reduce has run) and the
// reducer's arguments are written to `job` (mi_conv_wgrad_job_bytes() bytes of HOST memory).
extern "C" size_t mi_conv_wgrad_job_bytes(void) { return sizeof(WgradReduceJob); }
extern "C" int mi_conv_wgrad_partial(const void* dy, const void* x, float* dw, int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                                     int ksize, int stride, int pad, int dil, const float* scale_o, int accumulate, int out_map, int ncls,
                                     size_t dw_elems, void* workspace, size_t workspace_bytes, void* job, void* stream) {
    MI_REQUIRE(job, "mi_conv_wgrad_partial: null job");
    return mi_conv_wgrad_impl(dy, x, dw, B, Ha, Wa, I, Ho, Wo, O, ksize, stride, pad, dil, scale_o, accumulate, out_map, ncls, dw_elems, workspace, workspace_bytes,
                              stream, (WgradReduceJob*)job);
}
// jobs: n consecutive job records written by mi_conv_wgrad_partial (n <= 8): their reducers as ONE launch, each summing its slabs in its fixed order
extern "C" int mi_conv_wgrad_reduce(const void* jobs, int n, void* stream) {
    MI_REQUIRE(jobs && n >= 1 && n <= MI_REDUCE_MAX_JOBS, "mi_conv_wgrad_reduce: 1 .. %d jobs", MI_REDUCE_MAX_JOBS);
    WgradReduceTable tab;
    tab.n = n;
    int total = 0;
    for (int j = 0; j < n; ++j) {
        tab.job[j] = ((const WgradReduceJob*)jobs)[j];
        MI_REQUIRE(tab.job[j].slab && tab.job[j].dw && tab.job[j].gx > 0 && tab.job[j].gy > 0, "mi_conv_wgrad_reduce: job %d is not a record of mi_conv_wgrad_partial", j);
        tab.first[j] = total;
        total += tab.job[j].gx * tab.job[j].gy;
    }
    for (int j = n; j <= MI_REDUCE_MAX_JOBS; ++j) tab.first[j] = total;
    for (int j = n; j < MI_REDUCE_MAX_JOBS; ++j) tab.job[j] = tab.job[0];
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, tab);
    MI_CHECK_LAUNCH("mi_conv_wgrad_reduce");
    return MI_OK;
}
