// Implicit-GEMM convolution for gfx950: bf16 operands, fp32 MFMA accumulate, fused epilogue.
//
//   out[m][n] = epi( sum_t sum_c A[src(m,t)][c] * Wp[t][n][c] )          m = (b,ho,wo) pixel, n = out channel
//
// One kernel serves: dilated / plain 3x3 and 1x1 convs forward (gather mode FWD), their data gradients
// (gather mode DGRAD with transposed packed weights), and the two plain GEMMs of the ASPP head.
// Replaces nn.Conv2d at reference core/components/resnet.py:22-30 + FrozenBN/ReLU/residual at :93-113.
//
// Structure: 128(m) x 128(n) tile, BK = 64 channels of one tap per step, 4 waves (2x2), each wave 64x64 = 4x4
// MFMA 16x16x32 tiles, 2 workgroups per CU.  Operand tiles go global -> LDS directly (global_load_lds, 16 B per
// lane, 1 KiB = 8 rows per wave-instruction), double-buffered: the next tile's DMA is in flight during the current
// tile's MFMAs.  The LDS image is lane-linear per DMA, so the bank-conflict XOR swizzle of the 16-B chunks is applied
// to the per-lane SOURCE address and to the ds_read_b128 address (never to the destination).  Zero padding of the
// convolution (and M / N tails) is a per-lane source pointer into a zero page - never a materialised im2col.
// MFMA orientation: D rows = n (weights are the "A" operand), D cols = m.  The weight rows feeding MFMA tile i are
// permuted (row rho of tile i is channel 16*(rho>>2) + 4*i + (rho&3) of the wave's 64), so that a lane ends up with
// 16 CONTIGUOUS channels of one pixel: 4 lanes store a whole 128-B line, and residual / mask loads are 16-B wide.
#include "mi_common.h"
#include <stdlib.h>

namespace {

// Tile height is a template parameter: MT 16-row MFMA tiles per wave in M -> BM = 32*MT rows (128 or 160).  The
// launcher picks the one with fewer (rounds x rows) on the 512 resident workgroup slots: at M = 75 272, N = 256 the
// 128-row tile needs 1178 workgroups = 3 rounds, the 160-row tile 942 = 2 rounds.
constexpr int BN = 128, BK = 64;
constexpr int WTILE_BYTES = BN * BK * 2;         // 16 KiB weight tile
template <int MT> struct Geo {
    static constexpr int BM = 32 * MT;
    static constexpr int ATILE_BYTES = BM * BK * 2;
    static constexpr int STAGE_BYTES = ATILE_BYTES + WTILE_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // double buffer: 64 KiB (MT 4) / 72 KiB (MT 5) / 80 KiB (MT 6)
};

__device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];   // source of every padded / out-of-range chunk

struct IgemmParams {
    const __bf16* A;
    const __bf16* Wp;
    void* out;
    const float* scale;
    const float* bias;
    const __bf16* res;
    const __bf16* msk;            // bf16 tensor (MI_EPI_MASK) or packed sign bits, uint16 per 16 channels (MI_EPI_BITMASK)
    uint16_t* mask_out;           // MI_EPI_WRITE_MASK: bit c%16 of word [m][c/16] = (out[m][c] > 0)
    int M, N, Ca, T;
    int Ho, Wo, Ha, Wa;
    int ksz, stride, pad, dil, mode;
    int flags, zgw;
    float alpha;                  // LeakyReLU negative slope (MI_EPI_LEAKY)
    int m_tiles, n_tiles;
};

__device__ __forceinline__ bool tap_src(const IgemmParams& p, int ho, int wo, int ky, int kx, int& ha, int& wa) {
    // general gather (any stride), branch-free
    const bool fwd = p.mode == MI_GATHER_FWD;
    const int fh = ho * p.stride + ky * p.dil - p.pad, fw = wo * p.stride + kx * p.dil - p.pad;
    const int nh = ho + p.pad - ky * p.dil, nw = wo + p.pad - kx * p.dil;
    const int qh = nh / p.stride, qw = nw / p.stride;              // stride >= 1; exactness checked below
    const bool dg_ok = (nh >= 0) & (nw >= 0) & (qh * p.stride == nh) & (qw * p.stride == nw);
    ha = fwd ? fh : qh;
    wa = fwd ? fw : qw;
    return (fwd | dg_ok) & ((unsigned)ha < (unsigned)p.Ha) & ((unsigned)wa < (unsigned)p.Wa);
}

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// PREF: the epilogue's residual rows and mask bits are requested BEFORE the main loop (they are 150-300 MB of HBM traffic
// per launch on the 1024/2048-channel tensors) so that they land behind the MFMA work instead of after it.
template <int MT, bool UNIT, bool PREF>
__global__ __launch_bounds__(256, 2) void igemm_nt_kernel(IgemmParams p) {
    constexpr int BM = Geo<MT>::BM, ATILE_BYTES = Geo<MT>::ATILE_BYTES, STAGE_BYTES = Geo<MT>::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.m_tiles * p.n_tiles;
    const int tile = mi_xcd_remap(blockIdx.x, nwg);
    const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- DMA assignment: wave w moves pieces 4w..4w+3 of each operand tile; a piece = 8 rows x 128 B.
    //      lane -> (row = piece*8 + lane>>3, physical chunk = lane&7); it fetches logical chunk physical ^ s(row).
    const int prow = lane >> 3, pch = lane & 7;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    // Per-row state.  UNIT (stride 1, the 100+ launches per step that matter): the source pixel of tap t is the row's own
    // pixel plus a tap offset that is THE SAME for every row, so a tap change costs a handful of VALU ops per row:
    // a 64-bit base pointer per row, one scalar byte offset per tap, and a 9-bit per-row validity mask computed once.
    // (General path, stride 2: coordinates are kept and the source recomputed per tap.)
    int a_img[MT], a_ho[MT], a_wo[MT];
    const char* a_base[MT];
    unsigned a_mask[MT];
    const int HoWo = p.Ho * p.Wo;
    const int a_chunk = (pch ^ (prow & 7)) * 16;                       // s_A(row) = row & 7
    const int sgn = (p.mode == MI_GATHER_FWD) ? 1 : -1;                // FWD: src = out + tap*dil - pad ; DGRAD: out + pad - tap*dil
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wave * MT + i) * 8 + prow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / HoWo, rem = mm - b * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_ho[i] = ho;
        a_wo[i] = wo;
        a_img[i] = ok ? b * p.Ha * p.Wa : -1;
        if (UNIT) {
            const int h0 = ho - sgn * p.pad, w0 = wo - sgn * p.pad;   // tap (0,0) source; Ha == Ho, Wa == Wo here
            a_base[i] = reinterpret_cast<const char*>(p.A + ((long)b * p.Ha * p.Wa + (long)h0 * p.Wa + w0) * p.Ca) + a_chunk;
            unsigned msk = 0;
            for (int t = 0; t < p.T; ++t) {
                const int ky = t / p.ksz, kx = t - ky * p.ksz;
                const int hs = h0 + sgn * ky * p.dil, ws = w0 + sgn * kx * p.dil;
                msk |= (unsigned)(ok & ((unsigned)hs < (unsigned)p.Ha) & ((unsigned)ws < (unsigned)p.Wa)) << t;
            }
            a_mask[i] = msk;
        }
    }
    int w_row[4], w_chunk[4];
    const char* w_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w_row[i] = n0 + (wave * 4 + i) * 8 + prow;
        w_chunk[i] = (pch ^ ((prow & 3) | ((i >> 1) << 2))) * 16;      // s_W(row) = (row&3) | ((row>>4)&1)<<2
        w_base[i] = reinterpret_cast<const char*>(p.Wp + (long)(w_row[i] < p.N ? w_row[i] : 0) * p.Ca) + w_chunk[i];
    }

    const int cpt = p.Ca >> 6;          // 64-channel chunks per tap
    const int nk = p.T * cpt;
    int ld_t = 0, ld_cc = 0;            // tap / chunk of the NEXT tile to stage
    const char* a_ptr[MT];
    const char* w_ptr[4];
    int a_inc[MT], w_inc[4];
    const long w_tap_bytes = (long)p.N * p.Ca * 2;

    auto set_tap = [&](int t) __attribute__((always_inline)) {
        const int ky = t / p.ksz, kx = t - ky * p.ksz;
        if (UNIT) {
            const long toff = (long)sgn * ((long)ky * p.dil * p.Wa + (long)kx * p.dil) * p.Ca * 2;   // wave-uniform
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bool ok = (a_mask[i] >> t) & 1u;
                a_ptr[i] = ok ? a_base[i] + toff : zero;
                a_inc[i] = ok ? 128 : 0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int ha = 0, wa = 0;
                const bool ok = tap_src(p, a_ho[i], a_wo[i], ky, kx, ha, wa) && a_img[i] >= 0;
                const char* ptr = reinterpret_cast<const char*>(p.A + ((long)((ok ? a_img[i] : 0) + (ok ? ha : 0) * p.Wa + (ok ? wa : 0))) * p.Ca) + a_chunk;
                a_ptr[i] = ok ? ptr : zero;
                a_inc[i] = ok ? 128 : 0;
            }
        }
        const long woff = (long)t * w_tap_bytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = w_row[i] < p.N;
            w_ptr[i] = ok ? w_base[i] + woff : zero;
            w_inc[i] = ok ? 128 : 0;
        }
    };
    set_tap(0);

    auto stage = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES + wave * (MT * 1024);
        char* sb = smem + buf * STAGE_BYTES + ATILE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            glds16(a_ptr[i], sa + i * 1024);
            a_ptr[i] += a_inc[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(w_ptr[i], sb + i * 1024);
            w_ptr[i] += w_inc[i];
        }
        if (++ld_cc == cpt) {
            ld_cc = 0;
            ++ld_t;
            if (ld_t < p.T) set_tap(ld_t);
        }
    };

    const int wm = wave & 1, wn = wave >> 1;
    f32x4 acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int wrow0 = wn * 64 + 16 * (frow >> 2) + (frow & 3);          // + 4*i : permuted weight row of MFMA tile i
    bf16x8 pres[PREF ? MT : 1][2];
    unsigned pbits[PREF ? MT : 1];
    if (PREF) {
        const int nbp = n0 + wn * 64 + 16 * fq;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + wm * (MT * 16) + j * 16 + frow;
            const long o = (long)(m < p.M ? m : 0) * p.N + (nbp < p.N ? nbp : 0);
            if (p.flags & MI_EPI_RESIDUAL) {
                pres[j][0] = *reinterpret_cast<const bf16x8*>(p.res + o);
                pres[j][1] = *reinterpret_cast<const bf16x8*>(p.res + o + 8);
            }
            if (p.flags & MI_EPI_BITMASK) pbits[j] = reinterpret_cast<const uint16_t*>(p.msk)[o >> 4];
        }
    }
    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE_BYTES;
        const char* sb = sa + ATILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = (((kk * 4 + fq) ^ (frow & 7)) << 4);         // same expression for both operands (see header)
            bf16x8 wf[4], af[MT];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + (wrow0 + 4 * i) * 128 + sw);
#pragma unroll
            for (int j = 0; j < MT; ++j) af[j] = *reinterpret_cast<const bf16x8*>(sa + (wm * (MT * 16) + j * 16 + frow) * 128 + sw);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    };

    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1);          // DMA of the next tile flies during this tile's MFMAs
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (p.flags & (1 << 30)) {   // perf experiment: main loop only (keeps the accumulators alive, stores nothing)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // ---- epilogue: lane owns pixel m (D col) and the 16 contiguous channels nb .. nb+15 (4 per MFMA tile i) -------
    const int flags = p.flags;
    const int nb = n0 + wn * 64 + 16 * fq;
    f32x4 esc[4], ebi[4];               // FrozenBN scale / shift of this lane's 16 channels: loaded once, not per pixel
    if (flags & MI_EPI_SCALE_BIAS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (nb + 4 * i < p.N) ? nb + 4 * i : 0;
            esc[i] = *reinterpret_cast<const f32x4*>(p.scale + n);
            ebi[i] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * (MT * 16) + j * 16 + frow;
        if (m >= p.M) continue;
        const long o = (long)m * p.N + nb;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[i][j];
        if (flags & MI_EPI_SCALE_BIAS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] * esc[i] + ebi[i];
        }
        if (flags & MI_EPI_RESIDUAL) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 8 * h < p.N) {
                    const bf16x8 r = PREF ? pres[PREF ? j : 0][h] : *reinterpret_cast<const bf16x8*>(p.res + o + 8 * h);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[2 * h + (e >> 2)][e & 3] += (float)r[e];
                }
            }
        }
        if (flags & MI_EPI_RELU) {
            const float neg = (flags & MI_EPI_LEAKY) ? p.alpha : 0.f;       // LeakyReLU(alpha) when MI_EPI_LEAKY is set too
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] > 0.f ? v[i][e] : neg * v[i][e];
        }
        if (flags & MI_EPI_MASK) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 8 * h < p.N) {
                    const bf16x8 k = *reinterpret_cast<const bf16x8*>(p.msk + o + 8 * h);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[2 * h + (e >> 2)][e & 3] = ((float)k[e] > 0.f) ? v[2 * h + (e >> 2)][e & 3] : 0.f;
                }
            }
        }
        if (flags & MI_EPI_BITMASK) {
            if (nb < p.N) {
                const unsigned bits = PREF ? pbits[PREF ? j : 0] : reinterpret_cast<const uint16_t*>(p.msk)[(o >> 4)];
                const float neg = (flags & MI_EPI_LEAKY) ? p.alpha : 0.f;   // backward of LeakyReLU: gradient x alpha where the sign bit is 0
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c >> 2][c & 3] = ((bits >> c) & 1u) ? v[c >> 2][c & 3] : neg * v[c >> 2][c & 3];
            }
        }
        if (flags & MI_EPI_WRITE_MASK) {
            if (nb < p.N) {
                unsigned bits = 0;
#pragma unroll
                for (int c = 0; c < 16; ++c) bits |= (v[c >> 2][c & 3] > 0.f ? 1u : 0u) << c;
                p.mask_out[o >> 4] = (uint16_t)bits;
            }
        }
        if (flags & MI_EPI_ZSPLIT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nb + 4 * i;
                if (n < p.N) {
                    const int g = n / p.zgw, nn = n - g * p.zgw;
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + ((long)g * p.M + m) * p.zgw + nn) = v[i];
                }
            }
        } else if (flags & MI_EPI_OUT_F32) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (nb + 4 * i < p.N) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o + 4 * i) = v[i];
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 8 * h < p.N) {
                    bf16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (__bf16)v[2 * h + (e >> 2)][e & 3];
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.out) + o + 8 * h) = hv;
                }
            }
        }
    }
}

}  // namespace

extern "C" int mi_conv_gemm(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                            int ksize, int stride, int pad, int dil, int gather_mode, const float* scale, const float* bias,
                            const void* res, const void* msk, void* mask_out, int flags, int zgw, float alpha, void* stream) {
    MI_REQUIRE(a && wp && out, "mi_conv_gemm: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0 && N > 0, "mi_conv_gemm: non-positive dimension");
    MI_REQUIRE(Ca > 0 && Ca % 64 == 0, "mi_conv_gemm: Ca=%d must be a multiple of 64", Ca);
    MI_REQUIRE(N % 8 == 0, "mi_conv_gemm: N=%d must be a multiple of 8", N);
    MI_REQUIRE(ksize == 1 || ksize == 3, "mi_conv_gemm: ksize=%d (1 or 3)", ksize);
    MI_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0, "mi_conv_gemm: bad stride/dil/pad");
    MI_REQUIRE(gather_mode == MI_GATHER_FWD || gather_mode == MI_GATHER_DGRAD, "mi_conv_gemm: gather_mode");
    MI_REQUIRE(mi_aligned16(a) && mi_aligned16(wp) && mi_aligned16(out), "mi_conv_gemm: operands must be 16-byte aligned");
    MI_REQUIRE(!(flags & MI_EPI_SCALE_BIAS) || (scale && bias && mi_aligned16(scale) && mi_aligned16(bias)), "mi_conv_gemm: scale/bias");
    MI_REQUIRE(!(flags & MI_EPI_RESIDUAL) || (res && mi_aligned16(res)), "mi_conv_gemm: residual");
    MI_REQUIRE(!(flags & MI_EPI_MASK) || (msk && mi_aligned16(msk)), "mi_conv_gemm: mask");
    MI_REQUIRE(!(flags & MI_EPI_BITMASK) || (msk && N % 16 == 0 && !(flags & MI_EPI_MASK)), "mi_conv_gemm: bit mask needs N %% 16 == 0");
    MI_REQUIRE(!(flags & MI_EPI_WRITE_MASK) || (mask_out && N % 16 == 0), "mi_conv_gemm: mask_out needs N %% 16 == 0");
    MI_REQUIRE(!(flags & MI_EPI_ZSPLIT) || (zgw > 0 && zgw % 4 == 0 && N % zgw == 0), "mi_conv_gemm: zgw");
    const long M = (long)B * Ho * Wo;
    MI_REQUIRE(M < (1L << 31) && (long)B * Ha * Wa < (1L << 31), "mi_conv_gemm: pixel count overflows int32");
    if (gather_mode == MI_GATHER_FWD) {
        MI_REQUIRE((Ho - 1) * stride - pad < Ha && (Wo - 1) * stride - pad < Wa, "mi_conv_gemm: output larger than the input supports");
    }
    IgemmParams p;
    p.A = (const __bf16*)a;
    p.Wp = (const __bf16*)wp;
    p.out = out;
    p.scale = scale;
    p.bias = bias;
    p.res = (const __bf16*)res;
    p.msk = (const __bf16*)msk;
    p.mask_out = (uint16_t*)mask_out;
    p.M = (int)M;
    p.N = N;
    p.Ca = Ca;
    p.T = ksize * ksize;
    p.Ho = Ho;
    p.Wo = Wo;
    p.Ha = Ha;
    p.Wa = Wa;
    p.ksz = ksize;
    p.stride = stride;
    p.pad = pad;
    p.dil = dil;
    p.mode = gather_mode;
    p.flags = flags;
    p.zgw = zgw > 0 ? zgw : 4;
    p.alpha = alpha;
    p.n_tiles = (N + BN - 1) / BN;
    // tile height: fewer (rounds x rows) on the 512 resident workgroup slots wins; ties go to the smaller tile
    // cost ~ rounds x rows, discounted by the L2 bytes a taller tile saves per flop (the kernel is L2->LDS bound:
    // bytes per K-step ~ (bm + 128) for bm*128 outputs)
    auto cost = [&](int bm) {
        const long tiles = ((M + bm - 1) / bm) * p.n_tiles;
        return (double)(((tiles + 511) / 512) * bm) * (0.5 + 0.5 * (double)(bm + 128) / (2.0 * bm));
    };
    int mt_sel = 4;
    if (cost(160) < cost(128) * 0.999) mt_sel = 5;
    if (cost(192) < cost(mt_sel * 32) * 0.999) mt_sel = 6;
    static int force_mt = -1;
    if (force_mt < 0) {
        const char* e = getenv("MI_IGEMM_MT");
        force_mt = e ? atoi(e) : 0;
    }
    if (force_mt >= 4 && force_mt <= 6) mt_sel = force_mt;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<4, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<4>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<5, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<5>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<6, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<6>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<4>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<5, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<5>::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel<4, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<4>::LDS_BYTES);
        attr_set = true;
    }
    const bool unit = stride == 1 && Ha == Ho && Wa == Wo && ksize * ksize <= 9;
    static int pref_on = -1;
    if (pref_on < 0) {
        const char* e = getenv("MI_IGEMM_PREF");
        pref_on = e ? atoi(e) : 1;
    }
    const bool pref = pref_on && unit && (flags & MI_EPI_RESIDUAL) && N % 16 == 0;
    if (!unit) mt_sel = 4;                               // the general (strided) gather exists in the 128-row shape only
    if (pref && mt_sel == 6) mt_sel = 5;                 // the prefetched rows cost 8 VGPRs per 16-row MFMA tile
    const int bm = mt_sel * 32;
    p.m_tiles = (int)((M + bm - 1) / bm);
    const dim3 grid(p.m_tiles * p.n_tiles);
    if (!unit)
        hipLaunchKernelGGL((igemm_nt_kernel<4, false, false>), grid, dim3(256), Geo<4>::LDS_BYTES, (hipStream_t)stream, p);
    else if (pref && mt_sel == 5)
        hipLaunchKernelGGL((igemm_nt_kernel<5, true, true>), grid, dim3(256), Geo<5>::LDS_BYTES, (hipStream_t)stream, p);
    else if (pref)
        hipLaunchKernelGGL((igemm_nt_kernel<4, true, true>), grid, dim3(256), Geo<4>::LDS_BYTES, (hipStream_t)stream, p);
    else if (mt_sel == 6)
        hipLaunchKernelGGL((igemm_nt_kernel<6, true, false>), grid, dim3(256), Geo<6>::LDS_BYTES, (hipStream_t)stream, p);
    else if (mt_sel == 5)
        hipLaunchKernelGGL((igemm_nt_kernel<5, true, false>), grid, dim3(256), Geo<5>::LDS_BYTES, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((igemm_nt_kernel<4, true, false>), grid, dim3(256), Geo<4>::LDS_BYTES, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_gemm");
    return MI_OK;
}
