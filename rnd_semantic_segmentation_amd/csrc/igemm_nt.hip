// Implicit-GEMM convolution for gfx950: bf16 operands, fp32 MFMA accumulate, fused epilogue.
//
//   out[m][n] = epi( sum_t sum_c A[src(m,t)][c] * Wp[t][n][c] )          m = (b,ho,wo) pixel, n = out channel
//
// One kernel serves: dilated / plain 3x3 and 1x1 convs forward (gather mode FWD), their data gradients
// (gather mode DGRAD with transposed packed weights), and the two plain GEMMs of the ASPP head.
// Replaces nn.Conv2d at reference core/components/resnet.py:22-30 + FrozenBN/ReLU/residual at :93-113.
//
// Structure (round 1): 128(m) x 128(n) tile, BK = 64 channels of one tap per step, 4 waves (2x2), each wave
// 64x64 = 4x4 MFMA 16x16x32 tiles.  Operand tiles are staged global -> VGPR -> LDS (16-B chunks, XOR-swizzled
// 128-B rows so ds_read_b128 is conflict-free), double-buffered with the next tile's global loads issued
// before the current tile's MFMAs.  Zero padding is predication in the gather, never a materialised im2col.
// MFMA orientation: D rows = n (weights are the "A" operand), D cols = m, so a lane owns 4 consecutive
// channels of one pixel and stores them as one 8-byte (bf16) or 16-byte (fp32) vector.
#include "mi_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // double buffer = 64 KiB

struct IgemmParams {
    const __bf16* A;
    const __bf16* Wp;
    void* out;
    const float* scale;
    const float* bias;
    const __bf16* res;
    const __bf16* msk;
    int M, N, Ca, T;
    int Ho, Wo, Ha, Wa;
    int ksz, stride, pad, dil, mode;
    int flags, zgw;
    int m_tiles, n_tiles;
};

__device__ __forceinline__ bool tap_src(const IgemmParams& p, int ho, int wo, int ky, int kx, int& ha, int& wa) {
    if (p.mode == MI_GATHER_FWD) {
        ha = ho * p.stride + ky * p.dil - p.pad;
        wa = wo * p.stride + kx * p.dil - p.pad;
        return (unsigned)ha < (unsigned)p.Ha && (unsigned)wa < (unsigned)p.Wa;
    }
    int nh = ho + p.pad - ky * p.dil, nw = wo + p.pad - kx * p.dil;
    if (nh < 0 || nw < 0) return false;
    if (p.stride > 1) {
        if ((nh % p.stride) | (nw % p.stride)) return false;
        nh /= p.stride;
        nw /= p.stride;
    }
    ha = nh;
    wa = nw;
    return nh < p.Ha && nw < p.Wa;
}

__global__ __launch_bounds__(256, 2) void igemm_nt_kernel(IgemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = p.m_tiles * p.n_tiles;
    const int tile = mi_xcd_remap(blockIdx.x, nwg);
    const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread staging assignment: 4 A chunks + 4 B chunks of 16 B; rows r0 + 32*i, chunk kc ----
    const int kc = tid & 7, r0 = tid >> 3;
    int a_img[4], a_ho[4], a_wo[4];   // a_img = b*Ha*Wa (or -1 if the row is past M)
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            a_ho[i] = rem / p.Wo;
            a_wo[i] = rem - a_ho[i] * p.Wo;
            a_img[i] = b * p.Ha * p.Wa;
        } else {
            a_img[i] = -1;
            a_ho[i] = a_wo[i] = 0;
        }
    }
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b_ok[i] = (n0 + r0 + 32 * i) < p.N;

    const int cpt = p.Ca >> 6;          // 64-channel chunks per tap
    const int nk = p.T * cpt;
    int ld_t = 0, ld_cc = 0;            // tap / chunk of the NEXT tile to load
    long a_off[4];                      // element offset of this thread's chunk for the current tap (or -1)

    auto set_tap = [&](int t) {
        const int ky = t / p.ksz, kx = t - ky * p.ksz;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ha, wa;
            if (a_img[i] >= 0 && tap_src(p, a_ho[i], a_wo[i], ky, kx, ha, wa))
                a_off[i] = ((long)(a_img[i] + ha * p.Wa + wa)) * p.Ca + kc * 8;
            else
                a_off[i] = -1;
        }
    };
    set_tap(0);

    u32x4 ra[4], rb[4];
    auto load_tile = [&]() {
        const int cbase = ld_cc * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (a_off[i] >= 0) ? *reinterpret_cast<const u32x4*>(p.A + a_off[i] + cbase) : u32x4{0, 0, 0, 0};
        }
        const long wrow = (long)ld_t * p.N + n0 + r0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rb[i] = b_ok[i] ? *reinterpret_cast<const u32x4*>(p.Wp + (wrow + 32 * i) * p.Ca + cbase + kc * 8) : u32x4{0, 0, 0, 0};
        }
        if (++ld_cc == cpt) {
            ld_cc = 0;
            ++ld_t;
            if (ld_t < p.T) set_tap(ld_t);
        }
    };
    auto store_tile = [&](int stage) {
        char* sa = smem + stage * STAGE_BYTES;
        char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 32 * i;
            const int off = r * 128 + ((kc ^ (r & 7)) << 4);
            *reinterpret_cast<u32x4*>(sa + off) = ra[i];
            *reinterpret_cast<u32x4*>(sb + off) = rb[i];
        }
    };

    const int wm = wave & 1, wn = wave >> 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE_BYTES;
        const char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = (((kk * 4 + fq) ^ (frow & 7)) << 4);
            bf16x8 wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + (wn * 64 + i * 16 + frow) * 128 + sw);
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const bf16x8*>(sa + (wm * 64 + j * 16 + frow) * 128 + sw);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    };

    load_tile();
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) load_tile();
        compute(cur);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns pixel m (col of D) and channels n..n+3 (rows of D) -----------------------
    const int flags = p.flags;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fq * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (flags & MI_EPI_SCALE_BIAS) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.scale + n);
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
                v = v * s + b;
            }
            const long o = (long)m * p.N + n;
            if (flags & MI_EPI_RESIDUAL) {
                const bf16x4 r = *reinterpret_cast<const bf16x4*>(p.res + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
            }
            if (flags & MI_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            if (flags & MI_EPI_MASK) {
                const bf16x4 k = *reinterpret_cast<const bf16x4*>(p.msk + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ((float)k[e] > 0.f) ? v[e] : 0.f;
            }
            if (flags & MI_EPI_ZSPLIT) {
                const int g = n / p.zgw, nn = n - g * p.zgw;
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + ((long)g * p.M + m) * p.zgw + nn) = v;
            } else if (flags & MI_EPI_OUT_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = v;
            } else {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + o) = h;
            }
        }
    }
}

}  // namespace

extern "C" int mi_conv_gemm(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                            int ksize, int stride, int pad, int dil, int gather_mode, const float* scale, const float* bias,
                            const void* res, const void* msk, int flags, int zgw, void* stream) {
    MI_REQUIRE(a && wp && out, "mi_conv_gemm: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0 && N > 0, "mi_conv_gemm: non-positive dimension");
    MI_REQUIRE(Ca > 0 && Ca % 64 == 0, "mi_conv_gemm: Ca=%d must be a multiple of 64", Ca);
    MI_REQUIRE(N % 4 == 0, "mi_conv_gemm: N=%d must be a multiple of 4", N);
    MI_REQUIRE(ksize == 1 || ksize == 3, "mi_conv_gemm: ksize=%d (1 or 3)", ksize);
    MI_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0, "mi_conv_gemm: bad stride/dil/pad");
    MI_REQUIRE(gather_mode == MI_GATHER_FWD || gather_mode == MI_GATHER_DGRAD, "mi_conv_gemm: gather_mode");
    MI_REQUIRE(mi_aligned16(a) && mi_aligned16(wp) && mi_aligned16(out), "mi_conv_gemm: operands must be 16-byte aligned");
    MI_REQUIRE(!(flags & MI_EPI_SCALE_BIAS) || (scale && bias && mi_aligned16(scale) && mi_aligned16(bias)), "mi_conv_gemm: scale/bias");
    MI_REQUIRE(!(flags & MI_EPI_RESIDUAL) || (res && ((uintptr_t)res & 7) == 0), "mi_conv_gemm: residual");
    MI_REQUIRE(!(flags & MI_EPI_MASK) || (msk && ((uintptr_t)msk & 7) == 0), "mi_conv_gemm: mask");
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
    p.m_tiles = (int)((M + BM - 1) / BM);
    p.n_tiles = (N + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm_nt_kernel, dim3(p.m_tiles * p.n_tiles), dim3(256), LDS_BYTES, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_gemm");
    return MI_OK;
}
