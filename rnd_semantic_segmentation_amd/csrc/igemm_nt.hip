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
// permuted (row rho of tile i is channel 32*(i>>1) + 8*(rho>>2) + 4*(i&1) + (rho&3) of the wave's 64), so that a lane
// (pixel = lane&15, q = lane>>4) ends up with two groups of 8 CONTIGUOUS channels, 8q.. and 32+8q..: one 16-B store /
// residual load / mask byte per group, and the 4 lanes of a pixel cover 64 CONTIGUOUS bytes per instruction (two full
// 32-B sectors; two instructions fill the 128-B line).
#include "igemm_common.h"
#include <stdlib.h>

namespace {

// Tile shape is a template parameter: MT 16-row MFMA tiles per wave in M and a 2 x NWN wave grid, each wave owning 16*MT x 64
// outputs -> block tile BM = 32*MT rows x BNT = 64*NWN columns:
//   NWN 2: 4 waves, 128/160/192 x 128, two workgroups per CU;
//   NWN 4: 8 waves, 128/160/192 x 256, one workgroup per CU - 30 % fewer L2 bytes per FLOP, but slower with this loop structure
//          (see the launcher); opt-in with MI_IGEMM_BN=256.
// The launcher picks the shape with the lowest modelled time: rounds on the resident-workgroup slots x per-step tile time
// (L2 bytes (BM+BNT)*128 at the per-CU L2 rate vs MFMA time) - at M = 75 272 the 160-row tiles waste the least of the last round.
// BKT (channels per K-step) is 64; a 32-channel variant (half the LDS, 3-4 workgroups per CU) compiles from the same template
// and measured equal or slower on every shape (the short-K convs are bound by L2/HBM traffic, not per-workgroup latency).
template <int MT, int BKT = 64, int NWN = 2> struct Geo {
    static constexpr int NW = 2 * NWN;                      // waves per workgroup
    static constexpr int BM = 32 * MT, BNT = 64 * NWN;
    static constexpr int ROWB = BKT * 2;                    // bytes of one tile row in LDS
    static constexpr int RPP = 1024 / ROWB;                 // rows per DMA piece (one 1-KiB wave instruction)
    static constexpr int CPR = ROWB / 16;                   // 16-B chunks per row
    static constexpr int PA = BM / RPP;                     // A pieces per stage; wave w moves PA/NW of them (+1 for w < PA%NW)
    static constexpr int NPA = (PA + NW - 1) / NW;
    static constexpr int NPW = BNT / RPP / NW;              // W pieces per wave per stage
    static constexpr int ATILE_BYTES = BM * ROWB;
    static constexpr int WTILE_BYTES = BNT * ROWB;
    static constexpr int STAGE_BYTES = ATILE_BYTES + WTILE_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // double buffer; NWN 2: 64 / 72 / 80 KiB (MT 4 / 5 / 6), NWN 4: 96 / 104 / 112 KiB
    static constexpr int OCC = NWN == 4 ? 1 : (BKT == 64 ? 2 : (MT <= 4 ? 4 : 3));
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

// PREF: the epilogue's residual rows and mask bits are requested BEFORE the main loop (they are 150-300 MB of HBM traffic
// per launch on the 1024/2048-channel tensors) so that they land behind the MFMA work instead of after it.
// EPI >= 0: the epilogue flag set is a compile-time constant (the four sets the ResNet bottlenecks launch 200x per step get a
// straight-line epilogue: no per-flag branches, no dead ZSPLIT / fp32 paths); EPI < 0: flags are read from the parameters.
// STG: residual in / bf16 tile out through an LDS image with row-contiguous lanes (igemm_common.h); 128-column tiles with a
// compile-time epilogue flag set only.
template <int MT, bool UNIT, bool PREF, int BKT, int EPI, int NWN, bool STG = false>
__global__ __launch_bounds__((Geo<MT, BKT, NWN>::NW * 64), (Geo<MT, BKT, NWN>::OCC)) void igemm_nt_kernel(IgemmParams p) {
    static_assert(!STG || (EPI >= 0 && !PREF && NWN == 2 && BKT == 64 && !(EPI & (MI_EPI_OUT_F32 | MI_EPI_ZSPLIT))), "staged epilogue: hot bf16 sets only");
    using G = Geo<MT, BKT, NWN>;
    constexpr int BM = G::BM, BN = G::BNT, ATILE_BYTES = G::ATILE_BYTES, STAGE_BYTES = G::STAGE_BYTES;
    constexpr int ROWB = G::ROWB, RPP = G::RPP, CPR = G::CPR, NPA = G::NPA, NPW = G::NPW, NW = G::NW;
    static_assert(BM % RPP == 0 && BN % (RPP * NW) == 0, "tiles must split into whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = p.m_tiles * p.n_tiles;
    const int tile = mi_xcd_remap(blockIdx.x, nwg);
    const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- DMA assignment: wave w moves pieces w*NP .. w*NP+NP-1 of each operand tile; a piece = RPP rows x ROWB bytes = 1 KiB.
    //      lane -> (row = piece*RPP + lane/CPR, physical chunk = lane%CPR); it fetches logical chunk physical ^ s(row).
    //      Swizzles: BKT 64: s_A(row) = row&7, s_W(row) = (row&3) | ((row>>3)&1)<<2;  BKT 32: s_A(row) = (row>>2)&3,
    //      s_W(row) = (row>>3)&3 - chosen so that, for the rows a quarter-wave reads (A: base + f, W: the permuted rows
    //      below), both equal the same function of the lane (f&7, resp. (f>>2)&3) and the 16 lanes hit 16 distinct slots.
    const int prow = lane / CPR, pch = lane % CPR;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    // Per-row state.  UNIT (stride 1, the 100+ launches per step that matter): the source pixel of tap t is the row's own
    // pixel plus a tap offset that is THE SAME for every row, so a tap change costs a handful of VALU ops per row:
    // a 64-bit base pointer per row, one scalar byte offset per tap, and a 9-bit per-row validity mask computed once.
    // (General path, stride 2: coordinates are kept and the source recomputed per tap.)
    int a_img[NPA], a_ho[NPA], a_wo[NPA];
    const char* a_base[NPA];
    unsigned a_mask[NPA];
    const int HoWo = p.Ho * p.Wo;
    const int a_chunk = (pch ^ (BKT == 64 ? (prow & 7) : ((prow >> 2) & 3))) * 16;
    constexpr int PA_BASE = G::PA / NW, PA_REM = G::PA % NW;
    const int a_first = wave * PA_BASE + (wave < PA_REM ? wave : PA_REM);      // first A piece of this wave
    const int a_count = PA_BASE + (wave < PA_REM ? 1 : 0);                     // wave-uniform
    const int sgn = (p.mode == MI_GATHER_FWD) ? 1 : -1;                // FWD: src = out + tap*dil - pad ; DGRAD: out + pad - tap*dil
    // Pixel coordinates: a 1x1 conv needs none (the source row IS the output row); otherwise ONE pair of integer divisions per
    // lane for its first row, the following rows (RPP pixels further each) are stepped incrementally.
    const bool pointwise = UNIT && p.T == 1 && p.pad == 0;             // workgroup-uniform
    int cb_ = 0, cho = 0, cwo = 0;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + (a_first + i) * RPP + prow;
        const bool ok = (m < p.M) & (i < a_count);
        if (pointwise) {
            a_base[i] = reinterpret_cast<const char*>(p.A + (long)(ok ? m : 0) * p.Ca) + a_chunk;
            a_mask[i] = ok ? 1u : 0u;
            a_ho[i] = a_wo[i] = 0;
            a_img[i] = -1;
            continue;
        }
        if (!UNIT || i == 0) {
            const int mm = ok ? m : 0;
            cb_ = mm / HoWo;
            const int rem = mm - cb_ * HoWo;
            cho = rem / p.Wo;
            cwo = rem - cho * p.Wo;
        } else {
            cwo += RPP;
            while (cwo >= p.Wo) {
                cwo -= p.Wo;
                ++cho;
            }
            while (cho >= p.Ho) {
                cho -= p.Ho;
                ++cb_;
            }
        }
        const int b = cb_, ho = cho, wo = cwo;
        a_ho[i] = ho;
        a_wo[i] = wo;
        a_img[i] = ok ? b * p.Ha * p.Wa : -1;
        if (UNIT) {
            const int h0 = ho - sgn * p.pad, w0 = wo - sgn * p.pad;   // tap (0,0) source; Ha == Ho, Wa == Wo here
            a_base[i] = reinterpret_cast<const char*>(p.A + ((long)b * p.Ha * p.Wa + (long)h0 * p.Wa + w0) * p.Ca) + a_chunk;
            // validity of tap (ky, kx) is separable: 3 row bits x 3 column bits, no loop over taps (ksz is 1 or 3 here)
            unsigned rb = 0, cb = 0;
#pragma unroll
            for (int kq = 0; kq < 3; ++kq) {
                const bool live = kq < p.ksz;
                rb |= (unsigned)(live & ((unsigned)(h0 + sgn * kq * p.dil) < (unsigned)p.Ha)) << kq;
                cb |= (unsigned)(live & ((unsigned)(w0 + sgn * kq * p.dil) < (unsigned)p.Wa)) << kq;
            }
            unsigned msk;
            if (p.ksz == 1) msk = rb & cb & 1u;                                    // bit t = ky*ksz + kx
            else msk = ((rb & 1u) ? cb : 0u) | ((rb & 2u) ? cb << 3 : 0u) | ((rb & 4u) ? cb << 6 : 0u);
            a_mask[i] = ok ? msk : 0u;
        }
    }
    int w_row[NPW], w_chunk[NPW];
    const char* w_base[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int rl = (wave * NPW + i) * RPP + prow;                   // row within the weight tile
        w_row[i] = n0 + rl;
        w_chunk[i] = (pch ^ (BKT == 64 ? ((rl & 3) | (((rl >> 3) & 1) << 2)) : ((rl >> 3) & 3))) * 16;
        w_base[i] = reinterpret_cast<const char*>(p.Wp + (long)(w_row[i] < p.N ? w_row[i] : 0) * p.Ca) + w_chunk[i];
    }

    const int cpt = p.Ca / BKT;         // K-steps per tap
    const int nk = p.T * cpt;
    int ld_t = 0, ld_cc = 0;            // tap / chunk of the NEXT tile to stage
    const char* a_ptr[NPA];
    const char* w_ptr[NPW];
    int a_inc[NPA], w_inc[NPW];
    const long w_tap_bytes = (long)p.N * p.Ca * 2;

    auto set_tap = [&](int t) __attribute__((always_inline)) {
        const int ky = t / p.ksz, kx = t - ky * p.ksz;
        if (UNIT) {
            const long toff = (long)sgn * ((long)ky * p.dil * p.Wa + (long)kx * p.dil) * p.Ca * 2;   // wave-uniform
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const bool ok = (a_mask[i] >> t) & 1u;
                a_ptr[i] = ok ? a_base[i] + toff : zero;
                a_inc[i] = ok ? ROWB : 0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                int ha = 0, wa = 0;
                const bool ok = tap_src(p, a_ho[i], a_wo[i], ky, kx, ha, wa) && a_img[i] >= 0;
                const char* ptr = reinterpret_cast<const char*>(p.A + ((long)((ok ? a_img[i] : 0) + (ok ? ha : 0) * p.Wa + (ok ? wa : 0))) * p.Ca) + a_chunk;
                a_ptr[i] = ok ? ptr : zero;
                a_inc[i] = ok ? ROWB : 0;
            }
        }
        const long woff = (long)t * w_tap_bytes;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const bool ok = w_row[i] < p.N;
            w_ptr[i] = ok ? w_base[i] + woff : zero;
            w_inc[i] = ok ? ROWB : 0;
        }
    };
    set_tap(0);

    auto stage = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES + a_first * 1024;
        char* sb = smem + buf * STAGE_BYTES + ATILE_BYTES + wave * (NPW * 1024);
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            if (PA_REM == 0 || i < a_count) glds16(a_ptr[i], sa + i * 1024);
            a_ptr[i] += a_inc[i];
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
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
    const int wrow0 = wn * 64 + 8 * (frow >> 2) + (frow & 3);           // + 32*(i>>1) + 4*(i&1): permuted weight row of MFMA tile i
    const int flags = EPI >= 0 ? EPI : p.flags;
    bf16x8 pres[MT][2];
    unsigned pbits[MT];
    if (PREF) igemm_fetch_epilogue<MT, EPI>(p, m0, n0, wm, wn, frow, fq, pres, pbits);
    auto compute = [&](int buf) {
        const char* sa = smem + buf * STAGE_BYTES;
        const char* sb = sa + ATILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BKT / 32; ++kk) {
            // same expression for both operands (see the swizzle note above)
            const int sw = BKT == 64 ? (((kk * 4 + fq) ^ (frow & 7)) << 4) : ((fq ^ ((frow >> 2) & 3)) << 4);
            bf16x8 wf[4], af[MT];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + (wrow0 + 32 * (i >> 1) + 4 * (i & 1)) * ROWB + sw);
#pragma unroll
            for (int j = 0; j < MT; ++j) af[j] = *reinterpret_cast<const bf16x8*>(sa + (wm * (MT * 16) + j * 16 + frow) * ROWB + sw);
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

    if (EPI < 0 && (p.flags & (1 << 30))) {   // perf experiment: main loop only (keeps the accumulators alive, stores nothing)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if constexpr (STG) {
        constexpr bool RES = (EPI & MI_EPI_RESIDUAL) != 0, BITS_IN = (EPI & MI_EPI_BITMASK) != 0, BITS_OUT = (EPI & MI_EPI_WRITE_MASK) != 0;
        const bool mask_lds = (p.N & 127) == 0;          // uniform: 16-B aligned mask rows, no partial tile
        if (RES) igemm_residual_to_lds<MT>(p, m0, n0, wave, lane, zero, smem);
        if (BITS_IN) {
            if (mask_lds) igemm_mask_to_lds<MT>(p, m0, n0, tid, smem);
            else igemm_fetch_epilogue<MT, EPI, false>(p, m0, n0, wm, wn, frow, fq, pres, pbits);      // mask bytes straight to registers
        }
        if (RES || BITS_IN) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (RES) igemm_residual_from_lds<MT>(smem, wm, wn, frow, fq, pres);
            if (BITS_IN && mask_lds) igemm_mask_from_lds<MT>(smem, wm, wn, frow, fq, pbits);
        }
        igemm_epilogue<MT, EPI, true>(p, acc, m0, n0, wm, wn, frow, fq, pres, pbits, smem, mask_lds);
        __syncthreads();
        igemm_store_staged<MT>(p, m0, n0, tid, smem);
        if (BITS_OUT && mask_lds) igemm_store_mask_staged<MT>(p, m0, n0, tid, smem);
        return;
    }
    if (!PREF) igemm_fetch_epilogue<MT, EPI>(p, m0, n0, wm, wn, frow, fq, pres, pbits);
    igemm_epilogue<MT, EPI>(p, acc, m0, n0, wm, wn, frow, fq, pres, pbits);
}

inline bool staged_on() { return mi_sw().igemm_staged != 0; }        // MI_IGEMM_STAGED=0: MFMA-layout loads / stores straight from registers

template <int MT, bool UNIT, bool PREF, int BKT, int EPI, int NWN, bool STG = false>
void launch_one(dim3 grid, hipStream_t stream, const IgemmParams& p) {
    static std::atomic<uint64_t> attr_done{0};        // per instantiation, one bit per device
    auto kern = igemm_nt_kernel<MT, UNIT, PREF, BKT, EPI, NWN, STG>;
    constexpr int lds = Geo<MT, BKT, NWN>::LDS_BYTES;
    mi_allow_dynamic_lds((const void*)kern, lds, attr_done);
    hipLaunchKernelGGL(kern, grid, dim3(Geo<MT, BKT, NWN>::NW * 64), lds, stream, p);
}

// hot epilogue sets: 69 = FrozenBN + ReLU + sign bits (conv1 / conv2 forward), 71 = the same + residual (conv3 forward),
// 128 = ReLU-backward from sign bits (data gradients), 130 = the same + residual-gradient add (conv1 data gradient)
template <int MT, bool UNIT, bool PREF, int BKT, int NWN>
void launch_variant(dim3 grid, hipStream_t stream, const IgemmParams& p) {
    const int fl = p.flags;
    const bool staged = staged_on();
    if constexpr (UNIT) {
        if constexpr (!PREF && NWN == 2 && BKT == 64) {
            if (staged) {     // the residual sets only: 5-7 % faster staged; without a residual tile to read it is a wash
                if (fl == 71) return launch_one<MT, UNIT, PREF, BKT, 71, NWN, true>(grid, stream, p);
                if (fl == 130) return launch_one<MT, UNIT, PREF, BKT, 130, NWN, true>(grid, stream, p);
            }
        }
        if constexpr (!PREF) {
            if (fl == 69) return launch_one<MT, UNIT, PREF, BKT, 69, NWN>(grid, stream, p);
            if (fl == 128) return launch_one<MT, UNIT, PREF, BKT, 128, NWN>(grid, stream, p);
        }
        if (fl == 71) return launch_one<MT, UNIT, PREF, BKT, 71, NWN>(grid, stream, p);
        if (fl == 130) return launch_one<MT, UNIT, PREF, BKT, 130, NWN>(grid, stream, p);
    }
    if constexpr (!PREF && NWN == 2) {
        if (fl == MI_EPI_STATS) return launch_one<MT, UNIT, PREF, BKT, MI_EPI_STATS, NWN>(grid, stream, p);      // plain store + BatchNorm tile statistics
    }
    launch_one<MT, UNIT, PREF, BKT, -1, NWN>(grid, stream, p);
}

}  // namespace

extern "C" int mi_conv_gemm_pp(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize,
                               int stride, int pad, int dil, int gather_mode, const float* scale, const float* bias, const void* res,
                               const void* msk, void* mask_out, int flags, int zgw, float alpha, int mtg, void* stream);

// 1 = the launch goes to the wide-tile ping-pong main loop (igemm_pp_kernel), 0 = to igemm_nt_kernel
extern "C" int mi_conv_gemm_route(int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride, int flags) {
    const int pp_on = mi_sw().igemm_pp;
    const int min_k = mi_sw().igemm_pp_mink;         // 512: in the step 704 -> 2048 (ASPP data gradient) 246 vs 280 us, 512 -> 1024 101 vs 116; K = 256: no difference
    const long M = (long)B * Ho * Wo;
    // igemm_pp_kernel addresses its operands with 32-bit buffer offsets: 2 GiB tensors stay on the pointer-arithmetic kernel
    const long a_bytes = (long)B * Ha * Wa * Ca * 2 + 64L * (Wa + 1) * Ca * 2, w_bytes = (long)ksize * ksize * N * Ca * 2;
    if (a_bytes >= (1L << 31) - (1L << 20) || w_bytes >= (1L << 31) - (1L << 20)) return 0;
    return pp_on && stride == 1 && Ha == Ho && Wa == Wo && !(flags & (MI_EPI_RESIDUAL | MI_EPI_MASK | MI_EPI_LEAKY)) && (long)ksize * ksize * Ca >= min_k &&
           Ca % 32 == 0 && M >= 320 * 64 && N >= 256;
}

extern "C" int mi_conv_gemm(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                            int ksize, int stride, int pad, int dil, int gather_mode, const float* scale, const float* bias,
                            const void* res, const void* msk, void* mask_out, int flags, int zgw, float alpha, void* stream) {
    return mi_conv_gemm_impl(a, wp, out, B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, gather_mode, scale, bias, res, msk, mask_out, flags, zgw, alpha,
                             stream, nullptr);
}

extern "C" size_t mi_conv_gemm_stats_workspace(long M, int N) {
    const long rows = (M + 63) / 64 + 2;               // the shortest wave row tile is 64 rows
    return ((size_t)rows * 2 * (size_t)N + mi_bn_reduce_tmp_floats((int)rows, N)) * sizeof(float);
}

extern "C" int mi_conv_gemm_stats(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride,
                                  int pad, int dil, const float* pilot, float* sums, void* workspace, size_t workspace_bytes, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                  float* out4, void* stream) {
    MI_REQUIRE(pilot && sums && workspace && mi_aligned16(pilot) && mi_aligned16(workspace), "mi_conv_gemm_stats: null or unaligned statistics operand");
    MI_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && N > 0 && N % 8 == 0, "mi_conv_gemm_stats: N=%d", N);
    const long M = (long)B * Ho * Wo;
    MI_REQUIRE(workspace_bytes >= mi_conv_gemm_stats_workspace(M, N), "mi_conv_gemm_stats: workspace of %zu bytes, need %zu", workspace_bytes,
               mi_conv_gemm_stats_workspace(M, N));
    MI_REQUIRE((gamma == nullptr) == (out4 == nullptr), "mi_conv_gemm_stats: gamma and out4 go together (finalize in the same call) or are both NULL");
    MiConvStats st{(float*)workspace, pilot, 0};
    const int rc = mi_conv_gemm_impl(a, wp, out, B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, MI_GATHER_FWD, nullptr, nullptr, nullptr, nullptr, nullptr,
                                     MI_EPI_STATS, 0, 0.f, stream, &st);
    if (rc != MI_OK) return rc;
    float* tmp = (float*)workspace + ((M + 63) / 64 + 2) * 2 * (long)N;
    const MiBnFinal fin{pilot, (double)M, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, out4};
    return mi_bn_reduce_partials((const float*)workspace, st.nparts, N, tmp, sums, sums + N, gamma ? &fin : nullptr, stream);
}

int mi_conv_gemm_impl(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride, int pad,
                      int dil, int gather_mode, const float* scale, const float* bias, const void* res, const void* msk, void* mask_out, int flags,
                      int zgw, float alpha, void* stream, MiConvStats* st) {
    MI_REQUIRE(a && wp && out, "mi_conv_gemm: null operand");
    MI_REQUIRE(((flags & MI_EPI_STATS) != 0) == (st != nullptr) && (!st || flags == MI_EPI_STATS), "mi_conv_gemm: MI_EPI_STATS comes alone, through mi_conv_gemm_stats");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0 && N > 0, "mi_conv_gemm: non-positive dimension");
    // Ca % 64 == 0: either main loop.  Ca % 32 == 0 only (the 32-padded operands of the general family's packs: HarDNet's 466 -> 480 channel
    // gathers): the 32-channel slabs of igemm_pp_kernel take it, whatever the dispatch rule says - stride-1 / same-size launches without a residual tile
    const bool pp_only = Ca > 0 && Ca % 64 != 0;
    MI_REQUIRE(Ca > 0 && Ca % 32 == 0, "mi_conv_gemm: Ca=%d must be a multiple of 32", Ca);
    MI_REQUIRE(!pp_only || (stride == 1 && Ha == Ho && Wa == Wo && !(flags & (MI_EPI_RESIDUAL | MI_EPI_MASK | MI_EPI_LEAKY))),
               "mi_conv_gemm: Ca=%d is not a multiple of 64: stride-1, same-size launches without residual / mask / LeakyReLU only", Ca);
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
    // N % 128 == 0: the staged epilogue moves a tile row's 16 mask bytes at once (otherwise masks are touched bytewise)
    MI_REQUIRE(!(flags & MI_EPI_BITMASK) || N % 128 != 0 || mi_aligned16(msk), "mi_conv_gemm: the packed sign bits must be 16-byte aligned");
    MI_REQUIRE(!(flags & MI_EPI_WRITE_MASK) || N % 128 != 0 || mi_aligned16(mask_out), "mi_conv_gemm: mask_out must be 16-byte aligned");
    MI_REQUIRE(!(flags & MI_EPI_ZSPLIT) || (zgw > 0 && zgw % 4 == 0 && N % zgw == 0), "mi_conv_gemm: zgw");
    const long M = (long)B * Ho * Wo;
    MI_REQUIRE(M < (1L << 31) && (long)B * Ha * Wa < (1L << 31), "mi_conv_gemm: pixel count overflows int32");
    if (gather_mode == MI_GATHER_FWD) {
        MI_REQUIRE((Ho - 1) * stride - pad < Ha && (Wo - 1) * stride - pad < Wa, "mi_conv_gemm: output larger than the input supports");
    }
    // Long contractions without a residual tile go to the wide-tile ping-pong main loop (igemm_pp.hip): 7.0 instead of 13-14
    // L2 bytes per kFLOP.  Measured per shape against this kernel in one process (tools/ppexp.py, B = 8, 97 x 97): 3x3 256 +16 %,
    // 3x3 512 +13 %, 1x1 2048->512 +18 %, 1x1 1024->256 / 1024->2048 / ASPP forward +6 %; the short contractions (K < 512)
    // are epilogue-bound and stay here.  MI_IGEMM_PP=0 switches the dispatch off.
    if (pp_only) {
        MI_REQUIRE((long)B * Ha * Wa * Ca * 2 + 2L * ((long)(ksize - 1) * dil + pad) * (Wa + 1) * Ca * 2 < (1L << 31) - (1L << 20) &&
                       (long)ksize * ksize * N * Ca * 2 < (1L << 31) - (1L << 20),
                   "mi_conv_gemm: Ca=%d is not a multiple of 64 and the operands exceed the 2 GiB the 32-channel main loop addresses", Ca);
        return mi_conv_gemm_pp_impl(a, wp, out, B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, gather_mode, scale, bias, res, msk, mask_out, flags, zgw,
                                    alpha, 0, stream, st);
    }
    if (mi_conv_gemm_route(B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, flags) &&
        (long)B * Ha * Wa * Ca * 2 + 2L * ((long)(ksize - 1) * dil + pad) * (Wa + 1) * Ca * 2 < (1L << 31) - (1L << 20))     // its exact 32-bit offset bound
        return mi_conv_gemm_pp_impl(a, wp, out, B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, gather_mode, scale, bias, res, msk, mask_out, flags, zgw,
                                    alpha, 0, stream, st);
    IgemmParams p;
    p.korder = 0;
    p.stats = st ? st->partial : nullptr;
    p.pilot = st ? st->pilot : nullptr;
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
    const bool unit = stride == 1 && Ha == Ho && Wa == Wo && ksize * ksize <= 9;
    const int force_mt = mi_sw().igemm_mt, force_bn = mi_sw().igemm_bn, pref_on = mi_sw().igemm_pref;
    // Modelled time of one launch with tile bm x bn: rounds on the resident-workgroup slots x workgroups sharing a CU x per-K-step
    // tile time, where a CU's share of the L2 request rate serves (bm+bn)*128 B per step (~47.6 GB/s per CU measured) and its
    // MFMA pipes need bm*bn*128 FLOP at ~9.8 TFLOP/s per CU; the two overlap imperfectly (20 % of the shorter one is exposed).
    auto cost = [&](int bm, int bn) {
        const int per_cu = bn == 256 ? 1 : 2;
        const long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        const long rounds = (tiles + 256 * per_cu - 1) / (256 * per_cu);
        const double t_l2 = (bm + bn) * 128.0 / 47.6e3, t_mfma = bm * (double)bn * 128.0 / 9.77e6;      // microseconds
        return rounds * per_cu * ((t_l2 > t_mfma ? t_l2 : t_mfma) + 0.2 * (t_l2 > t_mfma ? t_mfma : t_l2));
    };
    // The 256-wide tile is selectable (MI_IGEMM_BN=256) but not the default: with this loop structure (one barrier and a full
    // vmcnt(0) drain per K-step) it measured 2-8 % slower than the 128-wide tiles on every shape of the network (bench 216 vs
    // 223.5 images/s) although it moves 30 % fewer L2 bytes - the single resident workgroup has nothing to overlap its drains
    // with.  It is the geometry a deeper-pipelined schedule (counted vmcnt, prefetch in flight across barriers) needs.
#ifdef MI_EXPERIMENTS
    const bool wide_ok = unit && N % 256 == 0 && !(flags & MI_EPI_ZSPLIT) && force_bn == 256;
#else
    const bool wide_ok = false;                          // the 256-wide tile is compiled into experiment builds only
    (void)force_bn;
#endif
    int mt_sel = 4, bn = 128;
    double best = cost(128, 128);
    for (int mt = 5; mt <= 6; ++mt)
        if (cost(32 * mt, 128) < best * 0.999) best = cost(32 * mt, 128), mt_sel = mt;
    if (force_mt >= 4 && force_mt <= 6) mt_sel = force_mt, best = cost(32 * mt_sel, 128);
    if (wide_ok) {
        int wmt = 0;
        double wbest = force_bn == 256 ? 1e30 : best;
        for (int mt = 4; mt <= 6; ++mt)
            if ((force_mt == 0 || force_mt == mt) && cost(32 * mt, 256) < wbest * 0.999) wbest = cost(32 * mt, 256), wmt = mt;
        if (wmt) mt_sel = wmt, bn = 256;
    }
    const bool hot_set = flags == 71 || flags == 130;
    const bool pref = bn == 128 && pref_on && unit && (flags & MI_EPI_RESIDUAL) && N % 16 == 0 && !(staged_on() && hot_set);
    if (!unit) mt_sel = 4;                               // the general (strided) gather exists in the 128-row shape only
    if (pref && mt_sel == 6) mt_sel = 5;                 // the prefetched rows cost 8 VGPRs per 16-row MFMA tile
    const int bm = mt_sel * 32;
    p.m_tiles = (int)((M + bm - 1) / bm);
    p.n_tiles = (N + bn - 1) / bn;
    const dim3 grid(p.m_tiles * p.n_tiles);
    if (st) st->nparts = 2 * p.m_tiles;                  // one partial row per wave row (MT*16 rows)
    const hipStream_t sq = (hipStream_t)stream;
    if (!unit)
        launch_variant<4, false, false, 64, 2>(grid, sq, p);
#ifdef MI_EXPERIMENTS
    else if (bn == 256 && mt_sel == 6)
        launch_variant<6, true, false, 64, 4>(grid, sq, p);
    else if (bn == 256 && mt_sel == 5)
        launch_variant<5, true, false, 64, 4>(grid, sq, p);
    else if (bn == 256)
        launch_variant<4, true, false, 64, 4>(grid, sq, p);
#endif
    else if (pref && mt_sel == 5)
        launch_variant<5, true, true, 64, 2>(grid, sq, p);
    else if (pref)
        launch_variant<4, true, true, 64, 2>(grid, sq, p);
    else if (mt_sel == 6)
        launch_variant<6, true, false, 64, 2>(grid, sq, p);
    else if (mt_sel == 5)
        launch_variant<5, true, false, 64, 2>(grid, sq, p);
    else
        launch_variant<4, true, false, 64, 2>(grid, sq, p);
    MI_CHECK_LAUNCH("mi_conv_gemm");
    return MI_OK;
}
