// General implicit-GEMM convolution for gfx950: any kh x kw taps, per-axis stride / padding / dilation, any channel counts,
// operands given as channel-slice VIEWS of NHWC bf16 tensors (pointer to channel 0 of the slice + elements per pixel row),
// so that torch.split / torch.cat of the reference's Res2Net bottleneck (core/models/classifiers/pranet/Res2Net_v1b.py:68-84)
// and of the RFB / partial decoder (PraNet_Res2Net.py:50-57,84-91) are views, never copies.
//
//   forward    out[m][n] = sum_{t,c} a[src(m,t)][c] * wp[t][n][c]  (+ bias[n])      m = (b,ho,wo)
//   data grad  the same kernel in gather mode MI_GATHER_DGRAD on the transposed pack
//   weight grad  dw[o][i][t] = sum_m dy[m][o] * x[src(m,t)][i]   (gwgrad_kernel: split-K slabs + fixed-order reducer)
//
// Replaces nn.Conv2d forward / convolution_backward for every conv of the PraNet path: 26/52/104/208-channel
// 3x3 group convs, 1x3 / 3x1 / 1x5 / 5x1 / 1x7 / 7x1 and dilated 3 / 5 / 7 convs of RFB_modified, 5x5 reverse-attention convs,
// the 3-channel stem conv and the one-channel side outputs.  Channel counts that are not multiples of the 32-channel K chunk
// are padded in LDS (zero fill on load), never in HBM.
//
// Tile: 128 pixels x BN channels (BN = 16 | 32 | 64 | 80 | 112 | 128, chosen per launch by a cost model: see mi_gconv), 4 waves, each 32 rows x BN columns of v_mfma_f32_16x16x32_bf16;
// operands are register-staged (global -> VGPR -> LDS, double buffered): the sources are arbitrary-alignment slices, which the
// LDS-DMA path of igemm_nt.hip (16-byte granules) cannot fetch.  The epilogue stages the tile in LDS, stores rows with
// the widest access the view's alignment allows, and (optionally) emits per-tile column sums / sums of squares of the ROUNDED
// outputs - the first level of the BatchNorm batch statistics (nn.BatchNorm2d in train(), PraNet_Res2Net.py:13,17-19) - so
// that no separate pass over the conv output is needed for them.
#include "mi_common.h"
#include <type_traits>
#include <atomic>
#include <vector>
#include <string.h>
#include <stdlib.h>

// Measurement switches (MI_GC_DBG / MI_GW_DBG: skip the main loop, the statistics or the stores of a launch to price its phases) exist only in experiment builds
// (tools/experiments/build.sh, -DMI_EXPERIMENTS): in the product library the tests below are the constant 0 and the environment variables are not read - an
// exported variable cannot corrupt a training run.
#ifdef MI_EXPERIMENTS
#define MI_DBG_BIT(p, b) ((p).dbg & (b))
#else
#define MI_DBG_BIT(p, b) 0
#endif

namespace {

constexpr int GBM = 128;       // pixels per tile
// Two register sets of operand prefetch: a K step stashes the next chunk into LDS while the chunk after next is in flight.  More sets do not pay: three
// sets everywhere (96 VGPRs + 96 AGPRs instead of 80 + 48, two workgroups per CU instead of four) took PraNet 795 -> 731 images/s and GALD 130 -> 117,
// four sets for the launches of at most two workgroups per CU (three K steps for a load to arrive) 819 -> 810: a K step of these launches is not
// waiting for its loads but for its own serial chain (LDS write -> barrier -> LDS read -> MFMAs -> ~85 address instructions), 600-800 ns with one
// workgroup per CU (tools/gkshape.py).
// K chunk per main-loop step: KC = 32 channels (one MFMA k) or 64 (two; half as many steps, twice the bytes in flight per step - the main loop
// of these small convs is bound by the round trip of a step's loads, not by the matrix pipe); LDS row stride KC + 8 elements (80 / 144 B)

struct GConvP {
    const __bf16* A;
    const __bf16* Wp;
    void* out;
    const float* bias;
    float* stats;
    long lda, ldo;
    int M, N, Ca, Cpad, Npad, T;
    int Ha, Wa, Ho, Wo;
    int kw, sh, sw, ph, pw, dh, dw, mode, nchunks, remap, dbg;
    // in-launch BatchNorm finalize (fin_out != null; the launch has at most MI_INLAUNCH_MAX_PARTS row tiles): the arguments of mi_gbn_finalize
    unsigned* fin_ticket;      // one zeroed word per column tile
    float* fin_out;            // [4][N]: mean, invstd, scale, shift
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    double count;
    float momentum, eps;
};

__device__ __attribute__((aligned(256))) uint32_t g_gzero[64];        // source of every padded / out-of-range access (zero-initialised)

// Loads are UNCONDITIONAL (an out-of-range access reads the zero page instead of being branched around): with control flow around a load
// the compiler cannot count the loads in flight and waits for all of them (s_waitcnt vmcnt(0)) at the first use - which also drains the
// prefetch of the chunk after next, i.e. every K step would pay a full memory round trip.
// 16 bytes from a 4-byte aligned address (a channel slice whose offset / row stride is even but not a multiple of 8 elements: HarDNet's 466-, 142-,
// 68-channel tensors, Res2Net's 26-channel groups): ONE global_load_dwordx4 - the hardware takes dword-aligned addresses for it - instead of four
// dword loads.  8 channels c0 .. c0+7 of a row of C (even, >= 8) channels: a chunk that would run past the row end is fetched as the row's LAST 16
// bytes and shifted down (zero fill), so that nothing outside [src, src + C) is ever read; channels >= C and !ok rows read the zero page.
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;

__device__ __forceinline__ bf16x8 gload8_a4(const __bf16* src, int c0, int C, bool ok) {
    const int rem = C - c0;
    const bool live = ok && rem > 0, part = live && rem < 8;
    const __bf16* ptr = live ? (part ? src + C - 8 : src + c0) : reinterpret_cast<const __bf16*>(g_gzero);
    const u32x4 v = *reinterpret_cast<const u32x4_a4*>(ptr);
    // (selects, no branch: control flow between a load and its use costs the main loop its load counting)
    const int drop = part ? (8 - rem) >> 1 : 0;          // dwords of the window that belong to channels below c0
    u32x4 r;
    r[0] = drop == 0 ? v[0] : (drop == 1 ? v[1] : (drop == 2 ? v[2] : v[3]));
    r[1] = drop == 0 ? v[1] : (drop == 1 ? v[2] : (drop == 2 ? v[3] : 0u));
    r[2] = drop == 0 ? v[2] : (drop == 1 ? v[3] : 0u);
    r[3] = drop == 0 ? v[3] : 0u;
    return __builtin_bit_cast(bf16x8, r);
}

template <int VEC>
__device__ __forceinline__ void gload16(const __bf16* src, int c0, int C, bool ok, bf16x8 (&r)[2]) {
    // 16 channels c0 .. c0+15 of one pixel row; channels >= C (and everything when !ok) read as zero
    const __bf16* zero = reinterpret_cast<const __bf16*>(g_gzero);
    if constexpr (VEC == 8) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + 8 * h;
            r[h] = *reinterpret_cast<const bf16x8*>((ok && c < C) ? src + c : zero);
        }
    } else if constexpr (VEC == 4) {
        r[0] = gload8_a4(src, c0, C, ok);
        r[1] = gload8_a4(src, c0 + 8, C, ok);
    } else {
        union { bf16x8 v[2]; uint16_t u[16]; } x;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = c0 + j;
            x.u[j] = *reinterpret_cast<const uint16_t*>((ok && c < C) ? src + c : zero);
        }
        r[0] = x.v[0];
        r[1] = x.v[1];
    }
}

template <int BN, int KC, bool OUTF32>
struct GSmem {
    static constexpr int AB = 2 * (GBM + BN) * (KC + 8) * 2;                          // double-buffered operand tiles
    static constexpr int CS = OUTF32 ? GBM * (BN + 4) * 4 : GBM * (BN + 8) * 2;       // staged output tile
    static constexpr int RG = BN <= 32 ? 8 : (BN <= 64 ? 4 : 2);                      // row groups of the statistics pass (a power of two, RG * BN <= 256)
    static constexpr int RED = 2 * RG * BN * 4;                                       // stats partials of the row groups
    static constexpr int BYTES = (AB > CS + RED ? AB : CS + RED);
};

// The conv epilogue (shared by gconv_kernel and gconv3_kernel): accumulators (+ bias) -> LDS image of the tile -> row stores, the column statistics of the
// rounded values, and (small launches) the in-launch BatchNorm finalize.  acc[i][j]: rows wave * 32 + i * 16 + (lane & 15) hold FOUR CONSECUTIVE CHANNELS
// j * 16 + (lane >> 4) * 4 .. + 3 per lane (MFMA with the weights as the first operand).
template <int BN, int OVEC, bool OUTF32>
__device__ __forceinline__ void gconv_epilogue(const GConvP& p, char* smem, f32x4 (&acc)[2][BN / 16], int tid, int m0, int n0, int mt, int nt, int m_tiles) {
    constexpr int NT = BN / 16;
    const int lane = tid & 63, wave = tid >> 6, frow = lane & 15;
    // ---- epilogue: (+ bias) -> LDS image of the tile -> row stores (+ column statistics of the rounded values)
    const int fq = lane >> 4;
    if constexpr (OUTF32) {
        float* Cs = reinterpret_cast<float*>(smem);
        constexpr int CSW = BN + 4;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + j * 16 + fq * 4 + r;
                bv[r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<f32x4*>(Cs + (wave * 32 + i * 16 + frow) * CSW + j * 16 + fq * 4) = f32x4{acc[i][j][0] + bv[0], acc[i][j][1] + bv[1], acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]};
        }
        __syncthreads();
        float* out = reinterpret_cast<float*>(p.out);
        for (int idx = tid; idx < GBM * BN; idx += 256) {
            const int row = idx / BN, col = idx - row * BN;
            const int m = m0 + row, n = n0 + col;
            if (m < p.M && n < p.N) out[(long)m * p.ldo + n] = Cs[row * CSW + col];
        }
    } else {
        __bf16* Cs = reinterpret_cast<__bf16*>(smem);
        constexpr int CSW = BN + 8;
#pragma unroll
        for (int j = 0; j < NT; ++j) {                       // (one 8-byte LDS write per 16 x 16 block and lane: its four consecutive channels of pixel row frow)
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + j * 16 + fq * 4 + r;
                bv[r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                union { __bf16 h[4]; uint64_t u; } pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk.h[r] = (__bf16)(acc[i][j][r] + bv[r]);
                *reinterpret_cast<uint64_t*>(Cs + (wave * 32 + i * 16 + frow) * CSW + j * 16 + fq * 4) = pk.u;
            }
        }
        __syncthreads();
        __bf16* out = reinterpret_cast<__bf16*>(p.out);
        constexpr int OW = OVEC == 4 ? 8 : OVEC;     // channels per store slot
        constexpr int G = BN / OW;                   // store slots per row
        if constexpr (OVEC == 8) {
            // (the tile's LDS reads are issued together, then the stores: a read inside the bounds branch waits for itself before every store)
            constexpr int TRIPS = GBM * G / 256;
            bf16x8 v[TRIPS];
#pragma unroll
            for (int k = 0; k < TRIPS; ++k) {
                const int idx = tid + k * 256;
                const int row = idx / G, cg = idx - row * G;
                v[k] = *reinterpret_cast<const bf16x8*>(Cs + row * CSW + cg * OW);
            }
#pragma unroll
            for (int k = 0; k < TRIPS; ++k) {
                const int idx = tid + k * 256;
                const int row = idx / G, cg = idx - row * G;
                const int m = m0 + row, n = n0 + cg * OW;
                if (m < p.M && n < p.N && !MI_DBG_BIT(p, 4)) *reinterpret_cast<bf16x8*>(out + (long)m * p.ldo + n) = v[k];
            }
        } else
        for (int idx = tid; idx < GBM * G; idx += 256) {
            const int row = idx / G, cg = idx - row * G;
            const int m = m0 + row, n = n0 + cg * OW;
            if (m < p.M && n < p.N && !MI_DBG_BIT(p, 4)) {
                __bf16* dst = out + (long)m * p.ldo + n;
                const __bf16* src = Cs + row * CSW + cg * OW;
                if constexpr (OVEC == 8) *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(src);
                else if constexpr (OVEC == 4) {       // 4-byte aligned rows, even N: one 16-byte store per full chunk, dword stores for the row's tail
                    if (n + 8 <= p.N) {
                        *reinterpret_cast<u32x4_a4*>(dst) = *reinterpret_cast<const u32x4*>(src);
                    } else {
                        for (int e = 0; n + e < p.N; e += 2) *reinterpret_cast<uint32_t*>(dst + e) = *reinterpret_cast<const uint32_t*>(src + e);
                    }
                } else *dst = *src;
            }
        }
        if (p.stats && !MI_DBG_BIT(p, 2)) {
            // thread -> (column, row group): fixed-order sums over the group's rows, then over the groups
            constexpr int RG = GSmem<BN, 32, false>::RG;         // 8 | 4 | 2 row groups
            constexpr int RPG = GBM / RG;
            float* red = reinterpret_cast<float*>(smem + GSmem<BN, 32, false>::CS);
            const int col = tid % BN, rg = tid / BN;
            if (rg < RG) {
                float s1 = 0.f, s2 = 0.f;
                // (unconditional LDS reads + a select: with the read inside the row-bound branch every one of the 16 - 64 rows was its own LDS round trip -
                //  15 of the 63 us of a 104 -> 256 conv at 16 x 88 x 88; adding 0 for a row past M leaves the sums' bits as they were)
#pragma unroll 8
                for (int r = 0; r < RPG; ++r) {
                    const int row = rg * RPG + r;
                    const float raw = (float)Cs[row * CSW + col];
                    const float v = m0 + row < p.M ? raw : 0.f;
                    s1 += v;
                    s2 += v * v;
                }
                red[(rg * 2 + 0) * BN + col] = s1;
                red[(rg * 2 + 1) * BN + col] = s2;
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.N) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    s1 += red[(g * 2 + 0) * BN + tid];
                    s2 += red[(g * 2 + 1) * BN + tid];
                }
                float* st = p.stats + (long)mt * 2 * p.N;
                if (p.fin_out) {                       // handed to the launch's last workgroup: write-through
                    mi_st_sc1(st + n0 + tid, s1);
                    mi_st_sc1(st + p.N + n0 + tid, s2);
                } else {
                    st[n0 + tid] = s1;
                    st[p.N + n0 + tid] = s2;
                }
            }
            if constexpr (BN == 32 || BN == 64) {
            if (p.fin_out) {
                // The column tile's last workgroup turns the row tiles' sums into mean / invstd / folded affine and updates the running statistics with
                // mi_gbn_finalize's arithmetic in its order - for at most MI_INLAUNCH_MAX_PARTS = 64 row tiles (the launcher's limit) every tile is its own lane and the lanes are added in ascending order
                // (double): the same bits as the separate launch.  The loads of a channel are spread over the L = 256 / BN threads that share it (a few
                // independent loads each, no register arrays that would cost the main loop its occupancy) and meet in LDS, one statistic at a time
                // ([64][BN] floats: 16 KB at BN = 64); one thread per channel adds them.
                if (mi_last_arriver(p.fin_ticket + nt, m_tiles, reinterpret_cast<int*>(smem))) {
                    mi_acquire_partials();
                    constexpr int L = 256 / BN;
                    float* lanes = reinterpret_cast<float*>(smem + 16);                       // [MI_INLAUNCH_MAX_PARTS][BN]
                    const int ch = tid % BN, j = tid / BN, tiles = m_tiles;
                    const int cc = n0 + ch;
                    double tot[2] = {0.0, 0.0};
#pragma unroll
                    for (int which = 0; which < 2; ++which) {
                        if (cc < p.N) {
#pragma unroll
                            for (int q = 0; q < MI_INLAUNCH_MAX_PARTS / L; ++q) {
                                const int t = j + q * L;
                                lanes[t * BN + ch] = p.stats[(long)(t < tiles ? t : tiles - 1) * 2 * p.N + which * p.N + cc];
                            }
                        }
                        __syncthreads();
                        if (tid < BN && n0 + tid < p.N) {
                            double sum = 0.0;
                            for (int t = 0; t < tiles; ++t) sum += (double)lanes[t * BN + tid];
                            tot[which] = sum;
                        }
                        __syncthreads();
                    }
                    const int c = n0 + tid;
                    if (tid < BN && c < p.N) {
                        const double s1 = tot[0], s2 = tot[1];
                        const MiBnFin f = mi_bn_finalize_channel(s1, s2, p.count, p.eps, p.gamma ? p.gamma[c] : 1.f, p.beta ? p.beta[c] : 0.f);
                        p.fin_out[c] = f.mean;
                        p.fin_out[p.N + c] = f.invstd;
                        p.fin_out[2 * p.N + c] = f.scale;
                        p.fin_out[3 * p.N + c] = f.shift;
                        if (p.running_mean) {
                            p.running_mean[c] = mi_bn_running(p.running_mean[c], p.momentum, f.mean);
                            p.running_var[c] = mi_bn_running(p.running_var[c], p.momentum, mi_bn_unbiased(f.var, p.count));
                        }
                    }
                }
            }
            }
        }
    }
}

// KS = 2 (launches that put at most ~one workgroup on a CU: the 22 x 22 / 11 x 11 maps): 512 threads, two wave groups that each run this main loop
// over every other K chunk of the SAME tile through their own LDS stages - half as many serial K steps with nobody to share the CU with anyway - and meet
// once at the end: group 1 hands its accumulators over through LDS (group 0's sum + group 1's sum, a fixed order), leaves, and group 0 runs the epilogue.
template <int BN, int KC, int AVEC, int OVEC, bool OUTF32, bool GEN = false, int KS = 1>
__global__ __launch_bounds__(256 * KS) void gconv_kernel(GConvP p) {
    constexpr int NT = BN / 16, GRS = KC + 8, AQ = KC / 32;
    char* smem;
    if constexpr (KS == 1) {
        __shared__ __attribute__((aligned(16))) char smem_static[GSmem<BN, KC, OUTF32>::BYTES];
        smem = smem_static;
    } else {
        extern __shared__ __attribute__((aligned(16))) char smem_dynamic[];          // KS * AB bytes (the launcher asks for them)
        smem = smem_dynamic;
    }
    const int grp = KS > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;      // wave group (wave-uniform)
    __bf16* As = reinterpret_cast<__bf16*>(smem + grp * GSmem<BN, KC, OUTF32>::AB);        // [2][GBM][GRS]
    __bf16* Bs = As + 2 * GBM * GRS;                                      // [2][BN][GRS]
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;                   // (position inside the wave group)
    // Tile order: the hardware dispatches workgroups x-fastest and round-robin over the 8 XCDs (each with its own L2).  Remapped so that one XCD
    // owns a CONTIGUOUS run of row tiles and walks the column tiles of a row tile back to back: the column tiles' re-reads of the A rows and the
    // halo rows that neighbouring row tiles of a 3x3 share are served by that XCD's L2 instead of being fetched once per XCD (MI_GCONV_REMAP=0:
    // the plain (x, y) order).
    int mt = blockIdx.x, nt = blockIdx.y;
    const int m_tiles = gridDim.x;
    if (p.remap) {
        const int n_tiles = gridDim.y;
        const int logical = mi_xcd_remap(blockIdx.y * m_tiles + blockIdx.x, m_tiles * n_tiles);
        mt = logical / n_tiles;
        nt = logical - mt * n_tiles;
    }
    const int m0 = mt * GBM, n0 = nt * BN;

    // A loader: thread -> (row, half): KC / 2 channels of one pixel per K chunk
    const int arow = tid >> 1, ahalf = tid & 1;
    const int am = m0 + arow;
    const bool am_ok = am < p.M;
    int ab, aoh, aow;
    {
        const int hw = p.Ho * p.Wo;
        const int mm = am_ok ? am : 0;
        ab = mm / hw;
        const int rem = mm - ab * hw;
        aoh = rem / p.Wo;
        aow = rem - aoh * p.Wo;
    }
    // B loader: BN rows x KC / 8 chunks of 8 channels, chunk index = tid + 256 j -> (row, chunk)
    constexpr int BCH = KC / 8, BLOADS = BN * BCH, BROWS = (BLOADS + 255) / 256;

    // two register sets: the chunk after next is in flight while the next one waits in registers and the current one is in LDS
    bf16x8 ra0[2 * AQ], rb0[BROWS], ra1[2 * AQ], rb1[BROWS];
    const int total_all = MI_DBG_BIT(p, 1) ? 0 : p.T * p.nchunks;
    const int total = (total_all + KS - 1) / KS;          // K steps of this wave group: it takes the chunks grp, grp + KS, ...
    // load() is called for it = 0, 1, 2, ... in order: (tap, chunk, ky, kx) advance with it instead of being divided out of it on every call
    // (counters of one launch: 113 VALU + 143 SALU instructions per wave and K step beside 16 MFMAs - the waves were issuing index arithmetic half
    // of their time)
    // The main loop below is ONE basic block: every tap / chunk counter update, every padding test and every tail is a select, never a branch.
    // With control flow in the loop (the first version: mode and stride branches around the address arithmetic, a predicated LDS write, a second
    // loop exit) hipcc split it into 20 blocks, shuffled the accumulators through VGPRs at the block boundaries and - those VGPRs being load
    // destinations - put s_waitcnt vmcnt(0) at the loop header: the prefetch in flight was drained every second K step.
    // GEN = false: source pixel = base + tap * step (forward with any stride; data gradient of a stride-1 conv);  GEN = true: data gradient of a
    // strided conv (the tap's source exists only where the division is exact).
    int l_tap = 0, l_kc = 0, l_ky = 0, l_kx = 0;
    const bool fwd = p.mode == MI_GATHER_FWD;
    const int th = fwd ? p.dh : -p.dh, tw = fwd ? p.dw : -p.dw;
    const int base_h = fwd ? aoh * p.sh - p.ph : aoh + p.ph, base_w = fwd ? aow * p.sw - p.pw : aow + p.pw;
    const __bf16* arow0 = p.A + (((long)ab * p.Ha + base_h) * p.Wa + base_w) * p.lda;      // this thread's source row for tap (0, 0) (an address only: used where the tap lies inside the image)
    auto advance = [&]() {                                 // (tap, chunk, ky, kx) one chunk further
        const bool wrap = l_kc + 1 == p.nchunks;
        const bool roww = wrap && l_kx + 1 == p.kw;
        l_kc = wrap ? 0 : l_kc + 1;
        l_tap += wrap ? 1 : 0;
        l_kx = roww ? 0 : (wrap ? l_kx + 1 : l_kx);
        l_ky += roww ? 1 : 0;
    };
    if (KS > 1 && grp) advance();                          // group 1 starts at chunk 1
    long boff[BROWS];                                      // this thread's weight rows: offset inside a tap's [Npad][Cpad] plane, channel offset, row in range
    int bch8[BROWS];
    bool brow_ok[BROWS];
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
        const int idx = (tid + j * 256) % BLOADS;          // (BLOADS < 256: the upper threads repeat the lower threads' chunks - same value to the same LDS address)
        const int nr = idx / BCH, ch = idx - nr * BCH;
        bch8[j] = ch * 8;
        brow_ok[j] = n0 + nr < p.Npad;
        boff[j] = (long)(n0 + nr) * p.Cpad + ch * 8;
    }
    auto load = [&](int it, bf16x8 (&ra)[2 * AQ], bf16x8 (&rb)[BROWS]) {
        const bool live = it * KS + grp < total_all;   // past the end: every lane reads the zero page (cheap, and keeps the issue unconditional)
        const int tap = l_tap, kc = l_kc, ky = l_ky, kx = l_kx;
#pragma unroll
        for (int a = 0; a < KS; ++a) advance();
        bool ok = am_ok & live;                        // (& not &&: a short-circuit chain becomes branches)
        const __bf16* arow;
        if constexpr (GEN) {
            const int nh = base_h + ky * th, nw = base_w + kx * tw;
            const int ih = nh / p.sh, iw = nw / p.sw;
            ok = ok & (nh >= 0) & (nw >= 0) & (ih * p.sh == nh) & (iw * p.sw == nw) & ((unsigned)ih < (unsigned)p.Ha) & ((unsigned)iw < (unsigned)p.Wa);
            arow = p.A + (((long)ab * p.Ha + min(max(ih, 0), p.Ha - 1)) * p.Wa + min(max(iw, 0), p.Wa - 1)) * p.lda;       // always a real pixel; !ok lanes read the zero page
        } else {
            // the tap's displacement is the same for every pixel: a wave-uniform 64-bit offset (scalar unit) added to the thread's row of tap (0, 0) - two
            // vector instructions per step instead of the clamp / multiply chain of the first version (45 VALU per K step beside 8 MFMAs, ten of them
            // quarter-rate 32 x 32 multiplies: the large launches were bound by address arithmetic); a row outside the image is never dereferenced
            const int dhh = ky * th, dww = kx * tw;
            ok = ok & ((unsigned)(base_h + dhh) < (unsigned)p.Ha) & ((unsigned)(base_w + dww) < (unsigned)p.Wa);
            arow = arow0 + ((long)dhh * p.Wa + dww) * p.lda;
        }
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            bf16x8 two[2];
            gload16<AVEC>(arow, kc * KC + ahalf * (KC / 2) + q * 16, p.Ca, ok, two);
            ra[2 * q] = two[0];
            ra[2 * q + 1] = two[1];
        }
        const __bf16* wt = p.Wp + ((long)tap * p.Npad) * p.Cpad + kc * KC;
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const bool bok = live && brow_ok[j] && kc * KC + bch8[j] < p.Cpad;
            rb[j] = *reinterpret_cast<const bf16x8*>(bok ? wt + boff[j] : reinterpret_cast<const __bf16*>(g_gzero));
        }
    };
    auto stash = [&](int buf, const bf16x8 (&ra)[2 * AQ], const bf16x8 (&rb)[BROWS]) {
        __bf16* a = As + buf * GBM * GRS + arow * GRS + ahalf * (KC / 2);
#pragma unroll
        for (int q = 0; q < 2 * AQ; ++q) *reinterpret_cast<bf16x8*>(a + 8 * q) = ra[q];
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const int idx = (tid + j * 256) % BLOADS;
            const int nr = idx / BCH, ch = idx - nr * BCH;
            *reinterpret_cast<bf16x8*>(Bs + buf * BN * GRS + nr * GRS + ch * 8) = rb[j];
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = (lane >> 4) * 8;
    auto compute = [&](int buf) {
        const __bf16* a = As + buf * GBM * GRS + (wave * 32 + frow) * GRS + fk;
        const __bf16* b = Bs + buf * BN * GRS + frow * GRS + fk;
#pragma unroll
        for (int ks = 0; ks < AQ; ++ks) {
            bf16x8 fa[2], fb[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a + i * 16 * GRS + ks * 32);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(b + j * 16 * GRS + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);      // D rows = channels, D columns = pixels: a lane ends up with FOUR CONSECUTIVE CHANNELS of one pixel
        }
    };
    // every load and every LDS write below is issued unconditionally (past the last chunk the lanes read the zero page into a buffer nobody reads): with a
    // branch around an issue the compiler no longer knows how many loads are in flight and falls back to draining them all
    // (the step count is rounded up to the unroll factor: a step past the last chunk multiplies zero-page operands - cheaper than a second loop exit;
    //  sched_barrier: left alone, hipcc sinks a step's loads below its MFMAs, i.e. right in front of the wait for them)
    {
        load(0, ra0, rb0);
        load(1, ra1, rb1);
        stash(0, ra0, rb0);
        __syncthreads();
        for (int it = 0; it < total; it += 2) {
            load(it + 2, ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            stash(1, ra1, rb1);
            __syncthreads();
            load(it + 3, ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            stash(0, ra0, rb0);
            __syncthreads();
        }
    }

    if constexpr (KS > 1) {
        // the two wave groups' partial sums meet: group 1 writes its accumulators (element-major: conflict-free), leaves; group 0 adds them to its own.
        // The barriers after this point wait for the surviving waves only (s_barrier does not count terminated waves).
        float* rbuf = reinterpret_cast<float*>(smem);          // [2 * NT * 4][256] floats (<= 32 KB; the operand stages are dead after the loop's last barrier)
        if (grp) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rbuf[((i * NT + j) * 4 + r) * 256 + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (grp) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += rbuf[((i * NT + j) * 4 + r) * 256 + tid];
        __syncthreads();
    }

    gconv_epilogue<BN, OVEC, OUTF32>(p, smem, acc, tid, m0, n0, mt, nt, m_tiles);
}

// ---- kernel-row window variant (round 4): stride-1 convs with three kernel columns whose output has the input's size (pad = dilation: the 3x3 / 1x3 convs of
// HarDNet, Res2Net, RFB, the decoders) at large M.  Counters of gconv_kernel on such a launch (profiles/r04_gconv_144x72_counters.txt): 1.83 GB of L1 -> L2 reads
// for 150 MB of tensors, 54 % of them the A rows that each of the nine taps fetches again.  The sources of 128 consecutive output pixels for the three taps of a
// kernel row are ONE window of 128 + 2 d consecutive input pixels: a K step here is (kernel row, 32-channel chunk) - the window staged once, the three taps'
// weight tiles beside it, 3 x 2 x NT MFMAs per wave from it (tap kx reads the window kx d rows further down).  A pixel whose tap leaves the image row reads a zero
// row instead (two validity bits per pixel, as gwgrad3_kernel does on its dy side); a window row whose source lies outside the image / the tensor is staged as
// zeros.  A third of the A loads, a third of the barriers; same epilogue, same accumulation order over (ky, chunk, kx) as ... no: gconv_kernel runs (tap, chunk),
// this kernel (ky, chunk, kx) - results agree to fp32 rounding.
constexpr int G3_MAXD = 8;                                   // largest column dilation the window holds
constexpr int G3_ROWS = GBM + 2 * G3_MAXD;                   // 144 window rows (+ a zero row + a dummy row for the loader's out-of-window writes)
constexpr int G3_GRS = 32 + 8;                               // LDS row stride (elements): 32-channel chunks
template <int BN>
struct G3Smem {
    static constexpr int A = (G3_ROWS + 2) * G3_GRS * 2;
    static constexpr int B = 3 * BN * G3_GRS * 2;
    static constexpr int STAGE = A + B;
    static constexpr int EPI = GSmem<BN, 32, false>::CS + GSmem<BN, 32, false>::RED;
    static constexpr int BYTES = 2 * STAGE > EPI ? 2 * STAGE : EPI;
};

template <int BN, int AVEC, int OVEC>
__global__ __launch_bounds__(256) void gconv3_kernel(GConvP p) {
    constexpr int NT = BN / 16, GRS = G3_GRS, KC = 32;
    __shared__ __attribute__((aligned(16))) char smem[G3Smem<BN>::BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int mt = blockIdx.x, nt = blockIdx.y;
    const int m_tiles = gridDim.x;
    if (p.remap) {
        const int n_tiles = gridDim.y;
        const int logical = mi_xcd_remap(blockIdx.y * m_tiles + blockIdx.x, m_tiles * n_tiles);
        mt = logical / n_tiles;
        nt = logical - mt * n_tiles;
    }
    const int m0 = mt * GBM, n0 = nt * BN;
    const bool fwd = p.mode == MI_GATHER_FWD;
    const int sgn = fwd ? 1 : -1, dwc = p.dw, hw = p.Ha * p.Wa;
    const int wrows = GBM + 2 * dwc;                          // window rows in use

    // ---- A loader: thread -> (window row, 16-channel half); rows 0 .. 127 by all threads, rows 128 .. 127 + 2 d by the first 4 d thread slots
    const int ahalf = tid & 1;
    int wr[2];                                               // this thread's two window rows (the second: a real row or the dummy row)
    wr[0] = tid >> 1;
    wr[1] = GBM + (tid >> 1);
    const __bf16* arow0[2];
    bool aok0[2];
    int ah[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long pb = (long)m0 - dwc + wr[h];               // the window row's pixel for the centre kernel row (flat index)
        const bool in_win = wr[h] < wrows;
        aok0[h] = in_win && pb >= 0 && pb < (long)p.M;
        const int pc = aok0[h] ? (int)pb : 0;
        ah[h] = (pc % hw) / p.Wa;
        arow0[h] = p.A + (long)pc * p.lda;
        if (!in_win) wr[h] = G3_ROWS + 1;                    // dummy row: the write stays unconditional
    }
    // ---- B loader: three taps x BN rows x 4 chunks of 8 channels
    constexpr int BCH = KC / 8, BLOADS = BN * BCH, BROWS = (BLOADS + 255) / 256;
    long boff[BROWS];
    int bch8[BROWS], bnr[BROWS];
    bool brow_ok[BROWS];
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
        const int idx = (tid + j * 256) % BLOADS;
        const int nr = idx / BCH, ch = idx - nr * BCH;
        bnr[j] = nr;
        bch8[j] = ch * 8;
        brow_ok[j] = n0 + nr < p.Npad;
        boff[j] = (long)(n0 + nr) * p.Cpad + ch * 8;
    }
    const int kh = p.T / 3;
    const int total = MI_DBG_BIT(p, 1) ? 0 : kh * p.nchunks;      // K steps: (kernel row, 32-channel chunk), chunk fastest
    int l_ky = 0, l_kc = 0;
    bf16x8 ra0[4], rb0[3 * BROWS], ra1[4], rb1[3 * BROWS];
    auto load = [&](int it, bf16x8 (&ra)[4], bf16x8 (&rb)[3 * BROWS]) {
        const bool live = it < total;
        const int ky = l_ky, kc = l_kc;
        {
            const bool wrap = l_kc + 1 == p.nchunks;
            l_kc = wrap ? 0 : l_kc + 1;
            l_ky += wrap ? 1 : 0;
        }
        const int off = sgn * (ky * p.dh - p.ph);             // source row - output row (wave-uniform)
        const long aoff = (long)off * p.Wa * p.lda;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = aok0[h] & live & ((unsigned)(ah[h] + off) < (unsigned)p.Ha);
            bf16x8 two[2];
            gload16<AVEC>(arow0[h] + aoff, kc * KC + ahalf * 16, p.Ca, ok, two);
            ra[2 * h] = two[0];
            ra[2 * h + 1] = two[1];
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const __bf16* wt = p.Wp + ((long)(ky * 3 + kx) * p.Npad) * p.Cpad + kc * KC;
#pragma unroll
            for (int j = 0; j < BROWS; ++j) {
                const bool bok = live & brow_ok[j];
                rb[kx * BROWS + j] = *reinterpret_cast<const bf16x8*>(bok ? wt + boff[j] : reinterpret_cast<const __bf16*>(g_gzero));
            }
        }
    };
    auto stash = [&](int buf, const bf16x8 (&ra)[4], const bf16x8 (&rb)[3 * BROWS]) {
        __bf16* aw = reinterpret_cast<__bf16*>(smem + buf * G3Smem<BN>::STAGE);
        __bf16* bw = reinterpret_cast<__bf16*>(smem + buf * G3Smem<BN>::STAGE + G3Smem<BN>::A);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<bf16x8*>(aw + wr[h] * GRS + ahalf * 16) = ra[2 * h];
            *reinterpret_cast<bf16x8*>(aw + wr[h] * GRS + ahalf * 16 + 8) = ra[2 * h + 1];
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < BROWS; ++j) *reinterpret_cast<bf16x8*>(bw + (kx * BN + bnr[j]) * GRS + bch8[j]) = rb[kx * BROWS + j];
    };
    // the zero rows of both stages (row G3_ROWS), once
    if (tid < 2 * (GRS / 2)) {
        const int b = tid / (GRS / 2), e = tid % (GRS / 2);
        reinterpret_cast<uint32_t*>(smem + b * G3Smem<BN>::STAGE + G3_ROWS * GRS * 2)[e] = 0u;
    }

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = (lane >> 4) * 8;
    // this lane's two pixels: window rows of the three taps (element offsets) and whether the outer taps stay inside the image row
    int arow_k[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int local = wave * 32 + i * 16 + frow;
        const int m = m0 + local;
        const int ow = (m < p.M ? m : 0) % p.Wa;
        const bool left = ow - dwc >= 0, right = ow + dwc < p.Wa;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sft = sgn * (kx - 1) * dwc;             // source column - output column
            const bool valid = kx == 1 || (sft < 0 ? left : right);
            arow_k[i][kx] = (valid ? local + dwc + sft : G3_ROWS) * GRS + fk;
        }
    }
    auto compute = [&](int buf) {
        const __bf16* aw = reinterpret_cast<const __bf16*>(smem + buf * G3Smem<BN>::STAGE);
        const __bf16* bw = reinterpret_cast<const __bf16*>(smem + buf * G3Smem<BN>::STAGE + G3Smem<BN>::A) + frow * GRS + fk;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bf16x8 fa[2], fb[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(aw + arow_k[i][kx]);
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bw + (kx * BN + j * 16) * GRS);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };
    {
        load(0, ra0, rb0);
        load(1, ra1, rb1);
        stash(0, ra0, rb0);
        __syncthreads();
        for (int it = 0; it < total; it += 2) {
            load(it + 2, ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            stash(1, ra1, rb1);
            __syncthreads();
            load(it + 3, ra1, rb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            stash(0, ra0, rb0);
            __syncthreads();
        }
    }
    gconv_epilogue<BN, OVEC, false>(p, smem, acc, tid, m0, n0, mt, nt, m_tiles);
}

template <int BN, int KC, int AVEC, int OVEC, bool OUTF32>
void glaunch(const GConvP& p, hipStream_t s) {
    dim3 grid((p.M + GBM - 1) / GBM, (p.N + BN - 1) / BN);
    if (p.mode != MI_GATHER_FWD && (p.sh != 1 || p.sw != 1)) {                     // data gradient of a strided conv: the general source map
        hipLaunchKernelGGL((gconv_kernel<BN, KC, AVEC, OVEC, OUTF32, true>), grid, dim3(256), 0, s, p);
        return;
    }
    if constexpr (!OUTF32 && (BN == 32 || BN == 64) && KC == 64 && AVEC >= 4 && OVEC >= 4) {       // two wave groups over the K chunks (see KS above)
        const int ks2_wgs = mi_sw().gconv_ks2_wgs;
        if ((int)(grid.x * grid.y) <= ks2_wgs && p.T * p.nchunks >= 8) {
            constexpr int bytes = 2 * GSmem<BN, KC, false>::AB;
            static_assert(bytes >= GSmem<BN, KC, false>::CS + GSmem<BN, KC, false>::RED && bytes >= 2 * (BN / 16) * 4 * 256 * 4, "LDS of the two-group launch");
            static std::atomic<uint64_t> attr;
            mi_allow_dynamic_lds((const void*)gconv_kernel<BN, KC, AVEC, OVEC, false, false, 2>, bytes, attr);
            hipLaunchKernelGGL((gconv_kernel<BN, KC, AVEC, OVEC, false, false, 2>), grid, dim3(512), bytes, s, p);
            return;
        }
    }
    hipLaunchKernelGGL((gconv_kernel<BN, KC, AVEC, OVEC, OUTF32>), grid, dim3(256), 0, s, p);
}

template <int BN, int KC, int AVEC>
void glaunch_o(const GConvP& p, int ovec, bool f32, hipStream_t s) {
    if (f32) {
        if constexpr (BN == 32) glaunch<32, KC, AVEC, 1, true>(p, s);
        return;
    }
    if (ovec == 8) glaunch<BN, KC, AVEC, 8, false>(p, s);
    else if (ovec == 4) glaunch<BN, KC, AVEC, 4, false>(p, s);
    else glaunch<BN, KC, AVEC, 1, false>(p, s);
}

template <int BN, int KC>
void glaunch_a(const GConvP& p, int avec, int ovec, bool f32, hipStream_t s) {
    if (avec == 8) glaunch_o<BN, KC, 8>(p, ovec, f32, s);
    else if (avec == 4) glaunch_o<BN, KC, 4>(p, ovec, f32, s);
    else glaunch_o<BN, KC, 1>(p, ovec, f32, s);
}

template <int BN>
void glaunch3(const GConvP& p, int avec, int ovec, hipStream_t s) {
    dim3 grid((p.M + GBM - 1) / GBM, (p.N + BN - 1) / BN);
#define G3L(AV, OV) hipLaunchKernelGGL((gconv3_kernel<BN, AV, OV>), grid, dim3(256), 0, s, p)
    if (avec == 8 && ovec == 8) G3L(8, 8);
    else if (avec == 8 && ovec == 4) G3L(8, 4);
    else if (avec == 8) G3L(8, 1);
    else if (avec == 4 && ovec == 8) G3L(4, 8);
    else if (avec == 4 && ovec == 4) G3L(4, 4);
    else if (avec == 4) G3L(4, 1);
    else if (ovec == 8) G3L(1, 8);
    else if (ovec == 4) G3L(1, 4);
    else G3L(1, 1);
#undef G3L
}

// BN = 128 keeps 32-channel chunks (its LDS image with 64 would pass the 64 KiB of static LDS)
template <int BN>
void glaunch_k(GConvP& p, int avec, int ovec, bool f32, hipStream_t s) {
    if constexpr (2 * (GBM + BN) * (64 + 8) * 2 <= 64 * 1024) {          // (BN <= 80: the 64-channel image fits the 64 KiB of static LDS)
        // 64-channel chunks halve the K steps; a conv on a large map (thousands of workgroups) gains more from the occupancy of
        // the 32-channel tile (30 KB of LDS instead of 55: 104 -> 256 at 16 x 88 x 88 54.7 vs 63.7 us, 208 -> 512 at 44 x 44 33.5 vs 41.0)
        const int kc32_wgs = mi_sw().gconv_kc32_wgs;
        const long wgs = (long)((p.M + GBM - 1) / GBM) * ((p.N + BN - 1) / BN);
        const bool small_k_big_m = wgs >= kc32_wgs && mi_sw().gconv_kc != 64;          // (3x3 convs on large maps too: GALD 38.4 -> 37.6 ms with 32-channel chunks)
        if (p.Cpad >= 64 && p.T * ((p.Cpad + 63) / 64) >= 2 && mi_sw().gconv_kc != 32 && !small_k_big_m) {
            p.nchunks = (p.Cpad + 63) / 64;
            glaunch_a<BN, 64>(p, avec, ovec, f32, s);
            return;
        }
    }
    p.nchunks = p.Cpad / 32;
    glaunch_a<BN, 32>(p, avec, ovec, f32, s);
}

int view_vec(const void* ptr, long ld, int C, bool load = true) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    if ((a & 15) == 0 && ld % 8 == 0 && C % 8 == 0) return 8;
    if ((a & 3) == 0 && ld % 2 == 0 && C % 2 == 0 && (C >= 8 || !load)) return 4;        // 16-byte accesses at 4-byte alignment (loads need a full window)
    return 1;
}

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int WTO = 64, WTI = 64, WKP = 64;     // output tile 64 (o) x 64 (i), 64 pixels (two MFMA k) per K step
constexpr int WRS = 144;                        // LDS bytes per pixel row of a tile: 64 channels bf16 + 16 B pad (8-B aligned for the transposed reads)

struct GWgP {
    const __bf16* dY;
    const __bf16* X;
    float* slab;
    long ldy, ldx;
    int M, O, I, T;
    int Ho, Wo, Ha, Wa;
    int kw, sh, sw, ph, pw, dh, dw;
    int S, rows_per_split, o_tiles, i_tiles;
    unsigned* ticket;      // in-launch reduction (S <= GW_INLAUNCH_S): one zeroed word per (tap, output tile); else null
    float* dwout;
    int accumulate, remap, dbg;
};
constexpr int GW_INLAUNCH_S = 16;      // K splits the last workgroup of a tile adds itself (64 KB of slabs at most); more: the reducer launch

template <int VEC>
__device__ __forceinline__ bf16x8 gload8(const __bf16* src, int c0, int C, bool ok) {
    union { bf16x8 v; uint32_t u[4]; uint16_t h[8]; } x;
    const __bf16* zero = reinterpret_cast<const __bf16*>(g_gzero);
    if constexpr (VEC == 8) {
        x.v = *reinterpret_cast<const bf16x8*>((ok && c0 < C) ? src + c0 : zero);
    } else if constexpr (VEC == 4) {
        x.v = gload8_a4(src, c0, C, ok);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x.h[j] = *reinterpret_cast<const uint16_t*>((ok && c0 + j < C) ? src + c0 + j : zero);
    }
    return x.v;
}

__device__ __forceinline__ s16x4 tr_read(const char* lds_generic) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_generic));
}

// One workgroup = one (K split, tap, o tile, i tile).  Both operands have the contraction index (pixel) as their memory row, so the
// tiles are staged pixel-major and the MFMA fragments come from ds_read_b64_tr_b16 (hardware transpose), as in igemm_tn.hip.
// (the body is shared by the one-conv launch and by the table-driven launch that runs the weight gradients of many convs at once: `bid` of `nblk`
//  workgroups of THIS conv)
constexpr int GW_SMEM = 2 * 2 * WKP * WRS;                                 // [buf][dy | x][64 pixels][144 B]
template <int YVEC, int XVEC>
__device__ __forceinline__ void gwgrad_body(const GWgP& p, int bid, int nblk, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = p.o_tiles * p.i_tiles;
    int id = p.remap ? mi_xcd_remap(bid, nblk) : bid;      // the (tile, tap) workgroups of a pixel split share its dy / x rows: keep them on one XCD's L2
    const int tile = id % tiles;
    id /= tiles;
    const int t = id % p.T, split = id / p.T;
    const int ot = tile / p.i_tiles, itile = tile - ot * p.i_tiles;
    const int o0 = ot * WTO, i0 = itile * WTI;
    const int ky = t / p.kw, kx = t - ky * p.kw;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nk = MI_DBG_BIT(p, 1) ? 0 : (m_end > m_begin ? (m_end - m_begin + WKP - 1) / WKP : 0);
    const int lpx = tid >> 3, lch = tid & 7;          // loader: pixel rows lpx and lpx + 32 of the step, 8-channel chunk
    const int hw = p.Ho * p.Wo;

    bf16x8 ry0[2], rx0[2], ry1[2], rx1[2];
    // load() is called for kt = 0, 1, 2, ... in order: the (image, row, column) of this thread's two pixel rows advance by 64 pixels per call instead of
    // being divided out of the pixel index every time (four integer divisions per step and thread in the first version)
    int cb[2], coh[2], cow[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = m_begin + lpx + 32 * h;
        cb[h] = m / hw;
        const int rem = m - cb[h] * hw;
        coh[h] = rem / p.Wo;
        cow[h] = rem - coh[h] * p.Wo;
    }
    // (selects only - the main loop stays one basic block, see gconv_kernel: a 64-pixel step is step_imgs images + step_rows rows + step_cols columns
    //  with step_rows < Ho, so every coordinate wraps at most once per step)
    const int step_imgs = WKP / hw, step_rem = WKP - step_imgs * hw;
    const int step_rows = step_rem / p.Wo, step_cols = step_rem - step_rows * p.Wo;
    auto load = [&](int kt, bf16x8 (&ry)[2], bf16x8 (&rx)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m_begin + kt * WKP + lpx + 32 * h;
            const bool ok = m < m_end;                 // (a step past the last one has every row >= m_end: zero-page reads)
            ry[h] = gload8<YVEC>(p.dY + (long)(ok ? m : 0) * p.ldy, o0 + lch * 8, p.O, ok);
            const int b = cb[h], oh = coh[h], ow = cow[h];
            {
                const int w1 = cow[h] + step_cols;
                const bool cw = w1 >= p.Wo;
                cow[h] = cw ? w1 - p.Wo : w1;
                const int h1 = coh[h] + step_rows + (cw ? 1 : 0);
                const bool ch = h1 >= p.Ho;
                coh[h] = ch ? h1 - p.Ho : h1;
                cb[h] += step_imgs + (ch ? 1 : 0);
            }
            const int ih = oh * p.sh + ky * p.dh - p.ph, iw = ow * p.sw + kx * p.dw - p.pw;
            const bool xok = ok & ((unsigned)ih < (unsigned)p.Ha) & ((unsigned)iw < (unsigned)p.Wa);
            const long pix = ((long)b * p.Ha + min(max(ih, 0), p.Ha - 1)) * p.Wa + min(max(iw, 0), p.Wa - 1);        // always a real pixel; !xok lanes read the zero page
            rx[h] = gload8<XVEC>(p.X + (ok ? pix : 0) * p.ldx, i0 + lch * 8, p.I, xok);
        }
    };
    auto stash = [&](int buf, const bf16x8 (&ry)[2], const bf16x8 (&rx)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            char* sy = smem + buf * (2 * WKP * WRS) + (lpx + 32 * h) * WRS + lch * 16;
            *reinterpret_cast<bf16x8*>(sy) = ry[h];
            *reinterpret_cast<bf16x8*>(sy + WKP * WRS) = rx[h];
        }
    };

    // D rows = i (A operand = X^T), D cols = o (B operand = dY^T); wave owns 32 (i) x 32 (o)
    const int wi = wave & 1, wo = wave >> 1;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposed read: lanes 16g .. 16g+15 fetch pixel rows 4g + q (q = (lane & 15) >> 2; +16 for the second half of k), the lane's
    // 8 bytes at channel 4 * (lane & 3) of the 16-channel block; lane i of the group receives channel i of the four rows
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int row_off = (g * 4 + q) * WRS + pc * 8;

    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < WKP / 32; ++ks) {
            const char* sy = smem + buf * (2 * WKP * WRS) + row_off + ks * 32 * WRS;
            const char* sx = sy + WKP * WRS;
            union { bf16x8 v; s16x4 h[2]; } xf[2], yf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const char* base = sx + (wi * 32 + a * 16) * 2;
                xf[a].h[0] = tr_read(base);
                xf[a].h[1] = tr_read(base + 16 * WRS);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const char* base = sy + (wo * 32 + b * 16) * 2;
                yf[b].h[0] = tr_read(base);
                yf[b].h[1] = tr_read(base + 16 * WRS);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a].v, yf[b].v, acc[a][b], 0, 0, 0);
        }
    };
    if (nk > 0) {
        // (steps rounded up to two: a step past the split's last one multiplies zero-page operands - cheaper than a second loop exit; sched_barrier:
        //  left alone hipcc sinks a step's loads below its MFMAs, right in front of the wait for them)
        load(0, ry0, rx0);
        load(1, ry1, rx1);
        stash(0, ry0, rx0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            load(kt + 2, ry0, rx0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            stash(1, ry1, rx1);
            __syncthreads();
            load(kt + 3, ry1, rx1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            stash(0, ry0, rx0);
            __syncthreads();
        }
    }
    // slab[split][t][o][i], o < O, i < roundup(I, 4) (i fastest): the lane owns o = column, four consecutive i = rows
    const int Ip = (p.I + 3) & ~3;
    float* slab = p.slab + ((long)(split * p.T + t) * p.O) * Ip;
    const int fcol = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int o = o0 + wo * 32 + b * 16 + fcol;
        if (o >= p.O) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int i = i0 + wi * 32 + a * 16 + fq * 4;
            if (i < Ip) {
                if (p.ticket) {                    // handed to the tile's last workgroup: write-through
                    mi_st2_sc1(slab + (long)o * Ip + i, acc[a][b][0], acc[a][b][1]);
                    mi_st2_sc1(slab + (long)o * Ip + i + 2, acc[a][b][2], acc[a][b][3]);
                } else {
                    *reinterpret_cast<f32x4*>(slab + (long)o * Ip + i) = acc[a][b];
                }
            }
        }
    }
    if (p.ticket) {
        // few K splits: the (tap, tile)'s last workgroup adds the slabs itself - gwgrad_reduce_kernel's arithmetic in its order (lane l of 8 adds the
        // splits l, l + 8, ... ascending, the eight sums are combined as the butterfly does: ((0+1)+(2+3))+((4+5)+(6+7))) - the same bits, one launch
        if (mi_last_arriver(p.ticket + (long)t * tiles + tile, p.S, reinterpret_cast<int*>(smem))) {
            mi_acquire_partials();
            // A thread owns eight float2 elements of the tile (rows tid / 32 + 8 k, columns 2 (tid % 32)), S <= 16 slabs each: 128 loads that all miss the L2
            // (the slabs were written by other XCDs).  They are issued in groups that fill ~64 registers - two elements at S > 8, four at S > 4, all eight
            // below - and only then added and stored: the first version went element by element, and since a store to dw may alias a slab as far as the
            // compiler knows, each element's two load batches waited for the previous element's store - sixteen dependent round trips, 24 of the 33 us of a
            // 3x3 104 -> 104 gradient at 16 x 22 x 22 (tools/gkshape.py, MI_GW_DBG=1).
            const long sstride = (long)p.T * p.O * Ip;
            const float* __restrict__ base = p.slab + (long)t * p.O * Ip;
            float* __restrict__ dwo = p.dwout;
            const int S = p.S, accumulate = p.accumulate;
            const int ip = (tid & 31) * 2, i = i0 + ip;
            auto group = [&](auto EGc, int k0) {
                constexpr int EG = decltype(EGc)::value, SL = 32 / EG;        // elements per group, slab loads per element (>= S)
                float2 v[EG][SL];
                float old0[EG], old1[EG];
#pragma unroll
                for (int k = 0; k < EG; ++k) {
                    const int o = o0 + (tid >> 5) + 8 * (k0 + k);
                    const bool live = o < p.O && i < Ip;
                    const long off = live ? (long)o * Ip + i : 0;
#pragma unroll
                    for (int l = 0; l < SL; ++l) v[k][l] = *reinterpret_cast<const float2*>(base + (l < S ? l : S - 1) * sstride + off);
                    old0[k] = (accumulate && live && i < p.I) ? dwo[((long)o * p.I + i) * p.T + t] : 0.f;
                    old1[k] = (accumulate && live && i + 1 < p.I) ? dwo[((long)o * p.I + i + 1) * p.T + t] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < EG; ++k) {
                    const int o = o0 + (tid >> 5) + 8 * (k0 + k);
                    // gwgrad_reduce_kernel's arithmetic in its order: lane l of 8 adds the splits l, l + 8 ascending, the eight sums meet as the butterfly does
                    float a0[8], a1[8];
#pragma unroll
                    for (int l = 0; l < 8; ++l) {
                        a0[l] = (l < SL && l < S) ? v[k][l < SL ? l : 0].x : 0.f;
                        a1[l] = (l < SL && l < S) ? v[k][l < SL ? l : 0].y : 0.f;
                        if constexpr (SL > 8) {
                            if (l + 8 < S) a0[l] += v[k][l + 8].x, a1[l] += v[k][l + 8].y;
                        }
                    }
                    const float r0 = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7]));
                    const float r1 = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
                    if (o < p.O && i < p.I) dwo[((long)o * p.I + i) * p.T + t] = accumulate ? old0[k] + r0 : r0;
                    if (o < p.O && i + 1 < p.I) dwo[((long)o * p.I + i + 1) * p.T + t] = accumulate ? old1[k] + r1 : r1;
                }
            };
            if (S > 8) {
#pragma unroll 1
                for (int k0 = 0; k0 < 8; k0 += 2) group(std::integral_constant<int, 2>{}, k0);
            } else if (S > 4) {
#pragma unroll 1
                for (int k0 = 0; k0 < 8; k0 += 4) group(std::integral_constant<int, 4>{}, k0);
            } else {
                group(std::integral_constant<int, 8>{}, 0);
            }
        }
    }
}

// Fused kernel row (round 4): one workgroup = (K split, kernel row ky, o tile, i tile) computes the THREE taps kx = 0, 1, 2 of that row from ONE staged
// window of x.  For a stride-1 conv whose output has the input's height and width (pad = dilation: every 3x3 / 1x3 conv of Res2Net, RFB, the partial
// decoder, HarDNet), the sources of 64 consecutive output pixels for tap (ky, kx) are 64 consecutive input pixels, shifted by kx * dw between the taps: the
// window of 64 + 2 dw pixel rows is loaded once and the transposed MFMA reads of tap kx start kx * dw rows further down.  What the per-tap kernel got from
// the zero padding for free - a tap that leaves the image row contributes nothing - is done on the dy side: the loader writes two validity bits per output
// pixel (left / right tap inside the row), and a lane whose pixel is invalid for the tap reads its dy fragment from a zero row instead.  x rows are valid
// when they lie in the tensor and the output row they serve (h_src - (ky dh - ph)) lies in the image.  Counters before (profiles/pmc_gald.json,
// pmc_pranet.json): the per-tap kernel moved 32.6 / 13.3 GB of HBM per step (GALD / PraNet), its L2 hit rate 0.57 / 0.42 - nine re-reads of dy and x.
constexpr int W3_MAXD = 8;                                  // largest column dilation the window holds
constexpr int W3_XROWS = WKP + 2 * W3_MAXD;                 // 80 pixel rows of x per stage
constexpr int W3_STAGE = (W3_XROWS + WKP) * WRS + 64 + 2 * WRS;      // x window | dy tile | 64 flag bytes | dummy area (target of the loader's out-of-window writes)
constexpr int W3_DUMMY_ROW = ((W3_XROWS + WKP) * WRS + 64 + WRS - 1) / WRS;      // the whole row that lies inside the dummy area, counted from the stage start
static_assert((W3_DUMMY_ROW + 1) * WRS <= W3_STAGE && W3_DUMMY_ROW * WRS >= (W3_XROWS + WKP) * WRS + 64 && W3_STAGE % 16 == 0, "dummy row inside the stage");
constexpr int GW3_SMEM = 2 * W3_STAGE + WRS;                               // two stages + one zero row
template <int YVEC, int XVEC>
__device__ __forceinline__ void gwgrad3_body(const GWgP& p, int bid, int nblk, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = p.o_tiles * p.i_tiles;
    const int kh = p.T / 3;
    int id = p.remap ? mi_xcd_remap(bid, nblk) : bid;      // the (tile, tap) workgroups of a pixel split share its dy / x rows: keep them on one XCD's L2
    const int tile = id % tiles;
    id /= tiles;
    const int ky = id % kh, split = id / kh;
    const int ot = tile / p.i_tiles, itile = tile - ot * p.i_tiles;
    const int o0 = ot * WTO, i0 = itile * WTI;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nk = MI_DBG_BIT(p, 1) ? 0 : (m_end > m_begin ? (m_end - m_begin + WKP - 1) / WKP : 0);
    const int lpx = tid >> 3, lch = tid & 7;
    const int hw = p.Ha * p.Wa;
    const int dyoff = ky * p.dh - p.ph;                      // source row - output row
    const int xrows = WKP + 2 * p.dw;
    char* zero_row = smem + 2 * W3_STAGE;
    if (tid < WRS / 4) reinterpret_cast<uint32_t*>(zero_row)[tid] = 0u;

    // coordinates of this thread's rows, advanced by 64 pixels per call of load(): dy rows lpx, lpx + 32; x window rows lpx, lpx + 32, lpx + 64
    int yh[2], yw[2], xh[3], xw[3];
    long xq[3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = m_begin + lpx + 32 * h;
        const int rem = m % hw;
        yh[h] = rem / p.Wa;
        yw[h] = rem - yh[h] * p.Wa;
    }
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        xq[h] = (long)m_begin + (long)dyoff * p.Wa - p.dw + lpx + 32 * h;
        const int rem = (int)(((xq[h] % hw) + hw) % hw);
        xh[h] = rem / p.Wa;
        xw[h] = rem - xh[h] * p.Wa;
    }
    // (selects only - the main loop stays one basic block, see gconv_kernel; only the row within the image matters here, so a step is
    //  (64 mod Ha*Wa) pixels = step_rows < Ha rows + step_cols columns and every coordinate wraps at most once)
    const int step_rem = WKP % hw;
    const int step_rows = step_rem / p.Wa, step_cols = step_rem - step_rows * p.Wa;
    auto advance = [&](int& hh, int& ww) {
        const int w1 = ww + step_cols;
        const bool cw = w1 >= p.Wa;
        ww = cw ? w1 - p.Wa : w1;
        const int h1 = hh + step_rows + (cw ? 1 : 0);
        hh = h1 >= p.Ha ? h1 - p.Ha : h1;
    };
    bf16x8 ry0[2], rx0[3], ry1[2], rx1[3];
    uint32_t rf0 = 0, rf1 = 0;                               // validity bits of the two dy rows (byte h)
    auto load = [&](int kt, bf16x8 (&ry)[2], bf16x8 (&rx)[3], uint32_t& rf) {
        rf = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m_begin + kt * WKP + lpx + 32 * h;
            const bool ok = m < m_end;
            ry[h] = gload8<YVEC>(p.dY + (long)(ok ? m : 0) * p.ldy, o0 + lch * 8, p.O, ok);
            const uint32_t f = (yw[h] >= p.dw ? 1u : 0u) | (yw[h] + p.dw < p.Wa ? 2u : 0u);
            rf |= f << (8 * h);
            advance(yh[h], yw[h]);
        }
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const bool in_window = lpx + 32 * h < xrows;
            const bool ok = in_window & (xq[h] >= 0) & (xq[h] < (long)p.M) & ((unsigned)(xh[h] - dyoff) < (unsigned)p.Ha) & (kt < nk);
            rx[h] = gload8<XVEC>(p.X + (ok ? xq[h] : 0L) * p.ldx, i0 + lch * 8, p.I, ok);
            xq[h] += WKP;
            advance(xh[h], xw[h]);
        }
    };
    auto stash = [&](int buf, const bf16x8 (&ry)[2], const bf16x8 (&rx)[3], uint32_t rf) {
        char* sx = smem + buf * W3_STAGE;
        char* sy = sx + W3_XROWS * WRS;
#pragma unroll
        for (int h = 0; h < 3; ++h) {                        // (rows past the window go to the stage's dummy row: an unconditional write, no branch)
            const int row = lpx + 32 * h < W3_XROWS ? lpx + 32 * h : W3_DUMMY_ROW;
            *reinterpret_cast<bf16x8*>(sx + row * WRS + lch * 16) = rx[h];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<bf16x8*>(sy + (lpx + 32 * h) * WRS + lch * 16) = ry[h];
            sy[WKP * WRS + lpx + 32 * h] = (char)((rf >> (8 * h)) & 0xff);            // (the eight threads of a pixel row write the same byte)
        }
    };

    const int wi = wave & 1, wo = wave >> 1;
    f32x4 acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int prow = g * 4 + q;                              // this lane's pixel row within a 16-row half
    const int row_off = prow * WRS + pc * 8;
    const int dwrs = p.dw * WRS;
    auto compute = [&](int buf) {
        const char* sx0 = smem + buf * W3_STAGE;
        const char* sy0 = sx0 + W3_XROWS * WRS;
        const unsigned char* flags = reinterpret_cast<const unsigned char*>(sy0 + WKP * WRS);
#pragma unroll
        for (int ks = 0; ks < WKP / 32; ++ks) {
            const char* sy = sy0 + row_off + ks * 32 * WRS;
            const unsigned f0 = flags[ks * 32 + prow], f1 = flags[ks * 32 + prow + 16];
#pragma unroll
            for (int t = 0; t < 3; ++t) {                    // one tap at a time: its dy fragments (masked for the outer taps), its x fragments, four MFMAs
                union { bf16x8 v; s16x4 h[2]; } yf[2], xf[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int col = (wo * 32 + b * 16) * 2;
                    const char* z = zero_row + pc * 8 + col;
                    const bool v0 = t == 1 || (f0 & (t == 0 ? 1u : 2u)), v1 = t == 1 || (f1 & (t == 0 ? 1u : 2u));
                    yf[b].h[0] = tr_read(v0 ? sy + col : z);
                    yf[b].h[1] = tr_read(v1 ? sy + col + 16 * WRS : z);
                }
                const char* sx = sx0 + row_off + ks * 32 * WRS + t * dwrs;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const char* base = sx + (wi * 32 + a * 16) * 2;
                    xf[a].h[0] = tr_read(base);
                    xf[a].h[1] = tr_read(base + 16 * WRS);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a].v, yf[b].v, acc[t][a][b], 0, 0, 0);
            }
        }
    };
    if (nk > 0) {
        load(0, ry0, rx0, rf0);
        load(1, ry1, rx1, rf1);
        stash(0, ry0, rx0, rf0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {                 // (steps rounded up to two, sched_barrier: see gwgrad_kernel)
            load(kt + 2, ry0, rx0, rf0);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            stash(1, ry1, rx1, rf1);
            __syncthreads();
            load(kt + 3, ry1, rx1, rf1);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            stash(0, ry0, rx0, rf0);
            __syncthreads();
        }
    }
    const int Ip = (p.I + 3) & ~3;
    const int fcol = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float* slab = p.slab + ((long)(split * p.T + ky * 3 + t) * p.O) * Ip;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int o = o0 + wo * 32 + b * 16 + fcol;
            if (o >= p.O) continue;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int i = i0 + wi * 32 + a * 16 + fq * 4;
                if (i < Ip) *reinterpret_cast<f32x4*>(slab + (long)o * Ip + i) = acc[t][a][b];
            }
        }
    }
}

// dw[o][i][t] (+)= sum over the K splits (bitwise reproducible); 8 lanes per output element (t, o, i: i fastest): lane l adds the splits
// l, l + 8, ... in ascending order, the eight partial sums are combined by a fixed butterfly
__device__ __forceinline__ void gwgrad_reduce_body(const float* __restrict__ slab, float* __restrict__ dw, int O, int I, int T, int S, int accumulate, long bid, long nblk) {
    const int Ip = (I + 3) & ~3;
    const long n = (long)T * O * I;
    const long sstride = (long)T * O * Ip;
    const int l = threadIdx.x & 7;
    for (long e = (bid * blockDim.x + threadIdx.x) >> 3; e < n; e += (nblk * blockDim.x) >> 3) {      // the 8 lanes of a group share e
        const int i = (int)(e % I);
        const long r = e / I;
        const int o = (int)(r % O), t = (int)(r / O);
        const float* src = slab + ((long)t * O + o) * Ip + i;
        float v = 0.f;
        for (int s = l; s < S; s += 8) v += src[s * sstride];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if (l == 0) {
            float* d = dw + ((long)o * I + i) * T + t;
            *d = accumulate ? *d + v : v;
        }
    }
}
__global__ __launch_bounds__(256) void gwgrad_reduce_kernel(const float* slab, float* dw, int O, int I, int T, int S, int accumulate) {
    gwgrad_reduce_body(slab, dw, O, I, T, S, accumulate, blockIdx.x, gridDim.x);
}

// ---- launches: one conv per launch (parameters as kernel arguments) ...
template <int YVEC, int XVEC>
__global__ __launch_bounds__(256, 4) void gwgrad_kernel(GWgP p) {       // four workgroups per CU (<= 128 VGPRs): the large launches are bandwidth-bound, 136 VGPRs cost them 25 %
    __shared__ __attribute__((aligned(16))) char smem[GW_SMEM];
    gwgrad_body<YVEC, XVEC>(p, blockIdx.x, gridDim.x, smem);
}
template <int YVEC, int XVEC>
__global__ __launch_bounds__(256, 2) void gwgrad3_kernel(GWgP p) {
    __shared__ __attribute__((aligned(16))) char smem[GW3_SMEM];
    gwgrad3_body<YVEC, XVEC>(p, blockIdx.x, gridDim.x, smem);
}

// ---- ... or the weight gradients of MANY convs in one launch (mi_gconv_wgrad_multi).  A training step of PraNet / GALD issues 140 - 200 weight
// gradients of 1 - 10 GFLOP; alone, each is a latency chain (launch, a handful of K steps of ~0.9 us, slab round trip, second-level sum: 15 - 25 us
// of its 25 - 60 us are fixed, tools/gkshape.py with MI_GW_DBG=1), and nothing reads a weight gradient before the optimizer.  The tape therefore
// queues them and flushes the queue once per backward: a descriptor table in device memory (written by gw_table_kernel from kernel arguments - no host
// buffer that a HIP graph would have to keep alive), one main launch per operand-alignment class in which every workgroup finds its conv by binary
// search over the table's first-block column, and ONE reducer launch for all slabs.  With the whole backward's gradients in flight together a conv needs
// no split count that fills the chip by itself: the K range per workgroup is ~48 steps (MI_GWM_STEPS; 8: 894, 16: 915, 24: 911, 48: 917 images/s on PraNet, GALD 44.0 / 42.7 / 42.5 / 42.2 ms) and the slabs shrink accordingly.
struct alignas(16) GWgD {
    GWgP p;
    int first_block, nblk;        // main launch of the conv's class: its workgroups are [first_block, first_block + nblk), first_block a multiple of 8 (XCD phase)
    int first_rblock, n_rblk;     // reducer launch
    int pad_[4];
};
static_assert(sizeof(GWgD) % 16 == 0, "descriptor rows are copied as 16-byte words");
constexpr int GW_CHUNK = 8;
struct GWgChunk { GWgD d[GW_CHUNK]; };

__global__ __launch_bounds__(256) void gw_table_kernel(GWgD* table, int first, int count, GWgChunk c) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&c);
    uint32_t* dst = reinterpret_cast<uint32_t*>(table + first);
    const int words = count * (int)(sizeof(GWgD) / 4);
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}

__device__ __forceinline__ int gw_find(const GWgD* __restrict__ table, int first, int n, int bid, bool reducer) {
    int lo = first, hi = first + n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((reducer ? table[mid].first_rblock : table[mid].first_block) <= bid) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

template <int YVEC, int XVEC>
__global__ __launch_bounds__(256, 4) void gwgrad_multi_kernel(const GWgD* __restrict__ table, int first, int n) {
    __shared__ __attribute__((aligned(16))) char smem[GW_SMEM];
    const int at = gw_find(table, first, n, blockIdx.x, false);
    const GWgP p = table[at].p;                          // (by value: scalar loads once, not re-read after the kernel's own stores)
    const int b = blockIdx.x - table[at].first_block;
    if (b >= table[at].nblk) return;                     // (padding up to the next multiple of 8)
    gwgrad_body<YVEC, XVEC>(p, b, table[at].nblk, smem);
}
template <int YVEC, int XVEC>
__global__ __launch_bounds__(256, 2) void gwgrad3_multi_kernel(const GWgD* __restrict__ table, int first, int n) {
    __shared__ __attribute__((aligned(16))) char smem[GW3_SMEM];
    const int at = gw_find(table, first, n, blockIdx.x, false);
    const GWgP p = table[at].p;
    const int b = blockIdx.x - table[at].first_block;
    if (b >= table[at].nblk) return;
    gwgrad3_body<YVEC, XVEC>(p, b, table[at].nblk, smem);
}

// The reducer of the table-driven launch: a thread owns four consecutive i of one (t, o) - the lanes of a wave read 1 KiB runs of a slab - and adds the
// splits in ascending order, eight loads in flight (the one-conv reducer spreads an element's splits over 8 lanes: with the 2 - 80 splits of this path most
// of those lanes idle, and the wave's reads are 32-byte pieces: 0.66 ms per PraNet step for ~300 MB).
__global__ __launch_bounds__(256) void gwgrad_reduce_multi_kernel(const GWgD* __restrict__ table, int n) {
    const int at = gw_find(table, 0, n, blockIdx.x, true);
    const GWgP p = table[at].p;
    const int bid = blockIdx.x - table[at].first_rblock, nblk = table[at].n_rblk;
    const int Ip = (p.I + 3) & ~3, I4 = Ip >> 2;
    const long n4 = (long)p.T * p.O * I4;
    const long sstride = (long)p.T * p.O * Ip;
    const float* __restrict__ slab = p.slab;
    float* __restrict__ dw = p.dwout;
    for (long e = (long)bid * 256 + threadIdx.x; e < n4; e += (long)nblk * 256) {
        const int i = (int)(e % I4) * 4;
        const long r = e / I4;
        const int o = (int)(r % p.O), t = (int)(r / p.O);
        const float* src = slab + e * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int s0 = 0;
        for (; s0 + 8 <= p.S; s0 += 8) {
            f32x4 b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = *reinterpret_cast<const f32x4*>(src + (s0 + k) * sstride);
#pragma unroll
            for (int k = 0; k < 8; ++k) v += b[k];
        }
        {
            f32x4 b[8];                                   // the tail: clamped loads, masked adds (same order)
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = *reinterpret_cast<const f32x4*>(src + (s0 + k < p.S ? s0 + k : p.S - 1) * sstride);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s0 + k < p.S) v += b[k];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (i + c < p.I) {
                float* d = dw + ((long)o * p.I + i + c) * p.T + t;
                *d = p.accumulate ? *d + v[c] : v[c];
            }
        }
    }
}

void gwgrad_plan(int M, int O, int I, int T, int* S, int* rows, int* ot, int* it) {
    *ot = (O + WTO - 1) / WTO;
    *it = (I + WTI - 1) / WTI;
    const int tiles = *ot * *it * T;
    int s = (768 + tiles - 1) / tiles;               // ~768 workgroups
    const int smax = (M + 8 * WKP - 1) / (8 * WKP);  // at least 8 K steps per split
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    int r = (M + s - 1) / s;
    r = rup(r, WKP);
    s = (M + r - 1) / r;
    *S = s;
    *rows = r;
}

// ------------------------------------------------------------------------------------------------ weight pack (table driven)
// table rows (int64[n][8]): {w_off, wp_off, wpt_off or -1, O, I, T, first_block, -}; a block packs 1024 elements of one conv's
// padded [T][Opad][Ipad] operand (Opad, Ipad = O, I rounded up to 32; the padding is written as zeros every time).
__global__ __launch_bounds__(256) void gpack_kernel(const float* wflat, __bf16* wp, __bf16* wpt, const long* table, int n_desc) {
    int lo = 0, hi = n_desc - 1;
    const long bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 6] <= bid) lo = mid;
        else hi = mid - 1;
    }
    const long* d = table + lo * 8;
    const int O = (int)d[3], I = (int)d[4], T = (int)d[5];
    const int Opad = (O + 31) & ~31, Ipad = (I + 31) & ~31;
    const long n = (long)T * Opad * Ipad;
    const float* w = wflat + d[0];
    const long base = (bid - d[6]) * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long e = base + k * 256 + threadIdx.x;
        if (e >= n) break;
        {
            const int i = (int)(e % Ipad);
            const long r = e / Ipad;
            const int o = (int)(r % Opad), t = (int)(r / Opad);
            const float v = (o < O && i < I) ? w[((long)o * I + i) * T + t] : 0.f;
            wp[d[1] + e] = (__bf16)v;
        }
        if (d[2] >= 0) {
            const int o = (int)(e % Opad);
            const long r = e / Opad;
            const int i = (int)(r % Ipad), t = (int)(r / Ipad);
            const float v = (o < O && i < I) ? w[((long)o * I + i) * T + t] : 0.f;
            wpt[d[2] + e] = (__bf16)v;
        }
    }
}

}  // namespace

extern "C" {

size_t mi_gconv_pack_elems(int O, int I, int kh, int kw) { return (size_t)kh * kw * rup(O, 32) * rup(I, 32); }

int mi_gconv_pack_multi(const float* wflat, void* wp_bf16, void* wpt_bf16, const int64_t* table_dev, int n_desc, int total_blocks, void* stream) {
    MI_REQUIRE(wflat && wp_bf16 && table_dev, "mi_gconv_pack_multi: null operand");
    MI_REQUIRE(n_desc > 0 && total_blocks > 0, "mi_gconv_pack_multi: empty table");
    hipLaunchKernelGGL(gpack_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, wflat, (__bf16*)wp_bf16, (__bf16*)wpt_bf16,
                       (const long*)table_dev, n_desc);
    MI_CHECK_LAUNCH("gpack_kernel");
    return MI_OK;
}

namespace {
struct GFin {
    unsigned* ticket;
    float* out;
    const float* gamma;
    const float* beta;
    float* rm;
    float* rv;
    double count;
    float momentum, eps;
};
}
static int gconv_impl(const void* a, long lda, const void* wp, void* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int mode, const float* bias, float* stats, int out_f32,
                      void* stream, const GFin* fin);

int mi_gconv(const void* a, long lda, const void* wp, void* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
             int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int mode, const float* bias, float* stats, int out_f32,
             void* stream) {
    return gconv_impl(a, lda, wp, out, ldo, B, Ha, Wa, Ca, Ho, Wo, N, kh, kw, sh, sw, ph, pw, dh, dw, mode, bias, stats, out_f32, stream, nullptr);
}

int mi_gconv_bn_inlaunch_max_pixels(void) { return MI_INLAUNCH_MAX_PARTS * GBM; }

int mi_gconv_bn(const void* a, long lda, const void* wp, void* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, const float* bias, float* stats, unsigned* tickets, const float* gamma,
                const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* fin_out, void* stream) {
    MI_REQUIRE(stats && tickets && fin_out, "mi_gconv_bn: null operand");
    MI_REQUIRE((long)B * Ho * Wo <= (long)MI_INLAUNCH_MAX_PARTS * GBM, "mi_gconv_bn: %ld pixels are more than %d row tiles - use mi_gconv + mi_gbn_finalize",
               (long)B * Ho * Wo, MI_INLAUNCH_MAX_PARTS);
    MI_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mi_gconv_bn: running_mean and running_var come together");
    GFin fin{tickets, fin_out, gamma, beta, running_mean, running_var, (double)((long)B * Ho * Wo), momentum, eps};
    return gconv_impl(a, lda, wp, out, ldo, B, Ha, Wa, Ca, Ho, Wo, N, kh, kw, sh, sw, ph, pw, dh, dw, MI_GATHER_FWD, bias, stats, 0, stream, &fin);
}

static int gconv_impl(const void* a, long lda, const void* wp, void* out, long ldo, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N,
                      int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int mode, const float* bias, float* stats, int out_f32,
                      void* stream, const GFin* fin) {
    MI_REQUIRE(a && wp && out, "mi_gconv: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0 && Ca > 0 && N > 0, "mi_gconv: empty shape");
    MI_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0, "mi_gconv: bad conv geometry");
    MI_REQUIRE(mode == MI_GATHER_FWD || mode == MI_GATHER_DGRAD, "mi_gconv: gather mode %d", mode);
    MI_REQUIRE(lda >= Ca && ldo >= N, "mi_gconv: a view's row stride is smaller than its channel count (lda %ld / Ca %d, ldo %ld / N %d)", lda, Ca, ldo, N);
    MI_REQUIRE((reinterpret_cast<uintptr_t>(wp) & 15) == 0, "mi_gconv: the packed weights must be 16-byte aligned");
    MI_REQUIRE(!(out_f32 && stats), "mi_gconv: batch statistics are taken from bf16 outputs");
    MI_REQUIRE((long)B * Ho * Wo < (1L << 31) && (long)B * Ha * Wa < (1L << 31), "mi_gconv: more than 2^31 pixels");
    if (mode == MI_GATHER_FWD) {
        MI_REQUIRE((Ha + 2 * ph - dh * (kh - 1) - 1) / sh + 1 == Ho && (Wa + 2 * pw - dw * (kw - 1) - 1) / sw + 1 == Wo,
                   "mi_gconv: output %dx%d does not follow from input %dx%d", Ho, Wo, Ha, Wa);
    } else {
        MI_REQUIRE((Ho + 2 * ph - dh * (kh - 1) - 1) / sh + 1 == Ha && (Wo + 2 * pw - dw * (kw - 1) - 1) / sw + 1 == Wa,
                   "mi_gconv: gradient input %dx%d does not follow from the forward input %dx%d", Ha, Wa, Ho, Wo);
    }
    GConvP p;
    p.A = (const __bf16*)a;
    p.Wp = (const __bf16*)wp;
    p.out = out;
    p.bias = bias;
    p.stats = stats;
    p.lda = lda;
    p.ldo = ldo;
    p.M = B * Ho * Wo;
    p.N = N;
    p.Ca = Ca;
    p.Cpad = rup(Ca, 32);
    p.Npad = rup(N, 32);
    p.T = kh * kw;
    p.Ha = Ha; p.Wa = Wa; p.Ho = Ho; p.Wo = Wo;
    p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    p.mode = mode;
    p.nchunks = p.Cpad / 32;
    p.remap = mi_sw().gconv_remap;
    p.dbg = mi_sw().gconv_dbg;
    p.fin_ticket = fin ? fin->ticket : nullptr;
    p.fin_out = fin ? fin->out : nullptr;
    p.gamma = fin ? fin->gamma : nullptr;
    p.beta = fin ? fin->beta : nullptr;
    p.running_mean = fin ? fin->rm : nullptr;
    p.running_var = fin ? fin->rv : nullptr;
    p.count = fin ? fin->count : 0.0;
    p.momentum = fin ? fin->momentum : 0.f;
    p.eps = fin ? fin->eps : 0.f;
    const int avec = view_vec(a, lda, Ca);
    int ovec = 1;
    if (!out_f32) ovec = view_vec(out, ldo, N, false);
    else MI_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3) == 0, "mi_gconv: fp32 output must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const long mt = (p.M + GBM - 1) / GBM;
    const int bn32_wgs = mi_sw().gconv_bn32_wgs;
    const bool bn_any = mi_sw().gconv_bn_any != 0;      // MI_GCONV_BN_ANY=0: tile widths 32 / 64 (/ 128) only
    // kernel-row window kernel (gconv3_kernel): three kernel columns, stride 1, output of the input's size, large maps (MI_GCONV3_WGS: from this many 64-wide
    // workgroups; 0 = never)
    const int g3_wgs = mi_sw().gconv3_wgs;       // (GALD 36.56 / 36.27 / 36.16 ms at 1024 / 512 / 256; PraNet indifferent)
    if (!out_f32 && !fin && g3_wgs > 0 && kw == 3 && sh == 1 && sw == 1 && pw == dw && dw <= G3_MAXD && Wo == Wa && Ho == Ha && 2 * ph == dh * (kh - 1) &&
        mt * ((N + 63) / 64) >= g3_wgs && (p.Cpad / 32 >= 4 || g3_wgs == 1)) {
        // measured per shape (tools/gkshape.py gald, MI_GCONV3_WGS = 0 | 1024): the window kernel wins where a kernel row has >= 4 chunks (3x3 142 -> 68 at
        // 6 x 180 x 320: 206 vs 247 us, 466 -> 168 at 90 x 160: 287 vs 391, 218 -> 78: 78 vs 114) and loses with fewer (data gradient 68 -> 142: 239 vs 222);
        // it has the widths 32 / 64 / 80 only (LDS), so a launch whose cost model wants 112 or 16 columns stays on gconv_kernel
        static const int widths3[] = {64, 80, 112, 32, 16};
        int best = 64;
        long best_cost = 1L << 60;
        for (int wdt : widths3) {
            const long cost = (long)((N + wdt - 1) / wdt) * (wdt + 64);
            if (cost < best_cost) best = wdt, best_cost = cost;
        }
        if (g3_wgs == 1 && best != 80 && best != 32) best = 64;          // (tests: every eligible conv)
        if (best == 80 || best == 64 || best == 32) {
            if (best == 80) glaunch3<80>(p, avec, ovec, s);
            else if (best == 32) glaunch3<32>(p, avec, ovec, s);
            else glaunch3<64>(p, avec, ovec, s);
            MI_CHECK_LAUNCH("gconv3_kernel");
            return MI_OK;
        }
    }
    if (out_f32) {
        MI_REQUIRE(N <= 32, "mi_gconv: fp32 outputs are the one-channel side maps (N <= 32), got N = %d", N);
        glaunch_k<32>(p, avec, 1, true, s);
    } else if (fin || !bn_any) {                            // (the in-launch finalize lives in the 32- / 64-wide instances)
        if (N <= 32) glaunch_k<32>(p, avec, ovec, false, s);
        else if (mt * ((N + 63) / 64) < bn32_wgs) glaunch_k<32>(p, avec, ovec, false, s);      // few pixels (1/16, 1/32 resolution): narrower tiles, more workgroups
        else if (N <= 64 || N % 128 == 64 || mt * ((N + 127) / 128) < 512 || !mi_sw().gconv_bn128 || fin) glaunch_k<64>(p, avec, ovec, false, s);
        else glaunch_k<128>(p, avec, ovec, false, s);
    } else if (N > 32 && mt * ((N + 63) / 64) < bn32_wgs) glaunch_k<32>(p, avec, ovec, false, s);
    else {
        // Tile width by a cost model fitted on HarDNet's shapes (tools/dbg/bn_width.sh: every width forced on every shape): a column tile costs its width
        // plus ~64 columns' worth of fixed work (the A rows it re-reads, prologue, epilogue), so cost = ceil(N / w) * (w + 64).  N = 68 -> one 80-wide tile
        // (248 vs 347 us for two 64-wide), 334 -> three 112-wide (221 vs 318), 256 -> four 64-wide; the least-padding rule tried first chose 16-wide tiles
        // for N = 168 (1 270 us against 393).  MI_GCONV_BN_FORCE: one width for everything (measurement).
        static const int widths[] = {64, 80, 112, 32, 16};
        const int force = mi_sw().gconv_bn_force;
        const int fixed = mi_sw().gconv_bn_c;
        int best = 64;
        long best_cost = 1L << 60;
        for (int wdt : widths) {
            const long cost = (long)((N + wdt - 1) / wdt) * (wdt + fixed);
            if (cost < best_cost) best = wdt, best_cost = cost;
        }
        // (only where the wider tiles still leave the chip full: on the 11 x 11 / 22 x 22 maps a 2048-channel data gradient took 26 us as 304 112-wide
        //  workgroups against 20 us as 512 64-wide ones)
        if (best > 64 && mt * ((N + best - 1) / best) < 384) best = 64;
        if (force) best = force;
        switch (best) {
            case 16: glaunch_k<16>(p, avec, ovec, false, s); break;
            case 32: glaunch_k<32>(p, avec, ovec, false, s); break;
            case 80: glaunch_k<80>(p, avec, ovec, false, s); break;
            case 112: glaunch_k<112>(p, avec, ovec, false, s); break;
            default: glaunch_k<64>(p, avec, ovec, false, s); break;
        }
    }
    MI_CHECK_LAUNCH("gconv_kernel");
    return MI_OK;
}

size_t mi_gconv_stats_elems(int B, int Ho, int Wo, int N) { return (size_t)(((long)B * Ho * Wo + GBM - 1) / GBM) * 2 * N; }

size_t mi_gconv_wgrad_workspace(int B, int Ho, int Wo, int O, int I, int kh, int kw) {
    int S, rows, ot, it;
    const long M = (long)B * Ho * Wo;
    if (M <= 0 || M >= (1L << 31)) return 0;          // mi_gconv_wgrad refuses such a shape
    gwgrad_plan((int)M, O, I, kh * kw, &S, &rows, &ot, &it);
    int S3 = 0;
    if (kw == 3) gwgrad_plan((int)M, O, I, kh, &S3, &rows, &ot, &it);       // the fused-row kernel splits K for a third of the workgroups
    if (S3 > S) S = S3;
    return (size_t)S * kh * kw * O * ((I + 3) & ~3) * sizeof(float);
}

int mi_gconv_wgrad(const void* dy, long ldy, const void* x, long ldx, float* dw, int B, int Ha, int Wa, int I, int Ho, int Wo, int O,
                   int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw_, int accumulate, void* workspace, size_t workspace_bytes,
                   unsigned* tickets, int n_tickets, void* stream) {
    MI_REQUIRE(dy && x && dw && workspace, "mi_gconv_wgrad: null operand");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && Ho > 0 && Wo > 0 && I > 0 && O > 0, "mi_gconv_wgrad: empty shape");
    MI_REQUIRE(ldy >= O && ldx >= I, "mi_gconv_wgrad: a view's row stride is smaller than its channel count");
    MI_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && dh > 0 && dw_ > 0 && ph >= 0 && pw >= 0, "mi_gconv_wgrad: bad conv geometry");
    MI_REQUIRE((long)B * Ho * Wo < (1L << 31) && (long)B * Ha * Wa < (1L << 31), "mi_gconv_wgrad: more than 2^31 pixels");
    MI_REQUIRE((Ha + 2 * ph - dh * (kh - 1) - 1) / sh + 1 == Ho && (Wa + 2 * pw - dw_ * (kw - 1) - 1) / sw + 1 == Wo,
               "mi_gconv_wgrad: output %dx%d does not follow from input %dx%d", Ho, Wo, Ha, Wa);
    MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "mi_gconv_wgrad: workspace must be 16-byte aligned");
    GWgP p;
    p.remap = mi_sw().gconv_remap;
    p.dbg = mi_sw().gw_dbg;
    p.dY = (const __bf16*)dy;
    p.X = (const __bf16*)x;
    p.slab = (float*)workspace;
    p.ldy = ldy;
    p.ldx = ldx;
    p.M = B * Ho * Wo;
    p.O = O; p.I = I; p.T = kh * kw;
    p.Ho = Ho; p.Wo = Wo; p.Ha = Ha; p.Wa = Wa;
    p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw_;
    // fused kernel row (gwgrad3_kernel): three-column kernels at stride 1 whose output has the input's height and width
    // (measured, bench.py --workload gald / pranet: GALD's weight gradients 9.96 -> 8.90 ms per step with every eligible conv fused, PraNet's 5.02 -> 5.23:
    //  on the small maps of the deep stages a third of the workgroups leaves the launch latency-bound on fewer CUs - fused from 65 536 pixels up, or when forced)
    const bool fused = kw == 3 && sh == 1 && sw == 1 && pw == dw_ && Ho == Ha && Wo == Wa && dw_ <= W3_MAXD &&
                       (mi_sw().gwgrad3 == 2 || (mi_sw().gwgrad3 == 1 && p.M >= 65536));
    gwgrad_plan(p.M, O, I, fused ? kh : p.T, &p.S, &p.rows_per_split, &p.o_tiles, &p.i_tiles);
    const size_t need = (size_t)p.S * p.T * O * ((I + 3) & ~3) * sizeof(float);
    if (workspace_bytes < need) return mi_set_error(MI_ENOMEM, "mi_gconv_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const int yv = view_vec(dy, ldy, O), xv = view_vec(x, ldx, I);
    if (fused) {
        p.ticket = nullptr;
        p.dwout = dw;
        p.accumulate = accumulate;
        const dim3 grid3(p.o_tiles * p.i_tiles * kh * p.S);
#define GW3(YV, XV) hipLaunchKernelGGL((gwgrad3_kernel<YV, XV>), grid3, dim3(256), 0, s, p)
        if (yv == 8 && xv == 8) GW3(8, 8);
        else if (yv == 8 && xv == 4) GW3(8, 4);
        else if (yv == 8) GW3(8, 1);
        else if (yv == 4 && xv == 8) GW3(4, 8);
        else if (yv == 4 && xv == 4) GW3(4, 4);
        else if (yv == 4) GW3(4, 1);
        else if (xv == 8) GW3(1, 8);
        else if (xv == 4) GW3(1, 4);
        else GW3(1, 1);
#undef GW3
        MI_CHECK_LAUNCH("gwgrad3_kernel");
        const long n3 = (long)O * I * p.T;
        const int blocks3 = (int)((n3 + 31) / 32 < 4096 ? (n3 + 31) / 32 : 4096);
        hipLaunchKernelGGL(gwgrad_reduce_kernel, dim3(blocks3), dim3(256), 0, s, p.slab, dw, O, I, p.T, p.S, accumulate);
        MI_CHECK_LAUNCH("gwgrad_reduce_kernel");
        return MI_OK;
    }
    // few K splits: the slabs are added by the last workgroup of each (tap, tile) inside this launch (needs one zeroed ticket word per (tap, tile))
    const bool inlaunch = tickets && p.S <= GW_INLAUNCH_S && p.o_tiles * p.i_tiles * p.T <= n_tickets;
    p.ticket = inlaunch ? tickets : nullptr;
    p.dwout = dw;
    p.accumulate = accumulate;
    const dim3 grid(p.o_tiles * p.i_tiles * p.T * p.S);
#define GW(YV, XV) hipLaunchKernelGGL((gwgrad_kernel<YV, XV>), grid, dim3(256), 0, s, p)
    if (yv == 8 && xv == 8) GW(8, 8);
    else if (yv == 8 && xv == 4) GW(8, 4);
    else if (yv == 8) GW(8, 1);
    else if (yv == 4 && xv == 8) GW(4, 8);
    else if (yv == 4 && xv == 4) GW(4, 4);
    else if (yv == 4) GW(4, 1);
    else if (xv == 8) GW(1, 8);
    else if (xv == 4) GW(1, 4);
    else GW(1, 1);
#undef GW
    MI_CHECK_LAUNCH("gwgrad_kernel");
    if (inlaunch) return MI_OK;
    const long n = (long)O * I * p.T;
    const int blocks = (int)((n + 31) / 32 < 4096 ? (n + 31) / 32 : 4096);          // 32 output elements (8 lanes each) per block
    hipLaunchKernelGGL(gwgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slab, dw, O, I, p.T, p.S, accumulate);
    MI_CHECK_LAUNCH("gwgrad_reduce_kernel");
    return MI_OK;
}


// ---- the weight gradients of many convs in one go (see GWgD above).  jobs: host array; table_dev: device buffer of mi_gconv_wgrad_multi_table_bytes(n)
// bytes; workspace: device buffer of mi_gconv_wgrad_multi_workspace(jobs, n) bytes.  Two jobs must not share a dw (the caller flushes between them).
static bool gwm_fused(const MiWgradJob& j, long M) {
    const int mode = mi_sw().gwgrad3;                    // 0: never fused; otherwise fused whenever the geometry allows (MI_GWM_FUSED3=0: the one-conv rule, from 65 536 pixels)
    const int always = mi_sw().gwm_fused3;
    const bool can = j.kw == 3 && j.sh == 1 && j.sw == 1 && j.pw == j.dw_ && j.Ho == j.Ha && j.Wo == j.Wa && j.dw_ <= W3_MAXD;
    return can && mode != 0 && (always || mode == 2 || M >= 65536);
}
static void gwm_plan(const MiWgradJob& j, GWgD& d) {
    const int steps = mi_sw().gwm_steps;
    GWgP& p = d.p;
    p.remap = mi_sw().gconv_remap;
    p.dbg = 0;
    p.dY = (const __bf16*)j.dy;
    p.X = (const __bf16*)j.x;
    p.ldy = j.ldy;
    p.ldx = j.ldx;
    p.M = j.B * j.Ho * j.Wo;
    p.O = j.O; p.I = j.I; p.T = j.kh * j.kw;
    p.Ho = j.Ho; p.Wo = j.Wo; p.Ha = j.Ha; p.Wa = j.Wa;
    p.kw = j.kw; p.sh = j.sh; p.sw = j.sw; p.ph = j.ph; p.pw = j.pw; p.dh = j.dh; p.dw = j.dw_;
    p.o_tiles = (j.O + WTO - 1) / WTO;
    p.i_tiles = (j.I + WTI - 1) / WTI;
    int S = (p.M + WKP * (steps > 0 ? steps : 48) - 1) / (WKP * (steps > 0 ? steps : 48));
    if (S < 1) S = 1;
    int r = rup((p.M + S - 1) / S, WKP);
    p.rows_per_split = r;
    p.S = (p.M + r - 1) / r;
    p.ticket = nullptr;
    p.dwout = j.dw;
    p.accumulate = j.accumulate;
    const bool fused = gwm_fused(j, p.M);
    d.nblk = p.o_tiles * p.i_tiles * (fused ? j.kh : p.T) * p.S;
    const long n4 = (long)j.O * ((j.I + 3) / 4) * p.T;                  // the reducer's threads: four consecutive i each
    d.n_rblk = (int)((n4 + 255) / 256 < 512 ? (n4 + 255) / 256 : 512);
}
static inline size_t gwm_slab_bytes(const GWgD& d) { return ((size_t)d.p.S * d.p.T * d.p.O * ((d.p.I + 3) & ~3) * sizeof(float) + 255) & ~(size_t)255; }

size_t mi_gconv_wgrad_multi_table_bytes(int n) { return (size_t)(n > 0 ? n : 0) * sizeof(GWgD); }

size_t mi_gconv_wgrad_multi_workspace(const MiWgradJob* jobs, int n) {
    size_t total = 0;
    for (int k = 0; k < n; ++k) {
        const MiWgradJob& j = jobs[k];
        if ((long)j.B * j.Ho * j.Wo <= 0 || (long)j.B * j.Ho * j.Wo >= (1L << 31) || j.O <= 0 || j.I <= 0 || j.kh <= 0 || j.kw <= 0) return 0;
        GWgD d;
        gwm_plan(j, d);
        total += gwm_slab_bytes(d);
    }
    return total;
}

int mi_gconv_wgrad_multi(const MiWgradJob* jobs, int n, void* table_dev, size_t table_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    if (n == 0) return MI_OK;
    MI_REQUIRE(jobs && n > 0 && table_dev && workspace, "mi_gconv_wgrad_multi: null operand");
    MI_REQUIRE(table_bytes >= mi_gconv_wgrad_multi_table_bytes(n), "mi_gconv_wgrad_multi: table buffer %zu < %zu bytes", table_bytes, mi_gconv_wgrad_multi_table_bytes(n));
    MI_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(table_dev) & 15) == 0, "mi_gconv_wgrad_multi: buffers must be 16-byte aligned");
    std::vector<GWgD> table((size_t)n);
    std::vector<int> cls((size_t)n);                        // 0 .. 8: per-tap kernel by (yv, xv); 9 .. 17: fused-row kernel
    auto vidx = [](int v) { return v == 8 ? 0 : (v == 4 ? 1 : 2); };
    std::vector<int> order;
    size_t need = 0;
    for (int k = 0; k < n; ++k) {
        const MiWgradJob& j = jobs[k];
        MI_REQUIRE(j.dy && j.x && j.dw, "mi_gconv_wgrad_multi: job %d: null operand", k);
        MI_REQUIRE(j.B > 0 && j.Ha > 0 && j.Wa > 0 && j.Ho > 0 && j.Wo > 0 && j.I > 0 && j.O > 0, "mi_gconv_wgrad_multi: job %d: empty shape", k);
        MI_REQUIRE(j.ldy >= j.O && j.ldx >= j.I, "mi_gconv_wgrad_multi: job %d: a view's row stride is smaller than its channel count", k);
        MI_REQUIRE(j.kh > 0 && j.kw > 0 && j.sh > 0 && j.sw > 0 && j.dh > 0 && j.dw_ > 0 && j.ph >= 0 && j.pw >= 0, "mi_gconv_wgrad_multi: job %d: bad conv geometry", k);
        MI_REQUIRE((long)j.B * j.Ho * j.Wo < (1L << 31) && (long)j.B * j.Ha * j.Wa < (1L << 31), "mi_gconv_wgrad_multi: job %d: more than 2^31 pixels", k);
        MI_REQUIRE((j.Ha + 2 * j.ph - j.dh * (j.kh - 1) - 1) / j.sh + 1 == j.Ho && (j.Wa + 2 * j.pw - j.dw_ * (j.kw - 1) - 1) / j.sw + 1 == j.Wo,
                   "mi_gconv_wgrad_multi: job %d: output %dx%d does not follow from input %dx%d", k, j.Ho, j.Wo, j.Ha, j.Wa);
        for (int q = 0; q < k; ++q) MI_REQUIRE(jobs[q].dw != j.dw, "mi_gconv_wgrad_multi: jobs %d and %d write the same gradient", q, k);
        gwm_plan(j, table[k]);
        cls[k] = vidx(view_vec(j.dy, j.ldy, j.O)) * 3 + vidx(view_vec(j.x, j.ldx, j.I)) + (gwm_fused(j, table[k].p.M) ? 9 : 0);
        need += gwm_slab_bytes(table[k]);
    }
    if (workspace_bytes < need) return mi_set_error(MI_ENOMEM, "mi_gconv_wgrad_multi: workspace %zu < %zu bytes", workspace_bytes, need);
    // table order: by class; inside a class the job order
    std::vector<GWgD> sorted;
    sorted.reserve((size_t)n);
    int cls_first[18], cls_n[18], cls_blocks[18];
    size_t off = 0;
    int rblocks = 0;
    for (int c = 0; c < 18; ++c) {
        cls_first[c] = (int)sorted.size();
        int blocks = 0;
        for (int k = 0; k < n; ++k) {
            if (cls[k] != c) continue;
            GWgD d = table[k];
            d.p.slab = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
            off += gwm_slab_bytes(d);
            d.first_block = blocks;
            blocks += (d.nblk + 7) & ~7;
            d.first_rblock = rblocks;
            rblocks += d.n_rblk;
            sorted.push_back(d);
        }
        cls_n[c] = (int)sorted.size() - cls_first[c];
        cls_blocks[c] = blocks;
    }
    hipStream_t s = (hipStream_t)stream;
    GWgD* tdev = static_cast<GWgD*>(table_dev);
    for (int first = 0; first < n; first += GW_CHUNK) {
        GWgChunk c;
        const int count = n - first < GW_CHUNK ? n - first : GW_CHUNK;
        memset(&c, 0, sizeof(c));
        for (int k = 0; k < count; ++k) c.d[k] = sorted[(size_t)first + k];
        hipLaunchKernelGGL(gw_table_kernel, dim3(1), dim3(256), 0, s, tdev, first, count, c);
    }
    MI_CHECK_LAUNCH("gw_table_kernel");
    static const int vv[3] = {8, 4, 1};
    for (int c = 0; c < 18; ++c) {
        if (!cls_n[c]) continue;
        const int yv = vv[(c % 9) / 3], xv = vv[c % 3];
        const dim3 grid((unsigned)cls_blocks[c]);
#define GWM(K, YV, XV) hipLaunchKernelGGL((K<YV, XV>), grid, dim3(256), 0, s, (const GWgD*)tdev, cls_first[c], cls_n[c])
#define GWM_ALL(K)                                   \
        if (yv == 8 && xv == 8) GWM(K, 8, 8);        \
        else if (yv == 8 && xv == 4) GWM(K, 8, 4);   \
        else if (yv == 8) GWM(K, 8, 1);              \
        else if (yv == 4 && xv == 8) GWM(K, 4, 8);   \
        else if (yv == 4 && xv == 4) GWM(K, 4, 4);   \
        else if (yv == 4) GWM(K, 4, 1);              \
        else if (xv == 8) GWM(K, 1, 8);              \
        else if (xv == 4) GWM(K, 1, 4);              \
        else GWM(K, 1, 1);
        if (c < 9) { GWM_ALL(gwgrad_multi_kernel) } else { GWM_ALL(gwgrad3_multi_kernel) }
#undef GWM_ALL
#undef GWM
    }
    MI_CHECK_LAUNCH("gwgrad_multi_kernel");
    hipLaunchKernelGGL(gwgrad_reduce_multi_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, (const GWgD*)tdev, n);
    MI_CHECK_LAUNCH("gwgrad_reduce_multi_kernel");
    return MI_OK;
}

}  // extern "C"
