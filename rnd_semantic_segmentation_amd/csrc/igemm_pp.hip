// Implicit-GEMM convolution, wide-tile "ping-pong" main loop for gfx950 (bf16 operands, fp32 MFMA accumulate).
//
//   out[m][n] = epi( sum_t sum_c A[src(m,t)][c] * Wp[t][n][c] )      same contract and epilogue as igemm_nt.hip, for the
// stride-1 convs / plain GEMMs with a long contraction (K = taps * Ca >= 512): the dilated 3x3 family of layer3 / layer4
// (reference core/components/resnet.py:22-25, 100) and the ASPP head's GEMMs (classifiers/aspp/classifier.py:26-29).
//
// Why a second main loop: the 128-wide kernel moves 13-14 bytes from the L2 into LDS per kFLOP, and with the step's cold operands
// what fills the LDS is the scarce resource - the lever is bytes per FLOP, i.e. the tile.  This kernel computes a (2*16*MTG) x 256
// tile per workgroup (320 x 256: 7.0 B/kFLOP against 13-14 for the 128-wide tiles) with ONE workgroup of 8 waves per CU, and
// replaces the co-resident second workgroup (which used to hide the DMA drain and the barrier of every K-step) by structure:
//   * K advances in 32-channel slabs through a 4-slot LDS ring; three slabs are always in flight (buffer_load ... lds, counted
//     s_waitcnt vmcnt, raw s_barrier - a __syncthreads() would drain the DMA queue).  A 3x3 contraction runs channel-chunk-major
//     (all nine taps of a 32-channel chunk, then the next chunk: the shifted re-reads of the input rows hit the L2);
//   * the 8 waves are two groups of 4 (one wave of each group per SIMD).  Group 0 owns the upper half of the tile's rows,
//     group 1 the lower half, each wave 16*MTG rows x 64 columns.  The groups run the same loop half a phase apart: while one
//     group issues its MFMAs for slab p, the other reads its fragments of slab p (or p+1) from LDS and issues its share of
//     the DMA for slab p+3.  The matrix pipe of a SIMD is thus handed from one wave to the other at every barrier and never
//     waits for LDS or DMA.
// Intervals (I_k lies between barriers B_k and B_k+1):    I_2p: G0 read(p) | G1 mfma(p-1)      I_2p+1: G0 mfma(p) | G1 read(p)
//   every wave waits for ITS DMA pieces of slab p+1 (vmcnt, two younger slabs stay in flight) at the end of I_2p+1, so that
//   after B_2p+2 the slab is visible to group 0 and after B_2p+3 to group 1;
//   slab p+3 lands in the slot of slab p-1, whose last reader (G1 in I_2p-1) finished its ds_reads (lgkmcnt(0)) before B_2p.
// (Tried and dropped: issuing a slab's DMA pieces inside the MFMA segment instead of the read segment - see PP_DMA_SEG below.)
// RR (template parameter; round 5, experiment builds only): the "rolling" main loop.  No groups and no hand-over: BOTH waves of a SIMD stream MFMAs all the
//   time, each wave software-pipelines its own fragment reads - the A fragment of the row tile four ahead goes into a five-slot register ring (20 VGPRs
//   instead of 40), the next slab's four W fragments into a second set right behind the slab's one barrier (which sits in the middle of the slab: own DMA
//   pieces of slab s+1 landed -> barrier -> issue the pieces of slab s+3, read slab s+1).  229 VGPRs, no spills, the waits are the compiler's counted
//   lgkmcnt.  Same tile, same LDS image, same order of the fp32 sums: bit-equal outputs.  Measured (tools/rrexp.py, profiles/r05_pp_rolling_loop_ab.txt):
//   3x3 256 85.4 vs 82.0 us (main loop alone 70.3 vs 69.4), 3x3 512 287.5 vs 278.9, 1x1 2048 -> 512 164 vs 152; the training step 291.5 vs 296.5 images/s.
//   The loop was not waiting for the hand-over: at 1.28 PFLOP/s on random operands it runs at the clock the chip allows under this load
//   (MI355X_MICROARCH.md "DVFS give-back": cycles saved in an MFMA-dense loop return as a lower clock), so a denser issue stream buys nothing.
// LDS image of a slab: A rows then W rows, 64 B (32 channels) per row, lane-linear per 1-KiB DMA piece (16 rows); the 16-B chunk
// c of row r sits at chunk c ^ s(r), applied to the per-lane SOURCE address and to the ds_read_b128 address, with
// s_A(r) = (-(r >> 2)) & 3 and s_W(r) = ((r >> 3) & 1) << 1: both make every ds_read_b128 lane group hit 64 distinct banks
// (brute-forced over the instruction's four 16-lane groups; W rows are read in the permuted order of igemm_nt.hip).
#include <type_traits>
#include "igemm_common.h"
#include <stdlib.h>

namespace {

template <int MTG> struct PPGeo {
    static constexpr int BMG = 16 * MTG, BM = 2 * BMG, BN = 256;
    static constexpr int SLAB_A = BM * 64, SLAB_W = BN * 64, SLAB = SLAB_A + SLAB_W;
    static constexpr int NSLOT = 4, LDS_BYTES = NSLOT * SLAB;          // MTG 10: 144 KiB, MTG 8: 128 KiB
    static constexpr int PA = BM / 16;                                // A pieces per slab (16 rows each), moved by group 0
    static constexpr int NPA = PA / 4;                                // ... per wave of group 0
    static constexpr int NPW = 4;                                     // W pieces per wave of group 1 (16 pieces per slab)
    static_assert(PA % 4 == 0, "the A pieces must split evenly over the four waves of group 0");
};

#ifndef PP_PRIO
#define PP_PRIO 0         // 0: no priority changes (default: 1 and 2 measure the same within 1 %, the 256-row tile 3 % slower with 1); 1: s_setprio 1 around every MFMA segment; 2: static priority 1 for the younger group (waves 4-7)
#endif
#ifndef PP_DMA_SEG
#define PP_DMA_SEG 0      // 0: the DMA pieces of the slab three ahead are issued in the read segment, after the fragment reads; 1: between the MFMAs
#endif
#ifdef MI_PP_TRACE
// Timeline experiment (tools/pptrace.py builds a second library with -DMI_PP_TRACE; never defined in the product build): wave 0 of each
// group of ONE workgroup stamps s_memtime at six points of every k-step into spare LDS, dumped to this buffer at the end.
constexpr int PP_TRACE_STEPS = 80, PP_TRACE_PTS = 6;
__device__ unsigned g_pp_trace[2 * PP_TRACE_STEPS * PP_TRACE_PTS];
__device__ unsigned g_pp_clock[4];
#define PP_T(k)                                                                                                        \
    if (tr_on && s < PP_TRACE_STEPS) {                                                                                 \
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();                                                    \
        if (lane == 0) tr[s * PP_TRACE_PTS + (k)] = t_;                                                                \
    }
#else
#define PP_T(k)
#endif

template <int MTG, int EPI, bool RR = false>
__global__ __launch_bounds__(512, 1) void igemm_pp_kernel(IgemmParams p) {
    using G = PPGeo<MTG>;
    constexpr int BM = G::BM, BN = G::BN, BMG = G::BMG, SLAB = G::SLAB, SLAB_A = G::SLAB_A, NPA = G::NPA, NPW = G::NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int nwg = p.m_tiles * p.n_tiles;
    const int tile = mi_xcd_remap(blockIdx.x, nwg);
    const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- DMA roles: group 0 moves the A rows (NPA pieces per wave), group 1 the W rows (NPW pieces per wave) ------------------
    // Buffer addressing: one resource per group (A shifted down by the largest negative tap excursion, or W), a 32-bit byte
    // offset per piece and lane (tap with the lowest address, channel 0, + swizzled chunk) and the tap / channel-chunk
    // displacement in the scalar offset.  A chunk that falls into the padding, or a row past M, gets bit 31 of its offset set:
    // out of range for the resource, which then returns zeros - two VALU instructions per piece and k-step, no 64-bit selects.
    const int prow = lane >> 2, pch = lane & 3;
    const int sgn = (p.mode == MI_GATHER_FWD) ? 1 : -1;
    const bool pointwise = p.T == 1 && p.pad == 0;
    constexpr int NPMIN = NPA < NPW ? NPA : NPW, NPMAX = NPA > NPW ? NPA : NPW;
    const int span = (p.ksz - 1) * p.dil;                              // distance between the first and the last tap of a row / column
    const int neg = pointwise ? 0 : (sgn > 0 ? p.pad : (span > p.pad ? span - p.pad : 0));     // rows / columns a source may lie before pixel 0
    const long bias = ((long)neg * p.Wa + neg) * p.Ca * 2;
    const char* rbase = grp == 0 ? reinterpret_cast<const char*>(p.A) - bias : reinterpret_cast<const char*>(p.Wp);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rbase), 0, 0x7ffffff0, 0x00020000);
    unsigned d_off[NPMAX];                           // per piece: byte offset of the lowest-address tap, channel 0 (+ swizzled chunk)
    unsigned d_inv[NPMAX];                           // A: bit t set = tap t of this row lies in the padding (or the row is past M); W: 0
    if (grp == 0) {
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int row = (wq * NPA + i) * 16 + prow;
            const int m = m0 + row;
            const bool ok = m < p.M;
            const int chunk = (pch ^ ((-(row >> 2)) & 3)) * 16;
            if (pointwise) {
                d_off[i] = (unsigned)((long)(ok ? m : 0) * p.Ca * 2 + chunk);
                d_inv[i] = ok ? 0u : 0x1ffu;
                continue;
            }
            const int mm = ok ? m : 0;
            const int b = mm / HoWo, rem = mm - b * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            const int h0 = ho - sgn * p.pad, w0 = wo - sgn * p.pad;          // tap (0,0) source; Ha == Ho, Wa == Wo (stride 1)
            const int hl = sgn > 0 ? h0 : h0 - span, wl = sgn > 0 ? w0 : w0 - span;       // the tap with the lowest address
            d_off[i] = (unsigned)(bias + (((long)b * p.Ha + hl) * p.Wa + wl) * p.Ca * 2 + chunk);
            unsigned rb = 0, cb = 0;
#pragma unroll
            for (int kq = 0; kq < 3; ++kq) {
                const bool live = kq < p.ksz;
                rb |= (unsigned)(live & ((unsigned)(h0 + sgn * kq * p.dil) < (unsigned)p.Ha)) << kq;
                cb |= (unsigned)(live & ((unsigned)(w0 + sgn * kq * p.dil) < (unsigned)p.Wa)) << kq;
            }
            unsigned msk;
            if (p.ksz == 1) msk = rb & cb & 1u;
            else msk = ((rb & 1u) ? cb : 0u) | ((rb & 2u) ? cb << 3 : 0u) | ((rb & 4u) ? cb << 6 : 0u);
            d_inv[i] = ok ? ~msk & 0x1ffu : 0x1ffu;
        }
#pragma unroll
        for (int i = NPA; i < NPMAX; ++i) d_off[i] = 0u, d_inv[i] = 0x1ffu;          // pieces only group 1 has
    } else {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int rl = (wq * NPW + i) * 16 + prow;
            const int n = n0 + rl;
            const int chunk = (pch ^ (((rl >> 3) & 1) << 1)) * 16;
            d_off[i] = (unsigned)((long)(n < p.N ? n : 0) * p.Ca * 2 + chunk);      // rows past N re-read row 0: their accumulator columns are never stored
            d_inv[i] = 0u;
        }
#pragma unroll
        for (int i = NPW; i < NPMAX; ++i) d_off[i] = 0u, d_inv[i] = 0x1ffu;          // pieces only group 0 has
    }
    const int spt = p.Ca >> 5;                          // slabs per tap
    const bool tap_inner = (p.korder & 1) != 0;
    const int ns = p.T * spt;
    const unsigned w_tap_bytes = (unsigned)p.N * p.Ca * 2;
    int ld_s = 0, ld_t = 0, ld_c = 0;                   // next slab to stage: index, tap, 32-channel chunk within the tap
    unsigned ld_toff = 0;                               // byte displacement of the tap (A: spatial shift from the lowest tap, W: tap plane)

    auto tap_offset = [&](int t) -> unsigned {
        if (grp != 0) return (unsigned)t * w_tap_bytes;
        const int ky = p.ksz == 3 ? (t * 11) >> 5 : 0, kx = t - ky * p.ksz;          // t / 3 for t < 9
        const int dy = sgn > 0 ? ky : p.ksz - 1 - ky, dx = sgn > 0 ? kx : p.ksz - 1 - kx;    // data gradient: tap (ky, kx) reads out - (k - 1) * dil
        return (unsigned)((dy * p.dil * p.Wa + dx * p.dil) * p.Ca * 2);
    };
    ld_toff = tap_offset(0);
    auto advance_slab = [&]() {
        ++ld_s;
        if (tap_inner) {
            // all taps of one 32-channel chunk back to back: the nine shifted reads of the same input rows follow each other
            // within microseconds and hit L2 (a workgroup's rows + halo are ~90 KB per chunk; TCC hit rate 71 % -> 91 %, fabric
            // fetches 6x lower at 3x3 256); tap-major order sweeps the whole 38-77 MB tensor between two reads of a row
            if (++ld_t == p.T) {
                ld_t = 0;
                ++ld_c;
            }
            ld_toff = tap_offset(ld_t);
        } else if (++ld_c == spt) {
            ld_c = 0;
            ++ld_t;
            ld_toff = tap_offset(ld_t);
        }
    };
    unsigned src[NPMAX];                                              // this wave's piece offsets of the slab being staged
    unsigned src_soff = 0;
    auto piece_sources = [&]() {                                      // read segment: VALU work under the fragment reads' latency
        const bool live = ld_s < ns;                                  // past the last slab: re-read slab 0 (scalar selects only)
        src_soff = live ? ld_toff + (unsigned)ld_c * 64u : tap_offset(0);
        const int tap = live ? ld_t : 0;
#pragma unroll
        for (int i = 0; i < NPMAX; ++i) {
            src[i] = d_off[i] | (((d_inv[i] >> tap) & 1u) << 31);
            asm volatile("" : "+v"(src[i]));                          // materialise here: left alone, hipcc sinks the work between the MFMAs
        }
    };
    auto stage_next = [&]() {                           // prologue: a whole slab's pieces back to back
        char* dst = smem + (ld_s & 3) * SLAB + (grp == 0 ? wq * (NPA * 1024) : SLAB_A + wq * (NPW * 1024));
        piece_sources();
#pragma unroll
        for (int i = 0; i < NPMAX; ++i)
            if (i < NPMIN || (NPA > NPW ? grp == 0 : grp == 1)) blds16(rsrc, src[i], src_soff, dst + i * 1024);
        advance_slab();
    };

    // ---- compute roles ---------------------------------------------------------------------------------------------------------
    f32x4 acc[4][MTG];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MTG; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    const int a_off = (grp * BMG + frow) * 64 + ((fq ^ ((-(frow >> 2)) & 3)) << 4);                       // + j * 1024
    const int w_off = SLAB_A + (wq * 64 + 8 * (frow >> 2) + (frow & 3)) * 64 + ((fq ^ (((frow >> 2) & 1) << 1)) << 4);   // + (32*(i>>1) + 4*(i&1)) * 64
    bf16x8 wf[4], af[MTG];
    auto read_frags = [&](int s) {
        const char* base = smem + (s & 3) * SLAB;
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + w_off + (32 * (i >> 1) + 4 * (i & 1)) * 64);
#pragma unroll
        for (int j = 0; j < MTG; ++j) af[j] = *reinterpret_cast<const bf16x8*>(base + a_off + j * 1024);
    };
    // Where the DMA pieces of the slab three ahead are issued (PP_DMA_SEG).  s_memtime timeline of one workgroup
    // (tools/pptrace.py; the stamps cost 13-28 % themselves, so only the proportions count), 3x3 256:
    //   * round-2 first version: pieces at the top of the read segment, each source a 64-bit select against the zero page
    //     (7 VALU per piece).  The partner wave on the SIMD is in its MFMA segment at s_setprio 1, and the low-priority wave gets
    //     about one vector issue slot per MFMA: the 35 VALU + 5 DMA took 600-770 ticks, the read segment 1.2k against an MFMA
    //     segment of 0.8k (2.64k ticks per k-step).
    //   * buffer addressing (2 VALU per piece, below), pieces still in the read segment: read segment 0.65k, MFMA segment
    //     0.9-0.95k, 2.1k ticks per k-step; wall time of the launch 90 -> 83 us (3x3 256), 297 -> 280 us (3x3 512).
    //   * pieces between the MFMAs instead (PP_DMA_SEG 1): each costs the MFMA stream 33-40 ticks (the in-order wave cannot issue
    //     its next MFMA until the texture path has taken the instruction); 1-3 % slower in wall time.
    // The shader clock during the loop is 1.7-1.9 GHz (s_memtime / s_memrealtime; a bare MFMA loop on every CU runs at 2.1-2.4 GHz,
    // tools/micro/l2lds.hip clock probe): the 2.5 PFLOP/s peak is a 2.4 GHz figure this kernel's power draw does not allow.
    // Past the last slab the pieces re-read slab 0 into the slot that slab s-1 has left (nobody reads it again): the vmcnt
    // arithmetic is the same in every step.
    auto mfma_and_stage = [&]() {
        char* dst = smem + (ld_s & 3) * SLAB + (grp == 0 ? wq * (NPA * 1024) : SLAB_A + wq * (NPW * 1024));
#pragma unroll
        for (int j = 0; j < MTG; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
            if ((j & 1) == 0 && (j >> 1) < NPMAX) {
                const int i = j >> 1;
#if PP_DMA_SEG
                __builtin_amdgcn_sched_barrier(0);
                // one instruction stream for both groups (W rows carry an all-ones mask); only the pieces one group has more of
                // than the other sit behind a (wave-uniform) branch
                if (i < NPMIN || (NPA > NPW ? grp == 0 : grp == 1)) blds16(rsrc, src[i], src_soff, dst + i * 1024);
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        advance_slab();
    };

    if constexpr (RR) {
        // ---- rolling main loop (round 5; the header's "RR" paragraph) ------------------------------------------------------------
        constexpr int RING = (MTG % 5 == 0) ? 5 : 4, LEAD = RING - 1, BR = MTG - NPMAX;
        static_assert(MTG % RING == 0 && BR >= 0 && BR <= MTG - LEAD && MTG >= 4, "rolling loop: ring / barrier row do not fit this tile height");
        for (int s = 0; s < 3 && s < ns; ++s) stage_next();
        if (ns >= 3) {
            if (grp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        bf16x8 wr[2][4], ar[RING];
        auto ld_w = [&](int slab, int i) { return *reinterpret_cast<const bf16x8*>(smem + (slab & 3) * SLAB + w_off + (32 * (i >> 1) + 4 * (i & 1)) * 64); };
        auto ld_a = [&](int slab, int j) { return *reinterpret_cast<const bf16x8*>(smem + (slab & 3) * SLAB + a_off + j * 1024); };
#pragma unroll
        for (int i = 0; i < 4; ++i) wr[0][i] = ld_w(0, i);
#pragma unroll
        for (int j = 0; j < LEAD; ++j) ar[j] = ld_a(0, j);
        __builtin_amdgcn_sched_barrier(0);
        auto slab_body = [&](auto par, int s) {
            constexpr int P = decltype(par)::value;
            char* dst = nullptr;
#pragma unroll
            for (int j = 0; j < MTG; ++j) {
                if (j == BR) {
                    // every wave's pieces of slab s+1 have landed (its pieces of slab s+2 may be in flight); after the barrier slab s+1 is readable by
                    // everyone, and slab s-1 - whose last fragment was read MTG rows ago - gives its slot to slab s+3
                    if (grp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
                    __builtin_amdgcn_s_barrier();
                    piece_sources();
                    dst = smem + (ld_s & 3) * SLAB + (grp == 0 ? wq * (NPA * 1024) : SLAB_A + wq * (NPW * 1024));
                }
                // fragment LEAD rows ahead, into the ring slot the previous row's MFMAs have just released
                ar[(j + LEAD) % RING] = (j + LEAD < MTG) ? ld_a(s, j + LEAD) : ld_a(s + 1, j + LEAD - MTG);
                if (j == BR || j == BR + 1) {          // the next slab's weight fragments, right behind the barrier: three rows or more before their first use
                    wr[P ^ 1][2 * (j - BR)] = ld_w(s + 1, 2 * (j - BR));
                    wr[P ^ 1][2 * (j - BR) + 1] = ld_w(s + 1, 2 * (j - BR) + 1);
                }
                if (j >= BR && j - BR < NPMAX) {
                    const int i = j - BR;
                    if (i < NPMIN || (NPA > NPW ? grp == 0 : grp == 1)) blds16(rsrc, src[i], src_soff, dst + i * 1024);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[P][i], ar[j % RING], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            advance_slab();
        };
        int s = 0;
        for (; s + 1 < ns; s += 2) {
            slab_body(std::integral_constant<int, 0>{}, s);
            slab_body(std::integral_constant<int, 1>{}, s + 1);
        }
        if (s < ns) slab_body(std::integral_constant<int, 0>{}, s);
    } else {
    // ---- prologue: three slabs in flight, slab 0 landed -------------------------------------------------------------------------
    for (int s = 0; s < 3 && s < ns; ++s) stage_next();
    if (ns >= 3) {
        if (grp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPA) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                   // B_0
    if (grp == 1) __builtin_amdgcn_s_barrier();                     // group 1 runs one interval behind (B_1)

#ifdef MI_PP_TRACE
    unsigned* tr = reinterpret_cast<unsigned*>(smem + 4 * SLAB) + grp * PP_TRACE_STEPS * PP_TRACE_PTS;
    const bool tr_on = blockIdx.x == p.korder >> 8 && wq == 0;
    const unsigned long long tr_c0 = __builtin_readcyclecounter(), tr_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#if PP_PRIO == 2
    if (grp == 1) __builtin_amdgcn_s_setprio(1);
#endif
#if defined(PP_DBG) && (PP_DBG & 1)
    read_frags(0);                                                  // timeline experiment: the loop below issues no LDS reads
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    for (int s = 0; s < ns; ++s) {
        // ---- read segment (G0: I_2s, G1: I_2s+1) ----
        PP_T(0)
        PP_T(1)
#if !(defined(PP_DBG) && (PP_DBG & 1))
        read_frags(s);
#endif
        piece_sources();
#if !PP_DMA_SEG && !(defined(PP_DBG) && (PP_DBG & 2))
        {
            char* dst = smem + (ld_s & 3) * SLAB + (grp == 0 ? wq * (NPA * 1024) : SLAB_A + wq * (NPW * 1024));
#pragma unroll
            for (int i = 0; i < NPMAX; ++i)
                if (i < NPMIN || (NPA > NPW ? grp == 0 : grp == 1)) blds16(rsrc, src[i], src_soff, dst + i * 1024);
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        PP_T(2)
        // end of an odd interval: group 1's pieces of slab s+1 must have landed (only slab s+2 may still be in flight: it stages
        // slab s+3 in the MFMA segment below)
#if !(defined(PP_DBG) && (PP_DBG & 2))
        if (grp == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PP_DMA_SEG ? NPW : 2 * NPW) : "memory");
#endif
        __builtin_amdgcn_s_barrier();
        PP_T(3)
        // ---- MFMA segment (G0: I_2s+1, G1: I_2s+2) ----
#if PP_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
        mfma_and_stage();                                           // + slab s+3 -> the slot slab s-1 has left
#if PP_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        PP_T(4)
        if (grp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPA) : "memory");     // slab s+1 landed (s+2, s+3 in flight)
        __builtin_amdgcn_s_barrier();
        PP_T(5)
    }
#ifdef MI_PP_TRACE
    if (tr_on && lane == 0) {
        // loop duration in s_memtime ticks and in s_memrealtime ticks (100 MHz): the clock the loop ran at
        g_pp_clock[grp * 2 + 0] = (unsigned)(__builtin_readcyclecounter() - tr_c0);
        g_pp_clock[grp * 2 + 1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - tr_r0);
        for (int i = 0; i < PP_TRACE_STEPS * PP_TRACE_PTS; ++i) g_pp_trace[grp * PP_TRACE_STEPS * PP_TRACE_PTS + i] = tr[i];
    }
#endif
    if (grp == 0) __builtin_amdgcn_s_barrier();                     // group 0 leaves one interval early: keep the barrier counts equal
    }   // ping-pong loop

    if (EPI < 0 && (p.flags & (1 << 30))) {   // perf experiment: main loop only (keeps the accumulators alive, stores nothing)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MTG; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    bf16x8 pres[MTG][2];
    unsigned pbits[MTG];
    igemm_fetch_epilogue<MTG, EPI>(p, m0, n0, grp, wq, frow, fq, pres, pbits);
    igemm_epilogue<MTG, EPI>(p, acc, m0, n0, grp, wq, frow, fq, pres, pbits);
}

// =====================================================================================================================
// igemm_pw_kernel: the ping-pong main loop for 3x3 convs with the three taps of a kernel row sharing ONE A window.
//
// The 3x3 launches of igemm_pp_kernel are held back by the L2 -> LDS stream (35 us of DMA beside 42 us of MFMA at 256 -> 256,
// issued from the read segments, which then outlast the MFMA segments).  The taps (ky, 0..2) read the SAME input row at
// columns w-d, w, w+d: one LDS window of 320 + 2d pixel rows serves all three, so the A traffic falls to a third
// (68.5 KiB instead of 108 KiB per three phases) and group 0 issues its DMA once per three phases.
// As in the fused-row weight gradient the tile rows are PADDED pixel coordinates q = (b*H + h) * (W + 2d) + wp with d zero
// slots on either side of every image row: the A row of output row q for tap kx is window row (q - q0) + d * (1 + sgn*(kx-1))
// for every q, inside the image the right source pixel, outside a pad slot = the convolution's zero padding - no per-tap
// validity.  Pad rows are computed (2d / W more rows) and not stored; the epilogue maps q back to the pixel.
// K order: ky, 32-channel chunk c, kx.  LDS: two A windows (336 rows x 64 B) + a six-slot ring of W slabs (256 x 64 B), 138 KiB.
// Group 0 stages window (ky, c)+1 during the first phase of window (ky, c) and waits for it (vmcnt(0): it issues nothing else)
// at the end of the window's last MFMA segment, five intervals later; group 1 streams the W slabs four phases ahead.
// sgn = +1 forward (source = out + (tap-1)*d), -1 data gradient (source = out - (tap-1)*d; weights packed [tap][i][o]).
template <int MTG> struct PWGeo {
    static constexpr int BMG = 16 * MTG, BM = 2 * BMG, BN = 256;
    static constexpr int WROWS = 336, WIN = WROWS * 64;             // A window: BM + 2d rows rounded up to whole 16-row pieces (d <= 8)
    static constexpr int NWIN = 2, NWS = 6, SLAB_W = BN * 64;
    static constexpr int W0 = NWIN * WIN, LDS_BYTES = W0 + NWS * SLAB_W;     // 42 KiB + 96 KiB
    static constexpr int PA = WROWS / 16;                           // 21 pieces: wave wq of group 0 moves pieces wq*5 .. wq*5+4, wave 0 also piece 20
    static constexpr int NPW = 4;
    static_assert(BM + 16 <= WROWS, "window too small");
};

#ifdef MI_EXPERIMENTS   // opt-in kernel that does not win: experiment builds only, not part of libmi355seg.so
template <int MTG, int EPI>
__global__ __launch_bounds__(512, 1) void igemm_pw_kernel(IgemmParams p) {
    using G = PWGeo<MTG>;
    constexpr int BM = G::BM, BN = G::BN, BMG = G::BMG, WIN = G::WIN, W0 = G::W0, SLAB_W = G::SLAB_W, NPW = G::NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int nwg = p.m_tiles * p.n_tiles;
    const int tile = mi_xcd_remap(blockIdx.x, nwg);
    const int mt = tile / p.n_tiles, nt = tile - mt * p.n_tiles;
    const int q0 = mt * BM, n0 = nt * BN;
    const int d = p.dil, WP = p.Wa + 2 * d, H = p.Ha, W = p.Wa;
    const int Q = p.M;                                   // padded coordinates: B * H * (W + 2d) (set by the launcher; p.zgw holds the pixel count)
    const int sgn = (p.mode == MI_GATHER_FWD) ? 1 : -1;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const int prow = lane >> 2, pch = lane & 3;
    const int spt = p.Ca >> 5;                           // 32-channel chunks
    const int nwnd = 3 * spt, nph = 9 * spt;

    // ---- group 0: A windows --------------------------------------------------------------------------------------------------
    constexpr int NPA = 6;
    const char* a_base[NPA];       // source of the row for ky = 1, chunk 0 (+ swizzled 16-B chunk)
    unsigned a_ok[NPA];            // bit ky: the row exists for that kernel row
    const int a_np = (grp == 0) ? (wq == 0 ? 6 : 5) : 0;
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int piece = (i < 5) ? wq * 5 + i : 20;
            const int row = piece * 16 + prow;
            const int chunk = (pch ^ ((-(row >> 2)) & 3)) * 16;
            const int qw = q0 - d + row;
            const int qs = qw + WP;                                           // >= 0
            const int bh = qs / WP - 1, wp = qs - (bh + 1) * WP;
            const bool real = (unsigned)qw < (unsigned)Q && (unsigned)(wp - d) < (unsigned)W;
            const int h = bh < 0 ? 0 : bh % H;
            unsigned ok = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) ok |= (unsigned)(real && (unsigned)(h + sgn * (ky - 1) * d) < (unsigned)H) << ky;
            a_ok[i] = (i < a_np) ? ok : 0u;
            a_base[i] = reinterpret_cast<const char*>(p.A + ((long)(real ? bh : 0) * W + (real ? wp - d : 0)) * p.Ca) + chunk;
        }
    }
    int lw = 0, lw_ky = 0, lw_c = 0;                     // next window to stage
    const long ky_bytes = (long)sgn * d * W * p.Ca * 2;  // one kernel row up / down
    auto stage_window = [&]() {
        char* dst = smem + (lw & 1) * WIN;
        const long off = (long)(lw_ky - 1) * ky_bytes + (long)lw_c * 64;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            if (i < a_np) {
                const int piece = (i < 5) ? wq * 5 + i : 20;
                glds16(((a_ok[i] >> lw_ky) & 1u) ? a_base[i] + off : zero, dst + piece * 1024);
            }
        }
        ++lw;
        if (++lw_c == spt) {
            lw_c = 0;
            ++lw_ky;
        }
    };
    // ---- group 1: W slabs -------------------------------------------------------------------------------------------------------
    const char* w_base[NPW];
    bool w_ok[NPW];
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int rl = (wq * NPW + i) * 16 + prow;
            const int n = n0 + rl;
            const int chunk = (pch ^ (((rl >> 3) & 1) << 1)) * 16;
            w_ok[i] = n < p.N;
            w_base[i] = reinterpret_cast<const char*>(p.Wp + (long)(w_ok[i] ? n : 0) * p.Ca) + chunk;
        }
    }
    const long w_tap_bytes = (long)p.N * p.Ca * 2;
    int ls = 0, ls_ky = 0, ls_c = 0, ls_kx = 0;          // next W slab to stage (phase order: ky, c, kx)
    auto stage_w = [&]() {
        char* dst = smem + W0 + (ls % G::NWS) * SLAB_W + wq * (NPW * 1024);
        const long off = (long)(ls_ky * 3 + ls_kx) * w_tap_bytes + (long)ls_c * 64;
#pragma unroll
        for (int i = 0; i < NPW; ++i) glds16(w_ok[i] ? w_base[i] + off : zero, dst + i * 1024);
        ++ls;
        if (++ls_kx == 3) {
            ls_kx = 0;
            if (++ls_c == spt) {
                ls_c = 0;
                ++ls_ky;
            }
        }
    };

    // ---- compute roles ---------------------------------------------------------------------------------------------------------
    f32x4 acc[4][MTG];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MTG; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int rb = grp * BMG + frow + d * (1 + sgn * (kx - 1));          // window row of this lane's row 0 for tap kx
        a_off[kx] = rb * 64 + ((fq ^ ((-(rb >> 2)) & 3)) << 4);               // (rb + 16 j) >> 2 has the same low two bits
    }
    const int w_off = (wq * 64 + 8 * (frow >> 2) + (frow & 3)) * 64 + ((fq ^ (((frow >> 2) & 1) << 1)) << 4);
    bf16x8 wf[4], af[MTG];
    auto read_frags = [&](int wnd, int kx, int ph) {
        const char* wb = smem + W0 + (ph % G::NWS) * SLAB_W + w_off;
        const char* ab = smem + (wnd & 1) * WIN + (kx == 0 ? a_off[0] : (kx == 1 ? a_off[1] : a_off[2]));
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wb + (32 * (i >> 1) + 4 * (i & 1)) * 64);
#pragma unroll
        for (int j = 0; j < MTG; ++j) af[j] = *reinterpret_cast<const bf16x8*>(ab + j * 1024);
    };
    auto mfma_all = [&]() {
#pragma unroll
        for (int j = 0; j < MTG; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: window 0 and W slabs 0..3 in flight; window 0 and slab 0 landed ------------------------------------------------
    if (grp == 0) {
        stage_window();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int s = 0; s < 4 && s < nph; ++s) stage_w();
        if (nph >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                   // B_0
    if (grp == 1) __builtin_amdgcn_s_barrier();                     // group 1 runs one interval behind

    int wnd = 0, kx = 0;
    for (int ph = 0; ph < nph; ++ph) {
        // ---- read segment (G0: I_2ph, G1: I_2ph+1) ----
        if (grp == 0) {
            if (kx == 0 && wnd + 1 < nwnd) stage_window();          // window wnd+1 -> the slot window wnd-1 has left (last read in I_2ph-1)
        } else {
            if (ph + 4 < nph) stage_w();                            // slab ph+4 -> the slot of slab ph-2
        }
        read_frags(wnd, kx, ph);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) {                                             // end of an odd interval: my pieces of slab ph+1 must have landed
            if (ph + 4 < nph) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        // ---- MFMA segment (G0: I_2ph+1, G1: I_2ph+2) ----
        __builtin_amdgcn_s_setprio(1);
        mfma_all();
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0 && kx == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // window wnd+1 (issued five intervals ago)
        __builtin_amdgcn_s_barrier();
        if (++kx == 3) {
            kx = 0;
            ++wnd;
        }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue: padded coordinate of each of the lane's rows -> pixel (or -1: pad slot / beyond the last image) ------------------
    int mrow[MTG];
    {
        const int qb = q0 + grp * BMG + frow;
        int bh = qb / WP, wp = qb - bh * WP;
#pragma unroll
        for (int j = 0; j < MTG; ++j) {
            const bool real = (qb + 16 * j) < Q && (unsigned)(wp - d) < (unsigned)W;
            mrow[j] = real ? bh * W + wp - d : -1;
            wp += 16;                                    // WP >= 16 (launcher): at most one carry
            if (wp >= WP) {
                wp -= WP;
                ++bh;
            }
        }
    }
    IgemmParams pe = p;
    pe.M = p.zgw;                                        // the epilogue's bound is the pixel count
    bf16x8 pres[MTG][2];
    unsigned pbits[MTG];
    igemm_fetch_epilogue<MTG, EPI>(pe, 0, n0, grp, wq, frow, fq, pres, pbits, mrow);
    igemm_epilogue<MTG, EPI>(pe, acc, 0, n0, grp, wq, frow, fq, pres, pbits, nullptr, false, mrow);
}

template <int EPI>
void launch_pw(dim3 grid, hipStream_t stream, const IgemmParams& p) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = igemm_pw_kernel<10, EPI>;
    constexpr int lds = PWGeo<10>::LDS_BYTES;
    mi_allow_dynamic_lds((const void*)kern, lds, attr_done);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);
}
#endif  // MI_EXPERIMENTS

template <int MTG, int EPI, bool RR>
void launch_pp(dim3 grid, hipStream_t stream, const IgemmParams& p) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = igemm_pp_kernel<MTG, EPI, RR>;
#ifdef MI_PP_TRACE
    constexpr int lds = PPGeo<MTG>::LDS_BYTES + 2 * PP_TRACE_STEPS * PP_TRACE_PTS * 4;
#else
    constexpr int lds = PPGeo<MTG>::LDS_BYTES;
#endif
    mi_allow_dynamic_lds((const void*)kern, lds, attr_done);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);
}

template <int MTG, bool RR>
void launch_pp_flags(dim3 grid, hipStream_t stream, const IgemmParams& p) {
    if (p.flags == 69) return launch_pp<MTG, 69, RR>(grid, stream, p);       // FrozenBN + ReLU + sign bits (conv forward)
    if (p.flags == 128) return launch_pp<MTG, 128, RR>(grid, stream, p);     // ReLU backward from sign bits (data gradient)
    if (p.flags == 0) return launch_pp<MTG, 0, RR>(grid, stream, p);         // plain bf16 store (downsample data gradient, ASPP data gradient)
    if (p.flags == 1) return launch_pp<MTG, 1, RR>(grid, stream, p);         // FrozenBN only (downsample forward)
    if (p.flags == 48) return launch_pp<MTG, 48, RR>(grid, stream, p);       // fp32 tap planes (ASPP forward)
    if (p.flags == MI_EPI_STATS) return launch_pp<MTG, MI_EPI_STATS, RR>(grid, stream, p);      // plain store + BatchNorm tile statistics
    launch_pp<MTG, -1, RR>(grid, stream, p);
}

}  // namespace

#ifdef MI_PP_TRACE
extern "C" int mi_pp_trace_read(unsigned* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_trace), sizeof(unsigned) * n);
}
extern "C" int mi_pp_clock_read(unsigned* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_clock), sizeof(unsigned) * 4); }
#endif

// Same contract as mi_conv_gemm, restricted to stride 1 / Ha == Ho / Ca % 32 == 0.  mi_conv_gemm dispatches here by its cost
// model; exported so that the two main loops can be compared in one process (tools/kexp.py).  mtg = 0: choose the tile height.
extern "C" int mi_conv_gemm_pp(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize,
                               int stride, int pad, int dil, int gather_mode, const float* scale, const float* bias, const void* res,
                               const void* msk, void* mask_out, int flags, int zgw, float alpha, int mtg, void* stream) {
    return mi_conv_gemm_pp_impl(a, wp, out, B, Ha, Wa, Ca, Ho, Wo, N, ksize, stride, pad, dil, gather_mode, scale, bias, res, msk, mask_out, flags, zgw, alpha,
                                mtg, stream, nullptr);
}

int mi_conv_gemm_pp_impl(const void* a, const void* wp, void* out, int B, int Ha, int Wa, int Ca, int Ho, int Wo, int N, int ksize, int stride, int pad,
                         int dil, int gather_mode, const float* scale, const float* bias, const void* res, const void* msk, void* mask_out, int flags,
                         int zgw, float alpha, int mtg, void* stream, MiConvStats* st) {
    MI_REQUIRE(a && wp && out, "mi_conv_gemm_pp: null operand");
    MI_REQUIRE(((flags & MI_EPI_STATS) != 0) == (st != nullptr) && (!st || flags == MI_EPI_STATS), "mi_conv_gemm_pp: MI_EPI_STATS comes alone, through mi_conv_gemm_stats");
    MI_REQUIRE(B > 0 && Ha > 0 && Wa > 0 && N > 0, "mi_conv_gemm_pp: non-positive dimension");
    MI_REQUIRE(stride == 1 && Ha == Ho && Wa == Wo, "mi_conv_gemm_pp: stride-1, same-size convolutions only");
    MI_REQUIRE(Ca > 0 && Ca % 32 == 0, "mi_conv_gemm_pp: Ca=%d must be a multiple of 32", Ca);
    MI_REQUIRE(N % 8 == 0, "mi_conv_gemm_pp: N=%d must be a multiple of 8", N);
    MI_REQUIRE(ksize == 1 || ksize == 3, "mi_conv_gemm_pp: ksize=%d (1 or 3)", ksize);
    MI_REQUIRE(dil >= 1 && pad >= 0, "mi_conv_gemm_pp: bad dil/pad");
    MI_REQUIRE(gather_mode == MI_GATHER_FWD || gather_mode == MI_GATHER_DGRAD, "mi_conv_gemm_pp: gather_mode");
    MI_REQUIRE(mi_aligned16(a) && mi_aligned16(wp) && mi_aligned16(out), "mi_conv_gemm_pp: operands must be 16-byte aligned");
    MI_REQUIRE(!(flags & MI_EPI_SCALE_BIAS) || (scale && bias && mi_aligned16(scale) && mi_aligned16(bias)), "mi_conv_gemm_pp: scale/bias");
    MI_REQUIRE(!(flags & MI_EPI_RESIDUAL) || (res && mi_aligned16(res)), "mi_conv_gemm_pp: residual");
    MI_REQUIRE(!(flags & MI_EPI_MASK) || (msk && mi_aligned16(msk)), "mi_conv_gemm_pp: mask");
    MI_REQUIRE(!(flags & MI_EPI_BITMASK) || (msk && N % 16 == 0 && !(flags & MI_EPI_MASK)), "mi_conv_gemm_pp: bit mask needs N %% 16 == 0");
    MI_REQUIRE(!(flags & MI_EPI_WRITE_MASK) || (mask_out && N % 16 == 0), "mi_conv_gemm_pp: mask_out needs N %% 16 == 0");
    MI_REQUIRE(!(flags & MI_EPI_ZSPLIT) || (zgw > 0 && zgw % 4 == 0 && N % zgw == 0), "mi_conv_gemm_pp: zgw");
    const long M = (long)B * Ho * Wo;
    MI_REQUIRE(M < (1L << 31), "mi_conv_gemm_pp: pixel count overflows int32");
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
    p.stride = 1;
    p.pad = pad;
    p.dil = dil;
    p.mode = gather_mode;
    p.flags = flags;
    p.zgw = zgw > 0 ? zgw : 4;
    p.alpha = alpha;
    p.stats = st ? st->partial : nullptr;
    p.pilot = st ? st->pilot : nullptr;
    p.korder = ksize > 1 ? mi_sw().pp_korder : 0;              // MI_IGEMM_PP_KORDER=0: tap-major contraction (the order of igemm_nt_kernel; bit-equal to it)
#ifdef MI_PP_TRACE
    p.korder |= mi_sw().pp_trace_wg << 8;
#endif
    // 3x3 with the hot epilogues: the shared-window kernel (mtg == 0 only: an explicit 8 / 10 selects igemm_pp_kernel).
    // MI_IGEMM_PW: 0 = off (default: measured 90 vs 85 us at 3x3 256, 311 vs 290 at 3x3 512 - see DESIGN.md section 8), 1 = by rule,
    // 2 = always when the geometry allows (tests on tiny shapes)
#ifdef MI_EXPERIMENTS
    const int pw_mode = mi_sw().igemm_pw;
    const int WPad = Wa + 2 * dil;
    const long Qp = (long)B * Ha * WPad;
    if ((mtg == 3 || (mtg == 0 && pw_mode)) && ksize == 3 && pad == dil && dil <= 8 && WPad >= 16 && (flags == 69 || flags == 128) && Qp < (1L << 31) &&
        (pw_mode == 2 || mtg == 3 || M >= 320 * 64)) {
        p.M = (int)Qp;                     // the kernel's rows are padded coordinates; the pixel count travels in zgw (unused by these epilogues)
        p.zgw = (int)M;
        p.m_tiles = (int)((Qp + 319) / 320);
        p.n_tiles = (N + 255) / 256;
        const dim3 gridw(p.m_tiles * p.n_tiles);
        if (flags == 69) launch_pw<69>(gridw, (hipStream_t)stream, p);
        else launch_pw<128>(gridw, (hipStream_t)stream, p);
        MI_CHECK_LAUNCH("mi_conv_gemm_pp (shared window)");
        return MI_OK;
    }
#endif
    if (mtg == 3) return mi_set_error(MI_EINVAL, "mi_conv_gemm_pp: mtg 3 (shared-window kernel, experiment builds only) needs a 3x3 conv with pad == dil <= 8 and flags 69 or 128");
    {   // 32-bit buffer offsets (bit 31 marks a padded chunk): operands and the largest tap excursion must stay below 2 GiB
        const long a_bytes = (long)B * Ha * Wa * Ca * 2 + 2L * ((long)(ksize - 1) * dil + pad) * (Wa + 1) * Ca * 2;
        const long w_bytes = (long)ksize * ksize * N * Ca * 2;
        MI_REQUIRE(a_bytes < (1L << 31) - (1L << 20) && w_bytes < (1L << 31) - (1L << 20), "mi_conv_gemm_pp: operand larger than 2 GiB");
    }
    bool rolling = mi_sw().pp_loop != 0;   // MI_IGEMM_PP_LOOP (read in experiment builds only); an explicit mtg of 108 / 110 (8 / 10) selects the rolling (ping-pong) loop
    if (mtg == 108 || mtg == 110) rolling = true, mtg -= 100;
    else if (mtg == 8 || mtg == 10) rolling = false;
    if (mtg != 8 && mtg != 10) {           // fewest rounds on 256 CUs, then the least padding
        auto rounds = [&](int bm) { return (((M + bm - 1) / bm) * ((N + 255) / 256) + 255) / 256; };
        mtg = rounds(320) < rounds(256) ? 10 : (rounds(256) < rounds(320) ? 8 : (((M + 319) / 320) * 320 <= ((M + 255) / 256) * 256 ? 10 : 8));
    }
    const int bm = 32 * mtg;
    p.m_tiles = (int)((M + bm - 1) / bm);
    p.n_tiles = (N + 255) / 256;
    const dim3 grid(p.m_tiles * p.n_tiles);
    if (st) st->nparts = 2 * p.m_tiles;                    // one partial row per wave group (MTG*16 rows)
#ifdef MI_EXPERIMENTS      // the rolling loop does not win (header): experiment builds only (tools/experiments/build.sh, tools/rrexp.py)
    if (rolling) {
        if (mtg == 10) launch_pp_flags<10, true>(grid, (hipStream_t)stream, p);
        else launch_pp_flags<8, true>(grid, (hipStream_t)stream, p);
        MI_CHECK_LAUNCH("mi_conv_gemm_pp (rolling loop)");
        return MI_OK;
    }
#else
    if (rolling) return mi_set_error(MI_EINVAL, "mi_conv_gemm_pp: the rolling main loop (mtg 108 / 110, MI_IGEMM_PP_LOOP=1) exists in experiment builds only");
#endif
    if (mtg == 10) launch_pp_flags<10, false>(grid, (hipStream_t)stream, p);
    else launch_pp_flags<8, false>(grid, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_gemm_pp");
    return MI_OK;
}
