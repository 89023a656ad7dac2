// Two chained 1x1 convolutions of the bottleneck sequence in ONE kernel for gfx950: the wide tensor between them is written once and never re-read.
//
//   forward  (reference core/components/resnet.py:105-113 of block i, then :93-95 of block i+1):
//       x1 = relu(scale1 * (a2 . W3^T) + shift1 + x0)         conv3 + FrozenBN + residual + ReLU      [M][256] -> [M][1024]
//       a1 = relu(scale2 * (x1 . W1^T) + shift2)              conv1 + FrozenBN + ReLU of the next block [M][1024] -> [M][256]
//   backward (the data gradients of the same two convs, in the order backward meets them: conv1 of block i, then conv3 of block i-1):
//       gx = ((ga1 . W1t^T) + g) masked by the sign bits of x     [M][256] -> [M][1024]   (FrozenBN scale folded into the packed weights)
//       ga2 = (gx . W3t^T) masked by the sign bits of a2          [M][1024] -> [M][256]
// As two launches (igemm_nt with the residual epilogue, then igemm_pp) the [M][1024] tensor (154 MB at B = 8, 769 x 769) is written by the first and read
// back by the second; both launches are bound by those bytes, not by the matrix pipe.  Here a wave OWNS its 16 pixel rows through
// both products: the first product's accumulator tile, after its epilogue, IS the second product's B operand (cdna_hip_programming.md section 3, "An
// accumulator tile as the next MFMA's operand") - no LDS round trip, no barrier between the two products.
//
// MFMA orientation as in igemm_nt.hip: D rows = output channels (weights are the A operand), D columns = pixels.  With the weight-row permutation of
// that file a lane (pixel = lane & 15, q = lane >> 4) holds, of every 64-channel chunk of x1, channels 8q .. 8q+7 and 32+8q .. 32+8q+7 - exactly the
// elements the B operand of v_mfma_f32_16x16x32_bf16 wants from that lane for k-steps 0 and 1 of the chunk (k = 8q + j), in natural k order: the
// second product reads the ordinary packed weights [N2][1024].
//
// Structure: one persistent workgroup of 8 waves per CU over a contiguous range of 16-row tiles; a pass covers 8 tiles (one per wave).  Per pass both
// weight matrices (2 x 512 KB) stream through a 5-slot LDS ring of 16-KB blocks shared by the waves (buffer_load ... lds, counted s_waitcnt vmcnt).  x1 is
// produced in sixteen 64-channel chunks, four ring blocks each:
//   A0 / A1  first product, output channels 0-31 / 32-63 of the chunk (MFMA tiles 0,1 / 2,3), all of K1 = 256: 16 MFMAs per wave each;
//   B0 / B1  second product, k-step 0 / 1 of the chunk (the channels of A0 / A1), all 256 outputs: 16 MFMAs per wave each.
// After A0 the lane has the final values of half of the chunk: its epilogue half (residual from a wave-private LDS ring filled by LDS-DMA two chunks ahead,
// FrozenBN / ReLU / sign bits or the backward mask, bf16 in place) runs right there and yields B0's operand; after A1 the other half, then the chunk leaves
// for HBM with row-contiguous lanes (8 rows x 128 B per store instruction).
// The first version ran one wave per SIMD (4 waves, 32 rows each): correct, and exactly as fast as the two launches - s_memtime stamps showed the wave's
// MFMAs (2.0k cycles per chunk), its DMA / fragment-read issue (2.0k) and its epilogue arithmetic (2.1k) strictly one after the other.  Now the 8 waves
// are two groups of four (one wave of each per SIMD) that run the SAME program one interval apart, as in igemm_pp.hip: every block has a READ phase (issue
// the DMA pieces of the block four ahead, 16 ds_read_b128 of this block's fragments) and an MFMA phase (16 MFMAs, epilogue halves); while group 0 is in
// READ(x), group 1 is in MFMA(x-1), and vice versa, a raw s_barrier between intervals.  One fragment set of 16 per wave; ~230 registers.
//   I_2x: G0 READ(x) | G1 MFMA(x-1)        I_2x+1: G0 MFMA(x) | G1 READ(x)
//   block x+1 must have landed before I_2x+2, where group 0 reads it: group 0 waits for ITS pieces at the end of MFMA(x), group 1 at the end of READ(x);
//   block x+4 lands in the slot of block x-1, last read by group 1 in I_2x-1 (reads retired by lgkmcnt(0) before the barrier).
// Every vector-memory operation is a builtin the compiler can see and their order per chunk is fixed, so the waits are compile-time counts
// (tools/dbg/chain_vmcnt.py replays the order and prints them).
#include <type_traits>
#include "igemm_common.h"

namespace {

struct ChainParams {
    const __bf16* A;            // [M][K1]
    const __bf16* Wa;           // first product's weights  [N1][K1]
    const __bf16* Wb;           // second product's weights [N2][N1]
    const __bf16* res;          // [M][N1]
    __bf16* mid;                // [M][N1]  (x1 / gx)
    __bf16* out;                // [M][N2]
    const float* scale1;        // forward: FrozenBN of the first conv [N1]
    const float* shift1;
    const float* scale2;        // forward: FrozenBN of the second conv [N2]
    const float* shift2;
    const uint8_t* bits1_in;    // backward: sign bits that mask mid  [M][N1/8]
    const uint8_t* bits2_in;    // backward: sign bits that mask out  [M][N2/8]
    uint8_t* bits1_out;         // forward: sign bits of mid
    uint8_t* bits2_out;         // forward: sign bits of out
    int M;
};

#ifndef CHAIN_DBG
#define CHAIN_DBG 0            // measurement builds only (tools/dbg/chain_variants.sh): 1 no MFMAs, 2 no residual DMA / mid stores, 4 no epilogue arithmetic, 8 no fragment reads, 16 weight DMA fetches nothing, 32 no DMA instructions at all, 64 no barriers
#endif
#ifndef CHAIN_ROT
#define CHAIN_ROT 0            // chunk rotation per workgroup: first chunk = (blockIdx * CHAIN_ROT) mod 16; 0 = every workgroup starts at chunk 0 (measured: no difference)
#endif
#ifndef CHAIN_MID_AUX
#define CHAIN_MID_AUX 2        // cache policy of the wide tensor's stores: 2 = non-temporal (as igemm_store_staged), 0 = default
#endif

struct ChainGeo {
    static constexpr int K1 = 256, N1 = 1024, N2 = 256, NC = 64, NCH = N1 / NC, NW = 8;
    static constexpr int WBLK = 16384, NB = 5;                           // weight ring: blocks of 16 KB, four in flight beside the one being read
    static constexpr int PW = 16 / NW;                                   // DMA pieces (1 KB) per wave and block
    static constexpr int RES_SLOT = 2048, RES_DEPTH = 3;                 // per wave: 16 rows x 128 B, chunks c, c+1, c+2
    static constexpr int OFF_RES = NB * WBLK;
    static constexpr int OFF_SS = OFF_RES + NW * RES_DEPTH * RES_SLOT;   // scale1 | shift1 | scale2 | shift2 (fp32)
    static constexpr int SS_BYTES = (2 * N1 + 2 * N2) * 4;
    static constexpr int OFF_BITS1 = OFF_SS + SS_BYTES;                  // per wave: 16 rows x 128 B of sign bits of mid
    static constexpr int BITS1_WAVE = 16 * (N1 / 8);
    static constexpr int LDS_BYTES = OFF_BITS1 + NW * BITS1_WAVE;        // 154 KB
    // vm operations younger than the awaited ones at each wait (tools/dbg/chain_vmcnt.py 2 2 2; steady state, the first chunks have more in flight)
    static constexpr int WAIT_W = 10, WAIT_R0 = 6, WAIT_R1 = 16, WAIT_R = 24;
    static_assert(LDS_BYTES <= MI_LDS_MAX, "LDS budget");
};

#ifdef CHAIN_TRACE
// Timeline experiment (tools/chaintrace.py builds a second library with -DCHAIN_TRACE=<pass>; never defined in the product build): lane 0 of waves 0 and 4 of
// ONE workgroup stamps s_memtime at sixteen points of every chunk of that pass into the LDS the kernel leaves free, dumped at the end of the pass.
__device__ unsigned g_chain_trace[2 * 16 * 16 + 4];
#define CT(k)                                                                                                          \
    if (tr_on) {                                                                                                       \
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();                                                    \
        if (lane == 0) reinterpret_cast<unsigned*>(smem + G::LDS_BYTES)[grp * 256 + ci * 16 + (k)] = t_;               \
    }
#else
#define CT(k)
#endif

// s_waitcnt as the builtin, not inline asm: the compiler's own wait insertion then knows what has already been waited for (with asm waits it put
// s_waitcnt lgkmcnt(14) in front of MFMAs whose fragments the asm lgkmcnt(0) had long retired).  gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4],
// lgkmcnt [11:8]; the fields not meant are left at their maxima.
#define CHAIN_VMCNT(n) (((n) & 15) | (((n) >> 4) << 14) | 0x70 | 0xF00)
template <int N> __device__ __forceinline__ void chain_wait_vm() { __builtin_amdgcn_s_waitcnt(CHAIN_VMCNT(N)); }
__device__ __forceinline__ void chain_wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }     // lgkmcnt(0)
__device__ __forceinline__ void chain_bar() {
    __builtin_amdgcn_sched_barrier(0);
    if (!(CHAIN_DBG & 64)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <bool BWD>
__global__ __launch_bounds__(512, 2) void chain_kernel(ChainParams p) {
    using G = ChainGeo;
    constexpr int K1 = G::K1, N1 = G::N1, N2 = G::N2, NCH = G::NCH, NB = G::NB, WBLK = G::WBLK, NW = G::NW, PW = G::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int frow = lane & 15, q = lane >> 4;

    // ---- this workgroup's 16-row tiles: a contiguous range, as even as 16-row granules allow ----------------------------------
    const int mt_total = (p.M + 15) >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int per = mt_total / nwg, rem = mt_total - per * nwg;
    const int t_lo = bid * per + (bid < rem ? bid : rem), t_hi = t_lo + per + (bid < rem ? 1 : 0);
    const int npass = (t_hi - t_lo + NW - 1) / NW;
    if (npass == 0) return;
    const int row_hi = (t_hi * 16 < p.M) ? t_hi * 16 : p.M;            // first row that is not this workgroup's
    const int total_blocks = npass * NCH * 4;
    const int rot = CHAIN_ROT ? (int)((bid * CHAIN_ROT) & (NCH - 1)) : 0;

    if (!BWD) {                                                        // FrozenBN coefficients -> LDS (read per chunk with ds_read_b128 broadcasts)
        float* ss = reinterpret_cast<float*>(smem + G::OFF_SS);
        for (int i = tid; i < N1; i += 512) ss[i] = p.scale1[i], ss[N1 + i] = p.shift1[i];
        for (int i = tid; i < N2; i += 512) ss[2 * N1 + i] = p.scale2[i], ss[2 * N1 + N2 + i] = p.shift2[i];
        __syncthreads();
    }

    const __amdgpu_buffer_rsrc_t rWa = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Wa), 0, N1 * K1 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rWb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Wb), 0, N2 * N1 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.A), 0, p.M * (K1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.res), 0, p.M * (N1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rMid = __builtin_amdgcn_make_buffer_rsrc(p.mid, 0, p.M * (N1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.M * (N2 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB1 = __builtin_amdgcn_make_buffer_rsrc(BWD ? const_cast<uint8_t*>(p.bits1_in) : p.bits1_out, 0, p.M * (N1 / 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(BWD ? const_cast<uint8_t*>(p.bits2_in) : p.bits2_out, 0, p.M * (N2 / 8), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;        // an offset past every resource: loads return zeros, stores are dropped

    // ---- weight ring: per-lane DMA sources (constant over the kernel) and fragment read addresses ---------------------------------
    // A block: 32 rows (two MFMA tiles of output channels) x 512 B (K1); 16-B slot u of row r sits at u ^ fa(r).  DMA piece = 2 rows.
    // B block: 256 rows (N2) x 64 B (one 32-wide k-step); slot u of row r at u ^ fb(r).  DMA piece = 16 rows.
    // Both brute-forced so that every ds_read_b128 lane group of the permuted-row fragment reads hits 16 distinct slots (tools/dbg/chain_swizzle.py).
    auto fa = [](int r) { return (r & 15) ^ ((r & 1) << 2); };
    auto fb = [](int r) { return (r >> 2) & 3; };
    unsigned wsa[PW], wsb[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int pc = PW * wave + i;
        const int ra = 2 * pc + (lane >> 5), rb = 16 * pc + (lane >> 2);
        wsa[i] = (unsigned)(ra * (K1 * 2) + 16 * ((lane & 31) ^ fa(ra)));
        wsb[i] = (unsigned)(rb * (N1 * 2) + 16 * ((lane & 3) ^ fb(rb)));
    }
    // fragment addresses inside a block.  A: tile t (0, 1), k-step ks: row r = 4 t + 8 (frow >> 2) + (frow & 3), slot 4 ks + q -> base + 64 (ks ^ hi);
    // B: tile t (0 .. 15): row 64 (t >> 2) + 32 ((t >> 1) & 1) + 4 (t & 1) + 8 (frow >> 2) + (frow & 3), slot q (fb depends on t & 1 only)
    int fra_base[2], fra_hi[2], frb_base[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = 4 * t + 8 * (frow >> 2) + (frow & 3);
        fra_base[t] = r * 512 + 16 * (q ^ (fa(r) & 3));
        fra_hi[t] = fa(r) >> 2;
        frb_base[t] = r * 64 + 16 * (q ^ fb(r));
    }
    int slot_wr = 0, slot_rd = 0;      // ring slots of the next block to issue / to read
    int bk_issue = 0;                  // blocks issued so far
    // issue one block: TYPE 0 / 1 = the two halves of the chunk's output channels of the first product, 2 / 3 = the two k-steps of the second product
    auto issue_w = [&](auto type_c) {
        constexpr int TYPE = decltype(type_c)::value;
        const int c = ((bk_issue >> 2) + rot) & (NCH - 1);
        const unsigned kill = (bk_issue < total_blocks && !(CHAIN_DBG & 16)) ? 0u : OOB;             // past the end: the same instruction count, no traffic
        char* dst = smem + slot_wr * WBLK + wave * (PW * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            if (CHAIN_DBG & 32) continue;
            if (TYPE < 2) blds16(rWa, wsa[i] | kill, (unsigned)((c * 64 + TYPE * 32) * (K1 * 2)), dst + i * 1024);
            else blds16(rWb, wsb[i] | kill, (unsigned)(c * 128 + (TYPE - 2) * 64), dst + i * 1024);
        }
        ++bk_issue;
        slot_wr = slot_wr == NB - 1 ? 0 : slot_wr + 1;
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;

    // Fragments: two sets of eight (half a block each).  While the MFMAs of one half run, the ds_read_b128 of the next half are in flight.
    // A block halves: k-steps 0-3 / 4-7 of both tiles (index 2 (ks & 3) + t); B block halves: tiles 0-7 / 8-15.
    bf16x8 P[8], Q[8];
    bool rd_on = true;                 // false for a wave without rows in this pass: it keeps its DMA share and the barriers, not the LDS reads
    auto read_a = [&](bf16x8 (&w)[8], auto half_c, bool advance) {
        constexpr int HALF = decltype(half_c)::value;
        const char* blk = smem + slot_rd * WBLK;
        if (advance) slot_rd = slot_rd == NB - 1 ? 0 : slot_rd + 1;
        if (!rd_on) return;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const char* a = blk + fra_base[t] + (((4 * HALF + k) ^ fra_hi[t]) << 6);
                if (CHAIN_DBG & 8) asm volatile("" : "=v"(w[2 * k + t]) : "v"(a));
                else w[2 * k + t] = *reinterpret_cast<const bf16x8*>(a);
            }
    };
    auto read_b = [&](bf16x8 (&w)[8], auto half_c, bool advance) {
        constexpr int HALF = decltype(half_c)::value;
        const char* blk = smem + slot_rd * WBLK;
        if (advance) slot_rd = slot_rd == NB - 1 ? 0 : slot_rd + 1;
        if (!rd_on) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = 8 * HALF + k;
            const char* a = blk + frb_base[t & 1] + (64 * (t >> 2) + 32 * ((t >> 1) & 1)) * 64;
            if (CHAIN_DBG & 8) asm volatile("" : "=v"(w[k]) : "v"(a));
            else w[k] = *reinterpret_cast<const bf16x8*>(a);
        }
    };

    // ---- wave-private rows: one 16-row tile per pass ----------------------------------------------------------------------------------
    // Row-contiguous pieces (8 rows x 128 B of a 64-channel chunk): lane -> row 8 i + (lane >> 3), 16-B slot (lane & 7) ^ (row & 7) (residual DMA
    // source and mid store target; the LDS image is lane-linear).
    char* const res_base = smem + G::OFF_RES + wave * (G::RES_DEPTH * G::RES_SLOT);
    auto row_offsets = [&](int pass, unsigned (&off)[2], unsigned rowbytes, unsigned lane_bytes) {
        const int m0 = (t_lo + pass * NW + wave) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + 8 * i + (lane >> 3);
            off[i] = (pass < npass && m < row_hi) ? (unsigned)m * rowbytes + lane_bytes : OOB;
        }
    };
    const unsigned piece_lane = 16u * (unsigned)((lane & 7) ^ ((lane >> 3) & 7));
    unsigned ro_cur[2], ro_nxt[2];                     // mid / res row offsets of this pass and of the next
    row_offsets(0, ro_cur, N1 * 2, piece_lane);
    row_offsets(1, ro_nxt, N1 * 2, piece_lane);
    int gc = 0;                                        // chunks done (over all passes); residual ring slot = chunk % 3
    auto issue_res = [&](int gcp, const unsigned (&off)[2]) {        // residual of global chunk gcp
        char* dst = res_base + (gcp % G::RES_DEPTH) * G::RES_SLOT;
        const unsigned so = (unsigned)(((gcp + rot) & (NCH - 1)) * 128);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (CHAIN_DBG & 32) continue;
            blds16(rRes, (CHAIN_DBG & 2) ? OOB : off[i], so, dst + i * 1024);
        }
    };

    f32x4 acc2[16];
    bf16x8 af[8];
    const int rsw = frow & 7;
    const int re0 = frow * 128 + 16 * (q ^ rsw), re1 = frow * 128 + 16 * ((4 + q) ^ rsw);     // this lane's two 16-B slots of its row
    char* const bits1 = smem + G::OFF_BITS1 + wave * G::BITS1_WAVE;

    // ---- prologue: five weight blocks and two residual chunks in flight; the first half of block 0 on its way to registers ---------------
    issue_w(T0{});
    issue_w(T1{});
    issue_w(T2{});
    issue_w(T3{});
    issue_w(T0{});
    issue_res(0, ro_cur);
    issue_res(1, ro_cur);
    chain_wait_vm<4 * PW + 4>();       // my pieces of block 0
    chain_bar();
    read_a(P, T0{}, false);

    for (int pass = 0; pass < npass; ++pass) {
        const int tile = t_lo + pass * NW + wave;
        const int m0 = tile * 16;
        const bool active = tile < t_hi;               // wave-uniform: a wave without rows in the last pass keeps its DMA share and the barriers only
        rd_on = active;
#ifdef CHAIN_TRACE
        const bool tr_on = bid == 37 && (wave & 3) == 0 && pass == CHAIN_TRACE;
        const unsigned long long tr_c0 = __builtin_readcyclecounter(), tr_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        // the a2 rows of this wave as B fragments: lane (pixel frow, q) holds k = 32 s + 8 q .. + 7 of its pixel for the 8 k-steps
        {
            const int m = m0 + frow;
            const unsigned vo = m < row_hi ? (unsigned)m * (K1 * 2) + 16u * q : OOB;
#pragma unroll
            for (int s = 0; s < 8; ++s) af[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rA, vo, 64 * s, 0));
        }
        unsigned bo_cur[2];                            // sign bits of mid: 8 rows x 128 B pieces, lane -> row 8 i + (lane >> 3), bytes 16 (lane & 7)
        row_offsets(pass, bo_cur, N1 / 8, 16u * (lane & 7));
        if (BWD) {
#pragma unroll
            for (int i = 0; i < 2; ++i) blds16(rB1, bo_cur[i], 0, bits1 + i * 1024);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ci = 0; ci < NCH; ++ci) {
            const int c = (ci + rot) & (NCH - 1);          // the chunk of mid this iteration produces and consumes
            f32x4 acc1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 xf[2];
            char* const slot = res_base + (gc % G::RES_DEPTH) * G::RES_SLOT;
            auto mma_a = [&](const bf16x8 (&w)[8], auto h_c, auto half_c) {
                constexpr int H = decltype(h_c)::value, HALF = decltype(half_c)::value;
                if (active) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            if (CHAIN_DBG & 1) asm volatile("" : "+v"(acc1[2 * H + t]) : "v"(w[2 * k + t]), "v"(af[4 * HALF + k]));
                            else acc1[2 * H + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2 * k + t], af[4 * HALF + k], acc1[2 * H + t], 0, 0, 0);
                        }
                }
            };
            auto mma_b = [&](const bf16x8 (&w)[8], auto ks_c, auto half_c) {
                constexpr int KS = decltype(ks_c)::value, HALF = decltype(half_c)::value;
                if (active) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (CHAIN_DBG & 1) asm volatile("" : "+v"(acc2[8 * HALF + k]) : "v"(w[k]), "v"(xf[KS]));
                        else acc2[8 * HALF + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[k], xf[KS], acc2[8 * HALF + k], 0, 0, 0);
                    }
                }
            };
            // between the two halves of a block: my pieces of the next block have landed, this block is in registers (or on its way: lgkmcnt(0));
            // after the barrier the next block is readable and this block's slot takes the block five ahead
            auto sync = [&]() {
                chain_wait_vm<G::WAIT_W>();
                chain_wait_lds();
                chain_bar();
            };
            // epilogue of one half of the chunk (channels 32 s + 8 q .. + 7 of the lane's pixel): residual in, bf16 out in place, the k-step's B fragment
            auto epi = [&](auto s_c) {
                constexpr int S = decltype(s_c)::value;
                if (!active) return;
                char* cell = slot + (S ? re1 : re0);
                f32x4 v[2] = {acc1[2 * S], acc1[2 * S + 1]};
                if (CHAIN_DBG & 4) {      // measurement: the chunk without its epilogue arithmetic
                    bf16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (__bf16)v[e >> 2][e & 3];
                    xf[S] = hv;
                    *reinterpret_cast<bf16x8*>(cell) = hv;
                    return;
                }
                const bf16x8 r = *reinterpret_cast<const bf16x8*>(cell);
                if (!BWD) {
                    const float* ss = reinterpret_cast<const float*>(smem + G::OFF_SS) + c * 64 + 32 * S + 8 * q;
                    v[0] = v[0] * *reinterpret_cast<const f32x4*>(ss) + *reinterpret_cast<const f32x4*>(ss + N1);
                    v[1] = v[1] * *reinterpret_cast<const f32x4*>(ss + 4) + *reinterpret_cast<const f32x4*>(ss + N1 + 4);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] += (float)r[e];
                unsigned bits = 0;
                if (!BWD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) asm("v_max_f32 %0, 0, %1" : "=v"(v[e >> 2][e & 3]) : "v"(v[e >> 2][e & 3]));
#pragma unroll
                    for (int e = 7; e >= 0; --e) bits = __builtin_amdgcn_alignbit(bits, 0u - __float_as_uint(v[e >> 2][e & 3]), 31);
                    *reinterpret_cast<uint8_t*>(bits1 + frow * 128 + 8 * c + 4 * S + q) = (uint8_t)bits;
                } else {
                    bits = *reinterpret_cast<const uint8_t*>(bits1 + frow * 128 + 8 * c + 4 * S + q);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)bits, e, 1);
                        v[e >> 2][e & 3] = __uint_as_float(__float_as_uint(v[e >> 2][e & 3]) & keep);
                    }
                }
                bf16x8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (__bf16)v[e >> 2][e & 3];
                xf[S] = hv;
                *reinterpret_cast<bf16x8*>(cell) = hv;
            };

            // ================= block A0 (P holds its first half) =================
            CT(0)
            read_a(Q, T1{}, true);
            __builtin_amdgcn_sched_barrier(0);
            mma_a(P, T0{}, T0{});
            CT(1)
            sync();
            CT(2)
            issue_w(T1{});
            issue_res(gc + 2, (ci + 2 < NCH) ? ro_cur : ro_nxt);
            read_a(P, T0{}, false);
            __builtin_amdgcn_sched_barrier(0);
            CT(3)
            mma_a(Q, T0{}, T1{});
            // the residual of this chunk was issued two chunks ago (steady state: 24 younger operations; fewer for the first two chunks of the launch);
            // the first chunk of a backward pass also waits for the pass's sign bits (the youngest operations)
            if (BWD && ci == 0) chain_wait_vm<0>();
            else if (gc == 0) chain_wait_vm<G::WAIT_R0>();
            else if (gc == 1) chain_wait_vm<G::WAIT_R1>();
            else chain_wait_vm<G::WAIT_R>();
            __builtin_amdgcn_sched_barrier(0);
            epi(T0{});
            __builtin_amdgcn_sched_barrier(0);
            // ================= block A1 =================
            CT(4)
            read_a(Q, T1{}, true);
            __builtin_amdgcn_sched_barrier(0);
            mma_a(P, T1{}, T0{});
            CT(5)
            sync();
            CT(6)
            issue_w(T2{});
            read_b(P, T0{}, false);
            __builtin_amdgcn_sched_barrier(0);
            CT(7)
            mma_a(Q, T1{}, T1{});
            epi(T1{});
            // the chunk leaves for HBM: 8 rows x 128 B per instruction
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(slot + i * 1024 + 16 * lane);
                __builtin_amdgcn_raw_buffer_store_b128(v, rMid, (CHAIN_DBG & 2) ? OOB : ro_cur[i], c * 128, CHAIN_MID_AUX);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ================= block B0 =================
            CT(8)
            read_b(Q, T1{}, true);
            __builtin_amdgcn_sched_barrier(0);
            mma_b(P, T0{}, T0{});
            CT(9)
            sync();
            CT(10)
            issue_w(T3{});
            read_b(P, T0{}, false);
            __builtin_amdgcn_sched_barrier(0);
            CT(11)
            mma_b(Q, T0{}, T1{});
            __builtin_amdgcn_sched_barrier(0);
            // ================= block B1 =================
            CT(12)
            read_b(Q, T1{}, true);
            __builtin_amdgcn_sched_barrier(0);
            mma_b(P, T1{}, T0{});
            CT(13)
            sync();
            CT(14)
            issue_w(T0{});
            read_a(P, T0{}, false);
            __builtin_amdgcn_sched_barrier(0);
            CT(15)
            mma_b(Q, T1{}, T1{});
            __builtin_amdgcn_sched_barrier(0);
            ++gc;
        }
#ifdef CHAIN_TRACE
        if (tr_on && lane == 0) {
            for (int i = 0; i < 256; ++i) g_chain_trace[grp * 256 + i] = reinterpret_cast<unsigned*>(smem + G::LDS_BYTES)[grp * 256 + i];
            g_chain_trace[512 + 2 * grp] = (unsigned)(__builtin_readcyclecounter() - tr_c0);
            g_chain_trace[513 + 2 * grp] = (unsigned)(__builtin_amdgcn_s_memrealtime() - tr_r0);
        }
#endif
        // ================= end of the pass: the second product's epilogue =================
        if (!BWD) {     // sign bits of mid: the LDS image [row][128 B] leaves with row-contiguous lanes
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(bits1 + i * 1024 + 16 * lane);
                __builtin_amdgcn_raw_buffer_store_b128(v, rB1, bo_cur[i], 0, 0);
            }
        }
        {
            const int m = m0 + frow;
            const bool ok = m < row_hi;
            const unsigned vo = ok ? (unsigned)m * (N2 * 2) + 16u * q : OOB;
            const unsigned vb = ok ? (unsigned)m * (N2 / 8) + (unsigned)q : OOB;
            unsigned mbits[8];
            if (BWD) {      // this pixel's 32 bytes of sign bits, the 8 bytes this lane needs: 8 g + 4 s + q
#pragma unroll
                for (int k = 0; k < 8; ++k) mbits[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rB2, vb, 4 * k, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f32x4 v[2] = {acc2[4 * g + 2 * s], acc2[4 * g + 2 * s + 1]};
                    unsigned bits = 0;
                    if (!BWD) {
                        const float* ss = reinterpret_cast<const float*>(smem + G::OFF_SS) + 2 * N1 + 64 * g + 32 * s + 8 * q;
                        v[0] = v[0] * *reinterpret_cast<const f32x4*>(ss) + *reinterpret_cast<const f32x4*>(ss + N2);
                        v[1] = v[1] * *reinterpret_cast<const f32x4*>(ss + 4) + *reinterpret_cast<const f32x4*>(ss + N2 + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) asm("v_max_f32 %0, 0, %1" : "=v"(v[e >> 2][e & 3]) : "v"(v[e >> 2][e & 3]));
#pragma unroll
                        for (int e = 7; e >= 0; --e) bits = __builtin_amdgcn_alignbit(bits, 0u - __float_as_uint(v[e >> 2][e & 3]), 31);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, rB2, vb, 8 * g + 4 * s, 0);
                    } else {
                        bits = mbits[2 * g + s];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)bits, e, 1);
                            v[e >> 2][e & 3] = __uint_as_float(__float_as_uint(v[e >> 2][e & 3]) & keep);
                        }
                    }
                    bf16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (__bf16)v[e >> 2][e & 3];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rOut, vo, (64 * g + 32 * s) * 2, 0);
                }
            }
        }
        ro_cur[0] = ro_nxt[0], ro_cur[1] = ro_nxt[1];
        row_offsets(pass + 2, ro_nxt, N1 * 2, piece_lane);
    }
    chain_wait_vm<0>();                                   // no LDS-DMA may land after the workgroup has left its LDS
}

template <bool BWD>
void chain_launch(int grid, hipStream_t stream, const ChainParams& p) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = chain_kernel<BWD>;
#ifdef CHAIN_TRACE
    constexpr int lds = ChainGeo::LDS_BYTES + 2048;
#else
    constexpr int lds = ChainGeo::LDS_BYTES;
#endif
    mi_allow_dynamic_lds((const void*)kern, lds, attr_done);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
}

}  // namespace

#ifdef CHAIN_TRACE
extern "C" int mi_chain_trace_read(unsigned* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chain_trace), sizeof(unsigned) * n); }
#endif

extern "C" int mi_conv_chain(const void* a, const void* w_first, const void* res, void* mid, const void* w_second, void* out, long M, int K1, int N1, int N2,
                             const float* scale1, const float* shift1, const float* scale2, const float* shift2, const void* bits1, const void* bits2,
                             void* bits1_out, void* bits2_out, int backward, int grid, void* stream) {
    MI_REQUIRE(a && w_first && res && mid && w_second && out, "mi_conv_chain: null operand");
    MI_REQUIRE(K1 == 256 && N1 == 1024 && N2 == 256, "mi_conv_chain: built for 256 -> 1024 -> 256 channels (layer3 of the ResNet), got %d -> %d -> %d", K1, N1, N2);
    MI_REQUIRE(M > 0 && M * (long)N1 * 2 < (1L << 31), "mi_conv_chain: M=%ld (32-bit buffer offsets)", M);
    MI_REQUIRE(mi_aligned16(a) && mi_aligned16(w_first) && mi_aligned16(res) && mi_aligned16(mid) && mi_aligned16(w_second) && mi_aligned16(out),
               "mi_conv_chain: operands must be 16-byte aligned");
    if (backward) MI_REQUIRE(bits1 && bits2 && mi_aligned16(bits1) && mi_aligned16(bits2), "mi_conv_chain: backward needs the two sign-bit tensors");
    else
        MI_REQUIRE(scale1 && shift1 && scale2 && shift2 && bits1_out && bits2_out && mi_aligned16(bits1_out) && mi_aligned16(bits2_out),
                   "mi_conv_chain: forward needs the FrozenBN coefficients and the two sign-bit outputs");
    static std::atomic<int> n_cu{0};
    int cus = n_cu.load(std::memory_order_relaxed);
    if (cus == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu.store(cus, std::memory_order_relaxed);
    }
    const int mt_total = (int)((M + 15) / 16);
    if (grid <= 0) grid = cus;
    if (grid > mt_total) grid = mt_total;
    ChainParams p;
    p.A = (const __bf16*)a;
    p.Wa = (const __bf16*)w_first;
    p.Wb = (const __bf16*)w_second;
    p.res = (const __bf16*)res;
    p.mid = (__bf16*)mid;
    p.out = (__bf16*)out;
    p.scale1 = scale1, p.shift1 = shift1, p.scale2 = scale2, p.shift2 = shift2;
    p.bits1_in = (const uint8_t*)bits1, p.bits2_in = (const uint8_t*)bits2;
    p.bits1_out = (uint8_t*)bits1_out, p.bits2_out = (uint8_t*)bits2_out;
    p.M = (int)M;
    if (backward) chain_launch<true>(grid, (hipStream_t)stream, p);
    else chain_launch<false>(grid, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_chain");
    return MI_OK;
}
