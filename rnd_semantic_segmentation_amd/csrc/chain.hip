// Two chained 1x1 convolutions of the bottleneck sequence in ONE kernel for gfx950: the wide tensor between them is written once and never re-read.
//
//   forward  (reference core/components/resnet.py:105-113 of block i, then :93-95 of block i+1):
//       x1 = relu(scale1 * (a2 . W3^T) + shift1 + x0)         conv3 + FrozenBN + residual + ReLU      [M][256] -> [M][1024]
//       a1 = relu(scale2 * (x1 . W1^T) + shift2)              conv1 + FrozenBN + ReLU of the next block [M][1024] -> [M][256]
//   backward (the data gradients of the same two convs, in the order backward meets them: conv1 of block i, then conv3 of block i-1):
//       gx = ((ga1 . W1t^T) + g) masked by the sign bits of x     [M][256] -> [M][1024]   (FrozenBN scale folded into the packed weights)
//       ga2 = (gx . W3t^T) masked by the sign bits of a2          [M][1024] -> [M][256]
// As two launches (igemm_nt with the residual epilogue, then igemm_pp) the [M][1024] tensor (154 MB at B = 8, 769 x 769) is written by the first and read
// back by the second; both launches are bound by those bytes, not by the matrix pipe.  Here a wave OWNS its pixel rows through both
// products: the first product's accumulator tile, after its epilogue, IS the second product's B operand (cdna_hip_programming.md section 3, "An
// accumulator tile as the next MFMA's operand") - no LDS round trip, no barrier between the two products.
//
// MFMA orientation as in igemm_nt.hip: D rows = output channels (weights are the A operand), D columns = pixels.  With the weight-row permutation of
// that file a lane (pixel = lane & 15, q = lane >> 4) holds, of every 64-channel chunk of x1, channels 8q .. 8q+7 and 32+8q .. 32+8q+7 - exactly the
// elements the B operand of v_mfma_f32_16x16x32_bf16 wants from that lane for k-steps 0 and 1 of the chunk (k = 8q + j), in natural k order: the
// second product reads the ordinary packed weights [N2][1024].
//
// Structure: one workgroup of 4 waves per CU (one wave per SIMD, up to 512 registers each), persistent over a contiguous range of pixel rows.  A pass
// covers 4 * MT 16-row tiles (wave w: MT of them); per pass the whole of both weight matrices (2 x 512 KB) streams through a 5-slot LDS ring of 16-KB
// blocks shared by the four waves (buffer_load ... lds, four blocks in flight, counted s_waitcnt vmcnt, one raw s_barrier per block).  x1 is produced in
// sixteen 64-channel chunks: GEMM1 of the chunk (K = 256, two ring blocks), epilogue (the residual chunk arrives by LDS-DMA two chunks ahead into a
// wave-private ring; the bf16 result overwrites it in place and leaves for HBM with row-contiguous lanes: 8 rows x 128 B per store instruction), then
// the chunk is the K = 64 slice of GEMM2 (two ring blocks).  Registers per lane at MT = 2: 128 (second accumulator) + 32 (first) + 64 (the a2 rows as B
// fragments, loaded once per pass) + fragments and addresses.
// Every vector-memory operation is a builtin the compiler can see; their order per chunk is fixed (W W+R W+S W W), so the waits are compile-time counts.
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
#define CHAIN_DBG 0            // measurement builds only (tools/dbg/chain_variants.sh): 1 no MFMAs, 2 no residual DMA / mid stores, 4 no epilogue arithmetic, 8 no fragment reads, 16 weight DMA fetches nothing
#endif
#ifndef CHAIN_ROT
#define CHAIN_ROT 0            // chunk rotation per workgroup: first chunk = (blockIdx * CHAIN_ROT) mod 16; 0 = every workgroup starts at chunk 0
#endif
#ifndef CHAIN_MID_AUX
#define CHAIN_MID_AUX 2       // cache policy of the wide tensor's stores: 2 = non-temporal (as igemm_store_staged), 0 = default
#endif

#ifdef CHAIN_TRACE
// Timeline experiment (tools/chaintrace.py builds a second library with -DCHAIN_TRACE=<pass>; never defined in the product build): lane 0 of wave 0 of ONE
// workgroup stamps s_memtime at sixteen points of every chunk of that pass into the 2 KB of LDS the kernel leaves free, dumped at the end of the pass.
__device__ unsigned g_chain_trace[16 * 16 + 4];
#define CT(k)                                                                                  \
    if (tr_on) {                                                                               \
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();                            \
        if (lane == 0) reinterpret_cast<unsigned*>(smem + G::LDS_BYTES)[ci * 16 + (k)] = t_;   \
    }
#else
#define CT(k)
#endif

template <int MT> struct ChainGeo {
    static constexpr int K1 = 256, N1 = 1024, N2 = 256, NC = 64, NCH = N1 / NC;
    static constexpr int WBLK = 16384, NB = 5, LOOK = 4;                 // weight ring: blocks of 16 KB, LOOK in flight
    static constexpr int RES_SLOT = MT * 2048, RES_DEPTH = 3;            // per wave: MT*16 rows x 128 B, chunks c, c+1, c+2
    static constexpr int OFF_RES = NB * WBLK;
    static constexpr int OFF_SS = OFF_RES + 4 * RES_DEPTH * RES_SLOT;    // scale1 | shift1 | scale2 | shift2 (fp32)
    static constexpr int SS_BYTES = (2 * N1 + 2 * N2) * 4;
    static constexpr int OFF_BITS1 = OFF_SS + SS_BYTES;                  // per wave: MT*16 rows x 128 B of sign bits of mid
    static constexpr int BITS1_WAVE = MT * 16 * (N1 / 8);
    static constexpr int OFF_BITS2 = OFF_BITS1 + 4 * BITS1_WAVE;         // per wave: MT*16 rows x 32 B of sign bits of out (backward: read)
    static constexpr int BITS2_WAVE = 1024 * ((MT * 16 * (N2 / 8) + 1023) / 1024);
    static constexpr int LDS_BYTES = OFF_BITS2 + 4 * BITS2_WAVE;
    static constexpr int WAITN = 12 + 4 * MT;                            // vm operations younger than a ring block at the wait for it
    static_assert(LDS_BYTES <= MI_LDS_MAX, "LDS budget");
    static_assert(36 + 8 * MT <= 63, "vmcnt field");
};

// s_waitcnt as the builtin, not inline asm: the compiler's own wait insertion then knows what has already been waited for (with asm waits it put
// s_waitcnt lgkmcnt(14) in front of MFMAs whose fragments the asm lgkmcnt(0) had long retired).  gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4],
// lgkmcnt [11:8]; the fields not meant are left at their maxima.
#define CHAIN_VMCNT(n) (((n) & 15) | (((n) >> 4) << 14) | 0x70 | 0xF00)
__device__ __forceinline__ void chain_wait_vm(int n) {       // n is one of a few compile-time values at every call site
    switch (n) {
#define CW(k) case k: __builtin_amdgcn_s_waitcnt(CHAIN_VMCNT(k)); break;
        CW(0) CW(16) CW(20) CW(24) CW(28) CW(36) CW(42) CW(52) CW(60)
#undef CW
    }
}
__device__ __forceinline__ void chain_wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }     // lgkmcnt(0)

template <int MT, bool BWD>
__global__ __launch_bounds__(256, 1) void chain_kernel(ChainParams p) {
    using G = ChainGeo<MT>;
    constexpr int K1 = G::K1, N1 = G::N1, N2 = G::N2, NCH = G::NCH, NB = G::NB, WBLK = G::WBLK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, q = lane >> 4;

    // ---- this workgroup's 16-row tiles: a contiguous range, as even as 16-row granules allow ----------------------------------
    const int mt_total = (p.M + 15) >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int per = mt_total / nwg, rem = mt_total - per * nwg;
    const int t_lo = bid * per + (bid < rem ? bid : rem), t_hi = t_lo + per + (bid < rem ? 1 : 0);
    const int npass = (t_hi - t_lo + 4 * MT - 1) / (4 * MT);
    if (npass == 0) return;
    const int row_hi = (t_hi * 16 < p.M) ? t_hi * 16 : p.M;            // first row that is not this workgroup's
    const int total_blocks = npass * NCH * 4;
    // Every workgroup streams the same 1 MB of weights; in step they would all pull the same few 4-KB pages - the same L2 channels - at the same time
    // (measured: a block took 1.2k cycles with nothing but its DMA in the loop).  Workgroup b therefore walks the sixteen 64-channel chunks of mid starting at
    // chunk rot(b): the second product's K order differs per workgroup (fp32 rounding only; fixed for a given M and grid, so runs repeat bit for bit).
    const int rot = CHAIN_ROT ? (int)((bid * CHAIN_ROT) & (NCH - 1)) : 0;

    if (!BWD) {                                                        // FrozenBN coefficients -> LDS (read per chunk with ds_read_b128 broadcasts)
        float* ss = reinterpret_cast<float*>(smem + G::OFF_SS);
        for (int i = tid; i < N1; i += 256) ss[i] = p.scale1[i], ss[N1 + i] = p.shift1[i];
        for (int i = tid; i < N2; i += 256) ss[2 * N1 + i] = p.scale2[i], ss[2 * N1 + N2 + i] = p.shift2[i];
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
    // first product's block: 64 rows (the chunk's output channels) x 256 B (one half of K1); 16-B slot u of row r sits at u ^ fa(r)
    // second product's block: 128 rows (one half of N2) x 128 B (the chunk's 64 k); slot u of row r at u ^ fb(r)
    // both brute-forced so that every ds_read_b128 lane group of the permuted-row fragment reads hits 16 distinct slots (tools/dbg/chain_swizzle.py)
    auto fa = [](int r) { return (r & 15) ^ ((r & 1) << 2); };
    auto fb = [](int r) { return (r & 7) ^ (((r >> 3) & 1) << 2); };
    unsigned wsa[4], wsb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = 4 * wave + i;
        const int ra = 4 * pc + (lane >> 4), rb = 8 * pc + (lane >> 3);
        wsa[i] = (unsigned)(ra * (K1 * 2) + 16 * ((lane & 15) ^ fa(ra)));
        wsb[i] = (unsigned)(rb * (N1 * 2) + 16 * ((lane & 7) ^ fb(rb)));
    }
    int fra[4][4], frb[8][2];          // fragment addresses inside a block
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = 32 * (t >> 1) + 4 * (t & 1) + 8 * (frow >> 2) + (frow & 3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fra[t][ks] = r * 256 + 16 * ((4 * ks + q) ^ fa(r));
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int r = 64 * (t >> 2) + 32 * ((t >> 1) & 1) + 4 * (t & 1) + 8 * (frow >> 2) + (frow & 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) frb[t][ks] = r * 128 + 16 * ((4 * ks + q) ^ fb(r));
    }
    int slot_wr = 0, slot_rd = 0;      // ring slots of the next block to issue / to read
    int bk_issue = 0;                  // blocks issued so far
    // issue one block: TYPE 0 / 1 = the two K halves of the first product's chunk, 2 / 3 = the two N2 halves of the second product's chunk
    auto issue_w = [&](auto type_c) {
        constexpr int TYPE = decltype(type_c)::value;
        const int c = ((bk_issue >> 2) + rot) & (NCH - 1);
        const unsigned kill = (bk_issue < total_blocks && !(CHAIN_DBG & 16)) ? 0u : OOB;             // past the end: the same instruction count, no traffic
        char* dst = smem + slot_wr * WBLK + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (TYPE < 2) blds16(rWa, wsa[i] | kill, (unsigned)(c * (64 * K1 * 2) + TYPE * 256), dst + i * 1024);
            else blds16(rWb, wsb[i] | kill, (unsigned)((TYPE - 2) * (128 * N1 * 2) + c * 128), dst + i * 1024);
        }
        ++bk_issue;
        slot_wr = slot_wr == NB - 1 ? 0 : slot_wr + 1;
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;

    // ---- wave-private rows ---------------------------------------------------------------------------------------------------------
    // a pass covers tiles t_lo + pass * 4 MT + wave * MT + (0 .. MT-1).  Row-contiguous pieces (8 rows x 128 B of a 64-channel chunk): lane -> row
    // 8 i + (lane >> 3), 16-B slot (lane & 7) ^ (row & 7) (residual DMA source and mid store target; the LDS image is lane-linear).
    char* const res_base = smem + G::OFF_RES + wave * (G::RES_DEPTH * G::RES_SLOT);
    auto row_offsets = [&](int pass, unsigned (&off)[2 * MT], unsigned rowbytes, unsigned lane_bytes) {
        const int m0 = (t_lo + pass * 4 * MT + wave * MT) * 16;
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            const int m = m0 + 8 * i + (lane >> 3);
            off[i] = (pass < npass && m < row_hi) ? (unsigned)m * rowbytes + lane_bytes : OOB;
        }
    };
    const unsigned piece_lane = 16u * (unsigned)((lane & 7) ^ ((lane >> 3) & 7));
    unsigned ro_cur[2 * MT], ro_nxt[2 * MT];           // mid / res row offsets of this pass and of the next
    row_offsets(0, ro_cur, N1 * 2, piece_lane);
    row_offsets(1, ro_nxt, N1 * 2, piece_lane);
    int gc = 0;                                        // chunks done (over all passes); residual ring slot = chunk % 3
    auto issue_res = [&](int gcp, const unsigned (&off)[2 * MT]) {        // residual of global chunk gcp
        char* dst = res_base + (gcp % G::RES_DEPTH) * G::RES_SLOT;
        const unsigned so = (unsigned)(((gcp + rot) & (NCH - 1)) * 128);
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) blds16(rRes, (CHAIN_DBG & 2) ? OOB : off[i], so, dst + i * 1024);
    };

    f32x4 acc2[16][MT];
    bf16x8 af[8][MT];
    const int rsw = frow & 7;
    const int re0 = frow * 128 + 16 * (q ^ rsw), re1 = frow * 128 + 16 * ((4 + q) ^ rsw);     // this lane's two 16-B slots of its row (+ j * 2048)
    char* const bits1 = smem + G::OFF_BITS1 + wave * G::BITS1_WAVE;
    char* const bits2 = smem + G::OFF_BITS2 + wave * G::BITS2_WAVE;

    // ---- block pipeline ---------------------------------------------------------------------------------------------------------------
    // The 16 fragments of block b+1 are read into one of two register sets (X, Y) in the MIDDLE of block b's MFMAs: there the wave waits for its DMA
    // pieces of block b+1, joins the barrier (all pieces of b+1 landed; every wave has block b in registers, so its slot is free), issues block b+5 into
    // that slot and the 16 ds_read_b128 of block b+1, whose latency the second half of block b's MFMAs covers.
    bf16x8 X[16], Y[16];
    auto read_a = [&](bf16x8 (&w)[16]) {
        const char* blk = smem + slot_rd * WBLK;
        slot_rd = slot_rd == NB - 1 ? 0 : slot_rd + 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (CHAIN_DBG & 8) asm volatile("" : "=v"(w[4 * ks + t]) : "v"(blk + fra[t][ks]));
                else w[4 * ks + t] = *reinterpret_cast<const bf16x8*>(blk + fra[t][ks]);
            }
    };
    auto read_b = [&](bf16x8 (&w)[16]) {
        const char* blk = smem + slot_rd * WBLK;
        slot_rd = slot_rd == NB - 1 ? 0 : slot_rd + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (CHAIN_DBG & 8) asm volatile("" : "=v"(w[8 * ks + t]) : "v"(blk + frb[t][ks]));
                else w[8 * ks + t] = *reinterpret_cast<const bf16x8*>(blk + frb[t][ks]);
            }
    };
    auto mid_sync = [&](int n) {
        chain_wait_vm(n);
        chain_wait_lds();                                        // the block being computed is in registers
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: five weight blocks and two residual chunks in flight; block 0 in registers ---------------------------------------
    issue_w(T0{});
    issue_w(T1{});
    issue_w(T2{});
    issue_w(T3{});
    issue_w(T0{});
    issue_res(0, ro_cur);
    issue_res(1, ro_cur);
    mid_sync(16 + 4 * MT);
    read_a(X);
    __builtin_amdgcn_sched_barrier(0);

    for (int pass = 0; pass < npass; ++pass) {
        const int m0 = (t_lo + pass * 4 * MT + wave * MT) * 16;
#ifdef CHAIN_TRACE
        const bool tr_on = bid == 37 && wave == 0 && pass == CHAIN_TRACE;
        const unsigned long long tr_c0 = __builtin_readcyclecounter(), tr_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        // the a2 rows of this wave as B fragments: lane (pixel frow, q) holds k = 32 s + 8 q .. + 7 of its pixel for the 8 k-steps
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + 16 * j + frow;
            const unsigned vo = m < row_hi ? (unsigned)m * (K1 * 2) + 16u * q : OOB;
#pragma unroll
            for (int s = 0; s < 8; ++s) af[s][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rA, vo, 64 * s, 0));
        }
        unsigned bo_cur[2 * MT];                       // sign bits of mid: 8 rows x 128 B pieces, lane -> row 8 i + (lane >> 3), bytes 16 (lane & 7)
        row_offsets(pass, bo_cur, N1 / 8, 16u * (lane & 7));
        if (BWD) {
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) blds16(rB1, bo_cur[i], 0, bits1 + i * 1024);
            {   // sign bits of out: rows of 32 B; lane -> row lane >> 1, bytes 16 (lane & 1); MT * 16 rows = MT / 2 pieces (MT even) or one partial piece
                const int m = m0 + (lane >> 1);
#pragma unroll
                for (int i = 0; i < (MT + 1) / 2; ++i) {
                    const int mm = m + 32 * i;
                    blds16(rB2, (mm < row_hi && (lane >> 1) + 32 * i < 16 * MT) ? (unsigned)mm * (N2 / 8) + 16u * (lane & 1) : OOB, 0, bits2 + i * 1024);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc2[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ci = 0; ci < NCH; ++ci) {
            const int c = (ci + rot) & (NCH - 1);          // the chunk of mid this iteration produces and consumes
            CT(0)
            f32x4 acc1[4][MT];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc1[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // ================= first product: two ring blocks (K halves); X holds block A0 =================
            auto mma_a = [&](const bf16x8 (&w)[16], auto h_c, auto half_c) {
                constexpr int H = decltype(h_c)::value, HALF = decltype(half_c)::value;
#pragma unroll
                for (int ks = 2 * HALF; ks < 2 * HALF + 2; ++ks)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < MT; ++j) {
                            if (CHAIN_DBG & 1) asm volatile("" : "+v"(acc1[t][j]) : "v"(w[4 * ks + t]), "v"(af[4 * H + ks][j]));
                            else acc1[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[4 * ks + t], af[4 * H + ks][j], acc1[t][j], 0, 0, 0);
                        }
            };
            mma_a(X, T0{}, T0{});
            CT(1)
            mid_sync(G::WAITN);
            CT(2)
            issue_w(T1{});
            issue_res(gc + 2, (ci + 2 < NCH) ? ro_cur : ro_nxt);
            read_a(Y);
            CT(3)
            __builtin_amdgcn_sched_barrier(0);
            mma_a(X, T0{}, T1{});
            __builtin_amdgcn_sched_barrier(0);
            mma_a(Y, T1{}, T0{});
            CT(4)
            mid_sync(G::WAITN);
            CT(5)
            issue_w(T2{});
            read_b(X);
            CT(6)
            __builtin_amdgcn_sched_barrier(0);
            mma_a(Y, T1{}, T1{});
            __builtin_amdgcn_sched_barrier(0);
            CT(7)
            // ================= epilogue of the chunk: residual in, bf16 chunk out (in place in LDS), B fragments of the second product =================
            // the residual of this chunk was issued two chunks ago; vm operations issued since (steady state): 36 + 8 MT, fewer for the first two chunks
            // of the launch; the first chunk of a pass with sign bits to read waits for everything (those DMAs are the youngest operations)
            if (BWD && ci == 0) chain_wait_vm(0);
            else if (gc == 0) chain_wait_vm(8 + 4 * MT);
            else if (gc == 1) chain_wait_vm(24 + 6 * MT);
            else chain_wait_vm(36 + 8 * MT);
            __builtin_amdgcn_sched_barrier(0);
            CT(8)
            char* const slot = res_base + (gc % G::RES_DEPTH) * G::RES_SLOT;
            bf16x8 xf[2][MT];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 sc[2], sh[2];
                if (!BWD) {
                    const float* ss = reinterpret_cast<const float*>(smem + G::OFF_SS) + c * 64 + 32 * s + 8 * q;
                    sc[0] = *reinterpret_cast<const f32x4*>(ss), sc[1] = *reinterpret_cast<const f32x4*>(ss + 4);
                    sh[0] = *reinterpret_cast<const f32x4*>(ss + N1), sh[1] = *reinterpret_cast<const f32x4*>(ss + N1 + 4);
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    char* cell = slot + j * 2048 + (s ? re1 : re0);
                    f32x4 v[2] = {acc1[2 * s][j], acc1[2 * s + 1][j]};
                    if (CHAIN_DBG & 4) {      // measurement: the chunk without its epilogue arithmetic
                        bf16x8 hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = (__bf16)v[e >> 2][e & 3];
                        xf[s][j] = hv;
                        *reinterpret_cast<bf16x8*>(cell) = hv;
                        continue;
                    }
                    const bf16x8 r = *reinterpret_cast<const bf16x8*>(cell);
                    if (!BWD) {
                        v[0] = v[0] * sc[0] + sh[0];
                        v[1] = v[1] * sc[1] + sh[1];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] += (float)r[e];
                    unsigned bits = 0;
                    if (!BWD) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) asm("v_max_f32 %0, 0, %1" : "=v"(v[e >> 2][e & 3]) : "v"(v[e >> 2][e & 3]));
#pragma unroll
                        for (int e = 7; e >= 0; --e) bits = __builtin_amdgcn_alignbit(bits, 0u - __float_as_uint(v[e >> 2][e & 3]), 31);
                        *reinterpret_cast<uint8_t*>(bits1 + (16 * j + frow) * 128 + 8 * c + 4 * s + q) = (uint8_t)bits;
                    } else {
                        bits = *reinterpret_cast<const uint8_t*>(bits1 + (16 * j + frow) * 128 + 8 * c + 4 * s + q);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)bits, e, 1);
                            v[e >> 2][e & 3] = __uint_as_float(__float_as_uint(v[e >> 2][e & 3]) & keep);
                        }
                    }
                    bf16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (__bf16)v[e >> 2][e & 3];
                    xf[s][j] = hv;
                    *reinterpret_cast<bf16x8*>(cell) = hv;
                }
            }
            // the chunk leaves for HBM: 8 rows x 128 B per instruction
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(slot + i * 1024 + 16 * lane);
                __builtin_amdgcn_raw_buffer_store_b128(v, rMid, (CHAIN_DBG & 2) ? OOB : ro_cur[i], c * 128, CHAIN_MID_AUX);
            }
            CT(9)
            // ================= second product: two ring blocks (N2 halves), K = this chunk; X holds block B0 =================
            auto mma_b = [&](const bf16x8 (&w)[16], auto h_c, auto half_c) {
                constexpr int H = decltype(h_c)::value, HALF = decltype(half_c)::value;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int j = 0; j < MT; ++j) {
                        if (CHAIN_DBG & 1) asm volatile("" : "+v"(acc2[8 * H + t][j]) : "v"(w[8 * HALF + t]), "v"(xf[HALF][j]));
                        else acc2[8 * H + t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[8 * HALF + t], xf[HALF][j], acc2[8 * H + t][j], 0, 0, 0);
                    }
            };
            __builtin_amdgcn_sched_barrier(0);
            mma_b(X, T0{}, T0{});
            CT(10)
            mid_sync(G::WAITN);
            CT(11)
            issue_w(T3{});
            read_b(Y);
            CT(12)
            __builtin_amdgcn_sched_barrier(0);
            mma_b(X, T0{}, T1{});
            __builtin_amdgcn_sched_barrier(0);
            mma_b(Y, T1{}, T0{});
            CT(13)
            mid_sync(G::WAITN);
            CT(14)
            issue_w(T0{});
            read_a(X);
            CT(15)
            __builtin_amdgcn_sched_barrier(0);
            mma_b(Y, T1{}, T1{});
            __builtin_amdgcn_sched_barrier(0);
            ++gc;
        }
#ifdef CHAIN_TRACE
        if (tr_on && lane == 0) {
            for (int i = 0; i < 256; ++i) g_chain_trace[i] = reinterpret_cast<unsigned*>(smem + G::LDS_BYTES)[i];
            g_chain_trace[256] = (unsigned)(__builtin_readcyclecounter() - tr_c0);
            g_chain_trace[257] = (unsigned)(__builtin_amdgcn_s_memrealtime() - tr_r0);
        }
#endif
        // ================= end of the pass: the second product's epilogue =================
        if (!BWD) {     // sign bits of mid: the LDS image [row][128 B] leaves with row-contiguous lanes
#pragma unroll
            for (int i = 0; i < 2 * MT; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(bits1 + i * 1024 + 16 * lane);
                __builtin_amdgcn_raw_buffer_store_b128(v, rB1, bo_cur[i], 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + 16 * j + frow;
            const bool ok = m < row_hi;
            const unsigned vo = ok ? (unsigned)m * (N2 * 2) + 16u * q : OOB;
            const unsigned vb = ok ? (unsigned)m * (N2 / 8) + (unsigned)q : OOB;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f32x4 v[2] = {acc2[4 * g + 2 * s][j], acc2[4 * g + 2 * s + 1][j]};
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
                        bits = *reinterpret_cast<const uint8_t*>(bits2 + (16 * j + frow) * 32 + 8 * g + 4 * s + q);
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
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) ro_cur[i] = ro_nxt[i];
        row_offsets(pass + 2, ro_nxt, N1 * 2, piece_lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may land after the workgroup has left its LDS
}

template <int MT, bool BWD>
void chain_launch(int grid, hipStream_t stream, const ChainParams& p) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = chain_kernel<MT, BWD>;
#ifdef CHAIN_TRACE
    constexpr int lds = ChainGeo<MT>::LDS_BYTES + 1024;
#else
    constexpr int lds = ChainGeo<MT>::LDS_BYTES;
#endif
    mi_allow_dynamic_lds((const void*)kern, lds, attr_done);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
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
    if (backward) chain_launch<2, true>(grid, (hipStream_t)stream, p);
    else chain_launch<2, false>(grid, (hipStream_t)stream, p);
    MI_CHECK_LAUNCH("mi_conv_chain");
    return MI_OK;
}
