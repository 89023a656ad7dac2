// EXPERIMENT, NOT BUILT (kept for the record).  Correct (passed tests/test_gpu_ops.py::
// test_pointwise_wide_conv_all_hot_epilogues_vs_torch incl. the BASELINE shape) but no faster than the tile kernel:
// 256 -> 1024 with residual 131.9 us vs 129.1 us, without residual 97 vs 88 us.  To build it, add it to SOURCES in
// __graft_entry__.py and call mi_try_astat(&p, stream) from mi_conv_gemm before the tile selection.
// A-stationary 1x1 convolution for gfx950: the short-K, wide-N GEMMs around the ResNet bottlenecks
//
//   out[m][n] = epi( sum_c A[m][c] * Wp[n][c] )        K = Ca in {64, 128, 256},  N = 4*K (conv3 forward, conv1 data gradient)
//
// In the tile-per-workgroup kernel (igemm_nt.hip) these launches spend two thirds of a workgroup's life outside the MFMA loop:
// K = 256 is four K-steps, every one of the N/128 workgroups of a row block re-reads the same A rows from L2, waits out an
// HBM round trip for its residual tile before its first MFMA and stores only after its last.  Here ONE workgroup owns BM
// output rows and ALL N columns:
//   * the A panel [BM][K] is DMA'd into LDS once and stays (80 KiB at BM = 160, K = 256);
//   * the weight tiles [128 n][64 k] (16 KiB, L2-resident: the whole operand is <= 512 KiB) stream through a three-slot ring,
//     two stages ahead, across the N-tile boundaries - the loop never drains the DMA queue;
//   * the residual rows / mask bytes of N-tile t are requested at its first K-step and land behind its MFMAs, its stores retire
//     behind the first K-step of tile t+1.
// Five waves.  vmcnt retires loads, stores and LDS-DMA of a wave together in issue order, so a wave that waits for a weight DMA
// also waits for every older residual load (an HBM round trip) and output store it issued; measured, that cost more than the
// structure gained.  Hence the weight DMAs live in a wave of their own:
//   wave 4 (loader):   barrier_s -> DMA W(s+2) into slot (s+2)%3 (= slot of step s-1, which every compute wave has left) ->
//                      s_waitcnt vmcnt(16): W(s+1) landed, W(s+2) may fly (16 DMAs per tile, out-of-range rows read the zero
//                      page, so the count is exact)
//   waves 0-3:         barrier_s -> [first step of an N-tile: residual / mask loads] -> MFMAs on slot s%3; after the last
//                      K-step of the tile the fused epilogue.  They never wait on a DMA; the compiler's own waits cover
//                      their register loads.
// Both sides execute exactly S = NT*KC barriers.
// Same lane layout, swizzles and fused epilogue as igemm_nt.hip (igemm_common.h).  Replaces nn.Conv2d 1x1 + FrozenBN / residual /
// ReLU at reference core/components/resnet.py:99-113 for those shapes.
#include "igemm_common.h"
#include <stdlib.h>

namespace {

constexpr int WTILE = 128 * 128;      // [128 n][64 k] bf16
constexpr int NSLOT = 3;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MT, int KC>
struct AGeo {
    static constexpr int BM = 32 * MT;
    static constexpr int ATILE = BM * 128;                       // one 64-channel chunk of the panel
    static constexpr int LDS_BYTES = KC * ATILE + NSLOT * WTILE;
};

template <int MT, int KC, int EPI>
__global__ __launch_bounds__(320, 1) void igemm_astat_kernel(IgemmParams p) {
    using G = AGeo<MT, KC>;
    constexpr int BM = G::BM, ATILE = G::ATILE;
    static_assert(EPI >= 0, "compile-time epilogue flag set");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sW = smem + KC * ATILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    const int prow = lane >> 3, pch = lane & 7;
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const int NT = p.n_tiles, S = NT * KC;
    const long row_bytes = (long)p.Ca * 2;

    if (wave == 4) {
        // ---- loader wave: all weight-tile DMAs.  Its vmcnt only ever holds those DMAs, so the counted waits are exact and are
        //      never held up by residual loads or output stores (vmcnt retires in issue order per wave).
        //      A tile = 16 pieces (8 rows x 128 B): piece i holds rows 8i + prow.
        const char* w_src[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            w_src[i] = reinterpret_cast<const char*>(p.Wp) + (long)(i * 8 + prow) * row_bytes + ((pch ^ ((prow & 3) | ((i & 1) << 2))) << 4);
        auto stage_w = [&](int s, int slot) __attribute__((always_inline)) {
            const int nt = s / KC, kc = s - nt * KC;
            const long off = (long)nt * 128 * row_bytes + kc * 128;
            char* dst = sW + slot * WTILE;
#pragma unroll
            for (int i = 0; i < 16; ++i) glds16((nt * 128 + i * 8 + prow < p.N) ? w_src[i] + off : zero, dst + i * 1024);
        };
        stage_w(0, 0);
        if (S > 1) {
            stage_w(1, 1);
            wait_vm<16>();
        } else {
            wait_vm<0>();
        }
        int slot = 0;
        for (int s = 0; s < S; ++s) {
            __builtin_amdgcn_s_barrier();          // publishes W(s); every compute wave has finished step s-1, whose slot is free
            const bool more = s + 2 < S;
            if (more) stage_w(s + 2, slot == 0 ? 2 : slot - 1);
            slot = slot == 2 ? 0 : slot + 1;
            if (more) wait_vm<16>();               // W(s+1) has landed, W(s+2) may fly
            else wait_vm<0>();
        }
        return;
    }

    // ---- compute waves 0..3: the A panel (wave w moves pieces w*MT .. w*MT+MT-1 of every 64-channel chunk), then MFMAs + epilogues
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + (wave * MT + i) * 8 + prow;
        const bool ok = m < p.M;
        const char* src = ok ? reinterpret_cast<const char*>(p.A) + (long)m * row_bytes + ((pch ^ (prow & 7)) << 4) : zero;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) glds16(src + (ok ? kc * 128 : 0), sA + kc * ATILE + (wave * MT + i) * 1024);
    }
    const int wm = wave & 1, wn = wave >> 1;
    const int frow = lane & 15, fq = lane >> 4;
    const int wrow0 = wn * 64 + 8 * (frow >> 2) + (frow & 3);
    f32x4 acc[4][MT];
    auto compute = [&](int kc, int slot) __attribute__((always_inline)) {
        const char* sa = sA + kc * ATILE;
        const char* sb = sW + slot * WTILE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = (((kk * 4 + fq) ^ (frow & 7)) << 4);
            bf16x8 wf[4], af[MT];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sb + (wrow0 + 32 * (i >> 1) + 4 * (i & 1)) * 128 + sw);
#pragma unroll
            for (int j = 0; j < MT; ++j) af[j] = *reinterpret_cast<const bf16x8*>(sa + (wm * (MT * 16) + j * 16 + frow) * 128 + sw);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    };
    wait_vm<0>();                      // this wave's pieces of the panel; the first barrier publishes everyone's
    int slot = 0;
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = nt * 128;
        bf16x8 pres[MT][2];
        unsigned pbits[MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            __builtin_amdgcn_s_barrier();
            if (kc == 0) igemm_fetch_epilogue<MT, EPI>(p, m0, n0, wm, wn, frow, fq, pres, pbits);   // lands behind this tile's MFMAs
            compute(kc, slot);
            slot = slot == 2 ? 0 : slot + 1;
        }
        igemm_epilogue<MT, EPI>(p, acc, m0, n0, wm, wn, frow, fq, pres, pbits);
    }
}

template <int MT, int KC, int EPI>
void launch_astat(const IgemmParams& p, hipStream_t stream) {
    static bool attr_done = false;
    auto kern = igemm_astat_kernel<MT, KC, EPI>;
    constexpr int lds = AGeo<MT, KC>::LDS_BYTES;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.m_tiles), dim3(320), lds, stream, p);
}

template <int MT, int KC>
bool launch_astat_epi(const IgemmParams& p, hipStream_t stream) {
    switch (p.flags) {
        case 69: launch_astat<MT, KC, 69>(p, stream); return true;
        case 71: launch_astat<MT, KC, 71>(p, stream); return true;
        case 128: launch_astat<MT, KC, 128>(p, stream); return true;
        case 130: launch_astat<MT, KC, 130>(p, stream); return true;
        default: return false;
    }
}

}  // namespace

// Called by mi_conv_gemm (igemm_nt.hip) for stride-1 1x1 convs; returns 1 if it took the launch, 0 if the shape / flag set is not
// one it is built for.  `params` is the caller's IgemmParams.
__attribute__((visibility("hidden"))) int mi_try_astat(const void* params, void* stream) {
    IgemmParams p = *reinterpret_cast<const IgemmParams*>(params);
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("MI_IGEMM_ASTAT");
        enabled = e ? atoi(e) : 1;
    }
    if (!enabled || p.T != 1 || p.pad != 0 || p.stride != 1 || p.N < 256 || p.N < 2 * p.Ca) return 0;
    if (p.Ca != 64 && p.Ca != 128 && p.Ca != 256) return 0;
    p.n_tiles = (p.N + 127) / 128;
    // 160-row panels unless 128-row ones waste less of the last round (one workgroup per CU at K = 256)
    const long t5 = (p.M + 159) / 160, t4 = (p.M + 127) / 128;
    const bool mt5 = ((t5 + 255) / 256) * 160 <= ((t4 + 255) / 256) * 128;
    p.m_tiles = (int)(mt5 ? t5 : t4);
    const hipStream_t st = (hipStream_t)stream;
    const int kc = p.Ca / 64;
    bool ok = false;
    if (mt5) {
        if (kc == 4) ok = launch_astat_epi<5, 4>(p, st);
        else if (kc == 2) ok = launch_astat_epi<5, 2>(p, st);
        else ok = launch_astat_epi<5, 1>(p, st);
    } else {
        if (kc == 4) ok = launch_astat_epi<4, 4>(p, st);
        else if (kc == 2) ok = launch_astat_epi<4, 2>(p, st);
        else ok = launch_astat_epi<4, 1>(p, st);
    }
    return ok ? 1 : 0;
}
