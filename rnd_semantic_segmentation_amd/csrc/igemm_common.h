// Shared pieces of the implicit-GEMM conv kernel (igemm_nt.hip): parameter block, zero page, LDS-DMA helper
// and the fused epilogue.  See igemm_nt.hip for the tile / lane layout the epilogue assumes.
#pragma once
#include "mi_common.h"

namespace {

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
    int korder;                   // igemm_pp_kernel: 0 = contraction runs tap-major (tap, then channels), 1 = channel-chunk-major (all taps of a chunk)
    // MI_EPI_STATS (compile-time epilogue sets only): per wave row tile (MT*16 rows) and channel, sum (o - pilot) and sum (o - pilot)^2 of the
    // bf16-rounded outputs o: stats[row tile][2][N] fp32, row tile = first row of the wave / (MT*16); reduced in a fixed order afterwards
    float* stats;
    const float* pilot;
};

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4); every lane receives the total
__device__ __forceinline__ float igemm_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the same with the non-temporal cache policy (aux bit 1 = nt on gfx94x / gfx950): for tensors a launch reads exactly once
__device__ __forceinline__ void glds16_nt(const char* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

// Buffer form of the same DMA: per-lane 32-bit byte offset + scalar offset against a buffer resource; a lane whose offset is out
// of range for the resource (bit 31 set, num_records < 2^31) writes zeros.
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}


// The epilogue's residual rows and ReLU sign bits (150-300 MB of HBM traffic per launch on the 1024/2048-channel tensors) are
// fetched as ONE batch of independent loads per lane - before the main loop (they land behind the MFMA work) or right after
// it (the registers of the operand fragments are free then).  Left inside the per-row loop they would be serialised behind
// the stores (out / res may alias as far as the compiler knows): one HBM round trip per 16 rows.
template <int MT, int EPI, bool WANT_RES = true>
__device__ __forceinline__ void igemm_fetch_epilogue(const IgemmParams& p, int m0, int n0, int wm, int wn, int frow, int fq, bf16x8 (&pres)[MT][2],
                                                     unsigned (&pbits)[MT], const int* mrow = nullptr) {
    const int flags = (EPI >= 0 ? EPI : p.flags) & (WANT_RES ? ~0 : ~MI_EPI_RESIDUAL);
    const int nbp = n0 + wn * 64 + 8 * fq;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        // mrow (igemm_pw_kernel): the tile's rows are padded pixel coordinates; mrow[j] is the pixel of this lane's row j, or -1
        const int m = mrow ? (mrow[j] < 0 ? p.M : mrow[j]) : m0 + wm * (MT * 16) + j * 16 + frow;
        const long o0 = (long)(m < p.M ? m : 0) * p.N + (nbp < p.N ? nbp : 0);
        const long o1 = (long)(m < p.M ? m : 0) * p.N + (nbp + 32 < p.N ? nbp + 32 : 0);
        if (flags & MI_EPI_RESIDUAL) {
            pres[j][0] = *reinterpret_cast<const bf16x8*>(p.res + o0);
            pres[j][1] = *reinterpret_cast<const bf16x8*>(p.res + o1);
        }
        if (flags & MI_EPI_BITMASK) {
            const uint8_t* mb = reinterpret_cast<const uint8_t*>(p.msk);
            pbits[j] = (unsigned)mb[o0 >> 3] | ((unsigned)mb[o1 >> 3] << 8);
        }
    }
}

// EPI >= 0: the epilogue flag set is a compile-time constant (straight-line code); EPI < 0: flags are read from the parameters.
// acc[i][j]: MFMA tile i (channels, see below) x 16-row tile j of the wave's 16*MT x 64 outputs; (wm, wn) = wave position in the
// block tile whose first row / column are m0 / n0; frow = lane & 15, fq = lane >> 4.
// STAGED: the bf16 results go to an LDS image of the block tile ([BM rows][256 B], 16-B chunk c of row r at c ^ (r & 15)) instead of
// global memory; igemm_store_staged() then writes the tile out with row-contiguous lanes.
template <int MT, int EPI, bool STAGED = false>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[4][MT], int m0, int n0, int wm, int wn, int frow, int fq,
                                               bf16x8 (&pres)[MT][2], unsigned (&pbits)[MT], char* stage = nullptr, bool mask_lds = false,
                                               const int* mrow = nullptr) {
    const int flags = EPI >= 0 ? EPI : p.flags;
    constexpr bool STATS = EPI >= 0 && (EPI & MI_EPI_STATS) != 0;
    float st0[STATS ? 16 : 1], st1[STATS ? 16 : 1];
    // ---- epilogue: lane owns pixel m (D col) and two groups of 8 contiguous channels: h = 0 -> nb .. nb+7 (MFMA tiles 0, 1),
    //      h = 1 -> nb+32 .. nb+39 (tiles 2, 3); tile i holds channels nb + 32*(i>>1) + 4*(i&1) + (0..3) ------------------------
    const int nb = n0 + wn * 64 + 8 * fq;
    f32x4 esc[4], ebi[4];               // FrozenBN scale / shift of this lane's 16 channels: loaded once, not per pixel
    if (flags & MI_EPI_SCALE_BIAS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ni = nb + 32 * (i >> 1) + 4 * (i & 1);
            const int n = (ni < p.N) ? ni : 0;
            esc[i] = *reinterpret_cast<const f32x4*>(p.scale + n);
            ebi[i] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }
    }
    f32x4 epil[4];                      // MI_EPI_STATS: the pilot (running mean) of this lane's 16 channels
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ni = nb + 32 * (i >> 1) + 4 * (i & 1);
            epil[i] = *reinterpret_cast<const f32x4*>(p.pilot + ((ni < p.N) ? ni : 0));
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) st0[k] = st1[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(epil[i]));
    }
    // Everything the row loop below consumes must have LANDED before the loop: the loop body sits behind per-row exec-mask
    // branches (m < M, n < N), and hipcc places the s_waitcnt for a pending load inside the first conditional block that uses
    // it - a path that skips that block still has the load pending, so EVERY later block gets its own s_waitcnt vmcnt(0),
    // which also waits for the previous block's stores: ten dependent HBM round trips per tile.  A no-op asm that reads the
    // registers puts the one wait here, in straight-line code.
    if (flags & MI_EPI_SCALE_BIAS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(esc[i]), "v"(ebi[i]));
    }
    if (flags & MI_EPI_RESIDUAL) {
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(pres[j][0]), "v"(pres[j][1]));
    }
    if (flags & MI_EPI_BITMASK) {
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(pbits[j]));
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = mrow ? (mrow[j] < 0 ? p.M : mrow[j]) : m0 + wm * (MT * 16) + j * 16 + frow;
        if (m >= p.M) continue;
        const long o = (long)m * p.N + nb;       // group h starts at o + 32*h
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
                if (nb + 32 * h < p.N) {
                    const bf16x8 r = pres[j][h];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[2 * h + (e >> 2)][e & 3] += (float)r[e];
                }
            }
        }
        if (flags & MI_EPI_RELU) {
            if (flags & MI_EPI_LEAKY) {                                     // LeakyReLU(alpha)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] > 0.f ? v[i][e] : p.alpha * v[i][e];
            } else {                                                        // one v_max_f32 per value; the result is never -0
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, 0, %1" : "=v"(v[i][e]) : "v"(v[i][e]));   // fmaxf() costs two (canonicalize)
            }
        }
        if (flags & MI_EPI_MASK) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 32 * h < p.N) {
                    const bf16x8 k = *reinterpret_cast<const bf16x8*>(p.msk + o + 32 * h);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[2 * h + (e >> 2)][e & 3] = ((float)k[e] > 0.f) ? v[2 * h + (e >> 2)][e & 3] : 0.f;
                }
            }
        }
        if (flags & MI_EPI_BITMASK) {
            // the packed sign bits are addressed as bytes here: bit c%8 of byte c/8 == bit c%16 of the uint16 word c/16
            const float neg = (flags & MI_EPI_LEAKY) ? p.alpha : 0.f;       // backward of LeakyReLU: gradient x alpha where the sign bit is 0
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 32 * h < p.N) {
                    const unsigned bits = (pbits[j] >> (8 * h)) & 0xffu;
                    if (flags & MI_EPI_LEAKY) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[2 * h + (c >> 2)][c & 3] = ((bits >> c) & 1u) ? v[2 * h + (c >> 2)][c & 3] : neg * v[2 * h + (c >> 2)][c & 3];
                    } else {      // two ops per value: sign-extend bit c to a word mask (v_bfe_i32), AND it onto the float
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const unsigned keep = (unsigned)__builtin_amdgcn_sbfe((int)bits, c, 1);
                            v[2 * h + (c >> 2)][c & 3] = __uint_as_float(__float_as_uint(v[2 * h + (c >> 2)][c & 3]) & keep);
                        }
                    }
                }
            }
        }
        if (flags & MI_EPI_WRITE_MASK) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 32 * h < p.N) {
                    unsigned bits = 0;
                    if ((flags & MI_EPI_RELU) && !(flags & MI_EPI_LEAKY)) {
                        // after ReLU every value is >= +0, so (v > 0) == (bits of v != 0) == sign of (0 - bits): two ops per value,
                        // v_sub_u32 and v_alignbit_b32 (shift the collected bits left by one and append that sign)
#pragma unroll
                        for (int c = 7; c >= 0; --c) bits = __builtin_amdgcn_alignbit(bits, 0u - __float_as_uint(v[2 * h + (c >> 2)][c & 3]), 31);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) bits |= (v[2 * h + (c >> 2)][c & 3] > 0.f ? 1u : 0u) << c;
                    }
                    if (STAGED && mask_lds)     // byte wn*8 + h*4 + fq of the row's 16 mask bytes, behind the bf16 image
                        reinterpret_cast<uint8_t*>(stage)[MT * 32 * 256 + (wm * (MT * 16) + j * 16 + frow) * 16 + wn * 8 + h * 4 + fq] = (uint8_t)bits;
                    else
                        reinterpret_cast<uint8_t*>(p.mask_out)[(o + 32 * h) >> 3] = (uint8_t)bits;
                }
            }
        }
        if (flags & MI_EPI_ZSPLIT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nb + 32 * (i >> 1) + 4 * (i & 1);
                if (n < p.N) {
                    const int g = n / p.zgw, nn = n - g * p.zgw;
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + ((long)g * p.M + m) * p.zgw + nn) = v[i];
                }
            }
        } else if (flags & MI_EPI_OUT_F32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nb + 32 * (i >> 1) + 4 * (i & 1);
                if (n < p.N) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (long)m * p.N + n) = v[i];
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 32 * h < p.N) {
                    bf16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (__bf16)v[2 * h + (e >> 2)][e & 3];
                    if constexpr (STATS) {      // of the value as stored: what a statistics pass over the output tensor would read
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float d = (float)hv[e] - epil[2 * h + (e >> 2)][e & 3];
                            st0[8 * h + e] += d;
                            st1[8 * h + e] += d * d;
                        }
                    }
                    if (STAGED) {
                        const int row = wm * (MT * 16) + j * 16 + frow;
                        *reinterpret_cast<bf16x8*>(stage + row * 256 + (((wn * 8 + h * 4 + fq) ^ (row & 15)) << 4)) = hv;
                    } else {
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.out) + o + 32 * h) = hv;
                    }
                }
            }
        }
    }
    if constexpr (STATS) {
        // rows: this lane's MT pixels are summed above; the 16 lanes of a DPP row hold the other pixels of the same 16 channels
#pragma unroll
        for (int k = 0; k < 16; ++k) st0[k] = igemm_row16_sum(st0[k]), st1[k] = igemm_row16_sum(st1[k]);
        if (frow == 0) {
            const long rt = (m0 + wm * (MT * 16)) / (MT * 16);
            float* dst = p.stats + rt * 2 * (long)p.N;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (nb + 32 * h < p.N) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        *reinterpret_cast<f32x4*>(dst + nb + 32 * h + 4 * q) = f32x4{st0[8 * h + 4 * q], st0[8 * h + 4 * q + 1], st0[8 * h + 4 * q + 2], st0[8 * h + 4 * q + 3]};
                        *reinterpret_cast<f32x4*>(dst + p.N + nb + 32 * h + 4 * q) = f32x4{st1[8 * h + 4 * q], st1[8 * h + 4 * q + 1], st1[8 * h + 4 * q + 2], st1[8 * h + 4 * q + 3]};
                    }
                }
            }
        }
    }
}

// ---- LDS-staged tile I/O for the 128-column block tile ---------------------------------------------------------------------
// A tile-shaped copy runs 18 % faster when a wave instruction covers whole 256-B tile rows than with the MFMA lane layout
// (4 lanes x 16 B per row, 16 rows per instruction) - tools/micro/tilecopy.hip - so the residual tile comes in and the output
// tile goes out through an LDS image with row-contiguous lanes.  The image is free LDS: the operand stages are dead once every
// wave has passed the barrier after the last K-step.
template <int MT>
__device__ __forceinline__ void igemm_residual_to_lds(const IgemmParams& p, int m0, int n0, int wave, int lane, const char* zero, char* stage) {
    // piece q = 4 rows x 256 B = one 1-KiB wave DMA; lane -> (row 4q + lane/16, physical chunk lane%16), fetching logical chunk
    // physical ^ (row & 15); wave w moves pieces w, w+4, ...
    const int prow = lane >> 4, pch = lane & 15;
#pragma unroll
    for (int i = 0; i < 2 * MT; ++i) {
        const int q = wave + 4 * i;
        const int row = 4 * q + prow, m = m0 + row;
        const int n = n0 + ((pch ^ (row & 15)) << 3);
        const char* src = (m < p.M && n < p.N) ? reinterpret_cast<const char*>(p.res + (long)m * p.N + n) : zero;
        glds16(src, stage + q * 1024);          // (glds16_nt here: 8.09 -> 8.03 ms for the class, inside the noise; not used)
    }
}

template <int MT>
__device__ __forceinline__ void igemm_residual_from_lds(const char* stage, int wm, int wn, int frow, int fq, bf16x8 (&pres)[MT][2]) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int row = wm * (MT * 16) + j * 16 + frow;
#pragma unroll
        for (int h = 0; h < 2; ++h) pres[j][h] = *reinterpret_cast<const bf16x8*>(stage + row * 256 + (((wn * 8 + h * 4 + fq) ^ (row & 15)) << 4));
    }
}

// The packed sign bits of a 128-column tile are 16 bytes per row.  From the MFMA layout they are touched 4 bytes per row per
// instruction; through the LDS image behind the bf16 tile ([BM][16 B]) one lane moves a whole row's 16 bytes.  Only for
// N % 128 == 0 (16-B aligned rows, no partial tile); otherwise the byte path stays.
template <int MT>
__device__ __forceinline__ void igemm_mask_to_lds(const IgemmParams& p, int m0, int n0, int tid, char* stage) {
    if (tid < MT * 32) {
        const int m = m0 + tid;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m < p.M) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(p.msk) + (((long)m * p.N + n0) >> 3));
        *reinterpret_cast<u32x4*>(stage + MT * 32 * 256 + tid * 16) = v;
    }
}

template <int MT>
__device__ __forceinline__ void igemm_mask_from_lds(const char* stage, int wm, int wn, int frow, int fq, unsigned (&pbits)[MT]) {
    const uint8_t* img = reinterpret_cast<const uint8_t*>(stage) + MT * 32 * 256;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int base = (wm * (MT * 16) + j * 16 + frow) * 16 + wn * 8 + fq;
        pbits[j] = (unsigned)img[base] | ((unsigned)img[base + 4] << 8);
    }
}

template <int MT>
__device__ __forceinline__ void igemm_store_mask_staged(const IgemmParams& p, int m0, int n0, int tid, const char* stage) {
    if (tid < MT * 32) {
        const int m = m0 + tid;
        if (m < p.M)
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(p.mask_out) + (((long)m * p.N + n0) >> 3)) =
                *reinterpret_cast<const u32x4*>(stage + MT * 32 * 256 + tid * 16);
    }
}

template <int MT>
__device__ __forceinline__ void igemm_store_staged(const IgemmParams& p, int m0, int n0, int tid, const char* stage) {
    const int c = tid & 15, r_in = tid >> 4;
    const int n = n0 + c * 8;
    if (n >= p.N) return;
#pragma unroll
    for (int it = 0; it < 2 * MT; ++it) {
        const int row = it * 16 + r_in, m = m0 + row;
        if (m < p.M) {
            // non-temporal: the 154-308 MB outputs of the residual launches are not read again before they have left every cache, and
            // written through the normal policy they evict the operands the neighbouring workgroups still share (K = 256 <-> N = 1024
            // class 3.70 -> 3.97 TB/s, +0.6 % images/s).  The direct-store epilogues (38-77 MB outputs, read by the very next launch)
            // keep the default policy: with nt there the next launch misses (igemm_pp +2 %, step -1 %).
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(stage + row * 256 + ((c ^ (row & 15)) << 4)),
                                        reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(p.out) + (long)m * p.N + n));
        }
    }
}

}  // namespace
