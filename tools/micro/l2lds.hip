// Microbenchmark: how fast does one CU pull an operand tile from L2 / Infinity Cache / HBM into LDS, as a function of the
// contiguous segment per row (the k-slab width of an implicit GEMM: 64 B = 32 bf16 channels, 128 B = 64, 256 B = 128) and of the
// path: LDS-DMA (global_load_lds_dwordx4), global_load_dwordx4 -> VGPR -> ds_write_b128, or global_load only.
// Each workgroup sweeps its own region: blocks of R rows x `stride` bytes, k-slab by k-slab, 16 KB per stage, three stages in flight.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o l2lds.so l2lds.hip ; run: python tools/micro/run_l2lds.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 512, U = 2, STAGE = NT * 16 * U, SLOTS = 4;

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

struct Sweep {
    const char* base;
    size_t block_bytes;
    int stride, seg, blocks, slabs, rows_per_instr, my_off, blk, slab;
    __device__ Sweep(const char* src, size_t wg_region, int stride_, int seg_, int blocks_) {
        stride = stride_, seg = seg_, blocks = blocks_;
        const int lpr = seg / 16;
        rows_per_instr = NT / lpr;
        slabs = stride / seg;
        block_bytes = (size_t)rows_per_instr * U * stride;
        base = src + (size_t)blockIdx.x * wg_region;
        my_off = (threadIdx.x / lpr) * stride + (threadIdx.x % lpr) * 16;
        blk = slab = 0;
    }
    __device__ const char* addr(int u) const { return base + blk * block_bytes + (size_t)u * rows_per_instr * stride + slab * seg + my_off; }
    __device__ void advance() {
        if (++slab == slabs) {
            slab = 0;
            if (++blk == blocks) blk = 0;
        }
    }
};

template <int MODE, bool BARRIER>
__global__ __launch_bounds__(NT) void l2lds_kernel(const char* src, size_t wg_region, int stride, int seg, int blocks, int iters, unsigned* sink) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    Sweep sw(src, wg_region, stride, seg, blocks);
    unsigned acc = 0;
    if (MODE == 0) {
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int u = 0; u < U; ++u) glds16(sw.addr(u), lds + s * STAGE + (u * NT + wave * 64) * 16);
            sw.advance();
        }
        for (int it = 0; it < iters; ++it) {
            char* slot = lds + ((it + 3) & (SLOTS - 1)) * STAGE;
#pragma unroll
            for (int u = 0; u < U; ++u) glds16(sw.addr(u), slot + (u * NT + wave * 64) * 16);
            sw.advance();
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (BARRIER) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = *(unsigned*)(lds + tid * 4);
    } else {
        u32x4 r[3][U];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int u = 0; u < U; ++u) r[s][u] = *(const u32x4*)sw.addr(u);
            sw.advance();
        }
        for (int it = 0; it < iters; it += 3) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                char* slot = lds + ((it + s) & (SLOTS - 1)) * STAGE;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (MODE == 1)
                        *(u32x4*)(slot + (u * NT + tid) * 16) = r[s][u];
                    else
                        acc ^= r[s][u].x ^ r[s][u].y ^ r[s][u].z ^ r[s][u].w;
                    r[s][u] = *(const u32x4*)sw.addr(u);
                }
                sw.advance();
                if (BARRIER) __builtin_amdgcn_s_barrier();
            }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= r[s][u].x;
        if (MODE == 1) {
            __syncthreads();
            acc ^= *(unsigned*)(lds + tid * 4);
        }
    }
    if (acc == 0x12345679u) sink[blockIdx.x] = acc;
}

// Clock probe: the shader clock (s_memtime ticks per s_memrealtime 100 MHz tick) while every CU runs back-to-back MFMAs on
// `waves` waves per SIMD (mfma = 1) or an LDS-DMA stream from L2 (mfma = 0), for `iters` iterations.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) void clock_probe_kernel(const char* src, int mfma, int iters, unsigned* out) {
    extern __shared__ char lds[];
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[8], b;
    {   // operands: pseudo-random bf16 bit patterns in [-2, 2) (eight different A operands: no common subexpressions)
        unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 8; ++i) {
                h = h * 1664525u + 1013904223u;
                a[j][i] = (__bf16)(((float)(h >> 8) / 8388608.0f - 1.0f) * 2.0f);
            }
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            b[i] = (__bf16)(((float)(h >> 8) / 8388608.0f - 1.0f) * 0.01f);
        }
    }
    if (mfma == 2) {
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        f32x16 c[4];
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 16; ++k) c[i][k] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, c[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) acc[i][0] = c[i][0] + c[i][7];
    } else if (mfma) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i], 0, 0, 0);
    } else {
        const char* base = src + (size_t)blockIdx.x * 65536 + threadIdx.x * 16;
        for (int it = 0; it < iters; ++it) {
            glds16(base + (it & 3) * 8192, lds + (it & 3) * 8192 + (threadIdx.x >> 6) * 1024);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += acc[i][0];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2 + 0] = (unsigned)(c1 - c0);
        out[blockIdx.x * 2 + 1] = (unsigned)(r1 - r0);
    }
    if (sum == 1.2345f) out[0] = 0;
}

extern "C" int run_clock_probe(const void* src, int mfma, int iters, int grid, int threads, void* out, void* stream) {
    hipFuncSetAttribute((const void*)clock_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(grid), dim3(threads), 65536, (hipStream_t)stream, (const char*)src, mfma, iters, (unsigned*)out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int run_l2lds(const void* src, size_t src_bytes, int mode, int barrier, int stride, int seg, int blocks, int iters, int grid, void* sink, void* stream) {
    const int lpr = seg / 16;
    if (seg % 16 || stride % seg || NT % lpr || iters % 3) return -1;
    const size_t wg_region = (size_t)(NT / lpr) * U * stride * blocks;
    if (wg_region * grid > src_bytes) return -2;
    const int lds = SLOTS * STAGE;
#define GO(M, B)                                                                                                      \
    {                                                                                                                 \
        hipFuncSetAttribute((const void*)l2lds_kernel<M, B>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);        \
        hipLaunchKernelGGL((l2lds_kernel<M, B>), dim3(grid), dim3(NT), lds, (hipStream_t)stream, (const char*)src, wg_region, stride, seg, blocks, iters, \
                           (unsigned*)sink);                                                                          \
    }
    if (mode == 0 && barrier) GO(0, true)
    else if (mode == 0) GO(0, false)
    else if (mode == 1 && barrier) GO(1, true)
    else if (mode == 1) GO(1, false)
    else if (barrier) GO(2, true)
    else GO(2, false)
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
