// Micro-benchmark: copy a [M][N] bf16 matrix where each workgroup moves one BM x BNc tile (row segments of BNc*2 bytes at a
// stride of N*2 bytes), in the lane pattern of the igemm epilogue (16 rows x 64 B per wave instruction).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Same tile per workgroup, but every wave instruction covers 4 full tile rows (4 x BNc*2 = 256 B contiguous for BNc = 128): the lane
// pattern an LDS-transposed epilogue would produce.
template <bool READ, bool WRITE>
__global__ __launch_bounds__(256) void tilecopy_rows(const char* src, char* dst, int M, int N, int BM, int BNc, int n_tiles) {
    const int tile = blockIdx.x;
    const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
    const int lanes_per_row = BNc / 8;                       // 16-B chunks per tile row
    const int rows_per_pass = 256 / lanes_per_row;
    const int r_in = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 buf[8];
    const int passes = BM / rows_per_pass;
    for (int p0 = 0; p0 < passes; p0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mt * BM + (p0 + q) * rows_per_pass + r_in;
            buf[q] = u32x4{1, 2, 3, 4};
            if (READ && p0 + q < passes && m < M) buf[q] = *reinterpret_cast<const u32x4*>(src + ((long)m * N + nt * BNc + c * 8) * 2);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = mt * BM + (p0 + q) * rows_per_pass + r_in;
            if (p0 + q < passes && m < M) {
                if (WRITE) *reinterpret_cast<u32x4*>(dst + ((long)m * N + nt * BNc + c * 8) * 2) = buf[q];
                else acc += buf[q];
            }
        }
    }
    if (!WRITE && acc[0] == 0x12345678) dst[0] = 1;
}

template <bool READ, bool WRITE>
__global__ __launch_bounds__(256) void tilecopy(const char* src, char* dst, int M, int N, int BM, int BNc, int n_tiles, int remap) {
    int tile = blockIdx.x;
    if (remap) {   // XCD-aware: consecutive tiles on one XCD
        const int nwg = gridDim.x, per = (nwg + 7) / 8;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = xcd * per + idx;
        if (tile >= nwg) return;
    }
    const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int rows_per_wave = BM / 2;
    u32x4 acc = {0, 0, 0, 0};
    // loads of the whole tile first, then the stores (as the conv epilogue does: batched residual loads, then the row loop)
    u32x4 buf[6][4];
    const int nj = rows_per_wave / 16, nh = BNc / 64;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int m = mt * BM + wm * rows_per_wave + j * 16 + frow;
            buf[j][h] = u32x4{1, 2, 3, 4};
            if (READ && j < nj && h < nh && m < M)
                buf[j][h] = *reinterpret_cast<const u32x4*>(src + ((long)m * N + nt * BNc + wn * (BNc / 2) + h * 32 + 8 * fq) * 2);
        }
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int m = mt * BM + wm * rows_per_wave + j * 16 + frow;
            if (j < nj && h < nh && m < M) {
                if (WRITE) *reinterpret_cast<u32x4*>(dst + ((long)m * N + nt * BNc + wn * (BNc / 2) + h * 32 + 8 * fq) * 2) = buf[j][h];
                else acc += buf[j][h];
            }
        }
    if (!WRITE && acc[0] == 0x12345678) dst[0] = 1;
}

extern "C" void run_tilecopy(const void* src, void* dst, int M, int N, int BM, int BNc, int mode, int remap, void* stream) {
    const int m_tiles = (M + BM - 1) / BM, n_tiles = N / BNc;
    dim3 grid(m_tiles * n_tiles);
    if (mode == 0) hipLaunchKernelGGL((tilecopy<true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, M, N, BM, BNc, n_tiles, remap);
    if (mode == 1) hipLaunchKernelGGL((tilecopy<true, false>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, M, N, BM, BNc, n_tiles, remap);
    if (mode == 2) hipLaunchKernelGGL((tilecopy<false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, M, N, BM, BNc, n_tiles, remap);
    if (mode == 10) hipLaunchKernelGGL((tilecopy_rows<true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, M, N, BM, BNc, n_tiles);
    if (mode == 11) hipLaunchKernelGGL((tilecopy_rows<true, false>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, M, N, BM, BNc, n_tiles);
    if (mode == 12) hipLaunchKernelGGL((tilecopy_rows<false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, M, N, BM, BNc, n_tiles);
}
