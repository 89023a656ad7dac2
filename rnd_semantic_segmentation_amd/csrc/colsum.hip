// Column sums over the pixel dimension (conv bias gradients), deterministic and wide:
//   mi_aspp_bias_grad   dbias4[r][n] (+)= sum_m dlow[m][n]        fp32 [M][K], replicated to the 4 ASPP branches
//   mi_bias_grad_bf16   db[n]        (+)= sum_m dy[m][n]          bf16 [M][N]
// Two levels with a fixed order: up to 512 workgroups each sum a contiguous range of rows (row lanes combined in lane
// order), then one thread per column adds the workgroup partials in ascending order.  No atomics -> bitwise reproducible.
// Replaces the bias half of convolution_backward for reference core/models/classifiers/aspp/classifier.py:12-20 and
// core/models/discriminator.py:34-50.
#include "mi_common.h"

namespace {

constexpr int MAX_BLOCKS = 512;

// VEC = 1: fp32 columns; VEC = 8: bf16x8 column groups.  CP = slots per row rounded up to a power of two (<= 256).
template <int VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ src, float* __restrict__ partial, long M, int slots,
                                                             int cp, long rows_per_block) {
    __shared__ float red[256 * VEC];
    const int slot = threadIdx.x % cp, rl = threadIdx.x / cp, lanes = 256 / cp;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (slot < slots) {
        for (long r = r0 + rl; r < r1; r += lanes) {
            if (VEC == 1) {
                acc[0] += reinterpret_cast<const float*>(src)[r * slots + slot];
            } else {
                const bf16x8 v = reinterpret_cast<const bf16x8*>(src)[r * slots + slot];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += (float)v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x * VEC + e] = acc[e];
    __syncthreads();
    if (rl == 0 && slot < slots) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float s = 0.f;
            for (int q = 0; q < lanes; ++q) s += red[(q * cp + slot) * VEC + e];
            partial[((long)blockIdx.x * slots + slot) * VEC + e] = s;
        }
    }
}

// 8 columns per workgroup, 32 lanes per column: lane q adds partials q, q+32, ... in ascending order, lane 0 then adds the 32 lane
// sums in ascending order (fixed association, independent of timing)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblocks, int N, int rep,
                                                           int accumulate) {
    __shared__ float red[32][8];
    const int c = threadIdx.x & 7, q = threadIdx.x >> 3;
    const int n = blockIdx.x * 8 + c;
    float s = 0.f;
    if (n < N)
        for (int b = q; b < nblocks; b += 32) s += partial[(long)b * N + n];
    red[q][c] = s;
    __syncthreads();
    if (q == 0 && n < N) {
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += red[i][c];
        for (int r = 0; r < rep; ++r) out[r * N + n] = accumulate ? out[r * N + n] + t : t;
    }
}

int launch(const void* src, float* out, long M, int N, int vec, int rep, int accumulate, void* workspace, size_t ws_bytes, void* stream,
           const char* who) {
    const int slots = N / vec;
    int cp = 1;
    while (cp < slots) cp <<= 1;
    if (cp > 256) return mi_set_error(MI_EINVAL, "%s: N=%d is too wide (at most %d columns)", who, N, 256 * vec);
    const int lanes = 256 / cp;
    long rows = (M + MAX_BLOCKS - 1) / MAX_BLOCKS;
    if (rows < 8L * lanes) rows = 8L * lanes;
    const int nb = (int)((M + rows - 1) / rows);
    if (ws_bytes < (size_t)nb * N * sizeof(float)) return mi_set_error(MI_ENOMEM, "%s: workspace too small", who);
    float* partial = (float*)workspace;
    if (vec == 1)
        hipLaunchKernelGGL(colsum_partial_kernel<1>, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, partial, M, slots, cp, rows);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<8>, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, partial, M, slots, cp, rows);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const float*)partial, out, nb, N, rep,
                       accumulate);
    return MI_OK;
}

}  // namespace

extern "C" size_t mi_colsum_workspace(int M, int N) {
    (void)M;
    return (size_t)MAX_BLOCKS * (size_t)(N > 0 ? N : 1) * sizeof(float);
}

extern "C" int mi_aspp_bias_grad(const float* dlow, float* dbias4, int M, int K, int accumulate, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    MI_REQUIRE(dlow && dbias4 && workspace && M > 0 && K > 0, "mi_aspp_bias_grad: bad argument");
    const int rc = launch(dlow, dbias4, M, K, 1, 4, accumulate, workspace, workspace_bytes, stream, "mi_aspp_bias_grad");
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH("mi_aspp_bias_grad");
    return MI_OK;
}

extern "C" int mi_bias_grad_bf16(const void* dy, float* db, int M, int N, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MI_REQUIRE(dy && db && workspace && M > 0 && N > 0 && N % 8 == 0 && mi_aligned16(dy), "mi_bias_grad_bf16: bad argument (N %% 8 == 0)");
    const int rc = launch(dy, db, M, N, 8, 1, accumulate, workspace, workspace_bytes, stream, "mi_bias_grad_bf16");
    if (rc != MI_OK) return rc;
    MI_CHECK_LAUNCH("mi_bias_grad_bf16");
    return MI_OK;
}
