// Timestep/class embedding path: sinusoidal embedding and small fp32 linear layers.
// These are a few MFLOP per evaluation; they stay in fp32 on the vector ALU.
#include "common.h"

namespace mad {

__global__ __launch_bounds__(256) void pos_embedding_kernel(const float* t, int t_count, const float* freqs,
                                                             float* emb, int rows, int half) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * half) return;
    const int b = idx / half, i = idx % half;
    const float ang = t[t_count > 1 ? b : 0] * freqs[i];
    emb[(size_t)b * 2 * half + i] = cosf(ang);
    emb[(size_t)b * 2 * half + half + i] = sinf(ang);
}

// One wave per output feature n: the weight row is read once (coalesced), every input row is
// dotted against it.  ROWS_PER_PASS input rows are processed together to reuse the weight row
// held in registers.
template <int ROWS_PER_PASS>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int rows, int K, int N, int act_in, int act_out,
                                                      int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (size_t)n * K;
    for (int r0 = 0; r0 < rows; r0 += ROWS_PER_PASS) {
        float acc[ROWS_PER_PASS];
#pragma unroll
        for (int r = 0; r < ROWS_PER_PASS; ++r) acc[r] = 0.0f;
        for (int k = lane; k < K; k += 64) {
            const float wv = w[k];
#pragma unroll
            for (int r = 0; r < ROWS_PER_PASS; ++r) {
                if (r0 + r < rows) {
                    float xv = x[(size_t)(r0 + r) * K + k];
                    if (act_in) xv = silu_f(xv);
                    acc[r] = fmaf(xv, wv, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_PASS; ++r) {
            float v = acc[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && r0 + r < rows) {
                if (bias) v += bias[n];
                float* dst = y + (size_t)(r0 + r) * N + n;
                if (accumulate) v += *dst;
                if (act_out) v = silu_f(v);
                *dst = v;
            }
        }
    }
}

}  // namespace mad

using namespace mad;

extern "C" int mad_pos_embedding(const float* t, int t_count, const float* freqs, float* emb, int rows, int half,
                                 void* stream) {
    MAD_REQUIRE(t && freqs && emb, "mad_pos_embedding: null pointer");
    MAD_REQUIRE(rows > 0 && half > 0 && (t_count == 1 || t_count == rows),
                "mad_pos_embedding: t_count must be 1 or rows (%d), got %d", rows, t_count);
    const int n = rows * half;
    hipLaunchKernelGGL(pos_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, t_count,
                       freqs, emb, rows, half);
    return check_launch("mad_pos_embedding");
}

extern "C" int mad_linear_f32(const float* x, const float* W, const float* bias, float* y, int rows, int K, int N,
                              int act_in, int act_out, int accumulate, void* stream) {
    MAD_REQUIRE(x && W && y, "mad_linear_f32: null pointer");
    MAD_REQUIRE(rows > 0 && K > 0 && N > 0, "mad_linear_f32: bad shape rows=%d K=%d N=%d", rows, K, N);
    hipLaunchKernelGGL(linear_kernel<8>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, bias, y, rows,
                       K, N, act_in, act_out, accumulate);
    return check_launch("mad_linear_f32");
}
