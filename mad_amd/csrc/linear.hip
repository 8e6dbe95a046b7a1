// Timestep/class embedding path: sinusoidal embedding and small fp32 linear layers.
// These are a few MFLOP per evaluation; they stay in fp32 on the vector ALU.
#include "common.h"
#include <algorithm>

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

// y = act_out(x W^T + bias [+ y]).  A block stages up to RT input rows (activated once) in LDS and
// each wave then produces FT output features for all of them: a weight row is read once per row
// tile, coalesced, and every staged row is dotted against it from LDS (lane k-strided, conflict-free).
// The block walks feature groups with a grid stride so the staging cost is shared.
constexpr int LIN_FT = 2;

template <int RT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int rows, int K, int N, int act_in, int act_out,
                                                      int accumulate) {
    extern __shared__ float xs[];            // RT x K
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int r0 = 0; r0 < rows; r0 += RT) {
        const int nr = min(RT, rows - r0);
        __syncthreads();
        // one row per wave at a time, four rows in flight: independent coalesced loads, no division
        for (int r = wave; r < RT; r += 4 * nwave) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * nwave;
                if (rr < RT) {
                    for (int k = lane; k < K; k += 64) {
                        float v = 0.0f;
                        if (rr < nr) {
                            v = x[(size_t)(r0 + rr) * K + k];
                            if (act_in) v = silu_f(v);
                        }
                        xs[rr * K + k] = v;
                    }
                }
            }
        }
        __syncthreads();
        for (int g = blockIdx.x * nwave + wave; g * LIN_FT < N; g += gridDim.x * nwave) {
            const int n0 = g * LIN_FT;
            float acc[LIN_FT][RT];
#pragma unroll
            for (int f = 0; f < LIN_FT; ++f)
#pragma unroll
                for (int r = 0; r < RT; ++r) acc[f][r] = 0.0f;
            for (int k = lane; k < K; k += 64) {
                float wv[LIN_FT];
#pragma unroll
                for (int f = 0; f < LIN_FT; ++f) wv[f] = n0 + f < N ? W[(size_t)(n0 + f) * K + k] : 0.0f;
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const float xv = xs[r * K + k];
#pragma unroll
                    for (int f = 0; f < LIN_FT; ++f) acc[f][r] = fmaf(xv, wv[f], acc[f][r]);
                }
            }
            // Transposing butterfly: at step `mask` a lane keeps the half of its values whose index has
            // that bit equal to the lane's and adds the partner's copy of them, so after log2(64) steps
            // lane l holds the complete sum of value l = f * RT + r (64 values when RT = 32).
            constexpr int NV = LIN_FT * RT;
            static_assert(NV <= 64 && (NV & (NV - 1)) == 0, "values per wave must be a power of two <= 64");
            float vals[NV];
#pragma unroll
            for (int f = 0; f < LIN_FT; ++f)
#pragma unroll
                for (int r = 0; r < RT; ++r) vals[f * RT + r] = acc[f][r];
#pragma unroll
            for (int mask = 32; mask >= NV; mask >>= 1)          // fewer values than lanes: plain halving first
#pragma unroll
                for (int j = 0; j < NV; ++j) vals[j] += __shfl_xor(vals[j], mask);
#pragma unroll
            for (int cnt = NV / 2; cnt >= 1; cnt >>= 1) {
                const bool up = (lane & cnt) != 0;
#pragma unroll
                for (int j = 0; j < cnt; ++j) {
                    const float keep = up ? vals[j + cnt] : vals[j];
                    const float send = up ? vals[j] : vals[j + cnt];
                    vals[j] = keep + __shfl_xor(send, cnt);
                }
            }
            const int vi = lane & (NV - 1), f = vi / RT, r = vi % RT;
            if (lane < NV && r < nr && n0 + f < N) {
                float v = vals[0];
                if (bias) v += bias[n0 + f];
                float* dst = y + (size_t)(r0 + r) * N + n0 + f;
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
    MAD_REQUIRE(K <= 4096, "mad_linear_f32: K=%d exceeds the staged-row limit 4096", K);
    const int groups = (N + LIN_FT - 1) / LIN_FT;
    const int grid = std::max(1, std::min((groups + 15) / 16, 512));   // ~4 feature groups per wave
    if (rows <= 4) {
        // one timestep and one label for the whole batch (the sampling loops): a single row, broadcast by the caller
        hipLaunchKernelGGL(linear_kernel<4>, dim3(grid), dim3(256), (size_t)4 * K * sizeof(float),
                           (hipStream_t)stream, x, W, bias, y, rows, K, N, act_in, act_out, accumulate);
    } else if ((size_t)32 * K * sizeof(float) <= 128 * 1024) {
        const size_t lds = (size_t)32 * K * sizeof(float);
        static int lds_dev = -1;
        MAD_REQUIRE(raise_lds_limit((const void*)linear_kernel<32>, 128 * 1024, &lds_dev),
                    "mad_linear_f32: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL(linear_kernel<32>, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, W, bias, y, rows, K,
                           N, act_in, act_out, accumulate);
    } else {
        hipLaunchKernelGGL(linear_kernel<4>, dim3(grid), dim3(256), (size_t)4 * K * sizeof(float),
                           (hipStream_t)stream, x, W, bias, y, rows, K, N, act_in, act_out, accumulate);
    }
    return check_launch("mad_linear_f32");
}
