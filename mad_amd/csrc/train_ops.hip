// Small kernels of the training step: backward of the fp32 linear layers (embedding MLP, per-block
// affines), dropout with a counter-based mask, the mean-squared-error loss and its gradient, and the
// fused parameter update (gradient clean-up, AdamW, EMA).  All HBM-bound or tiny.
#include "common.h"

namespace mad {

__device__ __forceinline__ float silu_grad_t(float u) {
    const float s = 1.0f / (1.0f + __expf(-u));
    return s * fmaf(u, 1.0f - s, 1.0f);
}

// part[z][r][k] = sum over the z-th share of n of dY[r,n] * W[n,k]; grid (ceil(K/256), rows, nsplit)
__global__ __launch_bounds__(256) void linear_dx_partial_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ part, int rows, int N, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (k >= K) return;
    const int per = (N + gridDim.z - 1) / gridDim.z;
    const int n0 = blockIdx.z * per, n1 = n0 + per < N ? n0 + per : N;
    const float* g = dy + (size_t)r * N;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int n = n0;
    for (; n + 4 <= n1; n += 4) {
        s0 = fmaf(g[n], w[(size_t)n * K + k], s0);
        s1 = fmaf(g[n + 1], w[(size_t)(n + 1) * K + k], s1);
        s2 = fmaf(g[n + 2], w[(size_t)(n + 2) * K + k], s2);
        s3 = fmaf(g[n + 3], w[(size_t)(n + 3) * K + k], s3);
    }
    for (; n < n1; ++n) s0 = fmaf(g[n], w[(size_t)n * K + k], s0);
    part[((size_t)blockIdx.z * rows + r) * K + k] = (s0 + s1) + (s2 + s3);
}

// dX[i] (+)= (sum_z part[z][i]) * (act_in ? silu'(xpre[i]) : 1), shares added in a fixed order
__global__ __launch_bounds__(256) void linear_dx_finish_kernel(const float* __restrict__ part, const float* __restrict__ xpre,
                                                                float* __restrict__ dx, int n, int nsplit, int act_in,
                                                                int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * n + i];
    if (act_in) s *= silu_grad_t(xpre[i]);
    dx[i] = accumulate ? dx[i] + s : s;
}

// dW[n,k] = sum_r dY[r,n] * (act_in ? silu(xpre[r,k]) : xpre[r,k]); grid (ceil(K/256), N)
__global__ __launch_bounds__(256) void linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ xpre,
                                                         float* __restrict__ dw, int rows, int N, int K, int act_in) {
    const int k = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (k >= K) return;
    float s = 0.0f;
    for (int r = 0; r < rows; ++r) {
        const float x = xpre[(size_t)r * K + k];
        s = fmaf(dy[(size_t)r * N + n], act_in ? silu_f(x) : x, s);
    }
    dw[(size_t)n * K + k] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, size_t nvec, DropMask m) {
    constexpr int VEC = Elem<T>::VEC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        Vec16<T> v; v.load(x + i * VEC);
        float f[VEC]; v.unpack(f);
        float keep[VEC];
        drop_factors<VEC>(m, i * VEC, keep);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] *= keep[j];
        v.pack(f);
        v.store(y + i * VEC);
    }
}

// grad = 2*(pred - target)/n; partial[block] = sum of squared differences of the block's elements
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                   float* __restrict__ grad, float* __restrict__ partial, size_t n,
                                                   float gscale) {
    __shared__ float red[256];
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        grad[i] = d * gscale;
        s = fmaf(d, d, s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

struct AdamArgs {
    float* param; const float* grad; float* m; float* v; float* ema;
    size_t n;
    float lr, beta1, beta2, eps, weight_decay, bias1, bias2_sqrt, grad_scale, ema_decay;
};

// torch.nan_to_num(grad, 0, 1e5, -1e5) (main.py:249-251), AdamW with decoupled weight decay
// (torch.optim.AdamW: main.py:173-178), then the EMA shadow update (main.py:256-257)
__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        float g = a.grad[i] * a.grad_scale;
        if (g != g) g = 0.0f;
        else if (g > 3.0e38f) g = 1.0e5f;
        else if (g < -3.0e38f) g = -1.0e5f;
        float p = a.param[i];
        p *= 1.0f - a.lr * a.weight_decay;
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
        a.m[i] = m; a.v[i] = v;
        const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
        p -= (a.lr / a.bias1) * (m / denom);
        a.param[i] = p;
        if (a.ema) { const float e = a.ema[i]; a.ema[i] = e - (1.0f - a.ema_decay) * (e - p); }
    }
}

// Working copies of every convolution's weights from the fp32 masters, in one launch: the forward pack
// (Cout,k,k,Cin) and the data-gradient pack (Cin,k,k,Cout) with the taps flipped.  A workgroup moves one
// 32 co x 32 ci x k*k tile through LDS, so that the master rows (ci-major, taps minor) are read in whole
// runs and both packs are written in 32-element runs.  `table` (device, 8 longs per conv): master offset,
// forward-pack offset, gradient-pack offset (elements), Cout, Cin, k, first tile, unused.
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* master, T* fwd, T* dgrad, const long* table,
                                                                 int nconv) {
    __shared__ float tile[32][9 * 32 + 1];
    int lo = 0, hi = nconv - 1;                           // the conv whose tile range holds blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 6] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long* e = table + lo * 8;
    const int Cout = (int)e[3], Cin = (int)e[4], k = (int)e[5], kk = k * k;
    const int t_local = (int)((long)blockIdx.x - e[6]);
    const int ncit = (Cin + 31) / 32;
    const int co0 = (t_local / ncit) * 32, ci0 = (t_local % ncit) * 32;
    const int cw = Cin - ci0 < 32 ? Cin - ci0 : 32, ch = Cout - co0 < 32 ? Cout - co0 : 32;
    const float* src = master + e[0];
    const int run = cw * kk;                              // contiguous floats of one co row inside the tile
    for (int i = threadIdx.x; i < ch * run; i += 256) {
        const int co = i / run, j = i - co * run;
        tile[co][j] = src[((size_t)(co0 + co) * Cin + ci0) * kk + j];
    }
    __syncthreads();
    T* f = fwd + e[1];
    for (int i = threadIdx.x; i < ch * kk * cw; i += 256) {
        const int ci = i % cw, t = (i / cw) % kk, co = i / (cw * kk);
        f[((size_t)(co0 + co) * kk + t) * Cin + ci0 + ci] = Elem<T>::store(tile[co][ci * kk + t]);
    }
    T* g = dgrad + e[2];
    for (int i = threadIdx.x; i < cw * kk * ch; i += 256) {
        const int co = i % ch, t = (i / ch) % kk, ci = i / (ch * kk);
        g[((size_t)(ci0 + ci) * kk + (kk - 1 - t)) * Cout + co0 + co] = Elem<T>::store(tile[co][ci * kk + t]);
    }
}

}  // namespace mad

using namespace mad;

static int linear_dx_splits(int N) { const int z = N / 512; return z < 1 ? 1 : (z > 64 ? 64 : z); }

extern "C" size_t mad_linear_bwd_input_workspace(int rows, int K, int N) {
    return (size_t)linear_dx_splits(N) * rows * K * sizeof(float);
}

extern "C" int mad_linear_bwd_input(const float* dy, const float* W, const float* xpre, float* dx, void* workspace,
                                    size_t workspace_bytes, int rows, int K, int N, int act_in, int accumulate,
                                    void* stream) {
    MAD_REQUIRE(dy && W && dx && workspace && (!act_in || xpre), "mad_linear_bwd_input: null pointer");
    MAD_REQUIRE(rows > 0 && K > 0 && N > 0, "mad_linear_bwd_input: bad shape");
    MAD_REQUIRE(workspace_bytes >= mad_linear_bwd_input_workspace(rows, K, N), "mad_linear_bwd_input: workspace too small");
    const int z = linear_dx_splits(N);
    hipLaunchKernelGGL(linear_dx_partial_kernel, dim3((K + 255) / 256, rows, z), dim3(256), 0, (hipStream_t)stream, dy, W,
                       (float*)workspace, rows, N, K);
    if (int rc = check_launch("mad_linear_bwd_input")) return rc;
    hipLaunchKernelGGL(linear_dx_finish_kernel, dim3((rows * K + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, xpre, dx, rows * K, z, act_in, accumulate);
    return check_launch("mad_linear_bwd_input (finish)");
}

extern "C" int mad_linear_bwd_weight(const float* dy, const float* xpre, float* dW, int rows, int K, int N, int act_in,
                                     void* stream) {
    MAD_REQUIRE(dy && xpre && dW, "mad_linear_bwd_weight: null pointer");
    MAD_REQUIRE(rows > 0 && K > 0 && N > 0, "mad_linear_bwd_weight: bad shape");
    hipLaunchKernelGGL(linear_dw_kernel, dim3((K + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, dy, xpre, dW, rows,
                       N, K, act_in);
    return check_launch("mad_linear_bwd_weight");
}

extern "C" int mad_dropout(const void* x, void* y, int dtype, size_t n, unsigned long long seed, float p, void* stream) {
    MAD_REQUIRE(x && y, "mad_dropout: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_dropout: bad dtype %d", dtype);
    MAD_REQUIRE(p >= 0.0f && p < 1.0f, "mad_dropout: p=%f out of [0,1)", (double)p);
    const int VEC = dtype == MAD_BF16 ? 8 : 4;
    MAD_REQUIRE(n % VEC == 0, "mad_dropout: element count must be a multiple of %d", VEC);
    const size_t nvec = n / VEC;
    if (nvec == 0) return 0;
    size_t g = (nvec + 255) / 256;
    if (g > 8192) g = 8192;
    const DropMask m = make_drop_mask(seed, p);
    if (dtype == MAD_BF16)
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (bf16_t*)y, nvec, m);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (float*)y, nvec, m);
    return check_launch("mad_dropout");
}

extern "C" int mad_mse_loss_grad(const float* pred, const float* target, float* grad, float* partial, int npartial,
                                 size_t n, void* stream) {
    MAD_REQUIRE(pred && target && grad && partial, "mad_mse_loss_grad: null pointer");
    MAD_REQUIRE(n > 0 && npartial > 0 && npartial <= 4096, "mad_mse_loss_grad: bad sizes");
    hipLaunchKernelGGL(mse_kernel, dim3(npartial), dim3(256), 0, (hipStream_t)stream, pred, target, grad, partial, n,
                       2.0f / (float)n);
    return check_launch("mad_mse_loss_grad");
}

extern "C" int mad_pack_conv_weights(const float* master, void* fwd, void* dgrad, const long* table, int nconv,
                                     long total_tiles, int dtype, void* stream) {
    MAD_REQUIRE(master && fwd && dgrad && table, "mad_pack_conv_weights: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_pack_conv_weights: bad dtype %d", dtype);
    MAD_REQUIRE(nconv > 0 && total_tiles > 0 && total_tiles < (1L << 31), "mad_pack_conv_weights: bad sizes");
    if (dtype == MAD_BF16)
        hipLaunchKernelGGL(pack_conv_weights_kernel<bf16_t>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                           master, (bf16_t*)fwd, (bf16_t*)dgrad, table, nconv);
    else
        hipLaunchKernelGGL(pack_conv_weights_kernel<float>, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                           master, (float*)fwd, (float*)dgrad, table, nconv);
    return check_launch("mad_pack_conv_weights");
}

extern "C" int mad_adamw_step(float* param, const float* grad, float* m, float* v, float* ema, size_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                              float ema_decay, void* stream) {
    MAD_REQUIRE(param && grad && m && v, "mad_adamw_step: null pointer");
    MAD_REQUIRE(step >= 1, "mad_adamw_step: step counts from 1");
    if (n == 0) return 0;
    AdamArgs a;
    a.param = param; a.grad = grad; a.m = m; a.v = v; a.ema = ema; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
    a.eps = eps; a.weight_decay = weight_decay; a.grad_scale = grad_scale; a.ema_decay = ema_decay;
    a.bias1 = (float)(1.0 - pow((double)beta1, step));
    a.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("mad_adamw_step");
}
