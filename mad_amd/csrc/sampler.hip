// Per-step sampler arithmetic as single coalesced elementwise passes over NCHW fp32 tensors.
//
// Compiled with -ffp-contract=off: every multiply, add and divide below is a separately rounded
// IEEE fp32 operation in the reference's evaluation order (each torch operator there is its own
// rounded elementwise pass), so results can agree with a CPU evaluation to the last bit.
#include "common.h"

namespace mad {

struct SamplerArgs {
    mad_sampler_coef c;
    const float* model_output;
    const float* sample;
    const float* noise;
    const float* original;
    const float* mask;
    const float* next_sample;
    float* out;
    float* x0_out;
    size_t n;         // B*C*H*W
    uint32_t chw;     // C*H*W
    uint32_t hw;      // H*W
    int mask_batch;
};

__device__ __forceinline__ float sampler_one(const SamplerArgs& a, float mo, float s, float z, float orig,
                                             float m, float nxt, float& x0_ret) {
    const mad_sampler_coef& c = a.c;
    float x0, eps;
    if (c.pred_type == MAD_PRED_EPSILON) {
        x0 = (s - c.sqrt_b_t * mo) / c.sqrt_a_t;
        eps = mo;
    } else if (c.pred_type == MAD_PRED_SAMPLE) {
        x0 = mo;
        eps = (s - c.sqrt_a_t * x0) / c.sqrt_b_t;
    } else {
        x0 = c.sqrt_a_t * s - c.sqrt_b_t * mo;
        eps = c.sqrt_a_t * mo + c.sqrt_b_t * s;
    }
    if (c.clip) x0 = fminf(fmaxf(x0, -c.clip_range), c.clip_range);
    x0_ret = x0;
    if (c.op == MAD_OP_INVERSE) {
        // sqrt_a_prev / sqrt_b_prev hold the HIGHER noise level here (scheduler.py:279-282)
        return c.sqrt_a_prev * x0 + c.sqrt_b_prev * eps;
    }
    float mean;
    if (c.algo == MAD_ALGO_DDPM) {
        mean = c.c0 * x0 + c.c1 * s;
    } else {
        if (c.use_clipped) eps = (s - c.sqrt_a_t * x0) / c.sqrt_b_t;
        mean = c.sqrt_a_prev * x0 + c.dir_coef * eps;
    }
    if (c.op == MAD_OP_COMPUTE_EPS) return (nxt - mean) / c.eps_div;
    float unknown = mean;
    if (c.add_variance) unknown = mean + c.std * z;
    if (a.mask == nullptr) return unknown;
    float known = c.sqrt_a_prev * orig;
    if (c.known_noise) known = known + c.sqrt_b_prev * z;
    return m * known + (1.0f - m) * unknown;
}

__global__ __launch_bounds__(256) void sampler_kernel(SamplerArgs a) {
    const size_t n4 = a.n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool need_z = a.noise != nullptr;
    const bool need_o = a.original != nullptr && a.mask != nullptr;
    const bool need_n = a.next_sample != nullptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const size_t e = i << 2;
        float4 mo = *reinterpret_cast<const float4*>(a.model_output + e);
        float4 s = *reinterpret_cast<const float4*>(a.sample + e);
        float4 z = need_z ? *reinterpret_cast<const float4*>(a.noise + e) : make_float4(0, 0, 0, 0);
        float4 o = need_o ? *reinterpret_cast<const float4*>(a.original + e) : make_float4(0, 0, 0, 0);
        float4 nx = need_n ? *reinterpret_cast<const float4*>(a.next_sample + e) : make_float4(0, 0, 0, 0);
        float4 m = make_float4(0, 0, 0, 0);
        if (need_o) {
            // hw is a multiple of 4, so the four elements share (b, c) and are consecutive pixels
            const size_t b = e / a.chw;
            const uint32_t pix = (uint32_t)(e % a.hw);
            m = *reinterpret_cast<const float4*>(a.mask + (a.mask_batch > 1 ? b * a.hw : 0) + pix);
        }
        float4 r, x0;
        r.x = sampler_one(a, mo.x, s.x, z.x, o.x, m.x, nx.x, x0.x);
        r.y = sampler_one(a, mo.y, s.y, z.y, o.y, m.y, nx.y, x0.y);
        r.z = sampler_one(a, mo.z, s.z, z.z, o.z, m.z, nx.z, x0.z);
        r.w = sampler_one(a, mo.w, s.w, z.w, o.w, m.w, nx.w, x0.w);
        *reinterpret_cast<float4*>(a.out + e) = r;
        if (a.x0_out) *reinterpret_cast<float4*>(a.x0_out + e) = x0;
    }
}

__global__ __launch_bounds__(256) void add_noise_kernel(const float* x0, const float* noise, const float* mask,
                                                         float* out, float sa, float sb, size_t n, uint32_t chw,
                                                         uint32_t hw, int mask_batch) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const size_t e = i << 2;
        float4 x = *reinterpret_cast<const float4*>(x0 + e);
        float4 z = *reinterpret_cast<const float4*>(noise + e);
        float4 r;
        r.x = sa * x.x + sb * z.x;
        r.y = sa * x.y + sb * z.y;
        r.z = sa * x.z + sb * z.z;
        r.w = sa * x.w + sb * z.w;
        if (mask) {
            const size_t b = e / chw;
            const uint32_t pix = (uint32_t)(e % hw);
            float4 m = *reinterpret_cast<const float4*>(mask + (mask_batch > 1 ? b * hw : 0) + pix);
            r.x = m.x * r.x + (1.0f - m.x) * x.x;
            r.y = m.y * r.y + (1.0f - m.y) * x.y;
            r.z = m.z * r.z + (1.0f - m.z) * x.z;
            r.w = m.w * r.w + (1.0f - m.w) * x.w;
        }
        *reinterpret_cast<float4*>(out + e) = r;
    }
}

__global__ __launch_bounds__(256) void finalize_image_kernel(const float* x, float* out, int normalize, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = fminf(fmaxf(x[i], -1.0f), 1.0f);
        if (normalize) v = v / 2.0f + 0.5f;
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void masked_copy_kernel(float* dst, const float* src, const float* mask, size_t n,
                                                           uint32_t chw, uint32_t hw, int mask_batch) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t b = i / chw;
        const uint32_t pix = (uint32_t)(i % hw);
        if (mask[(mask_batch > 1 ? b * hw : 0) + pix] > 0.0f) dst[i] = src[i];
    }
}

// Reflect index as torch's "reflect" padding does: -i -> i, n-1+i -> n-1-i.
__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// One thread per pixel, blockIdx.y = label map of the stack.  near21[v] / near11: is there a pixel with label v
// (1 or 6) inside the reflect-padded 21x21 / 11x11 window around this pixel.
__global__ __launch_bounds__(256) void keep_mask_kernel(const uint8_t* lab, float* keep, int H, int W,
                                                         const int32_t* comps, int ncomps, int dilate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    lab += (size_t)blockIdx.y * H * W;
    keep += (size_t)blockIdx.y * H * W;
    const int y = idx / W, x = idx % W;
    bool near21_1 = false, near21_6 = false, near11 = false;
    for (int dy = -10; dy <= 10; ++dy) {
        const int yy = reflect(y + dy, H);
        for (int dx = -10; dx <= 10; ++dx) {
            const int xx = reflect(x + dx, W);
            const int v = lab[yy * W + xx];
            const bool in11 = dy >= -5 && dy <= 5 && dx >= -5 && dx <= 5;
            if (v == 1) { near21_1 = true; near11 |= in11; }
            if (v == 6) { near21_6 = true; near11 |= in11; }
        }
    }
    const int v = lab[idx];
    float edit = 0.0f;
    for (int i = 0; i < ncomps; ++i) {
        const int c = comps[i];
        if (dilate && (c == 1 || c == 6)) {
            const bool ring = (c == 1 ? near21_1 : near21_6) && v != c;
            if (ring) edit = 1.0f;
        } else if (c == 4 || c == 8) {
            if (v == c) edit = 1.0f;
            if (near11) edit = 0.0f;
        } else {
            if (v == c) edit = 1.0f;
        }
    }
    const bool basic = v == 9 || v == 13 || v == 4 || v == 8 || v == 10 || v == 2 || v == 7 || v == 1 || v == 6;
    if (!basic) edit = 0.0f;
    keep[idx] = 1.0f - edit;
}

static inline int grid_for(size_t work_items) {
    size_t g = (work_items + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace mad

using namespace mad;

extern "C" int mad_sampler_update(const mad_sampler_coef* coef, const float* model_output, const float* sample,
                                  const float* noise, const float* original, const float* mask,
                                  const float* next_sample, float* out, float* x0_out, int B, int C, int H, int W,
                                  int mask_batch, void* stream) {
    MAD_REQUIRE(coef && model_output && sample && out, "mad_sampler_update: null pointer");
    MAD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "mad_sampler_update: bad shape");
    MAD_REQUIRE((H * W) % 4 == 0, "mad_sampler_update: H*W must be a multiple of 4");
    MAD_REQUIRE(coef->op >= 0 && coef->op <= 2 && coef->pred_type >= 0 && coef->pred_type <= 2 &&
                    coef->algo >= 0 && coef->algo <= 1,
                "mad_sampler_update: bad op/pred_type/algo");
    MAD_REQUIRE(coef->op != MAD_OP_COMPUTE_EPS || next_sample, "mad_sampler_update: compute_eps needs next_sample");
    MAD_REQUIRE(!(coef->op == MAD_OP_STEP && (coef->add_variance || (mask && coef->known_noise))) || noise,
                "mad_sampler_update: step with variance needs noise");
    MAD_REQUIRE(!mask || (original && (mask_batch == 1 || mask_batch == B)),
                "mad_sampler_update: mask needs original and mask_batch in {1,B}");
    SamplerArgs a;
    a.c = *coef;
    a.model_output = model_output;
    a.sample = sample;
    a.noise = noise;
    a.original = original;
    a.mask = (coef->op == MAD_OP_STEP) ? mask : nullptr;
    a.next_sample = next_sample;
    a.out = out;
    a.x0_out = x0_out;
    a.n = (size_t)B * C * H * W;
    a.chw = (uint32_t)(C * H * W);
    a.hw = (uint32_t)(H * W);
    a.mask_batch = mask_batch;
    hipLaunchKernelGGL(sampler_kernel, dim3(grid_for(a.n / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("mad_sampler_update");
}

extern "C" int mad_add_noise(const float* x0, const float* noise, const float* mask, float* out, float sa, float sb,
                             int B, int C, int H, int W, int mask_batch, void* stream) {
    MAD_REQUIRE(x0 && noise && out, "mad_add_noise: null pointer");
    MAD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (H * W) % 4 == 0, "mad_add_noise: bad shape");
    MAD_REQUIRE(!mask || mask_batch == 1 || mask_batch == B, "mad_add_noise: mask_batch must be 1 or B");
    const size_t n = (size_t)B * C * H * W;
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x0, noise, mask,
                       out, sa, sb, n, (uint32_t)(C * H * W), (uint32_t)(H * W), mask_batch);
    return check_launch("mad_add_noise");
}

extern "C" int mad_finalize_image(const float* x, float* out, int normalize, size_t n, void* stream) {
    MAD_REQUIRE(x && out && n > 0, "mad_finalize_image: bad arguments");
    hipLaunchKernelGGL(finalize_image_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out,
                       normalize, n);
    return check_launch("mad_finalize_image");
}

extern "C" int mad_masked_copy(float* dst, const float* src, const float* mask, int B, int C, int H, int W,
                               int mask_batch, void* stream) {
    MAD_REQUIRE(dst && src && mask, "mad_masked_copy: null pointer");
    MAD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (mask_batch == 1 || mask_batch == B), "mad_masked_copy: bad shape");
    const size_t n = (size_t)B * C * H * W;
    hipLaunchKernelGGL(masked_copy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, mask, n,
                       (uint32_t)(C * H * W), (uint32_t)(H * W), mask_batch);
    return check_launch("mad_masked_copy");
}

extern "C" int mad_keep_mask_batch(const uint8_t* label_maps, float* keep, int B, int H, int W, const int32_t* comps,
                                   int ncomps, int dilate, void* stream) {
    MAD_REQUIRE(label_maps && keep && comps, "mad_keep_mask: null pointer");
    MAD_REQUIRE(B > 0 && B <= 65535, "mad_keep_mask: B=%d label maps (1..65535)", B);
    MAD_REQUIRE(H > 10 && W > 10 && ncomps >= 0 && ncomps <= 16, "mad_keep_mask: needs H,W > 10 and <= 16 components");
    hipLaunchKernelGGL(keep_mask_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, label_maps,
                       keep, H, W, comps, ncomps, dilate);
    return check_launch("mad_keep_mask");
}

extern "C" int mad_keep_mask(const uint8_t* label_map, float* keep, int H, int W, const int32_t* comps, int ncomps,
                             int dilate, void* stream) {
    return mad_keep_mask_batch(label_map, keep, 1, H, W, comps, ncomps, dilate, stream);
}
