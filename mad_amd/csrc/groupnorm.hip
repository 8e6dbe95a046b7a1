// GroupNorm for NHWC activations: statistics in two passes (partial sums, then a finalize that
// folds mean/rstd/gamma/beta and the AdaGN scale/shift into one multiply-add per channel), and an
// apply pass y = act(x*a + d) that can also average 2x2 pixels (the encoder's downsample).
// All of it is HBM-bound: 16-byte vector accesses, wave-level reductions, fp32 statistics.
#include "common.h"
#include <algorithm>

#include <mutex>

namespace mad {

// ------------------------------------------------------------------------------- partial sums
// grid (nsplit, B), 256 threads.  Chunk = 16 bytes = VEC channels of one pixel.  A thread keeps the
// same channel chunk for every pixel it visits, so its sums stay in registers.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2,
                                                          int C2, float* __restrict__ partial, int HW, int nsplit) {
    constexpr int VEC = Elem<T>::VEC;
    const int C = C1 + C2;
    const int cpr = C / VEC;                 // chunks per pixel
    const int rows = 256 / cpr > 0 ? 256 / cpr : 1;
    const int b = blockIdx.y, split = blockIdx.x;
    const int pix_per_split = HW / nsplit;
    const int p0 = split * pix_per_split;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rows][cpr][2*VEC] (cpr<=256)
    float s[VEC], q[VEC];
    // cpr may exceed 256 (C > 2048 for bf16 never happens here; for fp32 C=1536 gives 384): loop chunks
    for (int cbase = 0; cbase < cpr; cbase += 256) {
        const int t = threadIdx.x;
        const int my_rows = (cpr >= 256) ? 1 : rows;
        const int chunk = cbase + ((cpr >= 256) ? t : t % cpr);
        const int row = (cpr >= 256) ? 0 : t / cpr;
        const bool active = chunk < cpr && row < my_rows;
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = q[i] = 0.0f;
        if (active) {
            const int c = chunk * VEC;
            const T* src;
            int cs, coff;
            if (c < C1) { src = x1; cs = C1; coff = c; } else { src = x2; cs = C2; coff = c - C1; }
            src += ((size_t)b * HW) * cs + coff;
            for (int p = p0 + row; p < p0 + pix_per_split; p += my_rows) {
                Vec16<T> v;
                v.load(src + (size_t)p * cs);
                float f[VEC];
                v.unpack(f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) { s[i] += f[i]; q[i] = fmaf(f[i], f[i], q[i]); }
            }
        }
        if (cpr >= 256) {
            if (active) {
                float* dst = partial + (((size_t)b * nsplit + split) * C + chunk * VEC) * 2;
#pragma unroll
                for (int i = 0; i < VEC; ++i) { dst[2 * i] = s[i]; dst[2 * i + 1] = q[i]; }
            }
        } else {
            __syncthreads();
            if (active) {
                float* r = red + ((size_t)row * cpr + (chunk - cbase)) * 2 * VEC;
#pragma unroll
                for (int i = 0; i < VEC; ++i) { r[2 * i] = s[i]; r[2 * i + 1] = q[i]; }
            }
            __syncthreads();
            // fixed-order reduction over the rows: deterministic
            for (int j = threadIdx.x; j < cpr * 2 * VEC; j += 256) {
                float acc = 0.0f;
                for (int r = 0; r < my_rows; ++r) acc += red[(size_t)r * cpr * 2 * VEC + j];
                partial[(((size_t)b * nsplit + split) * C) * 2 + j] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------- finalize
// grid (groups, B), NWV waves per (sample, group): one wave for the short sums, four for the layers whose
// producer wrote hundreds of tile partials per channel (the 256x256 stage: 1024 pairs per group, which one
// wave would walk in four dependent round trips to memory).  The channel axis is the concatenation of two
// partial-sum slabs (C1 channels in nsplit1 splits, C2 channels in nsplit2 splits).  Partials are
// combined in fp64 in a fixed order, then the lanes write the per-channel (a, d) pairs.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void gn_finalize_kernel(const float* __restrict__ partial1, int C1, int nsplit1,
                                                          const float* __restrict__ partial2, int C2, int nsplit2,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ ada, int ada_stride,
                                                          float* __restrict__ coef, int groups, int HW, float eps,
                                                          float out_scale) {
    constexpr int NT = 64 * NWV;
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;       // lane: index in the workgroup
    const int C = C1 + C2, cpg = C / groups;
    const int max_ns = nsplit1 > nsplit2 ? nsplit1 : nsplit2;
    double s = 0.0, q = 0.0;
    // four independent loads in flight per lane (the sums are accumulated in the order of a one-at-a-time loop)
    const int total = cpg * max_ns;
    for (int it0 = lane; it0 < total; it0 += 4 * NT) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = it0 + NT * u;
            const int itc = it < total ? it : 0;
            const int c = g * cpg + itc / max_ns, sp = itc % max_ns;
            const bool first = c < C1;
            const int ns = first ? nsplit1 : nsplit2;
            const int spc = sp < ns ? sp : ns - 1;
            const float* p = first ? partial1 + (((size_t)b * nsplit1 + spc) * C1 + c) * 2
                                   : partial2 + (((size_t)b * nsplit2 + spc) * C2 + (c - C1)) * 2;
            v[u] = *reinterpret_cast<const float2*>(p);
            if (it >= total || sp >= ns) v[u] = make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    if (NWV > 1) {
        __shared__ double wsum[NWV][2];
        if ((lane & 63) == 0) { wsum[lane >> 6][0] = s; wsum[lane >> 6][1] = q; }
        __syncthreads();
        s = 0.0; q = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { s += wsum[w][0]; q += wsum[w][1]; }
    }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)eps));
    for (int i = lane; i < cpg; i += NT) {
        const int c = g * cpg + i;
        const float rg = rstd_f * gamma[c];
        float a = rg;
        float d = beta[c] - mean_f * rg;
        if (ada) {
            const float sc = ada[(size_t)b * ada_stride + c] + 1.0f;
            const float sh = ada[(size_t)b * ada_stride + C + c];
            a = a * sc;
            d = fmaf(d, sc, sh);
        }
        coef[((size_t)b * C + c) * 2] = a * out_scale;
        coef[((size_t)b * C + c) * 2 + 1] = d * out_scale;
    }
}

// ------------------------------------------------------------------------------- apply
// grid (pixel shares, B), 256 threads.  A thread keeps ONE 16-byte channel chunk for all the (output) pixels
// it visits, so its (a, d) pairs are read once into registers.
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x1, int C1, const T* __restrict__ x2,
                                                        int C2, const float* __restrict__ coef, T* __restrict__ y,
                                                        int B, int H, int W, int act, DropMask dm) {
    constexpr int VEC = Elem<T>::VEC;
    const int C = C1 + C2;
    const int cpr = C / VEC;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W, HWo = Ho * Wo;
    const int b = blockIdx.y;
    const int per = (HWo + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = p0 + per < HWo ? p0 + per : HWo;
    const int rows = cpr >= 256 ? 1 : 256 / cpr;
    for (int cbase = 0; cbase < cpr; cbase += 256) {
        const int chunk = cbase + (cpr >= 256 ? (int)threadIdx.x : (int)threadIdx.x % cpr);
        const int row = cpr >= 256 ? 0 : (int)threadIdx.x / cpr;
        if (chunk >= cpr || row >= rows) continue;
        const int c = chunk * VEC;
        const T* src;
        int cs, coff;
        if (c < C1) { src = x1; cs = C1; coff = c; } else { src = x2; cs = C2; coff = c - C1; }
        float a[VEC], d[VEC];
        const float* cf = coef + ((size_t)b * C + c) * 2;
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
            const float4 v = *reinterpret_cast<const float4*>(cf + 2 * i);
            a[i] = v.x; d[i] = v.y; a[i + 1] = v.z; d[i + 1] = v.w;
        }
        for (int p = p0 + row; p < p1; p += rows) {
            float r[VEC];
            if (!POOL) {
                Vec16<T> v;
                v.load(src + ((size_t)b * HWo + p) * cs + coff);
                float f[VEC];
                v.unpack(f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float t = fmaf(f[i], a[i], d[i]);
                    r[i] = act ? silu_f(t) : t;
                }
            } else {
                const int oy = p / Wo, ox = p - oy * Wo;
#pragma unroll
                for (int i = 0; i < VEC; ++i) r[i] = 0.0f;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const size_t ip = ((size_t)b * H + (2 * oy + dy)) * W + (2 * ox + dx);
                        Vec16<T> v;
                        v.load(src + ip * cs + coff);
                        float f[VEC];
                        v.unpack(f);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            const float t = fmaf(f[i], a[i], d[i]);
                            r[i] += act ? silu_f(t) : t;
                        }
                    }
#pragma unroll
                for (int i = 0; i < VEC; ++i) r[i] *= 0.25f;
            }
            if (dm.thr) {                                   // dropout of the output element (its linear index in y)
                const unsigned long long e0 = ((unsigned long long)b * HWo + p) * C + c;
                float keep[VEC];
                drop_factors<VEC>(dm, e0, keep);
#pragma unroll
                for (int i = 0; i < VEC; ++i) r[i] *= keep[i];
            }
            Vec16<T> o;
            o.pack(r);
            o.store(y + ((size_t)b * HWo + p) * C + c);
        }
    }
}

}  // namespace mad

using namespace mad;

extern "C" int mad_gn_partial(const void* x1, int C1, const void* x2, int C2, int dtype, float* partial, int B, int HW,
                              int nsplit, void* stream) {
    MAD_REQUIRE(x1 && partial, "mad_gn_partial: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_gn_partial: bad dtype %d", dtype);
    const int VEC = dtype == MAD_BF16 ? 8 : 4;
    MAD_REQUIRE(C1 > 0 && C1 % VEC == 0 && C2 >= 0 && C2 % VEC == 0 && (C2 == 0 || x2),
                "mad_gn_partial: channels must be multiples of %d (C1=%d C2=%d)", VEC, C1, C2);
    MAD_REQUIRE(B > 0 && HW > 0 && nsplit > 0 && HW % nsplit == 0, "mad_gn_partial: HW=%d not divisible by nsplit=%d",
                HW, nsplit);
    const int cpr = (C1 + C2) / VEC;
    const int rows = cpr >= 256 ? 1 : 256 / cpr;
    const size_t lds = cpr >= 256 ? 0 : (size_t)rows * cpr * 2 * VEC * sizeof(float);
    dim3 grid(nsplit, B);
    if (dtype == MAD_BF16)
        hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x1, C1,
                           (const bf16_t*)x2, C2, partial, HW, nsplit);
    else
        hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)x1, C1,
                           (const float*)x2, C2, partial, HW, nsplit);
    return check_launch("mad_gn_partial");
}

extern "C" int mad_gn_finalize(const float* partial1, int C1, int nsplit1, const float* partial2, int C2, int nsplit2,
                               const float* gamma, const float* beta, const float* ada, int ada_stride, float* coef,
                               int B, int groups, int HW, float eps, void* stream) {
    return mad_gn_finalize_scaled(partial1, C1, nsplit1, partial2, C2, nsplit2, gamma, beta, ada, ada_stride, coef, B,
                                  groups, HW, eps, 1.0f, stream);
}

extern "C" int mad_gn_finalize_scaled(const float* partial1, int C1, int nsplit1, const float* partial2, int C2,
                                      int nsplit2, const float* gamma, const float* beta, const float* ada,
                                      int ada_stride, float* coef, int B, int groups, int HW, float eps, float scale,
                                      void* stream) {
    MAD_REQUIRE(partial1 && gamma && beta && coef, "mad_gn_finalize: null pointer");
    MAD_REQUIRE(C1 > 0 && nsplit1 > 0 && C2 >= 0 && (C2 == 0 || (partial2 && nsplit2 > 0)),
                "mad_gn_finalize: bad partial-sum layout C1=%d nsplit1=%d C2=%d nsplit2=%d", C1, nsplit1, C2, nsplit2);
    MAD_REQUIRE(B > 0 && groups > 0 && (C1 + C2) % groups == 0, "mad_gn_finalize: bad C=%d groups=%d", C1 + C2, groups);
    MAD_REQUIRE(HW > 0, "mad_gn_finalize: bad HW");
    const int longest = ((C1 + C2) / groups) * std::max(nsplit1, C2 ? nsplit2 : 1);     // pairs per group
    if (longest >= 512)
        hipLaunchKernelGGL(gn_finalize_kernel<4>, dim3(groups, B), dim3(256), 0, (hipStream_t)stream, partial1, C1,
                           nsplit1, C2 ? partial2 : partial1, C2, C2 ? nsplit2 : 1, gamma, beta, ada, ada_stride, coef,
                           groups, HW, eps, scale);
    else
        hipLaunchKernelGGL(gn_finalize_kernel<1>, dim3(groups, B), dim3(64), 0, (hipStream_t)stream, partial1, C1,
                           nsplit1, C2 ? partial2 : partial1, C2, C2 ? nsplit2 : 1, gamma, beta, ada, ada_stride, coef,
                           groups, HW, eps, scale);
    return check_launch("mad_gn_finalize");
}

static int gn_apply_impl(const void* x1, int C1, const void* x2, int C2, const float* coef, void* y, int dtype,
                         int B, int H, int W, int act, int pool, DropMask dm, void* stream) {
    MAD_REQUIRE(x1 && coef && y, "mad_gn_apply: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_gn_apply: bad dtype %d", dtype);
    const int VEC = dtype == MAD_BF16 ? 8 : 4;
    MAD_REQUIRE(C1 > 0 && C1 % VEC == 0 && C2 >= 0 && C2 % VEC == 0 && (C2 == 0 || x2),
                "mad_gn_apply: channels must be multiples of %d (C1=%d C2=%d)", VEC, C1, C2);
    MAD_REQUIRE(B > 0 && H > 0 && W > 0 && (!pool || (H % 2 == 0 && W % 2 == 0)), "mad_gn_apply: bad shape");
    // pixel shares: about 2048 workgroups in all, at least one sweep of pixels each
    const int hwo = (pool ? H / 2 : H) * (pool ? W / 2 : W);
    const int cpr = (C1 + C2) / VEC, rows = cpr >= 256 ? 1 : 256 / cpr;      // pixels a workgroup covers per sweep
    int shares = (2048 + B - 1) / B;
    if (shares > (hwo + rows - 1) / rows) shares = (hwo + rows - 1) / rows;
    if (shares < 1) shares = 1;
    dim3 grid(shares, B);
#define LAUNCH(T, P)                                                                                              \
    hipLaunchKernelGGL((gn_apply_kernel<T, P>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x1, C1,        \
                       (const T*)x2, C2, coef, (T*)y, B, H, W, act, dm)
    if (dtype == MAD_BF16) { if (pool) LAUNCH(bf16_t, true); else LAUNCH(bf16_t, false); }
    else { if (pool) LAUNCH(float, true); else LAUNCH(float, false); }
#undef LAUNCH
    return check_launch("mad_gn_apply");
}

extern "C" int mad_gn_apply(const void* x1, int C1, const void* x2, int C2, const float* coef, void* y, int dtype,
                            int B, int H, int W, int act, int pool, void* stream) {
    return gn_apply_impl(x1, C1, x2, C2, coef, y, dtype, B, H, W, act, pool, make_drop_mask(0, 0.0f), stream);
}

extern "C" int mad_gn_apply_dropout(const void* x1, int C1, const void* x2, int C2, const float* coef, void* y,
                                    int dtype, int B, int H, int W, int act, unsigned long long seed, float p,
                                    void* stream) {
    MAD_REQUIRE(p >= 0.0f && p < 1.0f, "mad_gn_apply_dropout: p=%f out of [0,1)", (double)p);
    return gn_apply_impl(x1, C1, x2, C2, coef, y, dtype, B, H, W, act, 0, make_drop_mask(seed, p), stream);
}

// ------------------------------------------------------------------------------- no-allocation variant
// mad_gn_partial + mad_gn_finalize in one call, for callers that cannot allocate the partial-sum slab between
// the two passes.  The slab lives in a device array of this code object; every GPU the library runs on holds
// its own copy at its own address, looked up once per device.  In host code the array's NAME is only a handle
// for that lookup: kernels are given the looked-up pointer.
namespace mad {

constexpr size_t GN_STATIC_FLOATS = (size_t)1 << 21;                 // 8 MiB of partial sums per GPU
static __device__ __align__(256) float scratch[GN_STATIC_FLOATS];

constexpr int GN_STATIC_MAX_DEVICES = 64;
static float* g_scratch_ptr[GN_STATIC_MAX_DEVICES];                  // per device; null until first use there
static std::mutex g_scratch_mutex;

// the current device's copy of `scratch` (0 = ok)
static int gn_static_slab(float** out) {
    int dev = -1;
    MAD_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < GN_STATIC_MAX_DEVICES,
                "mad_gn_coef_static: cannot identify the current device");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        MAD_REQUIRE(e == hipSuccess && p, "mad_gn_coef_static: hipGetSymbolAddress failed on device %d: %s", dev,
                    hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

}  // namespace mad

extern "C" size_t mad_gn_coef_static_max_floats(void) { return mad::GN_STATIC_FLOATS; }

extern "C" int mad_gn_coef_static(const void* x1, int C1, const void* x2, int C2, int dtype, const float* gamma,
                                  const float* beta, const float* ada, int ada_stride, float* coef, int B, int HW,
                                  int groups, float eps, void* stream) {
    MAD_REQUIRE(x1 && gamma && beta && coef, "mad_gn_coef_static: null pointer");
    MAD_REQUIRE(B > 0 && HW > 0 && C1 > 0 && C2 >= 0, "mad_gn_coef_static: bad shape");
    int nsplit = HW / 64;                                             // the split count the two-call form uses
    nsplit = nsplit < 1 ? 1 : (nsplit > 64 ? 64 : nsplit);
    MAD_REQUIRE(HW % nsplit == 0, "mad_gn_coef_static: HW=%d is not a multiple of its %d splits", HW, nsplit);
    const size_t need = (size_t)B * nsplit * (C1 + C2) * 2;
    MAD_REQUIRE(need <= GN_STATIC_FLOATS,
                "mad_gn_coef_static: B=%d x %d splits x %d channels needs %zu floats of partial sums, the static slab "
                "holds %zu: use mad_gn_partial + mad_gn_finalize with a caller-owned slab",
                B, nsplit, C1 + C2, need, (size_t)GN_STATIC_FLOATS);
    float* slab = nullptr;
    if (int rc = gn_static_slab(&slab)) return rc;
    // every element the second pass reads is written by the first: nothing has to start at zero
    if (int rc = mad_gn_partial(x1, C1, x2, C2, dtype, slab, B, HW, nsplit, stream)) return rc;
    return mad_gn_finalize(slab, C1 + C2, nsplit, nullptr, 0, 0, gamma, beta, ada, ada_stride, coef, B, groups, HW, eps,
                           stream);
}
