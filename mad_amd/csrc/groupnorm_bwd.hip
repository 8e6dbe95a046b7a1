// Backward of y = act(GroupNorm(x) * (1 + scale) + shift) for NHWC activations (training step).
//
// Forward keeps one (a, d) pair per (sample, channel): u = x*a + d, y = act(u).  With g = dy * act'(u):
//   dshift[b,c] = S1 = sum_pix g          dscale[b,c] = gamma*Gx + beta*S1      Gx = rstd*(S2 - mean*S1)
//   dbeta[c]    = sum_b (1+scale)*S1      dgamma[c]   = sum_b (1+scale)*Gx      S2 = sum_pix g*x
//   dx = g*p + x*q + r,  p = rstd*(1+scale)*gamma (per channel),  q = -rstd^2*Bg/N,  r = -rstd*Ag/N - mean*q
//   Ag = sum_{c in group} (1+scale)*gamma*S1,  Bg = sum_{c in group} (1+scale)*gamma*Gx,  N = HW * channels/group.
// Three HBM-bound passes mirror the forward's: partial sums (S1, S2) per (sample, split, channel), a
// finalize on the small tensors, and an elementwise apply that also adds up to two other gradients
// flowing into x (the residual branch, a skip connection), each possibly at another resolution.
#include "common.h"

namespace mad {

// How a gradient tensor relates to the resolution (H, W) of x.
//   0: same resolution
//   1: tensor is (H/2, W/2): every pixel receives a quarter of its 2x2 cell's value (backward of a 2x2 mean)
//   2: tensor is (2H, 2W): every pixel receives the sum of its 2x2 block (backward of a nearest x2)
template <typename T>
__device__ __forceinline__ void load_grad(const T* t, int mode, int b, int y, int x, int H, int W, int C, int c,
                                          float* out) {
    constexpr int VEC = Elem<T>::VEC;
    if (mode == 0) {
        Vec16<T> v; v.load(t + (((size_t)b * H + y) * W + x) * C + c);
        v.unpack(out);
    } else if (mode == 1) {
        Vec16<T> v; v.load(t + (((size_t)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C + c);
        v.unpack(out);
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] *= 0.25f;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                Vec16<T> v; v.load(t + (((size_t)b * (2 * H) + 2 * y + dy) * (2 * W) + 2 * x + dx) * C + c);
                float f[VEC]; v.unpack(f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) out[i] += f[i];
            }
    }
}

struct GnBwdArgs {
    const void* x1; const void* x2; const void* dy; const float* coef;
    int C1, C2, B, H, W, act, dy_mode;
    DropMask dm;               // dropout that was applied to the forward's output (dy_mode 0 only); thr == 0: none
};

// ------------------------------------------------------------------------------- partial sums
// grid (nsplit, B), 256 threads; the layout of gn_partial_kernel: a thread keeps one 16-byte channel chunk.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(GnBwdArgs a, float* __restrict__ partial, int nsplit) {
    constexpr int VEC = Elem<T>::VEC;
    const int C = a.C1 + a.C2, HW = a.H * a.W;
    const int cpr = C / VEC;
    const int rows = 256 / cpr > 0 ? 256 / cpr : 1;
    const int b = blockIdx.y, split = blockIdx.x;
    const int pix_per_split = HW / nsplit;
    const int p0 = split * pix_per_split;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rows][cpr][2*VEC]
    for (int cbase = 0; cbase < cpr; cbase += 256) {
        const int t = threadIdx.x;
        const int my_rows = (cpr >= 256) ? 1 : rows;
        const int chunk = cbase + ((cpr >= 256) ? t : t % cpr);
        const int row = (cpr >= 256) ? 0 : t / cpr;
        const bool active = chunk < cpr && row < my_rows;
        float s1[VEC], s2[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.0f;
        if (active) {
            const int c = chunk * VEC;
            const T* src; int cs, coff;
            if (c < a.C1) { src = (const T*)a.x1; cs = a.C1; coff = c; } else { src = (const T*)a.x2; cs = a.C2; coff = c - a.C1; }
            src += ((size_t)b * HW) * cs + coff;
            float ca[VEC], cd[VEC];
            const float* cf = a.coef + ((size_t)b * C + c) * 2;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { ca[i] = cf[2 * i]; cd[i] = cf[2 * i + 1]; }
            for (int p = p0 + row; p < p0 + pix_per_split; p += my_rows) {
                Vec16<T> v; v.load(src + (size_t)p * cs);
                float f[VEC], g[VEC]; v.unpack(f);
                load_grad<T>((const T*)a.dy, a.dy_mode, b, p / a.W, p % a.W, a.H, a.W, C, c, g);
                if (a.dm.thr) {
                    const unsigned long long e0 = ((unsigned long long)b * HW + p) * C + c;
                    float keep[VEC];
                    drop_factors<VEC>(a.dm, e0, keep);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) g[i] *= keep[i];
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float gi = a.act ? g[i] * silu_grad_f(fmaf(f[i], ca[i], cd[i])) : g[i];
                    s1[i] += gi; s2[i] = fmaf(gi, f[i], s2[i]);
                }
            }
        }
        if (cpr >= 256) {
            if (active) {
                float* dst = partial + (((size_t)b * nsplit + split) * C + chunk * VEC) * 2;
#pragma unroll
                for (int i = 0; i < VEC; ++i) { dst[2 * i] = s1[i]; dst[2 * i + 1] = s2[i]; }
            }
        } else {
            __syncthreads();
            if (active) {
                float* r = red + ((size_t)row * cpr + (chunk - cbase)) * 2 * VEC;
#pragma unroll
                for (int i = 0; i < VEC; ++i) { r[2 * i] = s1[i]; r[2 * i + 1] = s2[i]; }
            }
            __syncthreads();
            for (int j = threadIdx.x; j < cpr * 2 * VEC; j += 256) {
                float acc = 0.0f;
                for (int r = 0; r < my_rows; ++r) acc += red[(size_t)r * cpr * 2 * VEC + j];
                partial[(((size_t)b * nsplit + split) * C) * 2 + j] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------- finalize
// grid (groups, B), one wave per (sample, group): mean / rstd from the forward's statistics partials (the
// same fp64 combination as the forward), then the gradient sums.
struct GnBwdFinArgs {
    const float* stat1; const float* stat2; const float* gsum; const float* gamma; const float* beta; const float* ada;
    float* bcoef; float* dgb; float* dada;
    int C1, ns1, C2, ns2, nsg, ada_stride, groups, HW;
    float eps;
};

__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(GnBwdFinArgs a) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int C = a.C1 + a.C2, cpg = C / a.groups;
    const int max_ns = a.ns1 > a.ns2 ? a.ns1 : a.ns2;
    double s = 0.0, q = 0.0;
    // four independent loads in flight per lane (the sums are accumulated in the order of a one-at-a-time loop)
    const int total = cpg * max_ns;
    for (int it0 = lane; it0 < total; it0 += 256) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = it0 + 64 * u;
            const int itc = it < total ? it : 0;
            const int c = g * cpg + itc / max_ns, sp = itc % max_ns;
            const bool first = c < a.C1;
            const int ns = first ? a.ns1 : a.ns2;
            const int spc = sp < ns ? sp : ns - 1;
            const float* p = first ? a.stat1 + (((size_t)b * a.ns1 + spc) * a.C1 + c) * 2
                                   : a.stat2 + (((size_t)b * a.ns2 + spc) * a.C2 + (c - a.C1)) * 2;
            v[u] = *reinterpret_cast<const float2*>(p);
            if (it >= total || sp >= ns) v[u] = make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    const double n = (double)a.HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)a.eps));

    // gradient sums per channel: the lanes share the rows of partial sums of one channel at a time (a row per
    // 16x16 tile when the data-gradient conv produced them), butterfly-reduced so that every lane holds the totals
    double Ag = 0.0, Bg = 0.0;
    constexpr int CB = 4;                      // channels whose rows are requested together
    for (int i0 = 0; i0 < cpg; i0 += CB) {
        double S1[CB], S2[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) S1[k] = S2[k] = 0.0;
        for (int sp0 = lane; sp0 < a.nsg; sp0 += 256) {
            float2 v[CB][4];
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const int c = g * cpg + (i0 + k < cpg ? i0 + k : i0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int sp = sp0 + 64 * u;
                    v[k][u] = *reinterpret_cast<const float2*>(a.gsum + (((size_t)b * a.nsg + (sp < a.nsg ? sp : 0)) * C + c) * 2);
                    if (sp >= a.nsg) v[k][u] = make_float2(0.0f, 0.0f);
                }
            }
#pragma unroll
            for (int k = 0; k < CB; ++k)
#pragma unroll
                for (int u = 0; u < 4; ++u) { S1[k] += (double)v[k][u].x; S2[k] += (double)v[k][u].y; }
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            if (i0 + k >= cpg) break;
            const int c = g * cpg + i0 + k;
            double T1 = S1[k], T2 = S2[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { T1 += __shfl_xor(T1, off); T2 += __shfl_xor(T2, off); }
            const double sc = a.ada ? 1.0 + (double)a.ada[(size_t)b * a.ada_stride + c] : 1.0;
            const double gam = a.gamma[c], bet = a.beta[c];
            const double Gx = (double)rstd_f * (T2 - (double)mean_f * T1);
            if (lane == 0) {
                a.dgb[(size_t)b * 2 * C + c] = (float)(sc * Gx);
                a.dgb[(size_t)b * 2 * C + C + c] = (float)(sc * T1);
                if (a.dada) {
                    a.dada[(size_t)b * a.ada_stride + c] = (float)(gam * Gx + bet * T1);
                    a.dada[(size_t)b * a.ada_stride + C + c] = (float)T1;
                }
            }
            Ag += sc * gam * T1; Bg += sc * gam * Gx;
        }
    }
    const double qg = -(double)rstd_f * rstd_f * Bg / n;
    const double rg = -(double)rstd_f * Ag / n - (double)mean_f * qg;
    for (int i = lane; i < cpg; i += 64) {
        const int c = g * cpg + i;
        const float sc = a.ada ? 1.0f + a.ada[(size_t)b * a.ada_stride + c] : 1.0f;
        float4 o;
        o.x = rstd_f * sc * a.gamma[c]; o.y = (float)qg; o.z = (float)rg; o.w = 0.0f;
        *reinterpret_cast<float4*>(a.bcoef + ((size_t)b * C + c) * 4) = o;
    }
}

// ------------------------------------------------------------------------------- apply
struct GnBwdApplyArgs {
    GnBwdArgs g;
    const float* bcoef; const void* add1; const void* add2; void* dx1; void* dx2;
    int add1_mode, add2_mode;
};

// grid (pixel shares, B), 256 threads.  As in the partial-sum kernel a thread keeps ONE 16-byte channel chunk
// for all the pixels it visits, so the five per-channel coefficients (a, d, p, q, r) are read once into
// registers instead of once per element.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdApplyArgs a) {
    constexpr int VEC = Elem<T>::VEC;
    const GnBwdArgs& s = a.g;
    const int C = s.C1 + s.C2, cpr = C / VEC, HW = s.H * s.W;
    const int b = blockIdx.y;
    const int per = (HW + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * per, p1 = p0 + per < HW ? p0 + per : HW;
    const int rows = cpr >= 256 ? 1 : 256 / cpr;
    for (int cbase = 0; cbase < cpr; cbase += 256) {
        const int chunk = cbase + (cpr >= 256 ? (int)threadIdx.x : (int)threadIdx.x % cpr);
        const int row = cpr >= 256 ? 0 : (int)threadIdx.x / cpr;
        if (chunk >= cpr || row >= rows) continue;
        const int c = chunk * VEC;
        const T* src; T* dst; int cs, coff;
        if (c < s.C1) { src = (const T*)s.x1; dst = (T*)a.dx1; cs = s.C1; coff = c; }
        else { src = (const T*)s.x2; dst = (T*)a.dx2; cs = s.C2; coff = c - s.C1; }
        float ca[VEC], cd[VEC], kp[VEC], kq[VEC], kr[VEC];
        const float* cf = s.coef + ((size_t)b * C + c) * 2;
        const float* bc = a.bcoef + ((size_t)b * C + c) * 4;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            ca[i] = cf[2 * i]; cd[i] = cf[2 * i + 1];
            const float4 k = *reinterpret_cast<const float4*>(bc + 4 * i);
            kp[i] = k.x; kq[i] = k.y; kr[i] = k.z;
        }
        for (int p = p0 + row; p < p1; p += rows) {
            const size_t pix = (size_t)b * HW + p;
            const int y = p / s.W, x = p - y * s.W;
            Vec16<T> v; v.load(src + pix * cs + coff);
            float f[VEC], g[VEC], r[VEC]; v.unpack(f);
            load_grad<T>((const T*)s.dy, s.dy_mode, b, y, x, s.H, s.W, C, c, g);
            if (s.dm.thr) {
                const unsigned long long e0 = (unsigned long long)pix * C + c;
                float keep[VEC];
                drop_factors<VEC>(s.dm, e0, keep);
#pragma unroll
                for (int i = 0; i < VEC; ++i) g[i] *= keep[i];
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float gi = s.act ? g[i] * silu_grad_f(fmaf(f[i], ca[i], cd[i])) : g[i];
                r[i] = fmaf(gi, kp[i], fmaf(f[i], kq[i], kr[i]));
            }
            // the other gradients arriving at x span the whole (concatenated) channel axis, like dy
            if (a.add1) {
                float t[VEC]; load_grad<T>((const T*)a.add1, a.add1_mode, b, y, x, s.H, s.W, C, c, t);
#pragma unroll
                for (int i = 0; i < VEC; ++i) r[i] += t[i];
            }
            if (a.add2) {
                float t[VEC]; load_grad<T>((const T*)a.add2, a.add2_mode, b, y, x, s.H, s.W, C, c, t);
#pragma unroll
                for (int i = 0; i < VEC; ++i) r[i] += t[i];
            }
            Vec16<T> o; o.pack(r);
            o.store(dst + pix * cs + coff);
        }
    }
}

// dgamma[c], dbeta[c] (+)= sum over the batch of the per-image rows finalize wrote (dgb (B, 2C): dgamma then
// dbeta), in batch order; the two halves go to separate destinations (the flat gradient buffer).
__global__ __launch_bounds__(256) void gn_affine_grad_kernel(const float* dgb, int B, int C, float* dgamma, float* dbeta,
                                                              int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * C) return;
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += dgb[(size_t)b * 2 * C + c];
    float* d = c < C ? dgamma + c : dbeta + (c - C);
    *d = accumulate ? *d + s : s;
}

}  // namespace mad

using namespace mad;

static int gn_bwd_check(const char* who, const void* x1, int C1, const void* x2, int C2, const void* dy,
                        const float* coef, int dtype, int B, int H, int W, int dy_mode) {
    MAD_REQUIRE(x1 && dy && coef, "%s: null pointer", who);
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "%s: bad dtype %d", who, dtype);
    const int VEC = dtype == MAD_BF16 ? 8 : 4;
    MAD_REQUIRE(C1 > 0 && C1 % VEC == 0 && C2 >= 0 && C2 % VEC == 0 && (C2 == 0 || x2),
                "%s: channels must be multiples of %d (C1=%d C2=%d)", who, VEC, C1, C2);
    MAD_REQUIRE(B > 0 && H > 0 && W > 0, "%s: bad shape", who);
    MAD_REQUIRE(dy_mode >= 0 && dy_mode <= 2 && (dy_mode != 1 || (H % 2 == 0 && W % 2 == 0)),
                "%s: bad gradient resolution mode %d for %dx%d", who, dy_mode, H, W);
    return 0;
}

extern "C" int mad_gn_bwd_partial(const void* x1, int C1, const void* x2, int C2, const void* dy, int dy_mode,
                                  const float* coef, int act, unsigned long long drop_seed, float drop_p, int dtype,
                                  float* gsum, int B, int H, int W, int nsplit, void* stream) {
    MAD_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && (drop_p == 0.0f || dy_mode == 0),
                "mad_gn_bwd_partial: dropout needs p in [0,1) and a gradient at the same resolution");
    if (int rc = gn_bwd_check("mad_gn_bwd_partial", x1, C1, x2, C2, dy, coef, dtype, B, H, W, dy_mode)) return rc;
    MAD_REQUIRE(gsum && nsplit > 0 && (H * W) % nsplit == 0, "mad_gn_bwd_partial: HW=%d not divisible by nsplit=%d",
                H * W, nsplit);
    GnBwdArgs a;
    a.x1 = x1; a.x2 = C2 ? x2 : x1; a.dy = dy; a.coef = coef; a.C1 = C1; a.C2 = C2; a.B = B; a.H = H; a.W = W;
    a.act = act; a.dy_mode = dy_mode; a.dm = make_drop_mask(drop_seed, drop_p);
    const int VEC = dtype == MAD_BF16 ? 8 : 4;
    const int cpr = (C1 + C2) / VEC;
    const int rows = cpr >= 256 ? 1 : 256 / cpr;
    const size_t lds = cpr >= 256 ? 0 : (size_t)rows * cpr * 2 * VEC * sizeof(float);
    const dim3 grid(nsplit, B);
    if (dtype == MAD_BF16)
        hipLaunchKernelGGL(gn_bwd_partial_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, a, gsum, nsplit);
    else
        hipLaunchKernelGGL(gn_bwd_partial_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, a, gsum, nsplit);
    return check_launch("mad_gn_bwd_partial");
}

extern "C" int mad_gn_bwd_finalize(const float* stat1, int C1, int nsplit1, const float* stat2, int C2, int nsplit2,
                                   const float* gsum, int nsplit_g, const float* gamma, const float* beta,
                                   const float* ada, int ada_stride, float* bcoef, float* dgb, float* dada, int B,
                                   int groups, int HW, float eps, void* stream) {
    MAD_REQUIRE(stat1 && gsum && gamma && beta && bcoef && dgb, "mad_gn_bwd_finalize: null pointer");
    MAD_REQUIRE(C1 > 0 && nsplit1 > 0 && C2 >= 0 && (C2 == 0 || (stat2 && nsplit2 > 0)) && nsplit_g > 0,
                "mad_gn_bwd_finalize: bad partial-sum layout");
    MAD_REQUIRE(B > 0 && groups > 0 && (C1 + C2) % groups == 0 && HW > 0, "mad_gn_bwd_finalize: bad shape");
    MAD_REQUIRE(!dada || ada, "mad_gn_bwd_finalize: dada needs ada");
    GnBwdFinArgs a;
    a.stat1 = stat1; a.stat2 = C2 ? stat2 : stat1; a.gsum = gsum; a.gamma = gamma; a.beta = beta; a.ada = ada;
    a.bcoef = bcoef; a.dgb = dgb; a.dada = dada; a.C1 = C1; a.ns1 = nsplit1; a.C2 = C2; a.ns2 = C2 ? nsplit2 : 1;
    a.nsg = nsplit_g; a.ada_stride = ada_stride; a.groups = groups; a.HW = HW; a.eps = eps;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups, B), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch("mad_gn_bwd_finalize");
}

extern "C" int mad_gn_affine_grad(const float* dgb, int B, int C, float* dgamma, float* dbeta, int accumulate,
                                  void* stream) {
    MAD_REQUIRE(dgb && dgamma && dbeta, "mad_gn_affine_grad: null pointer");
    MAD_REQUIRE(B > 0 && C > 0, "mad_gn_affine_grad: bad shape");
    hipLaunchKernelGGL(gn_affine_grad_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, dgb, B, C,
                       dgamma, dbeta, accumulate);
    return check_launch("mad_gn_affine_grad");
}

extern "C" int mad_gn_bwd_apply(const void* x1, int C1, const void* x2, int C2, const void* dy, int dy_mode,
                                const float* coef, int act, unsigned long long drop_seed, float drop_p,
                                const float* bcoef, const void* add1, int add1_mode,
                                const void* add2, int add2_mode, void* dx1, void* dx2, int dtype, int B, int H, int W,
                                void* stream) {
    if (int rc = gn_bwd_check("mad_gn_bwd_apply", x1, C1, x2, C2, dy, coef, dtype, B, H, W, dy_mode)) return rc;
    MAD_REQUIRE(bcoef && dx1 && (C2 == 0 || dx2), "mad_gn_bwd_apply: null pointer");
    MAD_REQUIRE(add1_mode >= 0 && add1_mode <= 2 && add2_mode >= 0 && add2_mode <= 2, "mad_gn_bwd_apply: bad add mode");
    MAD_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f && (drop_p == 0.0f || dy_mode == 0),
                "mad_gn_bwd_apply: dropout needs p in [0,1) and a gradient at the same resolution");
    GnBwdApplyArgs a;
    a.g.x1 = x1; a.g.x2 = C2 ? x2 : x1; a.g.dy = dy; a.g.coef = coef; a.g.C1 = C1; a.g.C2 = C2; a.g.B = B; a.g.H = H;
    a.g.W = W; a.g.act = act; a.g.dy_mode = dy_mode; a.g.dm = make_drop_mask(drop_seed, drop_p);
    a.bcoef = bcoef; a.add1 = add1; a.add2 = add2; a.dx1 = dx1; a.dx2 = C2 ? dx2 : dx1;
    a.add1_mode = add1_mode; a.add2_mode = add2_mode;
    const int VEC = dtype == MAD_BF16 ? 8 : 4;
    // pixel shares: about 2048 workgroups in all, at least one sweep of pixels each
    const int cpr = (C1 + C2) / VEC, rows = cpr >= 256 ? 1 : 256 / cpr;      // pixels a workgroup covers per sweep
    int shares = (2048 + B - 1) / B;
    if (shares > (H * W + rows - 1) / rows) shares = (H * W + rows - 1) / rows;
    if (shares < 1) shares = 1;
    const dim3 grid(shares, B);
    if (dtype == MAD_BF16) hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("mad_gn_bwd_apply");
}
