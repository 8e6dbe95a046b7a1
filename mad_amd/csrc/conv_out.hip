// Last convolution of the denoiser (Cin -> at most 3 channels, 3x3, pad 1) with out_norm + SiLU folded into its
// loader, bf16 activations, NCHW fp32 output (the reference's out_conv(silu(out_norm(x))), model.py:589).
//
// With so few output channels the nine taps fit the N dimension of ONE matrix product: for every pixel p of the
// 18x18 halo of a 16x16 output tile
//        z[p][tap*Cout + co] = sum_c h[p][c] * w[co][tap][c],      h = act(x*a + d)   (zero outside the image)
// (M = 324 halo pixels in 11 blocks of 32, N = 9*Cout <= 32, K = Cin), and an output pixel is the sum of nine
// z entries of its neighbours, out[q][co] = bias[co] + sum_tap z[q + tap][tap*Cout + co].  Every halo pixel is read
// from HBM and transformed once per tile -- straight into the MFMA operand registers, 16 bytes per lane, no LDS
// staging of the input -- and the layer takes 8 MFMAs per 32 pixels instead of 72.  The weights (32 rows x Cin)
// stay in registers for the whole tile.  What bounds it is the transform's vector work and the one read of x.
#include "mma.h"

namespace mad {

struct ConvOutArgs {
    const bf16_t* x; const bf16_t* w; const float* coef; const float* bias; float* y;
    int B, H, W, Cin, Cout, act, tiles_x, tiles_y;
};

constexpr int CO_HS = 18;                  // halo side
constexpr int CO_PIX = CO_HS * CO_HS;      // 324 halo pixels
constexpr int CO_BLOCKS = (CO_PIX + 31) / 32;
constexpr int CO_ZLD = 36;                 // floats per z row: 16-byte aligned rows, 4 banks apart
constexpr int CO_MAXKS = 16;               // Cin <= 256

template <bool FUSE, int KS>               // KS = Cin / 16 K-steps
__global__ __launch_bounds__(256, 2) void conv_out_taps_kernel(ConvOutArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const z = reinterpret_cast<float*>(smem);                          // [CO_BLOCKS*32][CO_ZLD]
    float* const ctab = z + CO_BLOCKS * 32 * CO_ZLD;                         // [Cin][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    // tile coordinates (XCD-aware order: consecutive logical tiles share an L2)
    const int nwg = a.B * a.tiles_x * a.tiles_y;
    const int id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int tile = logical % (a.tiles_x * a.tiles_y);
    const int b = logical / (a.tiles_x * a.tiles_y);
    const int ty0 = (tile / a.tiles_x) * 16, tx0 = (tile % a.tiles_x) * 16;
    const int Cin = a.Cin;

    // weights as the first operand: row n = tap*Cout + co of lane&31, K half of lane>>5; rows >= 9*Cout are zero
    uint4 wf[KS];
    {
        const int n = fr, tap = n / a.Cout, co = n - tap * a.Cout;
        const bool real = n < 9 * a.Cout;
        const bf16_t* wp = a.w + ((size_t)(real ? co * 9 + tap : 0)) * Cin + 8 * fh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wf[ks] = *reinterpret_cast<const uint4*>(wp + ks * 16);
            if (!real) wf[ks] = make_uint4(0, 0, 0, 0);
        }
    }
    if (FUSE) {
        const float* cf = a.coef + (size_t)b * Cin * 2;
        for (int i = tid; i < Cin * 2; i += 256) ctab[i] = cf[i];
    }
    __syncthreads();
    const float xk1 = a.act ? -1.44269504088896340736f : 0.0f, xk0 = a.act ? 0.0f : -200.0f;

    // A wave takes the blocks wave, wave+4, wave+8; the pixels of the next block are requested before the
    // current one is transformed and multiplied.
    auto block_src = [&](int blk, bool& ok) {
        const int p = blk * 32 + fr;
        const int hy = p / CO_HS, hx = p - hy * CO_HS;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        ok = p < CO_PIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        // pixels outside the image read a pixel of their own image (discarded below)
        const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
        return a.x + ((size_t)(b * a.H + cy) * a.W + cx) * Cin + 8 * fh;
    };
    constexpr int NB = (CO_BLOCKS + 3) / 4;
    uint4 xr[2][KS];
    bool ok[2];
    {
        const bf16_t* src = block_src(wave, ok[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xr[0][ks] = *reinterpret_cast<const uint4*>(src + ks * 16);
    }
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
        const int blk = wave + 4 * ib;
        if (blk >= CO_BLOCKS) break;                       // wave-uniform
        if (ib + 1 < NB && blk + 4 < CO_BLOCKS) {
            const bf16_t* src = block_src(blk + 4, ok[(ib + 1) & 1]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xr[(ib + 1) & 1][ks] = *reinterpret_cast<const uint4*>(src + ks * 16);
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 v = xr[ib & 1][ks];
            if (FUSE) {
                const float* cf = ctab + (ks * 16 + 8 * fh) * 2;
                unsigned* w4 = reinterpret_cast<unsigned*>(&v);
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const float4 ad = *reinterpret_cast<const float4*>(cf + 4 * pr);
                    float t0 = __uint_as_float(w4[pr] << 16), t1 = __uint_as_float(w4[pr] & 0xffff0000u);
                    t0 = fmaf(t0, ad.x, ad.y); t1 = fmaf(t1, ad.z, ad.w);
                    const float u0 = 1.0f + __builtin_amdgcn_exp2f(fmaf(t0, xk1, xk0));
                    const float u1 = 1.0f + __builtin_amdgcn_exp2f(fmaf(t1, xk1, xk0));
                    t0 *= __builtin_amdgcn_rcpf(u0); t1 *= __builtin_amdgcn_rcpf(u1);
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t v2 = {t0, t1};
                    w4[pr] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2_t));
                }
            }
            if (!ok[ib & 1]) v = make_uint4(0, 0, 0, 0);
            mma_chunk<bf16_t>(acc, wf[ks], v);
        }
        // accumulator register e: row n = 8*(e/4) + 4*fh + e%4, column = pixel fr
        const int p = blk * 32 + fr;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v4 = {acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]};
            *reinterpret_cast<f32x4*>(z + p * CO_ZLD + 8 * g4 + 4 * fh) = v4;
        }
    }
    __syncthreads();

    // one output pixel per thread
    const int oy = tid >> 4, ox = tid & 15;
    float o[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* zp = z + ((oy + tap / 3) * CO_HS + ox + tap % 3) * CO_ZLD + tap * a.Cout;
#pragma unroll
        for (int co = 0; co < 3; ++co)
            if (co < a.Cout) o[co] += zp[co];
    }
#pragma unroll
    for (int co = 0; co < 3; ++co)
        if (co < a.Cout)
            a.y[(((size_t)b * a.Cout + co) * a.H + ty0 + oy) * a.W + tx0 + ox] = o[co] + a.bias[co];
}

// Whether the tap-major kernel takes the layer (bf16 activations assumed by the caller).
bool conv_out_taps_ok(int Cin, int H, int W, int Cout) {
    return Cout >= 1 && Cout <= 3 && Cin % 16 == 0 && Cin >= 16 && Cin / 16 <= CO_MAXKS && H % 16 == 0 && W % 16 == 0;
}

template <bool FUSE, int KS> static int launch_taps(const ConvOutArgs& a, hipStream_t s) {
    const int lds = (CO_BLOCKS * 32 * CO_ZLD + a.Cin * 2) * (int)sizeof(float);
    static int lds_dev = -1;
    MAD_REQUIRE(raise_lds_limit(reinterpret_cast<const void*>(&conv_out_taps_kernel<FUSE, KS>), lds, &lds_dev),
                "mad_conv_out_fused: cannot raise the dynamic LDS limit to %d", lds);
    hipLaunchKernelGGL((conv_out_taps_kernel<FUSE, KS>), dim3(a.B * a.tiles_x * a.tiles_y), dim3(256), lds, s, a);
    return check_launch("mad_conv_out_fused");
}

int conv_out_taps(const void* x, int Cin, const float* coef, int act, const void* w, const float* bias, float* y, int B,
                  int H, int W, int Cout, hipStream_t s) {
    ConvOutArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.coef = coef; a.bias = bias; a.y = y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = act; a.tiles_x = W / 16; a.tiles_y = H / 16;
    switch (Cin / 16) {
#define MAD_KS(K_) case K_: return coef ? launch_taps<true, K_>(a, s) : launch_taps<false, K_>(a, s);
        MAD_KS(1) MAD_KS(2) MAD_KS(3) MAD_KS(4) MAD_KS(5) MAD_KS(6) MAD_KS(7) MAD_KS(8)
        MAD_KS(9) MAD_KS(10) MAD_KS(11) MAD_KS(12) MAD_KS(13) MAD_KS(14) MAD_KS(15) MAD_KS(16)
#undef MAD_KS
    }
    set_error("mad_conv_out_fused: Cin=%d not handled by the tap-major kernel", Cin);
    return 1;
}

}  // namespace mad
