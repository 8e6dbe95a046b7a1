// Weight gradients of the convolutions (training step): dW[co][ky][kx][ci] = sum over (b, y, x) of
// dY[b,y,x,co] * X[b, y+ky-pad, x+kx-pad, ci], and the bias gradient (column sums of dY).
//
// The product reduces over PIXELS, while both operands are stored channels-last (a pixel's channels are
// contiguous).  MFMA operands need the reduction index contiguous per lane, so the tiles are staged in
// LDS as they lie in memory ([pixel][channel]) and read with the hardware-transposed LDS read
// (ds_read_b64_tr_b16): a 16-lane group fetches a 4-pixel x 16-channel block and lane i receives channel
// i of the four pixels.  Two such reads give the 8 reduction elements of a v_mfma_f32_32x32x16_bf16
// operand; A (dY) and B (X) use the same pixel order inside a 16-pixel step.
//
// One workgroup owns a (128 output channels) x (32 input channels) x (all taps) slice of dW and walks a
// strided share of the pixel tiles (4 rows x 16 columns of one image, or one 8x8 image); wave w keeps
// the 32 output channels 32w.. of every tap in registers (9 accumulators).  The input halo of a tile is
// staged once and the taps read it at shifted pixel offsets.  Workgroups that share a slice write their
// sums to separate partial buffers, which a second kernel adds in a fixed order (deterministic).
#include "mma.h"

namespace mad {

struct WgradArgs {
    const void* x; const void* dy; float* partial;
    float* bias_partial;                      // [ksplit][Cout] column sums of dY, or null
    int B, H, W, Cin, Cout, ksplit, ntiles, upsample;
};

namespace wg {
constexpr int YROW = 256;                  // bytes of dY per pixel in LDS: 128 bf16 output channels
constexpr int XROW = 64;                   // bytes of X per halo pixel in LDS: 32 bf16 input channels
// LDS of a workgroup with WCI groups of 32 input channels: two stages (the tile being multiplied and the tile
// being copied) of 64*WCI pixels of dY and WCI halo regions of up to 128 (WCI 1) or 256 (WCI 2) pixels
constexpr int x_bytes(int wci) { return wci == 1 ? 8 * 1024 : 16 * 1024; }    // whole rounds of four pieces
constexpr int lds_bytes(int wci) { return 2 * (64 * wci * YROW + wci * x_bytes(wci)); }
}  // namespace wg

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lo, const unsigned char* hi) {
    const bf16x4_t l = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)lo);
    const bf16x4_t h = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)hi);
    return bf16x8{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
}

// pointer select by masking: the compiler cannot turn it back into a branch around the address arithmetic
__device__ __forceinline__ const void* pick_ptr(bool ok, const void* p, const void* other) {
    const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
    return (const void*)(((uintptr_t)p & m) | ((uintptr_t)other & ~m));
}

// KS: 1 or 3.  A tile is TS = 4*WCI steps of 16 pixels.  SW: 16 = TS rows x 16 columns of one image
// (H % TS == 0, W % 16 == 0); 8 = TS/4 whole 8x8 images (consecutive images are consecutive rows).
// Two tiles are resident: the copies of the next tile are issued right after the barrier that hands over the
// current one and land while it is multiplied.  A dY piece's pixel is (compile-time piece row) + (per-lane
// pixel), so its address is a per-lane offset plus scalars.
// WCI: 1 = four waves, a 128 x 32 slice, 64-pixel tiles, two workgroups per CU; 2 = eight waves (wave = (ci group,
// co group)), a 128 x 64 slice, 128-pixel tiles, one workgroup per CU: the dY tile is copied once for both ci
// groups, half as many partial sums are written and half as many barriers are passed.
// BIAS: also produce the column sums of dY.
template <int KS, int SW, int WCI, bool BIAS>
__global__ __launch_bounds__(256 * WCI, 2 / WCI) void wgrad_mfma_kernel(WgradArgs a) {
    using namespace wg;
    constexpr int PAD = KS / 2, NT = KS * KS, TS = 4 * WCI, TPIX = 16 * TS;
    constexpr int NIMG = SW == 16 ? 1 : (TS + 3) / 4, SH = SW == 16 ? TS : 8;
    constexpr int HPW = SW + 2 * PAD, HPH = SH + 2 * PAD, HPIX = NIMG * HPH * HPW;
    constexpr int XPIECES = (HPIX + 15) / 16;
    constexpr int OFF_X = TPIX * YROW, X_BYTES = x_bytes(WCI);
    static_assert(XPIECES * 1024 <= X_BYTES, "halo region");
    constexpr int STAGE = OFF_X + WCI * X_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wco = wave & 3, wci = wave >> 2;

    const int ncic = a.Cin / (32 * WCI);
    const int ks = blockIdx.x % a.ksplit;
    const int rest = blockIdx.x / a.ksplit;
    const int ci0 = (rest % ncic) * 32 * WCI + 32 * wci, co0 = (rest / ncic) * 128;
    const bf16_t* X = (const bf16_t*)a.x;
    const bf16_t* DY = (const bf16_t*)a.dy;
    const int Hs = a.upsample ? a.H >> 1 : a.H, Ws = a.upsample ? a.W >> 1 : a.W;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    // The bias gradient (column sums of dY) is the weight gradient of an input channel of ones: the waves of the
    // first ci group of the first ci slice add one MFMA per step with an all-ones B operand.
    const bool with_bias = BIAS && rest % ncic == 0 && wci == 0;
    f32x16 accb;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.0f;
    const __bf16 one_bf = (__bf16)1.0f;
    const bf16x8 ones = {one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf};

    // transposed-read lane roles: group g = lane>>4 -> channel half (g&1), pixel half h = g>>1;
    // inside the group lane 4q+pp addresses pixel q, channels 4pp..4pp+3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = g >> 1;
    const int a_lchunk = 4 * wco + (g & 1) * 2 + (pp >> 1);            // 16-byte chunk of this wave's 32 channels
    const int a_off = ((a_lchunk ^ (q << 2)) << 4) + (pp & 1) * 8;      // chunk index XOR (pixel & 3) << 2
    const int b_off = (g & 1) * 32 + pp * 8;
    const int tpi = SW == 16 ? (a.H / TS) * (a.W >> 4) : 1;             // tiles per image
    const int tpr = SW == 16 ? (a.W >> 4) : 1;

    // dY copies, shared by all waves: piece j*4*WCI + wave holds pixels 16*WCI*j + pl, pl = 4*wave + lane/16;
    // chunk c of pixel p holds logical chunk c ^ ((p&3)<<2)
    const int pl = 4 * wave + (lane >> 4);
    const int y_px = pl % SW, y_py = pl / SW;
    const int y_co = co0 + (((lane & 15) ^ ((pl & 3) << 2)) << 3);
    const bool y_ok = y_co < a.Cout;                                      // Cout is a multiple of 8
    const int y_lane_off = (y_py * a.W + y_px) * a.Cout + y_co;
    const int y_piece_stride = (16 * WCI / SW) * a.W * a.Cout;            // 16*WCI pixels = this many rows

    // The copies of a tile are issued in NPART parts, one per multiply step, so that the waves of a SIMD do not
    // all sit in copy issue at the same time.  Part `part` (compile-time) is one dY piece or one X piece per
    // wave; `live` false (no next tile) turns every read into a read of the zero page.
    constexpr int YPARTS = TPIX / (16 * WCI), XPARTS = (XPIECES + 3) / 4, NPART = YPARTS + XPARTS;
    struct Origin { int b0, y0, x0; };
    auto origin = [&](int tile) {
        Origin o;
        if (SW == 16) { o.b0 = tile / tpi; const int r = tile - o.b0 * tpi; o.y0 = (r / tpr) * TS; o.x0 = (r % tpr) * 16; }
        else { o.b0 = tile * NIMG; o.y0 = 0; o.x0 = 0; }
        return o;
    };
    auto stage_part = [&](const Origin& o, bool live, int buf, int part) {
        const unsigned base = lds0 + buf * STAGE;
        if (part < YPARTS) {
            const int j = part;
            const bf16_t* src = DY + ((size_t)(o.b0 * a.H + o.y0) * a.W + o.x0) * a.Cout + y_lane_off +
                                (size_t)j * y_piece_stride;
            const bool img_ok = NIMG == 1 || o.b0 + (16 * WCI * j) / 64 < a.B;   // SW 8: the image of pixel 16*WCI*j
            glds16(pick_ptr(live & y_ok & img_ok, src, g_zero_page),
                   __builtin_amdgcn_readfirstlane(base + (j * 4 * WCI + wave) * 1024));
        } else if (part < NPART) {
            // X halo of this wave's ci group (copied by its four waves): pieces of 16 halo pixels x 64 bytes;
            // pieces past the halo exist in LDS and are filled with zeros
            const int piece = (part - YPARTS) * 4 + wco;
            const unsigned hp = piece * 16 + (lane >> 2);
            const unsigned img = NIMG == 1 ? 0u : (hp >= (unsigned)(HPH * HPW) ? 1u : 0u), r = hp - img * (HPH * HPW);
            const unsigned hy = r / HPW, hx = r - hy * HPW;
            const int yy = o.y0 + (int)hy - PAD, xx = o.x0 + (int)hx - PAD;
            const bool ok = live & (hp < (unsigned)HPIX) & ((unsigned)yy < (unsigned)a.H) &
                            ((unsigned)xx < (unsigned)a.W) & (o.b0 + (int)img < a.B);
            const int sy = (a.upsample ? yy >> 1 : yy) + (int)img * Hs, sx = a.upsample ? xx >> 1 : xx;
            const bf16_t* src = X + (size_t)o.b0 * Hs * Ws * a.Cin + ci0 + (lane & 3) * 8 + (sy * Ws + sx) * a.Cin;
            glds16(pick_ptr(ok, src, g_zero_page),
                   __builtin_amdgcn_readfirstlane(base + OFF_X + wci * X_BYTES + piece * 1024));
        }
    };

    int buf = 0;
    if (ks < a.ntiles) {
        const Origin o = origin(ks);
#pragma unroll
        for (int part = 0; part < NPART; ++part) stage_part(o, true, 0, part);
    }
    for (int tile = ks; tile < a.ntiles; tile += a.ksplit) {
        dma_wait_all();
        __syncthreads();                     // this tile has landed and the other stage has been consumed
        const bool more = tile + a.ksplit < a.ntiles;
        const Origin nxt = origin(more ? tile + a.ksplit : tile);
        const unsigned char* ys = smem + buf * STAGE;
        const unsigned char* xs = ys + OFF_X + wci * X_BYTES;
        // The shifted-X operand of tap row ky at step s covers halo row rowid(s, ky) (and the next one when a step
        // is two image rows).  Consecutive steps share halo rows, so an operand is read from LDS at its first
        // use and kept in registers until its last.
        auto rowid = [](int s, int ky) { return SW == 16 ? s + ky : (s >> 2) * HPH + 2 * (s & 3) + ky; };
        bf16x8 xf[NIMG * HPH][KS];
#pragma unroll
        for (int s = 0; s < TS; ++s) {
            const int pa = 16 * s + 4 * h + q;                           // tile pixel of the low half
            const bf16x8 af = tr_frag(ys + pa * YROW + a_off, ys + (pa + 8) * YROW + a_off);
#pragma unroll
            for (int part = s * NPART / TS; part < (s + 1) * NPART / TS; ++part) stage_part(nxt, more, buf ^ 1, part);
            if (BIAS && with_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ones, accb, 0, 0, 0);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int R = rowid(s, ky);
                if (s == 0 || R > rowid(s - 1, KS - 1)) {                // first use of halo row R
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const int hp_lo = R * HPW + kx + 4 * h + q, hp_hi = hp_lo + (SW == 16 ? 8 : HPW);
                        xf[R][kx] = tr_frag(xs + hp_lo * XROW + b_off, xs + hp_hi * XROW + b_off);
                    }
                }
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
                    acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xf[R][kx], acc[ky * KS + kx], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    // ---- partial[ks][co][tap][ci]
    float* out = a.partial + (size_t)ks * a.Cout * NT * a.Cin;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + 32 * wco + acc_row(e, lane);
            if (co < a.Cout) out[((size_t)co * NT + t) * a.Cin + ci0 + (lane & 31)] = acc[t][e];
        }
    if (BIAS && with_bias && (lane & 31) == 0) {                  // every column of accb holds the same sums
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + 32 * wco + acc_row(e, lane);
            if (co < a.Cout) a.bias_partial[(size_t)ks * a.Cout + co] = accb[e];
        }
    }
}

// Generic (any channel counts, fp32 or bf16, strided operands) weight gradient on the vector ALU: the
// first and last layers (3 image channels) and the fp32 mode.  One workgroup = one (co, tap) pair and a
// share of the pixels; threads run over ci.  Element (b,y,x,c) of an operand is at b*sb + y*sy + x*sx + c*sc.
struct WgradGenArgs {
    const void* x; const void* dy; float* partial;
    int B, H, W, Cin, Cout, ks, ksplit, upsample;
    long xsb, xsy, xsx, xsc, ysb, ysy, ysx, ysc;
};

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(WgradGenArgs a) {
    const int nt = a.ks * a.ks, pad = a.ks / 2;
    const int split = blockIdx.x % a.ksplit;
    const int rest = blockIdx.x / a.ksplit;
    const int t = rest % nt, co = rest / nt;
    const int ky = t / a.ks - pad, kx = t % a.ks - pad;
    const TX* X = (const TX*)a.x;
    const TY* DY = (const TY*)a.dy;
    const long npix = (long)a.B * a.H * a.W;
    for (int ci = threadIdx.x; ci < a.Cin; ci += 256) {
        float s = 0.0f;
        for (long p = split; p < npix; p += a.ksplit) {
            const int x = (int)(p % a.W), y = (int)((p / a.W) % a.H), b = (int)(p / ((long)a.W * a.H));
            const int yy = y + ky, xx = x + kx;
            if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
            const int sy = a.upsample ? yy >> 1 : yy, sx = a.upsample ? xx >> 1 : xx;
            const float g = Elem<TY>::load(DY[b * a.ysb + y * a.ysy + x * a.ysx + co * a.ysc]);
            s = fmaf(g, Elem<TX>::load(X[b * a.xsb + sy * a.xsy + sx * a.xsx + ci * a.xsc]), s);
        }
        a.partial[(((size_t)split * a.Cout + co) * nt + t) * a.Cin + ci] = s;
    }
}

// dst[i] (+)= sum over s of partial[s][i], in the order s = 0, 1, ... (taps == 0: plain vectors).
__global__ void wgrad_reduce_kernel(const float* partial, float* dst, size_t n, int nsplit, int accumulate, int taps,
                                    int Cin) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
        s0 += partial[(size_t)k * n + i]; s1 += partial[(size_t)(k + 1) * n + i];
        s2 += partial[(size_t)(k + 2) * n + i]; s3 += partial[(size_t)(k + 3) * n + i];
    }
    for (; k < nsplit; ++k) s0 += partial[(size_t)k * n + i];
    const float s = (s0 + s1) + (s2 + s3);
    dst[i] = accumulate ? dst[i] + s : s;
}

// The same sum for conv weights: the partials are (Cout, taps, Cin), the master weights (Cout, Cin, taps).
// One workgroup = one output channel x 32 input channels; the block is transposed through LDS so both
// sides stay coalesced.  grid (Cout, ceil(Cin/32)).
__global__ __launch_bounds__(288) void wgrad_reduce_oihw_kernel(const float* partial, float* dst, int Cout, int nsplit,
                                                                 int accumulate, int taps, int Cin,
                                                                 const float* bias_partial, float* dbias,
                                                                 int bias_accumulate) {
    __shared__ float blk[9 * 32];                         // [taps][32]
    const int co = blockIdx.x, c0 = blockIdx.y * 32;
    if (bias_partial != nullptr && blockIdx.y == 0 && threadIdx.x < 64) {      // wave 0: dbias[co], fixed order
        float sb = 0.0f;
        for (int k = threadIdx.x; k < nsplit; k += 64) sb += bias_partial[(size_t)k * Cout + co];
        for (int off = 32; off > 0; off >>= 1) sb += __shfl_down(sb, off, 64);
        if (threadIdx.x == 0) dbias[co] = bias_accumulate ? dbias[co] + sb : sb;
    }
    const int cw = Cin - c0 < 32 ? Cin - c0 : 32;
    const size_t n = (size_t)Cout * taps * Cin;
    const int t = threadIdx.x >> 5, cl = threadIdx.x & 31;
    if (t < taps && cl < cw) {
        const float* p = partial + ((size_t)co * taps + t) * Cin + c0 + cl;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {
            s0 += p[(size_t)k * n]; s1 += p[(size_t)(k + 1) * n]; s2 += p[(size_t)(k + 2) * n]; s3 += p[(size_t)(k + 3) * n];
        }
        for (; k < nsplit; ++k) s0 += p[(size_t)k * n];
        blk[t * 32 + cl] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    const int o = threadIdx.x;                            // position inside the (cw, taps) output block
    if (o < cw * taps) {
        const int ci = o / taps, tt = o - ci * taps;
        float* d = dst + ((size_t)co * Cin + c0) * taps + o;
        const float v = blk[tt * 32 + ci];
        *d = accumulate ? *d + v : v;
    }
}

// Column sums of a (rows, C) matrix, rows split over blockIdx.y: partial[split][c].
template <typename T> __global__ __launch_bounds__(256) void colsum_kernel(const T* x, float* partial, long rows, int C,
                                                                           long sr, long sc) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
    for (long r = blockIdx.y; r < rows; r += gridDim.y) s += Elem<T>::load(x[r * sr + c * sc]);
    partial[(size_t)blockIdx.y * C + c] = s;
}

// First / last layer: one operand is channels-last with many channels (V: dY of the first conv, X of the
// last), the other is the NCHW fp32 image side with at most 4 channels (S).  A thread owns one V channel
// and keeps all (S channel, tap) sums in registers, so V is read exactly once, coalesced over channels;
// the S values of a pixel's 3x3 neighbourhood are the same for every thread of a pixel row group.
//   acc[cs][t] = sum_p V[p][cv] * S[cs][p + sgn*(tap offset)]      (sgn = +1 first layer, -1 last layer)
struct WgradEdgeArgs {
    const void* v; const float* s; float* partial;
    int B, H, W, CV, CS, sgn, nsplit;
};

template <typename TV>
__global__ __launch_bounds__(256) void wgrad_edge_kernel(WgradEdgeArgs a) {
    __shared__ float red[256 * 36];
    const int rows = 256 / a.CV > 0 ? 256 / a.CV : 1;           // pixel streams per workgroup (CV <= 256)
    const int cv = threadIdx.x % a.CV, r = threadIdx.x / a.CV;
    const bool active = r < rows;
    const TV* V = (const TV*)a.v;
    const long npix = (long)a.B * a.H * a.W;
    const long per = (npix + a.nsplit - 1) / a.nsplit;
    const long p0 = (long)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    float acc[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.0f;
    if (active) {
        for (long p = p0 + r; p < p1; p += rows) {
            const int x = (int)(p % a.W), y = (int)((p / a.W) % a.H), b = (int)(p / ((long)a.W * a.H));
            const float v = Elem<TV>::load(V[p * a.CV + cv]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + a.sgn * (t / 3 - 1), xx = x + a.sgn * (t % 3 - 1);
                if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < a.CS) acc[c][t] = fmaf(v, a.s[(((long)b * a.CS + c) * a.H + yy) * a.W + xx], acc[c][t]);
            }
        }
    }
    // sum the pixel streams of this workgroup, then one row of partials per workgroup
    for (int c = 0; c < 4; ++c)
        for (int t = 0; t < 9; ++t) red[threadIdx.x * 36 + c * 9 + t] = active ? acc[c][t] : 0.0f;
    __syncthreads();
    if (threadIdx.x < a.CV) {
        for (int j = 0; j < 36; ++j) {
            float sum = 0.0f;
            for (int rr = 0; rr < rows; ++rr) sum += red[(rr * a.CV + cv) * 36 + j];
            a.partial[((size_t)blockIdx.x * a.CV + cv) * 36 + j] = sum;
        }
    }
}

// dw (OIHW) = sum over splits of partial[split][cv][cs*9+t]; first layer: (O, I) = (cv, cs); last: (cs, cv)
__global__ void wgrad_edge_reduce_kernel(const float* partial, float* dw, int nsplit, int CV, int CS, int v_is_out,
                                         int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CV * CS * 9) return;
    const int t = i % 9, cs = (i / 9) % CS, cv = i / (9 * CS);
    float s = 0.0f;
    for (int k = 0; k < nsplit; ++k) s += partial[((size_t)k * CV + cv) * 36 + cs * 9 + t];
    const int o = v_is_out ? (cv * CS + cs) * 9 + t : (cs * CV + cv) * 9 + t;
    dw[o] = accumulate ? dw[o] + s : s;
}

// `slots` workgroups are resident on the chip (256 CUs): one full round, since a second, partly filled
// round would idle CUs and every extra split adds a partial buffer to write and reduce.
static int pick_ksplit(int slices, int ntiles, int slots) {
    int k = slots / slices;
    if (k > ntiles) k = ntiles;
    if (k > 256) k = 256;
    return k < 1 ? 1 : k;
}

}  // namespace mad

using namespace mad;

static bool wgrad_mfma_ok(int dtype, int H, int W, int Cin, int Cout, int ksize) {
    if (dtype != MAD_BF16 || Cin % 32 != 0 || Cout % 8 != 0) return false;
    if (H == 8 && W == 8) return true;
    return H % 8 == 0 && W % 16 == 0;
}

// ci groups per workgroup of the MFMA kernel
static int wgrad_wci(int Cin) {
    const char* e = getenv("MAD_WGRAD_WCI");
    if (e && e[0] == '1') return 1;
    return Cin % 64 == 0 ? 2 : 1;
}

static int wgrad_ksplit(int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int* ntiles_out) {
    if (wgrad_mfma_ok(dtype, H, W, Cin, Cout, ksize)) {
        const int wci = wgrad_wci(Cin);
        const int ntiles = (H == 8 && W == 8) ? (B + wci - 1) / wci : B * (H / (4 * wci)) * (W / 16);
        if (ntiles_out) *ntiles_out = ntiles;
        return pick_ksplit(((Cout + 127) / 128) * (Cin / (32 * wci)), ntiles, 512 / wci);
    }
    if (ntiles_out) *ntiles_out = 0;
    int k = (1024 + Cout * ksize * ksize - 1) / (Cout * ksize * ksize);
    return k < 1 ? 1 : (k > 256 ? 256 : k);
}

// partial sums of dW, then room for the partial sums of the bias gradient (either form)
static size_t wgrad_main_bytes(int k, int Cin, int Cout, int ksize) {
    return (size_t)k * Cout * ksize * ksize * Cin * sizeof(float);
}

extern "C" size_t mad_conv_wgrad_workspace(int dtype, int B, int H, int W, int Cin, int Cout, int ksize) {
    const int k = wgrad_ksplit(dtype, B, H, W, Cin, Cout, ksize, nullptr);
    return wgrad_main_bytes(k, Cin, Cout, ksize) + (size_t)(k > 256 ? k : 256) * Cout * sizeof(float);
}

static int bias_grad_impl(const void* dy, int dtype, long rows, int C, float* dbias, float* workspace, int accumulate,
                          hipStream_t s);

extern "C" int mad_conv_wgrad(const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                              int dtype, int B, int H, int W, int Cin, int Cout, int ksize, int upsample,
                              int accumulate, void* stream) {
    return mad_conv_wgrad_bias(x, dy, dw, nullptr, workspace, workspace_bytes, dtype, B, H, W, Cin, Cout, ksize,
                               upsample, accumulate, 0, stream);
}

extern "C" int mad_conv_wgrad_bias(const void* x, const void* dy, float* dw, float* dbias, void* workspace,
                                   size_t workspace_bytes, int dtype, int B, int H, int W, int Cin, int Cout,
                                   int ksize, int upsample, int accumulate, int bias_accumulate, void* stream) {
    MAD_REQUIRE(x && dy && dw && workspace, "mad_conv_wgrad: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv_wgrad: bad dtype %d", dtype);
    MAD_REQUIRE(ksize == 1 || ksize == 3, "mad_conv_wgrad: ksize %d not supported", ksize);
    MAD_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "mad_conv_wgrad: bad shape");
    MAD_REQUIRE(!upsample || (H % 2 == 0 && W % 2 == 0), "mad_conv_wgrad: upsample needs even H, W");
    MAD_REQUIRE(workspace_bytes >= mad_conv_wgrad_workspace(dtype, B, H, W, Cin, Cout, ksize),
                "mad_conv_wgrad: workspace of %zu bytes is too small", workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int nt = ksize * ksize;
    int ntiles = 0;
    const int ksplit = wgrad_ksplit(dtype, B, H, W, Cin, Cout, ksize, &ntiles);
    float* bias_ws = (float*)((char*)workspace + wgrad_main_bytes(ksplit, Cin, Cout, ksize));
    const bool fused_bias = dbias && wgrad_mfma_ok(dtype, H, W, Cin, Cout, ksize);
    if (wgrad_mfma_ok(dtype, H, W, Cin, Cout, ksize)) {
        WgradArgs a;
        a.x = x; a.dy = dy; a.partial = (float*)workspace; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
        a.ksplit = ksplit; a.ntiles = ntiles; a.upsample = upsample;
        a.bias_partial = dbias ? bias_ws : nullptr;
        const int wci = wgrad_wci(Cin);
        const dim3 grid(((Cout + 127) / 128) * (Cin / (32 * wci)) * ksplit);
        const bool small = H == 8 && W == 8;
#define MAD_WG_LAUNCH2(KS, SW, BIAS)                                                                                 \
    do {                                                                                                             \
        if (wci == 2) {                                                                                              \
            static int lds_dev = -1;                                                                                 \
            MAD_REQUIRE(raise_lds_limit(reinterpret_cast<const void*>(&wgrad_mfma_kernel<KS, SW, 2, BIAS>),          \
                                        wg::lds_bytes(2), &lds_dev),                                                 \
                        "mad_conv_wgrad: cannot raise the dynamic LDS limit");                                       \
            hipLaunchKernelGGL((wgrad_mfma_kernel<KS, SW, 2, BIAS>), grid, dim3(512), wg::lds_bytes(2), s, a);       \
        } else {                                                                                                     \
            hipLaunchKernelGGL((wgrad_mfma_kernel<KS, SW, 1, BIAS>), grid, dim3(256), wg::lds_bytes(1), s, a);       \
        }                                                                                                            \
    } while (0)
#define MAD_WG_LAUNCH(KS, SW)                                                                                        \
    do {                                                                                                             \
        if (fused_bias) MAD_WG_LAUNCH2(KS, SW, true); else MAD_WG_LAUNCH2(KS, SW, false);                            \
    } while (0)
        if (ksize == 3) { if (small) MAD_WG_LAUNCH(3, 8); else MAD_WG_LAUNCH(3, 16); }
        else { if (small) MAD_WG_LAUNCH(1, 8); else MAD_WG_LAUNCH(1, 16); }
#undef MAD_WG_LAUNCH
#undef MAD_WG_LAUNCH2
    } else {
        WgradGenArgs a;
        a.x = x; a.dy = dy; a.partial = (float*)workspace; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
        a.ks = ksize; a.ksplit = ksplit; a.upsample = upsample;
        const int Hs = upsample ? H / 2 : H, Ws = upsample ? W / 2 : W;
        a.xsc = 1; a.xsx = Cin; a.xsy = (long)Ws * Cin; a.xsb = (long)Hs * Ws * Cin;
        a.ysc = 1; a.ysx = Cout; a.ysy = (long)W * Cout; a.ysb = (long)H * W * Cout;
        const dim3 grid(Cout * nt * ksplit);
        if (dtype == MAD_BF16) hipLaunchKernelGGL((wgrad_generic_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_generic_kernel<float, float>), grid, dim3(256), 0, s, a);
    }
    if (int rc = check_launch("mad_conv_wgrad")) return rc;
    hipLaunchKernelGGL(wgrad_reduce_oihw_kernel, dim3(Cout, (Cin + 31) / 32), dim3(288), 0, s,
                       (const float*)workspace, dw, Cout, ksplit, accumulate, nt, Cin,
                       fused_bias ? (const float*)bias_ws : (const float*)nullptr, dbias, bias_accumulate);
    if (int rc = check_launch("mad_conv_wgrad (reduce)")) return rc;
    if (dbias && !fused_bias) return bias_grad_impl(dy, dtype, (long)B * H * W, Cout, dbias, bias_ws, bias_accumulate, s);
    return 0;
}

extern "C" size_t mad_bias_grad_workspace(int C) { return (size_t)256 * C * sizeof(float); }

static int bias_grad_impl(const void* dy, int dtype, long rows, int C, float* dbias, float* workspace, int accumulate,
                          hipStream_t s) {
    // enough row shares to fill the chip for long inputs, few for short ones (the second pass adds them serially)
    long want = rows / 64;
    const int splits = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
    const dim3 grid((C + 255) / 256, splits);
    if (dtype == MAD_BF16)
        hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)dy, workspace, rows, C, (long)C,
                           1L);
    else
        hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)dy, workspace, rows, C, (long)C,
                           1L);
    if (int rc = check_launch("mad_bias_grad")) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)workspace, dbias,
                       (size_t)C, splits, accumulate, 0, 1);
    return check_launch("mad_bias_grad (reduce)");
}

extern "C" int mad_bias_grad(const void* dy, int dtype, long rows, int C, float* dbias, void* workspace,
                             size_t workspace_bytes, int accumulate, void* stream) {
    MAD_REQUIRE(dy && dbias && workspace, "mad_bias_grad: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_bias_grad: bad dtype %d", dtype);
    MAD_REQUIRE(rows > 0 && C > 0, "mad_bias_grad: bad shape");
    MAD_REQUIRE(workspace_bytes >= mad_bias_grad_workspace(C), "mad_bias_grad: workspace too small");
    return bias_grad_impl(dy, dtype, rows, C, dbias, (float*)workspace, accumulate, (hipStream_t)stream);
}

// First / last layer: operands in different dtypes and layouts (the image side is NCHW fp32).
static int edge_splits(int B, int H, int W) {
    const long npix = (long)B * H * W;
    long k = npix / 256;                                   // at least 256 pixels per workgroup
    return (int)(k < 1 ? 1 : (k > 1024 ? 1024 : k));
}

extern "C" size_t mad_conv_wgrad_edge_workspace(int B, int H, int W, int Cin, int Cout, int ksize) {
    const size_t generic = mad_conv_wgrad_workspace(MAD_F32, B, H, W, Cin, Cout, ksize);   // MAD_F32: the generic split
    const int cv = Cin > Cout ? Cin : Cout;
    const size_t edge = (size_t)edge_splits(B, H, W) * cv * 36 * sizeof(float);
    return generic > edge ? generic : edge;
}

extern "C" int mad_conv_wgrad_edge(const void* x, int x_dtype, int x_nchw, const void* dy, int dy_dtype, int dy_nchw,
                                   float* dw, void* workspace, size_t workspace_bytes, int B, int H, int W, int Cin,
                                   int Cout, int ksize, int accumulate, void* stream) {
    MAD_REQUIRE(x && dy && dw && workspace, "mad_conv_wgrad_edge: null pointer");
    MAD_REQUIRE((x_dtype == MAD_BF16 || x_dtype == MAD_F32) && (dy_dtype == MAD_BF16 || dy_dtype == MAD_F32),
                "mad_conv_wgrad_edge: bad dtype");
    MAD_REQUIRE(ksize == 1 || ksize == 3, "mad_conv_wgrad_edge: ksize %d not supported", ksize);
    MAD_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "mad_conv_wgrad_edge: bad shape");
    MAD_REQUIRE(workspace_bytes >= mad_conv_wgrad_edge_workspace(B, H, W, Cin, Cout, ksize),
                "mad_conv_wgrad_edge: workspace of %zu bytes is too small", workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int nt = ksize * ksize;
    // the two shapes of the denoiser: first conv (image in, channels-last gradient) and last conv
    const bool first = ksize == 3 && x_nchw && x_dtype == MAD_F32 && !dy_nchw && Cin <= 4 && Cout <= 256;
    const bool last = ksize == 3 && !x_nchw && dy_nchw && dy_dtype == MAD_F32 && Cout <= 4 && Cin <= 256;
    if (first || last) {
        WgradEdgeArgs e;
        e.v = first ? dy : x; e.s = (const float*)(first ? x : dy); e.partial = (float*)workspace;
        e.B = B; e.H = H; e.W = W; e.CV = first ? Cout : Cin; e.CS = first ? Cin : Cout; e.sgn = first ? 1 : -1;
        e.nsplit = edge_splits(B, H, W);
        const int vdt = first ? dy_dtype : x_dtype;
        if (vdt == MAD_BF16) hipLaunchKernelGGL(wgrad_edge_kernel<bf16_t>, dim3(e.nsplit), dim3(256), 0, s, e);
        else hipLaunchKernelGGL(wgrad_edge_kernel<float>, dim3(e.nsplit), dim3(256), 0, s, e);
        if (int rc = check_launch("mad_conv_wgrad_edge")) return rc;
        const int n = e.CV * e.CS * 9;
        hipLaunchKernelGGL(wgrad_edge_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)workspace, dw,
                           e.nsplit, e.CV, e.CS, first ? 1 : 0, accumulate);
        return check_launch("mad_conv_wgrad_edge (reduce)");
    }
    const int ksplit = wgrad_ksplit(MAD_F32, B, H, W, Cin, Cout, ksize, nullptr);
    WgradGenArgs a;
    a.x = x; a.dy = dy; a.partial = (float*)workspace; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.ks = ksize; a.ksplit = ksplit; a.upsample = 0;
    if (x_nchw) { a.xsx = 1; a.xsy = W; a.xsc = (long)H * W; a.xsb = (long)Cin * H * W; }
    else { a.xsc = 1; a.xsx = Cin; a.xsy = (long)W * Cin; a.xsb = (long)H * W * Cin; }
    if (dy_nchw) { a.ysx = 1; a.ysy = W; a.ysc = (long)H * W; a.ysb = (long)Cout * H * W; }
    else { a.ysc = 1; a.ysx = Cout; a.ysy = (long)W * Cout; a.ysb = (long)H * W * Cout; }
    const dim3 grid(Cout * nt * ksplit);
    if (x_dtype == MAD_F32 && dy_dtype == MAD_F32) hipLaunchKernelGGL((wgrad_generic_kernel<float, float>), grid, dim3(256), 0, s, a);
    else if (x_dtype == MAD_F32) hipLaunchKernelGGL((wgrad_generic_kernel<float, bf16_t>), grid, dim3(256), 0, s, a);
    else if (dy_dtype == MAD_F32) hipLaunchKernelGGL((wgrad_generic_kernel<bf16_t, float>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad_generic_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, a);
    if (int rc = check_launch("mad_conv_wgrad_edge")) return rc;
    hipLaunchKernelGGL(wgrad_reduce_oihw_kernel, dim3(Cout, (Cin + 31) / 32), dim3(288), 0, s,
                       (const float*)workspace, dw, Cout, ksplit, accumulate, nt, Cin, (const float*)nullptr,
                       (float*)nullptr, 0);
    return check_launch("mad_conv_wgrad_edge (reduce)");
}
