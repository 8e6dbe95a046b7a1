// Convolutions of the denoiser.
//
//   conv_igemm_kernel  3x3 / 1x1 stride-1 convolution as an implicit GEMM on MFMA:
//                      out[m][n] = sum_k A[m][k] * Wt[n][k],  m = (b,oy,ox), k = (ky,kx,c).
//                      Block tile 128 (pixels) x BN (output channels), K-tile = 128 bytes of
//                      channels of one tap.  Both operand tiles are brought in by asynchronous
//                      16-byte global->LDS copies (one tile in flight while the previous one is
//                      multiplied); padding pixels and rows past M read a zero page.  LDS rows are
//                      128 bytes, XOR-swizzled on the source side so the ds_read_b128 fragment
//                      reads are bank-conflict free.  The epilogue stages the fp32 tile through
//                      LDS and writes whole 16-byte chunks with bias and residual added.
//   conv_in_kernel     first layer (Cin = 3, NCHW fp32 in, NHWC out): direct, bandwidth bound.
//   conv_out_kernel    last layer (Cout = 3, NHWC in, NCHW fp32 out): direct, bandwidth bound.
#include "mma.h"
#include <algorithm>
#include <cstdlib>
#include <string>

namespace mad {

struct ConvArgs {
    const void* x1; const void* x2; const void* w; const float* bias; const void* residual; void* y;
    int C1, C2, B, H, W, Cout, ksize, upsample, res_mode;
    int M;          // B*H*W
    int Hs, Ws;     // source spatial size (H/2, W/2 when upsample)
    int mtiles, ntiles;
    float* stats;   // optional (M/64, Cout, 2): per-channel sum / sum of squares of each 64-pixel run of the output
    // split-K form: blockIdx.y takes one of `ksplit` contiguous shares of the K-tiles and writes its fp32 partial
    // tile to partial[share][M][Cout]; conv_splitk_finish_kernel adds the shares in order and does the epilogue
    float* partial; int ksplit;
    // 1x1, no resampling, 32-bit byte offsets suffice: the activation copies are addressed as a scalar base plus
    // a per-lane offset that never changes (no vector address arithmetic per K-tile)
    int fast1;
};

constexpr int ROWB = 128;   // bytes per LDS row = one K-tile of one operand row

template <int BM, int BN, int NST> constexpr int conv_lds_bytes() {
    constexpr int stage = NST * (BM + BN) * ROWB;
    constexpr int epi = BM * (BN + 4) * 4;
    return stage > epi ? stage : epi;
}

// BM x BN block tile, NST K-tile buffers in a ring.  Tile kt+NST-1 is requested while tile kt is
// multiplied; a wave waits only until its own copies of tile kt have landed (they retire in issue
// order), and the barrier that follows both publishes tile kt and frees the buffer of tile kt-1.
template <typename T, int BM, int BN, int NST>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
    constexpr int VEC = Elem<T>::VEC;             // elements per 16-byte chunk
    constexpr int CK = ROWB / (int)sizeof(T);     // channels per K-tile
    constexpr int TM = BM / 64, TN = BN / 64;     // 32x32 accumulator tiles per wave
    constexpr int APW = BM / 32, BPW = BN / 32;   // 1-KiB copy instructions per wave per K-tile
    constexpr int PPW = APW + BPW;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static_assert((NST - 2) * PPW < 64, "outstanding copies must fit the vmcnt field");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: blocks id, id+8, ... share an XCD (and its L2); give each XCD a
    // contiguous range of tiles so neighbouring pixel tiles (shared halo rows) and the n-tiles of
    // one pixel tile hit the same L2.
    const int nwg = a.mtiles * a.ntiles;
    const int id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int mt = logical / a.ntiles, nt = logical % a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int Ctot = a.C1 + a.C2;
    const int cchunks = Ctot / CK;
    const int ntaps = a.ksize * a.ksize;
    const int nk_all = ntaps * cchunks;
    const int kt_lo = a.partial ? (int)((long long)nk_all * blockIdx.y / a.ksplit) : 0;
    const int nk = (a.partial ? (int)((long long)nk_all * (blockIdx.y + 1) / a.ksplit) : nk_all) - kt_lo;
    const int pad = a.ksize >> 1;
    const int HW = a.H * a.W;

    // ---- per-thread copy assignment: instruction j of this wave covers rows (j*4+wave)*8 .. +7
    const int lrow = lane >> 3, pchunk = lane & 7;
    int a_oy[APW], a_ox[APW], a_b[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int m = m0 + (j * 4 + wave) * 8 + lrow;
        if (m < a.M) {
            const int b = m / HW, rem = m - b * HW;
            a_b[j] = b; a_oy[j] = rem / a.W; a_ox[j] = rem - a_oy[j] * a.W;
        } else {
            a_b[j] = -1; a_oy[j] = 0; a_ox[j] = 0;
        }
    }

    // Addressing without per-copy vector arithmetic.  Row r = (j*4+wave)*8 + lrow of a tile sits 32*j rows below
    // row wave*8 + lrow, and (r >> 1) & 7 does not depend on j: a copy is a scalar base (tile, tap, channel chunk,
    // j) plus ONE per-lane byte offset.  Always so for the weight rows; for the activation rows when the layer is
    // 1x1 without resampling (pixel m reads pixel m) and the tile is a whole one.
    const int lchunk_f = pchunk ^ (((wave & 1) * 4 + (lrow >> 1)) & 7);
    const unsigned woff = (unsigned)(((wave * 8 + lrow) * ntaps * Ctot + lchunk_f * VEC) * (int)sizeof(T));
    const bool fastA = a.fast1 && m0 + BM <= a.M;
    const unsigned aoff1 = (unsigned)(((m0 + wave * 8 + lrow) * a.C1 + lchunk_f * VEC) * (int)sizeof(T));
    const unsigned aoff2 = (unsigned)(((m0 + wave * 8 + lrow) * a.C2 + lchunk_f * VEC) * (int)sizeof(T));

    const T* apj[APW];       // general form: this lane's source rows for the (tap, source) pair a_key
    int a_key = -1;

    const unsigned smem_base = lds_addr(smem);
    auto stage = [&](int kt, int buf) {
        const int tap = kt / cchunks, cc = kt - tap * cchunks;
        const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
        const int c0 = cc * CK;
        const T* src; int cs, coff;
        if (c0 < a.C1) { src = (const T*)a.x1; cs = a.C1; coff = c0; }
        else { src = (const T*)a.x2; cs = a.C2; coff = c0 - a.C1; }
        const unsigned abuf = smem_base + buf * (A_BYTES + B_BYTES);
        const unsigned bbuf = abuf + A_BYTES;
        if (fastA) {
            const unsigned ao = c0 < a.C1 ? aoff1 : aoff2;
#pragma unroll
            for (int j = 0; j < APW; ++j)
                glds16s(src + coff + (size_t)(j * 32) * cs, ao, abuf + (j * 4 + wave) * 8 * ROWB);
        } else {
            // the row pointers of a (tap, source) pair are worked out once and serve all its channel chunks
            const int key = tap * 2 + (c0 < a.C1 ? 0 : 1);
            if (key != a_key) {
                a_key = key;
#pragma unroll
                for (int j = 0; j < APW; ++j) {
                    const int iy = a_oy[j] + ky - pad, ix = a_ox[j] + kx - pad;
                    const bool ok = a_b[j] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    // clamp so the address is always in range, then select the zero page for padding
                    const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
                    const int sy = a.upsample ? cy >> 1 : cy, sx = a.upsample ? cx >> 1 : cx;
                    const int bb = max(a_b[j], 0);
                    const T* p = src + ((size_t)((bb * a.Hs + sy) * a.Ws + sx)) * cs + lchunk_f * VEC;
                    apj[j] = ok ? p : reinterpret_cast<const T*>(g_zero_page) + lchunk_f * VEC;
                }
            }
#pragma unroll
            for (int j = 0; j < APW; ++j) glds16(apj[j] + coff, abuf + (j * 4 + wave) * 8 * ROWB);
        }
        const T* wb = (const T*)a.w + ((size_t)n0 * ntaps + tap) * Ctot + c0;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            glds16s(wb + (size_t)(j * 32) * ntaps * Ctot, woff, bbuf + (j * 4 + wave) * 8 * ROWB);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;
    for (int p = 0; p < NST - 1 && p < nk; ++p) stage(kt_lo + p, p);
    int buf = 0, nbuf = NST - 1;                 // buffer of tile kt, buffer tile kt+NST-1 goes to
    for (int kt = 0; kt < nk; ++kt) {
        // tiles kt+1 .. kt+NST-2 may still be in flight; near the end fewer were requested
        if (kt + NST - 2 < nk) dma_wait_upto<(NST - 2) * PPW>();
        else dma_wait_all();
        __syncthreads();
        if (kt + NST - 1 < nk) stage(kt_lo + kt + NST - 1, nbuf);
        const unsigned char* abuf = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* bbuf = abuf + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (BM / 2) + i * 32 + frow;
                af[i] = *reinterpret_cast<const uint4*>(abuf + r * ROWB + (((2 * kk + fh) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (BN / 2) + j * 32 + frow;
                bfr[j] = *reinterpret_cast<const uint4*>(bbuf + r * ROWB + (((2 * kk + fh) ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], af[i], bfr[j]);
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }

    if (a.partial) {
        // split-K: the raw fp32 accumulators go straight to this share's slab (128 contiguous bytes per row
        // and half wave); bias, residual, rounding and statistics belong to the finishing kernel
        float* dst = a.partial + (size_t)blockIdx.y * a.M * a.Cout;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + acc_row(e, lane);
                    const int c = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
                    if (m < a.M) dst[(size_t)m * a.Cout + c] = acc[i][j][e];
                }
        return;
    }

    // ---- epilogue: fp32 tile -> LDS -> coalesced 16-byte stores with bias / residual
    constexpr int CPR = BN / VEC;            // 16-byte output chunks per tile row
    constexpr int RPP = 256 / CPR;           // rows per pass
    constexpr int NPASS = BM / RPP;
    const int er = tid / CPR, ec = (tid % CPR) * VEC;
    const T* res = (const T*)a.residual;
    T* y = (T*)a.y;
    // residual rows are requested before the tile hand-over so their latency overlaps it
    constexpr int NPRE = NPASS <= 8 ? NPASS : 8;
    uint4 rpre[NPRE];
    if (a.res_mode == MAD_RES_SAME) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int m = min(m0 + er + k * RPP, a.M - 1);
            rpre[k] = *reinterpret_cast<const uint4*>(res + (size_t)m * a.Cout + n0 + ec);
        }
    }
    __syncthreads();
    float* ctile = reinterpret_cast<float*>(smem);
    constexpr int CLD = BN + 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * (BM / 2) + i * 32 + acc_row(e, lane);
                const int c = wn * (BN / 2) + j * 32 + (lane & 31);
                ctile[r * CLD + c] = acc[i][j][e];
            }
    __syncthreads();
    float bias_v[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) bias_v[i] = a.bias ? a.bias[n0 + ec + i] : 0.0f;
    constexpr int HALVES = BM / 64;          // statistics are kept per 64-pixel run
    float ssum[HALVES][VEC], ssq[HALVES][VEC];
#pragma unroll
    for (int h = 0; h < HALVES; ++h)
#pragma unroll
        for (int i = 0; i < VEC; ++i) { ssum[h][i] = 0.0f; ssq[h][i] = 0.0f; }
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int r = er + k * RPP;
        const int m = m0 + r;
        if (m >= a.M) break;
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(ctile + r * CLD + ec + i);
            v[i] = t.x + bias_v[i]; v[i + 1] = t.y + bias_v[i + 1];
            v[i + 2] = t.z + bias_v[i + 2]; v[i + 3] = t.w + bias_v[i + 3];
        }
        if (a.res_mode == MAD_RES_SAME) {
            Vec16<T> rv;
            if (k < NPRE) rv.raw = rpre[k < NPRE ? k : 0];
            else rv.load(res + (size_t)m * a.Cout + n0 + ec);
            float f[VEC]; rv.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] += f[i];
        } else if (a.res_mode == MAD_RES_UP) {
            const int b = m / HW, rem = m - b * HW, oy = rem / a.W, ox = rem - oy * a.W;
            const size_t rp = ((size_t)b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1);
            Vec16<T> rv; rv.load(res + rp * a.Cout + n0 + ec);
            float f[VEC]; rv.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] += f[i];
        } else if (a.res_mode == MAD_RES_DOWN) {
            const int b = m / HW, rem = m - b * HW, oy = rem / a.W, ox = rem - oy * a.W;
            float s[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) s[i] = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t rp = ((size_t)b * (2 * a.H) + (2 * oy + dy)) * (2 * a.W) + (2 * ox + dx);
                    Vec16<T> rv; rv.load(res + rp * a.Cout + n0 + ec);
                    float f[VEC]; rv.unpack(f);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) s[i] += f[i];
                }
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = fmaf(s[i], 0.25f, v[i]);
        }
        Vec16<T> o; o.pack(v);
        o.store(y + (size_t)m * a.Cout + n0 + ec);
        if (a.stats) {
            float f[VEC]; o.unpack(f);              // statistics of the values as stored
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                ssum[k / (NPASS / HALVES)][i] += f[i];
                ssq[k / (NPASS / HALVES)][i] = fmaf(f[i], f[i], ssq[k / (NPASS / HALVES)][i]);
            }
        }
    }
    if (a.stats) {
        // reduce the RPP row groups per channel in a fixed order, one 64-pixel run at a time
        float* red = reinterpret_cast<float*>(smem);      // [RPP][BN][2]
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                red[(er * BN + ec + i) * 2] = ssum[h][i];
                red[(er * BN + ec + i) * 2 + 1] = ssq[h][i];
            }
            __syncthreads();
            if (tid < 2 * BN && m0 + h * 64 < a.M) {
                const int c = tid >> 1, which = tid & 1;
                float t = 0.0f;
                for (int g = 0; g < RPP; ++g) t += red[(g * BN + c) * 2 + which];
                a.stats[((size_t)(m0 / 64 + h) * a.Cout + n0 + c) * 2 + which] = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------- split-K finish
// grid (M/64 rounded up, Cout/64), 256 threads: one 64-pixel run x 64 output channels.  Adds the ksplit fp32
// partial slabs in share order (fixed: the result does not depend on which workgroup finished first), then bias,
// residual, rounding to T, the store, and the run's per-channel sum / sum of squares, as the one-pass epilogue.
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(ConvArgs a) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CPR = 64 / VEC, RPP = 256 / CPR, NPASS = 64 / RPP;
    __shared__ float red[RPP * 64 * 2];
    const int tid = threadIdx.x;
    const int er = tid / CPR, ec = (tid % CPR) * VEC;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int HW = a.H * a.W;
    const T* res = (const T*)a.residual;
    T* y = (T*)a.y;
    const size_t slab = (size_t)a.M * a.Cout;
    float bias_v[VEC], ssum[VEC], ssq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { bias_v[i] = a.bias ? a.bias[n0 + ec + i] : 0.0f; ssum[i] = 0.0f; ssq[i] = 0.0f; }
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int m = m0 + er + k * RPP;
        if (m >= a.M) break;
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = 0.0f;
        const float* p = a.partial + (size_t)m * a.Cout + n0 + ec;
        for (int sidx = 0; sidx < a.ksplit; ++sidx) {
#pragma unroll
            for (int i = 0; i < VEC; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(p + sidx * slab + i);
                v[i] += t.x; v[i + 1] += t.y; v[i + 2] += t.z; v[i + 3] += t.w;
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] += bias_v[i];
        if (a.res_mode == MAD_RES_SAME) {
            Vec16<T> rv; rv.load(res + (size_t)m * a.Cout + n0 + ec);
            float f[VEC]; rv.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] += f[i];
        } else if (a.res_mode == MAD_RES_UP) {
            const int b = m / HW, rem = m - b * HW, oy = rem / a.W, ox = rem - oy * a.W;
            const size_t rp = ((size_t)b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1);
            Vec16<T> rv; rv.load(res + rp * a.Cout + n0 + ec);
            float f[VEC]; rv.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] += f[i];
        } else if (a.res_mode == MAD_RES_DOWN) {
            const int b = m / HW, rem = m - b * HW, oy = rem / a.W, ox = rem - oy * a.W;
            float s[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) s[i] = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t rp = ((size_t)b * (2 * a.H) + (2 * oy + dy)) * (2 * a.W) + (2 * ox + dx);
                    Vec16<T> rv; rv.load(res + rp * a.Cout + n0 + ec);
                    float f[VEC]; rv.unpack(f);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) s[i] += f[i];
                }
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = fmaf(s[i], 0.25f, v[i]);
        }
        Vec16<T> o; o.pack(v);
        o.store(y + (size_t)m * a.Cout + n0 + ec);
        if (a.stats) {
            float f[VEC]; o.unpack(f);              // statistics of the values as stored
#pragma unroll
            for (int i = 0; i < VEC; ++i) { ssum[i] += f[i]; ssq[i] = fmaf(f[i], f[i], ssq[i]); }
        }
    }
    if (a.stats) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[(er * 64 + ec + i) * 2] = ssum[i];
            red[(er * 64 + ec + i) * 2 + 1] = ssq[i];
        }
        __syncthreads();
        if (tid < 128) {
            const int c = tid >> 1, which = tid & 1;
            float t = 0.0f;
            for (int g = 0; g < RPP; ++g) t += red[(g * 64 + c) * 2 + which];
            a.stats[((size_t)blockIdx.x * a.Cout + n0 + c) * 2 + which] = t;
        }
    }
}

// --------------------------------------------------------------------------- first layer
// 64 consecutive pixels per block.  Thread (p = tid&63, cg = tid>>6) computes 32-channel groups
// cg, cg+4, ... of pixel p from its 27 input taps held in registers; weights sit in LDS transposed
// ([k][Cout]) so four output channels come from one broadcast ds_read_b128.  The output tile is
// staged in LDS and written out as whole contiguous rows.
template <typename T, int Cin>
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y, int B,
                                                       int H, int W, int Cout) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int K = Cin * 9;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_in[];
    float* wT = reinterpret_cast<float*>(smem_in);                    // [K][Cout]
    float* bl = wT + K * Cout;                                        // [Cout]
    T* otile = reinterpret_cast<T*>(bl + Cout);                       // [64][Cout]
    for (int i = threadIdx.x; i < Cout * K; i += 256) { const int n = i / K, k = i - n * K; wT[k * Cout + n] = w[i]; }
    for (int i = threadIdx.x; i < Cout; i += 256) bl[i] = bias[i];
    __syncthreads();
    const int HW = H * W;
    const size_t total = (size_t)B * HW;
    const size_t m0 = (size_t)blockIdx.x * 64;
    const int p = threadIdx.x & 63;
    const size_t m = m0 + p;
    float in[K];
    if (m < total) {
        const int b = (int)(m / HW), rem = (int)(m % HW), oy = rem / W, ox = rem % W;
#pragma unroll
        for (int c = 0; c < Cin; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = oy + ky - 1, ix = ox + kx - 1;
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    in[c * 9 + ky * 3 + kx] = ok ? x[((size_t)(b * Cin + c) * H + iy) * W + ix] : 0.0f;
                }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) in[k] = 0.0f;
    }
    for (int n0 = (threadIdx.x >> 6) * 32; n0 < Cout; n0 += 128) {
#pragma unroll
        for (int g = 0; g < 32; g += VEC) {
            float v[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = bl[n0 + g + i];
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int i = 0; i < VEC; i += 4) {
                    const float4 ww = *reinterpret_cast<const float4*>(wT + k * Cout + n0 + g + i);
                    v[i] = fmaf(in[k], ww.x, v[i]); v[i + 1] = fmaf(in[k], ww.y, v[i + 1]);
                    v[i + 2] = fmaf(in[k], ww.z, v[i + 2]); v[i + 3] = fmaf(in[k], ww.w, v[i + 3]);
                }
            }
            Vec16<T> o; o.pack(v);
            o.store(otile + p * Cout + n0 + g);
        }
    }
    __syncthreads();
    // 64 pixels x Cout channels are contiguous in NHWC: copy the tile out in 16-byte pieces
    const int chunks = 64 * Cout / VEC;
    T* dst = y + m0 * Cout;
    for (int i = threadIdx.x; i < chunks; i += 256) {
        if (m0 + (size_t)(i * VEC) / Cout < total)
            *reinterpret_cast<uint4*>(dst + (size_t)i * VEC) = *reinterpret_cast<const uint4*>(otile + (size_t)i * VEC);
    }
}

// (hi, lo) bf16 pairs of two floats: x = hi + lo up to 2^-17 |x| (lo = the rounded remainder of the rounded x)
__device__ __forceinline__ void split_bf16x2(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}

// First layer.  fp32 activations: the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, K = 27 padded to 28).  bf16
// activations: the image and the weights are both fp32, and on the fp32 pipe (1/16 of the bf16 rate) the layer
// would be bound by it (14.5 GFLOP at 256x256, batch 32: 92 us at best); instead both operands are split into
// bf16 (hi, lo) pairs and the product is taken as hi*hi + hi*lo + lo*hi on the bf16 pipe with fp32
// accumulation (K = 27 padded to 32: 24 instead of 56 four-times-longer instructions per tile).  What is
// dropped is lo*lo and the third-order remainders, 2^-16 of the product against the 2^-9 of the bf16 rounding
// of the result.  A workgroup walks 16x8-pixel tiles with a grid stride; the
// weights ([Cout][29] fp32, odd pitch = conflict-free operand reads) are loaded to LDS once per
// workgroup.  Per tile: the 3 x 10 x 18 input halo goes to LDS; each wave owns two tile rows (32
// pixels) x all Cout channels, reads its A operand straight from the halo at the tap's offset, and
// writes its 32 x Cout result through a wave-private LDS slab as whole 16-byte chunks (NHWC rows of
// one tile row are contiguous in memory).  With `stats` the per-channel sum / sum of squares of the stored
// values of every tile go to stats[b][tile][Cout][2] (the next GroupNorm's partial sums): lane-local over a
// wave's 32 pixels, then the four waves are added in order one tile later, behind the barrier the next tile's
// halo needs anyway.
template <typename T, int TN>
__global__ __launch_bounds__(256) void conv_in_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ y,
                                                            float* __restrict__ stats, int B, int H, int W) {
    constexpr int Cout = TN * 32, WP = 29, VEC = Elem<T>::VEC;
    constexpr int HALO_N = 3 * 10 * 18;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_m[];
    float* wl = reinterpret_cast<float*>(smem_m);                  // [Cout][WP]
    float* halo = wl + Cout * 32;                                  // [3][10][18]; the weight region holds Cout x 128 bytes
    T* oslab = reinterpret_cast<T*>(halo + ((HALO_N + 3) & ~3));   // [4 waves][32][Cout]
    float* red = reinterpret_cast<float*>(oslab + 4 * 32 * Cout);  // [4 waves][Cout][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (sizeof(T) == 2) {
        // bf16 form: the weights as (hi, lo) bf16 pairs, [hi | lo][k half st][channel][32 bytes = k 16 st .. + 15]
        // (k >= 27: zero): a lane's B fragment is one 16-byte read, lanes of a wave cover 2 KiB contiguously
        unsigned* wq = reinterpret_cast<unsigned*>(smem_m);
        for (int i = tid; i < Cout * 16; i += 256) {      // one k pair each
            const int n = i >> 4, kp = i & 15, k = 2 * kp;
            unsigned hi, lo;
            split_bf16x2(k < 27 ? w[n * 27 + k] : 0.0f, k + 1 < 27 ? w[n * 27 + k + 1] : 0.0f, hi, lo);
            const int st = kp >> 3, word = kp & 7;
            wq[((0 * 2 + st) * Cout + n) * 8 + word] = hi;
            wq[((1 * 2 + st) * Cout + n) * 8 + word] = lo;
        }
    } else {
        for (int i = tid; i < Cout * WP; i += 256) {
            const int n = i / WP, k = i - n * WP;
            wl[i] = k < 27 ? w[n * 27 + k] : 0.0f;
        }
    }
    const int row = lane & 31, kh = lane >> 5;
    const int pixoff = (wave * 2 + (row >> 4)) * 18 + (row & 15);
    int koff[14];                            // halo offset of this lane's k = 2s + kh
#pragma unroll
    for (int s2 = 0; s2 < 14; ++s2) {
        const int k = 2 * s2 + kh;
        const int c = k / 9, r = k - c * 9, ky = r / 3, kx = r - ky * 3;
        koff[s2] = k < 27 ? c * 180 + ky * 18 + kx : 0;           // k = 27 meets a zero weight
    }
    float bcol[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) bcol[jn] = bias[jn * 32 + row];
    // bf16 form: halo offsets of this lane's sixteen k = 16 st + 8 kh + j (k >= 27 meets a zero weight: any finite
    // halo value will do)
    int koff16[sizeof(T) == 2 ? 16 : 1];
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = 16 * (i >> 3) + 8 * kh + (i & 7);
            const int c = k / 9, r = k - c * 9, ky = r / 3, kx = r - ky * 3;
            koff16[i] = k < 27 ? c * 180 + ky * 18 + kx : 0;
        }
    }
    const int tiles_x = W / 16, tiles_y = H / 8;
    const int ntiles = B * tiles_x * tiles_y;
    T* const my_slab = oslab + wave * 32 * Cout;
    auto flush_stats = [&](int tile) {       // the four waves' sums of `tile`, added in wave order
        for (int i = tid; i < 2 * Cout; i += 256)
            stats[(size_t)tile * 2 * Cout + i] = ((red[i] + red[2 * Cout + i]) + red[4 * Cout + i]) + red[6 * Cout + i];
    };
    int prev = -1;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (tiles_x * tiles_y), tt = t - b * tiles_x * tiles_y;
        const int ty0 = (tt / tiles_x) * 8, tx0 = (tt % tiles_x) * 16;
        __syncthreads();                     // previous tile's halo reads are done (and the weights are in)
        if (stats && prev >= 0) flush_stats(prev);
        prev = t;
        for (int i = tid; i < HALO_N; i += 256) {
            const int c = i / 180, r = i - c * 180, hy = r / 18, hx = r - hy * 18;
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            halo[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((size_t)(b * 3 + c) * H + iy) * W + ix] : 0.0f;
        }
        __syncthreads();
        f32x16 acc[TN];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jn][e] = bcol[jn];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uint4 ah, al;            // this lane's pixel, k = 16 st + 8 kh .. + 7, as (hi, lo) bf16
                unsigned* ahp = reinterpret_cast<unsigned*>(&ah);
                unsigned* alp = reinterpret_cast<unsigned*>(&al);
#pragma unroll
                for (int j = 0; j < 8; j += 2)
                    split_bf16x2(halo[pixoff + koff16[st * 8 + j]], halo[pixoff + koff16[st * 8 + j + 1]], ahp[j / 2], alp[j / 2]);
                const unsigned char* wq = smem_m;
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    const uint4 wh = *reinterpret_cast<const uint4*>(wq + (((0 * 2 + st) * Cout + jn * 32 + row) * 8 + kh * 4) * 4);
                    const uint4 wlo = *reinterpret_cast<const uint4*>(wq + (((1 * 2 + st) * Cout + jn * 32 + row) * 8 + kh * 4) * 4);
                    mma_chunk<bf16_t>(acc[jn], ah, wh);
                    mma_chunk<bf16_t>(acc[jn], ah, wlo);
                    mma_chunk<bf16_t>(acc[jn], al, wh);
                }
            }
        } else {
#pragma unroll
        for (int s2 = 0; s2 < 14; ++s2) {
            const float av = halo[pixoff + koff[s2]];
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                const float bv = wl[(jn * 32 + row) * WP + 2 * s2 + kh];
                acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[jn], 0, 0, 0);
            }
        }
        }
        // accumulators -> wave-private slab (pixel-major) -> 16-byte global stores
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                my_slab[acc_row(e, lane) * Cout + jn * 32 + row] = Elem<T>::store(acc[jn][e]);
        if (stats) {
            // this lane's channel jn*32+row over the 16 pixels of its half, of the values as stored
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                float sm = 0.0f, sq = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float f = Elem<T>::load(Elem<T>::store(acc[jn][e]));
                    sm += f; sq = fmaf(f, f, sq);
                }
                sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
                if (kh == 0) {
                    red[(wave * Cout + jn * 32 + row) * 2] = sm;
                    red[(wave * Cout + jn * 32 + row) * 2 + 1] = sq;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the slab is private to this wave
        __builtin_amdgcn_wave_barrier();
        constexpr int CHUNKS = 32 * Cout / VEC;           // 16-byte chunks of the slab
#pragma unroll
        for (int i = 0; i < CHUNKS / 64; ++i) {
            const int ch = i * 64 + lane;
            const int p = ch / (Cout / VEC), cv = ch - p * (Cout / VEC);
            const int oy = ty0 + wave * 2 + (p >> 4), ox = tx0 + (p & 15);
            const uint4 v = *reinterpret_cast<const uint4*>(my_slab + p * Cout + cv * VEC);
            *reinterpret_cast<uint4*>(y + (((size_t)b * H + oy) * W + ox) * Cout + cv * VEC) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (stats && prev >= 0) {
        __syncthreads();
        flush_stats(prev);
    }
}

// --------------------------------------------------------------------------- last layer
// One thread per output pixel, all (<= 4) output channels; weights (Cout,3,3,Cin) fp32 in LDS are
// read as wave-wide broadcasts.
template <typename T>
__global__ __launch_bounds__(256) void conv_out_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B,
                                                        int Cin, int H, int W, int Cout) {
    constexpr int VEC = Elem<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [Cout][9][Cin]
    for (int i = threadIdx.x; i < Cout * 9 * Cin; i += 256) wl[i] = w[i];
    __syncthreads();
    const int HW = H * W;
    const size_t m = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= (size_t)B * HW) return;
    const int b = (int)(m / HW), rem = (int)(m % HW), oy = rem / W, ox = rem % W;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const T* px = x + ((size_t)(b * H + iy) * W + ix) * Cin;
        for (int c = 0; c < Cin; c += VEC) {
            Vec16<T> v; v.load(px + c);
            float f[VEC]; v.unpack(f);
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                if (co < Cout) {
                    const float* ww = wl + (co * 9 + tap) * Cin + c;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[co] = fmaf(f[i], ww[i], acc[co]);
                }
            }
        }
    }
    for (int co = 0; co < Cout; ++co) y[((size_t)(b * Cout + co) * H + oy) * W + ox] = acc[co] + bias[co];
}

template <typename T, int BM, int BN, int NST> static int launch_igemm(ConvArgs a, hipStream_t s) {
    constexpr int lds = conv_lds_bytes<BM, BN, NST>();
    static int lds_dev = -1;
    MAD_REQUIRE(raise_lds_limit(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, NST>), lds, &lds_dev),
                "mad_conv2d: cannot raise the dynamic LDS limit to %d", lds);
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = a.Cout / BN;
    const int shares = a.partial ? a.ksplit : 1;
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, NST>), dim3(a.mtiles * a.ntiles, shares), dim3(256), lds, s, a);
    if (a.partial) {
        hipLaunchKernelGGL((conv_splitk_finish_kernel<T>), dim3((a.M + 63) / 64, a.Cout / 64), dim3(256), 0, s, a);
    }
    return check_launch("mad_conv2d");
}

// Split-K plan.  A layer whose 128-row tiles do not put two workgroups on every CU, but whose K is long (the 3x3
// layers of the 8x8 stage: 4096 pixels x 768 channels with K = 6912 ...), is cut along K into 2, 4 or 8 shares:
// the smallest count that puts two workgroups on every CU, every share keeping at least 8 K-tiles.  (Measured
// on the 4096 x 768 x 6912 layer: 122 us in one pass, 94 / 85 / 91 us with 2 / 4 / 8 shares; odd counts leave a
// partial last round of workgroups and are slower than either neighbour.)  Returns 1 when the one-pass kernel
// is the better choice: short K (1x1 layers), or enough tiles.
static int plan_ksplit(int M, int Cout, int Cin, int ksize, int dtype) {
    if (const char* e = getenv("MAD_IGEMM_KSPLIT")) return std::max(1, atoi(e));     // tuning aid
    const int ck = dtype == MAD_BF16 ? 64 : 32;
    const int nk = ksize * ksize * (Cin / ck);
    const int tiles = ((M + 127) / 128) * (Cout % 128 == 0 ? Cout / 128 : Cout / 64);
    if (tiles >= 512 || nk < 32) return 1;
    int s = 2;
    while (tiles * s < 512 && s < 8) s *= 2;
    while (s > 1 && nk / s < 8) s /= 2;
    return s;
}

// Tile choice: 128x128 tiles when they fill the chip at least once; else 64x128 if those do; otherwise
// 64x64 tiles with a deeper ring, which quadruples the number of workgroups of the small late-stage layers.
template <typename T> static int dispatch_igemm(const ConvArgs& a, hipStream_t s) {
    const char* force = getenv("MAD_IGEMM_TILE");      // tuning aid: "128x128x2", "64x64x4", ...
    const std::string f = force ? force : "";
    if (a.partial && f.empty())
        return a.Cout % 128 == 0 ? launch_igemm<T, 128, 128, 2>(a, s) : launch_igemm<T, 128, 64, 2>(a, s);
    const int big_blocks = ((a.M + 127) / 128) * (a.Cout / 128);
    const bool wide = a.Cout % 128 == 0;
    if (f == "128x128x2" && wide) return launch_igemm<T, 128, 128, 2>(a, s);
    if (f == "128x128x3" && wide) return launch_igemm<T, 128, 128, 3>(a, s);
    if (f == "128x64x3") return launch_igemm<T, 128, 64, 3>(a, s);
    if (f == "128x64x2") return launch_igemm<T, 128, 64, 2>(a, s);
    if (f == "64x64x4") return launch_igemm<T, 64, 64, 4>(a, s);
    if (f == "64x128x3" && wide) return launch_igemm<T, 64, 128, 3>(a, s);
    if (wide && big_blocks >= 256) return launch_igemm<T, 128, 128, 2>(a, s);
    if (wide && ((a.M + 63) / 64) * (a.Cout / 128) >= 256) return launch_igemm<T, 64, 128, 3>(a, s);
    if (!wide && ((a.M + 127) / 128) * (a.Cout / 64) >= 256) return launch_igemm<T, 128, 64, 2>(a, s);
    return launch_igemm<T, 64, 64, 4>(a, s);
}

}  // namespace mad

using namespace mad;

static int conv2d_impl(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias,
                       const void* residual, int res_mode, void* y, float* stats, int dtype, int B, int H, int W,
                       int Cout, int ksize, int upsample, void* stream, void* workspace = nullptr,
                       size_t workspace_bytes = 0) {
    MAD_REQUIRE(x1 && w && y, "mad_conv2d: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv2d: bad dtype %d", dtype);
    MAD_REQUIRE(ksize == 1 || ksize == 3, "mad_conv2d: ksize must be 1 or 3, got %d", ksize);
    MAD_REQUIRE(C1 > 0 && C1 % 64 == 0 && C2 >= 0 && C2 % 64 == 0 && (C2 == 0 || x2),
                "mad_conv2d: C1=%d, C2=%d must be multiples of 64 (x2 given when C2 > 0)", C1, C2);
    MAD_REQUIRE(Cout > 0 && Cout % 64 == 0, "mad_conv2d: Cout=%d must be a multiple of 64", Cout);
    MAD_REQUIRE((size_t)(C1 > C2 ? C1 : C2) * (dtype == MAD_BF16 ? 2 : 4) <= (size_t)MAD_ZERO_PAGE_BYTES,
                "mad_conv2d: %d channels per source exceed the zero page", C1 > C2 ? C1 : C2);
    MAD_REQUIRE(B > 0 && H > 0 && W > 0 && (size_t)B * H * W < (1u << 31), "mad_conv2d: bad shape B=%d H=%d W=%d", B, H, W);
    MAD_REQUIRE(!upsample || (H % 2 == 0 && W % 2 == 0), "mad_conv2d: upsample needs even H, W");
    MAD_REQUIRE(res_mode >= MAD_RES_NONE && res_mode <= MAD_RES_DOWN && (res_mode == MAD_RES_NONE || residual),
                "mad_conv2d: bad residual mode %d", res_mode);
    MAD_REQUIRE(res_mode != MAD_RES_UP || (H % 2 == 0 && W % 2 == 0), "mad_conv2d: RES_UP needs even H, W");
    ConvArgs a;
    a.x1 = x1; a.x2 = C2 ? x2 : x1; a.w = w; a.bias = bias; a.residual = residual; a.y = y;
    a.C1 = C1; a.C2 = C2; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.ksize = ksize; a.upsample = upsample;
    a.res_mode = res_mode; a.M = B * H * W; a.stats = stats;
    a.partial = nullptr; a.ksplit = 1;
    a.fast1 = ksize == 1 && !upsample &&
              (size_t)B * H * W * (C1 > C2 ? C1 : C2) * (dtype == MAD_BF16 ? 2 : 4) < ((size_t)1 << 32);
    if (workspace) {
        const int ks = plan_ksplit(a.M, Cout, C1 + C2, ksize, dtype);
        if (ks > 1) {
            MAD_REQUIRE((size_t)ks * a.M * Cout * sizeof(float) <= workspace_bytes,
                        "mad_conv2d_ws: workspace of %zu bytes is below mad_conv2d_workspace_bytes (%zu)",
                        workspace_bytes, (size_t)ks * a.M * Cout * sizeof(float));
            a.partial = (float*)workspace; a.ksplit = ks;
        }
    }
    MAD_REQUIRE(!stats || (H * W) % 64 == 0, "mad_conv2d_stats: H*W=%d must be a multiple of 64", H * W);
    a.Hs = upsample ? H / 2 : H; a.Ws = upsample ? W / 2 : W;
    hipStream_t s = (hipStream_t)stream;
    // 1x1 layers with enough pixels: the 256-pixel-tile GEMM
    if (dtype == MAD_BF16 && ksize == 1 && !upsample && !a.partial && (res_mode == MAD_RES_NONE || res_mode == MAD_RES_SAME) &&
        conv1x1_ok(a.M, C1, C2, Cout))
        return conv1x1(x1, C1, a.x2, C2, w, bias, residual, y, stats, a.M, Cout, s);
    return dtype == MAD_BF16 ? dispatch_igemm<bf16_t>(a, s) : dispatch_igemm<float>(a, s);
}

extern "C" int mad_conv2d(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias,
                          const void* residual, int res_mode, void* y, int dtype, int B, int H, int W, int Cout,
                          int ksize, int upsample, void* stream) {
    return conv2d_impl(x1, C1, x2, C2, w, bias, residual, res_mode, y, nullptr, dtype, B, H, W, Cout, ksize, upsample,
                       stream);
}

extern "C" int mad_conv2d_stats(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias,
                                const void* residual, int res_mode, void* y, float* stats, int dtype, int B, int H,
                                int W, int Cout, int ksize, int upsample, void* stream) {
    MAD_REQUIRE(stats, "mad_conv2d_stats: null statistics pointer");
    return conv2d_impl(x1, C1, x2, C2, w, bias, residual, res_mode, y, stats, dtype, B, H, W, Cout, ksize, upsample,
                       stream);
}

extern "C" size_t mad_conv2d_workspace_bytes(int dtype, int B, int H, int W, int Cin, int Cout, int ksize) {
    if ((dtype != MAD_BF16 && dtype != MAD_F32) || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const int ks = plan_ksplit(B * H * W, Cout, Cin, ksize, dtype);
    return ks > 1 ? (size_t)ks * B * H * W * Cout * sizeof(float) : 0;
}

extern "C" int mad_conv2d_ws(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias,
                             const void* residual, int res_mode, void* y, float* stats, int dtype, int B, int H,
                             int W, int Cout, int ksize, int upsample, void* workspace, size_t workspace_bytes,
                             void* stream) {
    return conv2d_impl(x1, C1, x2, C2, w, bias, residual, res_mode, y, stats, dtype, B, H, W, Cout, ksize, upsample,
                       stream, workspace, workspace_bytes);
}

// Shapes the matrix-pipe form of the first layer covers (the only form that can take statistics).
static bool conv_in_mfma_ok(int dtype, int H, int W, int Cout) {
    const size_t esz = dtype == MAD_BF16 ? 2 : 4;
    const size_t lds_m = ((size_t)Cout * 32 + 544) * sizeof(float) + (size_t)4 * 32 * Cout * esz + (size_t)8 * Cout * 4;
    const int tn = Cout / 32;
    return H % 8 == 0 && W % 16 == 0 && Cout % 32 == 0 && lds_m <= 160 * 1024 &&
           (tn == 1 || tn == 2 || tn == 3 || tn == 4 || tn == 6 || tn == 8);
}

static int conv_in_impl(const float* x, const float* w, const float* bias, void* y, float* stats, int dtype, int B,
                        int Cin, int H, int W, int Cout, void* stream) {
    MAD_REQUIRE(x && w && bias && y, "mad_conv_in: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv_in: bad dtype %d", dtype);
    MAD_REQUIRE(Cin == 3, "mad_conv_in: Cin=%d, only 3 input channels are built", Cin);
    MAD_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cout <= 256, "mad_conv_in: Cout=%d must be a multiple of 32, <= 256", Cout);
    MAD_REQUIRE(B > 0 && H > 0 && W > 0, "mad_conv_in: bad shape");
    const size_t M = (size_t)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    MAD_REQUIRE(!stats || conv_in_mfma_ok(dtype, H, W, Cout),
                "mad_conv_in_stats: H=%d W=%d Cout=%d is outside the tiled form (H %% 8, W %% 16, Cout in 32..256)", H, W,
                Cout);
    if (conv_in_mfma_ok(dtype, H, W, Cout) && (stats || !getenv("MAD_CONV_IN_DIRECT"))) {
        // matrix-pipe form; tiles are walked with a grid stride so the weights are staged once per workgroup
        const int ntiles = B * (H / 8) * (W / 16);
        const dim3 grid((unsigned)std::min(ntiles, 2048));
        const size_t esz = dtype == MAD_BF16 ? 2 : 4;
        const size_t lds_m = ((size_t)Cout * 32 + 544) * sizeof(float) + (size_t)4 * 32 * Cout * esz + (size_t)8 * Cout * 4;
#define MAD_CONV_IN_CASE(TN_)                                                                                     \
    case TN_:                                                                                                     \
        if (dtype == MAD_BF16) {                                                                                  \
            static bool set_b = false;                                                                            \
            if (!set_b) { hipFuncSetAttribute((const void*)conv_in_mfma_kernel<bf16_t, TN_>,                      \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_b = true; } \
            hipLaunchKernelGGL((conv_in_mfma_kernel<bf16_t, TN_>), grid, dim3(256), lds_m, st, x, w, bias,        \
                               (bf16_t*)y, stats, B, H, W);                                                       \
        } else {                                                                                                  \
            static bool set_f = false;                                                                            \
            if (!set_f) { hipFuncSetAttribute((const void*)conv_in_mfma_kernel<float, TN_>,                       \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_f = true; } \
            hipLaunchKernelGGL((conv_in_mfma_kernel<float, TN_>), grid, dim3(256), lds_m, st, x, w, bias,         \
                               (float*)y, stats, B, H, W);                                                        \
        }                                                                                                         \
        return check_launch("mad_conv_in");
        switch (Cout / 32) {
            MAD_CONV_IN_CASE(1) MAD_CONV_IN_CASE(2) MAD_CONV_IN_CASE(3) MAD_CONV_IN_CASE(4)
            MAD_CONV_IN_CASE(6) MAD_CONV_IN_CASE(8)
            default: break;                  // other widths take the direct kernel below
        }
#undef MAD_CONV_IN_CASE
    }
    const size_t lds = ((size_t)Cout * Cin * 9 + Cout) * sizeof(float) + (size_t)64 * Cout * (dtype == MAD_BF16 ? 2 : 4);
    dim3 grid((unsigned)((M + 63) / 64));
    if (dtype == MAD_BF16)
        hipLaunchKernelGGL((conv_in_kernel<bf16_t, 3>), grid, dim3(256), lds, st, x, w, bias,
                           (bf16_t*)y, B, H, W, Cout);
    else
        hipLaunchKernelGGL((conv_in_kernel<float, 3>), grid, dim3(256), lds, st, x, w, bias,
                           (float*)y, B, H, W, Cout);
    return check_launch("mad_conv_in");
}

extern "C" int mad_conv_in(const float* x, const float* w, const float* bias, void* y, int dtype, int B, int Cin,
                           int H, int W, int Cout, void* stream) {
    return conv_in_impl(x, w, bias, y, nullptr, dtype, B, Cin, H, W, Cout, stream);
}

extern "C" int mad_conv_in_stats_ok(int dtype, int H, int W, int Cout) {
    return (dtype == MAD_BF16 || dtype == MAD_F32) && H > 0 && W > 0 && Cout > 0 && conv_in_mfma_ok(dtype, H, W, Cout);
}

extern "C" int mad_conv_in_stats(const float* x, const float* w, const float* bias, void* y, float* stats, int dtype,
                                 int B, int Cin, int H, int W, int Cout, void* stream) {
    MAD_REQUIRE(stats, "mad_conv_in_stats: null statistics pointer");
    return conv_in_impl(x, w, bias, y, stats, dtype, B, Cin, H, W, Cout, stream);
}

extern "C" int mad_conv_out(const void* x, int dtype, const float* w, const float* bias, float* y, int B, int Cin,
                            int H, int W, int Cout, void* stream) {
    MAD_REQUIRE(x && w && bias && y, "mad_conv_out: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv_out: bad dtype %d", dtype);
    MAD_REQUIRE(Cout >= 1 && Cout <= 4, "mad_conv_out: Cout=%d must be 1..4", Cout);
    MAD_REQUIRE(Cin > 0 && Cin % 8 == 0 && (size_t)Cout * 9 * Cin * 4 <= 64 * 1024,
                "mad_conv_out: Cin=%d must be a multiple of 8 with weights fitting 64 KiB of LDS", Cin);
    MAD_REQUIRE(B > 0 && H > 0 && W > 0, "mad_conv_out: bad shape");
    const size_t M = (size_t)B * H * W;
    const size_t lds = (size_t)Cout * 9 * Cin * sizeof(float);
    dim3 grid((unsigned)((M + 255) / 256));
    if (dtype == MAD_BF16)
        hipLaunchKernelGGL(conv_out_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x, w,
                           bias, y, B, Cin, H, W, Cout);
    else
        hipLaunchKernelGGL(conv_out_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, (const float*)x, w, bias,
                           y, B, Cin, H, W, Cout);
    return check_launch("mad_conv_out");
}
