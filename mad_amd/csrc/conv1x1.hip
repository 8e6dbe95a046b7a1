// 1x1 convolution (bf16) as a plain GEMM on the 16x16x32 MFMA: the attention blocks' qkv and output projections
// and the residual branch's channel change where it is not folded into a 3x3 launch.
//
//   out[m][n] = bias[n] + sum_k x[m][k] * w[n][k]  (+ residual[m][n]),   m = pixel (NHWC row), k over [x1 | x2]
//
//   Block  = 256 consecutive pixels x BN = 128 output channels, four waves 2 x 2; a wave owns 128 pixels
//            (eight 16-pixel row tiles) x BN/2 channels.  The weights are the FIRST operand of every product, so a
//            lane's four results are four consecutive channels of one pixel.
//   K loop = 32-channel chunks (64-byte rows of both operands) through a ring of three LDS stages filled by LDS-DMA
//            two chunks ahead and retired with counted vmcnt waits; per chunk a wave reads its 8 pixel fragments once
//            and walks the channel fragments with two of them live, products accumulating in place.
//   LDS    = row r of either operand at byte r*64, 16-byte chunk index XOR (-(r>>2) & 3): every ds_read_b128 of a
//            fragment (16 rows x one chunk per lane group) is bank-conflict free.
//   Epilogue = straight from the accumulators: lane pairs exchange four-channel quads (v_permlane16_swap) so that
//            each lane holds eight consecutive channels of its pixel and writes them with one 16-byte store; no LDS
//            hand-over and no barrier.  Statistics (optional): sum / sum of squares of the stored values per 64-pixel
//            run and channel PAIR (both sums in the slot of the even channel, zeros in the odd one's), reduced over
//            the 16 pixel columns by DPP: each wave writes the rows of its own two runs.
#include "mma.h"
#include <cstdlib>

namespace mad {

struct C1Args {
    const void* x1; const void* x2; const void* w; const float* bias; const void* residual; void* y; float* stats;
    int C1, C2, M, Cout, mtiles, ntiles;
};

namespace c1 {
constexpr int BM = 256, ROWB = 64, CK = 32, NST = 3;
template <int WN> constexpr int stage_bytes() { return (BM + 64 * WN) * ROWB; }
template <int WN> constexpr int lds_bytes() { return NST * stage_bytes<WN>(); }
}  // namespace c1

// WN: waves along the channels (2 x WN waves, BN = 64 * WN).  2 = four waves, two workgroups per CU, is the form in
// use; 4 (eight waves, one workgroup per CU, a third fewer copies per product) measured 0-10 % slower.
template <int WN>
__global__ __launch_bounds__(128 * WN, WN == 2 ? 2 : 1) void conv1x1_kernel(C1Args a) {
    using namespace c1;
    constexpr int TN = 2, NWAVES = 2 * WN;
    constexpr int BN = 64 * WN, TM16 = 8, TN16 = 2 * TN, NP = TN;
    constexpr int APW = BM / 16 / NWAVES, BPW = BN / 16 / NWAVES, PPW = APW + BPW;      // copy instructions per wave and chunk
    constexpr int STAGE = stage_bytes<WN>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fh = lane >> 4;

    // XCD-aware order: consecutive logical tiles (the channel tiles of one pixel tile first) share an L2
    const int nwg = a.mtiles * a.ntiles;
    const int id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int mt = logical / a.ntiles, nt = logical % a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int Ctot = a.C1 + a.C2;
    const int nk = Ctot / CK;

    // the accumulators start from the bias of their channels
    f32x4 acc[TM16][TN16];
    {
        f32x4 b4[TN16];
#pragma unroll
        for (int j = 0; j < TN16; ++j) {
            b4[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (a.bias) b4[j] = *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * (TN * 32) + j * 16 + 4 * fh);
        }
#pragma unroll
        for (int j = 0; j < TN16; ++j) asm volatile("" : "+v"(b4[j]));      // loaded before the first copy is issued
#pragma unroll
        for (int i = 0; i < TM16; ++i)
#pragma unroll
            for (int j = 0; j < TN16; ++j) acc[i][j] = b4[j];
    }

    // ---- copies.  Instruction j of a wave covers the 16 rows 16 * (j * NWAVES + wave) .. +15 of an operand tile,
    // four lanes per row; (-(r>>2)) & 3 depends on the row inside the block only, so a copy is a scalar base (tile,
    // chunk, block) plus ONE per-lane byte offset per source.
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);
    const unsigned aoff1 = (unsigned)((lrow * a.C1 + lchunk * 8) * 2);
    const unsigned aoff2 = (unsigned)((lrow * a.C2 + lchunk * 8) * 2);
    const unsigned woff = (unsigned)((lrow * Ctot + lchunk * 8) * 2);
    const unsigned zoff = (unsigned)lane * 16u;
    const unsigned smem_base = lds_addr(smem);
    auto stage = [&](int kt, int buf) {
        const int c0 = kt * CK;
        const bf16_t* src; int cs, coff; unsigned ao;
        if (c0 < a.C1) { src = (const bf16_t*)a.x1; cs = a.C1; coff = c0; ao = aoff1; }
        else { src = (const bf16_t*)a.x2; cs = a.C2; coff = c0 - a.C1; ao = aoff2; }
        const unsigned abuf = smem_base + buf * STAGE, bbuf = abuf + BM * ROWB;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            // 16-pixel blocks past the end of the tensor are filled from the zero page (same number of copies)
            const int r0 = 16 * (j * NWAVES + wave);
            const bool in = m0 + r0 < a.M;
            glds16s(in ? (const void*)(src + coff + (size_t)(m0 + r0) * cs) : (const void*)g_zero_page, in ? ao : zoff,
                    abuf + r0 * ROWB);
        }
        const bf16_t* wb = (const bf16_t*)a.w + (size_t)n0 * Ctot + c0;
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int r0 = 16 * (j * NWAVES + wave);
            glds16s(wb + (size_t)r0 * Ctot, woff, bbuf + r0 * ROWB);
        }
    };

    // fragment addresses: row r, 16-byte chunk fh of its 64 bytes; rows 16 apart share their swizzle
    int a_off, b_off;
    {
        const int ra = wm * 128 + fr, rb = wn * (TN * 32) + fr;
        a_off = ra * ROWB + ((fh ^ ((-(ra >> 2)) & 3)) << 4);
        b_off = BM * ROWB + rb * ROWB + ((fh ^ ((-(rb >> 2)) & 3)) << 4);
    }

    stage(0, 0);
    if (nk > 1) stage(1, 1);
    int buf = 0, nbuf = 2;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) dma_wait_upto<PPW>(); else dma_wait_all();
        __syncthreads();              // chunk kt is visible; every wave is done with the stage refilled below
        if (kt + 2 < nk) stage(kt + 2, nbuf);
        const unsigned char* st = smem + buf * STAGE;
        uint4 af[TM16], bfs[2];
        bfs[0] = *reinterpret_cast<const uint4*>(st + b_off);
#pragma unroll
        for (int i = 0; i < TM16; ++i) af[i] = *reinterpret_cast<const uint4*>(st + a_off + i * 16 * ROWB);
#pragma unroll
        for (int j = 0; j < TN16; ++j) {
            if (j + 1 < TN16) bfs[(j + 1) & 1] = *reinterpret_cast<const uint4*>(st + b_off + (j + 1) * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < TM16; ++i) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                             : "+v"(acc[i][j])
                             : "v"(__builtin_bit_cast(bf16x8, bfs[j & 1])), "v"(__builtin_bit_cast(bf16x8, af[i])));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    // the products are asm statements: the wait states a read of a matrix-pipe result needs are provided here
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

    // ---- epilogue (see the head of the file)
    const bf16_t* res = (const bf16_t*)a.residual;
    bf16_t* y = (bf16_t*)a.y;
    const int cb0 = wn * (TN * 32) + (fh & 1) * 16 + (fh >> 1) * 8;      // this lane's first channel, pair 0
    const int mw = m0 + wm * 128;                                          // first pixel of this wave
    auto row16_sum = [](float x) {
        // x may come straight out of an asm statement (the dot products above), where the compiler does not see a
        // vector write: the two wait states a DPP read of a freshly written register needs are provided here
        asm volatile("s_nop 1" : "+v"(x));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
        return x;
    };
    constexpr int RG = 4;                       // rows per group: one 64-pixel run
#pragma unroll
    for (int i0 = 0; i0 < TM16; i0 += RG) {
        if (mw + i0 * 16 >= a.M) break;         // runs past the end of the tensor (wave-uniform)
        uint4 rpre[RG][NP];
        if (res) {
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    rpre[r][p] = *reinterpret_cast<const uint4*>(res + (size_t)(mw + (i0 + r) * 16 + fr) * a.Cout + n0 + cb0 + p * 32);
        }
        float ssum[NP][4], ssq[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) { ssum[p][k] = 0.0f; ssq[p][k] = 0.0f; }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int i = i0 + r;
            const size_t m = (size_t)(mw + i * 16 + fr);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[i][2 * p][e]),
                                                                     __float_as_uint(acc[i][2 * p + 1][e]), false, false);
                    v[e] = __uint_as_float(sw[0]); v[4 + e] = __uint_as_float(sw[1]);
                }
                if (res) {
                    Vec16<bf16_t> rv; rv.raw = rpre[r][p];
                    float f[8]; rv.unpack(f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += f[e];
                }
                Vec16<bf16_t> o; o.pack(v);
                o.store(y + m * a.Cout + n0 + cb0 + p * 32);
                if (a.stats) {
                    const unsigned* w4 = reinterpret_cast<const unsigned*>(&o.raw);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ssum[p][k]) : "v"(w4[k]), "v"(0x3f803f80u));
                        asm("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(ssq[p][k]) : "v"(w4[k]));
                    }
                }
            }
        }
        if (a.stats) {
            const size_t run = (size_t)(mw + i0 * 16) / 64;
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sm = row16_sum(ssum[p][k]), sq = row16_sum(ssq[p][k]);
                    if (fr == 0)
                        *reinterpret_cast<float4*>(a.stats + (run * a.Cout + n0 + cb0 + p * 32 + 2 * k) * 2) =
                            make_float4(sm, sq, 0.0f, 0.0f);
                }
        }
    }
}

// Shapes the kernel covers, and where it pays: whole 64-pixel runs, whole 32-channel chunks per source, and enough
// tiles to give every CU a workgroup.  Measured against the implicit GEMM (batch 32 / 64, TFLOP/s): 32x32 256->768
// 410 -> 531, 32x32 384->1152 539 -> 686, 32x32 384->384 with residual and statistics 403 -> 488, 8x8 768->2304
// 488 -> 544; with fewer tiles than CUs, or at 64 channels per workgroup (odd multiples of 64), no gain: those
// layers stay on the implicit GEMM.  MAD_CONV1X1 (tuning / test aid): 0 = never, 2 = wherever the shape fits.
static int c1_plan(int M, int Cout) {          // output channels per workgroup; 0: not this kernel
    if (Cout % 128) return 0;
    const char* f = getenv("MAD_CONV1X1");
    if (f && atoi(f) == 0) return 0;
    if (f && atoi(f) == 2) return 128;
    return (long long)((M + 255) / 256) * (Cout / 128) >= 256 ? 128 : 0;
}
bool conv1x1_ok(int M, int C1, int C2, int Cout) {
    if (M % 64 || C1 <= 0 || C1 % 32 || C2 < 0 || C2 % 32) return false;
    return c1_plan(M, Cout) != 0;
}

template <int WN> static int launch_c1(C1Args a, hipStream_t s) {
    constexpr int lds = c1::lds_bytes<WN>();
    static int lds_dev = -1;
    MAD_REQUIRE(raise_lds_limit(reinterpret_cast<const void*>(&conv1x1_kernel<WN>), lds, &lds_dev),
                "mad_conv2d: cannot raise the dynamic LDS limit to %d", lds);
    a.mtiles = (a.M + c1::BM - 1) / c1::BM;
    a.ntiles = a.Cout / (64 * WN);
    hipLaunchKernelGGL((conv1x1_kernel<WN>), dim3(a.mtiles * a.ntiles), dim3(128 * WN), lds, s, a);
    return check_launch("mad_conv2d");
}

int conv1x1(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias, const void* residual,
            void* y, float* stats, int M, int Cout, hipStream_t s) {
    C1Args a;
    a.x1 = x1; a.x2 = C2 ? x2 : x1; a.w = w; a.bias = bias; a.residual = residual; a.y = y; a.stats = stats;
    a.C1 = C1; a.C2 = C2; a.M = M; a.Cout = Cout; a.mtiles = a.ntiles = 0;
    return launch_c1<2>(a, s);
}

}  // namespace mad
