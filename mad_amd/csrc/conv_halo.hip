// 3x3 convolution for the high-resolution layers: halo-tile implicit GEMM on MFMA with the
// GroupNorm apply + SiLU of the *input* fused into the operand loader and the GroupNorm
// statistics of the *output* produced by the epilogue.
//
//   Block  = one 16x16 output tile of one image x BN output channels, 4 waves laid out WM x WN; a
//            wave owns 16/WM tile rows x TN*32 channels.  BN = 128 (WM=2, WN=2, TN=2: 8 accumulator
//            tiles of 32x32 per wave) for the body of the network; BN = 32 (WM=4, WN=1, TN=1) for
//            the last layer, whose 3 output channels are zero-padded to 32 and written as NCHW fp32.
//   K loop = for each 64-byte channel chunk: the 18x18 input halo of the chunk is staged ONCE into
//            LDS, then the 9 taps are 9 K-tiles that read it at shifted addresses.  Everything that
//            comes from global memory inside the loop -- halo items, weight tiles (BN rows x 64 bytes
//            per tap, three buffers), the chunk's GroupNorm coefficients -- arrives by LDS-DMA issued
//            two K-tiles (weights) to three K-tiles (halo) ahead and is retired with COUNTED vmcnt
//            waits, so the loop never drains the memory pipeline.  The halo of chunk c+1 is filled
//            during chunk c, one 16-byte item per thread per tap; each thread then rewrites its own
//            item in place as y = silu(x*a+d) (zeros outside the image) three taps later.
//   LDS    = halo pixels at 64-byte pitch, 18 pixels per row, chunk index XOR ((x>>1)&3);
//            weight rows at 64-byte pitch, chunk index XOR (-(row>>2)&3): both make every
//            ds_read_b128 fragment read bank-conflict free for all nine tap shifts, with 32 rows x 2 chunks per
//            wave (32x32x16 MFMA) and with 16 rows x 4 chunks (16x16x32 MFMA).
//   Epilogue = per half tile: fp32 accumulators -> LDS -> coalesced 16-byte stores with bias and
//            residual; sum / sum of squares of the stored values go to a (B, tiles, Cout, 2) slab in a
//            fixed order (deterministic).  fp32: per channel.  bf16: per channel PAIR (2i, 2i+1), both sums in
//            slot 2i and zeros in slot 2i+1 -- a GroupNorm group holds whole pairs and only group sums are used.
//   bf16 on the 16x16x32 MFMA (the forms the denoiser runs in; the description above is the 32x32x16 / fp32 form):
//            the weights are the first operand, so a lane's four results are four consecutive channels of a pixel.
//            The window is walked column by column and a wave's eight pixel-row fragments stay in registers across
//            the three vertical taps (one new row per tap: 30 + 36 instead of 72 + 36 fragment reads per chunk);
//            the products are asm statements that accumulate in place; the output goes straight from the
//            accumulators to memory (lane pairs exchange quads with v_permlane16_swap: eight consecutive channels,
//            one 16-byte store per lane), statistics by v_dot2c_f32_bf16 per lane, DPP across the 16 pixel
//            columns and 1 KiB of LDS across waves -- no tile hand-over, no barrier between products and stores.
//            KS = 2: layers whose tiles put at most one workgroup on a CU run eight waves, two sets of four that
//            multiply alternate channel chunks out of their own staging buffers and then exchange half of their
//            accumulators, each set finishing half of the rows.
#include "mma.h"
#include <cstdlib>
#include <type_traits>

namespace mad {

struct HaloArgs {
    const void* x1; const void* x2; const void* w; const float* bias; const float* coef;
    const void* residual; void* y; float* stats;
    int C1, C2, B, H, W, Cout, upsample, res_mode, act;
    int Hs, Ws, tiles_x, tiles_y, ntiles_n;
    int n_off;                 // first output channel of this launch (a layer may be split over two launches)
    // optional 1x1 projection of a second input added to the result (extra K-tiles, centre tap only)
    const void* sx1; const void* sx2; const void* sw;
    int SC1, SC2;
    int stream_out;   // output larger than the on-die caches: written with streaming stores
    int wtiled;       // w is in the tiled layout (see mad_conv3x3_fused): every weight copy reads 1 KiB in one piece
    // GSTAT form: the output is the gradient dy reaching h = act(GroupNorm([gx1, gx2]) (affine in gcoef)), possibly
    // dropped out (gdm); `stats` then receives that GroupNorm backward's partial sums (sum g, sum g*x) per tile,
    // g = dy * keep * act'(a*x + d), instead of the output's own statistics.
    const void* gx1; const void* gx2; const float* gcoef;
    int gC1, gact;
    DropMask gdm;
};

constexpr int HT = 16;                 // output tile height (and width of the narrow tile)
constexpr int HS = 18;                 // halo rows
constexpr int HROW = 64;               // bytes per halo pixel / weight row per K-tile
// Output tile = 16 rows x TW columns, TW = 16 (four waves) or 32 (eight waves: two 16x16 halves side by side that
// share one stream of weight tiles -- half the weight copies per multiply -- and a 18 x 34 halo).
template <int TW> constexpr int halo_items() { return HS * (TW + 2) * 4; }      // 16-byte items per halo chunk
template <int TW> constexpr int halo_buf() { return (halo_items<TW>() + 63) / 64 * 64 * 16; }   // whole wave-instructions
constexpr int CTAB_BYTES = 1024;       // one DMA instruction's worth; 256 (bf16) / 128 (fp32) bytes used
// MFMA shape.  0: v_mfma_f32_32x32x16_bf16 everywhere; 1: v_mfma_f32_16x16x32_bf16 in the instantiations without the
// loader transform; 2: also in the bf16 four-wave, 128-channel instantiations with it (the ones that fit 256
// registers in that form); 3: everywhere.  (On this chip the 16x16x32 form sustains a higher clock for the same
// multiplies; it issues twice as many matrix instructions, which the loader transform's vector work competes with.)
#ifndef MAD_HALO_M16
#define MAD_HALO_M16 2
#endif
// 1: the loader transform of a tap runs as one block in front of the tap's multiplies instead of in 16 pieces
// between them (fewer registers live across the multiplies)
#ifndef MAD_XF_PHASED
#define MAD_XF_PHASED 0
#endif
// where in a tap (after which sixteenth of its MFMAs) the halo item and the weight tile of later taps are requested
#ifndef MAD_DMA_HALO_SIDX
#define MAD_DMA_HALO_SIDX 1
#endif
#ifndef MAD_DMA_W_SIDX
#define MAD_DMA_W_SIDX 3
#endif
// 1: the bf16 epilogue takes the output statistics over channel pairs with v_dot2c_f32_bf16 (see the epilogue)
#ifndef MAD_STATS_DOT2
#define MAD_STATS_DOT2 1
#endif
#ifndef MAD_HALO_PF
#define MAD_HALO_PF 2
#endif
// 1: the 16x16x32 forms keep a wave's pixel-row fragments in registers across the three vertical taps of a window
// column (30 instead of 72 pixel-fragment reads from LDS per channel chunk)
#ifndef MAD_HALO_SLIDE
#define MAD_HALO_SLIDE 1
#endif
// 1: the bf16 16x16x32 forms write their output straight from the accumulators (lane pairs exchange four-channel quads
// with v_permlane16_swap so that every lane holds eight consecutive channels of its pixel): no LDS hand-over, no
// barriers between the products and the stores, statistics reduced across pixel columns by DPP
#ifndef MAD_HALO_DSE
#define MAD_HALO_DSE 1
#endif
// 1: the products of the sliding-window form are asm statements that accumulate in place.  Left to the compiler, every
// product may write its result to other registers than it read the accumulator from, and the register assignment of
// 32 renamed accumulator quads beside the window's fragments no longer fits 256 registers (23-59 spilled, reloaded
// inside the loop); pinned in place the same loop needs 241.
#ifndef MAD_HALO_ASM_MMA
#define MAD_HALO_ASM_MMA 1
#endif
constexpr int PF = MAD_HALO_PF;        // weight tiles are copied PF K-tiles ahead of their use
constexpr int NWBUF = PF + 1;

#ifdef MAD_HALO_STAMPS
// Timing aid (separate build only): cycle stamps of every wave of one workgroup for the taps of its
// second channel chunk, [wave][tap][point]; points: tap top, copies landed, barrier passed,
// MFMAs issued, transform done.
__device__ unsigned g_stamps[8 * 9 * 8];
// Clock aid of the same build: every workgroup leaves (shader cycles, 100 MHz reference ticks) between its first and
// its last instruction; cycles / ticks * 100 MHz is the clock the chip held while the workgroup ran.  The values go to
// this array only, which nothing else in the kernel reads.
constexpr int CLOCK_SLOTS = 8192;
__device__ unsigned long long g_clock[CLOCK_SLOTS * 2];
#define MAD_STAMP(pt)                                                                      \
    if (stamp_on && cc == 1) {                                                             \
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();                        \
        if (lane == 0) stamp_lds[(wave * 9 + tap) * 8 + (pt)] = t_;                        \
    }
#define MAD_STAMP_PHASE(k)                                                                 \
    if (stamp_on) {                                                                        \
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();                        \
        if (lane == 0) stamp_lds[(wave * 9 + (k) % 9) * 8 + 5 + (k) / 9] = t_;             \
    }
#else
#define MAD_STAMP(pt)
#define MAD_STAMP_PHASE(k)
#endif

// fma1 is an asm statement: use it only on operands that come from ordinary vector instructions, LDS reads or
// scalars (never straight from v_exp / v_rcp or an MFMA: see div2)
__device__ __forceinline__ float fma1(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// t0 /= u0, t1 /= u1 as reciprocal + product, four one-element instructions in one statement.  On this chip a
// result of the transcendental unit needs one wait state before an ordinary vector instruction reads it; the
// compiler inserts it for its own instructions but not inside or in front of an asm statement, so the order
// below provides it: each product is separated from its reciprocal by one other instruction.
__device__ __forceinline__ void div2(float& t0, float& t1, float u0, float u1) {
    float r0, r1;
    asm("v_rcp_f32 %0, %4\n\t"
        "v_rcp_f32 %1, %5\n\t"
        "v_mul_f32 %2, %2, %0\n\t"
        "v_mul_f32 %3, %3, %1"
        : "=&v"(r0), "=&v"(r1), "+v"(t0), "+v"(t1)
        : "v"(u0), "v"(u1));
}

// staging buffers of one set of waves: two halo buffers, the weight-tile ring, two coefficient tables, one sink KiB
template <int BN, int TW> constexpr int halo_stage_bytes() { return 2 * halo_buf<TW>() + NWBUF * BN * HROW + 2 * CTAB_BYTES + 1024; }
// KS = 2: two sets of waves, each with its own staging buffers (see the kernel); their accumulator hand-over
// (half of each wave's quads, 16 bytes per lane) fits the same space
template <int BN, int TW, int KS = 1> constexpr int halo_lds_bytes() {
    constexpr int stage = KS * halo_stage_bytes<BN, TW>();
    constexpr int epi = KS == 1 ? 128 * (BN + 4) * 4 : 1024 * BN;
    static_assert(KS == 1 || epi <= stage, "the hand-over of the two-share form fits its staging space");
#ifdef MAD_HALO_STAMPS
    return (stage > epi ? stage : epi) + 4096;
#else
    return stage > epi ? stage : epi;
#endif
}

// OUT_NCHW: write the first a.Cout (<= BN) channels as NCHW fp32 with bias (last layer).
// PRE: the coefficient table holds (a, d) * -log2(e) and the weights carry a factor -ln(2): the loader computes
// u = x*a' + d' and u / (1 + 2^u), which is -log2(e) * silu(x*a + d), one multiply-add per element less.
// KS = 2 (two K shares in one workgroup): for layers whose tiles put no more than one workgroup on a CU.  The
// workgroup is two sets of WM x WN waves; both own the same pixels and channels, set g multiplies the channel chunks
// g, g + 2, ... out of staging buffers of its own (the loop below, unchanged, at twice the waves per SIMD), then the
// sets exchange half of their accumulators through LDS and each finishes half of the rows.
template <typename T, bool FUSE, int WM, int WN, int TN, bool OUT_NCHW, bool SKIP, bool GSTAT, int TW = 16, bool PRE = false,
          int KS = 1>
__global__ __launch_bounds__(64 * WM * WN * KS, KS == 2 ? 1 : 2) void conv3x3_halo_kernel(HaloArgs a) {
    static_assert(!PRE || (FUSE && !OUT_NCHW && !GSTAT), "the prescaled form belongs to the loader transform");
    static_assert(KS == 1 || KS == 2, "one or two K shares");
    // NWAVES, NT, `wave`, `tid` below: of ONE share (all the staging and fragment code is per share)
    constexpr int NWAVES = WM * WN, NT = 64 * NWAVES;
    static_assert(NWAVES == 4 || NWAVES == 8, "4 or 8 waves");
    static_assert(KS == 1 || (NWAVES == 4 && TW == 16), "two shares of four waves");
    static_assert(TW == 16 || TW == 32, "tile width");
    constexpr int HSX = TW + 2;                    // halo columns
    constexpr int HALO_ITEMS = halo_items<TW>(), HALO_BUF = halo_buf<TW>();
    constexpr int HALO_ROUNDS = (HALO_ITEMS + NT - 1) / NT;   // items per thread: 6 (16 wide, 4 waves) / 5 (32 wide, 8 waves)
    static_assert(HALO_ROUNDS <= 6, "one halo item per tap, transformed three taps later");
    constexpr int BN = WN * TN * 32;
    constexpr int WMH = WM / (TW / 16);            // waves stacked along the rows of one 16-column half
    static_assert(WMH >= 2 && WM % (TW / 16) == 0, "a wave covers at most 8 rows of one half");
    constexpr int TM = 8 / WMH;                    // 32-pixel accumulator tiles per wave (2 tile rows each)
    constexpr bool M16 = MAD_HALO_M16 >= 3 || (MAD_HALO_M16 >= 1 && !FUSE) ||
                         (MAD_HALO_M16 == 2 && TW == 16 && TN == 2 && sizeof(T) == 2);
    constexpr int TM16 = 2 * TM, TN16 = 2 * TN;   // M16: 16-pixel (one tile row) and 16-channel fragments per wave
    // pixel-row fragments kept in registers across the vertical taps (see the chunk body); the fp32 parity form has
    // no registers for it
    constexpr bool SLIDE = M16 && sizeof(T) == 2 && MAD_HALO_SLIDE;
    constexpr int NSLOT = TM16;
    constexpr bool DSE = M16 && sizeof(T) == 2 && !OUT_NCHW && !GSTAT && MAD_HALO_DSE;
    static_assert(KS == 1 || DSE, "the two-share form finishes straight from the accumulators");
    constexpr int WTILE_BYTES = BN * HROW;
    constexpr int EPI_LD = BN + 4;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CK = HROW / (int)sizeof(T);      // channels per chunk: 32 (bf16) / 16 (fp32)
    constexpr int NW = (BN / 16 + NWAVES - 1) / NWAVES;   // weight DMA instructions per wave per K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the wave index is read into a scalar register: LDS destinations of the copies and everything else derived
    // from it then live in scalar registers and cost no vector instructions
    const int tid_all = threadIdx.x;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int grp = KS == 2 ? wave_all / NWAVES : 0;          // K share of this wave
    const int tid = KS == 2 ? tid_all & (NT - 1) : tid_all, lane = tid & 63, wave = KS == 2 ? wave_all % NWAVES : wave_all;
    unsigned char* const halo0 = smem + grp * halo_stage_bytes<BN, TW>();
    unsigned char* const wt0 = halo0 + 2 * HALO_BUF;
    unsigned char* const ctab0 = wt0 + NWBUF * WTILE_BYTES;
    unsigned char* const sink0 = ctab0 + 2 * CTAB_BYTES;     // target of copies that only keep op counts equal

    const int wm = wave / WN, wn = wave % WN;

    // ---- tile coordinates (XCD-aware order: consecutive logical tiles share an L2)
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int nwg = a.B * tiles_img * a.ntiles_n;
    const int id = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
    int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int nt = logical % a.ntiles_n; logical /= a.ntiles_n;
    const int tile = logical % tiles_img;
    const int b = logical / tiles_img;
    const int ty0 = (tile / a.tiles_x) * HT, tx0 = (tile % a.tiles_x) * TW;
    const int n0 = a.n_off + nt * BN;

    const int Ctot = a.C1 + a.C2;
    const int nchunks = Ctot / CK;                 // of the layer; a share walks nchunks / KS of them:
    auto gchunk = [&](int cc) { return cc * KS + grp; };      // its cc-th chunk is the layer's chunk gchunk(cc)
    // The accumulators start from the bias of their channels (the epilogue then has nothing to add); the NCHW form
    // adds it when it writes.  Register e of a lane: M16 channel 16*j + 4*fh + e; 32x32: 32*j + 8*(e/4) + 4*fh + e%4.
    // Requested here, ahead of the tile geometry, and pinned before the first copy is issued: an ordinary load
    // waited for after the copies would make the compiler's wait cover the copies as well.
    constexpr int NB4 = M16 ? 2 * TN : 4 * TN;
    f32x4 bias4[NB4];
    {
        const bool bias_init = !OUT_NCHW && a.bias != nullptr && grp == 0;
        const int fh_ = M16 ? lane >> 4 : lane >> 5;
#pragma unroll
        for (int k = 0; k < NB4; ++k) {
            bias4[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (bias_init) bias4[k] = *reinterpret_cast<const f32x4*>(a.bias + n0 + wn * (TN * 32) + k * (M16 ? 16 : 8) + 4 * fh_);
        }
    }
#ifdef MAD_HALO_STAMPS
    const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned* const stamp_lds = reinterpret_cast<unsigned*>(smem + halo_lds_bytes<BN, TW, KS>() - 4096);
    const bool stamp_on = blockIdx.x == gridDim.x / 2 + 3;
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    if (stamp_on) {
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();
        if (lane_ == 0) stamp_lds[(wave_ * 9 + 0) * 8 + 5] = t_;
    }
#endif

    // ---- halo staging assignment.  Item = tid + NT*j sits at LDS offset item*16 (lane-linear, as LDS-DMA
    // requires): pixel p = item>>2, physical chunk tid&3, which holds logical chunk (tid&3) ^ ((hx>>1)&3).
    // In the last round only the first wave(s) have real items; the others copy the zero page into a sink
    // area that nobody reads, so that every wave issues the same number of memory operations (the counted
    // waits below rely on it) and no copy can land on an item that was already transformed in place.
    const bool last_round_real = NT * (HALO_ROUNDS - 1) + 64 * wave < HALO_BUF / 16;
    // source pixel index (b*Hs + sy)*Ws + sx of each item.  Padding / no item: -1 in the plain form (copied from the
    // zero page); in the FUSE form 0 with the item's bit set in padmask (the transform writes zeros there)
    // (FUSE: kept as 16-bit offsets from the tile's first in-image source pixel, two per register; the tile spans at
    // most 18 source rows, checked on the host)
    int src_pix[FUSE ? 1 : HALO_ROUNDS];
    unsigned rel16[FUSE ? (HALO_ROUNDS + 1) / 2 : 1] = {};
    const int oy0 = max(ty0 - 1, 0), ox0 = max(tx0 - 1, 0);
    const int sy0 = a.upsample ? oy0 >> 1 : oy0, sx0 = a.upsample ? ox0 >> 1 : ox0;
    const int base_pix = (b * a.Hs + sy0) * a.Ws + sx0;            // wave-uniform
    unsigned padmask = 0;
    unsigned lchunks = 0;            // 2 bits per round: logical chunk of this thread's item
#pragma unroll
    for (int j = 0; j < HALO_ROUNDS; ++j) {
        const int item = tid + NT * j;
        const int p = item >> 2;
        const int hy = p / HSX, hx = p - hy * HSX;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        // (unsigned compares and bitwise ands: one mask per item, no branches)
        const bool ok = (item < HALO_ITEMS) & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        const int sy = a.upsample ? iy >> 1 : iy, sx = a.upsample ? ix >> 1 : ix;
        if constexpr (FUSE) rel16[j / 2] |= (unsigned)(ok ? (sy - sy0) * a.Ws + (sx - sx0) : 0) << (16 * (j & 1));
        else src_pix[j] = ok ? (b * a.Hs + sy) * a.Ws + sx : -1;
        if (!ok) padmask |= 1u << j;
        lchunks |= (unsigned)((tid & 3) ^ ((hx >> 1) & 3)) << (2 * j);
    }
    // The last round of the 16-wide tile holds only LAST_REAL (16) items.  Its transform is not run as one more
    // whole-item round by every wave: the LAST_REAL * XPAIRS element pairs are spread over the lanes of wave 0, one
    // pair per lane (a quarter of an item's arithmetic, once).
    constexpr int XPAIRS = VEC / 2;
    constexpr int LAST_REAL = HALO_ITEMS - NT * (HALO_ROUNDS - 1);
    constexpr bool MINI = FUSE && HALO_ROUNDS > 1 && LAST_REAL * XPAIRS <= 64;
    // (its per-lane addresses and flags are worked out where they are used, once per chunk, rather than kept in
    // registers through the loop)
    auto mini_geom = [&](int& off, int& cf, bool& on, bool& pad) {
        const int it = NT * (HALO_ROUNDS - 1) + lane / XPAIRS, pr = lane % XPAIRS;
        const int p = it >> 2;
        const int hy = p / HSX, hx = p - hy * HSX;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        on = wave == 0 && lane < LAST_REAL * XPAIRS;
        pad = !(iy >= 0 && iy < a.H && ix >= 0 && ix < a.W);
        off = it * 16 + pr * (16 / XPAIRS);
        cf = (((it & 3) ^ ((hx >> 1) & 3)) * VEC + 2 * pr) * 2;
    };
    const unsigned halo_base = lds_addr(halo0);
    const unsigned sink_base = lds_addr(sink0);
    auto dma_item = [&](int cc, int j, int buf) {
        const int c0 = gchunk(cc) * CK;
        const T* src; int cs, coff;
        if (c0 < a.C1) { src = (const T*)a.x1; cs = a.C1; coff = c0; }
        else { src = (const T*)a.x2; cs = a.C2; coff = c0 - a.C1; }
        unsigned lcs = lchunks;
        asm volatile("" : "+v"(lcs));
        const int lc = (lcs >> (2 * j)) & 3;
        const bool sink = j == HALO_ROUNDS - 1 && !last_round_real;     // wave-uniform
        const unsigned dst = sink ? sink_base : halo_base + buf * HALO_BUF + (NT * j + 64 * wave) * 16;
        if constexpr (FUSE) {
            // scalar base = the source's channel chunk, 32-bit lane offset = pixel and logical chunk (pixel index
            // and row pitch both below 2^24; the tensor below 4 GB: checked on the host)
            unsigned packed = rel16[j / 2];
            asm volatile("" : "+v"(packed));      // unpacked where it is used, not once per round ahead of the loop
            const unsigned rel = (packed >> (16 * (j & 1))) & 0xffffu;
            const unsigned off = (__umul24(rel, (unsigned)cs) + lc * VEC) * (unsigned)sizeof(T);
            glds16s(src + coff + (size_t)base_pix * cs, off, __builtin_amdgcn_readfirstlane(dst));
        } else {
            const T* p = src + (size_t)max(src_pix[j], 0) * cs + coff + lc * VEC;
            const void* g = src_pix[j] >= 0 ? (const void*)p : (const void*)g_zero_page;
            glds16(sink ? (const void*)g_zero_page : g, __builtin_amdgcn_readfirstlane(dst));
        }
    };
    // GroupNorm coefficient table of one chunk: CK channels x (a, d) fp32, same copy from every wave
    const unsigned ctab_base = lds_addr(ctab0);
    auto dma_coef = [&](int cc, int buf) {
        constexpr int LANES = CK * 8 / 16;          // 16 (bf16) / 8 (fp32) lanes carry the table
        glds16s(a.coef + ((size_t)b * Ctot + gchunk(cc) * CK) * 2, (unsigned)(lane & (LANES - 1)) * 16u,
                __builtin_amdgcn_readfirstlane(ctab_base + buf * CTAB_BYTES));
    };
    // exponent of the sigmoid: -t*log2(e) with the activation, -200 without it (exp2 gives 0, the reciprocal
    // 1, the product t itself): no select on the way
    const float xk1 = a.act ? -1.44269504088896340736f : 0.0f, xk0 = a.act ? 0.0f : -200.0f;
    // denominator 1 + 2^u of the sigmoid from t = x*a + d (PRE: t already is the exponent)
    auto sig_den = [&](float t) { return 1.0f + __builtin_amdgcn_exp2f(PRE ? t : fma1(t, xk1, xk0)); };
    // in-place y = act(x*a + d) of this lane's element pair of the last round (MINI), zeros for padding
    auto xform_mini = [&](int buf) {
        int mini_off, mini_cf; bool mini_on, mini_pad;
        mini_geom(mini_off, mini_cf, mini_on, mini_pad);
        unsigned char* addr = halo0 + buf * HALO_BUF + mini_off;
        const float4 ad = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ctab0 + buf * CTAB_BYTES) + mini_cf);
        float f0, f1;
        if (sizeof(T) == 2) {
            const unsigned w = *reinterpret_cast<const unsigned*>(addr);
            f0 = __uint_as_float(w << 16); f1 = __uint_as_float(w & 0xffff0000u);
        } else {
            const float2 w = *reinterpret_cast<const float2*>(addr);
            f0 = w.x; f1 = w.y;
        }
        float t0 = fma1(f0, ad.x, ad.y), t1 = fma1(f1, ad.z, ad.w);
        const float u0 = sig_den(t0), u1 = sig_den(t1);
        div2(t0, t1, u0, u1);
        if (mini_pad) { t0 = 0.0f; t1 = 0.0f; }
        if (mini_on) {
            if (sizeof(T) == 2) {
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t v2 = {t0, t1};
                *reinterpret_cast<unsigned*>(addr) = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2_t));
            } else {
                *reinterpret_cast<float2*>(addr) = make_float2(t0, t1);
            }
        }
    };

    // The same transform cut into 16 short pieces, one behind each MFMA of a tap (in-loop form): the matrix
    // pipe runs the MFMA while the vector pipe runs the piece, instead of the whole transform standing behind
    // the tap's last MFMA.  Element pair p (one dword of bf16, two floats of fp32) goes through four stages in
    // the slots XSTRIDE*p .. XSTRIDE*p+3: unpack + multiply-add, exp, reciprocal + product, pack.  The item and
    // the first coefficient pair are requested at the top of the tap; each later coefficient pair one pair ahead.
    constexpr int XSTRIDE = 16 / XPAIRS;
    uint4 xraw; float4 xcf[1]; float xt[2], xu[2]; unsigned xout[4];
    // the "1" of the sigmoid's denominator 1 + 2^u: +infinity for an item that is padding, whose reciprocal is 0
    // and whose product with the (finite) t is the zero the convolution pads with -- no selects on the way out
    float xone = 1.0f;
    const unsigned char* xaddr = nullptr; const float* xcoef = nullptr;
    auto xf_begin = [&](int j, int buf) {
        xone = __uint_as_float(0x3f800000u + (((padmask >> j) & 1u) << 30));     // 1.0f, or 0x7f800000 = +inf
        const int item = tid + NT * j;
        xaddr = halo0 + buf * HALO_BUF + item * 16;
        xraw = *reinterpret_cast<const uint4*>(xaddr);
        unsigned lcs = lchunks;
        asm volatile("" : "+v"(lcs));
        const int lc = (lcs >> (2 * j)) & 3;
        xcoef = reinterpret_cast<const float*>(ctab0 + buf * CTAB_BYTES) + lc * VEC * 2;
        xcf[0] = *reinterpret_cast<const float4*>(xcoef);
    };
    auto xf_slot = [&](int s, int j) {
        const int pr = s / XSTRIDE, k = s % XSTRIDE;
        if (pr >= XPAIRS || k > 3) return;
        if (k == 0) {
            const unsigned* w = reinterpret_cast<const unsigned*>(&xraw);
            float f0, f1;
            if (sizeof(T) == 2) { f0 = __uint_as_float(w[pr] << 16); f1 = __uint_as_float(w[pr] & 0xffff0000u); }
            else { f0 = __uint_as_float(w[2 * pr]); f1 = __uint_as_float(w[2 * pr + 1]); }
            const float4 ad = xcf[0];
            // one-element instructions on purpose (fma1 / div2): left to itself the compiler pairs these into
            // v_pk_fma_f32 / v_pk_mul_f32, which needs three register moves per pair to line up (a0, a1) and
            // (d0, d1) from the interleaved table and, issued beside MFMAs, costs more than the two plain ones
            xt[0] = fma1(f0, ad.x, ad.y); xt[1] = fma1(f1, ad.z, ad.w);
            // exponent of the sigmoid: -t*log2(e) with the activation, -200 without it (exp2 gives 0, the
            // reciprocal 1, the product t itself): no select on the way.  PRE: t is the exponent already.
            if (!PRE) { xu[0] = fma1(xt[0], xk1, xk0); xu[1] = fma1(xt[1], xk1, xk0); }
        } else if (k == 1) {
            // the next pair's coefficients are requested into the registers the multiply-adds have just read
            if (pr + 1 < XPAIRS) xcf[0] = *reinterpret_cast<const float4*>(xcoef + 4 * (pr + 1));
            xu[0] = xone + __builtin_amdgcn_exp2f(PRE ? xt[0] : xu[0]); xu[1] = xone + __builtin_amdgcn_exp2f(PRE ? xt[1] : xu[1]);
        } else if (k == 2) {
            div2(xt[0], xt[1], xu[0], xu[1]);
        } else {
            if (sizeof(T) == 2) {
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const f32x2_t v2 = {xt[0], xt[1]};
                xout[pr] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2_t));
            } else { xout[2 * pr] = __float_as_uint(xt[0]); xout[2 * pr + 1] = __float_as_uint(xt[1]); }
            if (pr == XPAIRS - 1) {
                const uint4 o = make_uint4(xout[0], xout[1], xout[2], xout[3]);     // zeros already where padding
                const int item = tid + NT * j;
                if (j < HALO_ROUNDS - 1 || item < HALO_ITEMS)
                    *reinterpret_cast<uint4*>(const_cast<unsigned char*>(xaddr)) = o;
            }
        }
    };

    // The one-pair form of the last round (MINI), stages in slots 0..3
    auto xm_begin = [&](int buf) {
        int mini_off, mini_cf; bool mini_on, mini_pad;
        mini_geom(mini_off, mini_cf, mini_on, mini_pad);
        xaddr = halo0 + buf * HALO_BUF + mini_off;
        if (sizeof(T) == 2) xraw.x = *reinterpret_cast<const unsigned*>(xaddr);
        else { const uint2 w = *reinterpret_cast<const uint2*>(xaddr); xraw.x = w.x; xraw.y = w.y; }
        xcf[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ctab0 + buf * CTAB_BYTES) + mini_cf);
    };
    auto xm_slot = [&](int k) {
        if (k == 0) {
            float f0, f1;
            if (sizeof(T) == 2) { f0 = __uint_as_float(xraw.x << 16); f1 = __uint_as_float(xraw.x & 0xffff0000u); }
            else { f0 = __uint_as_float(xraw.x); f1 = __uint_as_float(xraw.y); }
            const float4 ad = xcf[0];
            xt[0] = fma1(f0, ad.x, ad.y); xt[1] = fma1(f1, ad.z, ad.w);
            if (!PRE) { xu[0] = fma1(xt[0], xk1, xk0); xu[1] = fma1(xt[1], xk1, xk0); }
        } else if (k == 1) {
            xu[0] = 1.0f + __builtin_amdgcn_exp2f(PRE ? xt[0] : xu[0]); xu[1] = 1.0f + __builtin_amdgcn_exp2f(PRE ? xt[1] : xu[1]);
        } else if (k == 2) {
            div2(xt[0], xt[1], xu[0], xu[1]);
        } else if (k == 3) {
            int mini_off, mini_cf; bool mini_on, mini_pad;
            mini_geom(mini_off, mini_cf, mini_on, mini_pad);
            if (mini_pad) { xt[0] = 0.0f; xt[1] = 0.0f; }
            if (mini_on) {
                if (sizeof(T) == 2) {
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t v2 = {xt[0], xt[1]};
                    *reinterpret_cast<unsigned*>(const_cast<unsigned char*>(xaddr)) =
                        __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2_t));
                } else {
                    *reinterpret_cast<float2*>(const_cast<unsigned char*>(xaddr)) = make_float2(xt[0], xt[1]);
                }
            }
        }
    };

    // ---- weight tile DMA: one wave-instruction covers 16 rows of 64 bytes; every wave issues NW of them
    // (for BN < 64 some waves repeat another wave's copy)
    const unsigned wt_base = lds_addr(wt0);
    unsigned wrow_off[NW];           // this lane's row and logical chunk inside the layer's weights, bytes
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int blk = (j * NWAVES + wave) % (BN / 16);   // 16-row block index
        const int r = blk * 16 + (lane >> 2);
        const int lchunk = (lane & 3) ^ ((-(r >> 2)) & 3);
        wrow_off[j] = a.wtiled ? (unsigned)lane * 16u : (unsigned)((r * 9 * Ctot + lchunk * VEC) * (int)sizeof(T));
    }
    auto stage_w = [&](int cc_share, int pos, int buf) {
        const int cc = gchunk(cc_share);
        // K-tiles of a chunk are multiplied column by column of the 3x3 window (loop position pos: kx = pos / 3,
        // ky = pos % 3; see the chunk body), the weights are stored row by row (tap = ky * 3 + kx)
        const int tap = (pos % 3) * 3 + pos / 3;
        // scalar base = first row of this workgroup's channels at (tap, chunk); the lane offset never changes.
        // Tiled weights: the 16 rows x 64 bytes of a copy are one KiB, already in LDS order, at
        // [16-row block][chunk][tap]: scalar base = that KiB, lane offset = lane * 16.
        const T* base = (const T*)a.w + ((size_t)n0 * 9 + tap) * Ctot + cc * CK;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int blk = (j * NWAVES + wave) % (BN / 16);
            const void* bj = a.wtiled ? (const void*)((const unsigned char*)a.w +
                                                      (((size_t)(n0 / 16 + blk) * nchunks + cc) * 9 + tap) * (16 * HROW))
                                      : (const void*)base;
            glds16s(bj, wrow_off[j], __builtin_amdgcn_readfirstlane(wt_base + buf * WTILE_BYTES + blk * 16 * HROW));
        }
    };

    // ---- prologue: the whole halo of chunk 0, its coefficients, weight tiles 0 and 1
#pragma unroll
    for (int k = 0; k < NB4; ++k) asm volatile("" : "+v"(bias4[k]));
    MAD_STAMP_PHASE(11)
    if (FUSE) dma_coef(0, 0);
#pragma unroll
    for (int j = 0; j < HALO_ROUNDS; ++j) dma_item(0, j, 0);
#pragma unroll
    for (int t = 0; t < PF; ++t) stage_w(0, t, t);
    // (the copies are in flight: what follows -- fragment addresses, the accumulators' start values -- runs under
    // their latency instead of in front of it)
    __builtin_amdgcn_sched_barrier(0);

    // ---- fragment addressing.  32x32x16 form: lane&31 = row of a 32-row fragment, lane>>5 = which of two adjacent
    // 16-byte chunks.  16x16x32 form (M16): lane&15 = row of a 16-row fragment (one tile row of pixels / 16
    // channels), lane>>4 = chunk of the 64-byte row.
    const int fr = M16 ? lane & 15 : lane & 31, fh = M16 ? lane >> 4 : lane >> 5;
    int a_off[3][2];     // [kx][kk]: byte offset of this lane's pixel chunk for this wave's first tile row (M16: kk = 0)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int hx = (wm / WMH) * 16 + (fr & 15) + kx;
        const int pix = ((wm % WMH) * (16 / WMH) + (M16 ? 0 : fr >> 4)) * HSX + hx;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a_off[kx][kk] = pix * HROW + ((((M16 ? 0 : 2 * kk) + fh) ^ ((hx >> 1) & 3)) << 4);
    }
    // weight chunk of this lane: row r, and (32x32x16) the K half kk.  Rows 16 apart share their swizzle, so the
    // fragments of a wave differ by a constant
    int b_off[2];
    {
        const int r = wn * (TN * 32) + fr;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b_off[kk] = r * HROW + ((((M16 ? 0 : 2 * kk) + fh) ^ ((-(r >> 2)) & 3)) << 4);
    }

    f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
    f32x4 acc16[M16 ? TM16 : 1][M16 ? TN16 : 1];
    if constexpr (M16) {
        static_assert(NB4 == TN16, "one bias quad per 16-channel fragment");
#pragma unroll
        for (int j = 0; j < TN16; ++j)
#pragma unroll
            for (int i = 0; i < TM16; ++i) acc16[i][j] = bias4[j];
    } else {
        static_assert(NB4 == 4 * TN, "four bias quads per 32-channel fragment");
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g4 + e] = bias4[4 * j + g4][e];
    }

    if (FUSE) {
        // One wait for the table and the whole halo (the weight tiles, issued last, may still be in flight), then
        // every item is read before any is rewritten and the coefficient rows of item j+1 are requested before
        // item j is computed: the items' dependent chains (LDS read, multiply-add, exp, reciprocal, product,
        // pack, LDS write) overlap instead of running one behind the other.
        static_assert(PF * NW < 64, "vmcnt range");
        MAD_STAMP_PHASE(12)
        dma_wait_upto<PF * NW>();
        MAD_STAMP_PHASE(13)
        constexpr int FULL = MINI ? HALO_ROUNDS - 1 : HALO_ROUNDS;
        uint4 raw[FULL];
#pragma unroll
        for (int j = 0; j < FULL; ++j) raw[j] = *reinterpret_cast<const uint4*>(halo0 + (tid + NT * j) * 16);
        auto load_cf = [&](int j, float4* out) {
            const float* cf = reinterpret_cast<const float*>(ctab0) + ((lchunks >> (2 * j)) & 3) * VEC * 2;
#pragma unroll
            for (int q = 0; q < XPAIRS; ++q) out[q] = *reinterpret_cast<const float4*>(cf + 4 * q);
        };
        float4 cfr[2][XPAIRS];
        load_cf(0, cfr[0]);
#pragma unroll
        for (int j = 0; j < FULL; ++j) {
            if (j + 1 < FULL) load_cf(j + 1, cfr[(j + 1) & 1]);
            Vec16<T> v; v.raw = raw[j];
            float f[VEC]; v.unpack(f);
#pragma unroll
            for (int q = 0; q < XPAIRS; ++q) {
                const float4 ad = cfr[j & 1][q];
                float t0 = fma1(f[2 * q], ad.x, ad.y), t1 = fma1(f[2 * q + 1], ad.z, ad.w);
                const float u0 = sig_den(t0), u1 = sig_den(t1);
                div2(t0, t1, u0, u1);
                f[2 * q] = t0; f[2 * q + 1] = t1;
            }
            v.pack(f);
            if ((padmask >> j) & 1) v.raw = make_uint4(0, 0, 0, 0);
            if (j < HALO_ROUNDS - 1 || tid + NT * j < HALO_ITEMS)
                *reinterpret_cast<uint4*>(halo0 + (tid + NT * j) * 16) = v.raw;
        }
        if (MINI) xform_mini(0);
    }
    MAD_STAMP_PHASE(14)
    dma_wait_all();

    // Operations a wave issues at tap t, in order: the coefficient copy (t == 0, FUSE), one halo item (t < 6),
    // then NW weight copies (K-tile t+PF).  Weight tile t was the last thing issued at tap t-PF, so at the top
    // of tap t everything issued at the taps in between may still be in flight:
    //   N(t) = sum over x = t-PF+1 .. t-1 of coef(x) + item(x) + NW          (indices mod 9)
    // and a halo item issued at tap j has landed when the wait of tap j+PF is over.
    MAD_STAMP_PHASE(1)
    // The body of one channel chunk.  The last chunk of a tile has nothing to fetch or transform for a next chunk:
    // its instantiation (LAST) requests only the weight tiles it still needs, and its waits count accordingly.
    auto chunk_body = [&](const int cc, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const unsigned char* halo = halo0 + (cc & 1) * HALO_BUF;
        const int nb = (cc + 1) & 1;                              // halo / table buffer of the next chunk
        uint4 arow[SLIDE ? NSLOT : 1];                            // SLIDE: the wave's pixel-row fragments
        const int ccn = cc + 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // `tap` is the loop position; the window is walked column by column (kx outer, ky inner), so that the
            // pixel rows a wave holds in registers for one ky serve the next ky shifted by one row (SLIDE, below)
            const int kx = tap / 3, ky = tap % 3;
            constexpr int C0 = FUSE ? 1 : 0;
            static_assert(PF >= 2 && PF <= 3, "the in-place transform of item j runs at tap j+3");
            int nwait = 0;
#pragma unroll
            for (int d = 1; d < PF; ++d) {
                const int x = (tap + 9 - d) % 9;
                // tap x of this chunk when x < tap, else of the chunk before (never a LAST one)
                if (LAST && x < tap) nwait += x + PF < 9 ? NW : 0;
                else nwait += (x == 0 ? C0 : 0) + (x < HALO_ROUNDS ? 1 : 0) + NW;
            }
            // weight tile `tap` has landed; after the barrier it and (at tap 0) the finished halo of this
            // chunk are visible to every wave, and every wave is done with the buffers refilled below
            MAD_STAMP(0)
            switch (nwait) {
                case 0: dma_wait_upto<0>(); break;
                case 1: dma_wait_upto<1>(); break;
                case 2: dma_wait_upto<2>(); break;
                case 3: dma_wait_upto<3>(); break;
                case 4: dma_wait_upto<4>(); break;
                case 5: dma_wait_upto<5>(); break;
                case 6: dma_wait_upto<6>(); break;
                case 7: dma_wait_upto<7>(); break;
                default: dma_wait_upto<8>(); break;
            }
            MAD_STAMP(1)
#ifndef MAD_KO_BAR
            __syncthreads();
#endif
            MAD_STAMP(2)
            // buffer of K-tile k = 9*cc + tap is k % NWBUF: a compile-time value when NWBUF divides 9
            auto wbuf = [&](int c_, int t_) { return 9 % NWBUF == 0 ? t_ % NWBUF : (c_ * (9 % NWBUF) + t_) % NWBUF; };
            const unsigned char* wt = wt0 + wbuf(cc, tap) * WTILE_BYTES;
            // item tap-3 of the next chunk was issued three taps ago, before weight tile `tap`: it has landed,
            // and is transformed in place in 16 pieces behind this tap's 16 MFMAs
#ifdef MAD_KO_XF
            const bool do_x = false;
#else
            const bool do_x = FUSE && !LAST && tap >= 3 && tap - 3 < HALO_ROUNDS;
#endif
            const bool mini_tap = MINI && do_x && tap - 3 == HALO_ROUNDS - 1;
            // what rides behind MFMA number sidx of the tap's NMM: the requests for later taps and (unless the
            // transform runs as one block) the transform pieces
            auto behind_mma = [&](int sidx, int NMM) {
#ifndef MAD_KO_DMA
                if (!LAST && sidx == MAD_DMA_HALO_SIDX * NMM / 16) {
                    if (FUSE && tap == 0) dma_coef(ccn, nb);
                    if (tap < HALO_ROUNDS) dma_item(ccn, tap, nb);
                }
                if (sidx == MAD_DMA_W_SIDX * NMM / 16) {
                    if (tap + PF < 9) stage_w(cc, tap + PF, wbuf(cc, tap + PF));
                    else if (!LAST) stage_w(ccn, tap + PF - 9, wbuf(cc + 1, tap + PF - 9));
                }
#endif
                if (do_x && !MAD_XF_PHASED) {
#pragma unroll
                    for (int xs = sidx * 16 / NMM; xs < (sidx + 1) * 16 / NMM; ++xs) {
                        if (mini_tap) xm_slot(xs); else xf_slot(xs, tap - 3);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (do_x && MAD_XF_PHASED) {
                if (mini_tap) xm_begin(nb); else xf_begin(tap - 3, nb);
#pragma unroll
                for (int xs = 0; xs < 16; ++xs) {
                    if (mini_tap) xm_slot(xs); else xf_slot(xs, tap - 3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // bf16: the fragments of a group are requested while the group before it is multiplied (measured
            // a little faster than requesting all of a tap's fragments at its top); the fp32 parity form has no
            // registers to spare and requests each group when it needs it
            constexpr bool AHEAD = sizeof(T) == 2;
            if constexpr (SLIDE) {
                // Sliding window of pixel-row fragments.  A wave multiplies TM16 rows of 16 pixels; at vertical tap ky
                // row i needs halo row i + ky of the wave's rows, at the column shift kx.  The TM16 rows stay in
                // registers: going from ky to ky + 1 replaces ONE row (row ky - 1 + TM16 + 1 takes the registers of row
                // ky - 1 ... i.e. the row that has just dropped out of the window), and during the last products of a
                // window column each row that has served is replaced by the row of the NEXT column that will use its
                // registers (row r of column kx lives in slot (r + 2 * kx) % TM16, so the next column's row i takes
                // over from this column's row i + 2).  LDS fragment reads per chunk: 30 pixel rows + 36 weight
                // fragments instead of 72 + 36.  Rows of the NEXT chunk cannot be requested early (its halo is
                // finished at the barrier of its first tap): the first tap of a chunk reads all its rows at its top.
                // The products run channel fragment by channel fragment (j outer): a weight fragment serves TM16
                // products and is then dead, so two of them are live at a time.
                constexpr int NMM = TM16 * TN16;
                auto slot = [&](int r, int kxg) { return (r + 2 * kxg) % NSLOT; };
                auto load_row = [&](int r, int kxg) {
                    arow[slot(r, kxg)] = *reinterpret_cast<const uint4*>(halo + a_off[kxg][0] + r * HSX * HROW);
                };
                uint4 bfs[2];
                auto load_b = [&](int j) { bfs[j & 1] = *reinterpret_cast<const uint4*>(wt + b_off[0] + j * 16 * HROW); };
                load_b(0);
                if (tap == 0) {
#pragma unroll
                    for (int r = 0; r < TM16; ++r) load_row(r, 0);
                } else if (ky > 0) {
                    load_row(TM16 - 1 + ky, kx);        // the row that enters the window; its first product is the tap's last
                }
                if (do_x && !MAD_XF_PHASED) { if (mini_tap) xm_begin(nb); else xf_begin(tap - 3, nb); }
#pragma unroll
                for (int j = 0; j < TN16; ++j) {
                    if (j + 1 < TN16) load_b(j + 1);
#pragma unroll
                    for (int i = 0; i < TM16; ++i) {
                        // weights are the first operand: a lane's four results are four consecutive channels of a pixel
#if MAD_HALO_ASM_MMA
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                                     : "+v"(acc16[i][j])
                                     : "v"(__builtin_bit_cast(bf16x8, bfs[j & 1])),
                                       "v"(__builtin_bit_cast(bf16x8, arow[slot(i + ky, kx)])));
#else
                        mma16_row<T>(acc16[i][j], bfs[j & 1], arow[slot(i + ky, kx)]);
#endif
                        behind_mma(j * TM16 + i, NMM);
                        // last use of row i + 2 of this column: its registers take the next column's row i
                        if (j == TN16 - 1 && ky == 2 && kx < 2) load_row(i, kx + 1);
                    }
                }
            } else if constexpr (M16) {
                constexpr int NG = TM16 / 2;           // groups of 2*TN16 MFMAs: a pair of pixel rows x all channels
                constexpr int NMM = TM16 * TN16;
                uint4 afs[2][2], bfs[TN16];
#ifdef MAD_KO_LDS
                auto opaque = [](uint4& v) { asm volatile("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); };
                auto read_group = [&](int g) {
                    if (g == 0) {
#pragma unroll
                        for (int q = 0; q < TN16; ++q) opaque(bfs[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) opaque(afs[g & 1][q]);
                };
#else
                auto read_group = [&](int g) {
                    if (g == 0) {
#pragma unroll
                        for (int q = 0; q < TN16; ++q) bfs[q] = *reinterpret_cast<const uint4*>(wt + b_off[0] + q * 16 * HROW);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        afs[g & 1][q] = *reinterpret_cast<const uint4*>(halo + a_off[kx][0] + (2 * g + q + ky) * HSX * HROW);
                };
#endif
                read_group(0);
                if (do_x && !MAD_XF_PHASED) { if (mini_tap) xm_begin(nb); else xf_begin(tap - 3, nb); }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (AHEAD ? g + 1 < NG : g > 0) read_group(AHEAD ? g + 1 : g);
#pragma unroll
                    for (int m = 0; m < 2 * TN16; ++m) {
                        // weights are the first operand: a lane's four results are four consecutive channels of a pixel
                        mma16_row<T>(acc16[2 * g + m / TN16][m % TN16], bfs[m % TN16], afs[g & 1][m / TN16]);
                        behind_mma(g * 2 * TN16 + m, NMM);
                    }
                }
            } else {
                static_assert(TM % 2 == 0, "pixel tiles are multiplied in pairs");
                constexpr int NG = TM;                     // groups of 2*TN MFMAs: (K half, pair of pixel tiles)
                constexpr int NMM = 2 * TM * TN;
                uint4 afs[2][2], bfs[2][TN];
                auto read_group = [&](int g) {
                    const int kk = g / (TM / 2), q0 = 2 * (g % (TM / 2));
                    if (q0 == 0) {
#pragma unroll
                        for (int q = 0; q < TN; ++q) bfs[kk][q] = *reinterpret_cast<const uint4*>(wt + b_off[kk] + q * 32 * HROW);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        afs[g & 1][q] = *reinterpret_cast<const uint4*>(halo + a_off[kx][kk] + (2 * (q0 + q) + ky) * HSX * HROW);
                };
                read_group(0);
                if (do_x && !MAD_XF_PHASED) { if (mini_tap) xm_begin(nb); else xf_begin(tap - 3, nb); }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int kk = g / (TM / 2), q0 = 2 * (g % (TM / 2));
                    if (AHEAD ? g + 1 < NG : g > 0) read_group(AHEAD ? g + 1 : g);
#pragma unroll
                    for (int m = 0; m < 2 * TN; ++m) {
                        // weights are the first operand: accumulator register e of a lane is channel 8*(e/4) +
                        // 4*(lane>>5) + e%4 of pixel lane&31, i.e. four consecutive channels per four registers
                        mma_chunk<T>(acc[q0 + m / TN][m % TN], bfs[kk][m % TN], afs[g & 1][m / TN]);
                        behind_mma(g * 2 * TN + m, NMM);
                    }
                }
            }
            MAD_STAMP(3)
            MAD_STAMP(4)
        }
    };
    for (int cc = 0; cc < nchunks / KS - 1; ++cc) chunk_body(cc, std::false_type{});
    chunk_body(nchunks / KS - 1, std::true_type{});
    dma_wait_all();
    if constexpr (SLIDE && MAD_HALO_ASM_MMA) {
        // the products above are asm statements: the compiler does not know that the accumulators are results of
        // the matrix pipe, so the wait states a read of such a result needs are provided here, once per tile
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    }

    // ---- 1x1 projection of the block input (the residual branch's channel change), accumulated into the
    // same tiles: K is extended by SC1+SC2 channels read at the centre tap from the raw, un-normalised
    // input.  The 16x16 pixels of a chunk are staged compactly (pixel p at byte p*64, chunk index XOR
    // ((p>>2)&3), the weight-tile layout), two buffers, one chunk in flight while one is multiplied.
    if constexpr (SKIP) {
        const int Cs = a.SC1 + a.SC2;
        const int nskip = Cs / CK / KS;            // per share
        constexpr int SK_ROUNDS = TW * 16 * 4 / NT;      // pixels in the order (half, row, column)
        auto dma_skip = [&](int sc, int buf) {
            const int c0 = gchunk(sc) * CK;
            const T* src; int cs, coff;
            if (c0 < a.SC1) { src = (const T*)a.sx1; cs = a.SC1; coff = c0; }
            else { src = (const T*)a.sx2; cs = a.SC2; coff = c0 - a.SC1; }
#pragma unroll
            for (int j = 0; j < SK_ROUNDS; ++j) {
                const int item = tid + NT * j;
                const int pp = item >> 2;
                const int lc = (tid & 3) ^ ((-(pp >> 2)) & 3);
                const size_t pix = ((size_t)b * a.H + ty0 + ((pp >> 4) & 15)) * a.W + tx0 + (pp >> 8) * 16 + (pp & 15);
                glds16(src + pix * cs + coff + lc * VEC,
                       __builtin_amdgcn_readfirstlane(halo_base + buf * HALO_BUF + (NT * j + 64 * wave) * 16));
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int blk = (j * NWAVES + wave) % (BN / 16);
                const int r = blk * 16 + (lane >> 2);
                const int lchunk = (lane & 3) ^ ((-(r >> 2)) & 3);
                const T* g = (const T*)a.sw + (size_t)(n0 + r) * Cs + c0 + lchunk * VEC;
                glds16(g, __builtin_amdgcn_readfirstlane(wt_base + buf * WTILE_BYTES + blk * 16 * HROW));
            }
        };
        int s_off[2];
        {
            const int r = wm * (TM * 32) + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) s_off[kk] = r * HROW + ((((M16 ? 0 : 2 * kk) + fh) ^ ((-(r >> 2)) & 3)) << 4);
        }
        __syncthreads();                     // every wave has finished reading the last 3x3 tiles
        dma_skip(0, 0);
        for (int sc = 0; sc < nskip; ++sc) {
            dma_wait_all();
            __syncthreads();
            if (sc + 1 < nskip) dma_skip(sc + 1, (sc + 1) & 1);
            const unsigned char* sa = halo0 + (sc & 1) * HALO_BUF;
            const unsigned char* wt = wt0 + (sc & 1) * WTILE_BYTES;
            if constexpr (M16) {
                uint4 af[TM16], bfr[TN16];
#pragma unroll
                for (int i = 0; i < TM16; ++i) af[i] = *reinterpret_cast<const uint4*>(sa + s_off[0] + i * 16 * HROW);
#pragma unroll
                for (int jn = 0; jn < TN16; ++jn) bfr[jn] = *reinterpret_cast<const uint4*>(wt + b_off[0] + jn * 16 * HROW);
#pragma unroll
                for (int i = 0; i < TM16; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN16; ++jn) mma16_row<T>(acc16[i][jn], bfr[jn], af[i]);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 af[TM], bfr[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(sa + s_off[kk] + i * 32 * HROW);
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) bfr[jn] = *reinterpret_cast<const uint4*>(wt + b_off[kk] + jn * 32 * HROW);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jn = 0; jn < TN; ++jn) mma_chunk<T>(acc[i][jn], bfr[jn], af[i]);
                }
            }
        }
    }
    MAD_STAMP_PHASE(2)

    // ---- epilogue, 128 pixels (8 rows of one 16-column half) per pass.  Pixels in wave order: L = wm*TM*32 +
    // tile*32 + row-in-tile = (half, row, column) = (L >> 8, (L >> 4) & 15, L & 15).  A lane hands over its
    // pixel's channels four at a time, as 16-byte LDS writes.
    float* ctile = reinterpret_cast<float*>(smem);
    constexpr int WAVE_PIX = TM * 32;                        // pixels owned by one wave
    constexpr int NPASS = TW * 16 / 128;
    static_assert(WAVE_PIX <= 128 && 128 % WAVE_PIX == 0, "a wave's pixels lie inside one pass");
    // this wave's accumulators -> ctile rows rbase.. (fp32, EPI_LD floats per pixel), four consecutive channels per write
    auto hand_over = [&](int rbase) {
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < TM16; ++i)
#pragma unroll
                for (int jn = 0; jn < TN16; ++jn)
                    *reinterpret_cast<f32x4*>(ctile + (rbase + i * 16 + fr) * EPI_LD + wn * (TN * 32) + jn * 16 + 4 * fh) = acc16[i][jn];
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 v4 = {acc[i][jn][4 * g4], acc[i][jn][4 * g4 + 1], acc[i][jn][4 * g4 + 2], acc[i][jn][4 * g4 + 3]};
                        *reinterpret_cast<f32x4*>(ctile + (rbase + i * 32 + fr) * EPI_LD + wn * (TN * 32) + jn * 32 + 8 * g4 + 4 * fh) = v4;
                    }
        }
    };
    auto pix_y = [&](int L) { return ty0 + ((L >> 4) & 15); };
    auto pix_x = [&](int L) { return tx0 + (L >> 8) * 16 + (L & 15); };
    if constexpr (OUT_NCHW) {
        float* yo = (float*)a.y;
#pragma unroll 1
        for (int half = 0; half < NPASS; ++half) {
            __syncthreads();
            if ((wm * WAVE_PIX) / 128 == half) {
                const int rbase = wm * WAVE_PIX - half * 128;
                hand_over(rbase);
            }
            __syncthreads();
            if (tid < 128) {
                const int r = tid;
                const int oy = pix_y(half * 128 + r), ox = pix_x(half * 128 + r);
                for (int n = 0; n < a.Cout; ++n)
                    yo[(((size_t)b * a.Cout + n) * a.H + oy) * a.W + ox] = ctile[r * EPI_LD + n] + a.bias[n];
            }
        }
        return;
    } else if constexpr (DSE) {
        // ---- epilogue straight from the accumulators.  Lane (fr, fh) of a wave holds, for pixel column fr of each
        // of its TM16 rows and each 16-channel block j, the four channels 16*j + 4*fh .. +3.  v_permlane16_swap
        // exchanges, between the lanes fh and fh ^ 1 of a pixel, the quad of block 2p+1 against the quad of block 2p:
        // afterwards an even fh holds channels 4*fh .. 4*fh+7 of block 2p and an odd fh channels 4*(fh-1) .. +7 of
        // block 2p+1 -- eight consecutive channels, one 16-byte store, and the four lanes of a pixel cover 64
        // contiguous bytes.  Residual rows are requested a row group ahead.  Statistics: per lane over its rows (pair
        // sums by v_dot2c_f32_bf16 on the packed output), then over the 16 pixel columns by a DPP butterfly, then over
        // the waves that share channels through LDS in wave order: a fixed order, so results are deterministic.
        constexpr int NP = TN16 / 2;                    // pairs of 16-channel blocks per wave
        constexpr int TMR = TM16 / KS;                  // rows this wave finishes
        if constexpr (KS == 2) {
            // The two shares hold partial sums of the same pixels and channels.  Share 0 finishes the first TMR rows of
            // a wave's TM16, share 1 the others: each hands the quads of the rows it does not finish to the wave with
            // its number in the other share (16 bytes per lane and quad, lane-linear), adds what it receives, and
            // share 1 moves its rows into the first TMR register rows so that the code below is the same for both.
            __syncthreads();                            // every wave is done with the staging buffers
            f32x4* xch = reinterpret_cast<f32x4*>(smem);
            auto slot = [&](int g, int i, int j) { return xch + (((g * NWAVES + wave) * TMR + i) * TN16 + j) * 64 + lane; };
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < TMR; ++i)
#pragma unroll
                    for (int j = 0; j < TN16; ++j) *slot(0, i, j) = acc16[TMR + i][j];
            } else {
#pragma unroll
                for (int i = 0; i < TMR; ++i)
#pragma unroll
                    for (int j = 0; j < TN16; ++j) *slot(1, i, j) = acc16[i][j];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < TMR; ++i)
#pragma unroll
                    for (int j = 0; j < TN16; ++j) acc16[i][j] += *slot(1, i, j);
            } else {
#pragma unroll
                for (int i = 0; i < TMR; ++i)
#pragma unroll
                    for (int j = 0; j < TN16; ++j) acc16[i][j] = *slot(0, i, j) + acc16[TMR + i][j];
            }
        }
        const T* res = (const T*)a.residual;
        T* y = (T*)a.y;
        const int cb0 = wn * (TN * 32) + (fh & 1) * 16 + (fh >> 1) * 8;     // this lane's first channel, pair 0
        float ssum[NP][4], ssq[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) { ssum[p][k] = 0.0f; ssq[p][k] = 0.0f; }
        const int half = (wm * WAVE_PIX) >> 8;          // 16-column half of a 32-wide tile this wave's rows lie in
        const int ox = tx0 + half * 16 + fr;
        const bool has_res = a.res_mode == MAD_RES_SAME || a.res_mode == MAD_RES_UP;
        constexpr int RG = 4;                           // rows per group of residual requests
        static_assert(TMR % RG == 0, "whole row groups");
        auto row_y = [&](int i) { return ty0 + (((wm * WAVE_PIX) >> 4) & 15) + grp * TMR + i; };
#pragma unroll
        for (int i0 = 0; i0 < TMR; i0 += RG) {
            uint4 rpre[RG][NP];
            if (has_res) {
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const int oy = row_y(i0 + r);
                    const size_t rp = a.res_mode == MAD_RES_SAME
                                          ? ((size_t)b * a.H + oy) * a.W + ox
                                          : ((size_t)b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1);
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        rpre[r][p] = *reinterpret_cast<const uint4*>(res + rp * a.Cout + n0 + cb0 + p * 32);
                }
            }
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int i = i0 + r;
                const int oy = row_y(i);
                const size_t m = ((size_t)b * a.H + oy) * a.W + ox;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc16[i][2 * p][e]),
                                                                         __float_as_uint(acc16[i][2 * p + 1][e]), false, false);
                        v[e] = __uint_as_float(sw[0]); v[4 + e] = __uint_as_float(sw[1]);
                    }
                    if (has_res) {
                        Vec16<T> rv; rv.raw = rpre[r][p];
                        float f[8]; rv.unpack(f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += f[e];
                    } else if (a.res_mode == MAD_RES_DOWN) {
                        float sd[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) sd[e] = 0.0f;
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                const size_t rp = ((size_t)b * (2 * a.H) + (2 * oy + dy)) * (2 * a.W) + (2 * ox + dx);
                                Vec16<T> rv; rv.load(res + rp * a.Cout + n0 + cb0 + p * 32);
                                float f[8]; rv.unpack(f);
#pragma unroll
                                for (int e = 0; e < 8; ++e) sd[e] += f[e];
                            }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaf(sd[e], 0.25f, v[e]);
                    }
                    Vec16<T> o; o.pack(v);
                    T* dst = y + m * a.Cout + n0 + cb0 + p * 32;
                    if (a.stream_out) o.store_nt(dst); else o.store(dst);
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
        }
        if (a.stats) {
            // over the 16 pixel columns: lanes ^1, ^2 inside a quad, then the mirrored quad, then the mirrored half row
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
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) { ssum[p][k] = row16_sum(ssum[p][k]); ssq[p][k] = row16_sum(ssq[p][k]); }
            __syncthreads();                              // every wave is done with the staging buffers
            float* red = reinterpret_cast<float*>(smem);  // [KS * WM][BN / 2 channel pairs][2]
            if (fr == 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int pr = (cb0 + p * 32) / 2 + k;
                        red[((grp * WM + wm) * (BN / 2) + pr) * 2] = ssum[p][k];
                        red[((grp * WM + wm) * (BN / 2) + pr) * 2 + 1] = ssq[p][k];
                    }
            }
            __syncthreads();
            if (tid_all < BN / 2) {
                float sm = 0.0f, sq = 0.0f;
#pragma unroll
                for (int w = 0; w < KS * WM; ++w) { sm += red[(w * (BN / 2) + tid) * 2]; sq += red[(w * (BN / 2) + tid) * 2 + 1]; }
                // one row of sums per 16x16 tile of the image, per channel pair: (sum, sum of squares) in the slot of the
                // even channel, zeros in the slot of the odd one; a 32-wide tile writes its sums to the row of its left
                // half and zeros to the row of its right half (the consumer adds the rows up)
                const int tile16 = (tile / a.tiles_x) * (a.W / 16) + (tile % a.tiles_x) * (TW / 16);
                const size_t tiles16_img = (size_t)(a.H / 16) * (a.W / 16);
                float4* srow = reinterpret_cast<float4*>(a.stats + ((b * tiles16_img + tile16) * a.Cout + n0) * 2);
                srow[tid] = make_float4(sm, sq, 0.0f, 0.0f);
                if (TW == 32) {
                    float4* srow1 = reinterpret_cast<float4*>(a.stats + ((b * tiles16_img + tile16 + 1) * a.Cout + n0) * 2);
                    srow1[tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
        }
    } else {
    constexpr int CPR = BN / VEC;            // 16-byte chunks per output row
    constexpr int RPP = NT / CPR;            // rows per pass
    const int er = tid / CPR, ec = (tid % CPR) * VEC;
    float ssum[VEC], ssq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { ssum[i] = 0.0f; ssq[i] = 0.0f; }
    const T* res = (const T*)a.residual;
    T* y = (T*)a.y;
    constexpr int RPT = 128 / RPP;           // rows per thread per pass
#pragma unroll 1
    for (int half = 0; half < NPASS; ++half) {
        // residual rows of this half are requested first: their latency overlaps the tile hand-over
        uint4 rpre[RPT];
        if (a.res_mode == MAD_RES_SAME || a.res_mode == MAD_RES_UP) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = er + k * RPP;
                const int oy = pix_y(half * 128 + r), ox = pix_x(half * 128 + r);
                const size_t rp = a.res_mode == MAD_RES_SAME
                                      ? ((size_t)b * a.H + oy) * a.W + ox
                                      : ((size_t)b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1);
                rpre[k] = *reinterpret_cast<const uint4*>(res + rp * a.Cout + n0 + ec);
            }
        }
        __syncthreads();
        MAD_STAMP_PHASE(3 + 3 * half)
        if ((wm * WAVE_PIX) / 128 == half) {
            const int rbase = wm * WAVE_PIX - half * 128;
            hand_over(rbase);
        }
        __syncthreads();
        MAD_STAMP_PHASE(4 + 3 * half)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = er + k * RPP;
            const int oy = pix_y(half * 128 + r), ox = pix_x(half * 128 + r);
            const size_t m = ((size_t)b * a.H + oy) * a.W + ox;
            float v[VEC];
#pragma unroll
            for (int i = 0; i < VEC; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(ctile + r * EPI_LD + ec + i);
                v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;      // the bias is in the accumulators
            }
            if (a.res_mode == MAD_RES_SAME || a.res_mode == MAD_RES_UP) {
                Vec16<T> rv; rv.raw = rpre[k];
                float f[VEC]; rv.unpack(f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] += f[i];
            } else if (a.res_mode == MAD_RES_DOWN) {
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
            if (a.stream_out) o.store_nt(y + m * a.Cout + n0 + ec);     // larger than the caches: do not allocate
            else o.store(y + m * a.Cout + n0 + ec);
            if constexpr (GSTAT) {
                // sums taken after both halves are stored (below): the accumulators are dead by then
            } else if (a.stats) {
#if MAD_STATS_DOT2
                if constexpr (sizeof(T) == 2) {
                    // sums over channel PAIRS, straight from the packed bf16 pairs (one instruction per pair and
                    // quantity): a GroupNorm group holds whole pairs, and only group sums are ever used
                    const unsigned* w4 = reinterpret_cast<const unsigned*>(&o.raw);
#pragma unroll
                    for (int k = 0; k < VEC / 2; ++k) {
                        asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ssum[2 * k]) : "v"(w4[k]), "v"(0x3f803f80u));
                        asm("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(ssq[2 * k]) : "v"(w4[k]));
                    }
                } else
#endif
                {
                float f[VEC]; o.unpack(f);          // statistics of the values as stored
#pragma unroll
                for (int i = 0; i < VEC; ++i) { ssum[i] += f[i]; ssq[i] = fmaf(f[i], f[i], ssq[i]); }
                }
            }
        }
        MAD_STAMP_PHASE(5 + 3 * half)
    }
    if constexpr (GSTAT) {
        // Each thread reads back the gradient chunks it has just stored (its own writes, in program order) together
        // with the GroupNorm input at the same pixels; with the accumulators dead eight row pairs are in flight
        // at a time.  g = dy * keep * act'(a*x + d); sums of g and g*x per channel.
        const int c = n0 + ec;
        float gca[VEC], gcd[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            gca[i] = a.gcoef[((size_t)b * a.Cout + c + i) * 2];
            gcd[i] = a.gcoef[((size_t)b * a.Cout + c + i) * 2 + 1];
        }
        const T* gsrc; int gcs;
        if (c < a.gC1) { gsrc = (const T*)a.gx1 + c; gcs = a.gC1; }
        else { gsrc = (const T*)a.gx2 + (c - a.gC1); gcs = a.Cout - a.gC1; }
        constexpr int KB = 8;                  // row pairs in flight per batch
        static_assert((NPASS * RPT) % KB == 0, "whole batches");
#pragma unroll 1
        for (int k0 = 0; k0 < NPASS * RPT; k0 += KB) {
            uint4 xq[KB], gq[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int r = er + (k0 + k) * RPP;
                const size_t m = ((size_t)b * a.H + pix_y(r)) * a.W + pix_x(r);
                xq[k] = *reinterpret_cast<const uint4*>(gsrc + m * gcs);
                gq[k] = *reinterpret_cast<const uint4*>(y + m * a.Cout + c);
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int r = er + (k0 + k) * RPP;
                const size_t m = ((size_t)b * a.H + pix_y(r)) * a.W + pix_x(r);
                float f[VEC], xv[VEC];
                Vec16<T> gr; gr.raw = gq[k]; gr.unpack(f);
                Vec16<T> xr; xr.raw = xq[k]; xr.unpack(xv);
                if (a.gdm.thr) {
                    float keep[VEC];
                    drop_factors<VEC>(a.gdm, (unsigned long long)m * a.Cout + c, keep);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) f[i] *= keep[i];
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float gi = a.gact ? f[i] * silu_grad_f(fmaf(xv[i], gca[i], gcd[i])) : f[i];
                    ssum[i] += gi; ssq[i] = fmaf(gi, xv[i], ssq[i]);
                }
            }
        }
    }
    if (a.stats) {
        // reduce the RPP row-groups per channel in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [RPP][BN][2]
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[(er * BN + ec + i) * 2] = ssum[i];
            red[(er * BN + ec + i) * 2 + 1] = ssq[i];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int c = tid >> 1, which = tid & 1;
            float s = 0.0f;
            for (int g = 0; g < RPP; ++g) s += red[(g * BN + c) * 2 + which];
            // one row of sums per 16x16 tile of the image: a 32-wide workgroup tile writes its sums to the row of
            // its left half and zeros to the row of its right half (the consumer adds the rows up)
            const int tile16 = (tile / a.tiles_x) * (a.W / 16) + (tile % a.tiles_x) * (TW / 16);
            const size_t tiles16_img = (size_t)(a.H / 16) * (a.W / 16);
            a.stats[((b * tiles16_img + tile16) * a.Cout + n0 + c) * 2 + which] = s;
            if (TW == 32) a.stats[((b * tiles16_img + tile16 + 1) * a.Cout + n0 + c) * 2 + which] = 0.0f;
        }
    }
    }
#ifdef MAD_HALO_STAMPS
    MAD_STAMP_PHASE(2 + 7)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MAD_STAMP_PHASE(2 + 8)
    if (threadIdx.x == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        g_clock[(blockIdx.x % CLOCK_SLOTS) * 2] = c1 - clk_c0;
        g_clock[(blockIdx.x % CLOCK_SLOTS) * 2 + 1] = r1 - clk_r0;
    }
    if (stamp_on) {
        __syncthreads();
        for (int i = tid; i < NWAVES * 9 * 8; i += NT) g_stamps[i] = stamp_lds[i];
    }
#endif
}

template <typename T, bool FUSE, int WM, int WN, int TN, bool OUT_NCHW, bool SKIP = false, bool GSTAT = false, int TW = 16,
          bool PRE = false, int KS = 1>
static int launch_halo(const HaloArgs& a0, hipStream_t s, const char* what) {
    constexpr int lds = halo_lds_bytes<WN * TN * 32, TW, KS>();
    static int lds_dev = -1;
    MAD_REQUIRE(raise_lds_limit(reinterpret_cast<const void*>(&conv3x3_halo_kernel<T, FUSE, WM, WN, TN, OUT_NCHW, SKIP, GSTAT, TW, PRE, KS>),
                                lds, &lds_dev),
                "%s: cannot raise the dynamic LDS limit to %d", what, lds);
    HaloArgs a = a0;
    a.tiles_x = a.W / TW;
    const int grid = a.B * a.tiles_x * a.tiles_y * a.ntiles_n;
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, FUSE, WM, WN, TN, OUT_NCHW, SKIP, GSTAT, TW, PRE, KS>), dim3(grid), dim3(64 * WM * WN * KS), lds, s, a);
    return check_launch(what);
}

}  // namespace mad

using namespace mad;

// the GroupNorm whose backward partial sums a launch also produces (GSTAT form of the kernel)
struct GnFold {
    const void* gx1; int gC1; const void* gx2; const float* gcoef; int gact; unsigned long long seed; float p;
};

static int conv3x3_fused_impl(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                              const void* w, const float* bias, const void* residual, int res_mode, void* y,
                              float* stats, int dtype, int B, int H, int W, int Cout, int upsample,
                              const void* sx1, int SC1, const void* sx2, int SC2, const void* sw, void* stream,
                              const GnFold* gn = nullptr) {
    MAD_REQUIRE(x1 && w && y, "mad_conv3x3_fused: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv3x3_fused: bad dtype %d", dtype);
    const int CK = dtype == MAD_BF16 ? 32 : 16;
    MAD_REQUIRE(C1 > 0 && C1 % CK == 0 && C2 >= 0 && C2 % CK == 0 && (C2 == 0 || x2),
                "mad_conv3x3_fused: C1=%d, C2=%d must be multiples of %d", C1, C2, CK);
    MAD_REQUIRE(SC1 >= 0 && SC1 % CK == 0 && SC2 >= 0 && SC2 % CK == 0 && (SC1 > 0 || SC2 == 0) &&
                    (SC1 == 0 || (sx1 && sw)) && (SC2 == 0 || sx2),
                "mad_conv3x3_fused_skip: SC1=%d, SC2=%d must be multiples of %d with their tensors given", SC1, SC2, CK);
    MAD_REQUIRE(Cout > 0 && Cout % 64 == 0, "mad_conv3x3_fused: Cout=%d must be a multiple of 64", Cout);
    MAD_REQUIRE(B > 0 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0,
                "mad_conv3x3_fused: H=%d, W=%d must be multiples of 16", H, W);
    MAD_REQUIRE((size_t)B * H * W < (1u << 31), "mad_conv3x3_fused: tensor too large for 32-bit pixel indices");
    // the loader-transform form addresses its input with 24-bit pixel indices and 32-bit byte offsets
    // the loader-transform form addresses a tile's input pixels with 16-bit offsets from the tile's first pixel (18
    // rows of the image) and 32-bit byte offsets
    MAD_REQUIRE(!coef || (W < 3600 && (size_t)18 * W * (C1 > C2 ? C1 : C2) * 4 < ((size_t)1 << 32)),
                "mad_conv3x3_fused: rows of W = %d pixels are too long for the fused loader", W);
    MAD_REQUIRE(res_mode >= MAD_RES_NONE && res_mode <= MAD_RES_DOWN && (res_mode == MAD_RES_NONE || residual),
                "mad_conv3x3_fused: bad residual mode %d", res_mode);
    // act: bit 0 SiLU, bit 1 prescaled coefficients and weights, bit 2 tiled weight layout
    MAD_REQUIRE(act >= 0 && act < 8 && (!(act & MAD_ACT_PRESCALED) || (coef && (act & MAD_ACT_SILU))),
                "mad_conv3x3_fused: act=%d: the prescaled form needs coef and SiLU", act);
    const bool pre = act & MAD_ACT_PRESCALED;
    HaloArgs a;
    a.x1 = x1; a.x2 = C2 ? x2 : x1; a.w = w; a.bias = bias; a.coef = coef; a.residual = residual; a.y = y;
    a.stats = stats;
    a.C1 = C1; a.C2 = C2; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.upsample = upsample; a.res_mode = res_mode;
    a.act = act & MAD_ACT_SILU; a.wtiled = (act & MAD_W_TILED) != 0;
    a.Hs = upsample ? H / 2 : H; a.Ws = upsample ? W / 2 : W;
    a.tiles_x = W / HT; a.tiles_y = H / HT;
    a.sx1 = sx1; a.sx2 = SC2 ? sx2 : sx1; a.sw = sw; a.SC1 = SC1; a.SC2 = SC2;
    a.stream_out = (size_t)B * H * W * Cout * (dtype == MAD_BF16 ? 2 : 4) > ((size_t)256 << 20);
    a.gx1 = a.gx2 = nullptr; a.gcoef = nullptr; a.gC1 = 0; a.gact = 0; a.gdm = make_drop_mask(0, 0.0f);
    if (gn) {
        a.gx1 = gn->gx1; a.gx2 = gn->gC1 < Cout ? gn->gx2 : gn->gx1; a.gcoef = gn->gcoef; a.gC1 = gn->gC1;
        a.gact = gn->gact; a.gdm = make_drop_mask(gn->seed, gn->p);
    }
    hipStream_t s = (hipStream_t)stream;
    const char* what = "mad_conv3x3_fused";
    // 128 output channels per workgroup unless that leaves the chip (256 CUs, two workgroups each) under-filled:
    // then 64 per workgroup.  An odd multiple of 64 is split over two launches: the wide form for the first
    // Cout - 64 channels and the narrow form for the last 64 (each reads the whole input).
    const int wg128 = B * a.tiles_x * a.tiles_y * (Cout / 128);
    // (the narrow tail launch has one workgroup per pixel tile: it must fill the chip on its own)
    const int wgtail = B * a.tiles_x * a.tiles_y;
    int mode = Cout % 128 == 0 ? (wg128 < 256 ? 64 : 128) : (Cout > 64 && wgtail >= 512 ? 192 : 64);
    if (const char* f = getenv("MAD_HALO_BN")) {                                  // tuning aid
        const int v = atoi(f);
        if (v == 64) mode = 64;
        else if (v == 128 && Cout % 128 == 0) mode = 128;
        else if (v == 192 && Cout % 128 == 64 && Cout > 64) mode = 192;
        else if (v == 0 && mode == 192) mode = 64;                                 // no split
    }
    // 16 x 32 pixel tiles on eight waves (one workgroup per CU, half the weight copies per multiply) where the
    // image is wide enough and such tiles still fill the chip; 16 x 16 tiles on four waves otherwise
    const bool wide_fits = dtype == MAD_BF16 && !gn && !pre && W % 32 == 0;
    // (measured: a gain of 3-4 % for layers of 256 and more output channels, a loss of 2-3 % at 128, whose
    // workgroups then have no second workgroup on their CU to overlap their first and last phases with; with the
    // loader transform the four-wave form on the 16x16x32 MFMA is now 1-4 % ahead at every size)
    bool wide_ok = wide_fits && !coef && Cout >= 256 && B * (W / 32) * a.tiles_y * (Cout / 128) >= 256;
    if (const char* f = getenv("MAD_HALO_WIDE"))              // tuning / test aid: 0 = never, 2 = wherever it fits
        wide_ok = atoi(f) == 2 ? wide_fits : (wide_ok && atoi(f) != 0);
    // 256 output channels per workgroup on eight waves (2 x 4, each wave 128 pixels x 64 channels as in the four-wave
    // form; one workgroup per CU): the staged halo and its loader transform serve twice the channels.  Measured
    // +3 % (128x128, 256 channels) to +9 % (32x32, 512 channels) where such workgroups still cover every CU, -25 %
    // where they do not.  The results equal the 128-channel form's to the last bit.
    const bool bn256_fits = pre && !gn && dtype == MAD_BF16 && mode == 128 && Cout % 256 == 0;
    bool bn256 = bn256_fits && B * a.tiles_x * a.tiles_y * (Cout / 256) >= 256;
    if (const char* f = getenv("MAD_HALO_BN256"))            // tuning / test aid: 0 = never, 2 = wherever it fits
        bn256 = atoi(f) == 2 ? bn256_fits : (bn256 && atoi(f) != 0);
    // Two K shares per workgroup (eight waves) where the tiles put no more than one four-wave workgroup on a CU:
    // the low-resolution layers of a small batch.  bf16, output written straight from the accumulators (no loader
    // transform, or the prescaled one at 128 channels per workgroup), whole pairs of channel chunks.
    const int grid1 = B * a.tiles_x * a.tiles_y * (Cout / (mode == 128 ? 128 : 64));
    const bool ks2_fits = dtype == MAD_BF16 && !gn && !wide_ok && !bn256 && mode != 192 && (!coef || (pre && mode == 128)) &&
                          (C1 + C2) % 64 == 0 && (SC1 + SC2) % 64 == 0;
    bool ks2 = ks2_fits && grid1 <= 256;
    if (const char* f = getenv("MAD_HALO_KS"))               // tuning / test aid: 0 = never, 2 = wherever it fits
        ks2 = atoi(f) == 2 ? ks2_fits : (ks2 && atoi(f) != 0);
    auto run = [&](bool narrow) -> int {
        if (ks2) {
            if (coef) return SC1 > 0 ? launch_halo<bf16_t, true, 2, 2, 2, false, true, false, 16, true, 2>(a, s, what)
                                     : launch_halo<bf16_t, true, 2, 2, 2, false, false, false, 16, true, 2>(a, s, what);
            if (narrow) return SC1 > 0 ? launch_halo<bf16_t, false, 2, 2, 1, false, true, false, 16, false, 2>(a, s, what)
                                       : launch_halo<bf16_t, false, 2, 2, 1, false, false, false, 16, false, 2>(a, s, what);
            return SC1 > 0 ? launch_halo<bf16_t, false, 2, 2, 2, false, true, false, 16, false, 2>(a, s, what)
                           : launch_halo<bf16_t, false, 2, 2, 2, false, false, false, 16, false, 2>(a, s, what);
        }
        if (wide_ok && !narrow) {
            if (SC1 > 0)
                return coef ? launch_halo<bf16_t, true, 4, 2, 2, false, true, false, 32>(a, s, what)
                            : launch_halo<bf16_t, false, 4, 2, 2, false, true, false, 32>(a, s, what);
            return coef ? launch_halo<bf16_t, true, 4, 2, 2, false, false, false, 32>(a, s, what)
                        : launch_halo<bf16_t, false, 4, 2, 2, false, false, false, 32>(a, s, what);
        }
#define MAD_HALO_DISPATCH(T_, TN_)                                                                    \
    do {                                                                                              \
        if (gn) return launch_halo<T_, false, 2, 2, TN_, false, false, true>(a, s, what);             \
        if (SC1 > 0) {                                                                                \
            if (pre && bn256 && TN_ == 2 && sizeof(T_) == 2)                                          \
                return launch_halo<bf16_t, true, 2, 4, 2, false, true, false, 16, true>(a, s, what);  \
            if (pre) return launch_halo<T_, true, 2, 2, TN_, false, true, false, 16, true>(a, s, what); \
            return coef ? launch_halo<T_, true, 2, 2, TN_, false, true>(a, s, what)                   \
                        : launch_halo<T_, false, 2, 2, TN_, false, true>(a, s, what);                 \
        }                                                                                             \
        if (pre && bn256 && TN_ == 2 && sizeof(T_) == 2)                                              \
            return launch_halo<bf16_t, true, 2, 4, 2, false, false, false, 16, true>(a, s, what);     \
        if (pre) return launch_halo<T_, true, 2, 2, TN_, false, false, false, 16, true>(a, s, what);  \
        return coef ? launch_halo<T_, true, 2, 2, TN_, false>(a, s, what)                             \
                    : launch_halo<T_, false, 2, 2, TN_, false>(a, s, what);                           \
    } while (0)
        if (dtype == MAD_BF16) {
            if (narrow) MAD_HALO_DISPATCH(bf16_t, 1);
            MAD_HALO_DISPATCH(bf16_t, 2);
        }
        if (narrow) MAD_HALO_DISPATCH(float, 1);
        MAD_HALO_DISPATCH(float, 2);
#undef MAD_HALO_DISPATCH
    };
    a.n_off = 0;
    if (mode == 192) {
        a.ntiles_n = Cout / 128;
        if (int rc = run(false)) return rc;
        a.n_off = Cout - 64; a.ntiles_n = 1;
        return run(true);
    }
    a.ntiles_n = bn256 ? Cout / 256 : Cout / mode;
    return run(mode == 64);
}

extern "C" int mad_conv3x3_fused(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                                 const void* w, const float* bias, const void* residual, int res_mode, void* y,
                                 float* stats, int dtype, int B, int H, int W, int Cout, int upsample, void* stream) {
    return conv3x3_fused_impl(x1, C1, x2, C2, coef, act, w, bias, residual, res_mode, y, stats, dtype, B, H, W, Cout,
                              upsample, nullptr, 0, nullptr, 0, nullptr, stream);
}

extern "C" int mad_conv3x3_fused_skip(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                                      const void* w, const float* bias, const void* sx1, int SC1, const void* sx2,
                                      int SC2, const void* sw, void* y, float* stats, int dtype, int B, int H, int W,
                                      int Cout, void* stream) {
    MAD_REQUIRE(SC1 > 0, "mad_conv3x3_fused_skip: SC1=%d, the projected input needs channels", SC1);
    return conv3x3_fused_impl(x1, C1, x2, C2, coef, act, w, bias, nullptr, MAD_RES_NONE, y, stats, dtype, B, H, W, Cout,
                              0, sx1, SC1, sx2, SC2, sw, stream);
}

extern "C" int mad_conv3x3_gn_bwd_sums(const void* x, int Cin, const void* w, void* y, float* gsum, const void* gx1,
                                       int gC1, const void* gx2, int gC2, const float* gcoef, int act,
                                       unsigned long long drop_seed, float drop_p, int dtype, int B, int H, int W,
                                       int Cout, void* stream) {
    MAD_REQUIRE(gsum && gx1 && gcoef, "mad_conv3x3_gn_bwd_sums: null pointer");
    const int VEC = dtype == MAD_BF16 ? 8 : 4;
    MAD_REQUIRE(gC1 > 0 && gC2 >= 0 && gC1 + gC2 == Cout && gC1 % VEC == 0 && (gC2 == 0 || gx2),
                "mad_conv3x3_gn_bwd_sums: the GroupNorm input has %d + %d channels, the gradient %d", gC1, gC2, Cout);
    MAD_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mad_conv3x3_gn_bwd_sums: p=%f out of [0,1)", (double)drop_p);
    const GnFold gn = {gx1, gC1, gx2, gcoef, act, drop_seed, drop_p};
    return conv3x3_fused_impl(x, Cin, nullptr, 0, nullptr, 0, w, nullptr, nullptr, MAD_RES_NONE, y, gsum, dtype, B, H, W,
                              Cout, 0, nullptr, 0, nullptr, 0, nullptr, stream, &gn);
}

#ifdef MAD_HALO_STAMPS
extern "C" int mad_debug_read_stamps(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned) * 8 * 9 * 8) == hipSuccess ? 0 : 1;
}
// (cycles, reference ticks) of the workgroups of the last launches: `n` pairs, n <= 8192
extern "C" int mad_debug_read_clock(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock), sizeof(unsigned long long) * 2 * (size_t)n) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int mad_conv_out_fused(const void* x, int Cin, const float* coef, int act, const void* w, const float* bias,
                                  float* y, int dtype, int B, int H, int W, int Cout, void* stream) {
    MAD_REQUIRE(x && w && bias && y, "mad_conv_out_fused: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv_out_fused: bad dtype %d", dtype);
    const int CK = dtype == MAD_BF16 ? 32 : 16;
    MAD_REQUIRE(Cin > 0 && Cin % CK == 0, "mad_conv_out_fused: Cin=%d must be a multiple of %d", Cin, CK);
    MAD_REQUIRE(Cout >= 1 && Cout <= 32, "mad_conv_out_fused: Cout=%d must be 1..32", Cout);
    MAD_REQUIRE(B > 0 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0,
                "mad_conv_out_fused: H=%d, W=%d must be multiples of 16", H, W);
    MAD_REQUIRE((size_t)B * H * W < (1u << 31), "mad_conv_out_fused: tensor too large for 32-bit pixel indices");
    // up to three output channels: the nine taps are the N dimension of one product per halo pixel (conv_out.hip)
    if (dtype == MAD_BF16 && conv_out_taps_ok(Cin, H, W, Cout) && !getenv("MAD_CONV_OUT_HALO"))
        return conv_out_taps(x, Cin, coef, act, w, bias, y, B, H, W, Cout, (hipStream_t)stream);
    HaloArgs a;
    a.x1 = x; a.x2 = x; a.w = w; a.bias = bias; a.coef = coef; a.residual = nullptr; a.y = y; a.stats = nullptr;
    a.C1 = Cin; a.C2 = 0; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.upsample = 0; a.res_mode = MAD_RES_NONE;
    a.act = act; a.Hs = H; a.Ws = W;
    a.sx1 = a.sx2 = a.sw = nullptr; a.SC1 = a.SC2 = 0; a.stream_out = 0; a.wtiled = 0;
    a.tiles_x = W / HT; a.tiles_y = H / HT; a.ntiles_n = 1; a.n_off = 0;
    hipStream_t s = (hipStream_t)stream;
    const char* what = "mad_conv_out_fused";
    if (dtype == MAD_BF16)
        return coef ? launch_halo<bf16_t, true, 4, 1, 1, true>(a, s, what) : launch_halo<bf16_t, false, 4, 1, 1, true>(a, s, what);
    return coef ? launch_halo<float, true, 4, 1, 1, true>(a, s, what) : launch_halo<float, false, 4, 1, 1, true>(a, s, what);
}
