// 3x3 convolution, role-split persistent form of the halo-tile kernel (bf16, 128 output channels
// per tile) for the large layers of the denoiser.
//
// One workgroup of 8 waves per CU walks a stream of 16x16-pixel output tiles.  The waves have roles:
//   waves 0-3  MFMA   own 128x64 of the 256x128 tile (8 accumulator tiles of 32x32 each); per K-tile
//                     (tap) they read fragments from LDS, issue 16 MFMAs and copy their quarter of the
//                     weight tile two taps ahead (LDS-DMA, counted vmcnt wait).
//   waves 4-5  LOAD   copy the 18x18 halo of the next 32-channel chunk (two 16-byte items per lane per
//                     tap, taps 0-5) and its GroupNorm coefficient table (LDS-DMA, counted waits).
//   waves 6-7  STORE  during taps 0-2 write the PREVIOUS tile's result -- which the MFMA waves left in
//                     LDS as bf16(acc + bias) -- to memory with the residual added, accumulating the
//                     per-channel sum / sum of squares for the next GroupNorm.
//   waves 4-7         rewrite the landed halo items in place as silu(x*a+d), one item per lane per tap
//                     (taps 3-8).
// One workgroup barrier per tap publishes what the LOAD / XS waves produced during the previous tap
// and frees the buffers they refill next.  The chunk stream runs on across tile boundaries, so a
// tile's prologue (first halo, first weights) and epilogue (store tail) overlap the neighbouring
// tiles' MFMA work instead of idling the matrix pipe.
//
// Same LDS images as the 4-wave kernel: halo pixels at 64-byte pitch, 18 per row, chunk index XOR
// ((x>>1)&3); weight rows at 64-byte pitch, chunk index XOR ((row>>2)&3).
//
// Rounding: the residual is added to the bf16-rounded (acc + bias), i.e. the result is rounded twice
// on residual layers (the 4-wave kernel rounds once).  Statistics are taken of the stored values.
#include "mma.h"
#include <cstdlib>

namespace mad {

struct Halo2Args {
    const bf16_t* x1; const bf16_t* x2; const bf16_t* w; const float* bias; const float* coef;
    const bf16_t* residual; bf16_t* y; float* stats;
    int C1, C2, B, H, W, Cout, upsample, res_mode, act;
    int Hs, Ws, tiles_x, tiles_y, ntiles_n, ntiles_total;
    int skew;       // start skew unit (64-cycle sleeps per step of the workgroup's skew index)
};

namespace h2 {
constexpr int HT = 16, HS = 18, HROW = 64, CK = 32, VEC = 8, BN = 128;
constexpr int HALO_ITEMS = HS * HS * 4;            // 1296 16-byte items per chunk
constexpr int HALO_BUF = 1344 * 16;                // whole wave-instructions
constexpr int WT_BYTES = BN * HROW;                // one tap's weight tile
constexpr int NWBUF = 3;
constexpr int CTAB = 1024;
constexpr int OUT_PITCH = (BN + 8) * 2;            // bytes per staged output row
constexpr int OFF_WT = 2 * HALO_BUF;
constexpr int OFF_CTAB = OFF_WT + NWBUF * WT_BYTES;
constexpr int OFF_SINK = OFF_CTAB + 2 * CTAB;
constexpr int OFF_OUT = OFF_SINK + 1024;
constexpr int OFF_RED = OFF_OUT + 256 * OUT_PITCH;
constexpr int LDS_BYTES = OFF_RED + 8 * BN * 2 * 4;
constexpr int ROUNDS = 11;                         // halo items per LOAD / XS lane per chunk (128 lanes)
constexpr int STORE_ITEMS = 32;                    // 16-byte output pieces per XS lane per tile
}  // namespace h2

// Per-tap barrier protocol.  Every wave executes exactly: 1 barrier after the prologue copies, one
// barrier at the top of each of the my_tiles * nchunks * 9 taps, and 2 barriers in the drain.  The three
// role loops below have identical trip counts, so the counts agree by construction.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding
// global stores (a release fence), which would stall the XS waves at every tap; nothing here is
// communicated between waves through global memory, so LDS completion + the barrier is enough.  The
// "memory" clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef MAD_H2_STAMPS
// Timing aid (separate build): cycles one wave of each role of workgroup 0 spent per segment, summed over the run.
__device__ unsigned long long g_h2_stamps[3 * 4];
#define H2_T(var) const unsigned long long var = __builtin_readcyclecounter();
#define H2_ACC(slot, t0, t1) acc_t[slot] += (t1) - (t0);
#define H2_DECL unsigned long long acc_t[4] = {0, 0, 0, 0};
#define H2_FLUSH(role) if (blockIdx.x == 0 && lane == 0 && (wave & 1) == 0) { for (int i_ = 0; i_ < 4; ++i_) g_h2_stamps[(role) * 4 + i_] = acc_t[i_]; }
#else
#define H2_T(var)
#define H2_ACC(slot, t0, t1)
#define H2_DECL
#define H2_FLUSH(role)
#endif

template <bool FUSE>
__global__ __launch_bounds__(512, 2) void conv3x3_halo2_kernel(Halo2Args a) {
    using namespace h2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo0 = smem;
    unsigned char* const wt0 = smem + OFF_WT;
    unsigned char* const ctab0 = smem + OFF_CTAB;
    unsigned char* const out0 = smem + OFF_OUT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rtid = tid & 127;                    // index inside the two-wave LOAD / XS group
    const int rw = wave & 1;                       // wave inside that group

    // ---- the workgroup's tile stream: tile k of this workgroup is logical tile k*G + perm(blockIdx),
    // perm giving each XCD (blockIdx & 7) a contiguous range so concurrent tiles of one L2 are neighbours
    const int G = gridDim.x;
    const int perm = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int my_tiles = (a.ntiles_total - perm + G - 1) / G;
    if (my_tiles <= 0) return;                     // whole workgroup: fewer tiles than workgroups
    // All workgroups run the same tap sequence; started together they would issue their stores and copies
    // in the same instants.  A start skew of up to about one tap (kept for the whole run, since every
    // workgroup's taps take the same time) spreads that traffic.
    if (a.skew > 0)
        for (int i = ((blockIdx.x * 5) & 15) * a.skew; i > 0; --i) __builtin_amdgcn_s_sleep(1);
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int Ctot = a.C1 + a.C2;
    const int nchunks = Ctot / CK;

    struct TileCoord { int b, ty0, tx0, n0, tile; };
    auto decode = [&](int k) {
        int L = k * G + perm;
        TileCoord t;
        const int nt = L % a.ntiles_n; L /= a.ntiles_n;
        t.tile = L % tiles_img; t.b = L / tiles_img;
        t.ty0 = (t.tile / a.tiles_x) * HT; t.tx0 = (t.tile % a.tiles_x) * HT; t.n0 = nt * BN;
        return t;
    };

    // ---- in-place transform of landed halo items, shared by waves 4-7: lane hid handles item hid + 256*r
    const int hid = tid - 256;
    constexpr int XROUNDS = (HALO_ITEMS + 255) / 256;                  // 6, the last one 16 items
    auto xf_setup = [&](const TileCoord& t, unsigned& lch, unsigned& pad) {
        lch = 0; pad = 0;
#pragma unroll
        for (int r = 0; r < XROUNDS; ++r) {
            const int item = hid + 256 * r;
            const int p = item >> 2;
            const int hy = p / HS, hx = p - hy * HS;
            const int iy = t.ty0 - 1 + hy, ix = t.tx0 - 1 + hx;
            const bool ok = item < HALO_ITEMS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            lch |= (unsigned)((hid & 3) ^ ((hx >> 1) & 3)) << (2 * r);
            pad |= (ok ? 0u : 1u) << r;
        }
    };
    auto xf_item = [&](int r, int buf, unsigned lch, unsigned pad) {   // y = act(x*a + d), zeros outside the image
        if (!FUSE) return;
        const int item = hid + 256 * r;
        if (r == XROUNDS - 1 && item >= HALO_ITEMS) return;
        unsigned char* addr = halo0 + buf * HALO_BUF + item * 16;
        Vec16<bf16_t> v; v.raw = *reinterpret_cast<const uint4*>(addr);
        const int lc = (lch >> (2 * r)) & 3;
        const float* cf = reinterpret_cast<const float*>(ctab0 + buf * CTAB) + lc * VEC * 2;
        float f[VEC]; v.unpack(f);
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
            const float4 ad = *reinterpret_cast<const float4*>(cf + 2 * i);
            const float t0 = fmaf(f[i], ad.x, ad.y), t1 = fmaf(f[i + 1], ad.z, ad.w);
            f[i] = a.act ? silu_f(t0) : t0;
            f[i + 1] = a.act ? silu_f(t1) : t1;
        }
        v.pack(f);
        if ((pad >> r) & 1) v.raw = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(addr) = v.raw;
    };

    if (wave < 4) {
        // ============================================================ MFMA waves
        const int wm = wave >> 1, wn = wave & 1;
        const int fr = lane & 31, fh = lane >> 5;
        int a_off[3][2], b_off[2][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = (fr & 15) + kx;
            const int pix = (wm * 8 + (fr >> 4)) * HS + hx;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) a_off[kx][kk] = pix * HROW + (((2 * kk + fh) ^ ((hx >> 1) & 3)) << 4);
        }
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int r = wn * 64 + jn * 32 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b_off[jn][kk] = r * HROW + (((2 * kk + fh) ^ ((r >> 2) & 3)) << 4);
        }
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        H2_DECL
        const unsigned wt_base = lds_addr(wt0);
        auto dma_weights = [&](const TileCoord& t, int cc, int tap, int buf) {   // this wave's 2 of the 8 16-row pieces
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int blk = wave * 2 + j;
                const int r = blk * 16 + (lane >> 2);
                const int lchunk = (lane & 3) ^ ((r >> 2) & 3);
                const bf16_t* g = a.w + ((size_t)(t.n0 + r) * 9 + tap) * Ctot + cc * CK + lchunk * VEC;
                glds16(g, __builtin_amdgcn_readfirstlane(wt_base + buf * WT_BYTES + blk * 16 * HROW));
            }
        };
        {
            const TileCoord t0 = decode(0);
            dma_weights(t0, 0, 0, 0);
            dma_weights(t0, 0, 1, 1);
            dma_wait_all();
        }
        wg_barrier();                               // prologue copies have landed
        int gc = 0, gw = 0;
        for (int k = 0; k < my_tiles; ++k) {
            const TileCoord cur_tc = decode(k);
            const bool has_next_tile = k + 1 < my_tiles;
            const TileCoord next_tc = decode(has_next_tile ? k + 1 : k);
            for (int cc = 0; cc < nchunks; ++cc, ++gc) {
                const unsigned char* halo = halo0 + (gc & 1) * HALO_BUF;
                const bool last_chunk = cc + 1 == nchunks;
                const int ncc = last_chunk ? (has_next_tile ? 0 : cc) : cc + 1;
                const TileCoord ntc = (last_chunk && has_next_tile) ? next_tc : cur_tc;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    H2_T(s0)
                    wg_barrier();
                    H2_T(s1)
                    // weight tile of the tap two ahead, into the buffer that was read one tap ago
                    const int wb = gw == 0 ? 2 : gw - 1;
                    if (tap + 2 < 9) dma_weights(cur_tc, cc, tap + 2, wb);
                    else dma_weights(ntc, ncc, tap + 2 - 9, wb);
                    H2_T(s2)
                    const int ky = tap / 3, kx = tap % 3;
                    const unsigned char* wt = wt0 + gw * WT_BYTES;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        uint4 af[4], bfr[2];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            af[i] = *reinterpret_cast<const uint4*>(halo + a_off[kx][kk] + (2 * i + ky) * HS * HROW);
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn) bfr[jn] = *reinterpret_cast<const uint4*>(wt + b_off[jn][kk]);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jn = 0; jn < 2; ++jn) mma_chunk<bf16_t>(acc[i][jn], af[i], bfr[jn]);
                    }
                    H2_T(s3)
                    dma_wait_upto<2>();                // the next tap's weight pieces (issued one tap ago) have landed
                    H2_T(s4)
                    H2_ACC(0, s0, s1) H2_ACC(1, s1, s2) H2_ACC(2, s2, s3) H2_ACC(3, s3, s4)
                    gw = gw == 2 ? 0 : gw + 1;
                }
            }
            // tile end: hand the result to the XS waves as bf16(acc + bias), restart the accumulators
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int col = wn * 64 + jn * 32 + fr;
                const float bv = a.bias ? a.bias[cur_tc.n0 + col] : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = wm * 128 + i * 32 + acc_row(e, lane);
                        *reinterpret_cast<bf16_t*>(out0 + row * OUT_PITCH + col * 2) = f32_to_bf16(acc[i][jn][e] + bv);
                        acc[i][jn][e] = 0.0f;
                    }
            }
        }
        dma_wait_all();
        wg_barrier();                               // drain 1: the last result is staged
        wg_barrier();                               // drain 2
        H2_FLUSH(0)
    } else if (wave < 6) {
        // ============================================================ LOAD waves (+ transform)
        const unsigned halo_base = lds_addr(halo0), ctab_base = lds_addr(ctab0);
        const unsigned sink_base = lds_addr(smem + OFF_SINK);
        int src_pix[ROUNDS];             // source pixel of this lane's item per copy round, -1 = padding
        unsigned lchunks = 0;            // 2 bits per copy round: logical 16-byte chunk of the item
        unsigned xlch = 0, xpad = 0;     // the same for this lane's transform items
        H2_DECL
        auto setup_items = [&](const TileCoord& t) {
            lchunks = 0;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int item = rtid + 128 * r;
                const int p = item >> 2;
                const int hy = p / HS, hx = p - hy * HS;
                const int iy = t.ty0 - 1 + hy, ix = t.tx0 - 1 + hx;
                const bool ok = item < HALO_ITEMS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const int sy = a.upsample ? iy >> 1 : iy, sx = a.upsample ? ix >> 1 : ix;
                src_pix[r] = ok ? (t.b * a.Hs + sy) * a.Ws + sx : -1;
                lchunks |= (unsigned)((rtid & 3) ^ ((hx >> 1) & 3)) << (2 * r);
            }
            xf_setup(t, xlch, xpad);
        };
        auto dma_item = [&](int cc, int r, int buf) {
            const int c0 = cc * CK;
            const bf16_t* src; int cs, coff;
            if (c0 < a.C1) { src = a.x1; cs = a.C1; coff = c0; }
            else { src = a.x2; cs = a.C2; coff = c0 - a.C1; }
            const int rr = r < ROUNDS ? r : ROUNDS - 1;                         // slot 11 only keeps the op count even
            const int lc = (lchunks >> (2 * rr)) & 3;
            const bf16_t* p = src + (size_t)max(src_pix[rr], 0) * cs + coff + lc * VEC;
            const bool sink = r >= ROUNDS || (r == ROUNDS - 1 && rw != 0);      // wave-uniform
            const void* g = (sink || src_pix[rr] < 0) ? (const void*)g_zero_page : (const void*)p;
            const unsigned dst = sink ? sink_base : halo_base + buf * HALO_BUF + (128 * r + 64 * rw) * 16;
            glds16(g, __builtin_amdgcn_readfirstlane(dst));
        };
        auto dma_coef = [&](const TileCoord& t, int cc, int buf) {   // CK channels x (a, d) fp32 = 256 bytes
            const float* g = a.coef + ((size_t)t.b * Ctot + cc * CK) * 2 + (lane & 15) * 4;
            glds16(g, __builtin_amdgcn_readfirstlane(ctab_base + buf * CTAB));
        };
        {
            const TileCoord t0 = decode(0);
            setup_items(t0);
            if (FUSE) dma_coef(t0, 0, 0);
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) dma_item(0, r, 0);
            dma_wait_all();
        }
        wg_barrier();                               // prologue copies have landed
#pragma unroll
        for (int r = 0; r < XROUNDS; ++r) xf_item(r, 0, xlch, xpad);
        int gc = 0;
        for (int k = 0; k < my_tiles; ++k) {
            const TileCoord cur_tc = decode(k);
            const bool has_next_tile = k + 1 < my_tiles;
            const TileCoord next_tc = decode(has_next_tile ? k + 1 : k);
            for (int cc = 0; cc < nchunks; ++cc, ++gc) {
                const int nb = (gc & 1) ^ 1;
                const bool last_chunk = cc + 1 == nchunks;
                // the chunk staged during this chunk's taps: the next one of the stream (itself at the very end)
                const int ncc = last_chunk ? (has_next_tile ? 0 : cc) : cc + 1;
                const TileCoord ntc = (last_chunk && has_next_tile) ? next_tc : cur_tc;
                if (last_chunk && has_next_tile) setup_items(next_tc);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    H2_T(s0)
                    wg_barrier();
                    H2_T(s1)
                    if (tap >= 3) xf_item(tap - 3, nb, xlch, xpad);
                    H2_T(s2)
                    if (FUSE && tap == 0) dma_coef(ntc, ncc, nb);
                    if (tap < 6) { dma_item(ncc, 2 * tap, nb); dma_item(ncc, 2 * tap + 1, nb); }
                    H2_T(s3)
                    // before the next barrier: everything issued up to tap-2 has landed (transformed from tap+1 on)
                    constexpr int C0 = FUSE ? 1 : 0;
                    const int tp = (tap + 8) % 9;
                    const int allowed = ((tp == 0 ? C0 : 0) + (tp < 6 ? 2 : 0)) + (tap == 0 ? C0 : 0) + (tap < 6 ? 2 : 0);
                    switch (allowed) {
                        case 0: dma_wait_upto<0>(); break;
                        case 1: dma_wait_upto<1>(); break;
                        case 2: dma_wait_upto<2>(); break;
                        case 3: dma_wait_upto<3>(); break;
                        case 4: dma_wait_upto<4>(); break;
                        default: dma_wait_upto<5>(); break;
                    }
                    H2_T(s4)
                    H2_ACC(0, s0, s1) H2_ACC(1, s1, s2) H2_ACC(2, s2, s3) H2_ACC(3, s3, s4)
                }
            }
        }
        dma_wait_all();
        wg_barrier();                               // drain 1
        wg_barrier();                               // drain 2
        H2_FLUSH(1)
    } else {
        // ============================================================ STORE waves (+ transform)
        float* const red = reinterpret_cast<float*>(smem + OFF_RED);
        unsigned xlch = 0, xpad = 0;         // per transform round: logical chunk (2 bits), item outside the image
        H2_DECL
        const int er = rtid >> 4, ecv = rtid & 15;     // rows er + 8k, 16-byte column piece ecv
        TileCoord prev_tc = decode(0);                 // tile whose staged result is being written out
        int stored = STORE_ITEMS;                      // pieces of prev_tc written so far (STORE_ITEMS = nothing pending)
        int stats_stage = 2;                           // 0 accumulating, 1 partials in LDS, 2 done
        float ssum[VEC], ssq[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { ssum[i] = 0.0f; ssq[i] = 0.0f; }
        constexpr int NPRE = 9;                        // residual ring: the piece written at tap t sits in slot t
        uint4 rpre[NPRE];
        auto res_load = [&](const TileCoord& t, int piece) -> uint4 {
            const int row = er + 8 * piece;
            const int oy = t.ty0 + (row >> 4), ox = t.tx0 + (row & 15);
            const size_t rp = a.res_mode == MAD_RES_SAME
                                  ? ((size_t)t.b * a.H + oy) * a.W + ox
                                  : ((size_t)t.b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1);
            return *reinterpret_cast<const uint4*>(a.residual + rp * a.Cout + t.n0 + ecv * VEC);
        };
        auto store_piece = [&](int k, int q) {                   // q: slot in rpre (compile-time after unrolling)
            const int row = er + 8 * k;
            const int oy = prev_tc.ty0 + (row >> 4), ox = prev_tc.tx0 + (row & 15);
            Vec16<bf16_t> v; v.raw = *reinterpret_cast<const uint4*>(out0 + row * OUT_PITCH + ecv * 16);
            if (a.res_mode != MAD_RES_NONE) {
                Vec16<bf16_t> rv; rv.raw = rpre[q];
                float f[VEC], g[VEC]; v.unpack(f); rv.unpack(g);
#pragma unroll
                for (int i = 0; i < VEC; ++i) f[i] += g[i];
                v.pack(f);
            }
            v.store(a.y + (((size_t)prev_tc.b * a.H + oy) * a.W + ox) * a.Cout + prev_tc.n0 + ecv * VEC);
            if (a.stats) {
                float f[VEC]; v.unpack(f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) { ssum[i] += f[i]; ssq[i] = fmaf(f[i], f[i], ssq[i]); }
            }
        };
        auto stats_to_lds = [&]() {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                red[(er * BN + ecv * VEC + i) * 2] = ssum[i];
                red[(er * BN + ecv * VEC + i) * 2 + 1] = ssq[i];
                ssum[i] = 0.0f; ssq[i] = 0.0f;
            }
        };
        auto stats_final = [&]() {                               // 256 (channel, which) sums by 128 lanes
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = rtid + 128 * h, c = idx >> 1, which = idx & 1;
                float s = 0.0f;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += red[(g * BN + c) * 2 + which];
                a.stats[(((size_t)prev_tc.b * tiles_img + prev_tc.tile) * a.Cout + prev_tc.n0 + c) * 2 + which] = s;
            }
        };
        xf_setup(decode(0), xlch, xpad);
        wg_barrier();                               // prologue copies have landed
#pragma unroll
        for (int r = 0; r < XROUNDS; ++r) xf_item(r, 0, xlch, xpad);
        // (the barrier at the top of the first tap publishes the transformed chunk)
        int gc = 0;
        const int ntaps = 9 * nchunks;                 // >= 36: one 16-byte piece per lane per tap covers the 32 pieces
        for (int k = 0; k < my_tiles; ++k) {
            const TileCoord cur_tc = decode(k);
            const bool has_next_tile = k + 1 < my_tiles;
            for (int cc = 0; cc < nchunks; ++cc, ++gc) {
                const int nb = (gc & 1) ^ 1;
                const bool last_chunk = cc + 1 == nchunks;
                if (last_chunk && has_next_tile) xf_setup(decode(k + 1), xlch, xpad);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    H2_T(s0)
                    wg_barrier();
                    H2_T(s1)
                    if (stats_stage == 1) { stats_final(); stats_stage = 2; }
                    // One piece of the previous tile's result per tap, so the writes are spread evenly in time.
                    // Its residual row was requested four taps ago into ring slot `tap`; the request for the
                    // piece four taps ahead follows (near the tile's end: the first pieces of THIS tile's result).
                    const int q = cc * 9 + tap;
                    if (stored < STORE_ITEMS && q < STORE_ITEMS) {
                        store_piece(q, tap);
                        if (++stored == STORE_ITEMS && a.stats) { stats_to_lds(); stats_stage = 1; }
                    }
                    if (a.res_mode != MAD_RES_NONE) {
                        if (q + 4 < STORE_ITEMS) { if (stored < STORE_ITEMS) rpre[(tap + 4) % 9] = res_load(prev_tc, q + 4); }
                        else if (q + 4 >= ntaps) rpre[(tap + 4) % 9] = res_load(cur_tc, q + 4 - ntaps);
                    }
                    H2_T(s2)
                    if (tap >= 3) xf_item(tap - 3, nb, xlch, xpad);
                    H2_T(s3)
                    H2_ACC(0, s0, s1) H2_ACC(1, s1, s2) H2_ACC(2, s2, s3)
                }
            }
            prev_tc = cur_tc;
            stored = 0;
            stats_stage = 0;
        }
        // drain: the last tile's result
        wg_barrier();
        if (stats_stage == 1) { stats_final(); stats_stage = 2; }
#pragma unroll 1
        for (int first = 0; first < STORE_ITEMS; first += 8) {
            if (a.res_mode != MAD_RES_NONE) {
#pragma unroll
                for (int q = 0; q < 8; ++q) rpre[q] = res_load(prev_tc, first + q);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) store_piece(first + q, q);
        }
        if (a.stats) stats_to_lds();
        wg_barrier();
        if (a.stats) stats_final();
        H2_FLUSH(2)
    }
}

}  // namespace mad

using namespace mad;

#ifdef MAD_H2_STAMPS
extern "C" int mad_debug_read_h2_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_h2_stamps), sizeof(unsigned long long) * 12) == hipSuccess ? 0 : 1;
}
#endif

// Entry used by mad_conv3x3_fused for shapes this kernel covers; returns -1 when it does not apply.
int mad_halo2_try(const void* x1, int C1, const void* x2, int C2, const float* coef, int act, const void* w,
                  const float* bias, const void* residual, int res_mode, void* y, float* stats, int B, int H, int W,
                  int Cout, int upsample, hipStream_t stream) {
    using namespace h2;
    // Opt-in: measured on MI355X it matches the 4-wave kernel with the loader transform on (858 vs 870 TFLOP/s
    // at 256x256, C = 128) and trails it without (955 vs 1038), so the 4-wave kernel stays the default.
    const char* f = getenv("MAD_HALO_V2");
    if (!f || atoi(f) == 0) return -1;
    if (Cout % BN != 0 || C1 % CK != 0 || C2 % CK != 0 || res_mode == MAD_RES_DOWN) return -1;
    const int Ctot = C1 + C2;
    if (Ctot / CK < 4) return -1;                      // the store tail is spread over >= 4 chunks (<= 8 pieces each)
    const int ntiles_total = B * (H / HT) * (W / HT) * (Cout / BN);
    int min_tiles = 512;                               // fewer than two tiles per CU: the 4-wave kernel fills the chip better
    if (const char* f = getenv("MAD_HALO_V2_MIN_TILES")) min_tiles = atoi(f);      // test / tuning aid
    if (ntiles_total < min_tiles) return -1;
    Halo2Args a;
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)(C2 ? x2 : x1); a.w = (const bf16_t*)w; a.bias = bias; a.coef = coef;
    a.residual = (const bf16_t*)residual; a.y = (bf16_t*)y; a.stats = stats;
    a.C1 = C1; a.C2 = C2; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.upsample = upsample; a.res_mode = res_mode;
    a.act = act; a.Hs = upsample ? H / 2 : H; a.Ws = upsample ? W / 2 : W;
    a.tiles_x = W / HT; a.tiles_y = H / HT; a.ntiles_n = Cout / BN; a.ntiles_total = ntiles_total;
    a.skew = 0;
    if (const char* f = getenv("MAD_HALO_V2_SKEW")) a.skew = atoi(f);          // tuning aid
    static int lds_dev[2] = {-1, -1};
    if (!raise_lds_limit((const void*)conv3x3_halo2_kernel<true>, LDS_BYTES, &lds_dev[0]) ||
        !raise_lds_limit((const void*)conv3x3_halo2_kernel<false>, LDS_BYTES, &lds_dev[1])) {
        set_error("mad_conv3x3_fused: cannot raise the dynamic LDS limit to %d", LDS_BYTES);
        return 1;
    }
    const int grid = 256;                              // one persistent workgroup per CU
    if (coef)
        hipLaunchKernelGGL(conv3x3_halo2_kernel<true>, dim3(grid), dim3(512), LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL(conv3x3_halo2_kernel<false>, dim3(grid), dim3(512), LDS_BYTES, stream, a);
    return check_launch("mad_conv3x3_fused");
}
