// 3x3 convolution for the high-resolution layers: halo-tile implicit GEMM on MFMA with the
// GroupNorm apply + SiLU of the *input* fused into the operand loader and the GroupNorm
// statistics of the *output* produced by the epilogue.
//
//   Block  = one 16x16 output tile of one image x BN output channels, 4 waves laid out WM x WN; a
//            wave owns 16/WM tile rows x TN*32 channels.  BN = 128 (WM=2, WN=2, TN=2: 8 accumulator
//            tiles of 32x32 per wave) for the body of the network; BN = 32 (WM=4, WN=1, TN=1) for
//            the last layer, whose 3 output channels are zero-padded to 32 and written as NCHW fp32.
//   K loop = for each 64-byte channel chunk: the 18x18 input halo of the chunk is staged ONCE into
//            LDS (through registers: load -> y = silu(x*a+d) -> ds_write; zeros outside the
//            image), then the 9 taps are 9 K-tiles that read it at shifted addresses.  Weight
//            tiles (128 rows x 64 bytes per tap) arrive by LDS-DMA, double-buffered; the halo is
//            double-buffered too and the staging of chunk c+1 is spread over the taps of chunk c
//            (one 16-byte item per thread per tap) so its VALU work overlaps the MFMAs.
//   LDS    = halo pixels at 64-byte pitch, 18 pixels per row, chunk index XOR ((x>>1)&3);
//            weight rows at 64-byte pitch, chunk index XOR ((row>>2)&3): both make every
//            ds_read_b128 fragment read bank-conflict free for all nine tap shifts.
//   Epilogue = per half tile: fp32 accumulators -> LDS -> coalesced 16-byte stores with bias and
//            residual; per-channel sum / sum of squares of the stored values go to a
//            (B, tiles, Cout, 2) slab in a fixed order (deterministic).
#include "mma.h"

namespace mad {

struct HaloArgs {
    const void* x1; const void* x2; const void* w; const float* bias; const float* coef;
    const void* residual; void* y; float* stats;
    int C1, C2, B, H, W, Cout, upsample, res_mode, act;
    int Hs, Ws, tiles_x, tiles_y, ntiles_n;
};

constexpr int HT = 16;                 // output tile edge
constexpr int HS = 18;                 // halo edge
constexpr int HROW = 64;               // bytes per halo pixel / weight row per K-tile
constexpr int HALO_BYTES = HS * HS * HROW;          // 20736
constexpr int HALO_ITEMS = HS * HS * 4;             // 16-byte items per halo chunk: 1296
constexpr int HALO_ROUNDS = (HALO_ITEMS + 255) / 256;   // 6

template <int BN> constexpr int halo_lds_bytes() {
    constexpr int stage = 2 * HALO_BYTES + 2 * BN * HROW;
    constexpr int epi = 128 * (BN + 4) * 4;
    return stage > epi ? stage : epi;
}

// OUT_NCHW: write the first a.Cout (<= BN) channels as NCHW fp32 with bias (last layer).
template <typename T, bool FUSE, int WM, int WN, int TN, bool OUT_NCHW>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloArgs a) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BN = WN * TN * 32;
    constexpr int TM = 8 / WM;                     // 32-pixel accumulator tiles per wave (2 tile rows each)
    constexpr int WTILE_BYTES = BN * HROW;
    constexpr int EPI_LD = BN + 4;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CK = HROW / (int)sizeof(T);      // channels per chunk: 32 (bf16) / 16 (fp32)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo0 = smem;
    unsigned char* const wt0 = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const int ty0 = (tile / a.tiles_x) * HT, tx0 = (tile % a.tiles_x) * HT;
    const int n0 = nt * BN;

    const int Ctot = a.C1 + a.C2;
    const int nchunks = Ctot / CK;

    // ---- halo staging assignment: item = tid + 256*j -> (pixel p, 16-byte chunk c4); c4 is fixed per thread
    const int c4 = tid & 3;
    int src_pix[HALO_ROUNDS];        // source pixel index (b*Hs + sy)*Ws + sx, or -1 for padding / no item
    int lds_off[HALO_ROUNDS];
#pragma unroll
    for (int j = 0; j < HALO_ROUNDS; ++j) {
        const int item = tid + 256 * j;
        const int p = item >> 2;
        const int hy = p / HS, hx = p - hy * HS;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool ok = item < HALO_ITEMS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const int sy = a.upsample ? iy >> 1 : iy, sx = a.upsample ? ix >> 1 : ix;
        src_pix[j] = ok ? (b * a.Hs + sy) * a.Ws + sx : -1;
        lds_off[j] = item < HALO_ITEMS ? p * HROW + ((c4 ^ ((hx >> 1) & 3)) << 4) : -1;
    }

    auto chunk_src = [&](int cc, const T*& src, int& cs, int& coff) {
        const int c0 = cc * CK;
        if (c0 < a.C1) { src = (const T*)a.x1; cs = a.C1; coff = c0; }
        else { src = (const T*)a.x2; cs = a.C2; coff = c0 - a.C1; }
    };
    auto load_item = [&](int cc, int j) -> uint4 {
        const T* src; int cs, coff;
        chunk_src(cc, src, cs, coff);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (src_pix[j] >= 0) v = *reinterpret_cast<const uint4*>(src + (size_t)src_pix[j] * cs + coff + c4 * VEC);
        return v;
    };
    // per-thread GroupNorm coefficients (a, d) of the chunk being staged: this thread's VEC channels
    float cf_a[VEC], cf_d[VEC];
    auto load_coef = [&](int cc) {
        if (!FUSE) return;
        const float* cf = a.coef + ((size_t)b * Ctot + cc * CK + c4 * VEC) * 2;
#pragma unroll
        for (int i = 0; i < VEC; i += 2) {
            const float4 ad = *reinterpret_cast<const float4*>(cf + 2 * i);
            cf_a[i] = ad.x; cf_d[i] = ad.y; cf_a[i + 1] = ad.z; cf_d[i + 1] = ad.w;
        }
    };
    auto put_item = [&](int j, uint4 raw, int buf) {
        if (lds_off[j] < 0) return;
        if (FUSE && src_pix[j] >= 0) {
            Vec16<T> v; v.raw = raw;
            float f[VEC]; v.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float t = fmaf(f[i], cf_a[i], cf_d[i]);
                f[i] = a.act ? silu_f(t) : t;
            }
            v.pack(f);
            raw = v.raw;
        }
        *reinterpret_cast<uint4*>(halo0 + buf * HALO_BYTES + lds_off[j]) = raw;
    };

    // ---- weight tile DMA: one wave-instruction covers 16 rows of 64 bytes; BN/16 instructions per tile
    const unsigned wt_base = lds_addr(wt0);
    auto stage_w = [&](int cc, int tap, int buf) {
#pragma unroll
        for (int j = 0; j < (BN / 16 + 3) / 4; ++j) {
            const int blk = j * 4 + wave;                 // 16-row block index
            if (blk < BN / 16) {
                const int r = blk * 16 + (lane >> 2);
                const int lchunk = (lane & 3) ^ ((r >> 2) & 3);
                const T* g = (const T*)a.w + ((size_t)(n0 + r) * 9 + tap) * Ctot + cc * CK + lchunk * VEC;
                glds16(g, __builtin_amdgcn_readfirstlane(wt_base + buf * WTILE_BYTES + blk * 16 * HROW));
            }
        }
    };

    // ---- fragment addressing
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[3][2];     // [kx][kk]: byte offset of this lane's A chunk for tile row wm*8, before the (2i+ky) row term
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int hx = (fr & 15) + kx;
        const int pix = (wm * (16 / WM) + (fr >> 4)) * HS + hx;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a_off[kx][kk] = pix * HROW + (((2 * kk + fh) ^ ((hx >> 1) & 3)) << 4);
    }
    int b_off[TN][2];    // [jn][kk]
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int r = wn * (TN * 32) + jn * 32 + fr;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b_off[jn][kk] = r * HROW + (((2 * kk + fh) ^ ((r >> 2) & 3)) << 4);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- prologue: halo of chunk 0 and the first weight tile
    stage_w(0, 0, 0);
    load_coef(0);
#pragma unroll
    for (int j = 0; j < HALO_ROUNDS; ++j) put_item(j, load_item(0, j), 0);

    int kt = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
        const unsigned char* halo = halo0 + (cc & 1) * HALO_BYTES;
        const bool more = cc + 1 < nchunks;
        uint4 pending = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++kt) {
            const int ky = tap / 3, kx = tap % 3;
            // weight tile kt and (at tap 0) the halo of this chunk are complete and visible after this
            // barrier; it also means every wave has finished the K-tile that last used the buffers
            // about to be refilled
            dma_wait_all();
            __syncthreads();
            if (tap < 8) stage_w(cc, tap + 1, (kt + 1) & 1);
            else if (more) stage_w(cc + 1, 0, (kt + 1) & 1);
            // staging of the next chunk's halo, one item per tap: issue the load of item `tap`, finish item `tap-1`
            uint4 fresh = make_uint4(0, 0, 0, 0);
            if (more && tap == 0) load_coef(cc + 1);
            if (more && tap < HALO_ROUNDS) fresh = load_item(cc + 1, tap);
            const unsigned char* wt = wt0 + (kt & 1) * WTILE_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const uint4*>(halo + a_off[kx][kk] + (2 * i + ky) * HS * HROW);
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) bfr[jn] = *reinterpret_cast<const uint4*>(wt + b_off[jn][kk]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) mma_chunk<T>(acc[i][jn], af[i], bfr[jn]);
            }
            if (more && tap >= 1 && tap <= HALO_ROUNDS) put_item(tap - 1, pending, (cc + 1) & 1);
            pending = fresh;
        }
    }

    // ---- epilogue, one half tile (8 tile rows = 128 pixels) at a time
    float* ctile = reinterpret_cast<float*>(smem);
    constexpr int ROWS_PER_WAVE = 256 / WM;                  // pixels owned by one wave row
    if constexpr (OUT_NCHW) {
        float* yo = (float*)a.y;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if ((wm * ROWS_PER_WAVE) / 128 == half) {
                const int rbase = wm * ROWS_PER_WAVE - half * 128;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            ctile[(rbase + i * 32 + acc_row(e, lane)) * EPI_LD + wn * (TN * 32) + jn * 32 + (lane & 31)] =
                                acc[i][jn][e];
            }
            __syncthreads();
            if (tid < 128) {
                const int r = tid;
                const int oy = ty0 + half * 8 + (r >> 4), ox = tx0 + (r & 15);
                for (int n = 0; n < a.Cout; ++n)
                    yo[(((size_t)b * a.Cout + n) * a.H + oy) * a.W + ox] = ctile[r * EPI_LD + n] + a.bias[n];
            }
        }
        return;
    } else {
    constexpr int CPR = BN / VEC;            // 16-byte chunks per output row
    constexpr int RPP = 256 / CPR;           // rows per pass
    const int er = tid / CPR, ec = (tid % CPR) * VEC;
    float bias_v[VEC], ssum[VEC], ssq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { bias_v[i] = a.bias ? a.bias[n0 + ec + i] : 0.0f; ssum[i] = 0.0f; ssq[i] = 0.0f; }
    const T* res = (const T*)a.residual;
    T* y = (T*)a.y;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((wm * ROWS_PER_WAVE) / 128 == half) {
            const int rbase = wm * ROWS_PER_WAVE - half * 128;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        ctile[(rbase + i * 32 + acc_row(e, lane)) * EPI_LD + wn * (TN * 32) + jn * 32 + (lane & 31)] =
                            acc[i][jn][e];
        }
        __syncthreads();
        for (int r = er; r < 128; r += RPP) {
            const int oy = ty0 + half * 8 + (r >> 4), ox = tx0 + (r & 15);
            const size_t m = ((size_t)b * a.H + oy) * a.W + ox;
            float v[VEC];
#pragma unroll
            for (int i = 0; i < VEC; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(ctile + r * EPI_LD + ec + i);
                v[i] = t.x + bias_v[i]; v[i + 1] = t.y + bias_v[i + 1];
                v[i + 2] = t.z + bias_v[i + 2]; v[i + 3] = t.w + bias_v[i + 3];
            }
            if (a.res_mode == MAD_RES_SAME) {
                Vec16<T> rv; rv.load(res + m * a.Cout + n0 + ec);
                float f[VEC]; rv.unpack(f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] += f[i];
            } else if (a.res_mode == MAD_RES_UP) {
                const size_t rp = ((size_t)b * (a.H >> 1) + (oy >> 1)) * (a.W >> 1) + (ox >> 1);
                Vec16<T> rv; rv.load(res + rp * a.Cout + n0 + ec);
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
            o.store(y + m * a.Cout + n0 + ec);
            if (a.stats) {
                float f[VEC]; o.unpack(f);          // statistics of the values as stored
#pragma unroll
                for (int i = 0; i < VEC; ++i) { ssum[i] += f[i]; ssq[i] = fmaf(f[i], f[i], ssq[i]); }
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
            a.stats[(((size_t)b * tiles_img + tile) * a.Cout + n0 + c) * 2 + which] = s;
        }
    }
    }
}

template <typename T, bool FUSE, int WM, int WN, int TN, bool OUT_NCHW>
static int launch_halo(const HaloArgs& a, hipStream_t s, const char* what) {
    constexpr int lds = halo_lds_bytes<WN * TN * 32>();
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<T, FUSE, WM, WN, TN, OUT_NCHW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int grid = a.B * a.tiles_x * a.tiles_y * a.ntiles_n;
    hipLaunchKernelGGL((conv3x3_halo_kernel<T, FUSE, WM, WN, TN, OUT_NCHW>), dim3(grid), dim3(256), lds, s, a);
    return check_launch(what);
}

}  // namespace mad

using namespace mad;

extern "C" int mad_conv3x3_fused(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                                 const void* w, const float* bias, const void* residual, int res_mode, void* y,
                                 float* stats, int dtype, int B, int H, int W, int Cout, int upsample, void* stream) {
    MAD_REQUIRE(x1 && w && y, "mad_conv3x3_fused: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv3x3_fused: bad dtype %d", dtype);
    const int CK = dtype == MAD_BF16 ? 32 : 16;
    MAD_REQUIRE(C1 > 0 && C1 % CK == 0 && C2 >= 0 && C2 % CK == 0 && (C2 == 0 || x2),
                "mad_conv3x3_fused: C1=%d, C2=%d must be multiples of %d", C1, C2, CK);
    MAD_REQUIRE(Cout > 0 && Cout % 128 == 0, "mad_conv3x3_fused: Cout=%d must be a multiple of 128", Cout);
    MAD_REQUIRE(B > 0 && H >= 16 && W >= 16 && H % 16 == 0 && W % 16 == 0,
                "mad_conv3x3_fused: H=%d, W=%d must be multiples of 16", H, W);
    MAD_REQUIRE((size_t)B * H * W < (1u << 31), "mad_conv3x3_fused: tensor too large for 32-bit pixel indices");
    MAD_REQUIRE(res_mode >= MAD_RES_NONE && res_mode <= MAD_RES_DOWN && (res_mode == MAD_RES_NONE || residual),
                "mad_conv3x3_fused: bad residual mode %d", res_mode);
    HaloArgs a;
    a.x1 = x1; a.x2 = C2 ? x2 : x1; a.w = w; a.bias = bias; a.coef = coef; a.residual = residual; a.y = y;
    a.stats = stats;
    a.C1 = C1; a.C2 = C2; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.upsample = upsample; a.res_mode = res_mode;
    a.act = act;
    a.Hs = upsample ? H / 2 : H; a.Ws = upsample ? W / 2 : W;
    a.tiles_x = W / HT; a.tiles_y = H / HT; a.ntiles_n = Cout / 128;
    hipStream_t s = (hipStream_t)stream;
    const char* what = "mad_conv3x3_fused";
    if (dtype == MAD_BF16)
        return coef ? launch_halo<bf16_t, true, 2, 2, 2, false>(a, s, what) : launch_halo<bf16_t, false, 2, 2, 2, false>(a, s, what);
    return coef ? launch_halo<float, true, 2, 2, 2, false>(a, s, what) : launch_halo<float, false, 2, 2, 2, false>(a, s, what);
}

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
    HaloArgs a;
    a.x1 = x; a.x2 = x; a.w = w; a.bias = bias; a.coef = coef; a.residual = nullptr; a.y = y; a.stats = nullptr;
    a.C1 = Cin; a.C2 = 0; a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.upsample = 0; a.res_mode = MAD_RES_NONE;
    a.act = act; a.Hs = H; a.Ws = W;
    a.tiles_x = W / HT; a.tiles_y = H / HT; a.ntiles_n = 1;
    hipStream_t s = (hipStream_t)stream;
    const char* what = "mad_conv_out_fused";
    if (dtype == MAD_BF16)
        return coef ? launch_halo<bf16_t, true, 4, 1, 1, true>(a, s, what) : launch_halo<bf16_t, false, 4, 1, 1, true>(a, s, what);
    return coef ? launch_halo<float, true, 4, 1, 1, true>(a, s, what) : launch_halo<float, false, 4, 1, 1, true>(a, s, what);
}
