// 3x3 convolution of the 8x8 layers: one workgroup = one whole image x 64 output channels.
//
// At 8x8 the implicit-GEMM form re-reads every input pixel nine times and, with few workgroups per CU,
// is bound by how fast its waves can issue copies (one wave issues about 1 KiB per 140 cycles).  Here the
// image's input -- its 10x10 zero-bordered halo, a slab of up to 384 (bf16) / 192 (fp32) channels at a
// time -- is brought to LDS ONCE per slab, optionally rewritten in place as silu(x*a+d) (the producer's
// GroupNorm, as in the halo-tile kernel), and the nine taps read it at shifted pixel offsets.  Only the
// weight K-tiles (64 rows x 128 bytes) stream during the K loop, two per step through a ring with
// counted vmcnt waits, one 1-KiB piece of each per wave.  The eight waves are two groups of four: both
// cover the same 64x64 output tile, group g multiplying the K-tiles 2j+g, and the two partial tiles are
// added in the epilogue (fp32 tile -> LDS -> 16-byte stores with bias and residual, plus the per-image
// sum / sum of squares for the next GroupNorm).
//
// LDS image of a slab: [K-tile][pixel 0..103][128 bytes], chunk index XOR swz(hy, hx) of the halo pixel
// (hy, hx), swz = (hy + 4*(hx>>1)) & 7.  A fragment read touches 32 output pixels = four 8-pixel runs of four
// 10-pixel halo rows, shifted by the tap; with this function the sixteen lanes that a ds_read_b128 serves in
// one LDS cycle always fall on sixteen different 16-byte bank groups, for all nine taps (the plain
// (pixel>>1)&7 of a dense row-major tile repeats after 16 pixels and collides across the row jumps).
#include "mma.h"

namespace mad {

struct Img8Args {
    const void* x1; const void* x2; const void* w; const float* bias; const float* coef; const void* residual;
    void* y; float* stats;
    int C1, C2, B, Cout, act, res_mode;
};

namespace i8 {
constexpr int NTHREADS = 512, NWAVES = 8;
constexpr int HP = 10, NPIX = HP * HP, PROWS = 104;     // halo pixels, padded to whole 8-row copies
constexpr int ROWB = 128;                               // bytes per pixel per K-tile
constexpr int KT_BYTES = PROWS * ROWB;                  // 13312
constexpr int SLAB_KT = 6;                              // K-tiles per slab
constexpr int BN = 64, WT_BYTES = BN * ROWB;            // one weight K-tile: 8 KiB
constexpr int NST = 4;                                  // ring of steps (two weight K-tiles each)
constexpr int OFF_WT = SLAB_KT * KT_BYTES;
constexpr int OFF_CTAB = OFF_WT + NST * 2 * WT_BYTES;
constexpr int CTAB_BYTES = SLAB_KT * 512;               // slab channels x (a, d) fp32
constexpr int LDS_BYTES = OFF_CTAB + CTAB_BYTES;
constexpr int CLD = BN + 4;                             // fp32 tile row pitch in the epilogue
__device__ __forceinline__ int swz(int hy, int hx) { return (hy + 4 * (hx >> 1)) & 7; }
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");
static_assert(2 * 64 * CLD * 4 <= OFF_WT, "the two partial tiles of the epilogue reuse the slab");
}  // namespace i8

template <typename T, bool FUSE>
__global__ __launch_bounds__(512) void conv3x3_img8_kernel(Img8Args a) {
    using namespace i8;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CK = ROWB / (int)sizeof(T);           // channels per K-tile: 64 / 32
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wt0 = smem + OFF_WT;
    float* const ctab = reinterpret_cast<float*>(smem + OFF_CTAB);
    const unsigned a_base = lds_addr(smem), wt_base = lds_addr(wt0), ctab_base = lds_addr(smem + OFF_CTAB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int ntn = a.Cout / BN;
    const int b = blockIdx.x / ntn, n0 = (blockIdx.x % ntn) * BN;
    const int Ctot = a.C1 + a.C2;
    const int nkt_total = Ctot / CK;                    // channel K-tiles
    const int nslabs = (nkt_total + SLAB_KT - 1) / SLAB_KT;

    // ---- slab copy: pieces of 8 halo pixels x 128 bytes, 13 per K-tile, dealt round-robin to the waves
    const int lrow = lane >> 3, pchunk = lane & 7;
    auto load_slab = [&](int kt0, int nkt) {
        for (int id = wave; id < nkt * (PROWS / 8); id += NWAVES) {
            const int k = id / (PROWS / 8), blk = id - k * (PROWS / 8);
            const int c0 = (kt0 + k) * CK;
            const T* src; int cs, coff;
            if (c0 < a.C1) { src = (const T*)a.x1; cs = a.C1; coff = c0; }
            else { src = (const T*)a.x2; cs = a.C2; coff = c0 - a.C1; }
            const int p = blk * 8 + lrow;
            const int hy = p / HP, hx = p - hy * HP;
            const bool ok = p < NPIX && hy >= 1 && hy <= 8 && hx >= 1 && hx <= 8;
            const int lchunk = pchunk ^ swz(hy, hx);
            const T* g = src + ((size_t)(b * 8 + ((hy - 1) & 7)) * 8 + ((hx - 1) & 7)) * cs + coff + lchunk * VEC;
            glds16(ok ? (const void*)g : (const void*)g_zero_page,
                   __builtin_amdgcn_readfirstlane(a_base + k * KT_BYTES + blk * 8 * ROWB));
        }
        if (FUSE) {                                     // this slab's (a, d) pairs: nkt*CK channels x 8 bytes
            const int bytes = nkt * CK * 8;
            if (wave * 1024 < bytes) {                  // SLAB_KT*512 bytes: at most one copy per wave
                const int off = min(wave * 1024 + lane * 16, bytes - 16);
                glds16(reinterpret_cast<const unsigned char*>(a.coef + ((size_t)b * Ctot + kt0 * CK) * 2) + off,
                       __builtin_amdgcn_readfirstlane(ctab_base + wave * 1024));
            }
        }
    };
    // in-place y = act(x*a + d) of the interior pixels of the slab (the border stays zero)
    auto transform_slab = [&](int nkt) {
        for (int it = tid; it < nkt * 64 * 8; it += NTHREADS) {
            const int k = it >> 9, rem = it & 511, px = rem >> 3, ch = rem & 7;
            const int p = (1 + (px >> 3)) * HP + 1 + (px & 7);
            unsigned char* addr = smem + k * KT_BYTES + p * ROWB + ((ch ^ swz(1 + (px >> 3), 1 + (px & 7))) << 4);
            Vec16<T> v; v.raw = *reinterpret_cast<const uint4*>(addr);
            const float* cf = ctab + (k * CK + ch * VEC) * 2;
            float f[VEC]; v.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; i += 2) {
                const float4 ad = *reinterpret_cast<const float4*>(cf + 2 * i);
                const float t0 = fmaf(f[i], ad.x, ad.y), t1 = fmaf(f[i + 1], ad.z, ad.w);
                f[i] = a.act ? silu_f(t0) : t0;
                f[i + 1] = a.act ? silu_f(t1) : t1;
            }
            v.pack(f);
            *reinterpret_cast<uint4*>(addr) = v.raw;
        }
    };
    // ---- one piece (8 rows x 128 bytes) of a weight K-tile: wave w copies rows 8w..8w+7
    const int wrow_ld = wave * 8 + lrow;
    const T* const wsrc = (const T*)a.w + (size_t)(n0 + wrow_ld) * 9 * Ctot + (pchunk ^ ((wrow_ld >> 1) & 7)) * VEC;
    auto stage_piece = [&](int tap, int ktg, int slot) {   // slot: K-tile slot of the ring (2 per step)
        glds16(wsrc + (size_t)tap * Ctot + ktg * CK,
               __builtin_amdgcn_readfirstlane(wt_base + slot * WT_BYTES + wave * 8 * ROWB));
    };

    // ---- fragment addressing
    const int fr = lane & 31, fh = lane >> 5;
    const int prow = wm * 32 + fr;                      // output pixel of this lane's A row
    const int pbase = (prow >> 3) * HP + (prow & 7);    // its halo pixel at tap (0,0)
    const int brow = wn * 32 + fr;
    const int bsw = (brow >> 1) & 7;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    for (int s = 0; s < nslabs; ++s) {
        const int kt0 = s * SLAB_KT, nkt = min(SLAB_KT, nkt_total - kt0);
        const int nk = 9 * nkt;                         // K-tiles of this slab, (tap, k) with k fastest
        const int nsteps = (nk + 1) >> 1;               // step j = K-tiles 2j (group 0) and 2j+1 (group 1)
        const bool odd = nk & 1;                        // the last step then holds one K-tile
        __syncthreads();                                // the previous slab and its weight tiles are consumed
        load_slab(kt0, nkt);
        // (tap, k) of the next K-tile to request; every wave requests one piece of every K-tile
        int st_tap = 0, st_k = 0, st_slot = 0, st_t = 0;
        auto request_step = [&]() {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (st_t < nk) {
                    stage_piece(st_tap, kt0 + st_k, st_slot + h);
                    if (++st_k == nkt) { st_k = 0; ++st_tap; }
                    ++st_t;
                }
            }
            st_slot = st_slot + 2 == 2 * NST ? 0 : st_slot + 2;
        };
#pragma unroll
        for (int j = 0; j < NST - 1; ++j) request_step();          // nk >= 9: these steps are all whole
        // the slab (issued first) has landed once at most the weight pieces are outstanding
        dma_wait_upto<2 * (NST - 1)>();
        __syncthreads();
        if (FUSE) transform_slab(nkt);

        // Tile fragments are read right after the barrier that publishes their step and multiplied one
        // step later, so the LDS latency runs under the previous step's MFMAs (two register sets).
        int rd_tap = grp / nkt, rd_k = grp % nkt, rd_slot = grp;     // this group's K-tile of the next step
        auto publish = [&](int j) {
            // steps j+1 .. j+NST-2 may stay in flight: two pieces per wave each unless the last one is odd
            if (j + NST - 2 < nsteps - 1 || (j + NST - 2 == nsteps - 1 && !odd)) dma_wait_upto<2 * (NST - 2)>();
            else dma_wait_all();
            __syncthreads();                            // step j is visible (at j = 0 also the transformed slab)
            if (j + NST - 1 < nsteps) request_step();
        };
        auto read_frags = [&](uint4 (&af)[4], uint4 (&bf)[4]) {
            const int ky = rd_tap / 3, kx = rd_tap - ky * 3;
            const int p = pbase + ky * HP + kx;
            const unsigned char* arow = smem + rd_k * KT_BYTES + p * ROWB;
            const int asw = swz((prow >> 3) + ky, (prow & 7) + kx);
            const unsigned char* wrow = wt0 + rd_slot * WT_BYTES + brow * ROWB;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                af[kk] = *reinterpret_cast<const uint4*>(arow + (((2 * kk + fh) ^ asw) << 4));
                bf[kk] = *reinterpret_cast<const uint4*>(wrow + (((2 * kk + fh) ^ bsw) << 4));
            }
            rd_k += 2;
            while (rd_k >= nkt) { rd_k -= nkt; ++rd_tap; }
            rd_slot = rd_slot + 2 >= 2 * NST ? rd_slot + 2 - 2 * NST : rd_slot + 2;
            __builtin_amdgcn_sched_barrier(0);          // keep the reads ahead of the MFMAs that follow
        };
        auto multiply = [&](const uint4 (&af)[4], const uint4 (&bf)[4]) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) mma_chunk<T>(acc, af[kk], bf[kk]);
        };
        // group 1 has no K-tile in an odd last step: it reads tap 8's last tile again (rd_tap would be 9)
        // into registers it never multiplies -- skip both instead
        const int mysteps = (odd && grp) ? nsteps - 1 : nsteps;
        uint4 a0[4], b0[4], a1[4], b1[4];
        publish(0);
        read_frags(a0, b0);
        int j = 1;
        for (; j + 1 < nsteps; j += 2) {
            publish(j);
            read_frags(a1, b1);
            multiply(a0, b0);
            publish(j + 1);
            if (j + 1 < mysteps) read_frags(a0, b0);
            multiply(a1, b1);
        }
        if (j < nsteps) {
            publish(j);
            if (j < mysteps) read_frags(a1, b1);
            multiply(a0, b0);
            if (j < mysteps) multiply(a1, b1);
        } else if (nsteps - 1 < mysteps) {
            multiply(a0, b0);
        }
    }

    // ---- epilogue: the groups' partial tiles -> LDS; sum -> 16-byte stores with bias / residual; statistics
    __syncthreads();
    float* ctile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 16; ++e)
        ctile[(grp * 64 + wm * 32 + acc_row(e, lane)) * CLD + wn * 32 + (lane & 31)] = acc[e];
    constexpr int CPR = BN / VEC, RPP = NTHREADS / CPR, NPASS = 64 / RPP;
    const int er = tid / CPR, ec = (tid % CPR) * VEC;
    const T* res = (const T*)a.residual;
    T* y = (T*)a.y;
    uint4 rpre[NPASS];
    if (a.res_mode == MAD_RES_SAME) {
#pragma unroll
        for (int k = 0; k < NPASS; ++k)
            rpre[k] = *reinterpret_cast<const uint4*>(res + ((size_t)b * 64 + er + k * RPP) * a.Cout + n0 + ec);
    }
    __syncthreads();
    float bias_v[VEC], ssum[VEC], ssq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { bias_v[i] = a.bias ? a.bias[n0 + ec + i] : 0.0f; ssum[i] = 0.0f; ssq[i] = 0.0f; }
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int r = er + k * RPP;
        float v[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(ctile + r * CLD + ec + i);
            const float4 u = *reinterpret_cast<const float4*>(ctile + (64 + r) * CLD + ec + i);
            v[i] = (t.x + u.x) + bias_v[i]; v[i + 1] = (t.y + u.y) + bias_v[i + 1];
            v[i + 2] = (t.z + u.z) + bias_v[i + 2]; v[i + 3] = (t.w + u.w) + bias_v[i + 3];
        }
        if (a.res_mode == MAD_RES_SAME) {
            Vec16<T> rv; rv.raw = rpre[k];
            float f[VEC]; rv.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] += f[i];
        } else if (a.res_mode == MAD_RES_DOWN) {        // residual (B,16,16,Cout): 2x2 mean
            const int oy = r >> 3, ox = r & 7;
            float sm[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) sm[i] = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const size_t rp = ((size_t)b * 16 + (2 * oy + dy)) * 16 + (2 * ox + dx);
                    Vec16<T> rv; rv.load(res + rp * a.Cout + n0 + ec);
                    float f[VEC]; rv.unpack(f);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) sm[i] += f[i];
                }
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = fmaf(sm[i], 0.25f, v[i]);
        }
        Vec16<T> o; o.pack(v);
        o.store(y + ((size_t)b * 64 + r) * a.Cout + n0 + ec);
        if (a.stats) {
            float f[VEC]; o.unpack(f);                  // statistics of the values as stored
#pragma unroll
            for (int i = 0; i < VEC; ++i) { ssum[i] += f[i]; ssq[i] = fmaf(f[i], f[i], ssq[i]); }
        }
    }
    if (a.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);    // [RPP][BN][2]
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[(er * BN + ec + i) * 2] = ssum[i];
            red[(er * BN + ec + i) * 2 + 1] = ssq[i];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int c = tid >> 1, which = tid & 1;
            float t = 0.0f;
            for (int g = 0; g < RPP; ++g) t += red[(g * BN + c) * 2 + which];
            a.stats[((size_t)b * a.Cout + n0 + c) * 2 + which] = t;
        }
    }
}

template <typename T, bool FUSE> static int launch_img8(const Img8Args& a, hipStream_t s) {
    static int lds_dev = -1;
    MAD_REQUIRE(raise_lds_limit(reinterpret_cast<const void*>(&conv3x3_img8_kernel<T, FUSE>), i8::LDS_BYTES, &lds_dev),
                "mad_conv3x3_img8: cannot raise the dynamic LDS limit to %d", i8::LDS_BYTES);
    hipLaunchKernelGGL((conv3x3_img8_kernel<T, FUSE>), dim3(a.B * (a.Cout / i8::BN)), dim3(i8::NTHREADS), i8::LDS_BYTES, s, a);
    return check_launch("mad_conv3x3_img8");
}

}  // namespace mad

using namespace mad;

extern "C" int mad_conv3x3_img8(const void* x1, int C1, const void* x2, int C2, const float* coef, int act,
                                const void* w, const float* bias, const void* residual, int res_mode, void* y,
                                float* stats, int dtype, int B, int Cout, void* stream) {
    MAD_REQUIRE(x1 && w && y, "mad_conv3x3_img8: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_conv3x3_img8: bad dtype %d", dtype);
    const int CK = dtype == MAD_BF16 ? 64 : 32;
    MAD_REQUIRE(C1 > 0 && C1 % CK == 0 && C2 >= 0 && C2 % CK == 0 && (C2 == 0 || x2),
                "mad_conv3x3_img8: C1=%d, C2=%d must be multiples of %d", C1, C2, CK);
    MAD_REQUIRE(Cout > 0 && Cout % 64 == 0, "mad_conv3x3_img8: Cout=%d must be a multiple of 64", Cout);
    MAD_REQUIRE(B > 0, "mad_conv3x3_img8: bad batch %d", B);
    MAD_REQUIRE(res_mode == MAD_RES_NONE || ((res_mode == MAD_RES_SAME || res_mode == MAD_RES_DOWN) && residual),
                "mad_conv3x3_img8: residual mode %d not supported (none / same / down)", res_mode);
    Img8Args a;
    a.x1 = x1; a.x2 = C2 ? x2 : x1; a.w = w; a.bias = bias; a.coef = coef; a.residual = residual; a.y = y;
    a.stats = stats; a.C1 = C1; a.C2 = C2; a.B = B; a.Cout = Cout; a.act = act; a.res_mode = res_mode;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MAD_BF16) return coef ? launch_img8<bf16_t, true>(a, s) : launch_img8<bf16_t, false>(a, s);
    return coef ? launch_img8<float, true>(a, s) : launch_img8<float, false>(a, s);
}
