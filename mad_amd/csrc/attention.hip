// Self-attention of the UNet blocks, flash style (no L x L matrix in memory), head_dim 64.
//
// One wave owns 32 queries; a workgroup of NW waves shares each 32-key K/V tile through LDS.
// Scores are computed transposed, S^T = K . Q^T (keys on the accumulator rows, queries on the
// lanes), so a query's softmax statistics are lane-local (16 registers + one cross-half
// exchange) and the probability tile is already laid out as the B operand of the second product
// O^T = V^T . P^T -- no shuffle or LDS round trip for P.  fp32 softmax, fp32 accumulation.
#include "mma.h"
#include <cstdlib>

namespace mad {

constexpr int HD = 64;   // head dim
constexpr int KT = 32;   // keys per tile

template <typename T> struct AttnLds;
// bf16: K rows are 128 bytes (8 chunks), V rows 128 bytes, unswizzled V for column gathers.
template <> struct AttnLds<bf16_t> { static constexpr int ROW = 128; };
template <> struct AttnLds<float> { static constexpr int ROW = 256; };

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                             float* __restrict__ lse, int L, int heads) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int ROW = AttnLds<T>::ROW;          // bytes per key row in LDS
    constexpr int CHUNKS = ROW / 16;              // 16-byte chunks per row
    constexpr int NT = NW * 64;
    // bf16 V rows sit at a 192-byte pitch: the four key rows of a transposed 4x16 block read then fall
    // on disjoint bank ranges (row pitch 48 dwords -> banks 0, 48, 32, 16)
    constexpr int VROW = sizeof(T) == 2 ? 192 : ROW;
    __shared__ __attribute__((aligned(16))) unsigned char k_lds[KT * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char v_lds[KT * VROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, head = blockIdx.y;
    const int C3 = heads * 3 * HD;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int qi = lane & 31, fh = lane >> 5;
    const T* base = qkv + (size_t)b * L * C3 + head * 3 * HD;

    // Q fragments (B operand): chunk pair kk covers d = kk*2*VEC .. ; scale by 1/sqrt(64) (exact)
    constexpr int NKK = CHUNKS / 2;
    uint4 qf[NKK];
    {
        const T* qrow = base + (size_t)(q0 + qi) * C3;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            Vec16<T> v;
            v.load(qrow + (2 * kk + fh) * VEC);
            float f[VEC];
            v.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] *= 0.125f;
            v.pack(f);
            qf[kk] = v.raw;
        }
    }

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.0f; o[1][e] = 0.0f; }
    float m_run = -1.0e30f, l_run = 0.0f;

    // K/V tile staging: global -> registers one tile ahead of the LDS copy, so the global latency of
    // tile t+1 is covered by the products of tile t
    constexpr int PER_THREAD = (KT * CHUNKS + NT - 1) / NT;
    uint4 kreg[PER_THREAD], vreg[PER_THREAD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NT;
            if (c < KT * CHUNKS) {
                const int r = c / CHUNKS, ch = c % CHUNKS;
                const T* krow = base + (size_t)(k0 + r) * C3 + HD;
                kreg[i] = *reinterpret_cast<const uint4*>(krow + ch * VEC);
                vreg[i] = *reinterpret_cast<const uint4*>(krow + HD + ch * VEC);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < L; k0 += KT) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NT;
            if (c < KT * CHUNKS) {
                const int r = c / CHUNKS, ch = c % CHUNKS;
                const int sw = (ROW == 128) ? (ch ^ ((r >> 1) & 7)) : (ch ^ (r & 15));
                *reinterpret_cast<uint4*>(k_lds + r * ROW + sw * 16) = kreg[i];
                *reinterpret_cast<uint4*>(v_lds + r * VROW + ch * 16) = vreg[i];
            }
        }
        __syncthreads();
        if (k0 + KT < L) fetch(k0 + KT);

        // S^T[key][query] = sum_d K[key][d] * Q[query][d]
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int ch = 2 * kk + fh;
            const int sw = (ROW == 128) ? (ch ^ ((qi >> 1) & 7)) : (ch ^ (qi & 15));
            const uint4 kf = *reinterpret_cast<const uint4*>(k_lds + qi * ROW + sw * 16);
            mma_chunk<T>(s, kf, qf[kk]);
        }

        // online softmax for this lane's query: 16 keys here, the other 16 in lane^32
        float tmax = s[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, s[e]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = __expf(s[e] - m_new); psum += s[e]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }

        // O^T[d][query] += sum_key V[key][d] * P^T[key][query]
        if constexpr (sizeof(T) == 2) {
            // k-step st uses accumulator registers 8st..8st+7 as the B fragment; element j of lane
            // half h is key 16st + 8(j>>2) + 4h + (j&3), so the A fragment gathers V at those keys.
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s[8 * st + j];
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    // A fragment = V^T[d][key]: two hardware-transposed reads of a 4-key x 16-d block each.
                    // In a 16-lane group lane 4q+p supplies the address of key row q, d columns 4p..4p+3,
                    // and lane i receives column i of the four rows.
                    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
                    const int dcol = dh * 32 + (g & 1) * 16 + 4 * pp;
                    const int krow0 = 16 * st + 4 * (g >> 1) + q;
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4*)(v_lds + krow0 * VROW + dcol * 2));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (lds_bf16x4*)(v_lds + (krow0 + 8) * VROW + dcol * 2));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dh], 0, 0, 0);
                }
            }
        } else {
            // fp32: one 32x32x2 MFMA per accumulator register; lane half h supplies key acc_row(j)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int key = (j & 3) + 8 * (j >> 2) + 4 * fh;
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    const float vv = *reinterpret_cast<const float*>(v_lds + key * ROW + (dh * 32 + qi) * 4);
                    o[dh] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[j], o[dh], 0, 0, 0);
                }
            }
        }
    }

    const float l = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l;
    const int C = heads * HD;
    // log-sum-exp of the query's scaled scores: what the backward needs to rebuild the probabilities
    if (lse && fh == 0) lse[((size_t)b * heads + head) * L + q0 + qi] = m_run + __logf(l);
    T* orow = out + ((size_t)b * L + q0 + qi) * C + head * HD;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = dh * 32 + 8 * g + 4 * fh;
            if constexpr (sizeof(T) == 2) {
                ushort4 pk;
                pk.x = f32_to_bf16(o[dh][4 * g + 0] * inv);
                pk.y = f32_to_bf16(o[dh][4 * g + 1] * inv);
                pk.z = f32_to_bf16(o[dh][4 * g + 2] * inv);
                pk.w = f32_to_bf16(o[dh][4 * g + 3] * inv);
                *reinterpret_cast<ushort4*>(orow + d) = pk;
            } else {
                float4 pk = make_float4(o[dh][4 * g + 0] * inv, o[dh][4 * g + 1] * inv, o[dh][4 * g + 2] * inv,
                                        o[dh][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(orow + d) = pk;
            }
        }
}

// ------------------------------------------------------------------------------- bf16 form
// The same algorithm arranged for the vector-issue budget of the bf16 case, where the softmax (not the
// matrix pipe) sets the pace at head_dim 64: per score element one multiply-add, one exp2, one add and
// half a conversion are all that is left in the tile loop.
//   * 64-key tiles, three K/V buffers in LDS filled by LDS-DMA two tiles ahead (no staging registers, no LDS
//     stores from the waves), counted waits, ONE barrier per tile.  A copy instruction writes 64 x 16 bytes
//     lane-linearly, i.e. 8 whole 128-byte rows, so the bank-conflict-free layouts are produced on the SOURCE side:
//     the lane that fills physical 16-byte piece p of row r fetches logical piece p ^ ((r>>1)&7) of a K row
//     (every ds_read_b128 of a fragment conflict-free) and p ^ 4*((r>>1)&1) of a V row (the 64-byte halves of
//     row pairs swapped: the four rows a 32-lane group of ds_read_b64_tr_b16 touches fall on distinct banks).
//   * exp2 domain: p = exp2(s*log2(e) - m2), the scale and the running maximum folded into one fma.
//   * The running maximum moves only when a tile's maximum exceeds it by more than RESCALE_THR (a
//     wave-uniform decision): p is then bounded by 2^RESCALE_THR instead of 1, which fp32 sums and the
//     bf16 probabilities (same relative precision at any scale) absorb, and the O-wide rescale runs on a
//     few tiles per query block instead of on all of them.
//   * The tile maximum is a chain of three-input maxima; the halves of a query's keys meet through one
//     lane swap.
constexpr int KT2 = 64;
#ifndef MAD_ATTN_THR
#define MAD_ATTN_THR 0.0f
#endif
constexpr float RESCALE_THR = MAD_ATTN_THR;

__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// max / sum of a value with the one held by lane ^ 32: v_permlane32_swap exchanges the upper half of its first
// operand with the lower half of its second, so with both operands x one result holds x of lanes 0..31 twice and
// the other x of lanes 32..63 twice
__device__ __forceinline__ float halves_max(float x) {
    const unsigned xi = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float halves_sum(float x) {
    const unsigned xi = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// QB: 32-query blocks per wave.  With two, every K fragment and every gathered V^T fragment read from LDS feeds two
// MFMAs, a K/V tile copied into LDS serves twice the queries, and the two blocks' softmax chains are independent
// instruction streams inside one wave (64 + 64 accumulator registers: two waves per SIMD).
template <int NW, int QB>
__global__ __launch_bounds__(NW * 64) void attention_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                  bf16_t* __restrict__ out,
                                                                  float* __restrict__ lse, int L, int heads, int B) {
    constexpr int ROW = 128, VEC = 8;
    constexpr int K_BYTES = KT2 * ROW, TILE_BYTES = 2 * K_BYTES, NBUF = 3;     // a buffer: the K rows, then the V rows
    __shared__ __attribute__((aligned(1024))) unsigned char kv_lds[NBUF * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are handed to the eight XCDs round-robin by their linear index, and every query block of a
    // (sample, head) reads that pair's whole K and V: the one-dimensional grid is mapped so that all query blocks of
    // a pair run on ONE XCD, whose L2 then holds its K and V once (with the pair as the slow grid dimension eight
    // XCDs each fetched them from memory).
    const int nq = L / (NW * 32 * QB), nbh = heads * B;
    int bh, qb;
    if ((nbh & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / nq) * 8 + xcd; qb = slot % nq;
    } else {
        bh = blockIdx.x / nq; qb = blockIdx.x % nq;
    }
    const int b = bh / heads, head = bh % heads;
    const int C3 = heads * 3 * HD;
    const int q0 = (qb * NW + wave) * 32 * QB;            // this wave's first query; block u holds q0 + 32u ..
    const int qi = lane & 31, fh = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * L * C3 + head * 3 * HD;

    // Q fragments (B operand), scaled by 1/sqrt(64): exact in bf16
    uint4 qf[QB][4];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const bf16_t* qrow = base + (size_t)(q0 + 32 * u + qi) * C3;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            Vec16<bf16_t> v;
            v.load(qrow + (2 * kk + fh) * VEC);
            float f[VEC];
            v.unpack(f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) f[i] *= 0.125f;
            v.pack(f);
            qf[u][kk] = v.raw;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the Q loads: the counted waits below see only the copies

    f32x16 o[QB][2];
    float m2[QB], l_run[QB];                    // running maximum in exp2 units, running sum
#pragma unroll
    for (int u = 0; u < QB; ++u) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[u][0][e] = 0.0f; o[u][1][e] = 0.0f; }
        m2[u] = -1.0e30f; l_run[u] = 0.0f;
    }
    constexpr float LOG2E = 1.44269504088896340736f;

    // A tile is 64 keys x (128 bytes of K + 128 bytes of V) = 16 copy instructions of 8 rows each, PER_WAVE per wave:
    // instruction i of the K (V) half covers rows 8i .. 8i+7.  Per-lane source offsets from the tile's first key:
    constexpr int PER_HALF = 8 / NW, PER_WAVE = 2 * PER_HALF;
    static_assert(8 % NW == 0, "whole copy instructions per wave");
    unsigned koff[PER_HALF], voff[PER_HALF];
#pragma unroll
    for (int j = 0; j < PER_HALF; ++j) {
        const int r = 8 * (wave + NW * j) + (lane >> 3), pc = lane & 7;
        koff[j] = (unsigned)((r * C3 + HD + (pc ^ ((r >> 1) & 7)) * VEC) * 2);
        voff[j] = (unsigned)((r * C3 + 2 * HD + (pc ^ (((r >> 1) & 1) << 2)) * VEC) * 2);
    }
    const unsigned lds0 = lds_addr(kv_lds);
    auto issue = [&](int tile, int buf) {
        const bf16_t* src = base + (size_t)tile * KT2 * C3;          // wave-uniform
#pragma unroll
        for (int j = 0; j < PER_HALF; ++j) {
            const int i = wave + NW * j;
            glds16s(src, koff[j], __builtin_amdgcn_readfirstlane(lds0 + buf * TILE_BYTES + i * 1024));
            glds16s(src, voff[j], __builtin_amdgcn_readfirstlane(lds0 + buf * TILE_BYTES + K_BYTES + i * 1024));
        }
    };
    const int ntiles = L / KT2;
    issue(0, 0);
    issue(ntiles > 1 ? 1 : 0, 1);

    // this lane's V^T gather: rows 4*(g>>1) + q (+ 8) of a 16-key step, 8 bytes at d = 32*dh + 16*(g&1) + 4*pp, the
    // 64-byte half picked against the row pair's swap
    int vlane[2];
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int sw = (q >> 1) & 1;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
            vlane[dh] = (4 * (g >> 1) + q) * ROW + ((dh ^ sw) << 6) + (g & 1) * 32 + 8 * pp;
    }

    for (int t = 0; t < ntiles; ++t) {
        // tile t has landed (the PER_WAVE copies of tile t+1 may still be in flight); after the barrier every wave's
        // part of it is visible and every wave is done with tile t-1, whose buffer tile t+2 now overwrites
        dma_wait_upto<PER_WAVE>();
        __syncthreads();
        // (past the end the last tile is fetched again into an idle buffer: every wave issues the same number of
        // copies per tile, which the counted wait relies on)
        issue(t + 2 < ntiles ? t + 2 : ntiles - 1, (t + 2) % NBUF);
        const unsigned char* kb = kv_lds + (t % NBUF) * TILE_BYTES;
        const unsigned char* vb = kb + K_BYTES;

        // S^T[key][query] for the two 32-key blocks of the tile: one K fragment per (block, k-step), QB products
        f32x16 s[QB][2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[u][blk][e] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ch = 2 * kk + fh;
                const int r = blk * 32 + qi;
                const uint4 kf = *reinterpret_cast<const uint4*>(kb + r * ROW + ((ch ^ ((r >> 1) & 7)) << 4));
#pragma unroll
                for (int u = 0; u < QB; ++u) mma_chunk<bf16_t>(s[u][blk], kf, qf[u][kk]);
            }
        }

#pragma unroll
        for (int u = 0; u < QB; ++u) {
            // this lane's query: 32 of the tile's keys here, the other 32 in lane^32
            // (four independent chains of three-input maxima, then their maximum: a third of one chain's depth; the
            // two halves of the query's keys meet through v_permlane32_swap, one vector instruction, no LDS trip)
            float c4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int blk = c >> 1, e0 = 8 * (c & 1);
                c4[c] = max3f(s[u][blk][e0], s[u][blk][e0 + 1], s[u][blk][e0 + 2]);
                c4[c] = max3f(c4[c], s[u][blk][e0 + 3], s[u][blk][e0 + 4]);
                c4[c] = max3f(c4[c], s[u][blk][e0 + 5], s[u][blk][e0 + 6]);
                c4[c] = fmaxf(c4[c], s[u][blk][e0 + 7]);
            }
            float tmax = fmaxf(max3f(c4[0], c4[1], c4[2]), c4[3]);
            tmax = halves_max(tmax);
            const float t2 = tmax * LOG2E;
            if (__builtin_amdgcn_ballot_w64(t2 > m2[u] + RESCALE_THR) != 0) {      // wave-uniform
                const float m_new = fmaxf(m2[u], t2);
                const float alpha = __builtin_amdgcn_exp2f(m2[u] - m_new);
                l_run[u] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) { o[u][0][e] *= alpha; o[u][1][e] *= alpha; }
                m2[u] = m_new;
            }
            float psum0 = 0.0f, psum1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[u][0][e] = __builtin_amdgcn_exp2f(fmaf(s[u][0][e], LOG2E, -m2[u]));
                s[u][1][e] = __builtin_amdgcn_exp2f(fmaf(s[u][1][e], LOG2E, -m2[u]));
                psum0 += s[u][0][e];
                psum1 += s[u][1][e];
            }
            l_run[u] += psum0 + psum1;
        }

        // O^T[d][query] += sum_key V[key][d] * P^T[key][query]; k-step st of block blk uses accumulator
        // registers 8st..8st+7 as the B fragment: element j of lane half h is key 32blk + 16st + 8(j>>2) + 4h
        // + (j&3), and the A fragment gathers V^T at those keys with two transposed 4-key x 16-d reads (one gather,
        // QB products)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf[QB];
#pragma unroll
                for (int u = 0; u < QB; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[u][j] = (__bf16)s[u][blk][8 * st + j];
#pragma unroll
                for (int dh = 0; dh < 2; ++dh) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
                    const unsigned char* vp = vb + (32 * blk + 16 * st) * ROW + vlane[dh];
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)vp);
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(vp + 8 * ROW));
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int u = 0; u < QB; ++u)
                        o[u][dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[u], o[u][dh], 0, 0, 0);
                }
            }
    }
    dma_wait_all();          // the copies past the end must not land in LDS that already belongs to another workgroup

    const int C = heads * HD;
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const float l = halves_sum(l_run[u]);
        const float inv = 1.0f / l;
        const int qrow = q0 + 32 * u + qi;
        // natural-log sum-exp of the query's scaled scores (what the backward rebuilds the probabilities from)
        if (lse && fh == 0) lse[((size_t)b * heads + head) * L + qrow] = m2[u] * 0.69314718055994530942f + __logf(l);
        bf16_t* orow = out + ((size_t)b * L + qrow) * C + head * HD;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dh * 32 + 8 * g + 4 * fh;
                ushort4 pk;
                pk.x = f32_to_bf16(o[u][dh][4 * g + 0] * inv);
                pk.y = f32_to_bf16(o[u][dh][4 * g + 1] * inv);
                pk.z = f32_to_bf16(o[u][dh][4 * g + 2] * inv);
                pk.w = f32_to_bf16(o[u][dh][4 * g + 3] * inv);
                *reinterpret_cast<ushort4*>(orow + d) = pk;
            }
    }
}

}  // namespace mad

using namespace mad;

static int attention_impl(const void* qkv, void* out, float* lse, int dtype, int B, int L, int heads, void* stream,
                          const char* who) {
    MAD_REQUIRE(qkv && out, "%s: null pointer", who);
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "%s: bad dtype %d", who, dtype);
    MAD_REQUIRE(B > 0 && heads > 0 && L >= 64 && L % 64 == 0, "%s: L=%d must be a multiple of 64", who, L);
    hipStream_t s = (hipStream_t)stream;
    static const bool old_form = getenv("MAD_ATTN_OLD") != nullptr;      // tuning aid: the generic kernel for bf16 too
    if (dtype == MAD_BF16 && !old_form) {
        // two query blocks per wave (256 queries per workgroup) where that still gives every CU its two workgroups
        // (measured: a gain at L = 1024, a loss at L = 256, where a (sample, head) is then a single workgroup)
        const char* qb_env = getenv("MAD_ATTN_QB");                                    // tuning / test aid: 1 / 2 forced
        const int qb_sel = qb_env ? atoi(qb_env) : 0;
        const bool two = qb_sel ? qb_sel == 2 : (L >= 512 && (size_t)(L / 256) * heads * B >= 512);
        if (L % 256 == 0 && two)
            hipLaunchKernelGGL((attention_bf16_kernel<4, 2>), dim3((L / 256) * heads * B), dim3(256), 0, s, (const bf16_t*)qkv,
                               (bf16_t*)out, lse, L, heads, B);
        else if (L % 128 == 0)
            hipLaunchKernelGGL((attention_bf16_kernel<4, 1>), dim3((L / 128) * heads * B), dim3(256), 0, s, (const bf16_t*)qkv,
                               (bf16_t*)out, lse, L, heads, B);
        else
            hipLaunchKernelGGL((attention_bf16_kernel<2, 1>), dim3((L / 64) * heads * B), dim3(128), 0, s, (const bf16_t*)qkv,
                               (bf16_t*)out, lse, L, heads, B);
        return check_launch(who);
    }
    if (L % 128 == 0) {
        dim3 grid(L / 128, heads, B);
        if (dtype == MAD_BF16)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 4>), grid, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, heads);
        else
            hipLaunchKernelGGL((attention_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)qkv, (float*)out, lse, L, heads);
    } else {
        dim3 grid(L / 64, heads, B);
        if (dtype == MAD_BF16)
            hipLaunchKernelGGL((attention_kernel<bf16_t, 2>), grid, dim3(128), 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, L, heads);
        else
            hipLaunchKernelGGL((attention_kernel<float, 2>), grid, dim3(128), 0, s, (const float*)qkv, (float*)out, lse, L, heads);
    }
    return check_launch(who);
}

extern "C" int mad_attention(const void* qkv, void* out, int dtype, int B, int L, int heads, void* stream) {
    return attention_impl(qkv, out, nullptr, dtype, B, L, heads, stream, "mad_attention");
}

extern "C" int mad_attention_lse(const void* qkv, void* out, float* lse, int dtype, int B, int L, int heads,
                                 void* stream) {
    MAD_REQUIRE(lse, "mad_attention_lse: null pointer");
    return attention_impl(qkv, out, lse, dtype, B, L, heads, stream, "mad_attention_lse");
}
