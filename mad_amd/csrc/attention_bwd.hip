// Backward of the UNet self-attention (training step), bf16, head_dim 64, flash style: the L x L
// probability matrix is recomputed tile by tile from Q, K and the forward's per-query log-sum-exp.
//
//   P = exp(Q.K^T/8 - lse)      dV = P^T.dO      dP = dO.V^T      dS = P o (dP - D),  D[q] = dO[q].O[q]
//   dQ = dS.K/8                 dK = dS^T.Q/8
//
// Two kernels with the forward's shape (a wave owns 32 rows of one side and walks 32-row tiles of the
// other side through LDS):
//   attn_bwd_dq : wave = 32 queries, tiles of K/V.  Scores transposed (keys on accumulator rows, queries
//                 on lanes), so lse and D are lane-local and dS^T is already the B operand of
//                 dQ^T = K^T . dS^T (K^T fetched with transposed LDS reads).  Also writes D.
//   attn_bwd_dkv: wave = 32 keys, tiles of Q/dO.  Scores untransposed (queries on accumulator rows, keys
//                 on lanes): P and dS are the B operands of dV^T = dO^T . P and dK^T = Q^T . dS.
#include "mma.h"

namespace mad {

namespace ab {
constexpr int HD = 64, TR = 32;            // head dim, rows per tile
constexpr int ROW = 128;                   // bytes per row, swizzled copy (chunk ^ ((row>>1)&7))
constexpr int TROW = 192;                  // row pitch of the copy read with transposed loads
}  // namespace ab

typedef __bf16 abf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) abf16x4 lds_abf16x4;

// A fragment of M^T (rows = d of half dh, reduction = the 16 tile rows of step st) from the 192-pitch copy
__device__ __forceinline__ bf16x8 tr_rows(const unsigned char* t_lds, int st, int dh, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int dcol = dh * 32 + (g & 1) * 16 + 4 * pp;
    const int r0 = 16 * st + 4 * (g >> 1) + q;
    const abf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_abf16x4*)(t_lds + r0 * ab::TROW + dcol * 2));
    const abf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_abf16x4*)(t_lds + (r0 + 8) * ab::TROW + dcol * 2));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// accumulator registers 8st..8st+7 as the bf16 B fragment of reduction step st
__device__ __forceinline__ bf16x8 acc_as_b(const f32x16& v, int st) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[8 * st + j];
    return r;
}

__device__ __forceinline__ uint4 scale_chunk(uint4 raw, float s) {
    Vec16<bf16_t> v; v.raw = raw;
    float f[8]; v.unpack(f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] *= s;
    v.pack(f);
    return v.raw;
}

// store the accumulators of M^T[d][row] (two d halves) as rows of M: row = lane & 31, 4 consecutive d per group
__device__ __forceinline__ void store_rows(bf16_t* rowp, const f32x16 (&m)[2], float scale, int fh) {
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ushort4 pk;
            pk.x = f32_to_bf16(m[dh][4 * g + 0] * scale); pk.y = f32_to_bf16(m[dh][4 * g + 1] * scale);
            pk.z = f32_to_bf16(m[dh][4 * g + 2] * scale); pk.w = f32_to_bf16(m[dh][4 * g + 3] * scale);
            *reinterpret_cast<ushort4*>(rowp + dh * 32 + 8 * g + 4 * fh) = pk;
        }
}

// ------------------------------------------------------------------------------------------- dQ (and D)
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                               const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                               float* __restrict__ dsum, bf16_t* __restrict__ dqkv, int L,
                                                               int heads) {
    using namespace ab;
    constexpr int NT = NW * 64, PER = (TR * 8 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char k_lds[TR * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char v_lds[TR * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char kt_lds[TR * TROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, head = blockIdx.y;
    const int C3 = heads * 3 * HD, C = heads * HD;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const int qi = lane & 31, fh = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * L * C3 + head * 3 * HD;

    uint4 qf[4], dof[4];
    float dpart = 0.0f;
    {
        const bf16_t* qrow = base + (size_t)(q0 + qi) * C3;
        const bf16_t* orow = out + ((size_t)b * L + q0 + qi) * C + head * HD;
        const bf16_t* drow = dout + ((size_t)b * L + q0 + qi) * C + head * HD;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[kk] = scale_chunk(*reinterpret_cast<const uint4*>(qrow + (2 * kk + fh) * 8), 0.125f);
            dof[kk] = *reinterpret_cast<const uint4*>(drow + (2 * kk + fh) * 8);
            Vec16<bf16_t> o, d; o.raw = *reinterpret_cast<const uint4*>(orow + (2 * kk + fh) * 8); d.raw = dof[kk];
            float fo[8], fd[8]; o.unpack(fo); d.unpack(fd);
#pragma unroll
            for (int i = 0; i < 8; ++i) dpart = fmaf(fo[i], fd[i], dpart);
        }
    }
    const float Dq = dpart + __shfl_xor(dpart, 32);
    const size_t srow = ((size_t)b * heads + head) * L + q0 + qi;
    if (fh == 0) dsum[srow] = Dq;
    const float lse_q = lse[srow];

    f32x16 dq[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dq[0][e] = 0.0f; dq[1][e] = 0.0f; }

    uint4 kreg[PER], vreg[PER];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * NT;
            if (c < TR * 8) {
                const bf16_t* krow = base + (size_t)(k0 + (c >> 3)) * C3 + HD;
                kreg[i] = *reinterpret_cast<const uint4*>(krow + (c & 7) * 8);
                vreg[i] = *reinterpret_cast<const uint4*>(krow + HD + (c & 7) * 8);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < L; k0 += TR) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * NT;
            if (c < TR * 8) {
                const int r = c >> 3, ch = c & 7, sw = ch ^ ((r >> 1) & 7);
                *reinterpret_cast<uint4*>(k_lds + r * ROW + sw * 16) = kreg[i];
                *reinterpret_cast<uint4*>(v_lds + r * ROW + sw * 16) = vreg[i];
                *reinterpret_cast<uint4*>(kt_lds + r * TROW + ch * 16) = kreg[i];
            }
        }
        __syncthreads();
        if (k0 + TR < L) fetch(k0 + TR);

        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.0f; dp[e] = 0.0f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int sw = ((2 * kk + fh) ^ ((qi >> 1) & 7)) * 16;
            const uint4 kf = *reinterpret_cast<const uint4*>(k_lds + qi * ROW + sw);
            const uint4 vf = *reinterpret_cast<const uint4*>(v_lds + qi * ROW + sw);
            mma_chunk<bf16_t>(s, kf, qf[kk]);             // S^T[key][query]
            mma_chunk<bf16_t>(dp, vf, dof[kk]);           // dP^T[key][query]
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = __expf(s[e] - lse_q) * (dp[e] - Dq);     // dS^T
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 dsf = acc_as_b(s, st);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
                dq[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_rows(kt_lds, st, dh, lane), dsf, dq[dh], 0, 0, 0);
        }
    }
    store_rows(dqkv + ((size_t)b * L + q0 + qi) * C3 + head * 3 * HD, dq, 0.125f, fh);
}

// ------------------------------------------------------------------------------------------- dK, dV
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                bf16_t* __restrict__ dqkv, int L, int heads) {
    using namespace ab;
    constexpr int NT = NW * 64, PER = (TR * 8 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char q_lds[TR * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char do_lds[TR * ROW];
    __shared__ __attribute__((aligned(16))) unsigned char qt_lds[TR * TROW];
    __shared__ __attribute__((aligned(16))) unsigned char dot_lds[TR * TROW];
    __shared__ float lse_lds[TR], d_lds[TR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, head = blockIdx.y;
    const int C3 = heads * 3 * HD, C = heads * HD;
    const int key0 = (blockIdx.x * NW + wave) * 32;
    const int ki = lane & 31, fh = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * L * C3 + head * 3 * HD;
    const bf16_t* dbase = dout + (size_t)b * L * C + head * HD;
    const float* lbase = lse + ((size_t)b * heads + head) * L;
    const float* sbase = dsum + ((size_t)b * heads + head) * L;

    uint4 kf[4], vf[4];
    {
        const bf16_t* krow = base + (size_t)(key0 + ki) * C3 + HD;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            kf[kk] = *reinterpret_cast<const uint4*>(krow + (2 * kk + fh) * 8);
            vf[kk] = *reinterpret_cast<const uint4*>(krow + HD + (2 * kk + fh) * 8);
        }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[0][e] = dk[1][e] = 0.0f; dv[0][e] = dv[1][e] = 0.0f; }

    uint4 qreg[PER], dreg[PER];
    float lreg = 0.0f, sreg = 0.0f;
    auto fetch = [&](int q0) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * NT;
            if (c < TR * 8) {
                qreg[i] = *reinterpret_cast<const uint4*>(base + (size_t)(q0 + (c >> 3)) * C3 + (c & 7) * 8);
                dreg[i] = *reinterpret_cast<const uint4*>(dbase + (size_t)(q0 + (c >> 3)) * C + (c & 7) * 8);
            }
        }
        if (tid < TR) { lreg = lbase[q0 + tid]; sreg = sbase[q0 + tid]; }
    };
    fetch(0);
    for (int q0 = 0; q0 < L; q0 += TR) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * NT;
            if (c < TR * 8) {
                const int r = c >> 3, ch = c & 7, sw = ch ^ ((r >> 1) & 7);
                const uint4 qs = scale_chunk(qreg[i], 0.125f);
                *reinterpret_cast<uint4*>(q_lds + r * ROW + sw * 16) = qs;
                *reinterpret_cast<uint4*>(qt_lds + r * TROW + ch * 16) = qs;
                *reinterpret_cast<uint4*>(do_lds + r * ROW + sw * 16) = dreg[i];
                *reinterpret_cast<uint4*>(dot_lds + r * TROW + ch * 16) = dreg[i];
            }
        }
        if (tid < TR) { lse_lds[tid] = lreg; d_lds[tid] = sreg; }
        __syncthreads();
        if (q0 + TR < L) fetch(q0 + TR);

        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.0f; dp[e] = 0.0f; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int sw = ((2 * kk + fh) ^ ((ki >> 1) & 7)) * 16;
            const uint4 qa = *reinterpret_cast<const uint4*>(q_lds + ki * ROW + sw);
            const uint4 da = *reinterpret_cast<const uint4*>(do_lds + ki * ROW + sw);
            mma_chunk<bf16_t>(s, qa, kf[kk]);             // S[query][key]
            mma_chunk<bf16_t>(dp, da, vf[kk]);            // dP[query][key]
        }
        f32x16 p;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = acc_row(e, lane);
            p[e] = __expf(s[e] - lse_lds[r]);
            s[e] = p[e] * (dp[e] - d_lds[r]);             // dS
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 pf = acc_as_b(p, st), dsf = acc_as_b(s, st);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                dv[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_rows(dot_lds, st, dh, lane), pf, dv[dh], 0, 0, 0);
                dk[dh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_rows(qt_lds, st, dh, lane), dsf, dk[dh], 0, 0, 0);
            }
        }
    }
    bf16_t* orow = dqkv + ((size_t)b * L + key0 + ki) * C3 + head * 3 * HD;
    store_rows(orow + HD, dk, 1.0f, fh);
    store_rows(orow + 2 * HD, dv, 1.0f, fh);
}

// ------------------------------------------------------------------------------------------- fp32 mode
// One wave per query (dQ, D) or per key (dK, dV), lane = head-dim index, dot products by wave reduction.
// The fp32 mode is the tight-tolerance check of the training step, not a throughput path.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(64) void attn_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                              const float* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ dsum, float* __restrict__ dqkv, int L,
                                                              int heads) {
    const int q = blockIdx.x, head = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
    const int C3 = heads * 192, C = heads * 64;
    const float* base = qkv + (size_t)b * L * C3 + head * 192;
    const float qd = base[(size_t)q * C3 + d] * 0.125f;
    const float dod = dout[((size_t)b * L + q) * C + head * 64 + d];
    const float D = wave_sum(dod * out[((size_t)b * L + q) * C + head * 64 + d]);
    const size_t srow = ((size_t)b * heads + head) * L + q;
    if (d == 0) dsum[srow] = D;
    const float l = lse[srow];
    float acc = 0.0f;
    for (int k = 0; k < L; ++k) {
        const float kd = base[(size_t)k * C3 + 64 + d], vd = base[(size_t)k * C3 + 128 + d];
        const float s = wave_sum(qd * kd), dp = wave_sum(dod * vd);
        acc = fmaf(__expf(s - l) * (dp - D), kd, acc);
    }
    dqkv[((size_t)b * L + q) * C3 + head * 192 + d] = acc * 0.125f;
}

__global__ __launch_bounds__(64) void attn_bwd_dkv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ dsum,
                                                               float* __restrict__ dqkv, int L, int heads) {
    const int k = blockIdx.x, head = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
    const int C3 = heads * 192, C = heads * 64;
    const float* base = qkv + (size_t)b * L * C3 + head * 192;
    const float kd = base[(size_t)k * C3 + 64 + d], vd = base[(size_t)k * C3 + 128 + d];
    const float* lrow = lse + ((size_t)b * heads + head) * L;
    const float* srow = dsum + ((size_t)b * heads + head) * L;
    float dk = 0.0f, dv = 0.0f;
    for (int q = 0; q < L; ++q) {
        const float qd = base[(size_t)q * C3 + d] * 0.125f;
        const float dod = dout[((size_t)b * L + q) * C + head * 64 + d];
        const float p = __expf(wave_sum(qd * kd) - lrow[q]);
        const float ds = p * (wave_sum(dod * vd) - srow[q]);
        dv = fmaf(p, dod, dv);
        dk = fmaf(ds, qd, dk);
    }
    float* orow = dqkv + ((size_t)b * L + k) * C3 + head * 192;
    orow[64 + d] = dk;
    orow[128 + d] = dv;
}

}  // namespace mad

using namespace mad;

extern "C" int mad_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum,
                                 void* dqkv, int dtype, int B, int L, int heads, void* stream) {
    MAD_REQUIRE(qkv && out && dout && lse && dsum && dqkv, "mad_attention_bwd: null pointer");
    MAD_REQUIRE(dtype == MAD_BF16 || dtype == MAD_F32, "mad_attention_bwd: bad dtype %d", dtype);
    MAD_REQUIRE(B > 0 && heads > 0 && L >= 64 && L % 64 == 0, "mad_attention_bwd: L=%d must be a multiple of 64", L);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MAD_F32) {
        const dim3 grid(L, heads, B);
        hipLaunchKernelGGL(attn_bwd_dq_f32_kernel, grid, dim3(64), 0, s, (const float*)qkv, (const float*)out,
                           (const float*)dout, lse, dsum, (float*)dqkv, L, heads);
        if (int rc = check_launch("mad_attention_bwd (dq)")) return rc;
        hipLaunchKernelGGL(attn_bwd_dkv_f32_kernel, grid, dim3(64), 0, s, (const float*)qkv, (const float*)dout, lse,
                           (const float*)dsum, (float*)dqkv, L, heads);
        return check_launch("mad_attention_bwd (dkv)");
    }
    const bf16_t* q = (const bf16_t*)qkv; const bf16_t* o = (const bf16_t*)out; const bf16_t* d = (const bf16_t*)dout;
    bf16_t* g = (bf16_t*)dqkv;
    if (L % 128 == 0) {
        const dim3 grid(L / 128, heads, B);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<4>), grid, dim3(256), 0, s, q, o, d, lse, dsum, g, L, heads);
        if (int rc = check_launch("mad_attention_bwd (dq)")) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<4>), grid, dim3(256), 0, s, q, d, lse, (const float*)dsum, g, L, heads);
    } else {
        const dim3 grid(L / 64, heads, B);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<2>), grid, dim3(128), 0, s, q, o, d, lse, dsum, g, L, heads);
        if (int rc = check_launch("mad_attention_bwd (dq)")) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<2>), grid, dim3(128), 0, s, q, d, lse, (const float*)dsum, g, L, heads);
    }
    return check_launch("mad_attention_bwd (dkv)");
}
