// Shared helpers for the gfx950 kernels of libmad_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mad_hip.h"

namespace mad {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// tap-major form of the last layer (conv_out.hip): bf16 activations, at most 3 output channels
bool conv_out_taps_ok(int Cin, int H, int W, int Cout);
int conv_out_taps(const void* x, int Cin, const float* coef, int act, const void* w, const float* bias, float* y, int B,
                  int H, int W, int Cout, hipStream_t s);

// 1x1 convolution as a GEMM on 256-pixel tiles (conv1x1.hip): bf16, M = B*H*W pixels, optional same-size residual
// and per-64-pixel-run statistics
bool conv1x1_ok(int M, int C1, int C2, int Cout);
int conv1x1(const void* x1, int C1, const void* x2, int C2, const void* w, const float* bias, const void* residual,
            void* y, float* stats, int M, int Cout, hipStream_t s);

#define MAD_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            mad::set_error(__VA_ARGS__);    \
            return 1;                       \
        }                                   \
    } while (0)

typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round-to-nearest-even conversion; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps
// NaNs as NaNs, which integer bit tricks do not.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;  // elements per 16 bytes
    __device__ static __forceinline__ float load(bf16_t v) { return bf16_to_f32(v); }
    __device__ static __forceinline__ bf16_t store(float f) { return f32_to_bf16(f); }
};
template <> struct Elem<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ float load(float v) { return v; }
    __device__ static __forceinline__ float store(float f) { return f; }
};

// 16-byte vector of T, unpacked to floats.
template <typename T> struct Vec16 {
    static constexpr int N = Elem<T>::VEC;
    uint4 raw;
    __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void store(T* p) const { *reinterpret_cast<uint4*>(p) = raw; }
    // streaming store: the line is not expected to be re-read by this kernel
    __device__ __forceinline__ void store_nt(T* p) const {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v = {raw.x, raw.y, raw.z, raw.w};
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    }
    __device__ __forceinline__ void unpack(float* f) const {
        const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int i = 0; i < N; ++i) f[i] = Elem<T>::load(e[i]);
    }
    __device__ __forceinline__ void pack(const float* f) {
        T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = Elem<T>::store(f[i]);
    }
};

// x * sigmoid(x); v_exp_f32 and v_rcp_f32 (1 ulp each)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Raise a kernel's dynamic-LDS limit.  The attribute belongs to the current device's copy of the kernel, so
// it is set again whenever the calling thread has moved to another device (`done_for` holds the last one).
inline bool raise_lds_limit(const void* kernel, int bytes, int* done_for) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (*done_for == dev) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    *done_for = dev;
    return true;
}

// d/du of u*sigmoid(u)
__device__ __forceinline__ float silu_grad_f(float u) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    return s * fmaf(u, 1.0f - s, 1.0f);
}

// Dropout keep-mask from a counter hash: element e of the tensor dropped with `seed` is kept with probability
// 1-p and scaled by 1/(1-p).  The same (seed, e) gives the same decision in the forward and in the backward.
// Decisions are drawn for groups of 8 consecutive elements: a two-round hash of the group index gives the
// first pair (16 bits each against a 16-bit threshold), one more round each further pair -- 5 rounds per 8
// elements, since the integer multiplies of a round cost more than the activation they sit beside.
__host__ __device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct DropMask {
    unsigned s0, s1, thr;      // thr (of 65536) == 0: no dropout
    float scale;
};
inline DropMask make_drop_mask(unsigned long long seed, float p) {
    DropMask m;
    m.s0 = mix32((unsigned)seed); m.s1 = mix32((unsigned)(seed >> 32) ^ 0x9e3779b9u);
    m.thr = p > 0.0f ? (unsigned)((double)p * 65536.0 + 0.5) : 0u;
    m.scale = m.thr ? 65536.0f / (float)(65536u - m.thr) : 1.0f;
    return m;
}
// keep factors of the N (4 or 8) elements e0 .. e0+N-1, e0 a multiple of N
template <int N> __device__ __forceinline__ void drop_factors(const DropMask& m, unsigned long long e0, float (&f)[N]) {
    static_assert(N == 4 || N == 8, "whole or half groups");
    const unsigned long long grp = e0 >> 3;
    const unsigned h0 = mix32(mix32((unsigned)grp ^ m.s0) + (unsigned)(grp >> 32) + m.s1);
    const unsigned first = N == 8 ? 0u : ((unsigned)e0 & 4u) >> 1;       // pair index of e0 inside its group
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        const unsigned w = first + (i >> 1);
        const unsigned hw = mix32(h0 + w * 0x9e3779b9u);
        const unsigned h = w == 0 ? h0 : hw;
        f[i] = (h & 0xffffu) >= m.thr ? m.scale : 0.0f;
        f[i + 1] = (h >> 16) >= m.thr ? m.scale : 0.0f;
    }
}

}  // namespace mad
