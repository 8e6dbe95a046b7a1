// MFMA fragment helpers shared by the convolution and attention kernels (gfx950, wave64).
//
// Operand convention used throughout: every lane reads ONE 16-byte chunk of a row-major row
// (lane&31 selects the row, lane>>5 selects which of two adjacent chunks).  For bf16 that is the
// native A/B layout of v_mfma_f32_32x32x16_bf16 (lane half h holds k = 8h..8h+7).  For fp32 the
// same two chunks (8 floats) are consumed by four v_mfma_f32_32x32x2_f32, element e of each lane
// per instruction; the k order inside the sum is permuted identically for A and B, which leaves
// the product unchanged.  One `mma_chunk` therefore always advances k by 32 bytes of a row.
#pragma once
#include "common.h"

namespace mad {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> __device__ __forceinline__ void mma_chunk(f32x16& acc, const uint4& a, const uint4& b);

template <> __device__ __forceinline__ void mma_chunk<bf16_t>(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                   0, 0, 0);
}

template <> __device__ __forceinline__ void mma_chunk<float>(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// The 16x16 forms: every lane reads ONE 16-byte chunk of a 64-byte row, lane&15 selects the row and lane>>4
// which of its four chunks.  bf16: v_mfma_f32_16x16x32_bf16 consumes the whole 64-byte row (k = 8*(lane>>4)..+7 per
// lane); fp32: four v_mfma_f32_16x16x4_f32, element e of each lane per instruction (k order permuted identically
// for both operands).  Accumulator register j of a lane is row 4*(lane>>4) + j of the FIRST operand, column
// lane&15 of the second.
template <typename T> __device__ __forceinline__ void mma16_row(f32x4& acc, const uint4& a, const uint4& b);

template <> __device__ __forceinline__ void mma16_row<bf16_t>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                   0, 0, 0);
}

template <> __device__ __forceinline__ void mma16_row<float>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// Row of the 32x32 accumulator held in register j of this lane (column = lane & 31).
__device__ __forceinline__ int acc_row(int j, int lane) { return (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5); }

// Asynchronous 16-byte global -> LDS copy (LDS-DMA).  The LDS destination of a wave-instruction
// is the wave-uniform byte address `lds_wave_base` + lane*16; the global source is per lane.
// Issued from inline asm so the compiler does not serialise it against the LDS reads of the
// tile being multiplied: the caller waits with dma_wait_all() before the barrier that publishes
// the tile.  M0 carries the LDS base and is left holding it (no save / restore: two scalar moves per copy less);
// the statement declares M0 clobbered, so the compiler's own uses of M0 (indexed register moves, GWS, messages,
// which it initialises right in front of each use) are never assumed to survive a copy.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_wave_base) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_wave_base)
        : "memory", "m0");
}

// The same copy with a wave-uniform 64-bit base address (held in scalar registers) and a 32-bit byte offset
// per lane: half the address registers and no 64-bit vector arithmetic per copy.
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_wave_base)
        : "memory", "m0");
}

__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Wait until at most N of this wave's vector-memory operations (LDS-DMA included) are outstanding;
// they retire in issue order, so everything older than the N youngest has landed.
template <int N> __device__ __forceinline__ void dma_wait_upto() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Source of LDS-DMA copies that must deliver zeros (conv padding, rows past the end).  16 KiB: a copy may read it
// at the channel offset of the row it stands in for (rows of up to MAD_ZERO_PAGE_BYTES - 16 bytes).
constexpr int MAD_ZERO_PAGE_BYTES = 16384;
__device__ __attribute__((aligned(256))) const unsigned char g_zero_page[MAD_ZERO_PAGE_BYTES] = {0};

// Byte address inside the workgroup's LDS allocation of a pointer into __shared__ memory.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

}  // namespace mad
