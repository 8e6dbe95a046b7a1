// Cost to the issuing wave of global -> LDS copy instructions: cycles to issue a burst of N pieces (no waits
// inside the burst), then cycles until all have landed; by waves per CU.  Source is L2-resident (256 KiB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

template <int N, int ROWS> __global__ __launch_bounds__(256) void burst(const uint4* src, int reps, unsigned long long* out) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    unsigned long long issue = 0, land = 0;
    // ROWS = 1: each piece is 1 KiB contiguous; ROWS = 8: eight 128-byte rows 9216 bytes apart
    const size_t lane_off = ROWS == 1 ? lane : (size_t)(lane >> 3) * 576 + (lane & 7);
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        const uint4* p = src + ((r * 37 + blockIdx.x * 11 + wave * 3) % 8) * 1024;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < N; ++i)
            glds16(p + i * 64 + lane_off, __builtin_amdgcn_readfirstlane(base + wave * 16384 + (i & 15) * 1024));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        issue += t1 - t0; land += t2 - t0;
    }
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = issue; out[(blockIdx.x * 4 + wave) * 2 + 1] = land; }
}

template <int N, int ROWS> void run(const uint4* src, unsigned long long* out, int blocks) {
    const int reps = 200;
    hipFuncSetAttribute((const void*)burst<N, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int k = 0; k < 2; ++k) { hipLaunchKernelGGL((burst<N, ROWS>), dim3(blocks), dim3(256), 65536, 0, src, reps, out); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), out, blocks * 64, hipMemcpyDeviceToHost);
    double is = 0, la = 0;
    for (int i = 0; i < blocks * 4; ++i) { is += h[2 * i]; la += h[2 * i + 1]; }
    is /= blocks * 4.0 * reps; la /= blocks * 4.0 * reps;
    printf("rows=%d blocks/CU=%d burst of %2d pieces: issue %.0f cyc (%.0f per piece), landed after %.0f cyc\n", ROWS, blocks / 256, N,
           is, is / N, la);
}

int main() {
    uint4* src; unsigned long long* out;
    hipMalloc(&src, 8 << 20); hipMalloc(&out, 1024 * 64);
    hipMemset(src, 1, 8 << 20);
    for (int blocks : {256, 512}) {
        run<1, 1>(src, out, blocks); run<2, 1>(src, out, blocks); run<4, 1>(src, out, blocks); run<8, 1>(src, out, blocks);
        run<16, 1>(src, out, blocks); run<32, 1>(src, out, blocks);
        run<2, 8>(src, out, blocks); run<8, 8>(src, out, blocks);
    }
    return 0;
}
