// Per-CU fill rate of global -> LDS copies (LDS-DMA) and global -> register loads, by working-set size.
// Build: hipcc -O3 --offload-arch=gfx950 -o fill_rate fill_rate.hip ; run: ./fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

// every block walks `iters` tiles of 8 KiB (256 threads x 2 x 16 B); tile index = (block*stride + i) mod ntiles
template <int MODE> __global__ __launch_bounds__(256) void fill(const uint4* src, size_t ntiles, int iters, int same,
                                                                  unsigned* sink, unsigned long long* cyc) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    uint4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    size_t t = same ? 0 : (size_t)blockIdx.x * 977;
    for (int i = 0; i < iters; ++i) {
        const uint4* p = src + (t % ntiles) * 512;
        if (MODE == 0) {
            glds16(p + wave * 64 + lane, __builtin_amdgcn_readfirstlane(base + (i & 7) * 8192 + wave * 1024));
            glds16(p + 256 + wave * 64 + lane, __builtin_amdgcn_readfirstlane(base + (i & 7) * 8192 + 4096 + wave * 1024));
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        } else {
            const uint4 a = p[wave * 64 + lane], b = p[256 + wave * 64 + lane];
            acc.x ^= a.x ^ b.x; acc.y ^= a.y ^ b.y; acc.z ^= a.z ^ b.z; acc.w ^= a.w ^ b.w;
        }
        t += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) acc.x = smem[tid * 4];
    if (acc.x == 0x12345678u && acc.y == 77u) sink[0] = acc.z ^ acc.w;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t maxbytes = (size_t)1 << 30;
    uint4* src; unsigned* sink; unsigned long long* cyc;
    hipMalloc(&src, maxbytes); hipMalloc(&sink, 4); hipMalloc(&cyc, 4096 * 8);
    hipMemset(src, 1, maxbytes);
    hipFuncSetAttribute((const void*)fill<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<unsigned long long> h(4096);
    for (int same = 0; same < 2; ++same)
    for (size_t ws : {(size_t)256 << 10, (size_t)2 << 20, (size_t)32 << 20, (size_t)1 << 30})
        for (int blocks : {256, 512, 1024})
            for (int mode = 0; mode < 2; ++mode) {
                const int iters = 2000;
                const size_t ntiles = ws / 8192;
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(256), 65536, 0, src, ntiles, iters, same, sink, cyc);
                    else hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(256), 0, 0, src, ntiles, iters, same, sink, cyc);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
                double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
                const double per_cu = (double)iters * 8192 / mean * (blocks / 256.0);
                printf("same=%d ws=%7zu KiB blocks/CU=%d %s: %.0f cycles/tile/block, %.1f B/clk/CU\n", same, ws >> 10,
                       blocks / 256, mode ? "regs" : "dma ", mean / iters, per_cu);
            }
    return 0;
}
