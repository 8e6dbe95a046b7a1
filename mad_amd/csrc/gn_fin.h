// GroupNorm finalize, shared by the stand-alone kernel (groupnorm.hip) and by producers that finish it in their own
// launch: per-tile partial sums (sum, sum of squares per channel) -> one (a, d) pair per (sample, channel) with
// y = x*a + d, folding mean, rstd, gain/offset and the optional AdaGN scale/shift.  Both users run the SAME
// functions in the same order, so their results are identical to the last bit.
#pragma once
#include "common.h"

namespace mad {

struct GnFin {
    const float* partial1; int C1, ns1;       // partial sums (B, ns1, C1, 2) of the first C1 channels
    const float* partial2; int C2, ns2;       // of the next C2 channels (C2 = 0: partial2 = partial1, ns2 = 1)
    const float* gamma; const float* beta;    // gain / offset over the C1 + C2 channels
    const float* ada; int ada_stride;         // AdaGN rows (scale first, shift at + C), or null
    float* coef;                              // out (B, C, 2); null: no finalize
    int groups, HW; float eps, scale;
    unsigned* counters;                       // in-launch form: one arrival counter per sample, zero between launches
    int arrivals;                             // in-launch form: workgroups per sample over the layer's launches
};

__device__ __forceinline__ int gn_max_ns(const GnFin& f) { return f.ns1 > f.ns2 ? f.ns1 : f.ns2; }

// Per-lane shares of the sums of GB groups gs[0..GB) of sample b (a negative entry: no group).  A team of NT lanes
// walks a group's cpg * max_ns (channel, split) items it = lane, lane + NT, ...; each lane adds its items in that
// order in fp64, four loads in flight per group (and GB groups side by side: their chains are independent).
template <int NT, int GB>
__device__ __forceinline__ void gn_lane_sums(const GnFin& f, int b, const int* gs, int lane, double* s, double* q) {
    const int C = f.C1 + f.C2, cpg = C / f.groups;
    const int max_ns = gn_max_ns(f);
    const int total = cpg * max_ns;
#pragma unroll
    for (int k = 0; k < GB; ++k) { s[k] = 0.0; q[k] = 0.0; }
    for (int it0 = lane; it0 < total; it0 += 4 * NT) {
        float2 v[GB][4];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            const int g = gs[k] < 0 ? 0 : gs[k];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = it0 + NT * u;
                const int itc = it < total ? it : 0;
                const int c = g * cpg + itc / max_ns, sp = itc % max_ns;
                const bool first = c < f.C1;
                const int ns = first ? f.ns1 : f.ns2;
                const int spc = sp < ns ? sp : ns - 1;
                const float* p = first ? f.partial1 + (((size_t)b * f.ns1 + spc) * f.C1 + c) * 2
                                       : f.partial2 + (((size_t)b * f.ns2 + spc) * f.C2 + (c - f.C1)) * 2;
                v[k][u] = *reinterpret_cast<const float2*>(p);
                if (it >= total || sp >= ns) v[k][u] = make_float2(0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int k = 0; k < GB; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[k] += (double)v[k][u].x; q[k] += (double)v[k][u].y; }
    }
}

// butterfly over the 64 lanes of a wave: every lane ends with the wave's sums
template <int GB> __device__ __forceinline__ void gn_wave_sums(double* s, double* q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < GB; ++k) { s[k] += __shfl_xor(s[k], off); q[k] += __shfl_xor(q[k], off); }
}

// the (a, d) pairs of group g from its sums; lanes lane, lane + NT, ... take its channels
template <int NT>
__device__ __forceinline__ void gn_write_coef(const GnFin& f, int b, int g, int lane, double s, double q) {
    const int C = f.C1 + f.C2, cpg = C / f.groups;
    const double n = (double)f.HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)f.eps));
    for (int i = lane; i < cpg; i += NT) {
        const int c = g * cpg + i;
        const float rg = rstd_f * f.gamma[c];
        float a = rg;
        float d = f.beta[c] - mean_f * rg;
        if (f.ada) {
            const float sc = f.ada[(size_t)b * f.ada_stride + c] + 1.0f;
            const float sh = f.ada[(size_t)b * f.ada_stride + C + c];
            a = a * sc;
            d = fmaf(d, sc, sh);
        }
        f.coef[((size_t)b * C + c) * 2] = a * f.scale;
        f.coef[((size_t)b * C + c) * 2 + 1] = d * f.scale;
    }
}

// four-wave teams are used for groups of 512 and more (channel, split) items, one wave otherwise (the stand-alone
// launcher and the in-launch form must agree: the team size is part of the summation order)
__device__ __host__ __forceinline__ bool gn_wide_team(int cpg, int max_ns) { return cpg * max_ns >= 512; }

// The whole finalize of sample b by one workgroup of NTH threads (256 or 512), as the last step of a producer's
// launch.  `wsum` is LDS scratch of at least 1 KiB that nobody else uses any more.  Every thread of the workgroup
// must call it (it contains workgroup barriers).  Results equal the stand-alone kernel's bit for bit: the same
// teams (four waves per group from 512 items on, one wave otherwise), the same per-lane order, the same butterfly,
// the same wave order; groups are merely processed eight at a time per team so that their memory round trips overlap.
template <int NTH>
__device__ __forceinline__ void gn_fin_tail(const GnFin& f, int b, int tid, double* wsum) {
    constexpr int GB = 8;
    const int C = f.C1 + f.C2, cpg = C / f.groups;
    if (gn_wide_team(cpg, gn_max_ns(f))) {
        constexpr int NT = 256, TEAMS = NTH / NT;
        const int team = tid / NT, lane = tid % NT, wv = lane >> 6;
        for (int g0 = 0; g0 < f.groups; g0 += GB * TEAMS) {
            int gs[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) { const int g = g0 + k * TEAMS + team; gs[k] = g < f.groups ? g : -1; }
            double s[GB], q[GB];
            gn_lane_sums<NT, GB>(f, b, gs, lane, s, q);
            gn_wave_sums<GB>(s, q);
            if ((lane & 63) == 0) {
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    wsum[((team * GB + k) * 4 + wv) * 2] = s[k];
                    wsum[((team * GB + k) * 4 + wv) * 2 + 1] = q[k];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                double st = 0.0, qt = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { st += wsum[((team * GB + k) * 4 + w) * 2]; qt += wsum[((team * GB + k) * 4 + w) * 2 + 1]; }
                if (gs[k] >= 0) gn_write_coef<NT>(f, b, gs[k], lane, st, qt);
            }
            __syncthreads();
        }
    } else {
        constexpr int NT = 64, TEAMS = NTH / NT;
        const int team = tid / NT, lane = tid % NT;
        for (int g0 = 0; g0 < f.groups; g0 += GB * TEAMS) {
            int gs[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) { const int g = g0 + k * TEAMS + team; gs[k] = g < f.groups ? g : -1; }
            double s[GB], q[GB];
            gn_lane_sums<NT, GB>(f, b, gs, lane, s, q);
            gn_wave_sums<GB>(s, q);
#pragma unroll
            for (int k = 0; k < GB; ++k)
                if (gs[k] >= 0) gn_write_coef<NT>(f, b, gs[k], lane, s[k], q[k]);
        }
    }
}

// Arrival of one workgroup at sample b's counter after it has stored its partial sums; true in the workgroup that
// arrives last (it then sees every other workgroup's sums and resets the counter for the next launch).  Every thread
// of the workgroup must call it; `flag` is one LDS word.  Visibility does not depend on where workgroups run: every
// storing wave drains its stores, the workgroup's barrier, an agent-scope release in front of the (relaxed) counter
// update, and an agent-scope acquire in the last arriver before its barrier and its plain loads.
__device__ __forceinline__ bool gn_fin_arrive(const GnFin& f, int b, int tid, unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(f.counters + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old + 1u == (unsigned)f.arrivals;
        if (last) {
            __hip_atomic_store(f.counters + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    return *flag != 0u;
}

}  // namespace mad
