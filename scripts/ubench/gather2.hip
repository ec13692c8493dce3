// Per-lane 56-byte record gathers through the vector L1 (a tree walk's node step), by the MIX of load widths: does the L1
// charge per load instruction, per byte, or per 16-byte load?  Every loaded component is consumed.
// build: hipcc --offload-arch=gfx950 -O3 -o gather2 gather2.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
// PATTERN: number of leading 16-byte loads (0..3); the rest of the 56 bytes comes as 8-byte loads (PATTERN 4: 14 4-byte loads)
template <int PATTERN>
__global__ __launch_bounds__(256) void k_gather(const float *tbl, uint32_t mask, uint32_t iters, float *out, unsigned long long lanes) {
    if (!((lanes >> (threadIdx.x & 63u)) & 1ull)) return; // a wave with part of its lanes switched off (a diverged walk)
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    for (uint32_t i = 0; i < iters; i++) {
        const float *p = tbl + (size_t)((idx >> 8) & mask) * 16u;
        if (PATTERN == 4) {
#pragma unroll
            for (int k = 0; k < 14; k++) acc += __builtin_nontemporal_load(p + k) * 0.0f + p[k];
        } else {
            int off = 0;
#pragma unroll
            for (int k = 0; k < PATTERN; k++) { const float4 a = *reinterpret_cast<const float4 *>(p + off); acc += (a.x + a.y) + (a.z + a.w); off += 4; }
#pragma unroll
            for (; off < 14; off += 2) { const float2 a = *reinterpret_cast<const float2 *>(p + off); acc += a.x + a.y; }
        }
        idx = idx * 1664525u + 1013904223u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int PATTERN>
static void run(const float *tbl, uint32_t records, float *out, const char *what, unsigned long long lanes = ~0ull) {
    const uint32_t iters = 4096, blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather<PATTERN>, dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, 64u, out, lanes);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_gather<PATTERN>, dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, iters, out, lanes);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int on = __builtin_popcountll(lanes);
    const double bytes = (double)blocks * threads * iters * 56.0 * on / 64.0;
    const char *mix[5] = {"7 x 8 B", "16 + 5 x 8 B", "2 x 16 + 3 x 8 B", "3 x 16 + 8 B (the node step)", "14 x 4 B"};
    std::printf("%-8s %2d lanes on, 56 B/lane as %-30s %8.1f GB/s, %6.2f ns per wave-gather per CU\n", what, on, mix[PATTERN], bytes / ms / 1e6,
                ms * 1e6 / ((double)blocks * threads / 64 * iters / 256.0));
}
int main() {
    const uint32_t sizes[2] = {256, 16384};
    const char *names[2] = {"16KB/L1", "1MB/L2"};
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int s = 0; s < 2; s++) {
        float *tbl;
        hipMalloc(&tbl, (size_t)sizes[s] * 64);
        hipMemset(tbl, 0, (size_t)sizes[s] * 64);
        run<3>(tbl, sizes[s], out, names[s]);
        run<2>(tbl, sizes[s], out, names[s]);
        run<1>(tbl, sizes[s], out, names[s]);
        run<0>(tbl, sizes[s], out, names[s]);
        // part of the wave switched off: does a gather cost per instruction or per active lane?
        run<3>(tbl, sizes[s], out, names[s], 0x00000000FFFFFFFFull);
        run<3>(tbl, sizes[s], out, names[s], 0x5555555555555555ull);
        run<3>(tbl, sizes[s], out, names[s], 0x000000000000FFFFull);
        run<3>(tbl, sizes[s], out, names[s], 0x1111111111111111ull);
        run<3>(tbl, sizes[s], out, names[s], 0x00000000000000FFull);
        run<3>(tbl, sizes[s], out, names[s], 0x0101010101010101ull);
        hipFree(tbl);
    }
    return 0;
}
