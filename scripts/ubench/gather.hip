// Per-lane record gathers through the vector L1: how many bytes per clock and CU the texture path delivers when every
// lane of a wave reads its own 64-byte-aligned record (what a tree walk's node step does).
// build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int BYTES>
__global__ __launch_bounds__(256) void k_gather(const float4 *tbl, uint32_t mask, uint32_t iters, float *out) {
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    for (uint32_t i = 0; i < iters; i++) {
        const float4 *p = tbl + (size_t)((idx >> 8) & mask) * 4u;
        float4 a = p[0];
        acc += a.x + a.w;
        if (BYTES >= 32 && BYTES != 24) { float4 b = p[1]; acc += b.x + b.w; }
        if (BYTES == 36) { float c = *reinterpret_cast<const float *>(p + 2); acc += c; }
        if (BYTES == 40) { float2 c = *reinterpret_cast<const float2 *>(p + 2); acc += c.x + c.y; }
        if (BYTES == 24) { float2 c = *reinterpret_cast<const float2 *>(p + 1); acc += c.x + c.y; }
        if (BYTES >= 48) { float4 c = p[2]; acc += c.x + c.w; }
        if (BYTES == 56) { float2 d = *reinterpret_cast<const float2 *>(p + 3); acc += d.x + d.y; }
        if (BYTES >= 64) { float4 d = p[3]; acc += d.x + d.w; }
        idx = idx * 1664525u + 1013904223u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int BYTES>
static void run(const float4 *tbl, uint32_t records, float *out, const char *what) {
    const uint32_t iters = 4096, blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gather<BYTES>, dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, 64u, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_gather<BYTES>, dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * threads * iters * BYTES;
    std::printf("%-10s %2d B/lane: %8.1f GB/s = %6.1f B/clk/CU at 2.4 GHz, %5.2f ns per wave-gather per CU\n", what, BYTES, bytes / ms / 1e6,
                bytes / (ms * 1e-3) / 256.0 / 2.4e9, ms * 1e6 / ((double)blocks * threads / 64 * iters / 256.0));
}
int main() {
    const uint32_t sizes[3] = {256, 16384, 1u << 20};
    // (24: x4 + x2; 36: x4 + x4 + x1; 40: x4 + x4 + x2; 56: x4 + x4 + x4 + x2) // records of 64 B: 16 KB (L1), 1 MB (L2), 64 MB (MALL / HBM)
    const char *names[3] = {"16KB/L1", "1MB/L2", "64MB"};
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int s = 0; s < 3; s++) {
        float4 *tbl;
        hipMalloc(&tbl, (size_t)sizes[s] * 64);
        hipMemset(tbl, 0, (size_t)sizes[s] * 64);
        run<16>(tbl, sizes[s], out, names[s]);
        run<24>(tbl, sizes[s], out, names[s]);
        run<32>(tbl, sizes[s], out, names[s]);
        run<36>(tbl, sizes[s], out, names[s]);
        run<40>(tbl, sizes[s], out, names[s]);
        run<48>(tbl, sizes[s], out, names[s]);
        run<56>(tbl, sizes[s], out, names[s]);
        run<64>(tbl, sizes[s], out, names[s]);
        hipFree(tbl);
    }
    return 0;
}
