// What is the fixed part of a gather's cost?  gather.hip: a wave's gather of per-lane records costs ~18 ns per CU plus ~18 ns
// per 16 bytes.  Here a lane fetches A bytes from a random 64-byte record and then B bytes either from the OTHER half of the
// same 128-byte line or from a second, independent random record: is the fixed part paid per line touched?
// build: hipcc --offload-arch=gfx950 -O3 -o gather3 gather3.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__device__ __forceinline__ float fetch(const float *p) {
    float acc = 0.0f;
    int off = 0;
#pragma unroll
    for (; off + 4 <= BYTES / 4; off += 4) { const float4 a = *reinterpret_cast<const float4 *>(p + off); acc += (a.x + a.y) + (a.z + a.w); }
#pragma unroll
    for (; off + 2 <= BYTES / 4; off += 2) { const float2 a = *reinterpret_cast<const float2 *>(p + off); acc += a.x + a.y; }
#pragma unroll
    for (; off < BYTES / 4; off++) acc += p[off];
    return acc;
}
// WHERE 0: no second fetch; 1: the other 64-byte half of the same 128-byte line; 2: an independent random record
template <int A, int B, int WHERE>
__global__ __launch_bounds__(256) void k_gather(const float *tbl, uint32_t mask, uint32_t iters, float *out) {
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t r1 = ((idx >> 8) & mask) & ~1u, r2 = WHERE == 1 ? r1 + 1u : ((idx >> 3) & mask) & ~1u;
        acc += fetch<A>(tbl + (size_t)r1 * 16u);
        if (WHERE != 0) acc += fetch<B>(tbl + (size_t)r2 * 16u);
        idx = idx * 1664525u + 1013904223u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int A, int B, int WHERE>
static void run(const float *tbl, uint32_t records, float *out, const char *what) {
    const uint32_t iters = 4096, blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather<A, B, WHERE>), dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, 64u, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather<A, B, WHERE>), dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const char *where[3] = {"", "+ %2d B from the same 128-B line", "+ %2d B from another random line"};
    char second[64] = "";
    if (WHERE) std::snprintf(second, sizeof second, where[WHERE], B);
    std::printf("%-8s %2d B %-36s %6.2f ns per wave and CU\n", what, A, second, ms * 1e6 / ((double)blocks * threads / 64 * iters / 256.0));
}
int main() {
    const uint32_t sizes[2] = {256, 16384};
    const char *names[2] = {"16KB/L1", "1MB/L2"};
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int s = 0; s < 2; s++) {
        float *tbl;
        (void)hipMalloc(&tbl, (size_t)sizes[s] * 64);
        (void)hipMemset(tbl, 0, (size_t)sizes[s] * 64);
        run<16, 0, 0>(tbl, sizes[s], out, names[s]);
        run<32, 0, 0>(tbl, sizes[s], out, names[s]);
        run<16, 16, 1>(tbl, sizes[s], out, names[s]);
        run<16, 16, 2>(tbl, sizes[s], out, names[s]);
        run<56, 0, 0>(tbl, sizes[s], out, names[s]);
        run<36, 0, 0>(tbl, sizes[s], out, names[s]);
        run<56, 36, 1>(tbl, sizes[s], out, names[s]);
        run<56, 36, 2>(tbl, sizes[s], out, names[s]);
        run<36, 36, 1>(tbl, sizes[s], out, names[s]);
        run<36, 36, 2>(tbl, sizes[s], out, names[s]);
        run<56, 56, 1>(tbl, sizes[s], out, names[s]);
        run<56, 56, 2>(tbl, sizes[s], out, names[s]);
        (void)hipFree(tbl);
    }
    return 0;
}
