// Do the vector L1's data banks explain the gather rate?  In gather.hip every lane of a load instruction reads the SAME offset
// of its own 64-byte record (quad j of the record in instruction j), so the 64 addresses of one instruction agree modulo 64:
// if the L1 interleaves 16-byte banks across a 128-byte line they fall into 2 of 8 banks.  Here the quads of a record are read
// in an order rotated by the record's index (instruction j reads quad (j + r) & 3 of record r): the same 56 bytes per lane,
// but the addresses of one instruction spread over all offsets.
// build: hipcc --offload-arch=gfx950 -O3 -o gather4 gather4.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0: quad j at offset 16 j (gather.hip); 1: rotated by the record index; 2: rotated by the lane index;
//      3: rotated by the record index over the 8 quads of a 128-byte pair of records (bit 6 of the address too)
template <int MODE, int QUADS>
__global__ __launch_bounds__(256) void k_gather(const float *tbl, uint32_t mask, uint32_t iters, float *out) {
    uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t r = (idx >> 8) & mask;
        const float *p = tbl + (size_t)r * 16u;
        const uint32_t rot = MODE == 1 ? r : MODE == 2 ? threadIdx.x : 0u;
#pragma unroll
        for (int j = 0; j < QUADS; j++) {
            const float4 a = *reinterpret_cast<const float4 *>(p + (((uint32_t)j + rot) & 3u) * 4u);
            acc += (a.x + a.y) + (a.z + a.w);
        }
        idx = idx * 1664525u + 1013904223u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE, int QUADS>
static void run(const float *tbl, uint32_t records, float *out, const char *what) {
    const uint32_t iters = 4096, blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather<MODE, QUADS>), dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, 64u, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather<MODE, QUADS>), dim3(blocks), dim3(threads), 0, 0, tbl, records - 1, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const char *mode[3] = {"quad j at offset 16 j", "quads rotated by the record index", "quads rotated by the lane index"};
    std::printf("%-8s %d x 16 B, %-36s %7.2f ns per wave and CU, %8.1f GB/s\n", what, QUADS, mode[MODE],
                ms * 1e6 / ((double)blocks * threads / 64 * iters / 256.0), (double)blocks * threads * iters * 16.0 * QUADS / ms / 1e6);
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
        run<0, 1>(tbl, sizes[s], out, names[s]);
        run<1, 1>(tbl, sizes[s], out, names[s]);
        run<2, 1>(tbl, sizes[s], out, names[s]);
        run<0, 2>(tbl, sizes[s], out, names[s]);
        run<1, 2>(tbl, sizes[s], out, names[s]);
        run<0, 3>(tbl, sizes[s], out, names[s]);
        run<1, 3>(tbl, sizes[s], out, names[s]);
        run<0, 4>(tbl, sizes[s], out, names[s]);
        run<1, 4>(tbl, sizes[s], out, names[s]);
        run<2, 4>(tbl, sizes[s], out, names[s]);
        (void)hipFree(tbl);
    }
    return 0;
}
