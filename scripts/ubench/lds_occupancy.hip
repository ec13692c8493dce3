// How many 64-lane workgroups fit one CU for a given dynamic LDS size (the allocation granule decides where the steps are).
// build: hipcc --offload-arch=gfx950 -O2 -o lds_occupancy lds_occupancy.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *o) {
    extern __shared__ float s[];
    s[threadIdx.x] = 1.0f;
    __syncthreads();
    o[threadIdx.x] = s[63 - threadIdx.x];
}
int main() {
    int prev = -1;
    for (size_t lds = 1024; lds <= 16384; lds += 64) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k, 64, lds) != hipSuccess) return 1;
        if (blocks != prev) std::printf("lds %zu B -> %d workgroups per CU\n", lds, blocks);
        prev = blocks;
    }
    return 0;
}
