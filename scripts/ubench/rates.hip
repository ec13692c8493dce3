// Instruction issue-rate microbenchmark for gfx950 (not part of the product): cycles per wave-instruction
// with 8 waves/SIMD resident, so the SIMD's issue rate (not one wave's latency) is measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
#define UNROLL 16
#define REP(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, float fseed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = b * 3u + 1u, d = a + 7u;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p = {fseed, fseed*2}, q = {1.0001f, 0.9999f}, r2 = {0.5f, 0.25f}, r3 = {0.1f,0.2f};
  float x = fseed + threadIdx.x * 1e-3f, y = x * 1.0001f + 0.5f, z = y * 0.999f + 0.25f, w = 1.5f + x;
  for (int i = 0; i < N_ITER; i++) {
    if (OP == 0) { REP(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (OP == 1) { REP(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (OP == 2) { REP(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (OP == 3) { REP(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w) : "v"(y), "v"(z));) }
    if (OP == 4) { REP(asm volatile("v_sqrt_f32 %0, %0" : "+v"(x)); asm volatile("v_sqrt_f32 %0, %0" : "+v"(w));) }
    if (OP == 5) { REP(asm volatile("v_rcp_f32 %0, %0" : "+v"(x)); asm volatile("v_rcp_f32 %0, %0" : "+v"(w));) }
    if (OP == 6) { REP(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(d));) }
    if (OP == 7) { REP(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(c));) }
    if (OP == 8) { REP(asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)); asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(w) : "v"(y), "v"(z));) }
    if (OP == 9) { REP(asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(x) : "v"(a)); asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(w) : "v"(c));) }
    if (OP == 10) { REP(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));) }
    if (OP == 11) { REP(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(b)); asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(b));) }
    if (OP == 13) { REP(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r2) : "v"(q));) }
    if (OP == 14) { REP(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(q), "v"(r2)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r3) : "v"(q), "v"(r2));) }
    if (OP == 15) { REP(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r2) : "v"(q));) }
    if (OP == 16) { REP(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c) : "v"(d));) }
    if (OP == 17) { REP(asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x), "v"(y) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(w), "v"(z): "vcc");) }
    if (OP == 18) { REP(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(b)); asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(c) : "v"(d));) }
    if (OP == 19) { REP(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(y) : "vcc"); asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(w) : "v"(y), "v"(z));) }
    if (OP == 20) { REP(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a) : "v"(b)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(c) : "v"(d));) }
    if (OP == 21) { REP(asm volatile("v_cmp_lt_f32_e64 s[12:13], %0, %1" :: "v"(x), "v"(y) : "s12","s13"); asm volatile("v_cmp_gt_f32_e64 s[14:15], %0, %1" :: "v"(w), "v"(z): "s14","s15");) }
    if (OP == 22) { REP(asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(w) : "v"(z));) }
    if (OP == 23) { REP(asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y), "v"(z) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w) : "v"(z), "v"(y): "vcc");) }
    if (OP == 12) { REP(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(w) : "v"(z));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(x) ^ __float_as_uint(w) ^ __float_as_uint(p.x+p.y+r2.x+r2.y+r3.x+r3.y);
}

template <int OP> double run(const char* name, uint32_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * 8;  // 8 blocks of 256 = 32 waves per CU = 8 waves / SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u, 1.1f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts_per_simd = 8.0 * N_ITER * UNROLL * 2;  // waves/SIMD * instrs per wave
  double ns_per_inst = ms * 1e6 / insts_per_simd;
  printf("%-28s %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz)\n", name, ms, ns_per_inst, ns_per_inst * 2.4);
  return ns_per_inst;
}

int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_u32_u24", d); run<3>("v_fma_f32", d);
  run<4>("v_sqrt_f32", d); run<5>("v_rcp_f32", d); run<6>("v_mul_hi_u32", d); run<7>("v_xor/v_lshrrev", d);
  run<8>("v_div_fixup_f32", d); run<9>("v_cvt_f32_u32", d); run<10>("v_mad_u32_u24", d); run<11>("v_lshl_add/v_xad", d);
  run<12>("v_mul_f32/v_add_f32", d); run<13>("v_pk_mul_f32", d); run<14>("v_pk_fma_f32", d); run<15>("v_pk_add_f32", d); run<16>("v_cndmask_b32", d); run<17>("v_cmp_f32", d); run<18>("ds_bpermute+wait", d); run<19>("v_div_scale/v_div_fmas", d); run<20>("v_cndmask_e64 sgpr", d); run<21>("v_cmp_e64 sgpr", d); run<22>("v_max/v_min", d); run<23>("cmp+cndmask pair(x2 instr)", d);
  return 0;
}
