// valu_rate.hip -- cycles per wave64 VALU instruction on gfx950 for the instruction mix of the NN compare loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  unsigned long long u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
  float s = s0;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP8(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
    if (OP == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 2) { REP8(asm volatile("v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %2\n v_min_f64 %2, %2, %3\n v_min_f64 %3, %3, %0\n v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %2\n v_min_f64 %2, %2, %3\n v_min_f64 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 3) { REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %2\n v_cmp_lt_u64 vcc, %2, %3\n v_cmp_lt_u64 vcc, %3, %0\n v_cmp_lt_u64 vcc, %0, %2\n v_cmp_lt_u64 vcc, %1, %3\n v_cmp_lt_u64 vcc, %2, %0\n v_cmp_lt_u64 vcc, %3, %1" : : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "vcc");) }
    if (OP == 4) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");) }
    if (OP == 5) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    if (OP == 6) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %4" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");) }
    if (OP == 7) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 8) {  // the compare-loop mix: 3 sub(sgpr) 3 mul 2 add + v_min_f64 per pair, 8 independent pairs
      REP8(asm volatile(
        "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %8, %5\n v_sub_f32 %2, %8, %6\n"
        "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n"
        "v_add_f32 %0, %0, %1\n v_add_f32 %3, %0, %2\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(s));
        asm volatile("v_min_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));)
    }
    if (OP == 9) {  // same with v_cmp_lt_u64 + 2 cndmask
      REP8(asm volatile(
        "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %8, %5\n v_sub_f32 %2, %8, %6\n"
        "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n"
        "v_add_f32 %0, %0, %1\n v_add_f32 %3, %0, %2\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(s));
        asm volatile("v_cmp_lt_u64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(a4), "+v"(a5) : "v"(u0), "v"(u1) : "vcc");)
    }
    if (OP == 10) { REP8(asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %4\n v_min_u32 %4, %4, %5\n v_min_u32 %5, %5, %6\n v_min_u32 %6, %6, %7\n v_min_u32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(u0 + u1 + u2 + u3);
}

template <int OP>
void run(const char* name, int instr_per_iter, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves)
  float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 4000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(out, 100, 1.5f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<OP><<<blocks, 256>>>(out, iters, 1.5f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // wave-instructions per SIMD = waves_per_simd * iters * instr_per_iter
  const double winst = (double)waves_per_simd * iters * instr_per_iter;
  printf("%-34s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_sub_f32 (sgpr src)", 64, w);
    run<1>("v_mul_f32", 64, w);
    run<7>("v_fma_f32", 64, w);
    run<2>("v_min_f64", 64, w);
    run<3>("v_cmp_lt_u64", 64, w);
    run<6>("v_cmp_lt_f32", 64, w);
    run<4>("v_cndmask_b32", 64, w);
    run<10>("v_min_u32", 64, w);
    run<5>("v_pk_mul_f32", 64, w);
    run<8>("pair mix: 8 f32 + v_min_f64", 72, w);
    run<9>("pair mix: 8 f32 + cmp_u64 + 2 cnd", 88, w);
  }
  return 0;
}
