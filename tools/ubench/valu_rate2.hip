// valu_rate2.hip -- follow-up: is the half rate of v_sub_f32 due to its SGPR operand?  v_cndmask forms; full pair mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define A8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define I8(op) op " %0, %0, %1\n" op " %1, %1, %2\n" op " %2, %2, %3\n" op " %3, %3, %4\n" op " %4, %4, %5\n" op " %5, %5, %6\n" op " %6, %6, %7\n" op " %7, %7, %0"
#define S8(op) op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7"

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s0, unsigned long long m0) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1;
  float s = s0;
  unsigned long long msk = m0;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP8(asm volatile(I8("v_sub_f32") : A8);) }
    if (OP == 1) { REP8(asm volatile(S8("v_sub_f32") : A8 : "s"(s));) }
    if (OP == 2) { REP8(asm volatile(I8("v_add_f32") : A8);) }
    if (OP == 3) { REP8(asm volatile(S8("v_mul_f32") : A8 : "s"(s));) }
    if (OP == 4) { REP8(asm volatile(I8("v_min_f32") : A8);) }
    if (OP == 5) { REP8(asm volatile(I8("v_and_b32") : A8);) }
    if (OP == 6) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, %8\n v_cndmask_b32 %1, %1, %2, %8\n v_cndmask_b32 %2, %2, %3, %8\n v_cndmask_b32 %3, %3, %4, %8\n v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %8\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %0, %8" : A8 : "s"(msk));) }
    if (OP == 7) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : A8);) }
    if (OP == 8) {  // pair: VGPR query operands, 8 f32 + v_min_f64
      REP8(asm volatile(
        "v_sub_f32 %0, %7, %4\n v_sub_f32 %1, %7, %5\n v_sub_f32 %2, %7, %6\n"
        "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n"
        "v_add_f32 %0, %0, %1\n v_add_f32 %3, %0, %2\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        asm volatile("v_min_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));)
    }
    if (OP == 9) {  // pair: SGPR query operands, 8 f32 + v_min_f64
      REP8(asm volatile(
        "v_sub_f32 %0, %7, %4\n v_sub_f32 %1, %7, %5\n v_sub_f32 %2, %7, %6\n"
        "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n"
        "v_add_f32 %0, %0, %1\n v_add_f32 %3, %0, %2\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "s"(s));
        asm volatile("v_min_f64 %0, %0, %1" : "+v"(d0) : "v"(d1));)
    }
    if (OP == 10) {  // pair: VGPR queries, f32 compare + 2 cndmask (first-seen minimum)
      REP8(asm volatile(
        "v_sub_f32 %0, %7, %4\n v_sub_f32 %1, %7, %5\n v_sub_f32 %2, %7, %6\n"
        "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n"
        "v_add_f32 %0, %0, %1\n v_add_f32 %3, %0, %2\n"
        "v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %4, %4, %3, vcc\n v_cndmask_b32 %5, %5, %6, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(a6), "v"(a7) : "vcc");)
    }
    if (OP == 11) { REP8(asm volatile("v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %0\n v_max_f64 %0, %0, %1\n v_max_f64 %1, %1, %0\n v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %0\n v_max_f64 %0, %0, %1\n v_max_f64 %1, %1, %0" : "+v"(d0), "+v"(d1));) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1);
}

template <int OP>
void run(const char* name, int instr_per_iter, int w) {
  const int blocks = 256 * w;
  float* out; (void)hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 4000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(out, 100, 1.5f, 0x5555555555555555ull);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  k<OP><<<blocks, 256>>>(out, iters, 1.5f, 0x5555555555555555ull);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double winst = (double)w * iters * instr_per_iter;
  printf("%-44s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, w, ms, ms * 1e6 / winst);
  (void)hipFree(out);
}

int main() {
  for (int w : {2, 4}) {
    run<0>("v_sub_f32 vgpr,vgpr", 64, w);
    run<1>("v_sub_f32 sgpr,vgpr", 64, w);
    run<2>("v_add_f32 vgpr,vgpr", 64, w);
    run<3>("v_mul_f32 sgpr,vgpr", 64, w);
    run<4>("v_min_f32", 64, w);
    run<5>("v_and_b32", 64, w);
    run<6>("v_cndmask_b32 e64 sgpr mask", 64, w);
    run<7>("v_mov_b32", 64, w);
    run<11>("v_min/max_f64 dependent", 64, w);
    run<8>("pair: vgpr q, 8 f32 + v_min_f64 (9)", 72, w);
    run<9>("pair: sgpr q, 8 f32 + v_min_f64 (9)", 72, w);
    run<10>("pair: vgpr q, 8 f32 + cmp_f32 + 2 cnd (11)", 88, w);
  }
  return 0;
}
