// valu_rate3.hip -- integer VALU rates of the SIFT top-2 scan and their overlap with i8 MFMAs on gfx950.
// hipcc --offload-arch=gfx950 -O3 valu_rate3.hip -o valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define REP8(x) x x x x x x x x
#define A8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define I8(op) op " %0, %0, %1\n" op " %1, %1, %2\n" op " %2, %2, %3\n" op " %3, %3, %4\n" op " %4, %4, %5\n" op " %5, %5, %6\n" op " %6, %6, %7\n" op " %7, %7, %0"
#define T8(op) op " %0, %0, %1, %2\n" op " %1, %1, %2, %3\n" op " %2, %2, %3, %4\n" op " %3, %3, %4, %5\n" op " %4, %4, %5, %6\n" op " %5, %5, %6, %7\n" op " %6, %6, %7, %0\n" op " %7, %7, %0, %1"
#define L8(op) op " %0, %0, 8, %1\n" op " %1, %1, 8, %2\n" op " %2, %2, 8, %3\n" op " %3, %3, 8, %4\n" op " %4, %4, 8, %5\n" op " %5, %5, 8, %6\n" op " %6, %6, 8, %7\n" op " %7, %7, 8, %0"
#define C8(op) op " %0, %0, 8, 17\n" op " %1, %1, 8, 18\n" op " %2, %2, 8, 19\n" op " %3, %3, 8, 20\n" op " %4, %4, 8, 21\n" op " %5, %5, 8, 22\n" op " %6, %6, 8, 23\n" op " %7, %7, 8, 24"
// the scan of one accumulator value in both directions: 6 instructions
#define SCAN(acc) "v_lshl_add_u32 %4, " acc ", 8, 33\n v_med3_i32 %1, %0, %1, %4\n v_max_i32 %0, %0, %4\n" \
                  "v_lshl_add_u32 %5, " acc ", 8, %6\n v_med3_i32 %3, %2, %3, %5\n v_max_i32 %2, %2, %5\n"

template <int OP>
__global__ __launch_bounds__(256, 4) void k(int* out, int iters, int s0) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v16i c0 = {0}, c1 = {0};
  v4i fa = {a0, a1, a2, a3}, fb = {a4, a5, a6, a7};
  int t0 = 0, t1 = 0;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP8(asm volatile(I8("v_max_i32") : A8);) }
    if (OP == 1) { REP8(asm volatile(T8("v_med3_i32") : A8);) }
    if (OP == 2) { REP8(asm volatile(L8("v_lshl_add_u32") : A8);) }
    if (OP == 3) { REP8(asm volatile(C8("v_lshl_add_u32") : A8);) }
    if (OP == 4) { REP8(asm volatile(T8("v_max3_i32") : A8);) }
    if (OP == 5) { REP8(asm volatile(I8("v_add_u32") : A8);) }
    if (OP == 6) { REP8(asm volatile(L8("v_lshl_or_b32") : A8);) }
    if (OP == 11) { REP8(asm volatile(T8("v_mad_i32_i24") : A8);) }
    if (OP == 12) { REP8(asm volatile(T8("v_mad_u32_u24") : A8);) }
    if (OP == 13) { REP8(asm volatile(I8("v_lshlrev_b32") : A8);) }
    if (OP == 14) { REP8(asm volatile(I8("v_mul_u32_u24") : A8);) }
    if (OP == 15) { REP8(asm volatile(I8("v_max_u32") : A8);) }
    if (OP == 16) { REP8(asm volatile(I8("v_max_f32") : A8);) }
    if (OP == 17) { REP8(asm volatile(T8("v_med3_f32") : A8);) }
    if (OP == 18) { REP8(asm volatile(I8("v_pk_max_i16") : A8);) }
    if (OP == 19) { REP8(asm volatile(I8("v_max_i16") : A8);) }
    if (OP == 20) { REP8(asm volatile(T8("v_fma_f32") : A8);) }
    if (OP == 21) { REP8(asm volatile(T8("v_add3_u32") : A8);) }
    if (OP == 22) { REP8(asm volatile(T8("v_and_or_b32") : A8);) }
    if (OP == 23) { REP8(asm volatile(T8("v_perm_b32") : A8);) }
    if (OP == 24) { REP8(asm volatile(T8("v_alignbit_b32") : A8);) }
    if (OP == 25) { REP8(asm volatile(I8("v_or_b32") : A8);) }
#define CMP8(op) op " s[40:41], %0, %1\n" op " s[42:43], %1, %2\n" op " s[44:45], %2, %3\n" op " s[46:47], %3, %4\n" op " s[48:49], %4, %5\n" op " s[50:51], %5, %6\n" op " s[52:53], %6, %7\n" op " s[54:55], %7, %0"
#define CMPV8(op) op " vcc, %0, %1\n" op " vcc, %1, %2\n" op " vcc, %2, %3\n" op " vcc, %3, %4\n" op " vcc, %4, %5\n" op " vcc, %5, %6\n" op " vcc, %6, %7\n" op " vcc, %7, %0"
#define SCLOB "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "vcc"
    if (OP == 26) { REP8(asm volatile(CMP8("v_cmp_gt_i32_e64") : A8 : : SCLOB);) }
    if (OP == 27) { REP8(asm volatile(CMPV8("v_cmp_gt_i32_e32") : A8 : : SCLOB);) }
    if (OP == 28) { REP8(asm volatile(I8("v_sub_u32") : A8);) }
    if (OP == 29) { REP8(asm volatile(CMP8("v_cmp_gt_u32_e64") : A8 : : SCLOB);) }
    if (OP == 30) { REP8(asm volatile(CMP8("v_cmp_gt_i16_e64") : A8 : : SCLOB);) }
    if (OP == 31) { REP8(asm volatile(CMP8("v_cmp_gt_f32_e64") : A8 : : SCLOB);) }
    if (OP == 7) {   // 32 accumulator values scanned in both directions: 192 VALU
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm volatile(SCAN("%7") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1) : "v"(a4), "v"(c0[r]));
        asm volatile(SCAN("%7") : "+v"(a0), "+v"(a1), "+v"(a5), "+v"(a6), "=&v"(t0), "=&v"(t1) : "v"(a4), "v"(c1[r]));
      }
    }
    if (OP == 8) {   // 8 MFMAs alone
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, c1, 0, 0, 0);
      }
    }
    if (OP == 9 || OP == 10) {   // 8 MFMAs into one accumulator pair while the other pair is scanned (192 VALU); roles swap
      v16i d0 = {0}, d1 = {0};
#define PHASE(X0, X1, Y0, Y1)                                                                                          \
      _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                                  \
        X0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, X0, 0, 0, 0);                                               \
        if (OP == 10) X1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, X1, 0, 0, 0);                                 \
        _Pragma("unroll") for (int r = 4 * m; r < 4 * m + 2; ++r) {                                                   \
          asm volatile(SCAN("%7") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1) : "v"(a4), "v"(Y0[r])); \
          asm volatile(SCAN("%7") : "+v"(a0), "+v"(a1), "+v"(a5), "+v"(a6), "=&v"(t0), "=&v"(t1) : "v"(a4), "v"(Y1[r])); \
        }                                                                                                              \
        if (OP == 9) X1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, X1, 0, 0, 0);                                  \
        _Pragma("unroll") for (int r = 4 * m + 2; r < 4 * m + 4; ++r) {                                               \
          asm volatile(SCAN("%7") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1) : "v"(a4), "v"(Y0[r])); \
          asm volatile(SCAN("%7") : "+v"(a0), "+v"(a1), "+v"(a5), "+v"(a6), "=&v"(t0), "=&v"(t1) : "v"(a4), "v"(Y1[r])); \
        }                                                                                                              \
      }
      PHASE(d0, d1, c0, c1)
      PHASE(c0, c1, d0, d1)
      a7 += d0[3] + d1[5];
    }
  }
  int acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + t0 + t1;
  for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, int instr_per_iter, int w) {
  const int blocks = 256 * w;
  int* out; (void)hipMalloc(&out, blocks * 256 * sizeof(int));
  const int iters = 2000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<OP><<<blocks, 256>>>(out, 100, 1);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  k<OP><<<blocks, 256>>>(out, iters, 1);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double winst = (double)w * iters * instr_per_iter;
  printf("%-52s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD (%.1f ns per iteration per wave)\n", name, w, ms,
         ms * 1e6 / winst, ms * 1e6 / ((double)w * iters));
  (void)hipFree(out);
}

int main() {
  if (getenv("UBENCH_CMP_ONLY")) {
    for (int w : {2, 4}) {
      run<26>("v_cmp_gt_i32_e64 -> sgpr pair", 64, w);
      run<27>("v_cmp_gt_i32_e32 -> vcc", 64, w);
      run<28>("v_sub_u32", 64, w);
      run<5>("v_add_u32", 64, w);
      run<29>("v_cmp_gt_u32_e64", 64, w);
      run<30>("v_cmp_gt_i16_e64", 64, w);
      run<31>("v_cmp_gt_f32_e64", 64, w);
      run<0>("v_max_i32", 64, w);
    }
    return 0;
  }
  for (int w : {2, 4}) {
    run<0>("v_max_i32", 64, w);
    run<1>("v_med3_i32", 64, w);
    run<2>("v_lshl_add_u32 v,8,v", 64, w);
    run<3>("v_lshl_add_u32 v,8,inline", 64, w);
    run<4>("v_max3_i32", 64, w);
    run<5>("v_add_u32", 64, w);
    run<6>("v_lshl_or_b32", 64, w);
    run<11>("v_mad_i32_i24", 64, w);
    run<12>("v_mad_u32_u24", 64, w);
    run<13>("v_lshlrev_b32", 64, w);
    run<14>("v_mul_u32_u24", 64, w);
    run<15>("v_max_u32", 64, w);
    run<16>("v_max_f32", 64, w);
    run<17>("v_med3_f32", 64, w);
    run<18>("v_pk_max_i16", 64, w);
    run<19>("v_max_i16", 64, w);
    run<20>("v_fma_f32", 64, w);
    run<21>("v_add3_u32", 64, w);
    run<22>("v_and_or_b32", 64, w);
    run<23>("v_perm_b32", 64, w);
    run<24>("v_alignbit_b32", 64, w);
    run<25>("v_or_b32", 64, w);
    run<7>("scan of 32 accumulators, both directions (192)", 192, w);
    run<8>("8 x v_mfma_i32_32x32x32_i8", 8, w);
    run<9>("2 x (8 MFMA spread through the 192-VALU scan)", 400, w);
    run<10>("2 x (MFMAs in pairs, then 48 VALU)", 400, w);
  }
  return 0;
}
