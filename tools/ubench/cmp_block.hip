// cmp_block.hip -- does the hand-scheduled compare block of brick_kernel.h reach the full VALU rate in situ?
// The block of compare_point4 (one staged point against 4 queries: 2 v_mov + 4 x 9 VALU) is run back to back on
// registers only (MODE 0), with the tile's four ds_read_b128 + s_waitcnt lgkmcnt(0) in front of every 4 points as in
// the kernel (MODE 1), and with the operands pinned to chosen VGPR banks (MODE 2: q in banks != the point's).
// hipcc --offload-arch=gfx950 -O3 -o cmp_block cmp_block.hip && ./cmp_block
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PCD_CMP_PAIR(A, B)                                                                         \
  "v_sub_f32 v122, %[qx" #A "], %[px]\n\tv_sub_f32 v125, %[qx" #B "], %[px]\n\t"                       \
  "v_sub_f32 v123, %[qy" #A "], %[py]\n\tv_sub_f32 v126, %[qy" #B "], %[py]\n\t"                       \
  "v_sub_f32 v124, %[qz" #A "], %[pz]\n\tv_sub_f32 v127, %[qz" #B "], %[pz]\n\t"                       \
  "v_mul_f32 v122, v122, v122\n\tv_mul_f32 v125, v125, v125\n\t"                                     \
  "v_mul_f32 v123, v123, v123\n\tv_mul_f32 v126, v126, v126\n\t"                                     \
  "v_mul_f32 v124, v124, v124\n\tv_mul_f32 v127, v127, v127\n\t"                                     \
  "v_add_f32 v122, v122, v123\n\tv_add_f32 v125, v125, v126\n\t"                                     \
  "v_add_f32 v121, v122, v124\n\tv_add_f32 v119, v125, v127\n\t"                                     \
  "v_min_f64 %[b" #A "], %[b" #A "], v[120:121]\n\tv_min_f64 %[b" #B "], %[b" #B "], v[118:119]\n\t"

__device__ __forceinline__ void compare_point4(const f32x4 p, const float* qx, const float* qy, const float* qz, double* best) {
  asm volatile("v_mov_b32 v120, %[pw]\n\tv_mov_b32 v118, %[pw]\n\t"
      PCD_CMP_PAIR(0, 1) PCD_CMP_PAIR(2, 3)
      : [b0] "+v"(best[0]), [b1] "+v"(best[1]), [b2] "+v"(best[2]), [b3] "+v"(best[3])
      : [px] "v"(p.x), [py] "v"(p.y), [pz] "v"(p.z), [pw] "v"(p.w),
        [qx0] "v"(qx[0]), [qy0] "v"(qy[0]), [qz0] "v"(qz[0]), [qx1] "v"(qx[1]), [qy1] "v"(qy[1]), [qz1] "v"(qz[1]),
        [qx2] "v"(qx[2]), [qy2] "v"(qy[2]), [qz2] "v"(qz[2]), [qx3] "v"(qx[3]), [qy3] "v"(qy[3]), [qz3] "v"(qz[3])
      : "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

template <int MODE, int NQ4>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, float s0) {
  __shared__ __attribute__((aligned(16))) float4 tile[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += 256) ((float4*)tile)[i] = make_float4(i * 0.5f, i * 0.25f, i, 0.f);
  __syncthreads();
  float qx[8], qy[8], qz[8];
  double best[8];
  for (int k = 0; k < 8; ++k) { qx[k] = s0 + k; qy[k] = s0 * 2 + k; qz[k] = s0 * 3 + k; best[k] = 1e30 + k; asm volatile("" : "+v"(qx[k]), "+v"(qy[k]), "+v"(qz[k])); }
  f32x4 p[4];
  for (int k = 0; k < 4; ++k) p[k] = f32x4{(float)lane + k, (float)lane * 2, (float)lane * 3, (float)k};
  const unsigned rd = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)&tile[wave][0] + lane * 16;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 1)
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                   "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]) : "v"(rd) : "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      compare_point4(p[k], qx, qy, qz, best);
      if (NQ4 == 2) compare_point4(p[k], qx + 4, qy + 4, qz + 4, best + 4);
    }
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += best[k];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

template <int MODE, int NQ4>
void run(const char* name) {
  const int blocks = 256 * 4;
  float* out; (void)hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 4000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<MODE, NQ4><<<blocks, 256>>>(out, 100, 1.5f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  k<MODE, NQ4><<<blocks, 256>>>(out, iters, 1.5f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  const double valu = (double)iters * 4 * NQ4 * (2 + 36);   // per wave
  const double waves_per_simd = 4;
  printf("%-44s %.3f ms -> %.3f ns per VALU wave-instr per SIMD (%d VALU per 4-point tile)\n", name, ms,
         ms * 1e6 / (valu * waves_per_simd), 4 * NQ4 * 38);
  (void)hipFree(out);
}

int main() {
  run<0, 1>("registers only, 4 queries");
  run<0, 2>("registers only, 8 queries");
  run<1, 1>("ds_read_b128 x4 + wait per tile, 4 queries");
  run<1, 2>("ds_read_b128 x4 + wait per tile, 8 queries");
  return 0;
}
