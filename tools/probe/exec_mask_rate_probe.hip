// Does a wave with a quarter of its lanes active issue f64 VALU work faster, and how do several
// waves on one SIMD share the f64 pipe? One workgroup of W waves (wave i lands on SIMD i % 4), each
// wave runs 1024 v_fma_f64 / v_add_f64 / a VALU+SALU mix with all 64 lanes or with lanes 0..15 only;
// prints clocks per instruction per wave for W = 4, 8, 16 (1, 2, 4 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_mask_rate_probe exec_mask_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))
#define REP1024(x) REP4(REP256(x))

// s_memtime with its own wait, fenced: the compiler may otherwise move clock reads across the
// volatile asm blocks (the first version of this probe reported 0 clocks for its first block)
__device__ __forceinline__ long long now() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

__global__ void k(double* out, long long* cyc, double seed, int quarter) {
  double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0 + 1e-9, d = 0.25;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long t[4] = {0, 0, 0, 0};
  __syncthreads();
  if (!quarter || lane < 16) {
    __builtin_amdgcn_s_barrier();
    long long t0 = now();
    asm volatile(REP1024("v_fma_f64 %0, %1, %2, %3\n\t") : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "memory");
    long long t1 = now();
    t[0] = t1 - t0;
    __builtin_amdgcn_s_barrier();
    t0 = now();
    asm volatile(REP1024("v_add_f64 %0, %1, %2\n\t") : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "memory");
    t1 = now();
    t[1] = t1 - t0;
    __builtin_amdgcn_s_barrier();
    t0 = now();
    // 2 VALU : 1 SALU : (1 not-taken branch per 4) -- roughly the NUTS pass mix
    asm volatile(REP256("v_fma_f64 %0, %1, %2, %3\n\t s_mov_b32 s20, s21\n\t v_add_f64 %1, %0, %2\n\t"
                        "v_mul_f64 %0, %1, %2\n\t s_and_b32 s22, s20, s21\n\t v_fma_f64 %1, %0, %2, %3\n\t"
                        "s_cbranch_execz 1f\n\t1:\n\t")
                 : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "s20", "s21", "s22", "memory");
    t1 = now();
    t[2] = t1 - t0;
    __builtin_amdgcn_s_barrier();
    t0 = now();
    int xi = lane + 3, yi = 7;
    asm volatile(REP1024("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                 : "+v"(xi), "+v"(yi) :: "memory");
    a += xi + yi;
    t1 = now();
    t[3] = t1 - t0;
  }
  if (lane == 0)
    for (int i = 0; i < 4; i++) cyc[wave * 4 + i] = t[i];
  out[threadIdx.x] = a + b;
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&cyc, 16 * 4 * 8);
  const double n[4] = {1024, 1024, 256 * 7, 1024};
  const char* nm[4] = {"v_fma_f64", "v_add_f64", "mix 4 VALU + 2 SALU + 1 branch", "v_mov_b32_dpp"};
  for (int quarter = 0; quarter < 2; quarter++)
    for (int W = 4; W <= 16; W *= 2) {
      for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(64 * W), 0, 0, out, cyc, 1.0, quarter);
      long long h[64];
      (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
      for (int i = 0; i < 4; i++) {
        double worst = 0;
        for (int w = 0; w < W; w++) worst = h[w * 4 + i] > worst ? (double)h[w * 4 + i] : worst;
        printf("%s lanes, %d waves/SIMD  %-32s %6.2f clocks/instr/wave  (%.2f per SIMD)\n",
               quarter ? "16" : "64", W / 4, nm[i], worst / n[i], worst / n[i] / (W / 4));
      }
    }
  return 0;
}
