// One wave on one SIMD: cycles per instruction for dependent and independent f64 VALU streams
// (s_memtime around 256 instructions). The NUTS kernels run one wave per SIMD, so whether a
// dependent chain (Horner, Newton-Raphson, butterfly) issues every 4 cycles or waits for the
// pipeline decides whether restructuring for ILP pays.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_latency_probe valu_latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ void k(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0 + 1e-9, d = 0.25;
  double e0 = a, e1 = a + 1, e2 = a + 2, e3 = a + 3;
  long long t0, t1;
  // dependent v_fma_f64
  t0 = clock64();
  asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n\t" "v_fma_f64 %0, %0, %1, %2\n\t" "v_fma_f64 %0, %0, %1, %2\n\t" "v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(c), "v"(d));
  t1 = clock64();
  cyc[0] = t1 - t0;
  // 4 independent chains interleaved
  t0 = clock64();
  asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n\t" "v_fma_f64 %1, %1, %4, %5\n\t" "v_fma_f64 %2, %2, %4, %5\n\t" "v_fma_f64 %3, %3, %4, %5\n\t") : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(c), "v"(d));
  t1 = clock64();
  cyc[1] = t1 - t0;
  // dependent with s_nop 0 between
  t0 = clock64();
  asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n\t s_nop 0\n\t" "v_fma_f64 %0, %0, %1, %2\n\t s_nop 0\n\t" "v_fma_f64 %0, %0, %1, %2\n\t s_nop 0\n\t" "v_fma_f64 %0, %0, %1, %2\n\t s_nop 0\n\t") : "+v"(a) : "v"(c), "v"(d));
  t1 = clock64();
  cyc[2] = t1 - t0;
  // dependent v_add_f64
  t0 = clock64();
  asm volatile(REP64("v_add_f64 %0, %0, %1\n\t" "v_add_f64 %0, %0, %1\n\t" "v_add_f64 %0, %0, %1\n\t" "v_add_f64 %0, %0, %1\n\t") : "+v"(b) : "v"(d));
  t1 = clock64();
  cyc[3] = t1 - t0;
  // 2 independent chains
  t0 = clock64();
  asm volatile(REP64("v_fma_f64 %0, %0, %2, %3\n\t" "v_fma_f64 %1, %1, %2, %3\n\t" "v_fma_f64 %0, %0, %2, %3\n\t" "v_fma_f64 %1, %1, %2, %3\n\t") : "+v"(e0), "+v"(e1) : "v"(c), "v"(d));
  t1 = clock64();
  cyc[4] = t1 - t0;
  // dependent f64 fma alternating with an independent 32-bit op
  int x = threadIdx.x;
  t0 = clock64();
  asm volatile(REP64("v_fma_f64 %0, %0, %2, %3\n\t v_add_u32 %1, %1, %1\n\t" "v_fma_f64 %0, %0, %2, %3\n\t v_add_u32 %1, %1, %1\n\t" "v_fma_f64 %0, %0, %2, %3\n\t v_add_u32 %1, %1, %1\n\t" "v_fma_f64 %0, %0, %2, %3\n\t v_add_u32 %1, %1, %1\n\t") : "+v"(a), "+v"(x) : "v"(c), "v"(d));
  t1 = clock64();
  cyc[5] = t1 - t0;
  // dependent 32-bit integer adds
  t0 = clock64();
  asm volatile(REP64("v_add_u32 %0, %0, %0\n\t" "v_add_u32 %0, %0, %0\n\t" "v_add_u32 %0, %0, %0\n\t" "v_add_u32 %0, %0, %0\n\t") : "+v"(x));
  t1 = clock64();
  cyc[6] = t1 - t0;
  // dependent DPP move + add pairs (one butterfly stage per 3 instructions)
  int y = 0;
  t0 = clock64();
  asm volatile(REP64("v_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t v_add_u32 %0, %0, %1\n\t s_nop 1\n\t") : "+v"(x), "+v"(y));
  t1 = clock64();
  cyc[7] = t1 - t0;
  // dependent v_rcp_f64
  t0 = clock64();
  asm volatile(REP64("v_rcp_f64 %0, %0\n\t") : "+v"(b));
  t1 = clock64();
  cyc[8] = t1 - t0;
  // scalar ALU
  t0 = clock64();
  asm volatile(REP64("s_add_u32 s20, s20, 1\n\t s_add_u32 s20, s20, 1\n\t s_add_u32 s20, s20, 1\n\t s_add_u32 s20, s20, 1\n\t") ::: "s20");
  t1 = clock64();
  cyc[9] = t1 - t0;
  out[threadIdx.x] = a + b + e0 + e1 + e2 + e3 + x;
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
  long long h[16];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[10] = {"v_fma_f64 dependent (256)", "v_fma_f64 4 chains (256)", "v_fma_f64 dep + s_nop 0 (256+256)",
                        "v_add_f64 dependent (256)", "v_fma_f64 2 chains (256)", "v_fma_f64 dep + v_add_u32 (256+256)",
                        "v_add_u32 dependent (256)", "dpp mov + nop + add + nop (64 stages)", "v_rcp_f64 dependent (64)",
                        "s_add_u32 dependent (256)"};
  for (int i = 0; i < 10; i++) printf("%-42s %6lld clocks\n", nm[i], h[i]);
  return 0;
}
