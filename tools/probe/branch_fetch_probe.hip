// What does a taken branch cost a lone wave, and does it depend on the CU it runs on?
// One single-wave workgroup per launch (as the warmup kernel), 48 launches in a row so that the
// dispatcher walks over CUs; each launch reports its XCC / CU / SE and the cycles of
//   (a) 256 x 16 straight-line VALU instructions (no branch),
//   (b) the same work with a taken s_branch to the very next instruction after every 16,
//   (c) the same work with a taken s_branch over 2 KB of code after every 16 (all 512 KB... no:
//       32 targets x 2 KB = 64 KB footprint, larger than one instruction-cache way set),
//   (d) as (c) with a 256-byte stride (8 KB footprint, cache resident).
// Build: hipcc --offload-arch=gfx950 -O3 -o branch_fetch_probe branch_fetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)
#define WORK "v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t" \
             "v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t" \
             "v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t" \
             "v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t v_fma_f64 %0, %0, %1, %2\n\t"
// 2 KB of never-executed filler: 512 s_nop (4 bytes each)
#define PAD2K REP32(REP16("s_nop 0\n\t"))
#define PAD256 REP16(REP4("s_nop 0\n\t"))

__global__ void k(double* out, long long* rec, double seed, int slot) {
  double a = seed + threadIdx.x;
  const double c = 1.0 + 1e-9, d = 0.25;
  long long t[5];
  for (int pass = 0; pass < 3; pass++) {   // the last pass is reported (instruction cache warm)
    t[0] = clock64();
    for (int i = 0; i < 8; i++) asm volatile(REP32(WORK) : "+v"(a) : "v"(c), "v"(d));
    t[1] = clock64();
    for (int i = 0; i < 8; i++) asm volatile(REP32(WORK "s_branch 1f\n\t1:\n\t") : "+v"(a) : "v"(c), "v"(d));
    t[2] = clock64();
    for (int i = 0; i < 8; i++) asm volatile(REP32(WORK "s_branch 1f\n\t" PAD2K "1:\n\t") : "+v"(a) : "v"(c), "v"(d));
    t[3] = clock64();
    for (int i = 0; i < 8; i++) asm volatile(REP32(WORK "s_branch 1f\n\t" PAD256 "1:\n\t") : "+v"(a) : "v"(c), "v"(d));
    t[4] = clock64();
  }
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    rec[slot * 8 + 0] = (xcc & 0xf) * 1000 + ((hw >> 8) & 0xf) * 10 + ((hw >> 13) & 0x7);
    for (int i = 0; i < 4; i++) rec[slot * 8 + 1 + i] = t[i + 1] - t[i];
  }
  out[threadIdx.x] = a;
}

// one loop only, 4096 rounds, for counter collection (rocprofv3 --pmc SQ_WAIT_ANY ...): the kernels
// are told apart by their grid size (mode + 1 workgroups)
__global__ void k1(double* out, double seed, int mode) {
  double a = seed + threadIdx.x;
  const double c = 1.0 + 1e-9, d = 0.25;
  if (mode == 0) for (int i = 0; i < 4096; i++) asm volatile(REP32(WORK) : "+v"(a) : "v"(c), "v"(d));
  if (mode == 1) for (int i = 0; i < 4096; i++) asm volatile(REP32(WORK "s_branch 1f\n\t1:\n\t") : "+v"(a) : "v"(c), "v"(d));
  if (mode == 2) for (int i = 0; i < 4096; i++) asm volatile(REP32(WORK "s_branch 1f\n\t" PAD256 "1:\n\t") : "+v"(a) : "v"(c), "v"(d));
  out[threadIdx.x] = a;
}

int main(int argc, char** argv) {
  if (argc > 1) {
    double* o; (void)hipMalloc(&o, 64 * 8);
    for (int mode = 0; mode < 3; mode++) hipLaunchKernelGGL(k1, dim3(mode + 1), dim3(64), 0, 0, o, 1.0, mode);
    (void)hipDeviceSynchronize();
    return 0;
  }
  double* out; long long* rec;
  const int N = 48;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&rec, N * 8 * 8);
  for (int s = 0; s < N; s++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, rec, 1.0, s);
    (void)hipDeviceSynchronize();
  }
  long long h[N * 8];
  (void)hipMemcpy(h, rec, sizeof h, hipMemcpyDeviceToHost);
  printf("256 groups of 16 v_fma_f64; extra clocks per taken branch relative to (a)\n");
  printf("%-8s %10s %12s %12s %12s\n", "xcc/cu/se", "(a) clocks", "near", "far 2KB", "far 256B");
  for (int s = 0; s < N; s++) {
    const double a = (double)h[s * 8 + 1];
    printf("%-8lld %10.0f %12.1f %12.1f %12.1f\n", h[s * 8], a, (h[s * 8 + 2] - a) / 256.0,
           (h[s * 8 + 3] - a) / 256.0, (h[s * 8 + 4] - a) / 256.0);
  }
  return 0;
}
