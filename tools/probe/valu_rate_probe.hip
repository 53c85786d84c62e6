// Issue cost of individual gfx950 instructions for a lone wave (clocks per instruction over 256
// independent-ish issues). Guides which operations to avoid in the one-wave-per-SIMD NUTS loop.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

#define TIME(idx, body)                      \
  t0 = clock64();                            \
  asm volatile(REP256(body) : "+v"(a), "+v"(b), "+v"(x), "+v"(y), "+v"(u), "+v"(w) : "v"(c), "v"(d)); \
  t1 = clock64();                            \
  cyc[idx] = t1 - t0;

__global__ void k(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0 + 1e-9, d = 0.25;
  int x = threadIdx.x + 3, y = 7;
  unsigned long long u = 0x123456789abcdefULL + threadIdx.x, w = 77;
  long long t0, t1;
  TIME(0, "v_fma_f64 %0, %1, %6, %7\n\t")
  TIME(1, "v_mul_f64 %0, %1, %6\n\t")
  TIME(2, "v_add_f64 %0, %1, %6\n\t")
  TIME(3, "v_rndne_f64 %0, %1\n\t")
  TIME(4, "v_cvt_i32_f64 %2, %1\n\t")
  TIME(5, "v_cvt_f64_i32 %0, %3\n\t")
  TIME(6, "v_cvt_f64_u32 %0, %3\n\t")
  TIME(7, "v_ldexp_f64 %0, %1, %3\n\t")
  TIME(8, "v_rcp_f64 %0, %1\n\t")
  TIME(9, "v_div_scale_f64 %0, vcc, %1, %6, %1\n\t")
  TIME(10, "v_div_fmas_f64 %0, %1, %6, %7\n\t")
  TIME(11, "v_div_fixup_f64 %0, %1, %6, %7\n\t")
  TIME(12, "v_max_f64 %0, %1, %6\n\t")
  TIME(13, "v_cmp_lt_f64 vcc, %1, %6\n\t")
  TIME(14, "v_mad_u64_u32 %4, vcc, %2, %3, %5\n\t")
  TIME(15, "v_mul_lo_u32 %2, %3, %3\n\t")
  TIME(16, "v_mul_hi_u32 %2, %3, %3\n\t")
  TIME(17, "v_lshlrev_b64 %4, 7, %5\n\t")
  TIME(18, "v_lshl_add_u64 %4, %5, 3, %5\n\t")
  TIME(19, "v_cndmask_b32 %2, %3, %3, vcc\n\t")
  TIME(20, "v_mov_b32 %2, %3\n\t")
  TIME(21, "v_mov_b64 %0, %1\n\t")
  TIME(22, "v_readlane_b32 s20, %3, 3\n\t")
  TIME(23, "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
  TIME(24, "v_frexp_mant_f64 %0, %1\n\t")
  TIME(25, "v_frexp_exp_i32_f64 %2, %1\n\t")
  TIME(26, "v_sqrt_f64 %0, %1\n\t")
  TIME(27, "v_fma_f32 %2, %3, %3, %3\n\t")
  TIME(28, "s_mov_b64 s[20:21], exec\n\t")
  TIME(29, "v_xor_b32 %2, %3, %3\n\t")
  // branches: 256 taken s_branch to the next instruction / 256 not-taken s_cbranch_execz /
  // 256 s_cbranch_execz taken over one skipped instruction (exec = 0 inside a saveexec region)
  t0 = clock64();
  asm volatile(REP256("s_branch 1f\n\t1:\n\t") ::: "memory");
  t1 = clock64();
  cyc[30] = t1 - t0;
  t0 = clock64();
  asm volatile(REP256("s_cbranch_execz 1f\n\t1:\n\t") ::: "memory");
  t1 = clock64();
  cyc[31] = t1 - t0;
  t0 = clock64();
  asm volatile("s_mov_b64 s[22:23], exec\n\t s_mov_b64 exec, 0\n\t"
               REP256("s_cbranch_execz 1f\n\t v_mov_b32 %0, %0\n\t1:\n\t")
               "s_mov_b64 exec, s[22:23]\n\t" : "+v"(x) :: "s22", "s23", "memory");
  t1 = clock64();
  cyc[32] = t1 - t0;
  out[threadIdx.x] = a + b + x + y + (double)u + (double)w;
}

int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 40 * 8);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.0);
  long long h[40];
  (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[30] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_cvt_f64_i32",
                        "v_cvt_f64_u32", "v_ldexp_f64", "v_rcp_f64", "v_div_scale_f64", "v_div_fmas_f64",
                        "v_div_fixup_f64", "v_max_f64", "v_cmp_lt_f64", "v_mad_u64_u32", "v_mul_lo_u32",
                        "v_mul_hi_u32", "v_lshlrev_b64", "v_lshl_add_u64", "v_cndmask_b32", "v_mov_b32",
                        "v_mov_b64", "v_readlane_b32", "v_mov_b32_dpp", "v_frexp_mant_f64",
                        "v_frexp_exp_i32_f64", "v_sqrt_f64", "v_fma_f32", "s_mov_b64", "v_xor_b32"};
  for (int i = 0; i < 30; i++) printf("%-22s %6.2f clocks/instr\n", nm[i], (double)h[i] / 256.0);
  printf("%-34s %6.2f clocks each\n", "s_branch taken (to next instr)", (double)h[30] / 256.0);
  printf("%-34s %6.2f clocks each\n", "s_cbranch_execz not taken", (double)h[31] / 256.0);
  printf("%-34s %6.2f clocks each\n", "s_cbranch_execz taken (skip 1)", (double)h[32] / 256.0);
  return 0;
}
