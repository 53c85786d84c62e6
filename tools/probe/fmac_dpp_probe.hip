// v_fmac_f64_dpp with row_newbcast (the only DPP control the f64 ALU of gfx950 takes): does
// `acc = fma(x[lane i of the row], 1.0, acc)` for i = 0..9 give the left-to-right sum of ten lanes in
// every lane of the row, bit for bit, and what does it cost a lone wave per instruction?
// Build: hipcc --offload-arch=gfx950 -O3 -o fmac_dpp_probe fmac_dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ long long now() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

#define FM(i) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #i " row_mask:0xf bank_mask:0xf\n\t"
#define SUM10 "s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" \
              FM(1) FM(2) FM(3) FM(4) FM(5) FM(6) FM(7) FM(8) FM(9)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ void k(const double* in, double* out, long long* cyc) {
  const double x = in[threadIdx.x];
  double one = 1.0, acc;
  asm volatile("" : "+v"(one));
  asm volatile(SUM10 : "=&v"(acc) : "v"(x), "v"(one));
  out[threadIdx.x] = acc;
  double a2 = acc;
  long long t0 = now();
  asm volatile(REP64(SUM10) : "+&v"(a2) : "v"(x), "v"(one) : "memory");
  long long t1 = now();
  // the butterfly it would replace: 4 stages of 2 v_mov_b32_dpp + v_add_f64, timed the same way
  double b = x;
  int blo = __double2loint(b), bhi = __double2hiint(b);
  long long t2 = now();
#define ST(ctrl) "v_mov_b32_dpp %2, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 " ctrl \
                 " row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\tv_add_f64 %0, %0, %0\n\t"
  int lo, hi;
  asm volatile(REP64(ST("quad_perm:[1,0,3,2]") ST("quad_perm:[2,3,0,1]") ST("row_half_mirror") ST("row_mirror"))
               : "+v"(b), "=&v"(lo), "=&v"(hi) : "v"(blo), "v"(bhi) : "memory");
  long long t3 = now();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
  out[64 + threadIdx.x] = a2 + b + lo + hi;
}

int main() {
  double h[64], *din, *dout; long long* cyc;
  for (int i = 0; i < 64; i++) h[i] = 1.0 / (3.0 + i) + 1e-9 * i * i;
  (void)hipMalloc(&din, 64 * 8); (void)hipMalloc(&dout, 128 * 8); (void)hipMalloc(&cyc, 16);
  (void)hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, cyc);
  double o[64]; long long c[2];
  (void)hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int row = 0; row < 4; row++) {
    double s = h[row * 16];
    for (int i = 1; i < 10; i++) s = s + h[row * 16 + i];
    for (int l = 0; l < 16; l++) bad += std::memcmp(&s, &o[row * 16 + l], 8) != 0;
  }
  printf("fmac_dpp sum of 10 lanes: %s (%d lanes differ)\n", bad ? "MISMATCH" : "bit-exact left-to-right sum in every lane", bad);
  printf("sum10 by v_fmac_f64_dpp   : %6.1f clocks per sum (11 instructions incl. s_nop)\n", (double)c[0] / 64.0);
  printf("4-stage DPP butterfly     : %6.1f clocks per sum (16 instructions incl. s_nop)\n", (double)c[1] / 64.0);
  return bad != 0;
}
