// a + b as v_fma_f64 a, 1.0, b and a * b as v_fma_f64 a, b, neg(0): bit-identical to v_add_f64 / v_mul_f64 (signed
// zeros, infinities, NaN payload classes, denormals, random operands), and the clocks a lone wave pays for each form.
//   hipcc --offload-arch=gfx950 -O3 -o fma_form_probe fma_form_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

__global__ void k(const double* a, const double* b, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b[i];
  double s0, s1, p0, p1, d0, d1;
  asm volatile("v_add_f64 %0, %1, %2" : "=v"(s0) : "v"(x), "v"(y));
  asm volatile("v_fma_f64 %0, %1, 1.0, %2" : "=v"(s1) : "v"(x), "v"(y));
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p0) : "v"(x), "v"(y));
  asm volatile("v_fma_f64 %0, %1, %2, neg(0)" : "=v"(p1) : "v"(x), "v"(y));
  asm volatile("v_add_f64 %0, %1, -%2" : "=v"(d0) : "v"(x), "v"(y));
  asm volatile("v_fma_f64 %0, %2, -1.0, %1" : "=v"(d1) : "v"(x), "v"(y));
  out[6 * i + 0] = s0; out[6 * i + 1] = s1; out[6 * i + 2] = p0; out[6 * i + 3] = p1; out[6 * i + 4] = d0; out[6 * i + 5] = d1;
}

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP256(x) REP16(REP16(x))
__global__ void t(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0 + 1e-9;
  long long t0, t1;
#define TIME(idx, body) t0 = clock64(); asm volatile(REP256(body) : "+v"(a) : "v"(b), "v"(c)); t1 = clock64(); cyc[idx] = t1 - t0;
  TIME(0, "v_add_f64 %0, %1, %2\n\t")
  TIME(1, "v_fma_f64 %0, %1, 1.0, %2\n\t")
  TIME(2, "v_mul_f64 %0, %1, %2\n\t")
  TIME(3, "v_fma_f64 %0, %1, %2, neg(0)\n\t")
  TIME(4, "v_fma_f64 %0, %1, %2, %0\n\t")
  out[threadIdx.x] = a;
}

int main() {
  const int n = 1 << 16;
  static double a[n], b[n], out[6 * n];
  const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 4.9e-324, -4.9e-324, 2.2250738585072014e-308,
                       1.7976931348623157e308, -1.7976931348623157e308, 0.5, 3.0, 1e-300, 1e300};
  const int ns = sizeof sp / sizeof sp[0];
  srand(3);
  for (int i = 0; i < n; i++) {
    if (i < ns * ns) { a[i] = sp[i / ns]; b[i] = sp[i % ns]; }
    else { a[i] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 600 - 300); b[i] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 600 - 300); }
  }
  double *da, *db, *dout; long long* dc;
  hipMalloc(&da, sizeof a); hipMalloc(&db, sizeof b); hipMalloc(&dout, sizeof out); hipMalloc(&dc, 64);
  hipMemcpy(da, a, sizeof a, hipMemcpyHostToDevice); hipMemcpy(db, b, sizeof b, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
  hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
  long bad[3] = {0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) {
      const double u = out[6 * i + 2 * j], v = out[6 * i + 2 * j + 1];
      const bool same = memcmp(&u, &v, 8) == 0 || (u != u && v != v);   // NaNs: both NaN (payloads may differ)
      if (!same) { if (bad[j] < 5) printf("form %d: a=%a b=%a plain %a fma-form %a\n", j, a[i], b[i], u, v); bad[j]++; }
    }
  printf("fma_form_probe: %d operand pairs (%d special x special): add %ld, mul %ld, sub %ld mismatches\n", n, ns * ns, bad[0], bad[1], bad[2]);
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL(t, dim3(1), dim3(64), 0, 0, dout, dc, 1.0);
  long long h[8];
  hipMemcpy(h, dc, 40, hipMemcpyDeviceToHost);
  const char* nm[5] = {"v_add_f64", "v_fma_f64 a, 1.0, b", "v_mul_f64", "v_fma_f64 a, b, neg(0)", "v_fma_f64 a, b, c"};
  for (int i = 0; i < 5; i++) printf("  %-26s %.2f clocks/instr (lone wave)\n", nm[i], (double)h[i] / 256.0);
  return (bad[0] + bad[1] + bad[2]) != 0;
}
