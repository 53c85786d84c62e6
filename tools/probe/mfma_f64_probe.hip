// Probe v_mfma_f64_16x16x4_f64 on gfx950: fragment layout and the rounding/accumulation order of
// D = A*B + C, compared bit for bit with host candidates. Build: hipcc --offload-arch=gfx950
// -ffp-contract=off -o mfma_f64_probe mfma_f64_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* a, const double* b, const double* c, double* out) {
  const int lane = threadIdx.x;
  v4d acc;
  // C/D layout under test: col = lane & 15, row = (lane >> 4) + 4 * reg
  for (int r = 0; r < 4; r++) acc[r] = c[((lane >> 4) + 4 * r) * 16 + (lane & 15)];
  // A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16]
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(lane % 16) * 4 + lane / 16], b[(lane / 16) * 16 + lane % 16], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}
int main() {
  double A[64], B[64], Cm[256], D[256];
  srand(7);
  auto rnd = [] { double m = (rand() / (double)RAND_MAX - 0.5); int e = rand() % 30 - 15; return ldexp(m, e); };
  for (auto& x : A) x = rnd();
  for (auto& x : B) x = rnd();
  for (auto& x : Cm) x = rnd();
  double *da, *db, *dc, *dd;
  hipMalloc(&da, sizeof A); hipMalloc(&db, sizeof B); hipMalloc(&dc, sizeof Cm); hipMalloc(&dd, sizeof D);
  hipMemcpy(da, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(db, B, sizeof B, hipMemcpyHostToDevice);
  hipMemcpy(dc, Cm, sizeof Cm, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
  hipMemcpy(D, dd, sizeof D, hipMemcpyDeviceToHost);
  int ok_fwd = 0, ok_rev = 0, ok_mul = 0, ok_pair = 0, ok_exact = 0;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      double f = Cm[i * 16 + j], r = Cm[i * 16 + j], m = Cm[i * 16 + j];
      for (int kk = 0; kk < 4; kk++) f = fma(A[i * 4 + kk], B[kk * 16 + j], f);
      for (int kk = 3; kk >= 0; kk--) r = fma(A[i * 4 + kk], B[kk * 16 + j], r);
      for (int kk = 0; kk < 4; kk++) m = m + A[i * 4 + kk] * B[kk * 16 + j];
      double p = fma(A[i * 4 + 0], B[0 * 16 + j], fma(A[i * 4 + 1], B[1 * 16 + j], 0.0)) +
                 fma(A[i * 4 + 2], B[2 * 16 + j], fma(A[i * 4 + 3], B[3 * 16 + j], 0.0));
      p = p + Cm[i * 16 + j];
      long double e = Cm[i * 16 + j];
      for (int kk = 0; kk < 4; kk++) e += (long double)A[i * 4 + kk] * (long double)B[kk * 16 + j];
      double got = D[i * 16 + j];
      ok_fwd += !memcmp(&got, &f, 8); ok_rev += !memcmp(&got, &r, 8); ok_mul += !memcmp(&got, &m, 8);
      ok_pair += !memcmp(&got, &p, 8);
      double ed = (double)e; ok_exact += !memcmp(&got, &ed, 8);
    }
  printf("of 256: fma-chain k=0..3 %d, k=3..0 %d, mul+add %d, pairwise %d, long-double-exact %d\n",
         ok_fwd, ok_rev, ok_mul, ok_pair, ok_exact);
  return 0;
}
