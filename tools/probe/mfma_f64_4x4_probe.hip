// Probe v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks per instruction) on gfx950:
//   (1) which lanes' A and B operands feed which lane's output (one-hot A against numbered B),
//   (2) the rounding / accumulation order of the four products of an output (bit for bit against
//       host candidates),
//   (3) what it costs: clocks per instruction for a dependent chain, for independent ones, and for a
//       stream that alternates it with independent v_fma_f64 (does the vector pipe run beside it).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_f64_4x4_probe mfma_f64_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void onehot(double* out) {   // out[e][l]: A one-hot at lane e, B[l] = l + 1, C = 0
  const int lane = threadIdx.x;
  for (int e = 0; e < 64; e++) {
    const double a = (lane == e) ? 1.0 : 0.0;
    const double b = (double)(lane + 1);
    out[e * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  }
}
__global__ void rnd(const double* a, const double* b, const double* c, double* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], c[lane], 0, 0, 0);
}
__global__ void timing(long long* t, double* sink) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0, v0 = 1.0, v1 = 2.0, v2 = 3.0, v3 = 4.0;
  constexpr int N = 256;
  long long t0 = clock64();
  for (int i = 0; i < N; i++) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);   // dependent chain
  long long t1 = clock64();
  for (int i = 0; i < N; i += 4) {                                                           // four chains
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
  }
  long long t2 = clock64();
  for (int i = 0; i < N; i += 4) {                                                           // + 4 fma each
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
  }
  long long t3 = clock64();
  for (int i = 0; i < N; i += 4) {                                                           // the fma alone
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
    v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
  }
  long long t4 = clock64();
  if (lane == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = N; }
  sink[blockIdx.x * 64 + lane] = c0 + c1 + c2 + c3 + v0 + v1 + v2 + v3;
}

int main() {
  double *dout, *da, *db, *dc, *dd, *sink;
  long long* dt;
  hipMalloc(&dout, 64 * 64 * 8);
  hipLaunchKernelGGL(onehot, dim3(1), dim3(64), 0, 0, dout);
  static double O[64 * 64];
  hipMemcpy(O, dout, sizeof O, hipMemcpyDeviceToHost);
  // pa[l][n], pb[l][n]: the A lane and B lane of the n-th product of output lane l, by ascending A lane
  int pa[64][8], pb[64][8], np_[64];
  memset(np_, 0, sizeof np_);
  for (int e = 0; e < 64; e++)
    for (int l = 0; l < 64; l++)
      if (O[e * 64 + l] != 0.0 && np_[l] < 8) { pa[l][np_[l]] = e; pb[l][np_[l]] = (int)O[e * 64 + l] - 1; np_[l]++; }
  printf("lane: (A lane x B lane) products, by ascending A lane\n");
  for (int l = 0; l < 64; l++) {
    printf("%2d:", l);
    for (int n = 0; n < np_[l]; n++) printf(" (%2d x %2d)", pa[l][n], pb[l][n]);
    printf("\n");
  }
  // hypothesis: A[b][i][k] at lane 16k + 4b + i, B[b][k][j] at lane 16k + 4b + j, D[b][i][j] at lane 16i + 4b + j
  int hyp = 0;
  for (int l = 0; l < 64; l++) {
    const int i = l / 16, b = (l / 4) % 4, j = l % 4;
    int ok = np_[l] == 4;
    for (int k = 0; k < 4 && ok; k++) ok = pa[l][k] == 16 * k + 4 * b + i && pb[l][k] == 16 * k + 4 * b + j;
    hyp += ok;
  }
  printf("layout hypothesis A[b][i][k]@16k+4b+i, B[b][k][j]@16k+4b+j, D[b][i][j]@16i+4b+j: %d of 64 lanes\n", hyp);
  // (2) rounding order
  double A[64], B[64], Cm[64], D[64];
  srand(11);
  auto rndv = [] { double m = (rand() / (double)RAND_MAX - 0.5); int e = rand() % 30 - 15; return ldexp(m, e); };
  int fwd = 0, rev = 0, muladd = 0, tot = 0;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 512);
  for (int rep = 0; rep < 64; rep++) {
    for (int l = 0; l < 64; l++) { A[l] = rndv(); B[l] = rndv(); Cm[l] = rndv(); }
    hipMemcpy(da, A, 512, hipMemcpyHostToDevice); hipMemcpy(db, B, 512, hipMemcpyHostToDevice);
    hipMemcpy(dc, Cm, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(rnd, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    hipMemcpy(D, dd, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
      double f = Cm[l], r = Cm[l], m = Cm[l];
      for (int n = 0; n < np_[l]; n++) f = fma(A[pa[l][n]], B[pb[l][n]], f);
      for (int n = np_[l] - 1; n >= 0; n--) r = fma(A[pa[l][n]], B[pb[l][n]], r);
      for (int n = 0; n < np_[l]; n++) m = m + A[pa[l][n]] * B[pb[l][n]];
      fwd += !memcmp(&D[l], &f, 8); rev += !memcmp(&D[l], &r, 8); muladd += !memcmp(&D[l], &m, 8); tot++;
    }
  }
  printf("of %d outputs: fma chain from C in ascending A lane %d, descending %d, mul+add %d\n", tot, fwd, rev, muladd);
  // (3) cost
  hipMalloc(&dt, 64); hipMalloc(&sink, 1024 * 64 * 8);
  for (int waves = 1; waves <= 2; waves++) {
    // one workgroup per SIMD is not controllable from here; one wave (grid 1) and a full chip of 2 waves per SIMD
    const int grid = waves == 1 ? 1 : 2048;
    hipLaunchKernelGGL(timing, dim3(grid), dim3(64), 0, 0, dt, sink);
    long long T[8];
    hipMemcpy(T, dt, 40, hipMemcpyDeviceToHost);
    printf("grid %4d: clocks per MFMA: dependent chain %.1f, four chains %.1f; MFMA + 4 v_fma_f64: %.1f per group; 4 v_fma_f64 alone: %.1f per group\n",
           grid, (double)T[0] / T[4], (double)T[1] / T[4], (double)T[2] / T[4], (double)T[3] / T[4]);
  }
  return 0;
}
