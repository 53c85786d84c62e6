// The reduce-scatter forms of the 64-lane sums (exmc_device.hpp rs64_reduce2 / 4 / 6, rs64_allsum4)
// against group_allsum_n<64, N> in both of its spellings, bit for bit, on random doubles of mixed
// magnitude and sign (so that the order of the additions shows in the last bits), and the lanes the
// totals end up in.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -o allsum_rs_probe allsum_rs_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../exmc_amd/csrc/exmc_device.hpp"

using namespace exmc;

__global__ void k(const double* in, double* out) {   // one wave: in [6][64], out [5][6][64]
  const int l = threadIdx.x;
  double v[6];
  for (int j = 0; j < 6; j++) v[j] = in[j * 64 + l];
  double a[6], b[6];
  for (int j = 0; j < 6; j++) a[j] = b[j] = v[j];
  group_allsum_butterfly_n<64, 6, false>(a);
  group_allsum_butterfly_n<64, 6, true>(b);
  for (int j = 0; j < 6; j++) { out[(0 * 6 + j) * 64 + l] = a[j]; out[(1 * 6 + j) * 64 + l] = b[j]; }
  out[(2 * 6 + 0) * 64 + l] = rs64_reduce6(v);
  const double v4[4] = {v[0], v[1], v[2], v[3]};
  out[(2 * 6 + 1) * 64 + l] = rs64_reduce4(v4);
  const double v2[2] = {v[0], v[1]};
  out[(2 * 6 + 2) * 64 + l] = rs64_reduce2(v2);
  double c[4] = {v[0], v[1], v[2], v[3]};
  rs64_allsum4(c);
  for (int j = 0; j < 4; j++) out[(3 * 6 + j) * 64 + l] = c[j];
}

// 16-lane rows (round 5): row16_reduce_scatter<22> against group_allsum_n<16, 22>, four rows of a wave with
// different data; and rs64_allsum6 against the 64-lane butterfly.   in [22][64], out [22 + 2 + 6][64]
__global__ void k16(const double* in, double* out) {
  const int l = threadIdx.x;
  double v[22], a[22];
  for (int j = 0; j < 22; j++) v[j] = a[j] = in[j * 64 + l];
  group_allsum_n<16, 22>(a);
  for (int j = 0; j < 22; j++) out[j * 64 + l] = a[j];
  double t[2];
  row16_reduce_scatter<22>(v, t);
  out[22 * 64 + l] = t[0];
  out[23 * 64 + l] = t[1];
  double c[6] = {v[0], v[1], v[2], v[3], v[4], v[5]};
  rs64_allsum6(c);
  for (int j = 0; j < 6; j++) out[(24 + j) * 64 + l] = c[j];
}

// the generic 64-lane form (group_allsum_n<64, N> since round 5) against the plain butterfly, N = 22 and 7:
// in [22][64], out [2][22][64] (butterfly, reduce-scatter) then [2][7][64]
__global__ void kgen(const double* in, double* out) {
  const int l = threadIdx.x;
  double a[22], b[22], a7[7], b7[7];
  for (int j = 0; j < 22; j++) a[j] = b[j] = in[j * 64 + l];
  for (int j = 0; j < 7; j++) a7[j] = b7[j] = in[(j + 3) * 64 + l];
  group_allsum_butterfly_n<64, 22>(a);
  group_allsum_n<64, 22>(b);
  group_allsum_butterfly_n<64, 7>(a7);
  group_allsum_n<64, 7>(b7);
  for (int j = 0; j < 22; j++) { out[j * 64 + l] = a[j]; out[(22 + j) * 64 + l] = b[j]; }
  for (int j = 0; j < 7; j++) { out[(44 + j) * 64 + l] = a7[j]; out[(51 + j) * 64 + l] = b7[j]; }
}

int main() {
  double *din, *dout;
  static double in[6 * 64], out[5 * 6 * 64];
  hipMalloc(&din, sizeof in);
  hipMalloc(&dout, sizeof out);
  long bad = 0;
  srand(7);
  for (int trial = 0; trial < 2000; trial++) {
    for (int i = 0; i < 6 * 64; i++) {
      const double m = (double)rand() / RAND_MAX - 0.5;
      const int e = rand() % 40 - 20;
      in[i] = ldexp(m, e);
    }
    hipMemcpy(din, in, sizeof in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    auto at = [&](int form, int j, int lane) { return out[(form * 6 + j) * 64 + lane]; };
    auto same = [](double x, double y) { return memcmp(&x, &y, 8) == 0; };
    static const int lane6[6] = {0, 4, 2, 1, 5, 3};   // total j sits in lane lane6[j]
    static const int lane4[4] = {0, 2, 1, 3};
    for (int j = 0; j < 6; j++) {
      for (int l = 0; l < 64; l++) bad += !same(at(0, j, l), at(1, j, 0));          // the two butterflies, every lane
      bad += !same(at(2, 0, lane6[j]), at(0, j, 0));
      bad += !same(at(2, 0, lane6[j] + 8 * (trial % 8)), at(0, j, 0));             // and in the lanes with the same low bits
    }
    for (int j = 0; j < 4; j++) {
      bad += !same(at(2, 1, lane4[j] + 4 * (trial % 16)), at(0, j, 0));
      for (int l = 0; l < 64; l += 7) bad += !same(at(3, j, l), at(0, j, 0));
    }
    for (int j = 0; j < 2; j++) bad += !same(at(2, 2, j + 2 * (trial % 32)), at(0, j, 0));
  }
  {
    static double in16[22 * 64], out16[30 * 64], ref6[6 * 64];
    double *d16i, *d16o;
    hipMalloc(&d16i, sizeof in16);
    hipMalloc(&d16o, sizeof out16);
    for (int trial = 0; trial < 500; trial++) {
      for (int i = 0; i < 22 * 64; i++) in16[i] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 40 - 20);
      hipMemcpy(d16i, in16, sizeof in16, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, d16i, d16o);
      hipMemcpy(out16, d16o, sizeof out16, hipMemcpyDeviceToHost);
      auto same = [](double x, double y) { return memcmp(&x, &y, 8) == 0; };
      for (int l = 0; l < 64; l++) {
        bad += !same(out16[22 * 64 + l], out16[(l % 16) * 64 + l]);                       // quantity l of this row
        if (l % 16 < 6) bad += !same(out16[23 * 64 + l], out16[(16 + l % 16) * 64 + l]);  // quantity 16 + l
      }
      // rs64_allsum6 against the 64-lane butterfly of the first six quantities (computed on the host
      // copy of the butterfly kernel above: forms 0 of k)
      hipMemcpy(din, in16, 6 * 64 * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
      hipMemcpy(ref6, dout, sizeof ref6, hipMemcpyDeviceToHost);
      for (int j = 0; j < 6; j++)
        for (int l = 0; l < 64; l += 5) bad += !same(out16[(24 + j) * 64 + l], ref6[j * 64 + 0]);
    }
  }
  {
    static double ing[22 * 64], outg[58 * 64];
    double *dgi, *dgo;
    hipMalloc(&dgi, sizeof ing);
    hipMalloc(&dgo, sizeof outg);
    for (int trial = 0; trial < 500; trial++) {
      for (int i = 0; i < 22 * 64; i++) ing[i] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 40 - 20);
      hipMemcpy(dgi, ing, sizeof ing, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(kgen, dim3(1), dim3(64), 0, 0, dgi, dgo);
      hipMemcpy(outg, dgo, sizeof outg, hipMemcpyDeviceToHost);
      for (int j = 0; j < 22; j++)
        for (int l = 0; l < 64; l++) bad += memcmp(&outg[j * 64 + l], &outg[(22 + j) * 64 + l], 8) != 0;
      for (int j = 0; j < 7; j++)
        for (int l = 0; l < 64; l++) bad += memcmp(&outg[(44 + j) * 64 + l], &outg[(51 + j) * 64 + l], 8) != 0;
    }
  }
  printf("allsum_rs_probe: 2000 trials of 6 sums over 64 lanes, 500 of 22 sums over 16-lane rows, 500 of 22 and of 7 sums over 64 lanes (generic form), %ld mismatches\n", bad);
  return bad != 0;
}
