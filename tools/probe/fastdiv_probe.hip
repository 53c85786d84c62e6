// Checks the short f64 division (exmc_detmath.h: exmc_rcp_refined + exmc_div_core) against the
// compiler's full IEEE expansion of `/` on the device, bit for bit, over the operand ranges the
// kernels claim for it. Prints one line per distribution and exits non-zero on any mismatch.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -o fastdiv_probe fastdiv_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "exmc_detmath.h"

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}

// sign, exponent in [elo, ehi], random 52-bit mantissa
__device__ __forceinline__ double make(uint64_t bits, int elo, int ehi) {
  const uint64_t mant = bits & 0x000FFFFFFFFFFFFFULL;
  const uint64_t sign = (bits >> 63) << 63;
  const uint64_t e = (uint64_t)(elo + (int)((bits >> 52) % (uint64_t)(ehi - elo + 1)) + 1023);
  return exmc_from_bits(sign | (e << 52) | mant);
}

__global__ void k(int mode, uint64_t seed, unsigned long long* bad, double* first) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nbad = 0;
  for (int it = 0; it < 256; it++) {
    const uint64_t r1 = mix(seed ^ (id * 256 + it) * 2), r2 = mix(r1 ^ 0x1234567);
    double a, b;
    if (mode == 0) {            // both operands anywhere in the declared window
      a = make(r1, -380, 379);
      b = make(r2, -380, 379);
    } else if (mode == 1) {     // model constants as divisors: moderate b, wide a
      a = make(r1, -380, 379);
      b = make(r2, -4, 8);
    } else if (mode == 2) {     // log: f / (2 + f), f = m - 1, m in [sqrt(2)/2, sqrt(2))
      const double m = exmc_from_bits(0x3FE6A09E667F3BCDULL + (r1 % 0x000FFFFFFFFFFFFFULL));
      a = m - 1.0;
      b = 2.0 + a;
    } else if (mode == 3) {     // eight_schools' last quotient: a in e^+-200, b = 1 + (a*5/2)^2-like
      a = make(r1, -291, 291);
      const double zt = 2.5 * a;
      b = 1.0 + zt * zt;
    } else {                    // near-1 mantissas and exact powers of two (rounding edge cases)
      const uint64_t tiny = r1 & 0xFF;
      a = exmc_from_bits(((uint64_t)(1023 + (int)(r2 % 200) - 100) << 52) | tiny | ((r1 >> 63) << 63));
      b = exmc_from_bits(((uint64_t)(1023 + (int)((r2 >> 20) % 200) - 100) << 52) |
                         ((r2 >> 40) & 1 ? 0x000FFFFFFFFFFFFFULL - (r1 >> 8 & 0xFF) : (r1 >> 8 & 0xFF)));
    }
    const double fast = exmc_div_core(a, b, exmc_rcp_refined(b));
    const double ref = a / b;
    if (exmc_to_bits(fast) != exmc_to_bits(ref)) {
      if (nbad == 0 && atomicAdd(bad + 1, 1ULL) == 0) { first[0] = a; first[1] = b; first[2] = fast; first[3] = ref; }
      nbad++;
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned long long* bad; double* first;
  if (hipMalloc(&bad, 16) != hipSuccess || hipMalloc(&first, 32) != hipSuccess) { printf("no device\n"); return 2; }
  const char* names[5] = {"window x window", "wide / moderate constant", "log: f / (2 + f)",
                          "a / (1 + (2.5 a)^2), |a| in 2^+-291", "mantissa edges, powers of two"};
  int rc = 0;
  for (int mode = 0; mode < 5; mode++) {
    unsigned long long h[2] = {0, 0};
    double f[4];
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, mode, 0x5eedULL + mode, bad, first);
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
    printf("%-40s %llu quotients, %llu mismatches\n", names[mode], 4096ULL * 256 * 256, h[0]);
    if (h[0]) { printf("   first: %a / %a -> fast %a, div %a\n", f[0], f[1], f[2], f[3]); rc = 1; }
  }
  return rc;
}
