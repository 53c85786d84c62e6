// Device spellings of include/exmc_detmath.h against the host build of the same header, bit for
// bit: exmc_exp / exmc_log (compiler spelling), exmc_exp_v / exmc_log_v (asm polynomial cores) and
// the range-restricted variants (v_ldexp_f64 scaling, no special-case branches), each over its
// domain including the subnormal results of exp and every special value. Exits non-zero on a
// mismatch. Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -o detmath_probe detmath_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "exmc_detmath.h"

// exmc_log_tab on the device: the pair from a global copy of the table (the kernels also keep one in LDS)
static __device__ const double logtab_dev[2 * EXMC_LOGTAB_ENTRIES] = {EXMC_LOGTAB_VALUES};
static __device__ double log_tab_dev(double x) {
  double m;
  int e;
  const int i = exmc_logtab_split(x, &m, &e);
  return exmc_logtab_finish(m, e, logtab_dev[2 * i], logtab_dev[2 * i + 1]);
}

__global__ void k(const double* x, int n, int which, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r;
  switch (which) {
    case 0: r = exmc_exp(v); break;
    case 1: r = exmc_exp_v(v); break;
    case 2: r = exmc_log(v); break;
    case 3: r = exmc_log_v(v); break;
    case 4: r = exmc_exp_pm200(v); break;
    case 5: r = exmc_exp_pm200_v(v); break;
    case 6: r = exmc_exp_le0(v); break;
    case 7: r = exmc_exp_le0_v(v); break;
    case 8: r = exmc_log_ge1(v); break;
    case 9: r = exmc_log_ge1_v(v); break;
    case 10: r = exmc_log_unit(v); break;
    case 11: r = exmc_log_unit_v(v); break;
    default: r = log_tab_dev(v); break;
  }
  out[i] = r;
}

static uint64_t st = 88172645463325252ULL;
static double rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; }

int main() {
  const int N = 1 << 20;
  std::vector<double> any(N), pm200(N), le0(N), ge1(N), unit(N), pos(N), normal(N);
  for (int i = 0; i < N; i++) {
    any[i] = (rnd() * 2 - 1) * ((i & 3) == 0 ? 760.0 : 30.0);
    pm200[i] = (rnd() * 2 - 1) * 200.0;
    le0[i] = (i & 1) ? -rnd() * 800.0 : (-745.2 + rnd() * 1.5);   // half of them in the subnormal-result band
    ge1[i] = (i & 1) ? 1.0 + rnd() : std::exp(rnd() * 709.0);
    unit[i] = std::floor(rnd() * 9007199254740992.0) / 9007199254740992.0;
    pos[i] = std::exp((rnd() * 2 - 1) * 700.0);
    // normal positive arguments: clipped probabilities, the neighbourhood of 1, every binade
    normal[i] = (i & 3) == 0 ? 1e-7 + rnd() * (1.0 - 2e-7) : ((i & 3) == 1 ? 1.0 + (rnd() - 0.5) * 0.05 : std::exp((rnd() * 2 - 1) * 700.0));
  }
  normal[0] = 1.0; normal[1] = 2.2250738585072014e-308; normal[2] = 1.7976931348623157e308; normal[3] = 1.0 - 0x1p-53;
  normal[4] = 1.0 + 0x1p-52; normal[5] = 0.70710678118654752; normal[6] = 1.4142135623730951;
  const double sp_exp[] = {0.0, -0.0, -INFINITY, INFINITY, NAN, 709.782712893384, 709.79, -745.1332191019412, -745.14, -746.0, -1e300, 1e300};
  const double sp_log[] = {0.0, -0.0, -1.0, INFINITY, NAN, 1.0, 5e-324, 2.2250738585072014e-308, 1e-310, 1.7976931348623157e308};
  for (unsigned i = 0; i < sizeof sp_exp / 8; i++) any[i] = sp_exp[i];
  for (unsigned i = 0; i < sizeof sp_log / 8; i++) pos[i] = sp_log[i];
  const double sp_le0[] = {0.0, -0.0, -INFINITY, NAN, -745.1332191019412, -745.14, -746.0, -1e300, -5e-324};
  for (unsigned i = 0; i < sizeof sp_le0 / 8; i++) le0[i] = sp_le0[i];
  const double sp_ge1[] = {1.0, 2.0, NAN, 1.7976931348623157e308, 1.0000000000000002};
  for (unsigned i = 0; i < sizeof sp_ge1 / 8; i++) ge1[i] = sp_ge1[i];
  unit[0] = 0.0; unit[1] = 0x1p-53; unit[2] = 1.0 - 0x1p-53;
  pm200[0] = 200.0; pm200[1] = -200.0; pm200[2] = 0.0;

  struct Case { const char* name; int which; std::vector<double>* x; double (*host)(double); };
  Case cases[] = {
      {"exmc_exp", 0, &any, exmc_exp}, {"exmc_exp_v", 1, &any, exmc_exp},
      {"exmc_log", 2, &pos, exmc_log}, {"exmc_log_v", 3, &pos, exmc_log},
      {"exmc_exp_pm200", 4, &pm200, exmc_exp}, {"exmc_exp_pm200_v", 5, &pm200, exmc_exp},
      {"exmc_exp_le0", 6, &le0, exmc_exp}, {"exmc_exp_le0_v", 7, &le0, exmc_exp},
      {"exmc_log_ge1", 8, &ge1, exmc_log}, {"exmc_log_ge1_v", 9, &ge1, exmc_log},
      {"exmc_log_unit", 10, &unit, exmc_log}, {"exmc_log_unit_v", 11, &unit, exmc_log},
      {"exmc_log_tab", 12, &normal, exmc_log_tab}};
  double *dx, *dout;
  (void)hipMalloc(&dx, N * 8); (void)hipMalloc(&dout, N * 8);
  std::vector<double> out(N);
  long total_bad = 0;
  for (auto& c : cases) {
    (void)hipMemcpy(dx, c.x->data(), N * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, 0, dx, N, c.which, dout);
    if (hipMemcpy(out.data(), dout, N * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
    long bad = 0;
    for (int i = 0; i < N; i++) {
      const double h = c.host((*c.x)[i]);
      const bool both_nan = (h != h) && (out[i] != out[i]);   // NaN payloads need not agree
      if (!both_nan && std::memcmp(&h, &out[i], 8) != 0) {
        if (bad < 3) printf("  %s(%.17g): host %.17g device %.17g\n", c.name, (*c.x)[i], h, out[i]);
        bad++;
      }
    }
    printf("%-18s %d values, %ld mismatches\n", c.name, N, bad);
    total_bad += bad;
  }
  return total_bad != 0;
}
