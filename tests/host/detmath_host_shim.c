/* Host build of the range-restricted exp / log variants of include/exmc_detmath.h for
 * tests/test_detmath_ranges.py (compiled with -ffp-contract=off). Test infrastructure only. */
#include "../../include/exmc_detmath.h"

double h_exp(double x) { return exmc_exp(x); }
double h_log(double x) { return exmc_log(x); }
double h_exp_pm200(double x) { return exmc_exp_pm200(x); }
double h_exp_le0(double x) { return exmc_exp_le0(x); }
double h_log_ge1(double x) { return exmc_log_ge1(x); }
double h_log_unit(double x) { return exmc_log_unit(x); }
/* n values at once: out[i] = f_which(x[i]); returns the number of values whose bits differ from the
 * general function */
long h_compare(int which, const double* x, long n) {
  long bad = 0;
  for (long i = 0; i < n; i++) {
    double a, b;
    switch (which) {
      case 0: a = exmc_exp(x[i]); b = exmc_exp_pm200(x[i]); break;
      case 1: a = exmc_exp(x[i]); b = exmc_exp_le0(x[i]); break;
      case 2: a = exmc_log(x[i]); b = exmc_log_ge1(x[i]); break;
      default: a = exmc_log(x[i]); b = exmc_log_unit(x[i]); break;
    }
    bad += __builtin_memcmp(&a, &b, 8) != 0;
  }
  return bad;
}
