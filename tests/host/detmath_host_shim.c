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

/* the table-driven logarithm (round 6): values, and its error against the long double logarithm
 * (64-bit significand on x86-64) in units of the last place of the correctly rounded result, and
 * absolutely -- out[0] = worst ulp error over arguments with |log x| >= 0.02, out[1] = worst absolute
 * error over the rest (the neighbourhood of 1, where l_i + log1p(r) cancels), out[2] = worst ulp error there */
#include <math.h>
double h_log_tab(double x) { return exmc_log_tab(x); }
void h_log_tab_error(const double* x, long n, double* out) {
  out[0] = out[1] = out[2] = 0.0;
  for (long i = 0; i < n; i++) {
    const double a = exmc_log_tab(x[i]);
    const long double t = logl((long double)x[i]);
    const double b = (double)t;
    if (b == 0.0) {
      if (a != 0.0) out[0] = out[2] = 1e300;
      continue;
    }
    const double ulp = fabs(nextafter(b, INFINITY) - b);
    const double ea = (double)fabsl((long double)a - t), eu = ea / ulp;
    if (fabs(b) >= 0.02) {
      if (eu > out[0]) out[0] = eu;
    } else {
      if (ea > out[1]) out[1] = ea;
      if (eu > out[2]) out[2] = eu;
    }
  }
}
