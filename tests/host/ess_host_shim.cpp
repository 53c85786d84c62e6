// Host build of the device's per-series ESS routine (exmc_amd/csrc/exmc_ess.hpp) for
// tests/test_ess_series_host.py. Test infrastructure only.
#include "../../exmc_amd/csrc/exmc_ess.hpp"

extern "C" double ess_series_host(const double* x, long stride, int S) {
  return exmc::ess_series(x, (size_t)stride, S);
}
