// exmc_plugin_part.hip — one of the model-dependent kernels of a generated model's plug-in library as a
// translation unit of its own (exmc_amd/codegen.py build_plugin compiles the parts next to
// exmc_hip.hip in parallel processes: a plug-in is then ready in about the time of its slowest
// kernel instead of the sum). EXMC_PLUGIN_PART: 1 nuts_kernel, 2 nuts_kernel (stream form),
// 3 warmup_kernel (two-wave pipeline), 4 warmup_kernel (one wave), 5 the auxiliary kernels (vag_fn
// batches, chain init, step-size search), 6 the one-chain warmup form of a short lane layout, 7 indep_kernel
// (warmup + sampling per chain in one launch: sample_chains vectorized: false), 8 nuts_kernel_wg of a lane layout
// that asks for it (EXMC_GEN_WG). exmc_hip.hip, compiled with
// -DEXMC_PLUGIN_SPLIT, declares the same instantiations `extern template` and keeps everything else.
#include <hip/hip_runtime.h>

#include "exmc_kernels.hpp"

#ifndef EXMC_CUSTOM_HEADER
#error "a plug-in part is built around a generated model (-DEXMC_CUSTOM_HEADER)"
#endif

namespace exmc {

#include "exmc_plugin_kernels.inc"

}  // namespace exmc
