// exmc_common.hip — the model-independent kernels (Diagnostics.ess / rhat / rank scores of
// exmc_kernels.hpp, the tree kernels of exmc_native_tree.hpp) as an object of their own. Built once
// next to libexmc_hip.so (exmc_amd/build.py) and linked into every generated model's plug-in
// library, whose own translation units only declare these kernels (EXMC_COMMON_DECL_ONLY): a plug-in
// build does not compile them again (exmc_amd/codegen.py build_plugin).
#include <hip/hip_runtime.h>

#include "exmc_kernels.hpp"
#include "exmc_native_tree.hpp"
