"""Build libexmc_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "exmc_hip.hip")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libexmc_hip.so")
# the model-independent kernels as an object of their own, linked into every plug-in library of a
# generated model (exmc_amd/codegen.py build_plugin) instead of being compiled again with each
COMMON_SRC = os.path.join(HERE, "csrc", "exmc_common.hip")
COMMON_OBJ = os.path.join(OUT_DIR, "exmc_common.o")

DEPS = [
    SRC,
    os.path.join(HERE, "csrc", "exmc_kernels.hpp"),
    os.path.join(HERE, "csrc", "exmc_nuts.hpp"),
    os.path.join(HERE, "csrc", "exmc_native_tree.hpp"),
    os.path.join(HERE, "csrc", "exmc_models.hpp"),
    os.path.join(HERE, "csrc", "exmc_device.hpp"),
    os.path.join(HERE, "csrc", "exmc_ess.hpp"),
    os.path.join(HERE, "csrc", "exmc_plugin_part.hip"),
    COMMON_SRC,
    os.path.join(HERE, "csrc", "exmc_plugin_kernels.inc"),
    os.path.join(HERE, "csrc", "exmc_plugin_layouts.inc"),
    os.path.join(ROOT, "include", "exmc_hip.h"),
    os.path.join(ROOT, "include", "exmc_detmath.h"),
    os.path.join(ROOT, "include", "exmc_logtab.h"),
    os.path.join(ROOT, "include", "exmc_zig_tables.h"),
]

# -ffp-contract=off is part of the numeric contract (include/exmc_detmath.h): fma only where written.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libexmc_hip.so cannot be built (no CPU fallback exists)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build_common(force=False, verbose=False):
    """exmc_common.o (see COMMON_SRC); atomic, so that parallel builders never link a partial file."""
    if not force and os.path.exists(COMMON_OBJ):
        t = os.path.getmtime(COMMON_OBJ)
        if all(os.path.getmtime(d) <= t for d in DEPS):
            return COMMON_OBJ
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = "%s.%d.tmp" % (COMMON_OBJ, os.getpid())
    cmd = [hipcc()] + [f for f in FLAGS if f != "-shared"] + ["-c", "-o", tmp, COMMON_SRC]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd, cwd=os.path.join(HERE, "csrc"))
        os.replace(tmp, COMMON_OBJ)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return COMMON_OBJ


def build(force=False, verbose=False):
    if not force and up_to_date():
        build_common(verbose=verbose)
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    extra = os.environ.get("EXMC_EXTRA_FLAGS", "").split()
    cmd = [hipcc()] + FLAGS + extra + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=os.path.join(HERE, "csrc"))
    build_common(force=True, verbose=verbose)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
