#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every kernel in a built library, from the code
object's metadata notes:  python tools/kernel_notes.py exmc_amd/lib/libexmc_hip.so [filter]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def notes(so):
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "co")
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + so,
                               "--output=" + co], stderr=subprocess.DEVNULL) if False else None
        # the fat binary sits in the .hip_fatbin section of the shared object
        fat = os.path.join(td, "fat")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary",
                               "--only-section=.hip_fatbin", so, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                               "--output=" + co])
        return subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)


def main():
    so = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    txt = notes(so)
    demangle = "c++filt"
    for blk in txt.split("  - .agpr_count:")[1:]:
        get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]   # noqa: E731
        name = subprocess.check_output([demangle, get("name")], text=True).strip()
        if flt and flt not in name:
            continue
        print("%s\n    vgpr %s agpr %s sgpr %s  vgpr_spill %s sgpr_spill %s  scratch %s B  lds %s B" % (
            name[:150], get("vgpr_count"), blk.split()[0], get("sgpr_count"), get("vgpr_spill_count"),
            get("sgpr_spill_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
