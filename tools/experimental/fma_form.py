"""EXPERIMENT (round 5, measured, no gain -- kept as a tool, not part of the build): f64 additions and products in
their fma form, by a rewrite of the device assembly.

    PYTHONPATH=. python tools/experimental/fma_form.py <out.so> [-DEXMC_DEV_ES16_ONLY ...]

eight_schools 4096 x 1000 with 11 011 additions and 10 211 products rewritten: 14.03-14.10 ms against 13.97-14.10 ms,
same bits -- the clock a lone wave saves per instruction in the probe is hidden in the real instruction mix.

A wave that has its SIMD to itself (eight_schools 4096 x 16 lanes, radon 1024 x 64, every warmup chain)
pays 5.0 clocks for v_add_f64 / v_mul_f64 and 4.05 for v_fma_f64 (tools/probe/valu_rate_probe.hip,
fma_form_probe.hip). a + b IS fma(a, 1.0, b) and a * b IS fma(a, b, -0.0), bit for bit on every operand
pair (signed zeros, infinities, denormals: fma_form_probe checks 65 536 pairs), but the compiler folds
those back to an addition and a product. So the library is built in two halves: the device side to
assembly (`hipcc --cuda-device-only -S`), every

    v_add_f64 D, A, B   ->   v_fma_f64 D, A, 1.0, B
    v_mul_f64 D, A, B   ->   v_fma_f64 D, A, B, neg(0)

(source modifiers and scalar / inline operands carry over: an inline constant does not use the constant bus;
both forms are 64-bit VOP3 encodings, so no offset moves), then assembler, device link, offload bundle and
the host side with that bundle embedded -- the steps hipcc itself runs for `-shared`. The numeric contract
(include/exmc_detmath.h) is untouched: the same values in the same order, one rounding each."""
import os
import re
import shutil
import subprocess

_ADD = re.compile(r"^(\s*)v_add_f64(?:_e64)? ([^,]+), ([^,]+), ([^,\n]+?)\s*$")
_MUL = re.compile(r"^(\s*)v_mul_f64(?:_e64)? ([^,]+), ([^,]+), ([^,\n]+?)\s*$")


def rewrite(text):
    """-> (text with the two rewrites applied, number of additions, number of products rewritten)"""
    out, na, nm = [], 0, 0
    for ln in text.split("\n"):
        m = _ADD.match(ln)
        if m:
            out.append("%sv_fma_f64 %s, %s, 1.0, %s" % m.groups())
            na += 1
            continue
        m = _MUL.match(ln)
        if m:
            out.append("%sv_fma_f64 %s, %s, %s, neg(0)" % m.groups())
            nm += 1
            continue
        if re.match(r"^\s*v_(add|mul)_f64", ln):
            raise ValueError("an f64 add / mul this rewrite does not know: %r" % ln)   # modifiers, DPP ...
        out.append(ln)
    return "\n".join(out), na, nm


def llvm_bin(hipcc):
    """the LLVM tool directory of the ROCm install hipcc belongs to"""
    for cand in (os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin"),
                 "/opt/rocm/lib/llvm/bin"):
        if os.path.exists(os.path.join(cand, "clang-offload-bundler")):
            return os.path.realpath(cand)
    raise RuntimeError("clang-offload-bundler not found next to %s" % hipcc)


def build_shared(hipcc, flags, src, out, cwd, arch="gfx950", verbose=False, keep=None):
    """`hipcc <flags> -shared -o out src` with the device assembly rewritten on the way. flags: the compile
    flags WITHOUT -shared / -o. Raises on any failing step (the caller falls back to the plain build).
    keep: a directory to leave the intermediate files in."""
    import tempfile
    L = llvm_bin(hipcc)
    out, src = os.path.abspath(out), os.path.abspath(src)
    work = keep or tempfile.mkdtemp(prefix="exmc_fma_")
    os.makedirs(work, exist_ok=True)
    base = os.path.join(work, "dev")

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=cwd)

    try:
        run([hipcc] + flags + ["--cuda-device-only", "-S", "-o", base + ".s", src])
        text, na, nm = rewrite(open(base + ".s").read())
        open(base + ".fma.s", "w").write(text)
        run([os.path.join(L, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + arch, "-c",
             base + ".fma.s", "-o", base + ".o"])
        run([os.path.join(L, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o",
             base + ".out", base + ".o"])
        run([os.path.join(L, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--" + arch, "-input=/dev/null",
             "-input=" + base + ".out", "-output=" + base + ".hipfb"])
        run([hipcc] + flags + ["--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary", "-Xclang",
                               base + ".hipfb", "-o", base + ".host.o"])
        tmp = "%s.%d.tmp" % (out, os.getpid())
        run([hipcc, "-shared", base + ".host.o", "-o", tmp])
        os.replace(tmp, out)
        return na, nm
    finally:
        if keep is None:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    import sys
    from exmc_amd import build as B
    out = sys.argv[1]
    extra = sys.argv[2:]
    flags = [f for f in B.FLAGS if f != "-shared"] + extra
    print(build_shared(B.hipcc(), flags, B.SRC, out, os.path.join(B.HERE, "csrc"), verbose=True))
