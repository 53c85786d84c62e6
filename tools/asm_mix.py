"""Instruction mix of one kernel in a hipcc -S listing: per opcode and per basic block.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only \
          -o gpurun_out/asm/exmc_hip.s exmc_amd/csrc/exmc_hip.hip
    python tools/asm_mix.py gpurun_out/asm/exmc_hip.s 'nuts_kernelINS_12EightSchoolsILi16' [--blocks]
"""
import collections
import re
import sys


def kernel_lines(path, needle):
    out, on = [], False
    for line in open(path):
        if not on:
            if line.startswith("_Z") and needle in line and line.rstrip().split(":")[0].startswith("_Z"):
                on = True
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            break
        out.append(line.rstrip("\n"))
    return out


def main():
    path, needle = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = kernel_lines(path, needle)
    ops = collections.Counter()
    cls = collections.Counter()
    cur, per_block, order = "entry", collections.Counter(), ["entry"]
    for ln in lines:
        m = re.match(r"^(\.LBB[0-9_]+):", ln)
        if m:
            cur = m.group(1)
            order.append(cur)
            continue
        m = re.match(r"^\t([a-z_0-9]+)\b", ln)
        if not m or ln.startswith("\t."):
            continue
        op = m.group(1)
        ops[op] += 1
        per_block[cur] += 1
        if op.startswith("v_"):
            if "f64" in op:
                cls["valu f64"] += 1
            elif "dpp" in ln or "_dpp" in op:
                cls["valu dpp mov"] += 1
            elif op.startswith(("v_cndmask", "v_mov", "v_accvgpr")):
                cls["valu mov/select"] += 1
            elif op.startswith("v_cmp") or op.startswith("v_cmpx"):
                cls["valu cmp"] += 1
            else:
                cls["valu int/other"] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
        elif op.startswith("ds_"):
            cls["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            cls["vmem"] += 1
        else:
            cls["other"] += 1
    total = sum(ops.values())
    print("instructions: %d" % total)
    for k, v in cls.most_common():
        print("  %-18s %6d  %5.1f%%" % (k, v, 100.0 * v / total))
    print("top opcodes:")
    for k, v in ops.most_common(40):
        print("  %-28s %6d" % (k, v))
    if blocks:
        print("blocks (in layout order):")
        for b in order:
            if per_block[b] >= 8:
                print("  %-14s %5d" % (b, per_block[b]))


if __name__ == "__main__":
    main()
