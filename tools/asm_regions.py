"""Per-block table and per-region opcode histograms of one kernel in a `hipcc -S --cuda-device-only`
listing (tools/asm_mix.py gives the whole-kernel mix; this one answers "what is in the leaf loop").

    python tools/asm_regions.py <listing.s> <kernel needle> [--blocks] [LABEL_A:LABEL_B[:title] ...]

A region LABEL_A:LABEL_B is the text from basic-block label LABEL_A (e.g. .LBB8_85) up to, not
including, LABEL_B in layout order. Static counts: what a pass executes depends on its branches.
"""
import collections
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
import asm_mix  # noqa: E402


def main():
    path, needle = sys.argv[1], sys.argv[2]
    lines = asm_mix.kernel_lines(path, needle)
    label_at = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB[0-9_]+):", ln)
        if m:
            label_at[m.group(1)] = i
    if "--blocks" in sys.argv:
        cur, start, c = "entry", 0, collections.Counter()
        rows = []
        for i, ln in enumerate(lines + [".LBB_end:"]):
            m = re.match(r"^(\.LBB[0-9_a-z]+):", ln)
            if m:
                rows.append((cur, start, c))
                cur, start, c = m.group(1), i, collections.Counter()
                continue
            m = re.match(r"^\t([a-z_0-9]+)\b", ln)
            if m and not ln.startswith("\t."):
                c[m.group(1)] += 1
        print("%-12s %6s %5s %5s %5s %4s %4s %4s %4s" % ("block", "line", "instr", "valu", "f64am", "rdln", "wrln", "scld", "scst"))
        for name, st, c in rows:
            n = sum(c.values())
            if n < 6:
                continue
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            arith = sum(v for k, v in c.items() if re.match(r"v_(add|mul|fma|fmac)_f64", k))
            print("%-12s %6d %5d %5d %5d %4d %4d %4d %4d" % (name, st, n, valu, arith, c["v_readlane_b32"], c["v_writelane_b32"],
                                                         c["scratch_load_dwordx2"], c["scratch_store_dwordx2"]))
    for spec in sys.argv[3:]:
        if spec.startswith("--"):
            continue
        parts = spec.split(":")
        a, b = label_at[parts[0]], label_at[parts[1]]
        title = parts[2] if len(parts) > 2 else spec
        c = collections.Counter()
        for ln in lines[a:b]:
            m = re.match(r"^\t([a-z_0-9]+)\b", ln)
            if m and not ln.startswith("\t."):
                c[m.group(1)] += 1
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        arith = sum(v for k, v in c.items() if re.match(r"v_(add|mul|fma|fmac)_f64", k))
        print("== %s (%s .. %s): %d instructions, %d VALU, %d of them f64 add / mul / fma (%.0f %%)"
              % (title, parts[0], parts[1], tot, valu, arith, 100.0 * arith / max(valu, 1)))
        print("   " + ", ".join("%s %d" % kv for kv in c.most_common(40)))


if __name__ == "__main__":
    main()
