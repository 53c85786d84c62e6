"""Pins the checker's OTP :rand restatement (exsss seeding, uniform_s, the normal_s ziggurat) to a
real BEAM -- when a maintainer has produced tests/golden/otp_rand_vectors.txt with
`elixir tools/otp_rand_vectors.exs` (this pipeline has no Erlang/OTP: SURVEY.md 8c).
Without the file the test is skipped and draw-level parity with the reference stays unpinned."""
import ctypes as C
import os
import struct

import pytest

import oracle as O

PATH = os.path.join(os.path.dirname(__file__), "golden", "otp_rand_vectors.txt")


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


@pytest.mark.skipif(not os.path.exists(PATH), reason="no BEAM-generated vectors committed "
                    "(run `elixir tools/otp_rand_vectors.exs > tests/golden/otp_rand_vectors.txt`)")
def test_checker_rng_equals_otp_rand():
    by_seed = {}
    for line in open(PATH):
        parts = line.split()
        if len(parts) != 4 or parts[0].startswith("#"):
            continue
        seed, kind, idx, hexbits = int(parts[0]), parts[1], int(parts[2]), int(parts[3], 16)
        by_seed.setdefault(seed, []).append((kind, idx, hexbits))
    assert by_seed, "the vector file has no rows"
    L = O.lib()
    for seed, rows in by_seed.items():
        r = O.Rng()
        L.exo_rng_seed(C.byref(r), seed)
        for kind, idx, want in rows:          # file order = stream order
            got = L.exo_rng_uniform(C.byref(r)) if kind == "uniform" else L.exo_rng_normal(C.byref(r), 0)
            assert _bits(got) == want, (seed, kind, idx, got)


def test_the_script_and_the_test_agree_on_the_row_format():
    """The generator's row format is what the test parses (guards against drift between the two
    files while no BEAM is at hand)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "otp_rand_vectors.exs")).read()
    assert 'IO.puts("#{seed} uniform #{i} #{bits.(u)}")' in src
    assert 'IO.puts("#{seed} normal #{i} #{bits.(z)}")' in src
    assert ":rand.seed_s(:exsss, seed)" in src
