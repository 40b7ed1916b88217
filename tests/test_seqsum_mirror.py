"""The exact parallel evaluation of the sequential f32 sum of squares (csrc/gl3_seqsum.h) — mirrored step by step in C
(tests/cmirror/seqsum_mirror.c) and run against the plain sequential chain on adversarial inputs (exact ties, wide dynamic
range, one giant element, all zeros, constant vectors, ragged lengths).  Host-only; the GPU twin is tests/test_gpu_seqsum.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def mirror(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ssm") / "seqsum_mirror")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", os.path.join(HERE, "cmirror", "seqsum_mirror.c"), "-o", exe, "-lm"],
                   check=True)
    return exe


@pytest.mark.parametrize("seed", [12345, 7, 2026])
def test_mirror_is_bit_identical_to_the_sequential_chain(mirror, seed):
    out = subprocess.run([mirror, "12000", str(seed)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0" in out.stdout
    # the parallel path must actually be taken (fallbacks would make the test vacuous)
    fallbacks = int(out.stdout.split("fallbacks")[1].split()[0])
    assert fallbacks < 120, out.stdout
