"""No kernel of the product library may spill registers to scratch memory.

Builds every HIP source of ``revisiting-at_amd/csrc`` for gfx950 with ``-Rpass-analysis=kernel-resource-usage`` (device pass only,
the Makefile's flags; hipcc cross-compiles without a GPU, ~1 minute) and fails on ``ScratchSize > 0``.  Round 2 shipped three
depthwise-7x7 instantiations of the headline configuration with 88-92 bytes per lane of scratch after a launch-bound change made
for another configuration; nothing noticed.  A spill reload is a memory load on the same counter as the prefetches these kernels
are built around, so it is a correctness-of-the-design check, not a style check.
"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import resource_usage as RU  # noqa: E402


@pytest.fixture(scope="module")
def table():
    if not (os.path.exists(RU.HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    return RU.collect(jobs=6)


def test_every_source_was_analysed(table):
    files = {r["file"] for r in table}
    assert files == {os.path.basename(f) for f in RU.sources()}
    assert len(table) > 150                                   # the library has a few hundred kernel instantiations


def test_no_kernel_uses_scratch(table):
    bad = [(r["file"], r["name"], r["scratch"], r.get("vgpr_spill", 0)) for r in table if r.get("scratch", 0) > 0]
    assert not bad, "kernels with scratch (file, kernel, bytes/lane, spilled VGPRs):\n" + "\n".join(map(str, bad))


def test_no_vector_register_spills(table):
    # (scalar registers parked in spare VGPR lanes - "SGPRs Spill" - never touch memory and are not counted)
    bad = [(r["file"], r["name"], r["vgpr_spill"]) for r in table if r.get("vgpr_spill", 0)]
    assert not bad, f"kernels with spilled vector registers: {bad}"


def test_default_path_occupancy_of_the_rolling_depthwise_kernels(table):
    """The 256-thread builds serve maps up to 64 pixels wide (every ConvNeXt-T / -B / ViT shape); only the wide build of the
    65 ... 80-pixel maps carries the larger launch bound, and it stages one unit per thread."""
    roll = [r for r in table if "dwconv7x7_roll_kernel<" in r["name"]]
    assert roll
    for r in roll:
        assert r["scratch"] == 0
        if ", 384>" in r["name"]:
            assert ", 1, " in r["name"], f"wide rolling kernel with two staging units: {r['name']}"
