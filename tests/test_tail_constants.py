"""CPU: the band constants of the list threshold kernel fit together (csrc/marex_tails.hip: straggler passes shift the band
by TT_STEP bins; with TT_STEP > TT_BW - 8 a quantile bin between two tried bands is never found and the passes never end --
the build asserts it, this test reads the same constants from the source so a change shows up without a compiler)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _defines(path):
    src = open(path).read()
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define\s+(TT_[A-Z_]+)\s+(\d+)", src, re.M)}, src


def test_straggler_bands_overlap_and_the_assert_is_in_the_source():
    d, src = _defines(os.path.join(ROOT, "marex_amd", "csrc", "marex_tails.hip"))
    assert d["TT_STEP"] <= d["TT_BW"] - 8
    assert 2 * d["TT_MARGIN"] < d["TT_BW"] and d["TT_BW"] + 2 <= 2 * d["TT_LS"]   # levels 0 .. BW + 1 fit the lane column
    assert "static_assert(TT_STEP <= TT_BW - 8" in src
    assert "n_unresolved" in src and "pass_limit" in src                          # the limit is counted, not silent
