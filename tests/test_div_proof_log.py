"""The reciprocal-division shortcut of the anomaly kernel: the committed exhaustive run (oracle/proofs/div_by_const.c
over all 2^32 float32 numerators, divisors 1..64) must show what the kernel relies on."""
import os
import re

LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "proofs", "div_by_const_1_64.log")


def test_recip_division_exact_for_odd_and_power_of_two_divisors():
    rows = {}
    for line in open(LOG):
        m = re.match(r"b=\s*(\d+).*normal-result=(\d+)\s+subnormal-result=(\d+)\s+a=inf=(\d+)", line)
        if m:
            rows[int(m.group(1))] = tuple(int(g) for g in m.groups()[1:])
    assert sorted(rows) == list(range(1, 65))
    for b, (normal, sub, inf) in rows.items():
        assert normal == 0  # never wrong when the quotient is a normal number
        assert inf == 2     # +-inf: r = NaN, repaired by v_div_fixup_f32
        if b % 2 == 1 or b & (b - 1) == 0:
            assert sub == 1, b  # the single a = -0.0 case (sign of zero), repaired by v_div_fixup_f32
        else:
            assert sub > 1, b   # even, not a power of two: ties among subnormal quotients -> kernel divides for real
