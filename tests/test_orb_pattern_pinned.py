"""The one piece of reference DATA the product and the oracle share: the 256 x 4 ORB sampling table
(`/root/reference/extra/orb_detector.cc:62-319`), carried as `orb_pattern_31.inc` on both sides.  Both sides read the same text,
so a wrong integer would be invisible to every parity test — here the table is parsed out of the reference source again, by a
parser of its own, and compared integer by integer; the circular-patch row ends `umax_` (orb_detector.cc:325-348) are
recomputed from the reference's formula and compared with the constants the kernels carry.  Runs only where /root/reference
exists (the build container); the GPU box has no reference tree and skips it."""
import math
import os
import re

import pytest

REF = "/root/reference/extra/orb_detector.cc"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isfile(REF), reason="reference tree not present (GPU box)")


def inc_numbers(path):
    text = "\n".join(l for l in open(path).read().splitlines() if not l.lstrip().startswith("//"))
    return [int(t) for t in text.replace("\n", " ").split(",") if t.strip()]


def reference_table():
    lines = open(REF).read().splitlines()
    start = next(i for i, l in enumerate(lines) if "bit_pattern_31_[256*4]" in l)
    nums = []
    for l in lines[start + 1:]:
        if l.strip().startswith("};"):
            break
        body = l.split("/*")[0]          # every row ends in a /*mean (...), correlation (...)*/ comment
        nums += [int(t) for t in body.split(",") if t.strip()]
    return start + 1, nums


def test_orb_table_equals_the_reference_source():
    first_line, ref = reference_table()
    assert first_line == 62 and len(ref) == 1024          # declared at orb_detector.cc:62, closed at :319: 256 rows of {x0, y0, x1, y1}
    assert all(-15 <= v <= 15 for v in ref)                # inside the 31 x 31 patch
    prod = inc_numbers(os.path.join(ROOT, "slam-sdvl_amd", "csrc", "orb_pattern_31.inc"))
    orac = inc_numbers(os.path.join(ROOT, "oracle", "orb_pattern_31.inc"))
    assert prod == ref
    assert orac == ref


def cv_round(x):
    """cvRound = round half to even"""
    f = math.floor(x)
    d = x - f
    if d > 0.5 or (d == 0.5 and f % 2 == 1):
        return int(f) + 1
    return int(f)


def test_umax_equals_the_reference_formula():
    src = open(REF).read()
    # the statements being restated are the ones the reference has (orb_detector.cc:325-348)
    for stmt in ("int HALF_PATCH_SIZE = Config::ORBSize()/2;", "vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);",
                 "int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);", "umax_[v] = cvRound(sqrt(hp2 - v * v));",
                 "while (umax_[v0] == umax_[v0 + 1])", "umax_[v] = v0;"):
        assert stmt in src, stmt
    half = 31 // 2                                         # ORBSize 31 (config.cc:84)
    s2 = float.fromhex("0x1.6a09e6p+0")                    # sqrt(2.f) as a float
    vmax = math.floor(half * s2 / 2 + 1)
    vmin = math.ceil(half * s2 / 2)
    hp2 = float(half * half)
    umax = [0] * (half + 1)
    for v in range(vmax + 1):
        umax[v] = cv_round(math.sqrt(hp2 - v * v))
    v0 = 0
    for v in range(half, vmin - 1, -1):
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    dev = open(os.path.join(ROOT, "slam-sdvl_amd", "csrc", "sdvl_orb_device.h")).read()
    m = re.search(r"constexpr int umax\[16\] = \{([^}]*)\};", dev)
    assert m, "the kernels' umax table moved"
    assert [int(t) for t in m.group(1).split(",")] == umax
    assert umax == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
