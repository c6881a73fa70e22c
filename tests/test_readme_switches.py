"""The README's table of environment switches lists exactly the SDVL_* names the library reads (csrc/ and host/): a switch that is added
or removed without its row — or a row whose test no longer exists — fails here, on the CPU."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_readme_lists_every_library_switch_and_names_an_existing_test():
    read = set()
    for pat in ("slam-sdvl_amd/csrc/*.hip", "slam-sdvl_amd/csrc/*.h", "slam-sdvl_amd/host/*.cc", "slam-sdvl_amd/host/*.h"):
        for path in glob.glob(os.path.join(ROOT, pat)):
            if os.path.basename(path) in ("track_sequence.cc", "api_surface_check.cc", "threaded_mode_check.cc", "frontend_link_check.cc"):
                continue   # the example programs' own options are not library switches
            read |= set(re.findall(r'getenv\("(SDVL_[A-Z0-9_]+)"\)', open(path).read()))
    readme = open(os.path.join(ROOT, "README.md")).read()
    rows = re.findall(r"^\| `(SDVL_[A-Z0-9_]+)=[^`]*` \|[^|]*\| ([^|]*) \|$", readme, re.M)
    listed = {name for name, _ in rows}
    assert listed == read, (sorted(listed - read), sorted(read - listed))
    m = re.search(r"Environment switches of the library\*\* \((\w+);", readme)
    words = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve", "thirteen", "fourteen", "fifteen"]
    assert m and m.group(1) == words[len(read)], (m and m.group(1), len(read))
    assert len(read) <= 15
    tests_src = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "tests", "*.py")))
    for name, cell in rows:
        for t in re.findall(r"`(test_[a-z0-9_]+)", cell):
            assert "def " + t in tests_src, (name, t)
