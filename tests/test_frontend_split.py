"""The host layer's drop-in split: host/frontend.h + frontend.cc carry ONLY the hot path's classes —
Device, Frame, Feature, FastDetector, ORBDetector, ImageAlign, Matcher, FeatureAlign — written against the reference tree's own
Camera / Point / Map / Config / SE3 (host/frontend_deps.h), and must link without the rest of the host layer.  CPU checks:
  * the symbols frontend.o leaves undefined are the C-ABI of include/sdvl_hip.h, the C / C++ runtime, and exactly six members of the
    reference's Camera and Point — nothing of SDVL, SDVLBatch, the map stand-ins or the C API of the Python bindings;
  * `make frontend_link_check` links frontend.cc against host/minimal_deps.cc, a second, independent implementation of those six
    members, without standalone.cc / mapper.cc / capi.cc, and the program runs (without a GPU it says so; the GPU suite runs it for
    real: tests/test_gpu_tracker.py);
  * types.h's real-Eigen / real-OpenCV branch compiles: the whole layer against mock <Eigen/Dense> and <opencv2/core.hpp> headers
    (tests/mock_third_party; neither library exists in this image)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "slam-sdvl_amd", "host")
FLAGS = ["-std=c++17", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-pthread"]

DEPS = {  # the reference's own members (camera.cc:69-79, point.cc:102-142, point.h:60) frontend.cc calls and does not define
    "sdvl::Camera::Project(sdvl::Vec<double, 3> const&, sdvl::Vec<double, 2>*) const",
    "sdvl::Camera::Unproject(sdvl::Vec<double, 2> const&, sdvl::Vec<double, 3>*) const",
    "sdvl::Point::GetPosition() const",
    "sdvl::Point::GetStd()",
    "sdvl::Point::Promote()",
    "sdvl::Point::Unpromote()",
}


def undefined_symbols(tmp_path):
    obj = os.path.join(str(tmp_path), "frontend.o")
    subprocess.run(["g++", "-O1"] + FLAGS + ["-c", os.path.join(HOST, "frontend.cc"), "-o", obj], check=True, cwd=HOST)
    out = subprocess.run(["nm", "-uC", obj], check=True, capture_output=True, text=True).stdout
    return [l.split(None, 1)[1].strip() for l in out.splitlines() if l.strip().startswith("U ")]


def test_frontend_object_depends_on_six_reference_members_only(tmp_path):
    und = undefined_symbols(tmp_path)
    ours = {s for s in und if s.startswith("sdvl::")}
    assert ours == DEPS, sorted(ours ^ DEPS)
    # C-ABI calls: every one is declared in include/sdvl_hip.h (the product boundary), none comes from the Python bindings' C API
    header = open(os.path.join(ROOT, "include", "sdvl_hip.h")).read()
    cabi = {s for s in und if s.startswith("sdvl_")}
    assert cabi and not [s for s in cabi if s.startswith("sdvlh_")]
    missing = [s for s in cabi if not re.search(r"\b%s\s*\(" % re.escape(s), header)]
    assert not missing, missing
    # nothing else of this repository: the rest is the C / C++ runtime
    rest = [s for s in und if not s.startswith(("sdvl::", "sdvl_"))]
    assert not [s for s in rest if "sdvl" in s.lower() and not s.startswith(("std::", "typeinfo", "vtable", "operator"))], rest
    # and frontend.cc includes the front end's header only
    src = open(os.path.join(HOST, "frontend.cc")).read()
    assert '#include "sdvl_host.h"' not in src and '#include "host_internal.h"' in src
    assert '#include "frontend.h"' in open(os.path.join(HOST, "host_internal.h")).read()


def test_frontend_links_against_a_second_implementation_of_its_dependencies():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "slam-sdvl_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", HOST, "frontend_link_check"], check=True)
    rule = open(os.path.join(HOST, "Makefile")).read().split("frontend_link_check:", 1)[1].split("\n\n", 1)[0]
    for absent in ("standalone.cc", "mapper.cc", "capi.cc"):
        assert absent not in rule, absent
    for present in ("frontend.cc", "minimal_deps.cc"):
        assert present in rule
    minimal = open(os.path.join(HOST, "minimal_deps.cc")).read()
    for name in ("Camera::Project", "Camera::Unproject", "Point::GetPosition", "Point::GetStd", "Point::Promote", "Point::Unpromote"):
        assert name + "(" in minimal, name
    import torch
    if not torch.cuda.is_available():          # on the GPU box the GPU suite runs the program and checks its results
        out = subprocess.run([os.path.join(HOST, "frontend_link_check")], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "linked against minimal_deps.cc" in out.stdout, out.stdout + out.stderr


def test_types_h_compiles_against_eigen_and_opencv_headers():
    mock = os.path.join(ROOT, "tests", "mock_third_party")
    for src in (os.path.join(mock, "types_with_third_party.cc"), os.path.join(HOST, "frontend.cc"), os.path.join(HOST, "standalone.cc"),
                os.path.join(HOST, "mapper.cc")):
        out = subprocess.run(["g++", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter"] + FLAGS + ["-I" + mock, "-I" + HOST, src],
                             capture_output=True, text=True, cwd=HOST)
        assert out.returncode == 0, out.stderr[-3000:]
    # without the mocks on the include path the same file must refuse (the branch under test is the one with the real types)
    out = subprocess.run(["g++", "-fsyntax-only"] + FLAGS + ["-I" + HOST, os.path.join(mock, "types_with_third_party.cc")], capture_output=True, text=True)
    assert out.returncode != 0 and "were not picked up" in out.stderr
