"""Runs the C++ host-mirror test (tests/cpp/test_module.cpp over include/rcflow_module.hpp),
built by __graft_entry__.build()."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_host_mirror_frame_loop():
    exe = os.path.join(ROOT, "tests", "cpp", "test_module")
    assert os.path.exists(exe), "tests/cpp/test_module is not built: run __graft_entry__.build()"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "ripcurrents_amd"), os.path.join(ROOT, "oracle"),
                                              env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "test_module: ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_under_host_asan():
    """The same frame loop with the HOST code (include/rcflow_module.hpp + the test) under AddressSanitizer and
    UBSan -- host sanitizers only; the library and the HIP runtime are not instrumented (leak checking is off:
    the runtime keeps its allocations until exit)."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_module_asan")
    assert os.path.exists(exe), "tests/cpp/test_module_asan is not built: run __graft_entry__.build()"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "ripcurrents_amd"), os.path.join(ROOT, "oracle"),
                                              env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "test_module: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


@pytest.mark.gpu
def test_cv_dropin_adapter_compiled_and_run():
    """include/rcflow_cv.hpp -- rc::calcOpticalFlowFarneback with cv::calcOpticalFlowFarneback's own signature, the edit
    INTEGRATION.md proposes for ripcurrents.cpp:215 -- compiled against the minimal <opencv2/core.hpp> stand-in of
    tests/cpp/opencv_standin (test scaffolding, not OpenCV) and run with the reference's argument lists: flow allocated
    by the callee, the oracle's flow within tolerance (flags 0) / bit for bit (main.cpp:264's flags), cv::Exception on
    upstream's preconditions, two threads with two frame sizes at once (the context pool), PyrLK with vector<Point2f>."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_dropin")
    assert os.path.exists(exe), "tests/cpp/test_dropin is not built: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "test_dropin ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_frame_loop_bench_modes_agree():
    """tests/cpp/bench_loop.cpp on a short run: the reference-shaped loop from a C++ host in four forms (flow only;
    flow + analysis as separate C-ABI calls; rcflow_frame_loop_step as one hipGraph launch per frame; the same call
    without the graph) -- the program itself fails when their flow fields, edge masks or seed positions differ."""
    exe = os.path.join(ROOT, "tests", "cpp", "bench_loop")
    assert os.path.exists(exe), "tests/cpp/bench_loop is not built: run __graft_entry__.build()"
    r = subprocess.run([exe, "320", "240", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bench_loop ok" in r.stdout, r.stdout + r.stderr
