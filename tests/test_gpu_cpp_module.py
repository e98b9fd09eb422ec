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
