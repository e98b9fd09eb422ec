"""CPU tests of the drop-in boundary and the host logic (no GPU, no compute calls)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rcflow.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rcflow_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ripcurrents_amd import _lib
    lib = _lib.load()                       # raises if the HIP extension is not built
    syms = _declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), "include/rcflow.h declares %s but librcflow.so does not export it" % s
        assert s in _lib.SIGNATURES, "%s has no ctypes signature" % s
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.rcflow_abi_version() == 1


def test_no_device_fails_loudly_never_falls_back():
    """Without a GPU rcflow_create returns RC_ENODEV and the Python host refuses to run:
    there is no CPU fallback in the product path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ripcurrents_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.rcflow_create(ctypes.byref(h), 0, 640, 480, 1)
    assert rc == -4 and not h.value
    assert b"HIP device" in lib.rcflow_last_error()
    from ripcurrents_amd.api import Context
    with pytest.raises(RuntimeError):
        Context(640, 480)
    assert lib.rcflow_create(ctypes.byref(h), 0, -1, 480, 1) == -1


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under ripcurrents_amd/ may reference it."""
    for base, _, files in os.walk(os.path.join(ROOT, "ripcurrents_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                for pat in ("import oracle", "from oracle", "liboracle", "rc_oracle.h", "oracle/", "oracle."):
                    assert pat not in text, "%s references the oracle (%s)" % (os.path.join(base, f), pat)
    out = subprocess.run(["ldd", os.path.join(ROOT, "ripcurrents_amd", "librcflow.so")], capture_output=True, text=True).stdout
    assert "liboracle" not in out


def test_level_geometry_entry_point_matches_oracle(orc):
    from ripcurrents_amd import _lib
    lib = _lib.load()
    for (w, h, ps, lv) in [(1920, 1080, 0.5, 2), (3840, 2160, 0.5, 4), (640, 480, 0.5, 2), (500, 375, 0.8, 6), (40, 36, 0.5, 3)]:
        for k in range(0, 7):
            wk, hk = ctypes.c_int(), ctypes.c_int()
            L = lib.rcflow_level_geometry(w, h, ps, lv, k, ctypes.byref(wk), ctypes.byref(hk))
            g = orc.level_geometry(w, h, ps, lv, k)
            assert (L, wk.value, hk.value) == (g["levels"], g["w"], g["h"])
    assert lib.rcflow_level_geometry(640, 480, 1.0, 2, 0, None, None) == -1


def test_synthetic_clips_are_deterministic():
    from ripcurrents_amd import synth
    a = synth.surf_clip(160, 120, 3, seed=1234)
    b = synth.surf_clip(160, 120, 3, seed=1234)
    assert a.dtype == np.uint8 and a.shape == (3, 120, 160) and np.array_equal(a, b)
    assert not np.array_equal(a, synth.surf_clip(160, 120, 3, seed=1235))
    assert 100 < a.mean() < 156 and 20 < a.std() < 60
    t = synth.translating_clip(64, 48, 2)
    assert t.shape == (2, 48, 64)
    f = synth.rotation_field(640, 480)
    assert f[200, 200, 0] == np.float32(-(200 - 240.0) / 480 * 100)      # main.cpp:376


def test_bench_byte_model_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.survey_model_bytes_per_frame(1920, 1080, 2, 2) == 544838400      # 544.84 MB
    assert bench.survey_model_bytes_per_frame(1920, 1080, 2, 3) == 762566400      # 762.57 MB


def test_segment_bounds_cover_the_clip():
    from ripcurrents_amd.distributed import segment_bounds
    for nframes in (2, 9, 64, 65):
        for world in (1, 2, 3, 8):
            pairs = []
            for r in range(world):
                a, b = segment_bounds(nframes, world, r)
                pairs += [(t, t + 1) for t in range(a, b - 1)]
            assert pairs == [(t, t + 1) for t in range(nframes - 1)]


def test_headers_are_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/rcflow.h (and the oracle's header) compile as C99."""
    src = tmp_path / "c99.c"
    src.write_text('#include "rcflow.h"\n#include "rc_oracle.h"\nint main(void) { return RC_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_option_is_documented_in_the_header():
    """rcflow_set_option's names (csrc/rcflow_api.hip) all appear in include/rcflow.h's option list."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "ripcurrents_amd", "csrc", "rcflow_api.hip")).read()
    hdr = open(os.path.join(root, "include", "rcflow.h")).read()
    names = set(re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', src))
    assert len(names) >= 10
    missing = [n for n in sorted(names) if '"%s"' % n not in hdr]
    assert not missing, missing


def test_bench_self_launch_command():
    """`python bench.py --gpus N` without a launcher around it starts one itself (a child process, before any GPU call):
    the same torch.distributed.run line the driver uses, rendezvous on 127.0.0.1, the script's own arguments passed on.
    --dry-launch prints the command instead of running it (no GPU, no torch import)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
                          "--dry-launch"], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout)["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    # one GPU, or a launcher already around the script: nothing to start
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-launch"], capture_output=True,
                         text=True, timeout=60, env=env)
    assert json.loads(out.stdout)["launch"] is None
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch"], capture_output=True,
                         text=True, timeout=60, env=dict(env, WORLD_SIZE="8", RANK="0", LOCAL_RANK="0"))
    assert json.loads(out.stdout)["launch"] is None


def test_tile_chain_plan_invariants():
    """Host logic of the fused winsize-3 flow kernel's tile chains (option `chain`): the pair groups of a launch cover
    every pair exactly once in order, no group is longer than the option, the groups over the last `chain` pairs halve
    (the launch drains on short blocks), a launch too small to keep ~4 blocks per block slot of the GPU shortens its
    chains or gets none, and chain <= 1 means independent pairs.  Pure host code: runs without a GPU."""
    import ripcurrents_amd
    lib = ripcurrents_amd.load()
    fn = lib.rcflow_debug_chain_plan
    fn.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    fn.restype = ctypes.c_int

    def plan(w, h, pairs, chain, force=0):
        buf = (ctypes.c_int * 80)()
        n = fn(w, h, pairs, chain, force, buf, 80)
        assert n >= 0
        return [buf[i] for i in range(n + 1)] if n else []

    assert plan(1920, 1080, 32, 8) == [0, 8, 16, 24, 28, 30, 31, 32]          # 8, 8, 8, then 4, 2, 1, 1
    assert plan(1920, 1080, 32, 1) == [] and plan(1920, 1080, 1, 8) == []     # independent pairs
    assert plan(480, 270, 32, 8) == []                                        # scale 2 of 1080p: too few blocks for chains
    p1 = plan(960, 540, 32, 8)                                                # scale 1: chains of 4
    assert p1 and max(b - a for a, b in zip(p1, p1[1:])) == 4
    for (w, h, pairs, chain, force) in [(1920, 1080, 32, 8, 0), (3840, 2160, 8, 8, 0), (640, 480, 9, 4, 1), (257, 130, 11, 64, 1),
                                        (1920, 1080, 64, 3, 0), (1920, 1080, 5, 2, 1), (333, 251, 63, 8, 1)]:
        st = plan(w, h, pairs, chain, force)
        assert st, (w, h, pairs, chain)
        lens = [b - a for a, b in zip(st, st[1:])]
        assert st[0] == 0 and st[-1] == pairs and all(l >= 1 for l in lens), st
        assert max(lens) <= chain and len(st) - 1 <= 64
        assert lens[-1] == 1 or pairs <= 1                                    # the launch ends on chains of one
        tail = [l for l in lens if l < max(lens)]
        assert tail == sorted(tail, reverse=True), lens                       # halving tail, never growing again
