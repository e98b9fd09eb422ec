"""bench.py's output contract: one JSON line with the driver's keys plus `roofline` and
`cpu_baseline` (run as a child process so that it owns its own GPU context)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--pairs", "16",
                          "--cpu-pairs", "1", "--warmup-seconds", "0.2", "--repeats", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["unit"] == "frames/s"
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 1000 and abs(j["value"] - 16 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 0.01
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["avg_us"] > 0
    # a roofline fraction is physical: compulsory bytes of the launch as built / time / 8 TB/s
    assert 0.0 < r["frac"] < 1.0 and r["peak_measured"]["read"] > 1000
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9) / r["achieved"] < 0.01
    assert "traffic_source" in r and (r["traffic"] is None or r["traffic_source"])
    # the issue-rate evidence next to the byte roofline: present only for the launch shape the counters were taken on
    # (32 pairs per launch), then with a duration-derived figure of this run
    if "valu_issue" in r:
        assert r["valu_issue"]["vector_insts_per_launch"] > 0 and r["valu_issue"]["ns_per_inst_per_simd_this_run"] > 0
    assert 0.0 < j["pipeline"]["frac"] < 1.0 and all(0.0 < k["frac"] < 1.0 for k in j["kernels"])
    assert j["warmup_steps_run"] >= 1
    # the headline is the median of `repeats` timed windows of exactly `steps` steps each
    assert j["repeats"] == 4 and j["value_min"] <= j["value"] <= j["value_max"] and j["timed_seconds_total"] > 0
    # device evidence: one rank, one device, with something that identifies it
    assert j["ranks"] == 1 and j["distinct_devices"] == 1 and len(j["devices"]) == 1
    assert j["devices"][0]["rank"] == 0 and ("uuid" in j["devices"][0] or "pci" in j["devices"][0])
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0


@pytest.mark.parametrize("how", ["launcher", "plain"])
def test_bench_two_ranks_rehearsal(how):
    """The N>1 path of bench.py end to end on the one GPU of the test box, both ways the driver may start it: under
    torch.distributed.run, and as a plain `python bench.py --gpus 2` (no WORLD_SIZE: the script then starts the launcher
    itself, as a child, before it touches the GPU).  Both ranks on cuda:0 with gloo as the transport (RC_REHEARSE_GLOO=1;
    RCCL refuses two ranks on one device).  Segments per rank, the asynchronous histogram all-reduce consumed one step
    later, max-over-ranks timing, one JSON line from rank 0 that names every rank's device."""
    env = dict(os.environ, RC_REHEARSE_GLOO="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = 29600 + os.getpid() % 300
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "4", "--no-roof",
            "--warmup-seconds", "0.2", "--repeats", "3"]
    if how == "launcher":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["config"]["segments"] == 2
    assert "all_reduce int32[1887]" in j["config"]["collective"]
    assert abs(j["value"] - 2 * 4 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 0.01
    assert "cpu_baseline" not in j          # rank 0 at N = 1 only
    assert j["ranks"] == 2 and sorted(d["rank"] for d in j["devices"]) == [0, 1]
    assert j["distinct_devices"] == 1       # the rehearsal puts both ranks on the box's one GPU; a real run reports N


@pytest.mark.parametrize("collective", ["torch", "rccl"])
def test_bench_rccl_backend_rehearsal_with_one_rank(collective):
    """The N>1 code path of bench.py over the REAL backend, as far as one GPU allows: one rank under
    torch.distributed.run with RC_FORCE_DIST=1 initialises the `nccl` (= RCCL) process group and runs every collective of
    the multi-rank run -- the asynchronous int32[1887] histogram all-reduce consumed one step later (torch.distributed, or
    the library's own rcflow_allreduce_hist over a one-rank RCCL communicator whose id travels by broadcast), the
    warm-up broadcast, the barriers and the max-over-ranks reduction of the timing."""
    env = dict(os.environ, RC_FORCE_DIST="1")
    port = 29900 + os.getpid() % 90 + (0 if collective == "torch" else 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--pairs", "8", "--no-roof", "--no-cpu-baseline", "--warmup-seconds", "0.2", "--collective", collective]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 1000
    assert ("librccl" if collective == "rccl" else "torch.distributed") in j["config"]["collective"]
