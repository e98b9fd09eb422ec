#!/usr/bin/env python3
"""bench.py -- frames/sec of dense Farneback flow @1080p on N MI355X GPUs (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--repeats R]
  N>1: either under the launcher (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...,
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or plainly -- without WORLD_SIZE in the
  environment the script starts that launcher itself as a CHILD process before anything touches the GPU and
  exits with its status.  The timed region (exactly K steps between barrier + synchronize on both sides, MAX over
  ranks) is repeated R times; `value` is the median window, `value_min` / `value_max` the extremes.

Workload (BASELINE config 2; config 4 when N>1): a synthetic 1920x1080 surf clip resident in
HBM, 3 pyramid scales (levels=2), the parameter set of ripcurrents.cpp:215
(0.5, 2, 3, 2, 15, 1.2, flags 0).  One STEP = one pass of the hot path over one clip of
`--pairs`+1 frames: every frame is expanded once (pyramid + polynomial expansion, streaming
model) and `--pairs` flow fields are produced, then the segment's flow histogram is
accumulated (and, for N>1, all-reduced over RCCL: 1887 int32).  Each rank owns an
independent segment (seed 1234+rank): weak scaling, no data-path collective.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: compulsory bytes of the launch as
built / its mean duration from HIP events on the kernel's stream inside the timed region / 8 TB/s),
`pipeline` (the whole step against the same peak) and, at N=1, `cpu_baseline` (the CPU oracle timed
on this host on a bounded sample of the same clip).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (scripts/diag/membw.hip on this part: 6.3 TB/s streaming read, 4.5-5.7 write)
W, H = 1920, 1080
PARAMS = dict(pyr_scale=0.5, levels=2, winsize=3, iterations=2, poly_n=15, poly_sigma=1.2, flags=0)


def survey_model_bytes_per_frame(w, h, levels, iters):
    """SURVEY.md section 8(d) algorithmic bytes per frame (streaming model)."""
    n0 = w * h
    total = 0
    nk_prev = None
    sizes = [(round(w * 0.5 ** k), round(h * 0.5 ** k)) for k in range(levels + 1)]
    for k in range(levels, -1, -1):
        nk = sizes[k][0] * sizes[k][1]
        b = n0 + 4 * nk + 24 * nk + 60 * nk + (iters - 1) * 80 * nk + 28 * nk
        if nk_prev is not None:
            b += 8 * nk_prev
        total += b
        nk_prev = nk
    return total


PMC_FILE = "profiles/r03_pmc_traffic.json"


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` and per flow field of the whole step, from the committed
    rocprofv3 --pmc passes (PMC_FILE, collected by scripts/pmc_multi.sh on the same bench command in
    separate runs -- counters cannot be collected inside this process).  None when the launch shape
    differs from the profiled one or the file is absent."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
        chunk = args.chunk or 32
        if min(chunk, args.pairs) != d["pairs_per_launch"] or args.gaussian or args.exact:
            return None, None, None
        src = "%s (%s; commit %s)" % (PMC_FILE, d.get("collected", "separate --pmc passes"), d.get("commit", "?"))
        return d["kernels"][kernel]["traffic_bytes_per_launch"], d.get("pipeline_bytes_per_field"), src
    except Exception:
        return None, None, None


SQ_FILE = "profiles/r03_sq_counters.json"


def valu_issue(kernel, avg_us, args):
    """What the dominant kernel's time follows: its vector-instruction count against the SIMDs' issue
    rate.  Instruction and cycle counts come from the committed SQ counter passes (SQ_FILE, scripts/r3/sq_json.py; counters
    cannot be collected inside this process), the duration is this run's.  None when the launch shape differs or the file
    is absent."""
    try:
        d = json.load(open(os.path.join(ROOT, SQ_FILE)))
        chunk = args.chunk or 32
        if min(chunk, args.pairs) != d["pairs_per_launch"] or args.gaussian or args.exact:
            return None
        k = d["kernels"][kernel]
        return {"vector_insts_per_launch": k["insts_valu"], "waves": k["waves"],
                "cycles_per_inst_per_simd": k["cycles_per_valu_inst_per_simd"],
                "ns_per_inst_per_simd_this_run": round(avg_us * 1e3 * d["simds"] / k["insts_valu"], 3),
                "valu_active_share_of_launch": k["valu_active_share_of_launch"],
                "device_rates_cycles_per_inst": {"pure stream of simple fp32 / integer ops": "2.0 (2.3-2.6 measured at the nominal clock)",
                                                  "every other or mixed stream": "4.0 (4.4-4.9 measured)"},
                "source": "%s (commit %s), profiles/r03_valu_rates.md" % (SQ_FILE, d.get("commit", "?")),
                "note": "the kernel issues one vector instruction per ~4 cycles on every SIMD for the whole launch: "
                        "bound by instruction issue, not by HBM"}
    except Exception:
        return None


def measured_memory_roof():
    """Streaming read / write / copy rates of this device from scripts/diag/membw (a stand-alone HIP microbenchmark
    built by __graft_entry__.build(), run as a child process after the timed region; 2 GiB buffers, beyond the
    Infinity Cache).  None when the tool is absent or fails -- a yardstick next to the 8 TB/s spec, never the peak
    `roofline.frac` is quoted on."""
    import subprocess
    tool = os.path.join(ROOT, "scripts", "diag", "membw")
    if not os.path.exists(tool):
        return None
    try:
        out = subprocess.run([tool, "--json"], capture_output=True, text=True, timeout=120).stdout
        for line in out.splitlines():
            if line.startswith("{"):
                return json.loads(line)
    except Exception:
        pass
    return None


def launcher_command(argv, gpus, port=None):
    """The child command a plain `python bench.py --gpus N` (N > 1, no WORLD_SIZE) starts: the same launcher line the
    driver uses, one rank per GPU, rendezvous on 127.0.0.1."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus,
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def device_identity(torch, dev):
    """What tells two GPUs apart in a SCALE record: uuid and PCI location of the device this rank computes on."""
    out = {"index": dev.index}
    try:
        pr = torch.cuda.get_device_properties(dev)
        out["name"] = pr.name
        u = getattr(pr, "uuid", None)
        if u is not None:
            out["uuid"] = str(u)
        if hasattr(pr, "pci_bus_id"):
            out["pci"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
    except Exception as e:                      # identity is evidence, never a reason to fail the run
        out["error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=15,
                    help="the timed window of --steps steps is repeated this many times (each between barrier + synchronize); "
                         "value = the median window")
    ap.add_argument("--dry-launch", action="store_true", help="print the launcher command --gpus N would start, and exit")
    ap.add_argument("--warmup-seconds", type=float, default=1.0,
                    help="untimed steps continue after --warmup until this much time has passed (clocks and "
                         "caches at steady state however small --warmup is); 0 disables")
    ap.add_argument("--pairs", type=int, default=32, help="frame pairs (flow fields) per step")
    ap.add_argument("--chunk", type=int, default=0, help="pairs per launch (0 = library default)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="library option for same-box A/B runs (rcflow_set_option), e.g. --opt chain=1")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--event-every", type=int, default=20,
                    help="bracket every kernel of every Nth timed step with HIP events (1 = every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roof", action="store_true", help="skip the measured memory roof (read/fill/copy microbench)")
    ap.add_argument("--cpu-pairs", type=int, default=16)
    ap.add_argument("--collective", choices=["torch", "rccl"], default="torch",
                    help="N>1: torch.distributed all_reduce (default) or the library's C-ABI collective "
                         "(rcflow_allreduce_hist over librccl)")
    ap.add_argument("--sync-collective", action="store_true", help="N>1: all-reduce and thresholds inside the step (no one-step pipelining)")
    ap.add_argument("--clip-per-step", action="store_true", help="every step is an independent clip of pairs + 1 frames (pairs + 1 expansions)")
    ap.add_argument("--gaussian", action="store_true", help="main.cpp:264 variant (flags=256)")
    ap.add_argument("--exact", action="store_true",
                    help="option exact = 1: the HIP path in upstream's CPU operation order (bit-identical to the oracle); "
                         "own metric name, headline false")
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c5"], default="c2",
                    help="BASELINE.json configuration: c2 (default, the headline) 1080p 3 scales; c1 640x480 translating texture; "
                         "c3 3840x2160 5 scales (levels=4) + 250 seed streamlines and 5 streaklines per field; c5 16 lock-step 1080p "
                         "streams per GPU through hipGraph replay.  Other than c2: own metric name, headline false")
    ap.add_argument("--mode", choices=["clip", "frame", "host", "host-stateless"], default="clip",
                    help="clip (default, the headline): 32 resident frames per call.  The reference-shaped loops, "
                         "reported under their own metric name, never as the headline: frame = one resident frame per "
                         "call (rcflow_push_frame_dev); host = frames from host memory through the page-locked double "
                         "buffer, flow stays on the device (rcflow_push_frame_u8; PCIe-inclusive upload); host-stateless = "
                         "the literal two-image drop-in with host pointers and the flow copied back (rcflow_farneback_u8)")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.repeats < 1:
        ap.error("--gpus, --steps and --repeats must be positive")

    # N > 1 without a launcher around us: start it as a child, BEFORE torch is imported or any GPU call is made (a
    # process that has initialised the GPU must never be replaced or forked), and leave with its status
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        cmd = launcher_command([a for a in sys.argv[1:] if a != "--dry-launch"], args.gpus)
        if args.dry_launch:
            print(json.dumps({"launch": cmd}))
            sys.exit(0)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))
    if args.dry_launch:
        print(json.dumps({"launch": None}))
        sys.exit(0)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print("bench.py --gpus %d inside a launcher of WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    # RC_REHEARSE_GLOO=1: rehearsal of the N>1 path on a one-GPU box (every rank on cuda:0, gloo
    # transport); never set by the driver
    rehearse = os.environ.get("RC_REHEARSE_GLOO") == "1"
    # RC_FORCE_DIST=1: the N>1 code path (process group, collectives, barriers) with however many ranks there are --
    # with one rank under torch.distributed.run it rehearses the RCCL backend on a one-GPU box; never set by the driver
    multi = world > 1 or os.environ.get("RC_FORCE_DIST") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from ripcurrents_amd import synth
    from ripcurrents_amd.api import Context
    from ripcurrents_amd.distributed import allreduce_hist_words, allreduce_hist_words_async, init_comm

    global W, H
    params = dict(PARAMS)
    if args.gaussian:
        params["flags"] = 256
    if args.config != "c2" and args.mode != "clip":
        print("--mode applies to --config c2 only", file=sys.stderr)
        sys.exit(2)
    if args.config == "c1":
        W, H = 640, 480
    elif args.config == "c3":
        W, H = 3840, 2160
        params["levels"] = 4
        if args.pairs == 32:
            args.pairs = 8
    elif args.config == "c5":
        args.pairs = 16                      # 16 streams, one lock-step push per step
    T = args.pairs + 1
    if args.config == "c1":
        frames = synth.translating_clip(W, H, T, seed=1234 + rank, device=dev)
    elif args.config == "c5":
        # 16 independent streams: clips of 3 frames with different seeds, played forwards and backwards
        frames = torch.stack([synth.surf_clip(W, H, 3, seed=1234 + 100 * rank + s_, device=dev) for s_ in range(16)])   # [S,3,H,W]
    else:
        frames = synth.surf_clip(W, H, T, seed=1234 + rank, device=dev)      # generated in HBM
    flows = torch.empty((args.pairs, H, W, 2), dtype=torch.float32, device=dev)
    ctx = Context(W, H, device=local_rank, streams=1)
    if args.exact:
        ctx.set_option("exact", 1)
    if args.chunk:
        ctx.set_option("chunk", args.chunk)
    for o in args.opt:
        name, val = o.split("=")
        ctx.set_option(name, int(val))
    ctx.analysis_reset(W, H)
    hist_words = ctx.histogram_words()
    use_cabi = multi and args.collective == "rccl" and not rehearse
    if use_cabi:
        init_comm(ctx, rccl_for_one=os.environ.get("RC_FORCE_DIST") == "1")

    pending = []
    cabi_pending = []

    def finish_pending():
        while pending:
            ctx.thresholds_from_words(pending.pop(0).wait())
        while cabi_pending:
            ctx.allreduce_hist_join()
            ctx.thresholds_from_words(cabi_pending.pop())

    # The segment is a stream: every step pushes the next `pairs` frames (rcflow_push_clip_dev), the slot keeps
    # the last frame's expansion, so every frame is expanded exactly once (SURVEY 8(d)'s streaming model).  The
    # synthetic clip is played forwards, then backwards, then forwards ...: consecutive frames are always
    # neighbours in the clip.  --clip-per-step restores independent clips of pairs + 1 frames per step.
    if args.config == "c5":
        c5_order = [0, 1, 2, 1]              # frame index per lock-step push: consecutive frames are always neighbours
        c5_frames = [frames[:, i].contiguous() for i in range(3)]
        c5_stage = torch.empty((16, H, W), dtype=torch.uint8, device=dev)    # the fixed staging buffer the graph captures
        ctx.batch_reset()
        c5_stage.copy_(c5_frames[0])
        ctx.push_batch(c5_stage, flows, use_graph=True, **params)             # primes
        fwd = bwd = frames
    else:
        fwd = frames[1:]
        bwd = frames.flip(0)[1:].contiguous()
    c3_seeds = c3_streak = None
    if args.config == "c3":
        from ripcurrents_amd.api import Streakline
        g = torch.Generator(device="cpu").manual_seed(7)
        c3_seeds = torch.stack([torch.rand(250, generator=g) * W, torch.rand(250, generator=g) * H], dim=1).float().to(dev)
        c3_streak = torch.tensor([[W * (0.2 + 0.15 * i), H * 0.5] for i in range(5)], dtype=torch.float32, device=dev)
    host_fwd = host_bwd = host_flow = None
    host_last = [None]
    if args.mode in ("host", "host-stateless"):
        host_fwd = fwd.cpu().numpy()
        host_bwd = bwd.cpu().numpy()
        host_last[0] = frames[0].cpu().numpy()
        host_flow = np.empty((H, W, 2), np.float32)
    if args.mode == "host":
        ctx.stream_reset()
        ctx.push_frame_host(host_last[0], **params)
    elif args.config == "c5":
        pass
    elif args.mode == "frame" or not args.clip_per_step:
        if args.mode == "frame":
            ctx.set_option("frame_overlap", 2)       # the clip is resident: every frame is complete when it is pushed
        ctx.stream_reset()
        ctx.push_clip(frames[0:1], flows, **params)          # primes the stream, no flow
    nstep = [0]

    def step():
        # a step is one segment of `pairs` frames: its histogram starts from zero (int32 counters:
        # 32 x 2.07 M counts per step; cumulative over the run they would wrap after ~32 steps)
        ctx.histogram_reset()
        if args.config == "c5":
            nstep[0] += 1
            c5_stage.copy_(c5_frames[c5_order[nstep[0] % 4]])                 # the 16 streams' next frames arrive (33 MB device copy)
            ctx.push_batch(c5_stage, flows, use_graph=True, **params)
        elif args.mode == "frame":
            seq = fwd if nstep[0] % 2 == 0 else bwd
            for t in range(args.pairs):
                ctx.push_frame(seq[t], flows[t], **params)
            nstep[0] += 1
        elif args.mode == "host":
            seq = host_fwd if nstep[0] % 2 == 0 else host_bwd
            for t in range(args.pairs):
                f = ctx.push_frame_host(seq[t], **params)
                flows[t].copy_(f, non_blocking=True)       # device-side copy into the step's batch (the analysis input)
            nstep[0] += 1
        elif args.mode == "host-stateless":
            seq = host_fwd if nstep[0] % 2 == 0 else host_bwd
            prev = host_last[0]
            for t in range(args.pairs):
                ctx.calcOpticalFlowFarneback(prev, seq[t], host_flow, **params)
                prev = seq[t]
            host_last[0] = prev
            nstep[0] += 1
            return                                           # the flow is on the host: no device-side analysis
        elif args.clip_per_step:
            ctx.farneback_clip(frames, flows, **params)
        else:
            got = ctx.push_clip(fwd if nstep[0] % 2 == 0 else bwd, flows, **params)
            assert got.shape[0] == args.pairs
            nstep[0] += 1
        if args.config == "c3":
            # config 3's second half: the 250 seed streamlines (ripcurrents.cpp:170-172, :283-285) and the 5 streakline
            # generation points (main.cpp:118) advected through every field of the step
            for t in range(args.pairs):
                ctx.streamline(c3_seeds, flows[t], 2.0, 1, 100.0, variant=3)
                ctx.streamline(c3_streak, flows[t], 1.0, 1, 0.0, variant=4)
        ctx.histogram_accumulate_clip(flows)
        if multi:
            # global flow histogram: integer sum over RCCL, order independent; every
            # rank then derives the same global thresholds from the same integers.  The 7.5 KB
            # collective of step k runs on RCCL's stream beside the flow kernels of step k+1 and
            # its thresholds are derived one step later (all of them inside the timed region).
            if use_cabi:
                # C-ABI collective: starts on its own stream; the thresholds of step k are enqueued behind it
                # at step k+1 (or right away with --sync-collective)
                if not args.sync_collective and cabi_pending:
                    ctx.allreduce_hist_join()
                    ctx.thresholds_from_words(cabi_pending.pop())
                g = ctx.allreduce_hist()
                if args.sync_collective:
                    ctx.allreduce_hist_join()
                    ctx.thresholds_from_words(g)
                else:
                    cabi_pending.append(g)
            elif args.sync_collective:
                ctx.thresholds_from_words(allreduce_hist_words(hist_words))
            else:
                nxt = allreduce_hist_words_async(hist_words)
                finish_pending()
                pending.append(nxt)
        else:
            ctx.thresholds()

    t_w = time.perf_counter()
    warm_run = 0
    for _ in range(args.warmup):
        step()
        warm_run += 1
    # time-based floor: every rank runs the same number of extra untimed steps (rank 0's clock decides)
    while args.warmup_seconds > 0:
        torch.cuda.synchronize()
        more = torch.tensor([1 if time.perf_counter() - t_w < args.warmup_seconds else 0], device=dev)
        if multi:
            dist.broadcast(more, 0)
        if not int(more.item()):
            break
        for _ in range(4):
            step()
            warm_run += 1
    finish_pending()
    torch.cuda.synchronize()
    events = not args.no_kernel_events
    if events:
        ctx.profile_reset()
    # The timed window -- EXACTLY --steps steps between barrier + synchronize on both sides, MAX over ranks -- repeated
    # --repeats times back to back: the headline is the median window, so it rests on R x K steps of GPU time instead of K
    windows = []
    gstep = 0
    for rep in range(args.repeats):
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            # HIP events on the kernels' own stream, inside the timed region; sampled steps only,
            # because an event pair between two kernels keeps them from overlapping at all
            if events:
                ctx.profile_enable(gstep % args.event_every == 0)
            gstep += 1
            step()
        finish_pending()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        windows.append(el)
    elapsed = float(np.median(windows))
    sampled_steps = len(range(0, gstep, args.event_every))
    prof = []
    if events:
        ctx.profile_enable(False)
        prof = ctx.profile_read()
    # evidence that N ranks sat on N devices: every rank's device uuid / PCI location, gathered
    ident = device_identity(torch, dev)
    ident.update(rank=rank, local_rank=local_rank, host=os.uname().nodename)
    idents = [ident]
    if multi:
        idents = [None] * dist.get_world_size()
        dist.all_gather_object(idents, ident)
    roof = None
    if rank == 0 and world == 1 and not args.no_roof:
        roof = measured_memory_roof()

    if rank == 0:
        frames_done = world * args.pairs * args.steps
        fps = frames_done / elapsed
        distinct = len({(d.get("host"), d.get("uuid") or d.get("pci") or d.get("index")) for d in idents})
        model_b = survey_model_bytes_per_frame(W, H, params["levels"], params["iterations"])
        mode_names = {"clip": "", "frame": " [one resident frame per call]", "host": " [host frames, PCIe upload inclusive, flow resident]",
                      "host-stateless": " [two-image host-pointer drop-in, PCIe both ways, blocking]"}
        cfg_names = {"c2": "@1080p", "c1": "@640x480 (config 1, translating texture)", "c3": "@3840x2160, 5 scales + streamline / streakline advection (config 3)",
                     "c5": "@1080p, 16 lock-step streams per GPU, hipGraph replay (config 5)"}
        out = {
            "metric": "frames/sec dense Farneback flow " + cfg_names[args.config] + mode_names[args.mode]
                      + (" [option exact: upstream's operation order, bit-identical to the CPU oracle]" if args.exact else ""),
            "bench_config": args.config, "mode": args.mode, "exact": bool(args.exact),
            "headline": args.mode == "clip" and args.config == "c2" and not args.exact and not args.gaussian,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "repeats": args.repeats, "value_min": round(frames_done / max(windows), 2), "value_max": round(frames_done / min(windows), 2),
            "timed_seconds_total": round(sum(windows), 4),
            "ranks": dist.get_world_size() if multi else 1, "distinct_devices": distinct, "devices": idents,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s %dx%d synthetic %s clip, %d pyramid scales (levels=%d), winsize 3, iters 2, poly_n 15, sigma 1.2, "
                                   "flags %d; %d flow fields per step per GPU, %s; + the flow histogram and thresholds of the step's "
                                   "fields (counters reset per step)%s"
                                   % (args.config.upper(), W, H, "translating-texture" if args.config == "c1" else "surf",
                                      params["levels"] + 1, params["levels"], params["flags"], args.pairs,
                                      "16 independent streams advanced one frame in lock-step per step (rcflow_push_batch_dev, hipGraph replay)"
                                      if args.config == "c5" else
                                      "an independent clip of %d frames per step (%d expansions)" % (args.pairs + 1, args.pairs + 1)
                                      if args.clip_per_step else
                                      "the next %d frames of a continuing segment per step (streaming model: every frame "
                                      "expanded once; the clip is played forwards and backwards)" % args.pairs,
                                      "; + 250 seed streamlines and 5 streakline points advected through every field" if args.config == "c3" else ""),
                       "pairs_per_step": args.pairs, "segments": world,
                       "collective": ("all_reduce int32[1887] per step" + ("" if args.sync_collective else ", overlapped with the next step")
                                      + (" (rcflow_allreduce_hist, librccl)" if use_cabi else " (torch.distributed)")) if multi else "none"},
            "warmup_steps_run": warm_run,
            "survey_model": {"bytes_per_frame": model_b,
                             "frac_of_8TBs": round(fps / world * model_b / (HBM_PEAK_GBS * 1e9), 4),
                             "note": "SURVEY.md 8(d) UNFUSED byte model x fps / 8 TB/s: the pipeline as built never "
                                     "materialises M and fuses two iterations, so it moves far fewer bytes than this "
                                     "model -- a throughput yardstick (70 % = 10 278 frames/s), not a roofline fraction"},
        }
        if prof:
            tot = sum(p["total_ms"] for p in prof)
            dom = max(prof, key=lambda p: p["total_ms"])
            secs = dom["total_ms"] * 1e-3
            own = dom["alg_bytes"] / secs / 1e9            # compulsory bytes: inputs once + outputs once, as built
            traffic, pipe_bytes, src = pmc_traffic(dom["kernel"], args)
            out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(own, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(own / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "traffic_source": src,
                               "avg_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                               "alg_bytes_per_launch": dom["alg_bytes"] / dom["launches"],
                               "alg_bytes_per_px_pair": round(dom["alg_bytes"] / dom["launches"] / (W * H * min(args.pairs, args.chunk or 32)), 3),
                               "alg_bytes_def": ("bytes of the launches of this kind as staged through HBM (exact path: M, the "
                                                 "column sums and the flow between its kernels)" if args.exact else
                                                 "compulsory bytes of the launch as built, per pixel of the scale and pair: R1 20 "
                                                 "+ R0 20 x (pairs that read it from memory: the head of every tile chain; the "
                                                 "others take it from the previous pair's R1 window in LDS) + coarse flow 2 + "
                                                 "flow out 8 (M never exists in HBM, two iterations per launch)"),
                               "frac_of_traffic": round(traffic / secs * dom["launches"] / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                               "survey_model_equivalent": {"bytes_per_launch": dom["model_bytes"] / dom["launches"],
                                                           "GBs": round(dom["model_bytes"] / secs / 1e9, 1),
                                                           "note": "SURVEY 8(d) bytes of the stages this launch replaces; "
                                                                   "exceeds the peak because those bytes are never moved"},
                               "share_of_gpu_time": round(dom["total_ms"] / tot, 3)}
            vi = valu_issue(dom["kernel"], 1e3 * dom["total_ms"] / dom["launches"], args)
            if vi:
                out["roofline"]["valu_issue"] = vi
            # whole step: compulsory bytes of every launch of a sampled step / wall time of a step
            step_bytes = sum(p["alg_bytes"] for p in prof) / sampled_steps
            out["pipeline"] = {"compulsory_bytes_per_step": step_bytes,
                               "frac": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                               "measured_bytes_per_field": pipe_bytes,
                               "frac_measured": round(pipe_bytes * fps / world / 1e9 / HBM_PEAK_GBS, 4) if pipe_bytes else None,
                               "note": "whole step (expansions + flow + histogram) against 8 TB/s: compulsory bytes as "
                                       "built, and FETCH/WRITE_SIZE bytes from the committed --pmc passes"}
            if roof:
                # the rates this device streams at (scripts/diag/membw), reported next to the 8 TB/s spec
                out["roofline"]["peak_measured"] = roof
            out["kernels"] = [{"kernel": p["kernel"], "launches": p["launches"],
                               "avg_us": round(1e3 * p["total_ms"] / p["launches"], 2),
                               "GBs_compulsory": round(p["alg_bytes"] / (p["total_ms"] * 1e-3) / 1e9, 1),
                               "frac": round(p["alg_bytes"] / (p["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for p in prof]
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle   # checker timed as the CPU baseline, never the product
            host = frames[:args.cpu_pairs + 1].cpu().numpy()
            oracle.farneback(host[0][:270, :480], host[1][:270, :480])    # warm the library
            tc = time.perf_counter()
            for t in range(args.cpu_pairs):
                oracle.farneback(host[t], host[t + 1], **{("iters" if k == "iterations" else k): v
                                                          for k, v in params.items()})
            cpu_s = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": round(args.cpu_pairs / cpu_s, 4), "unit": "frames/s", "cores": 1,
                                   "kind": "port",
                                   "sample": "%d 1080p frame pairs of the same clip, stateless two-image calls, "
                                             "oracle/farneback_oracle.cpp -O3 single thread (%.1f s)"
                                             % (args.cpu_pairs, cpu_s)}
            # the same port row-striped over this process's host cores
            ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))   # 16 = one GPU's host share on the pool
            if ncores > 1:
                kw = {("iters" if k == "iterations" else k): v for k, v in params.items()}
                oracle.farneback(host[0], host[1], nthreads=ncores, **kw)
                tc = time.perf_counter()
                n_mt = 3 * args.cpu_pairs
                for t in range(n_mt):
                    oracle.farneback(host[t % args.cpu_pairs], host[t % args.cpu_pairs + 1], nthreads=ncores, **kw)
                mt_s = time.perf_counter() - tc
                out["cpu_baseline"]["all_cores"] = {"value": round(n_mt / mt_s, 4), "unit": "frames/s",
                                                    "cores": ncores,
                                                    "sample": "%d pairs, row-striped std::thread (%.1f s)" % (n_mt, mt_s)}
            # Opportunistic: a LOCALLY installed OpenCV -- never fetched -- is the real reference
            # implementation of this path; when one exists it is timed beside the port (absent from this pool's images)
            try:
                import cv2
            except Exception:
                cv2 = None
            if cv2 is not None:
                def cv_run(n):
                    t0 = time.perf_counter()
                    for t in range(n):
                        cv2.calcOpticalFlowFarneback(host[t % args.cpu_pairs], host[t % args.cpu_pairs + 1], None, params["pyr_scale"],
                                                     params["levels"], params["winsize"], params["iterations"], params["poly_n"],
                                                     params["poly_sigma"], params["flags"])
                    return time.perf_counter() - t0
                cv2.ocl.setUseOpenCL(False)
                cv2.setNumThreads(1)
                cv_run(1)
                one = cv_run(args.cpu_pairs)
                cv2.setNumThreads(-1)
                cv_run(1)
                many = cv_run(3 * args.cpu_pairs)
                out["cpu_baseline"]["opencv"] = {"version": cv2.__version__, "kind": "reference",
                                                 "single_thread": round(args.cpu_pairs / one, 4),
                                                 "default_threads": round(3 * args.cpu_pairs / many, 4), "unit": "frames/s"}
        print(json.dumps(out))
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
