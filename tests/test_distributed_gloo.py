"""The N>1 path on CPU: world_size 2, gloo.  Each rank owns an independent segment
(SURVEY.md section 8(e)); the only exchange is the sum of the 1887 histogram words, after
which every rank derives identical global thresholds.  The words are produced here by the
oracle (no GPU in this tier); the collective helper is the one bench.py uses."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _words_of(st):
    return np.concatenate([st.hist, st.hist2d.ravel(), np.int32([st.histsum.value]), st.histsum2d]).astype(np.int32)


def _segment_state(rank):
    from oracle import oracle
    from ripcurrents_amd import synth
    clip = synth.surf_clip(96, 80, 4, seed=1234 + rank)          # independent segments, seed 1234+rank
    st = oracle.HistState()
    for t in range(3):
        flow = oracle.farneback(clip[t], clip[t + 1])
        oracle.histogram_accumulate(oracle.flow_to_polar(flow), st)
    return st


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from ripcurrents_amd.distributed import allreduce_hist_words, split_hist_words
    st = _segment_state(rank)
    words = torch.from_numpy(_words_of(st))
    g = allreduce_hist_words(words)
    assert torch.equal(words, torch.from_numpy(_words_of(st)))     # the rank's own counters are untouched
    from ripcurrents_amd.distributed import allreduce_hist_words_async
    pend = allreduce_hist_words_async(words)                       # the form bench.py overlaps with the next step
    assert torch.equal(pend.wait(), g) and torch.equal(pend.wait(), g)
    assert torch.equal(words, torch.from_numpy(_words_of(st)))
    hist, hist2d, histsum, histsum2d = split_hist_words(g)
    gs = oracle.HistState()
    gs.hist[:] = hist; gs.hist2d[:] = hist2d; gs.histsum.value = histsum; gs.histsum2d[:] = histsum2d
    oracle.histogram_thresholds(gs)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), words=g.numpy(), UPPER=gs.UPPER, UPPER2d=gs.UPPER2d,
             prop=gs.prop_above_upper)
    dist.barrier()
    dist.destroy_process_group()


def test_histogram_allreduce_two_ranks(tmp_path):
    world, port = 2, 29517 + os.getpid() % 200
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["words"], r1["words"])
    assert r0["UPPER"] == r1["UPPER"] and np.array_equal(r0["UPPER2d"], r1["UPPER2d"])
    assert np.array_equal(r0["prop"], r1["prop"], equal_nan=True)
    # equals processing both segments sequentially on one rank (integer sums commute)
    sys.path.insert(0, ROOT)
    seq = _words_of(_segment_state(0)).astype(np.int64) + _words_of(_segment_state(1)).astype(np.int64)
    assert np.array_equal(r0["words"].astype(np.int64), seq)


def test_allreduce_is_identity_without_a_process_group():
    from ripcurrents_amd.distributed import allreduce_hist_words
    from ripcurrents_amd._lib import HIST_WORDS
    w = torch.arange(HIST_WORDS, dtype=torch.int32)
    g = allreduce_hist_words(w)
    from ripcurrents_amd.distributed import allreduce_hist_words_async
    assert torch.equal(allreduce_hist_words_async(w).wait(), g)
    assert torch.equal(g, w) and g.data_ptr() != w.data_ptr()
    with pytest.raises(ValueError):
        allreduce_hist_words(torch.zeros(5, dtype=torch.int32))


class _FakeCtx:
    """Records what distributed.init_comm asks of a context (no GPU in this tier)."""

    def __init__(self):
        self.calls = []

    def comm_unique_id(self):
        self.calls.append("unique_id")
        return bytes(range(128))

    def comm_init(self, rank, world, unique_id=None):
        self.calls.append(("init", rank, world, unique_id))


def _init_comm_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ripcurrents_amd.distributed import init_comm
    ctx = _FakeCtx()
    assert init_comm(ctx) == (rank, world)
    # only rank 0 creates the RCCL id; every rank joins with the SAME 128 bytes
    assert ("unique_id" in ctx.calls) == (rank == 0)
    assert ctx.calls[-1] == ("init", rank, world, bytes(range(128)))
    dist.barrier()
    dist.destroy_process_group()


def test_c_abi_communicator_bootstrap_over_torch_distributed(tmp_path):
    """distributed.init_comm: the 128-byte RCCL id travels from rank 0 to the others over torch.distributed (the
    out-of-band channel of rcflow_comm_init); without a process group the world is one rank."""
    sys.path.insert(0, ROOT)
    from ripcurrents_amd.distributed import init_comm
    solo = _FakeCtx()
    assert init_comm(solo) == (0, 1) and solo.calls == [("init", 0, 1, None)]
    world, port = 2, 29750 + os.getpid() % 200
    mp.spawn(_init_comm_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
