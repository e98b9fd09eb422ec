"""Multi-GPU layer: independent video segments per rank, one collective for the global
flow histogram.

The path shards along time/stream: each rank (one process per GPU) runs the whole hot path
on its own segment and no data-path collective exists.  The only exchange is the sum of
the cumulative flow-histogram counters -- hist[50] | hist2d[36*50] | histsum | histsum2d[36]
= 1887 int32 = 7548 B -- after which every rank derives the same global thresholds
(UPPER, UPPER2d, prop_above_upper) from the same integers.  Integer sums commute, so the
result equals the reference's histogram after processing all segments in any order.

Two transports for that one sum: torch.distributed (backend "nccl" is RCCL over xGMI on ROCm; "gloo" is
used by the CPU tests), and the library's own C-ABI collective (rcflow_comm_init / rcflow_allreduce_hist,
librccl opened directly) -- what a C++ host like the reference's uses; init_comm() below sets it up for a
Python host, with torch.distributed only as the out-of-band channel that carries the 128-byte RCCL id.
"""
import os

import torch
import torch.distributed as dist

from ._lib import HIST_BINS, HIST_DIRECTIONS, HIST_WORDS


def _collective_wanted(group):
    """An initialised group of more than one rank -- or of one rank under RC_FORCE_DIST=1, the rehearsal of the N-rank
    code path (backend included) on a one-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("RC_FORCE_DIST") == "1"


def allreduce_hist_words(words, group=None):
    """Sum the 1887-word histogram block over all ranks; returns a new tensor (the rank's
    own cumulative counters are left untouched)."""
    if words.numel() != HIST_WORDS or words.dtype != torch.int32:
        raise ValueError("expected %d int32 histogram words" % HIST_WORDS)
    g = words.clone()
    if _collective_wanted(group):
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g


class PendingHistSum:
    """An all-reduce of the histogram words in flight (allreduce_hist_words_async).  wait()
    orders the caller's stream after the collective (RCCL: a stream wait, the host does not
    block; gloo: the host blocks) and returns the summed words."""

    def __init__(self, work, words):
        self._work, self._words = work, words

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._words


def allreduce_hist_words_async(words, group=None):
    """Start the sum of the 1887-word block over all ranks and return a PendingHistSum: the
    collective (7.5 KB, latency-bound) then runs on RCCL's stream beside the next batch's flow
    kernels, and the global thresholds are derived when the caller comes back for it."""
    if words.numel() != HIST_WORDS or words.dtype != torch.int32:
        raise ValueError("expected %d int32 histogram words" % HIST_WORDS)
    g = words.clone()
    work = None
    if _collective_wanted(group):
        work = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return PendingHistSum(work, g)


def init_comm(ctx, group=None, rccl_for_one=False):
    """Joins the context's C-ABI communicator with the ranks of the torch.distributed group: rank 0
    creates the RCCL unique id, torch.distributed broadcasts its 128 bytes (any backend), every rank calls
    rcflow_comm_init.  Without an initialised process group the world is one rank (identity collective);
    `rccl_for_one` builds a one-rank RCCL communicator for an initialised group of one instead (rehearsal)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not rccl_for_one):
        ctx.comm_init(0, 1)
        return 0, 1
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [ctx.comm_unique_id() if rank == 0 else None]
    # `rank` is the rank inside `group`; broadcast's src is a GLOBAL rank: the group's first member, not always 0
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    ctx.comm_init(rank, world, box[0])
    return rank, world


def split_hist_words(words):
    """hist[50], hist2d[36][50], histsum, histsum2d[36] views of a words block."""
    w = words.cpu().numpy() if isinstance(words, torch.Tensor) else words
    hist = w[:HIST_BINS]
    hist2d = w[HIST_BINS:HIST_BINS + HIST_DIRECTIONS * HIST_BINS].reshape(HIST_DIRECTIONS, HIST_BINS)
    histsum = int(w[HIST_BINS + HIST_DIRECTIONS * HIST_BINS])
    histsum2d = w[HIST_BINS + HIST_DIRECTIONS * HIST_BINS + 1:]
    return hist, hist2d, histsum, histsum2d


def segment_bounds(nframes, world, rank):
    """Frames [a, b) of rank's segment when one clip is cut into `world` segments with the
    one-frame overlap a flow needs (the `previous` frame of a segment's first pair)."""
    pairs = nframes - 1
    per = pairs // world
    extra = pairs % world
    a = rank * per + min(rank, extra)
    n = per + (1 if rank < extra else 0)
    return a, a + n + 1 if n > 0 else a
