"""Multi-GPU layer: independent video segments per rank, one collective for the global
flow histogram.

The path shards along time/stream: each rank (one process per GPU) runs the whole hot path
on its own segment and no data-path collective exists.  The only exchange is the sum of
the cumulative flow-histogram counters -- hist[50] | hist2d[36*50] | histsum | histsum2d[36]
= 1887 int32 = 7548 B -- after which every rank derives the same global thresholds
(UPPER, UPPER2d, prop_above_upper) from the same integers.  Integer sums commute, so the
result equals the reference's histogram after processing all segments in any order.

torch.distributed is the transport: backend "nccl" is RCCL over xGMI on ROCm; "gloo" is
used by the CPU tests.
"""
import torch
import torch.distributed as dist

from ._lib import HIST_BINS, HIST_DIRECTIONS, HIST_WORDS


def allreduce_hist_words(words, group=None):
    """Sum the 1887-word histogram block over all ranks; returns a new tensor (the rank's
    own cumulative counters are left untouched)."""
    if words.numel() != HIST_WORDS or words.dtype != torch.int32:
        raise ValueError("expected %d int32 histogram words" % HIST_WORDS)
    g = words.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g


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
