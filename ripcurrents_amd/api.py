"""Host-side mirror of the reference's interface for the hot path, over librcflow's C ABI.

Names, argument order and meaning follow the reference (paths relative to
/root/reference/RipCurrents_main):
  calcOpticalFlowFarneback   cv:: call at ripcurrents.cpp:215, main.cpp:264,...
  create_histogram           ripcurrents_module.cpp:89-144   (ripcurrents.hpp:39)
  create_flow                ripcurrents_module.cpp:153-182  (ripcurrents.hpp:50)
  create_accumulationbuffer  ripcurrents_module.cpp:189-212  (ripcurrents.hpp:52)
  streamline_field           ripcurrents_module.cpp:608-648  (ripcurrents.hpp:22)
  streamline / _2 / _3       ripcurrents_module.cpp:486-606  (ripcurrents.hpp:23-25)
  get_delta                  ripcurrents_module.cpp:650-679  (ripcurrents.hpp:58)
  Streakline                 Streakline.hpp:8-20, Streakline.cpp:11-71
  subtructAverage / subtructMeanMagnitude / stabilizer / vectorToColor / shearRateToColor

torch is used for device memory and streams only; all compute is in the HIP library.
Arrays cross this layer as torch CUDA tensors (zero copy) or numpy arrays (copied).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import FarnebackParams, HIST_BINS, HIST_DIRECTIONS, HIST_WORDS, check

__all__ = ["Context", "FarnebackParams", "Streakline", "Timeline", "PopulationMap", "HistState"]


def _params(pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
    return FarnebackParams(float(pyr_scale), int(levels), int(winsize), int(iterations), int(poly_n),
                           float(poly_sigma), int(flags))


def _is_t(x):
    return isinstance(x, torch.Tensor)


class HistState:
    """Host copy of the caller-owned arrays of create_histogram (ripcurrents.cpp:147-154)."""

    def __init__(self):
        self.hist = np.zeros(HIST_BINS, np.int32)
        self.histsum = 0
        self.hist2d = np.zeros((HIST_DIRECTIONS, HIST_BINS), np.int32)
        self.histsum2d = np.zeros(HIST_DIRECTIONS, np.int32)
        self.UPPER = 100.0
        self.UPPER2d = np.zeros(HIST_DIRECTIONS, np.float32)
        self.prop_above_upper = np.zeros(HIST_DIRECTIONS, np.float32)


class Context:
    """One GPU context with `streams` independent stream slots (rcflow_create)."""

    def __init__(self, max_w, max_h, device=0, streams=1):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("ripcurrents_amd needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        check(self._lib.rcflow_create(C.byref(h), device, max_w, max_h, streams))
        self._h = h
        self.max_w, self.max_h = max_w, max_h
        self._own, self._bound = set(), {}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rcflow_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ------------------------------------------------------------------ plumbing
    def sync(self, stream=0):
        check(self._lib.rcflow_sync(self._h, stream))

    def use_torch_stream(self, stream=0, torch_stream=None):
        """Run the slot on a torch stream (default: torch's current stream)."""
        ts = torch_stream if torch_stream is not None else torch.cuda.current_stream(self.device)
        self._own.discard(stream)
        self._bound[stream] = ts.cuda_stream
        check(self._lib.rcflow_set_hip_stream(self._h, stream, C.c_void_p(ts.cuda_stream)))

    def use_own_stream(self, stream=0):
        self._own.add(stream)
        check(self._lib.rcflow_use_own_stream(self._h, stream))

    def _bind(self, stream):
        """Device entry points run on torch's current stream (so tensor lifetimes and
        torch.cuda events order against them) unless use_own_stream() was asked for."""
        if stream not in self._own:
            ts = torch.cuda.current_stream(self.device).cuda_stream
            if self._bound.get(stream) != ts:
                check(self._lib.rcflow_set_hip_stream(self._h, stream, C.c_void_p(ts)))
                self._bound[stream] = ts

    def set_option(self, name, value):
        check(self._lib.rcflow_set_option(self._h, name.encode(), int(value)))

    def _dev(self, a, dtype):
        if _is_t(a):
            if not a.is_cuda or a.dtype != dtype:
                raise TypeError("expected a CUDA tensor of dtype %s" % dtype)
            return a
        return torch.as_tensor(np.ascontiguousarray(a)).to(self.device, dtype)

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr())

    # ------------------------------------------------------------------ A: Farneback
    def calcOpticalFlowFarneback(self, prev, next, flow=None, pyr_scale=0.5, levels=2, winsize=3,
                                 iterations=2, poly_n=15, poly_sigma=1.2, flags=0, stream=0):
        """cv::calcOpticalFlowFarneback(prev, next, flow, ...) -> flow (HxWx2 float32).

        numpy inputs use the host-pointer entry point (copy in, compute, copy out);
        CUDA tensors use the device entry point and return a CUDA tensor (asynchronous).
        """
        if _is_t(prev):
            if prev.shape != next.shape or prev.dim() != 2:
                raise ValueError("prev and next must be HxW and equal in size")
            h, w = prev.shape
            if flow is None:
                flow = torch.empty((h, w, 2), dtype=torch.float32, device=prev.device)
            p = _params(pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)
            self._bind(stream)
            check(self._lib.rcflow_farneback_dev(
                self._h, stream, self._ptr(prev), prev.stride(0), self._ptr(next), next.stride(0), w, h,
                self._ptr(flow), flow.stride(0) * 4, C.byref(p)))
            return flow
        prev = np.asarray(prev)
        next = np.asarray(next)
        if prev.dtype != np.uint8 or next.dtype != np.uint8 or prev.ndim != 2 or prev.shape != next.shape:
            raise ValueError("prev and next must be HxW uint8 and equal in size")
        if prev.strides[1] != 1:
            prev = np.ascontiguousarray(prev)
        if next.strides[1] != 1:
            next = np.ascontiguousarray(next)
        h, w = prev.shape
        if flow is None:
            flow = np.empty((h, w, 2), np.float32)
        check(self._lib.rcflow_farneback_u8(
            self._h, stream, prev.ctypes.data, prev.strides[0], next.ctypes.data, next.strides[0], w, h,
            flow.ctypes.data, flow.strides[0], pyr_scale, levels, winsize, iterations, poly_n, poly_sigma,
            flags))
        return flow

    def push_frame(self, frame, flow=None, stream=0, **kw):
        """Streaming frame loop (ripcurrents.cpp:194-221): returns None for the first frame."""
        frame = self._dev(frame, torch.uint8)
        h, w = frame.shape
        p = _params(kw.get("pyr_scale", 0.5), kw.get("levels", 2), kw.get("winsize", 3),
                    kw.get("iterations", 2), kw.get("poly_n", 15), kw.get("poly_sigma", 1.2),
                    kw.get("flags", 0))
        if flow is None:
            flow = torch.empty((h, w, 2), dtype=torch.float32, device=frame.device)
        self._bind(stream)
        rc = check(self._lib.rcflow_push_frame_dev(self._h, stream, self._ptr(frame), frame.stride(0), w, h,
                                                   self._ptr(flow), flow.stride(0) * 4, C.byref(p)))
        return None if rc == 1 else flow

    def push_frame_host(self, frame, stream=0, **kw):
        """The frame loop with HOST frames (rcflow_push_frame_u8): numpy HxW uint8 in, page-locked
        double-buffered upload, the flow field stays on the device.  Returns None when the call primed
        the stream, else a CUDA tensor aliasing the slot's resident flow field (valid until the next push)."""
        frame = np.asarray(frame)
        if frame.dtype != np.uint8 or frame.ndim != 2 or frame.strides[1] != 1:
            frame = np.ascontiguousarray(frame, np.uint8)
        h, w = frame.shape
        p = _params(kw.get("pyr_scale", 0.5), kw.get("levels", 2), kw.get("winsize", 3),
                    kw.get("iterations", 2), kw.get("poly_n", 15), kw.get("poly_sigma", 1.2),
                    kw.get("flags", 0))
        self._bind(stream)
        rc = check(self._lib.rcflow_push_frame_u8(self._h, stream, frame.ctypes.data, frame.strides[0], w, h, C.byref(p)))
        if rc == 1:
            return None
        d = C.c_void_p()
        check(self._lib.rcflow_stream_flow_ptr(self._h, stream, C.byref(d), None, None))
        return _alias_tensor(d.value, h * w * 2, torch.float32, self.device).view(h, w, 2)

    def frame_buffer(self, w, h, stream=0):
        """The next page-locked staging buffer of the slot as a numpy HxW uint8 view (rcflow_frame_buffer_acquire): produce
        the frame into it, then push_frame_acquired() -- the frame loop without the staging copy.  The view is valid until
        that push; a later acquire with a LARGER frame reallocates the slot's staging memory, so never keep a view across
        a size change (take a new one per frame, as the loop does anyway)."""
        ptr, step = C.c_void_p(), C.c_size_t()
        check(self._lib.rcflow_frame_buffer_acquire(self._h, stream, w, h, C.byref(ptr), C.byref(step)))
        buf = (C.c_uint8 * (h * step.value)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h, step.value)[:, :w]

    def push_frame_acquired(self, w, h, stream=0, **kw):
        p = _params(kw.get("pyr_scale", 0.5), kw.get("levels", 2), kw.get("winsize", 3),
                    kw.get("iterations", 2), kw.get("poly_n", 15), kw.get("poly_sigma", 1.2),
                    kw.get("flags", 0))
        self._bind(stream)
        rc = check(self._lib.rcflow_push_frame_acquired(self._h, stream, C.byref(p)))
        if rc == 1:
            return None
        d = C.c_void_p()
        check(self._lib.rcflow_stream_flow_ptr(self._h, stream, C.byref(d), None, None))
        return _alias_tensor(d.value, h * w * 2, torch.float32, self.device).view(h, w, 2)

    def frame_loop_step(self, w, h, seeds=None, outmask=None, edges=None, dt=2.0, iterations=1, seed_dt=2.0,
                        seed_iterations=1, seed_upper=100.0, seed_variant=3, MID=0.5, LOWER=0.2, use_graph=False,
                        stream=0, **kw):
        """One iteration of the reference's frame loop (ripcurrents.cpp:194-479) on the frame produced into frame_buffer():
        flow against the previous frame, streamline_field, seed streamlines, histogram + thresholds, classify / accumulate
        (framecount counted on the device), mask edges -- rcflow_frame_loop_step.  Returns the resident flow field, or
        None for the call that primes the stream."""
        from ._lib import FrameLoop
        p = _params(kw.get("pyr_scale", 0.5), kw.get("levels", 2), kw.get("winsize", 3), kw.get("iterations_flow", 2),
                    kw.get("poly_n", 15), kw.get("poly_sigma", 1.2), kw.get("flags", 0))
        L = FrameLoop()
        L.dt, L.iterations = dt, iterations
        if seeds is not None:
            L.d_seeds, L.nseeds = self._ptr(seeds), seeds.shape[0]
        L.seed_variant, L.seed_dt, L.seed_iterations, L.seed_upper = seed_variant, seed_dt, seed_iterations, seed_upper
        L.MID, L.LOWER = MID, LOWER
        if outmask is not None:
            L.d_outmask, L.mask_step = self._ptr(outmask), outmask.stride(0)
        if edges is not None:
            L.d_edges, L.edges_step = self._ptr(edges), edges.stride(0)
        L.use_graph = 1 if use_graph else 0
        self._bind(stream)
        rc = check(self._lib.rcflow_frame_loop_step(self._h, stream, C.byref(p), C.byref(L)))
        if rc == 1:
            return None
        d = C.c_void_p()
        check(self._lib.rcflow_stream_flow_ptr(self._h, stream, C.byref(d), None, None))
        return _alias_tensor(d.value, h * w * 2, torch.float32, self.device).view(h, w, 2)

    def stream_flow_read(self, w, h, stream=0):
        out = np.empty((h, w, 2), np.float32)
        self._bind(stream)
        check(self._lib.rcflow_stream_flow_read(self._h, stream, out.ctypes.data, out.strides[0]))
        return out

    def push_clip(self, frames, flows=None, stream=0, **kw):
        """The next [T,H,W] frames of the slot's stream in one call (rcflow_push_clip_dev): every frame is
        expanded once however the segment is cut into calls.  Returns the flow fields written ([T,H,W,2] when
        the stream was primed -- flow 0 runs from the previous call's last frame to frames[0] -- else [T-1,...])."""
        frames = self._dev(frames, torch.uint8)
        T, h, w = frames.shape
        p = _params(kw.get("pyr_scale", 0.5), kw.get("levels", 2), kw.get("winsize", 3),
                    kw.get("iterations", 2), kw.get("poly_n", 15), kw.get("poly_sigma", 1.2),
                    kw.get("flags", 0))
        if flows is None:
            flows = torch.empty((T, h, w, 2), dtype=torch.float32, device=frames.device)
        self._bind(stream)
        n = check(self._lib.rcflow_push_clip_dev(
            self._h, stream, self._ptr(frames), frames.stride(0), frames.stride(1), T, w, h,
            self._ptr(flows), flows.stride(0) * 4, flows.stride(1) * 4, C.byref(p)))
        return flows[:n]

    def stream_reset(self, stream=0):
        check(self._lib.rcflow_stream_reset(self._h, stream))

    def farneback_clip(self, frames, flows=None, stream=0, pyr_scale=0.5, levels=2, winsize=3, iterations=2,
                       poly_n=15, poly_sigma=1.2, flags=0):
        """[T,H,W] uint8 CUDA clip -> [T-1,H,W,2] flows (pair t = frames t, t+1)."""
        frames = self._dev(frames, torch.uint8)
        T, h, w = frames.shape
        if flows is None:
            flows = torch.empty((T - 1, h, w, 2), dtype=torch.float32, device=frames.device)
        p = _params(pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)
        self._bind(stream)
        check(self._lib.rcflow_farneback_clip_dev(
            self._h, stream, self._ptr(frames), frames.stride(0), frames.stride(1), T, w, h,
            self._ptr(flows), flows.stride(0) * 4, flows.stride(1) * 4, C.byref(p)))
        return flows

    def push_batch(self, frames, flows=None, use_graph=True, stream=0, **kw):
        """Lockstep batch of streams: frames [S,H,W] uint8 -> flows [S,H,W,2] (None on the priming call)."""
        frames = self._dev(frames, torch.uint8)
        S, h, w = frames.shape
        p = _params(kw.get("pyr_scale", 0.5), kw.get("levels", 2), kw.get("winsize", 3), kw.get("iterations", 2),
                    kw.get("poly_n", 15), kw.get("poly_sigma", 1.2), kw.get("flags", 0))
        if flows is None:
            flows = torch.empty((S, h, w, 2), dtype=torch.float32, device=frames.device)
        self._bind(stream)
        rc = check(self._lib.rcflow_push_batch_dev(self._h, stream, self._ptr(frames), frames.stride(0), frames.stride(1),
                                                   S, w, h, self._ptr(flows), flows.stride(0) * 4, flows.stride(1) * 4,
                                                   C.byref(p), 1 if use_graph else 0))
        return None if rc == 1 else flows

    def batch_reset(self, stream=0):
        check(self._lib.rcflow_batch_reset(self._h, stream))

    def level_geometry(self, w, h, pyr_scale, levels, k):
        wk, hk = C.c_int(), C.c_int()
        L = check(self._lib.rcflow_level_geometry(w, h, pyr_scale, levels, k, C.byref(wk), C.byref(hk)))
        return L, wk.value, hk.value

    # stage-level entry points (parity tests)
    def stage_pyr_level(self, img, pyr_scale, k, stream=0):
        img = self._dev(img, torch.uint8)
        h, w = img.shape
        _, wk, hk = self.level_geometry(w, h, pyr_scale, 64, k)
        out = torch.empty((hk, wk), dtype=torch.float32, device=self.device)
        self._bind(stream)
        check(self._lib.rcflow_stage_pyr_level_dev(self._h, stream, self._ptr(img), img.stride(0), w, h,
                                                   pyr_scale, k, self._ptr(out)))
        return out

    def stage_polyexp(self, I, poly_n=15, poly_sigma=1.2, stream=0):
        I = self._dev(I, torch.float32).contiguous()
        h, w = I.shape
        R = torch.empty((h, w, 5), dtype=torch.float32, device=self.device)
        self._bind(stream)
        check(self._lib.rcflow_stage_polyexp_dev(self._h, stream, self._ptr(I), w, h, poly_n, poly_sigma,
                                                 self._ptr(R)))
        return R

    def stage_flow_iter(self, R0, R1, flow_in, winsize, flags, stream=0):
        R0 = self._dev(R0, torch.float32).contiguous()
        R1 = self._dev(R1, torch.float32).contiguous()
        h, w = R0.shape[:2]
        fin = None if flow_in is None else self._dev(flow_in, torch.float32).contiguous()
        out = torch.empty((h, w, 2), dtype=torch.float32, device=self.device)
        self._bind(stream)
        check(self._lib.rcflow_stage_flow_iter_dev(self._h, stream, self._ptr(R0), self._ptr(R1),
                                                   None if fin is None else self._ptr(fin), w, h, winsize,
                                                   flags, self._ptr(out)))
        return out

    # ------------------------------------------------------------------ B: analysis
    def analysis_reset(self, w, h, stream=0):
        self._bind(stream)
        check(self._lib.rcflow_analysis_reset(self._h, stream, w, h))

    def _flow(self, flow):
        flow = self._dev(flow, torch.float32)
        if flow.dim() != 3 or flow.shape[2] != 2 or flow.stride(2) != 1 or flow.stride(1) != 2:
            flow = flow.contiguous()
        return flow

    def create_histogram(self, current, st=None, stream=0):
        """create_histogram(current, hist, histsum, hist2d, histsum2d, UPPER, UPPER2d, prop_above_upper).

        `current` is the flow field (HxWx2); the polar conversion the reference does first
        (ripcurrents.cpp:305-309) is fused into the kernel.  The cumulative counters live in the
        slot; `st` (HistState) receives a host copy, like the reference's in/out arrays.
        """
        flow = self._flow(current)
        h, w = flow.shape[:2]
        self._bind(stream)
        check(self._lib.rcflow_histogram_dev(self._h, stream, self._ptr(flow), flow.stride(0) * 4, w, h))
        self._bind(stream)
        check(self._lib.rcflow_thresholds_dev(self._h, stream))
        if st is not None:
            self.histogram_read(st, stream)
        return st

    def histogram_accumulate(self, current, stream=0):
        flow = self._flow(current)
        h, w = flow.shape[:2]
        self._bind(stream)
        check(self._lib.rcflow_histogram_dev(self._h, stream, self._ptr(flow), flow.stride(0) * 4, w, h))

    def histogram_accumulate_clip(self, flows, stream=0):
        """Counts of a whole segment's flow fields ([T,H,W,2]) in one launch."""
        T, h, w = flows.shape[:3]
        self._bind(stream)
        check(self._lib.rcflow_histogram_clip_dev(self._h, stream, self._ptr(flows), flows.stride(0) * 4,
                                                  flows.stride(1) * 4, T, w, h))

    def thresholds(self, stream=0):
        self._bind(stream)
        check(self._lib.rcflow_thresholds_dev(self._h, stream))

    def thresholds_from_words(self, words, stream=0):
        """UPPER / UPPER2d / prop_above_upper from a device block of histogram words (e.g. the
        all-reduced global histogram); the slot's own counters stay as they are."""
        wds = self._dev(words, torch.int32).contiguous()
        if wds.numel() != HIST_WORDS:
            raise ValueError("expected %d histogram words" % HIST_WORDS)
        self._bind(stream)
        check(self._lib.rcflow_thresholds_words_dev(self._h, stream, self._ptr(wds)))

    def histogram_read(self, st=None, stream=0):
        st = st or HistState()
        hs, up = C.c_int32(), C.c_float()
        self._bind(stream)
        check(self._lib.rcflow_histogram_read(
            self._h, stream, st.hist.ctypes.data, st.hist2d.ctypes.data, C.addressof(hs),
            st.histsum2d.ctypes.data, C.addressof(up), st.UPPER2d.ctypes.data,
            st.prop_above_upper.ctypes.data))
        st.histsum, st.UPPER = hs.value, up.value
        return st

    def histogram_words(self, stream=0):
        """The RC_HIST_WORDS int32 block as a CUDA tensor aliasing the slot's counters
        (what torch.distributed.all_reduce sums across ranks)."""
        p = C.c_void_p()
        self._bind(stream)
        check(self._lib.rcflow_histogram_device_ptr(self._h, stream, C.byref(p)))
        return _alias_tensor(p.value, HIST_WORDS, torch.int32, self.device)

    def histogram_write(self, words, stream=0):
        words = np.ascontiguousarray(words, np.int32)
        assert words.size == HIST_WORDS
        self._bind(stream)
        check(self._lib.rcflow_histogram_write(self._h, stream, words.ctypes.data))

    def histogram_reset(self, stream=0):
        """Starts a new segment: zeroes the cumulative counters (asynchronous, on the slot's stream)."""
        self._bind(stream)
        check(self._lib.rcflow_histogram_reset_dev(self._h, stream))

    # ------------------------------------------------------------------ multi-GPU: global histogram (C ABI over RCCL)
    def comm_unique_id(self):
        """rank 0: the RCCL unique id (bytes) the host distributes to the other ranks."""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        check(self._lib.rcflow_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id=None):
        check(self._lib.rcflow_comm_init(self._h, unique_id, rank, world))

    def comm_destroy(self):
        check(self._lib.rcflow_comm_destroy(self._h))

    def allreduce_hist(self, stream=0, out=None):
        """Starts the all-rank sum of the slot's histogram counters (asynchronous, on the collective's own
        stream); returns the device tensor that will hold the result once allreduce_hist_join() has ordered
        the slot's stream after it."""
        self._bind(stream)
        if out is None:
            check(self._lib.rcflow_allreduce_hist(self._h, stream, None))
            p = C.c_void_p()
            check(self._lib.rcflow_allreduce_hist_result(self._h, C.byref(p)))
            return _alias_tensor(p.value, HIST_WORDS, torch.int32, self.device)
        check(self._lib.rcflow_allreduce_hist(self._h, stream, self._ptr(out)))
        return out

    def allreduce_hist_join(self, stream=0):
        self._bind(stream)
        check(self._lib.rcflow_allreduce_hist_join(self._h, stream))

    def allreduce_hist_status(self):
        """Host wait for the collective started last, then the verdict every rank shares (rcflow_allreduce_hist_status):
        raises RcflowError(RC_ESTATE) when the ranks together counted more pixels than an int32 histsum holds; returns the
        upper bound of the pixels counted otherwise."""
        n = C.c_longlong(0)
        check(self._lib.rcflow_allreduce_hist_status(self._h, C.byref(n)))
        return n.value

    def create_flow_accumulate(self, current, framecount, MID=0.5, LOWER=0.2, want=("polar", "waterclass",
                               "out", "outmask"), stream=0):
        """create_flow + create_accumulationbuffer (ripcurrents_module.cpp:153-212) in one pass.
        Returns a dict of the requested device outputs."""
        flow = self._flow(current)
        h, w = flow.shape[:2]
        outs = {}
        def mk(name, shape, dt):
            if name in want:
                outs[name] = torch.empty(shape, dtype=dt, device=self.device)
                return self._ptr(outs[name]), outs[name].stride(0) * outs[name].element_size()
            return None, 0
        pp, ps = mk("polar", (h, w, 3), torch.float32)
        wp, ws = mk("waterclass", (h, w, 3), torch.float32)
        op, os_ = mk("out", (h, w, 3), torch.float32)
        mp, ms = mk("outmask", (h, w), torch.uint8)
        self._bind(stream)
        check(self._lib.rcflow_classify_accumulate_dev(
            self._h, stream, self._ptr(flow), flow.stride(0) * 4, w, h, framecount, MID, LOWER, pp, ps, wp, ws,
            op, os_, mp, ms))
        return outs

    def accumulator(self, w, h, stream=0):
        acc = np.empty((h, w), np.float32)
        self._bind(stream)
        check(self._lib.rcflow_accumulator_read(self._h, stream, acc.ctypes.data))
        return acc

    def streamline_field(self, flow, dt, iterations, UPPER=-1.0, stream=0):
        """streamlines_mat.forEach(streamline_field(...)) ripcurrents.cpp:229-231; state in the slot."""
        flow = self._flow(flow)
        h, w = flow.shape[:2]
        self._bind(stream)
        check(self._lib.rcflow_advect_field_dev(self._h, stream, self._ptr(flow), flow.stride(0) * 4, w, h, dt,
                                                iterations, UPPER))

    def streamline_field_state(self, w, h, stream=0):
        pt = np.empty((h, w, 2), np.float32)
        dist = np.empty((h, w), np.float32)
        self._bind(stream)
        check(self._lib.rcflow_advect_field_read(self._h, stream, pt.ctypes.data, dist.ctypes.data))
        return pt, dist

    def streamline(self, pts, flow, dt, iterations, UPPER, variant=0, trace=False, stream=0):
        """Seed loops over streamline()/streamline_2()/streamline_3() (variant 0/1/2),
        ripcurrents.cpp's copy (3) and pathlines.cpp (4).  pts: n x 2, returns (pts, trace)."""
        flow = self._flow(flow)
        h, w = flow.shape[:2]
        d_pts = self._dev(np.ascontiguousarray(pts, np.float32) if not _is_t(pts) else pts, torch.float32).contiguous()
        n = d_pts.shape[0]
        iters = 100 if variant == 2 else iterations
        tr = torch.zeros((n, iters, 2), dtype=torch.float32, device=self.device) if trace else None
        self._bind(stream)
        check(self._lib.rcflow_advect_points_dev(
            self._h, stream, self._ptr(d_pts), n, self._ptr(flow), flow.stride(0) * 4, w, h, dt, iterations,
            UPPER, variant, None if tr is None else self._ptr(tr)))
        return d_pts, tr

    def get_delta_field(self, pt, flow, dt, UPPER, stream=0):
        flow = self._flow(flow)
        h, w = flow.shape[:2]
        pt = self._dev(pt, torch.float32).contiguous()
        self._bind(stream)
        check(self._lib.rcflow_get_delta_field_dev(self._h, stream, self._ptr(pt), pt.stride(0) * 4,
                                                   self._ptr(flow), flow.stride(0) * 4, w, h, dt, UPPER))
        return pt

    def _postop(self, fn, current, stream):
        flow = self._flow(current)
        h, w = flow.shape[:2]
        self._bind(stream)
        check(fn(self._h, stream, self._ptr(flow), flow.stride(0) * 4, w, h))
        return flow

    def subtructAverage(self, current, stream=0):
        return self._postop(self._lib.rcflow_subtract_average_dev, current, stream)

    def subtructMeanMagnitude(self, current, stream=0):
        return self._postop(self._lib.rcflow_subtract_mean_magnitude_dev, current, stream)

    def stabilizer(self, current, stream=0):
        return self._postop(self._lib.rcflow_stabilizer_dev, current, stream)

    def window_mean(self, avg, slot, cur, window, stream=0):
        self._bind(stream)
        check(self._lib.rcflow_window_mean_dev(self._h, stream, self._ptr(avg), self._ptr(slot), self._ptr(cur),
                                               avg.numel(), window))

    def vectorToColor(self, current, max_displacement, stream=0):
        flow = self._flow(current)
        h, w = flow.shape[:2]
        hsv = torch.zeros((h, w, 3), dtype=torch.uint8, device=self.device)
        md = C.c_float(max_displacement)
        self._bind(stream)
        check(self._lib.rcflow_vector_to_color_dev(self._h, stream, self._ptr(flow), flow.stride(0) * 4, w, h,
                                                   self._ptr(hsv), hsv.stride(0), C.byref(md)))
        return hsv, md.value

    def shearRateToColor(self, current, max_frobenius, hsv=None, stream=0):
        flow = self._flow(current)
        h, w = flow.shape[:2]
        if hsv is None:
            hsv = torch.zeros((h, w, 3), dtype=torch.uint8, device=self.device)
        mf = C.c_float(max_frobenius)
        self._bind(stream)
        check(self._lib.rcflow_shear_rate_to_color_dev(self._h, stream, self._ptr(flow), flow.stride(0) * 4, w, h,
                                                       self._ptr(hsv), hsv.stride(0), C.byref(mf)))
        return hsv, mf.value

    # ------------------------------------------------------------------ SURVEY 8(f) next rows
    def create_edges(self, outmask, stream=0):
        """create_edges(outmask) ripcurrents_module.cpp:216-220: returns the edge mask."""
        m = self._dev(outmask, torch.uint8)
        if m.stride(1) != 1:
            m = m.contiguous()
        h, w = m.shape
        out = torch.empty((h, w), dtype=torch.uint8, device=self.device)
        self._bind(stream)
        check(self._lib.rcflow_create_edges_dev(self._h, stream, self._ptr(m), m.stride(0), w, h, self._ptr(out),
                                                out.stride(0)))
        return out

    def create_output(self, subframe, outmask, stream=0):
        """create_output(subframe, outmask) ripcurrents_module.cpp:225-244: paints the edge mask into the
        red channel of the 8UC3 frame, in place when `subframe` is already a device tensor; returns it."""
        f = self._dev(subframe, torch.uint8).contiguous()
        m = self._dev(outmask, torch.uint8).contiguous()
        h, w = m.shape
        if tuple(f.shape) != (h, w, 3):
            raise ValueError("subframe must be HxWx3 uint8 of the mask's size")
        self._bind(stream)
        check(self._lib.rcflow_create_output_dev(self._h, stream, self._ptr(f), f.stride(0), self._ptr(m), m.stride(0), w, h))
        return f

    def resize_bgr_to_gray(self, frame, dw, dh, stream=0, interpolation="linear"):
        """resize(frame, Size(dw,dh), INTER_LINEAR) + cvtColor(BGR2GRAY) (ripcurrents.cpp:209-210);
        interpolation="area": INTER_AREA, as the reference resizes the first frame (ripcurrents.cpp:186)."""
        f = self._dev(frame, torch.uint8).contiguous()
        sh, sw = f.shape[:2]
        out = torch.empty((dh, dw), dtype=torch.uint8, device=self.device)
        self._bind(stream)
        fn = self._lib.rcflow_resize_area_bgr_to_gray_dev if interpolation == "area" else self._lib.rcflow_resize_bgr_to_gray_dev
        check(fn(self._h, stream, self._ptr(f), f.stride(0), sw, sh, self._ptr(out), out.stride(0), dw, dh))
        return out

    def _an_size(self, stream):
        w, h = C.c_int(0), C.c_int(0)
        check(self._lib.rcflow_analysis_size(self._h, stream, C.byref(w), C.byref(h)))
        if w.value <= 0 or h.value <= 0:
            raise RuntimeError("the slot has no analysis state yet")
        return w.value, h.value

    def streamline_display(self, which, stream=0):
        """streamline_displacement (0) / _total_motion (1) / _ratio (2), ripcurrents_module.cpp:13-40, on the
        slot's streamline field: returns (8UC3 BGR image on the device, the minMaxLoc maximum)."""
        w, h = self._an_size(stream)
        out = torch.empty((h, w, 3), dtype=torch.uint8, device=self.device)
        mx = C.c_float(0)
        self._bind(stream)
        check(self._lib.rcflow_streamline_display_dev(self._h, stream, int(which), self._ptr(out), out.stride(0),
                                                      C.byref(mx)))
        return out, mx.value

    def streamline_positions(self, stream=0):
        """streamline_positions ripcurrents_module.cpp:44-60: 32FC3 image, (1,1,1) where particles sit."""
        w, h = self._an_size(stream)
        out = torch.zeros((h, w, 3), dtype=torch.float32, device=self.device)
        self._bind(stream)
        check(self._lib.rcflow_streamline_positions_dev(self._h, stream, self._ptr(out), out.stride(0) * 4))
        return out

    def hsv_to_bgr(self, hsv, stream=0):
        """cvtColor(current, current, CV_HSV2BGR) on the 32FC3 display image (ripcurrents.cpp:405)."""
        a = self._dev(hsv, torch.float32).contiguous()
        h, w = a.shape[:2]
        out = torch.empty_like(a)
        self._bind(stream)
        check(self._lib.rcflow_hsv_to_bgr_dev(self._h, stream, self._ptr(a), a.stride(0) * 4, w, h, self._ptr(out),
                                              out.stride(0) * 4))
        return out

    def jet_lut(self):
        lut = np.zeros((256, 3), np.uint8)
        check(self._lib.rcflow_jet_lut(lut.ctypes.data))
        return lut

    def calcOpticalFlowPyrLK(self, prev, nxt, prev_pts, next_pts=None, win=(21, 21), max_level=3,
                             crit_type=3, max_count=30, epsilon=0.01, flags=0, min_eig_threshold=1e-4, stream=0):
        """cv::calcOpticalFlowPyrLK on 8UC1 images (Streakline.cpp:32, ripcurrents_module.cpp:716,738,775,
        1162).  crit_type: 1 = COUNT, 2 = EPS; flags: 4 = OPTFLOW_USE_INITIAL_FLOW, 8 =
        OPTFLOW_LK_GET_MIN_EIGENVALS.  Returns (next_pts [n,2] f32, status [n] u8, err [n] f32) on the device."""
        a = self._dev(prev, torch.uint8)
        b = self._dev(nxt, torch.uint8)
        if a.stride(1) != 1:
            a = a.contiguous()
        if b.stride(1) != 1:
            b = b.contiguous()
        h, w = a.shape
        if tuple(b.shape) != (h, w):
            raise ValueError("prev and next differ in size")
        p = self._dev(prev_pts, torch.float32).reshape(-1, 2).contiguous()
        n = p.shape[0]
        if next_pts is None:
            q = torch.zeros((n, 2), dtype=torch.float32, device=self.device)
        else:
            q = self._dev(next_pts, torch.float32).reshape(-1, 2).contiguous().clone()
        status = torch.zeros((n,), dtype=torch.uint8, device=self.device)
        err = torch.zeros((n,), dtype=torch.float32, device=self.device)
        self._bind(stream)
        check(self._lib.rcflow_pyrlk_dev(self._h, stream, self._ptr(a), a.stride(0), self._ptr(b), b.stride(0), w, h,
                                         self._ptr(p), self._ptr(q), n, self._ptr(status), self._ptr(err),
                                         int(win[0]), int(win[1]), int(max_level), int(crit_type), int(max_count),
                                         float(epsilon), int(flags), float(min_eig_threshold)))
        return q, status, err

    def calcOpticalFlowPyrLK_host(self, prev, nxt, prev_pts, win=(21, 21), max_level=3, crit_type=3, max_count=30,
                                  epsilon=0.01, flags=0, min_eig_threshold=1e-4, stream=0):
        """Host-pointer form (rcflow_pyrlk_u8): numpy in, numpy out, blocking."""
        a = np.ascontiguousarray(prev, np.uint8)
        b = np.ascontiguousarray(nxt, np.uint8)
        h, w = a.shape
        p = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        n = p.shape[0]
        q = np.zeros((n, 2), np.float32)
        status = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self._bind(stream)
        check(self._lib.rcflow_pyrlk_u8(self._h, stream, a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], w, h,
                                        p.ctypes.data, q.ctypes.data, n, status.ctypes.data, err.ctypes.data,
                                        int(win[0]), int(win[1]), int(max_level), int(crit_type), int(max_count),
                                        float(epsilon), int(flags), float(min_eig_threshold)))
        return q, status, err

    def pyrlk_levels(self, w, h, win, max_level):
        return check(self._lib.rcflow_pyrlk_levels(w, h, int(win[0]), int(win[1]), int(max_level)))

    # ------------------------------------------------------------------ measurement
    def profile_enable(self, on=True):
        check(self._lib.rcflow_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(self._lib.rcflow_profile_reset(self._h))

    def profile_read(self):
        cap = 256
        names = (C.c_char_p * cap)()
        launches = (C.c_int * cap)()
        ms = (C.c_double * cap)()
        by = (C.c_double * cap)()
        mb = (C.c_double * cap)()
        n = check(self._lib.rcflow_profile_read(self._h, cap, names, launches, ms, by, mb))
        return [dict(kernel=names[i].decode(), launches=launches[i], total_ms=ms[i], alg_bytes=by[i],
                     model_bytes=mb[i]) for i in range(n)]

    def profile_read_buckets(self):
        """GPU time per bucket of the reference's own timing printout (ripcurrents.cpp:518-524):
        {"farneback": ms, "polar": ms, "threshold": ..., "overlay", "erosion", "codec", "stream"}."""
        names = (C.c_char_p * 7)()
        ms = (C.c_double * 7)()
        n = check(self._lib.rcflow_profile_read_buckets(self._h, names, ms))
        return {names[i].decode(): ms[i] for i in range(n)}


def _alias_tensor(ptr, n, dtype, device):
    """A torch tensor aliasing `n` elements of device memory the library owns."""
    itemsize = torch.empty((), dtype=dtype).element_size()

    class _Holder:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i%d" % itemsize if dtype != torch.float32 else "<f4",
                                    "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Holder(), device=device)


class Streakline:
    """Streakline.hpp:8-20 / Streakline.cpp:11-71 with the vertices moved through the dense
    flow field (the compute_timelinesFarne precedent, main.cpp:961-977) instead of sparse LK.
    Fields as in the reference: generationPoint, vertices, numberOfVertices, frameCount."""

    def __init__(self, pixel):
        self.generationPoint = (float(pixel[0]), float(pixel[1]))
        self.vertices = [self.generationPoint]
        self.numberOfVertices = 1
        self.frameCount = 1

    def run(self, ctx, flow, width, height, dt=1.0, stream=0):
        """runLK's bookkeeping: move every vertex, reject jumps > 0.1*dim (Streakline.cpp:35-40),
        insert the generation point in front (:46-48)."""
        v = np.asarray(self.vertices, np.float32).reshape(-1, 2)
        # variant 4 = `p += delta*dt/iterations` with no cutoff; one step
        moved, _ = ctx.streamline(v, flow, dt, 1, 0.0, variant=4, stream=stream)
        nxt = moved.cpu().numpy()
        big = (np.abs(v[:, 0] - nxt[:, 0]) > width * 0.1) | (np.abs(v[:, 1] - nxt[:, 1]) > height * 0.1)
        nxt[big] = v[big]
        self.vertices = [self.generationPoint] + [tuple(map(float, p)) for p in nxt]
        self.numberOfVertices = len(self.vertices)
        self.frameCount += 1
        return self.vertices

    def runLK(self, ctx, u_prev, u_current, stream=0):
        """Streakline::runLK (Streakline.cpp:22-71) with the reference's own mover: PyrLK 50x50, maxLevel 3,
        COUNT+EPS (30, 0.1), flags 10, minEigThreshold 1e-4 (:32); XDIM/YDIM are the frame size."""
        height, width = u_prev.shape[:2]
        v = np.asarray(self.vertices, np.float32).reshape(-1, 2)
        q, _, _ = ctx.calcOpticalFlowPyrLK(u_prev, u_current, v, win=(50, 50), max_level=3, crit_type=3,
                                           max_count=30, epsilon=0.1, flags=10, min_eig_threshold=1e-4,
                                           stream=stream)
        nxt = q.cpu().numpy()
        big = (np.abs(v[:, 0] - nxt[:, 0]) > width * 0.1) | (np.abs(v[:, 1] - nxt[:, 1]) > height * 0.1)
        nxt[big] = v[big]
        self.vertices = [self.generationPoint] + [tuple(map(float, p)) for p in nxt]
        self.numberOfVertices = len(self.vertices)
        self.frameCount += 1
        return self.vertices


def _run_lk_all(ctx, vertices, u_prev, u_current, stream=0):
    """The PyrLK call shared by Timeline::runLK and PopulationMap::runLK (ripcurrents_module.cpp:775,
    :1162): 50x50 window, maxLevel 3, COUNT+EPS (30, 0.1), flags 10, minEigThreshold 1e-4; every vertex
    takes its tracked position (the jump rejection is commented out in the reference)."""
    v = np.asarray(vertices, np.float32).reshape(-1, 2)
    q, _, _ = ctx.calcOpticalFlowPyrLK(u_prev, u_current, v, win=(50, 50), max_level=3, crit_type=3, max_count=30,
                                       epsilon=0.1, flags=10, min_eig_threshold=1e-4, stream=stream)
    return [tuple(map(float, p)) for p in q.cpu().numpy()]


class Timeline:
    """Timeline (ripcurrents.hpp:64-75, ripcurrents_module.cpp:751-807): numberOfVertices + 1 points on
    the segment lineStart..lineEnd, moved by sparse PyrLK every frame; drawing stays with the caller."""

    def __init__(self, lineStart, lineEnd, numberOfVertices):
        diffX = np.float32(np.float32(lineEnd[0] - lineStart[0]) / np.float32(numberOfVertices))
        diffY = np.float32(np.float32(lineEnd[1] - lineStart[1]) / np.float32(numberOfVertices))
        self.vertices = [(float(np.float32(lineStart[0]) + diffX * np.float32(i)),
                          float(np.float32(lineStart[1]) + diffY * np.float32(i))) for i in range(numberOfVertices + 1)]

    def runLK(self, ctx, u_prev, u_current, stream=0):
        self.vertices = _run_lk_all(ctx, self.vertices, u_prev, u_current, stream)
        return self.vertices


class PopulationMap:
    """PopulationMap (ripcurrents.hpp:86-95, ripcurrents_module.cpp:1140-1196): random points
    rectStart + (rectEnd - rectStart) * (u + 1), u uniform in [0, 1] -- the reference's formula,
    which lands them in the rectangle mirrored beyond rectEnd (`rand()/RAND_MAX + 1`) -- moved by
    sparse PyrLK.  `rng` replaces the reference's sranddev()/rand() (not reproducible by design)."""

    def __init__(self, rectStart, rectEnd, numberOfVertices, rng=None):
        rng = rng or np.random.RandomState()
        self.vertices = []
        for _ in range(numberOfVertices):
            randX = np.float32((rectEnd[0] - rectStart[0]) * (rng.uniform(0.0, 1.0) + 1) + rectStart[0])
            randY = np.float32((rectEnd[1] - rectStart[1]) * (rng.uniform(0.0, 1.0) + 1) + rectStart[1])
            self.vertices.append((float(randX), float(randY)))

    def runLK(self, ctx, u_prev, u_current, stream=0):
        self.vertices = _run_lk_all(ctx, self.vertices, u_prev, u_current, stream)
        return self.vertices
