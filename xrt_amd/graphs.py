"""Whole iterations of a ray-tracing loop as HIP graphs.

A ``run_process`` over small beams (1e4 .. 1e6 rays) is bound by the host: every element call is
~30-90 us of Python + ctypes for a few microseconds of kernels. The kernels of one iteration --
ray generator, reflect passes, screens, histograms -- take no decision on the host (the optimistic
pass and its fall-back are decided on the device), so the iteration can be recorded
once into a HIP graph (``torch.cuda.graph`` = hipStreamBeginCapture on the stream the C ABI is
given) and replayed with one launch per iteration.

What a replay cannot do is anything the HOST did while the graph was recorded: numpy random
numbers (host ray generators, random diffraction orders, wave sampling), uploads of host
arrays, read-backs that steer the Python code (automatic plot limits). Uploads and read-backs
fail inside a capture by themselves (HIP refuses them on a capturing stream); the sites that
would merely repeat a host decision ask :func:`refuse`. The device ray generator takes its call
number from a device cell that the graph increments as its last node
(``xrt_hip_geosource.call_dev``), so that replay k draws the rays the k-th eager call would have
drawn -- wherever in the iteration the rays are made (the generator's own launch, the head of an
element's pass, a redo).

Host-side counters of an iteration (rays seen by a plot, calls of a source) are not advanced
while recording; they are handed to :func:`per_iteration` and run after every replay.
"""
import gc
import threading

import torch

_tls = threading.local()


class CaptureError(RuntimeError):
    """This iteration cannot be replayed from a HIP graph."""


def capturing():
    """The IterationGraph being recorded by this thread, or None."""
    return getattr(_tls, 'recording', None)


def refuse(what):
    """Called where the host takes a per-iteration decision a replay would repeat."""
    if capturing() is not None:
        raise CaptureError('%s: not possible in an iteration replayed from a HIP graph '
                           '(run_ray_tracing(graph=False))' % what)


def per_iteration(bookkeeping):
    """Host bookkeeping of one iteration: now (eager), or after every replay (recording)."""
    rec = capturing()
    if rec is None:
        bookkeeping()
    else:
        rec.after_replay.append(bookkeeping)


class IterationGraph(object):
    """``fn()`` recorded once, replayed by :meth:`replay`. *fn* must have run eagerly before
    (workspaces, tables and automatic limits exist; the first call of anything is not what a
    steady-state iteration looks like). The objects *fn* returned while recording are kept:
    their device arrays are what every replay overwrites."""

    def __init__(self, fn, keep_graph=False):
        self.after_replay = []
        self.before_end = []         # recorded after fn(): the last nodes of the graph
        self.pending_calls = {}      # source -> shine() calls recorded so far
        # (keep_graph: the hipGraph_t stays reachable -- kernel_nodes() counts its launches)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else \
            torch.cuda.CUDAGraph()
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        _tls.recording = self
        # no automatic garbage collection while recording: a collected tensor is harmless, a
        # collected graph of an earlier run is not (hipGraphDestroy on a capturing stream is an
        # error, thrown from a destructor: the process ends)
        collects = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.result = fn()
                for last in self.before_end:
                    last()
        except CaptureError:
            raise
        except Exception as e:          # noqa: BLE001  (HIP refuses syncs / uploads in a capture)
            raise CaptureError('this iteration cannot be recorded into a HIP graph (%s: %s); '
                               'run it with graph=False' % (type(e).__name__, e)) from e
        finally:
            _tls.recording = None
            if collects:
                gc.enable()
        self.replays = 0

    def kernel_nodes(self):
        """How many launches one iteration is (hipGraphGetNodes on the recorded graph: kernel,
        memcpy and memset nodes alike); needs ``keep_graph=True``."""
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_size_t)]
        count = ctypes.c_size_t(0)
        err = hip.hipGraphGetNodes(ctypes.c_void_p(self.graph.raw_cuda_graph()), None,
                                   ctypes.byref(count))
        if err:
            raise RuntimeError('hipGraphGetNodes: error %d' % err)
        return int(count.value)

    def close(self):
        """Drops the graph, the recorded beams and the hooks (which refer back to this object:
        without this the graph lives until some later garbage collection)."""
        self.after_replay = []
        self.before_end = []
        self.result = None      # (pending_calls stays: a recorded beam made later needs the count)
        self.graph = None

    def replay(self):
        for source in self.pending_calls:        # (eager calls in between move the call number)
            source._sync_replay_cell()
        self.graph.replay()
        self.replays += 1
        for bookkeeping in self.after_replay:
            bookkeeping()
        return self.result
