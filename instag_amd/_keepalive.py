"""Lifetime of tensors that cross streams inside a stream-capture window.

Outside a capture ``tensor.record_stream(stream)`` tells the caching allocator that `stream` uses the tensor.  INSIDE a
capture the allocator cannot record the events that mechanism needs: it queues them and inserts them when the capture
ends -- and a block that was marked this way and then released while the capture was still open is what crashed
``hipStreamEndCapture`` (``torch.cuda.graphs.capture_end``, ROCm 7.2) intermittently, depending on when the cyclic
garbage collector happened to drop a finished step's tensors.  So inside a capture no tensor is marked: it is appended
to a list that the owner of the capture (GraphedStep / GraphedStage) clears after the capture has ended.  That is safe
because every side stream of a step is joined back into the capturing stream before the step ends, and memory of the
capture's private pool is only reused by allocations ordered behind that join.
"""
import torch

_HELD = []


def cross_stream(tensor, stream):
    """`tensor` (allocated on another stream) is used on `stream`."""
    if tensor is None:
        return
    if tensor.is_cuda and torch.cuda.is_current_stream_capturing():
        _HELD.append(tensor)
    else:
        tensor.record_stream(stream)


def release():
    """Called by the owner of a capture once the capture has ended."""
    _HELD.clear()
