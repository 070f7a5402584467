"""Gradients a backward node wrote IN PLACE from a side stream.

The text path runs on a stream of its own beside the VGG stack (model.UMPR._forward), and autograd replays that split in
backward.  Its Functions write their parameter gradients straight into FusedAdam's flat arena and return None for them
(model._grad_targets), so no AccumulateGrad node runs on the side stream - and the engine's end-of-backward synchronisation,
which only covers the streams of AccumulateGrad nodes, does not make the caller's stream wait for those writes.  Whoever reads
the gradient arena next (FusedAdam.step, the remainder all-reduce of parallel.GradReducer.finish / allreduce_arenas) must:

    note_gradients_written(device)   in the backward node, after its kernels are enqueued on the side stream
    wait_for_gradients(device)       in the consumer, before its first kernel: the current stream waits for every noted event

(tools/check_exchange_world1.py caught the missing wait: at batch 4 the optimiser step overtook the tail of the ReviewNet
backward - the GRU's reverse-direction gradients - and two runs of the same three steps differed by 1e-4.)"""
import os as _os
import threading

import torch

_PENDING = {}
_LOCK = threading.Lock()


def note_gradients_written(device):
    dev = torch.device(device)
    st = torch.cuda.current_stream(dev)
    ev = torch.cuda.Event()
    ev.record(st)
    with _LOCK:      # one event per stream is enough (a later event of a stream covers the earlier ones): no growth without a consumer
        _PENDING.setdefault(dev.index or 0, {})[st.cuda_stream] = ev


def wait_for_gradients(device):
    dev = torch.device(device)
    if dev.type != "cuda" or _os.environ.get("UMPR_DEBUG_NO_GRAD_WAIT") == "1":     # (A/B of what the wait costs; unsafe)
        return
    with _LOCK:
        evs = _PENDING.pop(dev.index or 0, None)
    if evs:
        cur = torch.cuda.current_stream(dev)
        for ev in evs.values():
            cur.wait_event(ev)
