"""
Pipelined batch executor for the recognition path (no reference analogue: the reference runs one
synchronous ``nn(...)`` + host decode per batch, kraken/lib/vgsl/rpred.py:210-229).

A ``RecognitionEngine`` owns ``slots`` independent execution slots; each slot has its own
``krk_plan`` (weights are 6 MB -- replicating them is free on a 288 GB part), its own HIP stream,
device result buffers and pinned host mirrors.  ``submit`` enqueues forward + softmax + CTC
best-path decode (one ``krk_recognize`` call) and the device->host copy of the COMPACT label
tuples on the slot's stream and returns immediately; ``collect`` waits for that slot's event
only.  With two or more slots the small LSTM recurrent kernels of one batch (a few dozen CUs)
overlap the convolutions of the next, which is where single-batch latency leaves CUs idle.
"""
import ctypes as C
import os
from collections import deque
from typing import Optional

import numpy as np
import torch

from . import _lib
from .vgsl import DecodedBatch, TorchVGSLModel, _Plan


class _Slot:
    def __init__(self, model: TorchVGSLModel, dev: int, max_n: int, max_t: int):
        self.plan = _Plan(model.nn._specs, model.nn, model.input[1], model.input[2], dev, model.nn.precision)
        self.stream = torch.cuda.Stream(device=dev)
        self.event = torch.cuda.Event()
        d = torch.device(f'cuda:{dev}')
        # one int32 block [labels | starts | ends | conf bits | counts] so the D2H is a single copy
        self.max_n, self.max_t = max_n, max_t
        self.dev_buf = torch.empty(4 * max_n * max_t + max_n, dtype=torch.int32, device=d)
        self.host_buf = torch.empty(4 * max_n * max_t + max_n, dtype=torch.int32).pin_memory()
        self.olens = np.empty(max_n, dtype=np.int32)
        self.busy = False
        self.n = self.t = 0
        self.keep = None
        self.stage = None     # device copy of a host input batch


class RecognitionEngine:
    def __init__(self, model: TorchVGSLModel, device: int = 0, max_batch: int = 256, max_width: int = 2400,
                 slots: int = 2, temperature: float = 1.0):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.model = model
        self.device = device
        self.temperature = float(temperature)
        with torch.cuda.device(device):
            probe = _Plan(model.nn._specs, model.nn, model.input[1], model.input[2], device, model.nn.precision)
            self.classes, _, max_t = probe.out_shape(max_width)
            probe.close()
            self.max_batch, self.max_t = max_batch, max_t
            self.slots = [_Slot(model, device, max_batch, max_t) for _ in range(slots)]
        self._next = 0
        self._inflight = deque()
        # front event of the batch submitted last: the next batch's convolution block queues behind it, so the
        # full-chip convolution blocks of different batches run one after another (no convoy of all slots doing
        # convolutions together and then all doing recurrences together) -- see include/kraken_amd.h
        self._fronts = []
        self.chain_fronts = os.environ.get('KRK_NO_FRONT_CHAIN') is None

    def set_profiling(self, on: bool):
        for s in self.slots:
            _lib.check(self.lib.krk_plan_set_profiling(s.plan.handle, 1 if on else 0))

    def layer_times(self):
        """Per-slot list of (name, ms, flops) for the LAST batch each slot ran (profiling must be on)."""
        out = []
        for s in self.slots:
            n = self.lib.krk_plan_num_steps(s.plan.handle)
            ms = (C.c_float * n)()
            if min(self.lib.krk_plan_layer_ms(s.plan.handle, ms, n), 0) != 0:
                continue                      # this slot has not run a batch since profiling was switched on
            out.append([(self.lib.krk_plan_layer_name(s.plan.handle, i).decode(), float(ms[i]),
                         float(self.lib.krk_plan_layer_flops(s.plan.handle, i))) for i in range(n)])
        return out

    def submit(self, x: torch.Tensor, lens: Optional[np.ndarray] = None) -> int:
        """
        x: (N, C, H, W) float32 tensor, resident on this device -- or a (preferably pinned) HOST tensor, which is copied
        to a per-slot staging buffer on the slot's own stream so that the PCIe transfer of batch k+1 overlaps the kernels of
        batch k.  Returns a ticket.
        """
        slot_id = self._next
        slot = self.slots[slot_id]
        if slot.busy:
            raise RuntimeError('all slots busy: collect() a ticket before submitting more')
        if not x.is_cuda:
            if slot.stage is None or slot.stage.shape != x.shape:
                slot.stage = torch.empty(x.shape, dtype=torch.float32, device=f'cuda:{self.device}')
            with torch.cuda.stream(slot.stream):
                slot.stage.copy_(x, non_blocking=True)
            x = slot.stage
        N, _, _, W = x.shape
        _, _, T = slot.plan.out_shape(W)
        if N > slot.max_n or T > slot.max_t:
            raise ValueError(f'batch {N}x{T} exceeds the engine capacity {slot.max_n}x{slot.max_t}')
        nt = slot.max_n * slot.max_t
        base = slot.dev_buf.data_ptr()
        dec = _lib.KrkDecodeOut(base, base + 4 * nt, base + 8 * nt, base + 12 * nt, base + 16 * nt, slot.max_t)
        lens_arr = None
        if lens is not None:
            lens_arr = np.ascontiguousarray(np.asarray(lens, dtype=np.int32))
        cur = torch.cuda.current_stream(self.device)
        slot.stream.wait_stream(cur)   # the input may have been produced on the caller's stream
        lag = int(os.environ.get('KRK_FRONT_LAG', '1'))
        if self.chain_fronts and len(self._fronts) >= lag and len(self.slots) > 1:
            _lib.check(self.lib.krk_plan_wait_front(slot.plan.handle, self._fronts[-lag]))
        self._fronts.append(self.lib.krk_plan_front_event(slot.plan.handle))
        del self._fronts[:-4]
        with torch.cuda.stream(slot.stream):
            _lib.check(self.lib.krk_recognize(slot.plan.handle, x.data_ptr(),
                                              lens_arr.ctypes.data if lens_arr is not None else None, N, W,
                                              self.temperature, slot.stream.cuda_stream, None, None,
                                              slot.olens.ctypes.data, C.byref(dec)))
            slot.host_buf.copy_(slot.dev_buf, non_blocking=True)
            slot.event.record(slot.stream)
        slot.busy, slot.n, slot.t, slot.keep = True, N, T, x
        self._next = (slot_id + 1) % len(self.slots)
        self._inflight.append(slot_id)
        return slot_id

    def collect(self, ticket: Optional[int] = None) -> tuple[DecodedBatch, np.ndarray]:
        """Waits for the oldest (or the given) in-flight batch; returns (DecodedBatch, olens)."""
        if ticket is None:
            ticket = self._inflight[0]
        self._inflight.remove(ticket)
        slot = self.slots[ticket]
        slot.event.synchronize()
        _lib.check(self.lib.krk_plan_status(slot.plan.handle))     # a kernel that gave up waiting raises here, never hangs
        nt = slot.max_n * slot.max_t
        h = slot.host_buf.numpy()
        n, t = slot.n, slot.max_t
        view = lambda k: h[k * nt:(k + 1) * nt].reshape(slot.max_n, t)[:n]   # noqa: E731
        batch = DecodedBatch(view(0).copy(), view(1).copy(), view(2).copy(), view(3).copy().view(np.float32),
                             h[4 * nt:4 * nt + n].copy())
        slot.busy, slot.keep = False, None
        return batch, slot.olens[:n].copy()

    def free_slots(self) -> int:
        return sum(not s.busy for s in self.slots)

    def close(self):
        for s in self.slots:
            s.plan.close()
