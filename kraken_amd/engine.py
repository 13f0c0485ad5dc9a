"""
Pipelined batch executor for the recognition path (no reference analogue: the reference runs one
synchronous ``nn(...)`` + host decode per batch, kraken/lib/vgsl/rpred.py:210-229).

A ``RecognitionEngine`` owns ``slots`` independent execution slots; each slot has its own
``krk_plan`` (weights are 6 MB -- replicating them is free on a 288 GB part), its own HIP stream,
device result buffers and pinned host mirrors.  ``submit`` enqueues forward + softmax + CTC
best-path decode (one ``krk_recognize`` call) and the device->host copy of the COMPACT label
tuples on the slot's stream and returns immediately; ``collect`` waits for that slot's event
only.  With two or more slots the recurrent kernels of one batch (a quarter of the CUs) overlap
the convolutions of the next, and the host prepares / decodes other batches meanwhile.

Two ways to hand a batch over:

* ``submit(x, lens)`` -- ``x`` resident on the device (bench.py) or any host tensor;
* ``stage(n, w)`` -> pinned host array ``(n, C, H, w)`` of the next free slot, to be filled in place
  (line k at ``[k, :, :, :w_k]``, zeros to its right), then ``submit_staged(lens)``: what ``rpred`` /
  ``TorchVGSLModel.predict`` use, so that padding a batch is its only host copy.

Buffers grow on demand (batch size, width); nothing is sized by a guess that a page can exceed.
"""
import ctypes as C
import os
from collections import deque
from typing import Optional

import numpy as np
import torch

from . import _lib
from .vgsl import DecodedBatch, TorchVGSLModel


_HIP = None


def _hip_runtime():
    """ctypes handle of the HIP runtime THIS process already uses (torch ships its own copy: a dlopen by name could load a second one)."""
    global _HIP
    if _HIP is None:
        path = None
        try:
            with open('/proc/self/maps') as f:
                for ln in f:
                    if 'libamdhip64' in ln:
                        path = ln.split()[-1]
                        break
        except OSError:
            pass
        try:
            _HIP = C.CDLL(path) if path else False
        except OSError:
            _HIP = False
    return _HIP


class _Slot:
    def __init__(self, model: TorchVGSLModel, dev: int, twin: Optional['_Slot'] = None):
        # the first slot builds a plan (same f32 fallback -- and warning -- as a direct nn(x) call); the others share its packed
        # weights (krk_plan_clone: 7 ms of repack + upload per slot of kraken's default recogniser otherwise)
        # Round 6, later: the first slot is a clone too -- of the plan the MODEL keeps for this device and arithmetic (HipSequential.plan:
        # built when a recogniser is put on the device, TorchSeqRecognizer.to / prepare_for_inference, or by the first nn(x) call), so a
        # page's first pass does not pack and upload the weights again (12-15 ms of kraken's default recogniser).
        self.plan = model.nn.plan(dev).clone() if twin is None else twin.plan.clone()
        self.dev = dev
        self.stream = torch.cuda.Stream(device=dev)
        self.event = torch.cuda.Event()
        self.cap_n = self.cap_t = 0
        self.dev_buf = self.host_buf = None
        self.olens = np.empty(0, dtype=np.int32)
        self.busy = False
        self.n = self.t = 0
        self.keep = None
        self.stage_host = None     # pinned (flat) host staging of an input batch
        self.stage_dev = None      # its device copy
        self.staged = None         # (n, c, h, w) of the batch being staged
        self.probs = None          # (n, t, classes) softmax of the last batch, when asked for
        self.want_probs = False
        self.crops_host = self.crops_dev = None     # packed uint8 line crops (krk_prep_crops): pinned staging + device copy
        self.boxes_host = self.boxes_dev = None     # device-side line preprocessing (krk_prep_lines): crop boxes ...
        self.flags_dev = self.flags_host = None     # ... and the per-line "holds ink" flags it returns
        self.has_flags = False

    def ensure_results(self, n: int, t: int):
        """One int32 block [labels | starts | ends | conf bits | counts] so the D2H is a single copy."""
        if n <= self.cap_n and t <= self.cap_t:
            return
        self.cap_n, self.cap_t = max(n, self.cap_n), max(t, self.cap_t)
        d = torch.device(f'cuda:{self.dev}')
        size = 4 * self.cap_n * self.cap_t + self.cap_n
        self.dev_buf = torch.empty(size, dtype=torch.int32, device=d)
        self.host_buf = torch.empty(size, dtype=torch.int32).pin_memory()
        self.olens = np.empty(self.cap_n, dtype=np.int32)

    def ensure_stage(self, elems: int, host: bool = True):
        if self.stage_dev is None or self.stage_dev.numel() < elems:
            elems = int(elems * 1.25)
            self.stage_dev = torch.empty(elems, dtype=torch.float32, device=f'cuda:{self.dev}')
            self.stage_host = None
        if host and self.stage_host is None:
            self.stage_host = torch.empty(self.stage_dev.numel(), dtype=torch.float32).pin_memory()

    def ensure_crops(self, nbytes: int):
        if self.crops_host is None or self.crops_host.numel() < nbytes:
            cap = int(nbytes * 1.25) + 4096
            self.crops_host = torch.empty(cap, dtype=torch.uint8).pin_memory()
            self.crops_dev = torch.empty(cap, dtype=torch.uint8, device=f'cuda:{self.dev}')

    def ensure_boxes(self, n: int):
        if self.boxes_host is None or self.boxes_host.shape[0] < n:
            cap = max(n, 64) * 2
            d = f'cuda:{self.dev}'
            self.boxes_host = torch.empty((cap, 5), dtype=torch.int32).pin_memory()
            self.boxes_dev = torch.empty((cap, 5), dtype=torch.int32, device=d)
            self.flags_dev = torch.empty(cap, dtype=torch.int32, device=d)
            self.flags_host = torch.empty(cap, dtype=torch.int32).pin_memory()


class RecognitionEngine:
    def __init__(self, model: TorchVGSLModel, device: int = 0, max_batch: int = 256, max_width: int = 2400,
                 slots: int = 2, temperature: float = 1.0):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.model = model
        self.device = device
        self.temperature = float(temperature)
        self.in_channels, self.in_height = model.input[1], model.input[2]
        with torch.cuda.device(device):
            self.slots = [_Slot(model, device)]
            self.slots += [_Slot(model, device, self.slots[0]) for _ in range(slots - 1)]
            self.classes, _, max_t = self.slots[0].plan.out_shape(max_width)
            for s in self.slots:
                s.ensure_results(max_batch, max_t)
        # (round 6, measured and not kept: a helper thread that pays the ~13 ms first copy of every stream -- the runtime builds a
        # stream's copy queue on first use, profiles/r06_h2d_first_copy.txt -- while the caller prepares its first lines: first passes
        # of 119-160 ms with it against 167-169 without on the 2048-line page, 53 against 50 ms on the 40-line page: inside the
        # run-to-run spread of a fresh process, profiles/r06_cold_start.txt)
        self._pg_host, self._pg_ev, self._pg_i = [None, None], [None, None], 0
        self._pg_stream = torch.cuda.Stream(device=device)
        self._next = 0
        self._inflight = deque()
        self.in_use = False          # held by a LinePipeline (rpred.py: one consumer per engine at a time)
        self.closed = False
        # front event of the batch submitted last: the next batch's convolution block queues behind it, so the
        # full-chip convolution blocks of different batches run one after another (no convoy of all slots doing
        # convolutions together and then all doing recurrences together) -- see include/kraken_amd.h
        self._fronts = []
        self.chain_fronts = os.environ.get('KRK_NO_FRONT_CHAIN') is None
        self.front_lag = int(os.environ.get('KRK_FRONT_LAG', '1'))

    # ------------------------------------------------------------------ diagnostics
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

    # ------------------------------------------------------------------ submission
    def free_slots(self) -> int:
        return sum(not s.busy for s in self.slots)

    def in_flight(self) -> int:
        return len(self._inflight)

    def _free_slot(self) -> _Slot:
        slot = self.slots[self._next]
        if slot.busy:
            raise RuntimeError('all slots busy: collect() a ticket before submitting more')
        return slot

    def stage(self, n: int, w: int, height: Optional[int] = None) -> np.ndarray:
        """
        Pinned host array (n, C, H, w) of the next free slot; fill it in place -- line k at ``[k, :, :, :w_k]`` and ZEROS to
        its right (the array is reused, not cleared) -- and call ``submit_staged``.
        ``height`` overrides the model's input height (variable-height specs: one plan per height).
        """
        slot = self._free_slot()
        c = self.in_channels
        h = self.in_height if height is None else height
        slot.ensure_stage(n * c * h * w)
        slot.staged = (n, c, h, w)
        return slot.stage_host[:n * c * h * w].numpy().reshape(n, c, h, w)   # NOT cleared: the caller zeroes what it does not fill

    def upload_page(self, page: np.ndarray) -> torch.Tensor:
        """uint8 page (H, W) or (H, W, 3) -> device tensor for ``submit_boxes`` (one upload per page)."""
        t = torch.from_numpy(np.ascontiguousarray(page, dtype=np.uint8))
        return t.to(f'cuda:{self.device}', non_blocking=False)

    def page_buffer(self, shape) -> np.ndarray:
        """
        Pinned host memory for a page (or a band of one), shape (rows, width[, 3]) uint8: the caller's threads convert the
        image straight into it, ``upload_page_buffer`` then starts ONE asynchronous DMA -- no intermediate array, no pageable
        copy.  Two buffers alternate, so band k+1 is filled while band k is still on its way.
        """
        if not hasattr(self, '_pg_host'):
            self._pg_host, self._pg_ev, self._pg_i = [None, None], [None, None], 0
            self._pg_stream = torch.cuda.Stream(device=self.device)
        self._pg_i ^= 1
        i, need = self._pg_i, int(np.prod(shape))
        if self._pg_ev[i] is not None:
            self._pg_ev[i].synchronize()              # the upload that last used this buffer has finished
        if self._pg_host[i] is None or self._pg_host[i].numel() < need:
            self._pg_host[i] = torch.empty(need + need // 4, dtype=torch.uint8, pin_memory=True)
        self._pg_shape = tuple(int(v) for v in shape)
        return self._pg_host[i][:need].numpy().reshape(self._pg_shape)

    def upload_page_buffer(self) -> torch.Tensor:
        """Starts the upload of the buffer ``page_buffer`` handed out last; the crops wait for it on the device (``submit_boxes``)."""
        i, need = self._pg_i, int(np.prod(self._pg_shape))
        dev = torch.empty(self._pg_shape, dtype=torch.uint8, device=f'cuda:{self.device}')
        self._pg_stream.wait_stream(torch.cuda.current_stream(self.device))      # `dev` may recycle a block still in use there
        with torch.cuda.stream(self._pg_stream):
            dev.view(-1).copy_(self._pg_host[i][:need], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._pg_stream)
        dev.record_stream(self._pg_stream)
        self._pg_ev[i] = ev
        dev._krk_ready = ev
        return dev

    @staticmethod
    def pin_blocks(table, pins: dict, y0: int, y1: int) -> bool:
        """
        hipHostRegister of the Pillow blocks that hold rows [y0, y1) (once per block: ``pins`` maps block address -> bytes and belongs
        to the caller, who calls ``unpin_blocks`` before the image may go away).  True: every one of those blocks is page-locked, the
        band can be copied asynchronously.
        """
        blocks = pins.get('blocks')
        if blocks is None:
            rows, ls = table.rows, table.linesize
            cuts = np.flatnonzero(np.diff(rows) != ls) + 1
            st = np.concatenate(([0], cuts))
            en = np.concatenate((cuts, [len(rows)]))
            blocks = pins['blocks'] = [(int(a), int(b), int(rows[a]), int((b - a) * ls)) for a, b in zip(st, en)]
            pins['done'] = {}
        rt = torch.cuda.cudart()
        ok = True
        for a, b, addr, nbytes in blocks:
            if b <= y0 or a >= y1:
                continue
            state = pins['done'].get(addr)
            if state is None:
                probe = torch.from_numpy(np.frombuffer((C.c_ubyte * 1).from_address(addr), dtype=np.uint8))
                if probe.is_pinned():
                    state = False              # locked by someone else (another live run on this page): theirs to release, not ours
                else:
                    try:
                        state = int(rt.cudaHostRegister(addr, nbytes, 0)) == 0
                    except Exception:
                        state = False
                    if not state:
                        # (read-only mapping, memlock limit ...)  HIP keeps the failure as this thread's "last error" and the library's
                        # launch wrappers read that after every launch: take it off, or the next kernel is reported as failed
                        hip = _hip_runtime()
                        if hip:
                            hip.hipGetLastError()
                pins['done'][addr] = state
            ok = ok and state
        return ok

    @staticmethod
    def unpin_blocks(pins: dict):
        rt = torch.cuda.cudart()
        for addr, state in (pins.get('done') or {}).items():
            if state:
                try:
                    bad = int(rt.cudaHostUnregister(addr)) != 0
                except Exception:
                    bad = True
                if bad and _hip_runtime():                 # (see pin_blocks: a refusal must not stay behind as the thread's last error)
                    _hip_runtime().hipGetLastError()
        pins.clear()

    def upload_rows(self, table, y0: int, y1: int, pins: Optional[dict] = None) -> torch.Tensor:
        """
        Rows [y0, y1) of an image whose rows lie in Pillow's memory (``pilmem.RowTable``) -> device tensor (rows, width[, 4]), copied
        STRAIGHT from Pillow's blocks: one pageable host -> device copy per run of rows that are contiguous there (a block holds
        many rows).  On MI355X hosts a pageable copy runs at the pinned rate (460 MB: 9.8 ms against 4-6 ms of memmove on 8 threads
        + 8.5 ms of DMA) and, unlike a pinned staging buffer, costs nothing the first time: the FIRST DMA out of a fresh pinned
        buffer of that size took 159 ms (profiles/r06_h2d_paths.txt) -- most of a page's first pass.
        """
        ls = table.linesize
        shape = (y1 - y0, table.width) if table.pixelsize == 1 else (y1 - y0, table.width, table.pixelsize)
        dev = torch.empty(shape, dtype=torch.uint8, device=f'cuda:{self.device}')
        flat = dev.view(-1)
        rows = table.rows[y0:y1]
        cuts = np.flatnonzero(np.diff(rows) != ls) + 1
        starts = np.concatenate(([0], cuts)).tolist()
        ends = np.concatenate((cuts, [y1 - y0])).tolist()
        # ``pins``: the caller's registry of page-locked Pillow blocks (pin_blocks): the band's copies are then asynchronous
        locked = pins is not None and self.pin_blocks(table, pins, y0, y1)
        self._pg_stream.wait_stream(torch.cuda.current_stream(self.device))      # `dev` may recycle a block still in use there
        with torch.cuda.stream(self._pg_stream):
            for a, b in zip(starts, ends):
                n = (b - a) * ls
                src = np.frombuffer((C.c_ubyte * n).from_address(int(rows[a])), dtype=np.uint8)
                flat[a * ls:b * ls].copy_(torch.from_numpy(src), non_blocking=locked)   # pageable: returns when the bytes have left the host
            ev = torch.cuda.Event()
            ev.record(self._pg_stream)
        dev.record_stream(self._pg_stream)
        dev._krk_ready = ev
        return dev

    def _page_format(self, page_dev: torch.Tensor):
        """
        (rows, width, bytes per pixel) of an uploaded page: (H, W) is an 'L' page, (H, W, 3) packed RGB, (H, W, 4) Pillow's own
        R, G, B, X storage (``kraken_amd.pilmem``).  A 1-channel model reads colour pages through Pillow's 'L' conversion on the
        device (``krk_prep_lines_fmt``); a 3-channel model needs a colour page.
        """
        if page_dev.dtype != torch.uint8 or page_dev.dim() not in (2, 3) or not page_dev.is_contiguous():
            raise ValueError('page must be a contiguous uint8 tensor of shape (H, W) or (H, W, 3 | 4)')
        ps = 1 if page_dev.dim() == 2 else int(page_dev.shape[2])
        if ps not in (1, 3, 4) or (self.in_channels == 3 and ps == 1) or self.in_channels not in (1, 3):
            raise ValueError(f'page has {ps} bytes per pixel, the model takes {self.in_channels} channel(s)')
        return int(page_dev.shape[0]), int(page_dev.shape[1]), ps

    def submit_boxes(self, page_dev: torch.Tensor, boxes: np.ndarray, pad: int, want_probs: bool = False) -> int:
        """
        Recognises rectangular crops of an uploaded page: ``boxes`` int32 (n, 5) = x0, y0, x1, y1, resized width.  Crop,
        fixed-height LANCZOS resize, white padding and inversion run on the device (``krk_prep_lines``), straight into
        this slot's staging tensor -- the device-side counterpart of ``stage`` + ``submit_staged``.
        """
        slot = self._free_slot()
        boxes = np.ascontiguousarray(boxes, dtype=np.int32)
        n = len(boxes)
        c, h = self.in_channels, self.in_height
        widths = boxes[:, 4] + 2 * pad
        w = int(widths.max())
        slot.ensure_stage(n * c * h * w, host=False)
        slot.ensure_boxes(n)
        slot.boxes_host[:n].copy_(torch.from_numpy(boxes))
        ph, pw, ps = self._page_format(page_dev)
        page_dev.record_stream(slot.stream)          # the caller may drop the page before this slot's crops have run
        ready = getattr(page_dev, '_krk_ready', None)
        with torch.cuda.stream(slot.stream):
            if ready is not None:
                slot.stream.wait_event(ready)
            slot.boxes_dev[:n].copy_(slot.boxes_host[:n], non_blocking=True)
            _lib.check(self.lib.krk_prep_lines_fmt(page_dev.data_ptr(), ph, pw, pw * ps, ps, c, slot.boxes_dev.data_ptr(), n,
                                                   int((boxes[:, 3] - boxes[:, 1]).max()), h, int(pad), w,
                                                   slot.stage_dev.data_ptr(), slot.flags_dev.data_ptr(), slot.stream.cuda_stream))
            slot.flags_host[:n].copy_(slot.flags_dev[:n], non_blocking=True)
        slot.has_flags = True
        x = slot.stage_dev[:n * c * h * w].view(n, c, h, w)
        slot.page_keep = page_dev
        return self._launch(slot, x, widths.astype(np.int32), want_probs, wait_current=True)

    def submit_crops(self, crops: list, pad: int, want_probs: bool = False, pool=None) -> int:
        """
        Recognises line images that were cut out on the host: ``crops`` = uint8 arrays ``(h, w)`` (1-channel models) or
        ``(h, w, 3)``.  They travel packed, ONE byte per pixel and channel (a float tensor is four), and are resized to the
        model height, padded, scaled and inverted on the device (``krk_prep_crops``) -- the counterpart of ``stage`` +
        ``submit_staged`` for callers that have images, not tensors.  ``pool``: executor for the packing copies.
        """
        slot = self._free_slot()
        c, h = self.in_channels, self.in_height
        n = len(crops)
        desc = np.empty((n, 5), dtype=np.int32)                  # offset, w, h, out_w (+ 1 unused: the box buffers are 5 wide)
        off = 0
        for k, a in enumerate(crops):
            if a.dtype != np.uint8 or a.ndim != (2 if c == 1 else 3) or (c == 3 and a.shape[2] != 3):
                raise ValueError(f'crop {k}: expected a uint8 array of shape (h, w{", 3" if c == 3 else ""}), got {a.dtype} {a.shape}')
            ch, cw = int(a.shape[0]), int(a.shape[1])
            desc[k] = (off, cw, ch, int(cw * h / ch) if ch else 0, 0)
            off += (ch * cw * c + 15) & ~15                       # 16-byte aligned images
        widths = desc[:, 3] + 2 * pad
        w = int(widths.max())
        slot.ensure_crops(off)
        slot.ensure_stage(n * c * h * w, host=False)
        slot.ensure_boxes(n)
        buf = slot.crops_host.numpy()

        def pack(lo_hi):
            for k in range(*lo_hi):
                a = crops[k]
                buf[desc[k, 0]:desc[k, 0] + a.size] = np.ascontiguousarray(a).reshape(-1)
        if pool is not None and n >= 32:
            step = -(-n // 8)
            list(pool.map(pack, [(a, min(a + step, n)) for a in range(0, n, step)]))
        else:
            pack((0, n))
        slot.boxes_host[:n].copy_(torch.from_numpy(desc))
        with torch.cuda.stream(slot.stream):
            slot.crops_dev[:off].copy_(slot.crops_host[:off], non_blocking=True)          # PCIe copy on the slot's own stream
            slot.boxes_dev[:n].copy_(slot.boxes_host[:n], non_blocking=True)
            # krk_prep_crops reads 4-wide descriptors: compact them on the device side of the copy
            d4 = slot.boxes_dev[:n, :4].contiguous()
            _lib.check(self.lib.krk_prep_crops(slot.crops_dev.data_ptr(), c, d4.data_ptr(), n, int(desc[:, 2].max()), h, int(pad), w,
                                               slot.stage_dev.data_ptr(), slot.flags_dev.data_ptr(), slot.stream.cuda_stream))
            slot.flags_host[:n].copy_(slot.flags_dev[:n], non_blocking=True)
            d4.record_stream(slot.stream)
        slot.has_flags = True
        x = slot.stage_dev[:n * c * h * w].view(n, c, h, w)
        return self._launch(slot, x, widths.astype(np.int32), want_probs, wait_current=False)

    def measure_dewarp(self, crops: list, pool=None):
        """
        First half of the device-side CenterNormalizer dewarp (1-channel bbox lines, kraken/lib/lineest.py:34-65): uploads the
        uint8 line images ``(h, w)`` of ONE batch into the next free slot and measures centre line and spread on the device
        (``krk_dewarp_measure``).  Blocks until the per-line results are back -- a line's output width depends on its spread --
        and returns ``(r, ok, ink)`` int arrays: ``ok`` = the reference's band slices are full (otherwise the line must take the
        host transform), ``ink`` = the line is not flat.  ``submit_dewarped`` finishes the batch.
        """
        return self.measure_dewarp_begin(crops, pool).result()

    def measure_dewarp_begin(self, crops, pool=None, ahead: int = 0, page: Optional[torch.Tensor] = None):
        """
        The same without the wait: enqueues upload + ``krk_dewarp_measure`` on the slot ``ahead`` places behind the next free one
        and returns a handle whose ``result()`` blocks for ``(r, ok, ink)``.  With ``ahead=1`` the measurement of batch k+1 is
        in flight while the host finishes batch k (``submit_dewarped`` of the slot in front) -- the device -> host read-back of a
        batch then costs the host nothing (round 4: it was 15 % of the API path's wall time).  Between a ``begin`` and the
        ``submit_dewarped`` of its slot nothing else may be submitted.

        With ``page`` (a tensor from ``upload_page`` / ``upload_page_buffer``) ``crops`` is an int array (n, 4) of boxes
        x0, y0, x1, y1 INSIDE that page: the lines are read where they lie (``krk_dewarp_measure_page``), nothing is packed or
        uploaded per line; colour pages are read through Pillow's 'L' conversion.
        """
        from .transforms import dewarp_tables
        slot = self.slots[(self._next + ahead) % len(self.slots)]
        if slot.busy:
            raise RuntimeError('all slots busy: collect() a ticket before submitting more')
        if self.in_channels != 1:
            raise ValueError('the dewarp is defined for 1-channel models')
        n = len(crops)
        desc = np.empty((n, 8), dtype=np.int32)
        dev = f'cuda:{self.device}'
        st = slot.__dict__.setdefault('dw', {})
        if page is not None:
            ph, pw, ps = self._page_format(page)
            bx = np.ascontiguousarray(crops, dtype=np.int64).reshape(n, 4)
            cw, chh = bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]
            if (bx[:, 0] < 0).any() or (bx[:, 1] < 0).any() or (bx[:, 2] > pw).any() or (bx[:, 3] > ph).any() or (chh < 2).any() or \
                    (cw < 1).any():
                raise ValueError('dewarp boxes must lie inside the page and be at least 2 rows high')
            if ph * pw * ps >= 1 << 32:
                raise ValueError('page of 4 GiB or more: crop offsets are 32-bit')
            tables, index = dewarp_tables(int(v) for v in chh)
            tab = np.array([index[int(v)] for v in chh], dtype=np.int64).reshape(n, 4)
            area = 3 * chh * cw
            desc[:, 0] = ((bx[:, 1] * pw + bx[:, 0]) * ps).astype(np.uint32).view(np.int32)
            desc[:, 1], desc[:, 2] = cw, chh
            desc[:, 3] = np.concatenate(([0], np.cumsum(area)[:-1]))
            desc[:, 4:8] = tab
            soff = int(area.sum())
            page.record_stream(slot.stream)
            src = (page, pw * ps, ps)
        else:
            tables, index = dewarp_tables(a.shape[0] for a in crops)
            off = soff = 0
            for k, a in enumerate(crops):
                if a.dtype != np.uint8 or a.ndim != 2 or a.shape[0] < 2:
                    raise ValueError(f'crop {k}: expected a uint8 array of shape (h >= 2, w), got {a.dtype} {a.shape}')
                ch, cw = int(a.shape[0]), int(a.shape[1])
                woff, r0, r1, r2 = index[ch]
                desc[k] = (off, cw, ch, soff, woff, r0, r1, r2)
                off += (ch * cw + 15) & ~15
                soff += 3 * ch * cw
            slot.ensure_crops(off)
            buf = slot.crops_host.numpy()

            def pack(lo_hi):
                for k in range(*lo_hi):
                    a = crops[k]
                    buf[desc[k, 0]:desc[k, 0] + a.size] = np.ascontiguousarray(a).reshape(-1)
            if pool is not None and n >= 32:
                step = -(-n // 8)
                list(pool.map(pack, [(a, min(a + step, n)) for a in range(0, n, step)]))
            else:
                pack((0, n))
            src = (slot.crops_dev, 0, 1)
        maxw, maxh = int(desc[:, 1].max()), int(desc[:, 2].max())
        if soff >= 1 << 31:
            raise ValueError('dewarp batch too large: the scratch offsets are 32-bit (bound the batch by pixels)')
        st['src'] = src
        with torch.cuda.stream(slot.stream):
            if page is not None:
                ready = getattr(page, '_krk_ready', None)
                if ready is not None:
                    slot.stream.wait_event(ready)
            else:
                slot.crops_dev[:off].copy_(slot.crops_host[:off], non_blocking=True)
            st['desc'] = torch.from_numpy(desc).to(dev, non_blocking=True)
            st['wts'] = torch.from_numpy(tables).to(dev, non_blocking=True)
            if st.get('scratch') is None or st['scratch'].numel() < soff:
                st['scratch'] = torch.empty(int(soff * 1.25) + 1024, dtype=torch.float64, device=dev)
            st['work'] = torch.empty(2 * n + 2 * n * maxw, dtype=torch.int32, device=dev)
            info = torch.empty((n, 4), dtype=torch.int32, device=dev)
            _lib.check(self.lib.krk_dewarp_measure_page(src[0].data_ptr(), src[1], src[2], st['desc'].data_ptr(), n, maxw, maxh,
                                                        st['wts'].data_ptr(), st['scratch'].data_ptr(), st['work'].data_ptr(),
                                                        info.data_ptr(), slot.stream.cuda_stream))
            info_h = torch.empty((n, 4), dtype=torch.int32).pin_memory() if st.get('info_h') is None or st['info_h'].shape[0] < n \
                else st['info_h']
            st['info_h'] = info_h
            info_h[:n].copy_(info, non_blocking=True)
            done = torch.cuda.Event()
            done.record(slot.stream)
        st.update(n=n, maxw=maxw, host_desc=desc)

        class _Measured:
            def result(_self):
                done.synchronize()
                a = info_h[:n].numpy()
                return a[:, 0].copy(), a[:, 1].astype(bool), a[:, 2].astype(bool)
        return _Measured()

    def submit_dewarped(self, r: np.ndarray, use: np.ndarray, pad: int, want_probs: bool = False) -> int:
        """Second half: cut-out band, bilinear scaling to the model height, uint8 truncation, padding, inversion (``krk_dewarp_apply``)
        of the lines with ``use`` set, then the recognition of the batch.  Lines with ``use`` clear stay zero (flat: flag 0)."""
        slot = self._free_slot()
        st = slot.dw
        n, maxw, desc = st['n'], st['maxw'], st['host_desc']
        h = self.in_height
        geo = np.zeros((n, 4), dtype=np.int32)
        for k in range(n):
            if use[k]:
                scale = h * 1.0 / (2 * int(r[k]))
                geo[k] = (int(r[k]), int(scale * int(desc[k, 1])), 1, 0)
        widths = np.where(geo[:, 2] > 0, geo[:, 1] + 2 * pad, 1).astype(np.int32)
        w = int(widths.max())
        slot.ensure_stage(n * h * w, host=False)
        slot.ensure_boxes(n)
        dev = f'cuda:{self.device}'
        with torch.cuda.stream(slot.stream):
            geo_d = torch.from_numpy(geo).to(dev, non_blocking=True)
            src = st.pop('src')          # the slot lets go of the page / band tensor here (a full RGBX page is > 100 MB of HBM per slot)
            if src[0] is not slot.crops_dev:
                src[0].record_stream(slot.stream)
            _lib.check(self.lib.krk_dewarp_apply_page(src[0].data_ptr(), src[1], src[2], st['desc'].data_ptr(), n, maxw, st['work'].data_ptr(),
                                                      geo_d.data_ptr(), h, int(pad), w, slot.stage_dev.data_ptr(), slot.flags_dev.data_ptr(),
                                                      slot.stream.cuda_stream))
            slot.flags_host[:n].copy_(slot.flags_dev[:n], non_blocking=True)
            geo_d.record_stream(slot.stream)
        slot.has_flags = True
        x = slot.stage_dev[:n * h * w].view(n, 1, h, w)
        return self._launch(slot, x, widths, want_probs, wait_current=False)

    def submit_staged(self, lens=None, want_probs: bool = False) -> int:
        slot = self._free_slot()
        n, c, h, w = slot.staged
        k = n * c * h * w
        with torch.cuda.stream(slot.stream):
            slot.stage_dev[:k].copy_(slot.stage_host[:k], non_blocking=True)     # PCIe copy on the slot's own stream
        x = slot.stage_dev[:k].view(n, c, h, w)
        return self._launch(slot, x, lens, want_probs, wait_current=False)

    def submit(self, x: torch.Tensor, lens: Optional[np.ndarray] = None, want_probs: bool = False) -> int:
        """
        x: (N, C, H, W) float32 tensor, resident on this device -- or a (preferably pinned) HOST tensor, which is copied
        to a per-slot staging buffer on the slot's own stream so that the PCIe transfer of batch k+1 overlaps the kernels of
        batch k.  Returns a ticket.
        """
        slot = self._free_slot()
        if not x.is_cuda:
            k = x.numel()
            slot.ensure_stage(k)
            with torch.cuda.stream(slot.stream):
                slot.stage_dev[:k].copy_(x.reshape(-1), non_blocking=True)
            x = slot.stage_dev[:k].view(x.shape)
            return self._launch(slot, x, lens, want_probs, wait_current=False)
        return self._launch(slot, x, lens, want_probs, wait_current=True)

    def _launch(self, slot: _Slot, x: torch.Tensor, lens, want_probs: bool, wait_current: bool) -> int:
        slot_id = self._next
        N, _, H, W = x.shape
        plan = slot.plan
        if H != self.in_height:
            raise ValueError(f'engine built for input height {self.in_height}, got {H}')
        _, _, T = plan.out_shape(W)
        slot.ensure_results(N, T)
        nt = slot.cap_n * slot.cap_t
        base = slot.dev_buf.data_ptr()
        dec = _lib.KrkDecodeOut(base, base + 4 * nt, base + 8 * nt, base + 12 * nt, base + 16 * nt, slot.cap_t)
        lens_arr = None
        if lens is not None:
            lens_arr = np.ascontiguousarray(np.asarray(lens, dtype=np.int32))
        if wait_current or slot.want_probs:
            # the input may have been produced on the current stream; and when the slot's previous batch exposed its softmax
            # (`last_probs`), the consumer's clones were enqueued there too: order them before this batch's overwrite
            slot.stream.wait_stream(torch.cuda.current_stream(self.device))
        if self.chain_fronts and len(self._fronts) >= self.front_lag and len(self.slots) > 1:
            _lib.check(self.lib.krk_plan_wait_front(plan.handle, self._fronts[-self.front_lag]))
        self._fronts.append(self.lib.krk_plan_front_event(plan.handle))
        del self._fronts[:-4]
        slot.want_probs = bool(want_probs)
        with torch.cuda.stream(slot.stream):
            probs_ptr = None
            if want_probs:
                if slot.probs is None or slot.probs.numel() < N * T * self.classes:
                    slot.probs = torch.empty(N * T * self.classes, dtype=torch.float32, device=x.device)
                probs_ptr = slot.probs.data_ptr()
            _lib.check(self.lib.krk_recognize(plan.handle, x.data_ptr(),
                                              lens_arr.ctypes.data if lens_arr is not None else None, N, W,
                                              self.temperature, slot.stream.cuda_stream, None, probs_ptr,
                                              slot.olens.ctypes.data, C.byref(dec)))
            slot.host_buf.copy_(slot.dev_buf, non_blocking=True)
            slot.event.record(slot.stream)
        slot.busy, slot.n, slot.t, slot.keep = True, N, T, x
        slot.keep_lens = lens_arr
        self._next = (slot_id + 1) % len(self.slots)
        self._inflight.append(slot_id)
        return slot_id

    # ------------------------------------------------------------------ results
    def collect(self, ticket: Optional[int] = None) -> tuple[DecodedBatch, np.ndarray]:
        """Waits for the oldest (or the given) in-flight batch; returns (DecodedBatch, olens)."""
        if ticket is None:
            ticket = self._inflight[0]
        self._inflight.remove(ticket)
        slot = self.slots[ticket]
        slot.event.synchronize()
        x, lens = slot.keep, getattr(slot, 'keep_lens', None)
        slot.busy, slot.keep, slot.keep_lens = False, None, None    # the slot is free again whatever the status below says
        try:
            _lib.check(self.lib.krk_plan_status(slot.plan.handle))  # a kernel that gave up waiting raises here, never hangs
        except _lib.KrakenAmdError as e:
            # one retry of THIS batch on the streaming recurrent kernel (no inter-workgroup exchange) before the page is lost
            if not _lib.is_exchange_timeout(e) or x is None:
                raise
            import logging
            logging.getLogger(__name__).warning(f'{e}; running the batch again on the streaming recurrent kernel')
            # the retry is not a new batch of the pipeline: it neither waits for another plan's convolution front nor enters the
            # ring of fronts the next submissions order themselves by
            keep_next, inflight, fronts, chain = self._next, list(self._inflight), list(self._fronts), self.chain_fronts
            self._next, self.chain_fronts = ticket, False
            try:
                with _lib.streaming_recurrence(slot.plan.handle):  # this slot's plan only: the other slots keep the cluster kernel
                    self._launch(slot, x, lens, slot.want_probs, False)
                    slot.event.synchronize()
            finally:
                self._inflight, self._next, self.chain_fronts = deque(inflight), keep_next, chain
                self._fronts[:] = fronts
            slot.busy, slot.keep, slot.keep_lens = False, None, None
            _lib.check(self.lib.krk_plan_status(slot.plan.handle))
        nt = slot.cap_n * slot.cap_t
        h = slot.host_buf.numpy()
        n, t = slot.n, slot.cap_t
        view = lambda k: h[k * nt:(k + 1) * nt].reshape(slot.cap_n, t)[:n]   # noqa: E731
        batch = DecodedBatch(view(0).copy(), view(1).copy(), view(2).copy(), view(3).copy().view(np.float32),
                             h[4 * nt:4 * nt + n].copy())
        self.last_slot = slot
        self.last_flags = slot.flags_host[:n].numpy().copy() if slot.has_flags else None
        slot.has_flags = False
        return batch, slot.olens[:n].copy()

    def last_probs(self) -> Optional[torch.Tensor]:
        """(N, C, T) softmax of the batch returned by the last ``collect`` (device tensor; valid until that slot is reused)."""
        slot = getattr(self, 'last_slot', None)
        if slot is None or not slot.want_probs or slot.probs is None:
            return None
        return slot.probs[:slot.n * slot.t * self.classes].view(slot.n, slot.t, self.classes).permute(0, 2, 1)

    def reset(self):
        """
        Abandons whatever is in flight (a consumer that stopped iterating, an exception between submit and collect): waits for
        the slots' streams, frees every slot and clears a stale status word, so that the next run starts from a clean engine.
        """
        for s in self.slots:
            if s.busy:
                s.stream.synchronize()
            s.busy, s.keep, s.staged, s.has_flags, s.want_probs = False, None, None, False, False
            if s.plan.handle:
                self.lib.krk_plan_status(s.plan.handle)            # reading the word clears it
        self._inflight.clear()
        self._fronts.clear()
        self._next = 0
        self.last_slot = None
        self.last_flags = None

    def close(self):
        self.closed = True
        for s in self.slots:
            if s.busy:
                s.stream.synchronize()
            s.plan.close()
