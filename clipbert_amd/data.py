"""GPU-side input pipeline: what sits between the reference's DataLoader and ``ClipBert.forward``.

Reference: src/datasets/dataloader.py:86-152 ``PrefetchLoader`` -- a side stream that ``.cuda()``s the next collated batch
while the current one computes, then ``.float()`` and ``ImageNorm`` on the GPU (a 4-byte-per-pixel fp32 tensor is
materialised per batch).  Here

* frames stay **uint8** all the way into HBM (1 byte per pixel over PCIe and in HBM); the cast, mean / std and the RGB->BGR
  flip happen inside the stem's input pack (``cb_stem_pack(src_u8=1)``) -- ``img_normalize`` is therefore not a tensor op here
  but the (mean, std) pair handed to the model;
* host staging is **pinned and double-buffered**: two pinned slots per tensor key, batch i+1 is copied host->pinned->HBM on a
  side HIP stream while batch i computes; a slot is reused only after the event recorded behind its last H2D copy has
  completed, and the consumer stream waits on the copy's event (not on the whole side stream);
* ``InfiniteIterator`` as in the reference (:155-175).

Host code + HIP streams / events (through torch): plumbing, no arithmetic."""
from typing import Dict, Iterable, Iterator, Optional

import torch


class InfiniteIterator:
    """iterate an iterable object infinitely (src/datasets/dataloader.py:155-175)"""
    def __init__(self, iterable):
        self.iterable = iterable
        self.iterator = iter(iterable)

    def __iter__(self):
        while True:
            try:
                batch = next(self.iterator)
            except StopIteration:
                self.iterator = iter(self.iterable)
                batch = next(self.iterator)
            yield batch


class PrefetchLoader:
    """Drop-in for the reference's PrefetchLoader: iterate it to get batches whose tensors live on ``device``.

    ``img_normalize``: ignored as a callable -- pass the model's pixel statistics at model construction instead (frames are
    delivered as uint8 and normalised inside the stem pack).  (task, batch) tuples of the reference's MetaLoader pass
    through unchanged."""

    def __init__(self, loader: Iterable, device=None, img_normalize=None, slots: int = 2):
        self.loader = loader
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.img_normalize = img_normalize
        self.slots = max(2, slots)
        self._cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self._cuda else None
        self._pinned: Dict = {}           # (key, slot) -> pinned host tensor
        self._slot_free = [None] * self.slots      # event recorded after the slot's last H2D copy
        self._n = 0

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.__dict__["loader"], name)

    # ---- staging ---------------------------------------------------------------------------------------------------------
    def _stage(self, key, slot, t: torch.Tensor) -> torch.Tensor:
        """host tensor -> pinned slot (reallocated when the batch outgrows it) -> device, on the side stream"""
        if not self._cuda:
            return t.to(self.device)
        buf = self._pinned.get((key, slot))
        n = t.numel()
        if buf is None or buf.dtype != t.dtype or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=t.dtype).pin_memory()
            self._pinned[(key, slot)] = buf
        view = buf[:n].view(t.shape)
        view.copy_(t)                                   # pageable -> pinned (host memcpy)
        return view.to(self.device, non_blocking=True)  # pinned -> HBM, asynchronous on the side stream

    def _move(self, obj, slot, prefix=""):
        if torch.is_tensor(obj):
            return self._stage(prefix, slot, obj) if obj.device.type == "cpu" else obj.to(self.device, non_blocking=True)
        if isinstance(obj, dict):
            return {k: self._move(v, slot, f"{prefix}/{k}") for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            moved = [self._move(v, slot, f"{prefix}/{i}") for i, v in enumerate(obj)]
            return type(obj)(moved) if not hasattr(obj, "_fields") else type(obj)(*moved)
        return obj

    def _preload(self, it) -> Optional[tuple]:
        try:
            batch = next(it)
        except StopIteration:
            return None
        slot = self._n % self.slots
        self._n += 1
        if not self._cuda:
            return self._move(batch, slot), None
        ev_free = self._slot_free[slot]
        if ev_free is not None:
            ev_free.synchronize()                       # the copy that last read this slot's pinned buffers has finished
        with torch.cuda.stream(self.stream):
            moved = self._move(batch, slot)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._slot_free[slot] = ev
        return moved, ev

    @staticmethod
    def _record_stream(obj, stream):
        if torch.is_tensor(obj):
            if obj.is_cuda:
                obj.record_stream(stream)
        elif isinstance(obj, dict):
            for v in obj.values():
                PrefetchLoader._record_stream(v, stream)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                PrefetchLoader._record_stream(v, stream)

    def __iter__(self) -> Iterator:
        it = iter(self.loader)
        nxt = self._preload(it)
        while nxt is not None:
            batch, ev = nxt
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)                      # only this batch's copies, not everything queued behind them
                self._record_stream(batch, cur)
            nxt = self._preload(it)                     # batch i+1 travels while batch i computes
            yield batch
