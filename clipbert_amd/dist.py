"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" on CPU).

The reference averages every parameter gradient with Horovod/NCCL, one tensor at a time
(run_video_retrieval.py:298-301,432).  Here the gradients already live in ONE flat fp32 buffer laid
out transformer-groups-first, so the exchange is two large bucket all-reduces:

  bucket 0 = transformer parameters: issued from the end of the encoder backward, i.e. it travels over
             xGMI on RCCL's side stream WHILE the ResNet backward (the larger half of the step) runs;
  bucket 1 = CNN parameters: issued after the CNN backward.

Sums are averaged inside the fused AdamW (grad_scale = 1/world), so no extra pass touches the buffer.
The dead detectron2 RPN/ROI parameters the reference also all-reduces do not exist here.
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from .params import ParamBank


class GradSync:
    def __init__(self, bank: ParamBank, group=None, bucket_bytes: int = 0):
        self.bank = bank
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        t_end = bank.group_range[3][1]
        self.t_range = (0, t_end)
        self.c_range = (t_end, bank.n_train)
        self.bucket_elems = bucket_bytes // 4 if bucket_bytes > 0 else 0
        self._work: List = []

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def _reduce(self, a: int, b: int):
        if self.world == 1 or b <= a:
            return
        step = self.bucket_elems if self.bucket_elems > 0 else (b - a)
        for s in range(a, b, step):
            e = min(b, s + step)
            self._work.append(dist.all_reduce(self.bank.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_transformer(self):
        self._reduce(*self.t_range)

    def reduce_cnn(self):
        self._reduce(*self.c_range)

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []

    def broadcast_parameters(self, src: int = 0):
        """hvd.broadcast_parameters equivalent (run_video_retrieval.py:304): one flat buffer per kind."""
        if self.world == 1:
            return
        dist.broadcast(self.bank.master, src, group=self.group)
        dist.broadcast(self.bank.f_master, src, group=self.group)
        self.bank.sync_compute()
