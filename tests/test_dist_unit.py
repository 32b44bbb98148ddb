"""GradSync bookkeeping on a toy parameter bank (CPU, no process group): ranges, buckets, the bf16 wire image, the dry-run mode
bench.py uses to time the N-rank plan on one GPU (pretend_world), and the refusal to reduce a range twice before wait()."""
import pytest
import torch
from torch import nn

from clipbert_amd.dist import GradSync
from clipbert_amd.params import ParamBank


def _toy_bank():
    root = nn.Module()
    root.transformer = nn.Sequential(nn.Linear(24, 16), nn.Linear(16, 8))
    root.cnn = nn.Module()
    root.cnn.grid_encoder = nn.Linear(8, 8)
    root.cnn.feature = nn.Linear(8, 4)
    return ParamBank(root, "cpu", torch.float32)


def test_pretend_world_does_everything_but_the_collectives():
    bank = _toy_bank()
    bank.grad.normal_(generator=torch.Generator().manual_seed(0))
    g0 = bank.grad.clone()
    sync = GradSync(bank, compress="bf16", pretend_world=4, bucket_bytes=256)        # 64 floats per bucket: several buckets per range
    assert sync.dry and sync.world == 4 and sync.grad_scale == 0.25
    assert sync.t_range[0] == 0 and sync.t_range[1] == sync.c_range[0] and sync.c_range[1] == bank.n_train
    sync.broadcast_parameters(0)                          # no process group: must not try to communicate
    sync.reduce_transformer()
    sync.reduce_cnn()
    wire = sync.wire_gradients()
    assert wire is not None and wire.dtype == torch.bfloat16
    sync.wait(cast_back=False)
    assert torch.equal(wire[:bank.n_train], g0[:bank.n_train].bfloat16())            # the wire image of THIS rank's gradients
    assert torch.equal(bank.grad, g0)                                                 # cast_back=False leaves the fp32 buffer alone
    sync.reduce_transformer()
    sync.reduce_cnn()
    sync.wait()                                           # default: the (un-reduced, dry) wire image comes back as fp32
    torch.testing.assert_close(bank.grad[:bank.n_train], g0[:bank.n_train].bfloat16().float(), rtol=0, atol=0)


def test_a_range_is_reduced_once_per_step():
    bank = _toy_bank()
    sync = GradSync(bank, compress=None, pretend_world=2)
    sync.reduce_transformer()
    with pytest.raises(AssertionError):
        sync.reduce_transformer()                         # again before wait(): other ranks' sums would be counted twice
    sync.wait()
    sync.reduce_transformer()                             # next step: fine
    sync.wait()


def test_single_rank_sync_is_a_no_op():
    bank = _toy_bank()
    bank.grad.fill_(1.0)
    sync = GradSync(bank, compress="bf16")
    assert sync.world == 1 and not sync.dry and sync.wire_gradients() is None
    sync.reduce_transformer(); sync.reduce_cnn(); sync.wait()
    assert float(bank.grad.min()) == 1.0
