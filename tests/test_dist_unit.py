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


def test_wait_sends_whatever_was_forgotten():
    """a hook that never fired must not leave a range un-exchanged: wait() reduces the uncovered parts itself"""
    bank = _toy_bank()
    bank.grad.normal_(generator=torch.Generator().manual_seed(1))
    g0 = bank.grad.clone()
    sync = GradSync(bank, compress="bf16", pretend_world=2)
    sync.reduce_cnn()                                     # the transformer hook "did not fire"
    assert sync._uncovered() == [sync.t_range]
    sync.wait()
    assert sync.late_ranges == 1
    torch.testing.assert_close(bank.grad[:bank.n_train], g0[:bank.n_train].bfloat16().float(), rtol=0, atol=0)   # every range went through the wire
    sync.wait()                                           # nothing in flight: nothing to do, no new late range
    assert sync.late_ranges == 1


def test_cnn_range_in_two_parts():
    bank = _toy_bank()
    sync = GradSync(bank, compress=None, pretend_world=2)
    lo, hi = sync.c_range
    mid = bank.group_range[6][0]
    split = mid + (hi - mid) // 2 // 64 * 64
    sync.set_cnn_split(split)
    assert sync.c_early == [r for r in ((lo, mid), (split, hi)) if r[1] > r[0]] and sync._cnn_late() == [(mid, split)]
    sync.reduce_transformer()
    sync.reduce_cnn_early()
    sync.reduce_cnn()                                     # only the middle is left
    assert sorted(sync._inflight) == sorted([sync.t_range] + sync.c_early + [(mid, split)]) and not sync._uncovered()
    sync.wait()
    assert sync.late_ranges == 0
    sync.reduce_transformer()
    sync.reduce_cnn()                                     # early part not sent this step: the whole CNN range at once
    assert sync.c_range in sync._inflight
    sync.wait()


def test_wait_without_any_reduce_exchanges_the_whole_buffer(emul):
    """a step for which no reduce_* was called (hooks not armed) must not pass un-reduced gradients to the optimizer;
    a repeated wait() in the same step stays a no-op."""
    bank = _toy_bank()
    sync = GradSync(bank, compress="bf16", pretend_world=2)
    bank.zero_grad()                                      # a new gradient group begins
    bank.grad.normal_(generator=torch.Generator().manual_seed(2))
    g0 = bank.grad.clone()
    sync.wait()                                           # nobody reduced anything: everything goes out late
    assert sync.late_ranges == 1
    torch.testing.assert_close(bank.grad[:bank.n_train], g0[:bank.n_train].bfloat16().float(), rtol=0, atol=0)
    sync.wait()
    assert sync.late_ranges == 1


def test_owner_only_pieces_partition_every_bucket_and_survive_a_repeated_wait():
    """GradSync(shard=True): rank r owns the r-th 1/world of every bucket (groups are padded to 512 elements, so the pieces are whole
    64-element runs for world 2 / 4 / 8); a second wait() in the same step must not forget them."""
    bank = _toy_bank()
    assert all(a % ParamBank.GROUP_ALIGN == 0 and b % ParamBank.GROUP_ALIGN == 0 for a, b in bank.group_range) and bank.n_train % 512 == 0
    for world in (2, 4, 8):
        sync = GradSync(bank, compress=None, pretend_world=world, shard=True, bucket_bytes=4096)      # 1024-element buckets
        sync.reduce_transformer()
        sync.reduce_cnn()
        sync.wait()
        pieces = sync.owned_pieces()
        assert pieces and all((hi - lo) % 64 == 0 and lo % 64 == 0 for lo, hi in pieces)
        assert sum(hi - lo for lo, hi in pieces) * world == bank.n_train          # rank 0's share of every bucket
        sync.wait()
        assert sync.owned_pieces() == pieces
        sync.reduce_transformer(); sync.reduce_cnn(); sync.wait()
        assert sync.owned_pieces() == pieces                                       # the same partition every step


def test_native_communicator_bring_up_is_all_or_nothing():
    """NativeComm._bring_up_guarded: the multi-rank bring-up of the library's RCCL communicator either succeeds on every rank or raises on
    every rank -- whether the local attempt raised, answered with a wrong sum, never came back, or another rank reported a failure; the
    agreement collective is called exactly once in every case."""
    import time
    from clipbert_amd.dist import NativeComm
    calls = []

    def agree_with(others_ok):
        def agree(ok):
            calls.append(ok)
            return ok and others_ok
        return agree
    assert NativeComm._bring_up_guarded(lambda: ("comm", True), agree_with(True), 5.0) == "comm" and calls == [True]
    cases = {"another rank": (lambda: ("comm", True), False), "wrong sum": (lambda: ("comm", False), True),
             "RuntimeError('boom')": (lambda: (_ for _ in ()).throw(RuntimeError("boom")), True),
             "no answer within": (lambda: (time.sleep(3.0), ("comm", True))[1], True)}
    for needle, (work, others_ok) in cases.items():
        calls.clear()
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="bring-up") as ei:
            NativeComm._bring_up_guarded(work, agree_with(others_ok), 0.3, rank=1)
        assert needle in str(ei.value) and len(calls) == 1 and time.perf_counter() - t0 < 2.5, (needle, str(ei.value), calls)
