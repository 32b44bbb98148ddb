"""cb_comm_* / cb_allreduce_bucket: the C ABI's RCCL entry points (include/clipbert_hip.h).  On a GPU-less host: the symbols exist
and fail with a message before any communicator exists.  On the (single-GPU) box: a world-size-1 communicator, an in-place
all-reduce on a side stream, the same captured into a hipGraph, and GradSync(comm="native") wiring.  Multi-rank behaviour over
xGMI cannot be exercised on 1-GPU boxes: the arithmetic of the exchange itself is covered by tests/test_dp_gloo.py."""
import ctypes

import pytest
import torch


def test_allreduce_without_communicator_fails_with_message():
    from clipbert_amd import _lib, build
    lib = _lib.bind(ctypes.CDLL(build.build()), strict=True)
    x = (ctypes.c_float * 4)()
    assert lib.cb_allreduce_bucket(ctypes.cast(x, ctypes.c_void_p), 4, 0, None) != 0
    assert b"cb_comm_init" in lib.cb_last_error()
    assert lib.cb_comm_info(None, None) != 0
    assert lib.cb_comm_destroy() == 0                      # nothing to destroy: not an error
    bad = ctypes.create_string_buffer(128)
    assert lib.cb_comm_init(3, 2, ctypes.cast(bad, ctypes.c_void_p)) != 0 and b"rank" in lib.cb_last_error()


@pytest.mark.gpu
def test_native_comm_world1_allreduce_and_graph_capture():
    from clipbert_amd.dist import NativeComm
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = NativeComm.from_process_group()                 # torch.distributed not initialised: rank 0 of 1
    assert (comm.rank, comm.world) == (0, 1)
    try:
        g = torch.Generator(device="cpu").manual_seed(0)
        t16 = torch.randn(1 << 20, generator=g).bfloat16().to(dev)
        t32 = torch.randn((1 << 20) + 3, generator=g).to(dev)
        r16, r32 = t16.clone(), t32.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        comm.all_reduce_(t16, side)                        # sum over one rank: the values must come back unchanged
        comm.all_reduce_(t32[3:], side)                    # (a bucket = a slice of the flat buffer)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(t16, r16) and torch.equal(t32, r32)
        # the collective is stream-ordered work like a kernel launch: capturable
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            t32.mul_(2.0)
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                t32.add_(1.0)
                comm.all_reduce_(t32)
                t32.mul_(0.5)
            graph.replay()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        # the capture itself does not execute: one replay on top of 2*r32
        torch.testing.assert_close(t32, (2 * r32 + 1) * 0.5, rtol=0, atol=1e-6)
    finally:
        NativeComm.destroy()


@pytest.mark.gpu
def test_native_reduce_scatter_allgather_broadcast_world1():
    """The two halves of an all-reduce and the parameter broadcast (cb_reduce_scatter_bucket / cb_allgather_bucket /
    cb_broadcast_bucket) on a world-size-1 communicator, in place, eager and captured: values come back unchanged and the calls are
    stream-ordered work.  (Multi-rank: the driver's scaling run; the arithmetic of the exchange: tests/test_dp_gloo.py.)"""
    from clipbert_amd.dist import NativeComm
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = NativeComm.from_process_group()
    try:
        g = torch.Generator(device="cpu").manual_seed(1)
        t16 = torch.randn(1 << 18, generator=g).bfloat16().to(dev)
        t32 = torch.randn(1 << 18, generator=g).to(dev)
        r16, r32 = t16.clone(), t32.clone()
        mine = comm.reduce_scatter_(t16)
        assert mine.data_ptr() == t16.data_ptr() and mine.numel() == t16.numel()      # world 1: the shard is the whole bucket
        comm.all_gather_(t16)
        comm.reduce_scatter_(t32)
        comm.all_gather_(t32)
        comm.broadcast_(t32, 0)
        torch.cuda.synchronize()
        assert torch.equal(t16, r16) and torch.equal(t32, r32)
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                t32.add_(1.0)
                comm.reduce_scatter_(t32)
                t32.mul_(2.0)                             # "the optimizer on the shard"
                comm.all_gather_(t32)
            graph.replay()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        torch.testing.assert_close(t32, (r32 + 1) * 2, rtol=0, atol=1e-6)
    finally:
        NativeComm.destroy()
