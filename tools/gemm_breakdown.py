#!/usr/bin/env python
"""Log every GEMM call (cb_gemm and cb_gemm_group) of ONE eager step of the bench workload (bench.py --mode <mode>, default train =
the metric configuration) with its problems, family (clipbert_amd/gemm_log.py: by the roof that bounds it), algorithmic flops / bytes
and the NUMBER OF KERNELS the library launched for it, for joining with a rocprofv3 kernel trace / counter collection of the same
process (tools/pmc_traffic.py): the LAST sum(kernels) GEMM dispatches of the trace are this step's, in order.

The kernel count per call comes from the library's own launch trace (CB_GEMM_TRACE=1: one "cb_gemm:" line per single launch, one
"cb_gemm_group[i/n]:" line per problem of a grouped launch), read back per call from a file that stands in for stderr."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["CB_GEMM_TRACE"] = "1"
import torch  # noqa: E402

from clipbert_amd import gemm_log, ops  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "train"
    import importlib
    bench = importlib.import_module("bench")
    trace = tempfile.NamedTemporaryFile(prefix="cb_gemm_trace_", suffix=".txt", delete=False)
    real_stderr = os.dup(2)
    calls = []

    class Stop(Exception):
        pass

    real_log = bench.log
    state = {"steps": 0}

    def log_hook(msg):
        real_log(msg)
        if msg.startswith("eager warm-up done"):
            raise Stop()

    bench.log = log_hook
    argv = sys.argv
    sys.argv = ["bench.py", "--mode", mode.split(":", 1)[0], "--no-cpu-baseline", "--no-roofline"] + (mode.split(":", 1)[1].split() if ":" in mode else [])
    log = gemm_log.GemmLog()
    try:
        with log:
            # wrap the wrappers once more: note the trace file's size around every library call
            inner_gemm, inner_group = ops.gemm, ops.gemm_group

            def around(fn):
                def call(*a, **kw):
                    torch.cuda.synchronize()
                    os.fsync(2)
                    before = os.fstat(2).st_size
                    n0 = len(log.launches)
                    out = fn(*a, **kw)
                    os.fsync(2)
                    after = os.fstat(2).st_size
                    if len(log.launches) > n0:
                        log.launches[-1]["trace_span"] = (before, after)
                    return out
                return call
            ops.gemm, ops.gemm_group = around(inner_gemm), around(inner_group)
            os.dup2(trace.fileno(), 2)
            try:
                bench.main()
            except Stop:
                pass
            finally:
                os.dup2(real_stderr, 2)
                ops.gemm, ops.gemm_group = inner_gemm, inner_group
    finally:
        sys.argv = argv
        bench.log = real_log
    torch.cuda.synchronize()
    text = open(trace.name, "rb").read()
    launches = log.launches[len(log.launches) // 2:]           # the warm-up runs the step twice: keep one step's calls
    for ln in launches:
        a, b = ln.get("trace_span", (0, 0))
        lines = text[a:b].decode(errors="replace").splitlines()
        kernels = 0
        for t in lines:
            if t.startswith("cb_gemm:"):
                kernels += 1                                        # (+ its split-K reduce kernel, which the reader folds into it)
            elif t.startswith("cb_gemm_group[0/"):
                kernels += 1
        calls.append({"kernels": max(1, kernels), "grouped": len(ln["problems"]) > 1, "problems": ln["problems"]})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(calls, open(os.path.join(ROOT, "gpurun_out", "gemm_calls.json"), "w"))
    print("logged", len(calls), "library calls,", sum(c["kernels"] for c in calls), "GEMM kernels,", sum(len(c["problems"]) for c in calls), "problems")


if __name__ == "__main__":
    main()
