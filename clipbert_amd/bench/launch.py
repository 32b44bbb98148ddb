#!/usr/bin/env python
"""Launching and supervising bench.py's N > 1 runs (moved out of bench.py in round 5, VERDICT r4 item 8: frozen until a multi-GPU node
exists -- no behaviour change).  bench.py imports `supervise`, `_stub_worker` and re-exports ATTEMPTS / _run_attempts / _marker for
tests/test_zz_bench_cli.py."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# ---- launching N > 1 -----------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` (no launcher, WORLD_SIZE unset) starts its own N ranks under torch.distributed.run; under a launcher
# (`python -m torch.distributed.run ... bench.py --gpus N`, the driver's form) every rank process SUPERVISES one worker process.
# Either way a failed or hung attempt is retried with a more conservative data-parallel plan, because no multi-rank RCCL run of this
# code existed when it was written (1-GPU build boxes): the first number a multi-GPU node produces must not depend on the newest plan.
#   attempt 0: CB_BENCH_PLAN unset -> the whole step incl. its bucket collectives in ONE hipGraph, the library's own RCCL entry points
#   attempt 1: CB_BENCH_PLAN=split -> four hipGraphs with eager collectives between them (the round-2 / round-3 default)
#   attempt 2: CB_BENCH_PLAN=split CB_COMM=torch -> the same with torch.distributed carrying the buckets
# A user who sets CB_BENCH_PLAN / CB_COMM gets exactly that plan and no retry.  The attempt that produced the line is in config.attempt.
ATTEMPTS = ({}, {"CB_BENCH_PLAN": "split"}, {"CB_BENCH_PLAN": "split", "CB_COMM": "torch"})
ATTEMPT_TIMEOUT_S = float(os.environ.get("CB_BENCH_ATTEMPT_TIMEOUT", "420"))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _marker(job, attempt, what):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cb_bench_{job}_a{attempt}.{what}")


def _touch(path):
    with open(path, "w") as fh:
        fh.write(str(os.getpid()))


def _descendants(pid):
    """every live descendant of ``pid`` (a launcher puts its workers into sessions of their own: killing our child's process group
    alone would leave a hung worker holding its GPU)"""
    try:
        import psutil
        return [c.pid for c in psutil.Process(pid).children(recursive=True)]
    except Exception:                                            # noqa: BLE001  (no psutil: walk /proc)
        kids, todo = [], [pid]
        while todo:
            cur = todo.pop()
            for d in os.listdir("/proc"):
                if d.isdigit():
                    try:
                        with open(f"/proc/{d}/stat") as fh:
                            ppid = int(fh.read().rsplit(")", 1)[1].split()[1])
                    except (OSError, ValueError, IndexError):
                        continue
                    if ppid == cur:
                        kids.append(int(d)); todo.append(int(d))
        return kids


def _kill_tree(p):
    """end attempt ``p`` (a Popen started by _run_attempts, and nothing else): SIGTERM first -- a launcher then takes its workers down
    itself --, after 10 s SIGKILL to exactly the processes that descend from it"""
    import signal
    import subprocess
    tree = _descendants(p.pid)
    try:
        os.killpg(p.pid, signal.SIGTERM)
    except ProcessLookupError:
        pass
    try:
        p.wait(timeout=10)
    except subprocess.TimeoutExpired:
        pass
    for pid in tree + _descendants(p.pid) + [p.pid]:
        try:
            os.kill(pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    p.wait()


def _run_attempts(make_cmd_env, n_attempts, rank, job):
    """Run attempt 0, 1, ... until one finishes: exit code 0, or its done-marker exists (the timed region and the JSON line were
    completed; only the teardown failed).  Under a launcher every rank runs this loop for its own worker: the first supervisor that
    sees its worker fail or time out drops a `failed` marker for the attempt, every other supervisor sees it within a second, kills
    its (by then hung) worker and moves on to the next attempt with it."""
    import subprocess
    rc = 1
    for attempt in range(n_attempts):
        cmd, env = make_cmd_env(attempt)
        done, failed = _marker(job, attempt, f"r{rank}.done"), _marker(job, attempt, "failed")
        if os.path.exists(done):
            os.remove(done)
        p = subprocess.Popen(cmd, env=env, start_new_session=True)           # own process group: a hung attempt is killed as a whole
        t0 = time.perf_counter()
        why = None
        t_done = None
        while True:
            try:
                rc = p.wait(timeout=0.5)
                break
            except subprocess.TimeoutExpired:
                pass
            if os.path.exists(done):                                         # measured and printed: only the teardown is left
                t_done = t_done or time.perf_counter()
                if time.perf_counter() - t_done > 20.0:                      # ... and it does not come to an end: that is not a failure
                    rc = 0
                    break
                continue
            if time.perf_counter() - t0 > ATTEMPT_TIMEOUT_S:
                why = f"exceeded {ATTEMPT_TIMEOUT_S:.0f} s"
            elif os.path.exists(failed) and not os.path.exists(done):
                time.sleep(3.0)                                              # let a worker that is about to finish its own exit finish it
                why = "another rank's attempt failed"
            if why:
                rc = -9
                break
        if p.poll() is None:
            _kill_tree(p)
        ok = rc == 0 or os.path.exists(done)
        if os.path.exists(done):
            os.remove(done)
        if ok:
            return 0
        _touch(failed)
        print(f"[bench supervisor rank {rank}] attempt {attempt} ({ATTEMPTS[attempt] or 'default plan'}) failed: {why or 'exit code ' + str(rc)}"
              + (f"; retrying with {ATTEMPTS[attempt + 1]}" if attempt + 1 < n_attempts else ""), file=sys.stderr, flush=True)
    return rc if rc not in (0, None) else 1


def supervise(args):
    """see the comment above ATTEMPTS; returns an exit code, or None if this process is itself a worker"""
    if os.environ.get("CB_BENCH_WORKER") == "1" or args.gpus <= 1:
        return None
    pinned = "CB_BENCH_PLAN" in os.environ or "CB_COMM" in os.environ
    n_attempts = 1 if (pinned or args.mode == "infer16") else len(ATTEMPTS)
    me = os.path.join(ROOT, "bench.py")                  # (the workers run bench.py itself)
    if "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves (each of them is a plain worker; this process supervises the whole job)
        job = f"{_free_port()}_{os.getpid()}"

        def launch(attempt):
            port = _free_port()
            env = dict(os.environ, CB_BENCH_WORKER="1", CB_BENCH_ATTEMPT=str(attempt), CB_BENCH_JOB=job, MASTER_ADDR="127.0.0.1",
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **ATTEMPTS[attempt])
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), me] + sys.argv[1:]
            return cmd, env
        return _run_attempts(launch, n_attempts, 0, job)
    # under a launcher: this rank process supervises ONE worker with the same rank environment
    rank = int(os.environ.get("RANK", "0"))
    port0 = int(os.environ.get("MASTER_PORT", "29500"))
    job = f"{port0}_{os.getppid()}"                       # the launcher's pid: the same for every rank of the node

    def worker(attempt):
        env = dict(os.environ, CB_BENCH_WORKER="1", CB_BENCH_ATTEMPT=str(attempt), CB_BENCH_JOB=job, **ATTEMPTS[attempt])
        if attempt > 0:
            # the launcher's store (port0) still holds the keys of the failed attempt: a retry rendezvouses through its own TCP store,
            # hosted by its rank 0 on a port every rank derives the same way
            env["MASTER_PORT"] = str(port0 + 17 * attempt)
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
        return [sys.executable, me] + sys.argv[1:], env
    return _run_attempts(worker, n_attempts, rank, job)


def _stub_worker(args):
    """TEST HOOK (tests/test_zz_bench_cli.py, CPU): stands in for the measured worker so that the launcher / supervisor / retry path --
    rendezvous through the launcher's store on attempt 0, through a fresh TCP store on the derived port on a retry, failure markers,
    done markers -- runs end to end without a GPU.  CB_BENCH_TEST_STUB = "fail0:<rank>" makes that rank fail on attempt 0 (the others
    then hang in the collective, as real ranks would); "hang0:<rank>" makes it hang instead."""
    import torch
    import torch.distributed as dist
    mode, _, who = os.environ["CB_BENCH_TEST_STUB"].partition(":")
    rank, world, attempt = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("CB_BENCH_ATTEMPT", "0"))
    if attempt == 0 and who and rank == int(who):
        if mode == "fail0":
            raise SystemExit(7)
        time.sleep(10000)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    assert float(t) == world * (world + 1) / 2
    dist.barrier()
    if rank == 0:
        print(json.dumps({"stub": True, "n_gpus": world, "attempt": attempt, "attempt_env": ATTEMPTS[attempt], "plan_env": os.environ.get("CB_BENCH_PLAN", "")}), flush=True)
    if os.environ.get("CB_BENCH_JOB"):
        _touch(_marker(os.environ["CB_BENCH_JOB"], attempt, f"r{rank}.done"))
    dist.destroy_process_group()


