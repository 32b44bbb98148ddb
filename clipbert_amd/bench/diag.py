#!/usr/bin/env python
"""Diagnostic step variants of bench.py (never the reported metric), moved out of bench.py in round 5:

  CB_BENCH_PIPELINE=1   software-pipelined optimizer: the transformer groups' AdamW of step i at the start of step i+1's graph, on a side
                        stream beside the ResNet forward -- measured SLOWER than the plain plan (round 3: 11.70-11.81 vs 11.52-11.56 ms)
  CB_BENCH_CHAINS=2     the videos split into two independent forward + backward chains on two HIP streams

`build(ctx)` takes bench.py's objects (model, bank, opt, batch tensors, ...) as a namespace and returns the two capturable step functions."""
import os
from types import SimpleNamespace

import torch

T_GROUPS, C_GROUPS = (0, 1, 2, 3), (4, 5, 6, 7)


def build(ctx):
    from clipbert_amd import ops, tasks
    model, bank, opt, args = ctx.model, ctx.bank, ctx.opt, ctx.args
    frames, ids, mask, labels, counts, tcfg, one = ctx.frames, ctx.ids, ctx.mask, ctx.labels, ctx.counts, ctx.tcfg, ctx.one
    bv, nclip, T, rep, fold, train, world = ctx.bv, ctx.nclip, ctx.T, ctx.rep, ctx.fold, ctx.train, ctx.world
    # Software-pipelined optimizer (1 GPU, opt-in: CB_BENCH_PIPELINE=1 -- measured SLOWER than the plain plan, 11.70-11.81 vs 11.52-11.56 ms
    # on the same box, round 3: the streaming update beside the ResNet forward costs that forward more than it hides): the AdamW update of the TRANSFORMER groups (111 M of the
    # 148.6 M parameters, an HBM-streaming kernel) of step i runs at the start of step i+1's graph on a side stream, beside the
    # ResNet forward -- which reads only CNN weights and is latency-bound, not HBM-bound.  Step i itself ends with the grad-norm
    # reduction over ALL gradients and the update of the CNN groups.  Same arithmetic in the same per-parameter order (the deferred
    # launch reads step i's hyper-parameters and norm); the update left over after the last timed step is flushed INSIDE the timed region.
    pipe_stream = torch.cuda.Stream() if (train and world == 1) else None

    def device_step_pipelined():
        t_end = bank.group_range[3][1]
        cur = torch.cuda.current_stream()
        pipe_stream.wait_stream(cur)
        with torch.cuda.stream(pipe_stream):
            opt.launch(groups=T_GROUPS, prev=True, reuse_norm=True)          # step i-1's transformer update (no-op before the first step)
            bank.zero_grad_range(0, t_end, lazy=True)                        # ... then its gradients may go
        bank.grad_epoch = getattr(bank, "grad_epoch", 0) + 1
        bank.zero_grad_range(t_end, bank.grad.numel())
        bank.lazy_fresh = bank.lazy_span is not None
        model.rt.pending_encoder_nodes = model.rt.pending_cnn_nodes = 0
        vis = frames.view(bv * nclip, T, *frames.shape[2:]) if (fold and nclip > 1) else frames
        grid = model.grid_features(vis)                                      # ResNet forward beside the deferred update
        cur.wait_stream(pipe_stream)
        mini = dict(visual_inputs=grid, text_input_ids=ids, text_input_mask=mask, labels=None, n_examples_list=tasks._pair_counts(tcfg, counts))
        lg = model.forward_from_grid(mini, clip_fold=nclip)["logits"]
        stack = lg.reshape(nclip, lg.shape[0] // nclip, *lg.shape[1:])
        loss = tasks.training_loss(model, stack, labels, counts, args.pool)
        loss.backward(one)
        ops.counter_add(model.rt.seed_dev)
        opt.launch(groups=C_GROUPS)                                          # norm over ALL gradients + the CNN groups' update
        return loss

    # diagnostic (CB_BENCH_CHAINS=2): the videos split into two independent forward+backward chains on two HIP streams, so that
    # the launch ramps / tails of one chain's kernels overlap the other's main loops
    chains = int(os.environ.get("CB_BENCH_CHAINS", "1"))
    assert chains == 1 or (world == 1 and train), "CB_BENCH_CHAINS is a single-GPU training diagnostic"
    chain_streams = [torch.cuda.Stream() for _ in range(chains)] if chains > 1 else []

    def device_step_chains():
        opt.zero_grad(lazy=False)
        model.rt.pending_encoder_nodes = 0
        cur = torch.cuda.current_stream()
        per, prep = bv // chains, rep * (bv // chains)
        total = None
        for h, st in enumerate(chain_streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                sub = dict(visual_inputs=frames[h * per:(h + 1) * per], text_input_ids=ids[h * prep:(h + 1) * prep],
                           text_input_mask=mask[h * prep:(h + 1) * prep], labels=labels[h * (len(labels) // chains):(h + 1) * (len(labels) // chains)],
                           n_examples_list=counts[h * per:(h + 1) * per])
                stack = tasks.forward_clips_stack(model, sub, nclip, T, fold=fold, cfg=tcfg)
                loss_h = tasks.training_loss(model, stack, sub["labels"], sub["n_examples_list"], args.pool) / chains
                loss_h.backward()
                total = loss_h.detach()
        for st in chain_streams:
            cur.wait_stream(st)
        ops.counter_add(model.rt.seed_dev)
        opt.launch()
        return total

    return SimpleNamespace(device_step_pipelined=device_step_pipelined, device_step_chains=device_step_chains, chains=chains,
                           pipe_stream=pipe_stream, T_GROUPS=T_GROUPS, C_GROUPS=C_GROUPS)
