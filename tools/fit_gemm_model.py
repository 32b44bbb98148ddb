#!/usr/bin/env python
"""Launch-cost model of cb_gemm for problem shapes that are NOT in the tuned table (csrc/gemm_tuned.h).

  python tools/fit_gemm_model.py fit   [sweep.json ...]      # -> clipbert_amd/csrc/gemm_model.h + profiles/r03m_gemm_model_fit.json
  python tools/fit_gemm_model.py check [sweep.json ...]      # what the BUILT library picks (cb_gemm_plan, table off) against the sweep

The sweeps are tools/tune_gemm.py --cold outputs (every cb_gemm problem of the bench steps x tile x K-loop schedule x K split, timed
behind a cache flush -- the state kernels meet inside the training step).  The model predicts the duration of one launch of a
configuration (tile t, K split s, schedule) on a problem (form f in {fwd, dgrad, wgrad}, M, N, K, batch, taps):

    wg     = ceil(M / BM_t) * ceil(N / BN_t) * batch * s                        workgroups
    r      = wg / (256 * occ_t)                                                 rounds of the grid over the CUs
    rounds = q * ceil(r) + (1 - q) * max(1, r)                                  q = q4 (4-wave tiles) or q8 (8-wave: one workgroup per CU, occ = 1)
    T      = a_t + rounds * (b_tf + ceil(ceil(K / 64) / s) * c_tf * (1 + g_taps * [taps > 1]) * (1 + g_m2 * [schedule 2]))
             + (A + B bytes) / bw_ab + (C bytes) / bw_c + (8-wave split: slab bytes) / bw_red + (4-wave split: atomic bytes) / bw_atom
             + d_m2 * [schedule 2]

a, occ per tile, b, c per (tile, form), the rest global: 7 + 7 + 21 + 21 + 9 = 65 numbers fitted by least squares on log T over all
measured (problem, configuration) pairs.  cb_gemm evaluates it for every configuration that is legal for the call and launches the
argmin (csrc/gemm.hip).  `fit` reports the regret of those choices against the per-problem best of the sweep, in sample and 5-fold
cross-validated over PROBLEMS (the number that says how it does on shapes it has not seen)."""
import ctypes as C
import json
import math
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_SWEEPS = [os.path.join(ROOT, "profiles", f) for f in
                  ("r03f_gemm_tuning_cold.json",          # the bench workloads (train 2x2 / tgif / infer16 / 448 px x 4 clips)
                   "r03m_sweep_v10_t24_cold.json",        # train, 10 videos, 24 text tokens
                   "r03o_sweep_fit_cold.json")]           # train 6 videos x 40 tokens, 24 videos, 192 px, 4 clips x 1 frame; tgif 6 videos; infer16 x 32 captions
DEFAULT_HOLDOUT = os.path.join(ROOT, "profiles", "r03o_sweep_holdout_cold.json")    # train 12 videos x 20 tokens x 256 px; tgif 10 videos: never fitted
#            name         BM   BN  model index  cb_gemm tile id
TILES = {"64x64": (64, 64, 0, 2), "128x64": (128, 64, 1, 3), "128x128": (128, 128, 2, 1), "128x128o2": (128, 128, 3, 4),
         "8w256x256": (256, 256, 4, 5), "8w128x256": (128, 256, 5, 6), "8w256x128": (256, 128, 6, 7)}
NT, NG = 7, 9
FORM = {"fwd": 0, "dgrad": 1, "wgrad": 2}


def parse(cfg):
    parts = cfg.split("/")
    s = ([int(x[1:]) for x in parts[2:] if x[0] == "s"] or [0])[0]
    m = ([int(x[1:]) for x in parts[2:] if x[0] == "m"] or [None])[0]
    return parts[0], (parts[1] if len(parts) > 1 else "xcd"), s, m


def feats(p, cfg):
    tile, _xcd, s, m = parse(cfg)
    bm, bn, ti, _ = TILES[tile]
    M, N, K, b = p["M"], p["N"], p["K"], p["batch"]
    s_eff = max(s, 1) if ti >= 4 else max(s if s else p["split_k"], 1)
    wg = -(-M // bm) * -(-N // bn) * b * s_eff
    kt = -(-(-(-K // 64)) // s_eff)
    form = FORM[p["form"]]
    return dict(ti=ti, wg=wg, kt=kt, form=form, taps=int(p["taps"] > 1), m2=int(m == 2), ab=(M * K + N * K) * 2 * b,
                cb=M * N * b * (4 if form == 2 else 2), red=(M * N * b * s_eff * 8) if (ti >= 4 and s_eff > 1) else 0,
                atom=(M * N * b * s_eff * 4) if (ti < 4 and s_eff > 1) else 0)


def predict(x, f):
    ti, fo = f["ti"], f["form"]
    a, occ, b, c, g = x[ti], x[NT + ti], x[2 * NT + ti * 3 + fo], x[5 * NT + ti * 3 + fo], x[8 * NT:]
    r = f["wg"] / (256 * occ)
    q = g[7] if ti >= 4 else g[3]
    rounds = q * math.ceil(r - 1e-9) + (1 - q) * max(1.0, r)
    ck = c * (1 + g[4] * f["taps"]) * (1 + g[8] * f["m2"])
    return (a + rounds * (b + f["kt"] * ck) + f["ab"] / (g[0] * 1e6) + f["cb"] / (g[5] * 1e6) + f["red"] / (g[1] * 1e6) + f["atom"] / (g[2] * 1e6)
            + g[6] * f["m2"])


def load(paths):
    probs = []
    for path in paths:
        probs += [p for p in json.load(open(path))["problems"] if p.get("us")]
    return probs


def measured(p):
    return {k: v for k, v in p["us"].items() if v and k != "auto"}


def rows_of(probs):
    return [(i, feats(p, c), v) for i, p in enumerate(probs) for c, v in measured(p).items() if parse(c)[1] != "rr"]


def predict_vec(x, F):
    """predict() over arrays (F: dict of numpy arrays, one entry per measurement)"""
    ti, fo = F["ti"], F["form"]
    a, occ, b, c, g = x[ti], x[NT + ti], x[2 * NT + ti * 3 + fo], x[5 * NT + ti * 3 + fo], x[8 * NT:]
    r = F["wg"] / (256 * occ)
    q = np.where(ti >= 4, g[7], g[3])
    rounds = q * np.ceil(r - 1e-9) + (1 - q) * np.maximum(1.0, r)
    ck = c * (1 + g[4] * F["taps"]) * (1 + g[8] * F["m2"])
    return (a + rounds * (b + F["kt"] * ck) + F["ab"] / (g[0] * 1e6) + F["cb"] / (g[5] * 1e6) + F["red"] / (g[1] * 1e6) + F["atom"] / (g[2] * 1e6)
            + g[6] * F["m2"])


def fit_rows(rows):
    from scipy.optimize import least_squares
    F = {k: np.array([f[k] for _, f, _ in rows], dtype=np.int64 if k in ("ti", "form") else np.float64) for k in rows[0][1]}
    logv = np.log(np.array([v for _, _, v in rows]))
    x0 = np.array([5.0] * NT + [4, 2, 1, 1.9, 1, 1, 1] + [2.0] * (3 * NT) + [0.3] * 3 + [0.4] * 3 + [0.6] * 3 + [0.6] * 3 + [1.2] * 3 + [1.0] * 6
                  + [3.0, 3.0, 1.0, 0.5, 0.2, 3.0, 0.0, 0.0, 0.0])
    lb = np.array([0.0] * NT + [1.0, 1.0, 0.5, 0.5] + [0.999] * 3 + [0.0] * (3 * NT) + [0.01] * (3 * NT) + [0.3, 0.3, 0.1, 0, -0.5, 0.3, -5, 0, -0.5])
    ub = np.array([50.0] * NT + [6.0, 3.0, 1.001, 2.001] + [1.001] * 3 +      # (occupancy: at most what the kernels' LDS / registers allow)
                  [50.0] * (3 * NT) + [10.0] * (3 * NT) + [20, 20, 20, 1, 3, 20, 5, 1, 0.5])
    x0[8 * NT + 7] = 0.8                    # (the 8-wave tiles hold one workgroup per CU: their grid is quantised in whole rounds)
    res = lambda x: np.log(np.maximum(predict_vec(x, F), 1e-3)) - logv       # noqa: E731
    return least_squares(res, x0, bounds=(lb, ub), max_nfev=400).x


def choose(x, p):
    cands = [c for c in measured(p) if parse(c)[1] != "rr"]
    return min(cands, key=lambda c: predict(x, feats(p, c)))


def regret(pick, probs):
    tb = th = 0.0
    for p in probs:
        us = measured(p)
        w = sum(p["count"].values())
        tb += w * min(us.values())
        th += w * us[pick(p)]
    return tb, th


def write_header(x, path, note):
    a, occ, b, c, g = x[:NT], x[NT:2 * NT], x[2 * NT:5 * NT], x[5 * NT:8 * NT], x[8 * NT:]
    fmt = lambda v: ", ".join(f"{float(t):.6g}" for t in v)          # noqa: E731
    with open(path, "w") as fh:
        fh.write("// GENERATED by tools/fit_gemm_model.py fit -- do not edit.  " + note + "\n"
                 "// Launch-cost model of cb_gemm (microseconds) for shapes outside gemm_tuned.h; see the tool's docstring for the formula.\n"
                 "// model index: 0 64x64, 1 128x64, 2 128x128, 3 128x128 (2 blocks/CU), 4 8-wave 256x256, 5 8-wave 128x256, 6 8-wave 256x128\n"
                 "#pragma once\nnamespace cbgemm {\n"
                 f"static const double MODEL_A[7] = {{{fmt(a)}}};\n"
                 f"static const double MODEL_OCC[7] = {{{fmt(occ)}}};\n"
                 f"static const double MODEL_B[7][3] = {{{', '.join('{' + fmt(b[i * 3:i * 3 + 3]) + '}' for i in range(NT))}}};     // [tile][fwd, dgrad, wgrad]\n"
                 f"static const double MODEL_C[7][3] = {{{', '.join('{' + fmt(c[i * 3:i * 3 + 3]) + '}' for i in range(NT))}}};\n"
                 "// bw_ab, bw_red, bw_atom (bytes / us / 1e6), ceil weight q of the 4-wave tiles, g_taps, bw_c, d_m2 (us), ceil weight of the 8-wave tiles, g_m2\n"
                 f"static const double MODEL_G[9] = {{{fmt(g)}}};\n"
                 "}  // namespace cbgemm\n")


def rules_pick(p):
    """the round-2 rules of thumb this model replaced (4-wave tiles only), for comparison"""
    M, N, K, zmul = p["M"], p["N"], p["K"], p["split_k"] * p["batch"]
    t128, t12864, kred = -(-M // 128) * -(-N // 128) * zmul, -(-M // 128) * -(-N // 64) * zmul, K // p["split_k"]
    tile = "64x64"
    if p["a_mode"] == 2:
        if t128 >= 256 and kred >= 1024:
            tile = "128x128o2"
    elif N > 64:
        if t128 >= 350 and kred >= 256:
            tile = "128x128o2"
        elif t12864 >= 200 and kred >= 512:
            tile = "128x64"
    return tile + "/xcd"


# ---- what the built library picks -------------------------------------------------------------------------------------------------
def lib_pick(lib, GemmDesc, p, use_table=0):
    d = GemmDesc()
    C.memset(C.byref(d), 0, C.sizeof(d))
    d.dtype = 1
    for k in ("M", "N", "K"):
        setattr(d, k, p[k])
    d.a_mode, d.b_mode, d.batch, d.split_k = p["a_mode"], p["b_mode"], p["batch"], p["split_k"]
    M, N, K, b = p["M"], p["N"], p["K"], p["batch"]
    taps = p["taps"]
    wgrad = p["form"] == "wgrad"
    d.A = d.B = d.C = 1 << 20                                           # (aligned dummies: nothing is launched)
    d.a_bytes = d.b_bytes = 1 << 30
    d.lda = M if p["a_mode"] == 2 else K
    d.ldb = {0: K, 2: N, 3: N * taps, 4: 0}[p["b_mode"]]
    d.ldc = N
    if p["a_mode"] == 1 or p["b_mode"] in (3, 4):                       # conv forms: channels per tap from the mode's K / N contract
        d.R, d.S = (3, 3) if taps == 9 else (taps, 1)
        d.Cin = (N if p["b_mode"] == 4 else K) // taps
        d.H = d.W = 8
        d.sW = d.Cin
        d.sH = 8 * d.Cin
        d.a_tab = (1 << 20) if p["a_mode"] == 1 else 0
        d.b_tab = (1 << 20) if p["b_mode"] == 4 else 0
    d.c_f32 = 1 if wgrad else 0
    d.accumulate = 1 if wgrad else 0
    if b > 1:
        d.batch_stride_a, d.batch_stride_b, d.batch_stride_c = K * M, K * N, M * N
    d.splitk_ws = 1 << 20
    d.splitk_ws_bytes = 128 << 20
    out = (C.c_int32 * 4)()
    rc = lib.cb_gemm_plan(C.byref(d), use_table, out)
    if rc != 0:
        raise RuntimeError(lib.cb_last_error().decode())
    tile, split, sched, _xcd = out
    name = {v[3]: k for k, v in TILES.items()}[tile]
    if tile >= 5:
        return f"{name}/xcd/s{split}/m{sched - 1}"
    return f"{name}/xcd" + (f"/s{split}" if split != p["split_k"] else "")


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else "fit"
    paths = sys.argv[2:] or (DEFAULT_SWEEPS if cmd == "fit" else DEFAULT_SWEEPS + [DEFAULT_HOLDOUT])
    probs = load(paths)
    if cmd == "fit":
        rows = rows_of(probs)
        x = fit_rows(rows)
        err = np.array([math.log(predict(x, f)) - math.log(v) for _, f, v in rows])
        tb, th = regret(lambda p: choose(x, p), probs)
        random.seed(0)
        idx = list(range(len(probs)))
        random.shuffle(idx)
        TB = TH = 0.0
        for k in range(5):
            test = set(idx[k::5])
            xr = fit_rows([r for r in rows if r[0] not in test])
            b_, h_ = regret(lambda p: choose(xr, p), [probs[i] for i in test])
            TB, TH = TB + b_, TH + h_
        rules = lambda p: rules_pick(p) if rules_pick(p) in measured(p) else min(measured(p), key=measured(p).get)      # noqa: E731
        rb, rh = regret(rules, probs)
        rep = dict(sweeps=[os.path.relpath(p, ROOT) for p in paths], problems=len(probs), measurements=len(rows), rms_log_error=float(err.std()),
                   in_sample=dict(best_ms=tb / 1e3, model_ms=th / 1e3, regret_pct=100 * (th / tb - 1), round2_rules_regret_pct=100 * (rh / rb - 1)),
                   cross_validated_5fold=dict(best_ms=TB / 1e3, model_ms=TH / 1e3, regret_pct=100 * (TH / TB - 1)), parameters=[float(v) for v in x])
        if os.path.exists(DEFAULT_HOLDOUT) and DEFAULT_HOLDOUT not in [os.path.abspath(p) for p in paths]:
            hold = load([DEFAULT_HOLDOUT])
            hb, hh = regret(lambda p: choose(x, p), hold)
            qb, qh = regret(rules, hold)
            rep["held_out"] = dict(sweep=os.path.relpath(DEFAULT_HOLDOUT, ROOT), problems=len(hold), best_ms=hb / 1e3, model_ms=hh / 1e3,
                                   regret_pct=100 * (hh / hb - 1), round2_rules_regret_pct=100 * (qh / qb - 1))
        print(json.dumps({k: v for k, v in rep.items() if k != "parameters"}, indent=1))
        write_header(x, os.path.join(ROOT, "clipbert_amd", "csrc", "gemm_model.h"),
                     f"{len(rows)} measurements of {len(probs)} problems; regret vs the sweep's best: 5-fold CV {rep['cross_validated_5fold']['regret_pct']:.1f} %"
                     + (f", held-out workloads {rep['held_out']['regret_pct']:.1f} %" if "held_out" in rep else ""))
        json.dump(rep, open(os.path.join(ROOT, "profiles", "r03m_gemm_model_fit.json"), "w"), indent=1)
    else:
        sys.path.insert(0, ROOT)
        from clipbert_amd import _lib
        lib = _lib.get()
        picks, missing = {}, 0

        def pick(p):
            c = lib_pick(lib, _lib.GemmDesc, p)
            if c not in measured(p):
                nonlocal missing
                missing += 1
                tile = parse(c)[0]
                same = [k for k in measured(p) if parse(k)[0] == tile and parse(k)[1] != "rr"]
                c = min(same, key=lambda k: abs(math.log2(max(parse(k)[2], 1)) - math.log2(max(parse(c)[2], 1)))) if same else min(measured(p), key=measured(p).get)
            picks[(p["form"], p["M"], p["N"], p["K"], p["batch"], p["taps"])] = c
            return c
        for path in paths:
            ps = load([path])
            missing = 0
            tb, th = regret(pick, ps)
            rb, rh = regret(lambda p: rules_pick(p) if rules_pick(p) in measured(p) else min(measured(p), key=measured(p).get), ps)
            print(f"{os.path.relpath(path, ROOT)} ({len(ps)} problems): round-2 rules regret {100 * (rh / rb - 1):.1f} %, library (table off) "
                  f"{100 * (th / tb - 1):.1f} % (best {tb / 1e3:.2f} ms, picked {th / 1e3:.2f} ms; {missing} picks outside the sweep -> nearest measured split)")


if __name__ == "__main__":
    main()
