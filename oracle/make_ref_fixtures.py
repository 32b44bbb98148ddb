"""Golden vectors of the reference's pure host functions.  TEST INFRASTRUCTURE ONLY.   python -m oracle.make_ref_fixtures

tests/golden/ref_retrieval_metrics.json: seeded score rows -> the metric dicts the REFERENCE's eval_retrieval /
get_retrieval_metric_from_bool_matrix return (run_video_retrieval.py:519-625, executed from source by oracle/ref_functions.py).  The GPU
box (no /root/reference) compares clipbert_amd.tasks.eval_retrieval with them; here tests/test_reference_functions.py also runs the
reference functions live."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_functions as RF     # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_retrieval_metrics.json")


def cases():
    """(name, rows, gt) -- rows as the inference loop emits them (vid_id, txt_id, score); deterministic"""
    out = []
    for seed, (n_txt, n_vid, dup, ties) in enumerate([(12, 12, False, False), (40, 40, True, False), (25, 25, False, True), (7, 7, True, True)]):
        rng = np.random.default_rng(100 + seed)
        sm = rng.random((n_txt, n_vid)).astype(np.float32)
        if ties:
            sm = np.round(sm, 1)                       # many equal scores, like scores rounded to 4 places on a large pool
        perm = rng.permutation(n_vid)
        gt = {f"t{i}": f"v{int(perm[i])}" for i in range(n_txt)}
        rows = [dict(vid_id=f"v{j}", txt_id=f"t{i}", score=float(sm[i, j])) for j in range(n_vid) for i in range(n_txt)]   # video-major, as inference emits
        if dup:
            rows += [dict(vid_id="v0", txt_id=f"t{i}", score=9.0) for i in range(n_txt)]                                    # a video seen twice: ignored
        out.append((f"n{n_txt}x{n_vid}{'_dup' if dup else ''}{'_ties' if ties else ''}", rows, gt))
    return out


def main():
    fns = RF.retrieval_metric_functions()
    data = {}
    for name, rows, gt in cases():
        res = fns["eval_retrieval"](rows, gt, None)
        data[name] = {d: {k: float(v) for k, v in m.items()} for d, m in res.items()}
        print(name, data[name])
    with open(OUT, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
