"""Pure functions of the reference, EXECUTED FROM THEIR SOURCE where it lies.  TEST INFRASTRUCTURE ONLY (needs /root/reference).

The reference's runner modules import horovod / apex / detectron2 at module level and cannot be imported here; their metric and
checkpoint-conversion functions however are plain numpy / torch.  ``load(path, names)`` parses the file, takes the named top-level
functions (ast, no other module code runs) and execs them in a namespace holding what they use (np, torch, defaultdict, os, Dict, Any).
Nothing is copied into the repo: the source is read at call time.
"""
import ast
import os
from collections import defaultdict
from typing import Any, Dict

import numpy as np
import torch

from . import ref_shim

RETRIEVAL_RUNNER = os.path.join("src", "tasks", "run_video_retrieval.py")
LOAD_SAVE = os.path.join("src", "utils", "load_save.py")


def load(rel_path: str, names):
    path = os.path.join(ref_shim.REFERENCE_ROOT, rel_path)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), (rel_path, names, [n.name for n in body])
    for fn in body:
        fn.decorator_list = []
    ns = {"np": np, "torch": torch, "defaultdict": defaultdict, "os": os, "Dict": Dict, "Any": Any}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return {n: ns[n] for n in names}


def retrieval_metric_functions():
    """get_retrieval_metric_from_bool_matrix, get_retrieval_scores, eval_retrieval (run_video_retrieval.py:519-625)"""
    return load(RETRIEVAL_RUNNER, ["get_retrieval_metric_from_bool_matrix", "get_retrieval_scores", "eval_retrieval"])


def torchvision_converter():
    """convert_torchvision_ckpt_to_detectron2 (src/utils/load_save.py:315-363)"""
    return load(LOAD_SAVE, ["convert_torchvision_ckpt_to_detectron2"])["convert_torchvision_ckpt_to_detectron2"]
