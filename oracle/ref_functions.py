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
OPT_UTILS = os.path.join("src", "optimization", "utils.py")
GRID_FEAT = os.path.join("src", "modeling", "grid_feat.py")
DATA_UTILS = os.path.join("src", "datasets", "data_utils.py")
BASIC_UTILS = os.path.join("src", "utils", "basic_utils.py")


class _StripCuda(ast.NodeTransformer):
    """x.cuda() -> x  (the reference's ImageNorm pins its constants to a GPU in __init__; nothing else about it needs one)"""
    def visit_Call(self, node):
        self.generic_visit(node)
        if isinstance(node.func, ast.Attribute) and node.func.attr == "cuda" and not node.args and not node.keywords:
            return node.func.value
        return node


def load(rel_path: str, names, extra_ns=None, strip_cuda: bool = False):
    """the named top-level functions / classes of a reference file, executed from its source (no other module code runs)"""
    path = os.path.join(ref_shim.REFERENCE_ROOT, rel_path)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(body) == len(names), (rel_path, names, [n.name for n in body])
    for fn in body:
        fn.decorator_list = []
    if strip_cuda:
        body = [ast.fix_missing_locations(_StripCuda().visit(n)) for n in body]
    ns = {"np": np, "torch": torch, "nn": torch.nn, "defaultdict": defaultdict, "os": os, "Dict": Dict, "Any": Any}
    ns.update(extra_ns or {})
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return {n: ns[n] for n in names}


def load_method(rel_path: str, cls_name: str, method: str, extra_ns=None):
    """one METHOD of a reference class as a plain function ``f(self, ...)``, executed from its source: the class statement is found by
    name, the method's ``def`` is lifted out of it (decorators dropped) and compiled alone -- neither the class body nor the module's
    imports (detectron2, horovod, apex) run.  ``super()`` calls inside such a method would not work; none of the methods taken this way
    has one."""
    path = os.path.join(ref_shim.REFERENCE_ROOT, rel_path)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == method)
    fn.decorator_list = []
    ns = {"np": np, "torch": torch, "nn": torch.nn, "os": os}
    ns.update(extra_ns or {})
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns[method]


def _import_file(rel_path: str, mod_name: str):
    """a reference module that is plain torch / stdlib at module level, imported from where it lies"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(mod_name, os.path.join(ref_shim.REFERENCE_ROOT, rel_path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def adamw_class():
    """AdamW (src/optimization/adamw.py:12-103)"""
    return _import_file(os.path.join("src", "optimization", "adamw.py"), "_ref_adamw").AdamW


def sched_module():
    """noam_schedule / warmup_linear / multi_step_schedule / get_lr_sched (src/optimization/sched.py)"""
    return _import_file(os.path.join("src", "optimization", "sched.py"), "_ref_sched")


def group_builder():
    """build_e2e_optimizer_w_lr_mul (src/optimization/utils.py:131-161): the four parameter groups of one half of the model"""
    return load(OPT_UTILS, ["build_e2e_optimizer_w_lr_mul"])["build_e2e_optimizer_w_lr_mul"]


def conv3x3():
    """conv3x3 (src/modeling/grid_feat.py:16-34): the grid encoder's convolution factory"""
    return load(GRID_FEAT, ["conv3x3"])["conv3x3"]


def grid_encoder(backbone_channel_in_size: int, hidden_size: int):
    """the grid encoder EXACTLY as GridFeatBackbone.__init__ builds it (src/modeling/grid_feat.py:43-48): the right-hand side of the
    ``self.grid_encoder = nn.Sequential(...)`` statement is taken from the class source and evaluated with the reference's own conv3x3
    (the rest of __init__ needs detectron2 and is not run)"""
    from types import SimpleNamespace
    path = os.path.join(ref_shim.REFERENCE_ROOT, GRID_FEAT)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GridFeatBackbone")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    stmt = next(n for n in ast.walk(init) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Attribute) and n.targets[0].attr == "grid_encoder")
    ns = {"nn": torch.nn, "torch": torch, "conv3x3": conv3x3(),
          "config": SimpleNamespace(backbone_channel_in_size=backbone_channel_in_size, hidden_size=hidden_size)}
    return eval(compile(ast.Expression(body=stmt.value), path, "eval"), ns)


def grid_feat_forward():
    """GridFeatBackbone.forward (src/modeling/grid_feat.py:89-105) as ``f(self, x)``: the (B, T) -> (B T) reshape, the RGB -> BGR flip,
    backbone -> get_conv5_features -> grid_encoder, the (B, T, C, H, W) view and the channels-last permute.  ``self`` is supplied by the
    caller (a namespace with ``input_format``, ``feature.backbone``, ``feature.roi_heads`` and ``grid_encoder``): detectron2 builds the
    real ones and is absent from the image."""
    return load_method(GRID_FEAT, "GridFeatBackbone", "forward")


def roi_heads_name() -> str:
    """MODEL.ROI_HEADS.NAME of the detectron2 config the reference's JSON configs point at (src/configs/detectron2_configs/R-50-grid.yaml
    through its _BASE_): which class's get_conv5_features GridFeatBackbone.forward calls"""
    import yaml
    base = os.path.join(ref_shim.REFERENCE_ROOT, "src", "configs", "detectron2_configs")
    with open(os.path.join(base, "R-50-grid.yaml")) as fh:
        top = yaml.safe_load(fh)
    with open(os.path.join(base, top["_BASE_"])) as fh:
        parent = yaml.safe_load(fh)
    roi = dict(parent["MODEL"]["ROI_HEADS"])
    roi.update(top.get("MODEL", {}).get("ROI_HEADS", {}))
    assert roi["IN_FEATURES"] == ["res5"] and parent["MODEL"]["RESNETS"]["OUT_FEATURES"] == ["res5"]
    return roi["NAME"]


def get_conv5_features():
    """get_conv5_features of the ROI-heads class the config names (src/modeling/grid_feats/roi_heads.py:232-236 for
    AttributeStandardROIHeads: the identity select of the single in_feature) as ``f(self, features)``"""
    return load_method(os.path.join("src", "modeling", "grid_feats", "roi_heads.py"), roi_heads_name(), "get_conv5_features")


def clipbert_forward():
    """ClipBert.forward (src/modeling/e2e_model.py:29-39) as ``f(self, batch)`` with the reference's own repeat_tensor_rows in scope;
    ``self`` (cnn, transformer, retrieval) is the caller's"""
    return load_method(os.path.join("src", "modeling", "e2e_model.py"), "ClipBert", "forward", extra_ns={"repeat_tensor_rows": repeat_tensor_rows()})


def repeat_tensor_rows():
    """repeat_tensor_rows (src/datasets/data_utils.py:344-357) with its helper flat_list_of_lists (src/utils/basic_utils.py)"""
    helper = load(BASIC_UTILS, ["flat_list_of_lists"])
    return load(DATA_UTILS, ["repeat_tensor_rows"], extra_ns=helper)["repeat_tensor_rows"]


def image_norm_class():
    """ImageNorm (src/datasets/data_utils.py:256-276), its two .cuda() calls removed by an AST edit"""
    return load(DATA_UTILS, ["ImageNorm"], strip_cuda=True)["ImageNorm"]


def retrieval_metric_functions():
    """get_retrieval_metric_from_bool_matrix, get_retrieval_scores, eval_retrieval (run_video_retrieval.py:519-625)"""
    return load(RETRIEVAL_RUNNER, ["get_retrieval_metric_from_bool_matrix", "get_retrieval_scores", "eval_retrieval"])


def torchvision_converter():
    """convert_torchvision_ckpt_to_detectron2 (src/utils/load_save.py:315-363)"""
    return load(LOAD_SAVE, ["convert_torchvision_ckpt_to_detectron2"])["convert_torchvision_ckpt_to_detectron2"]
