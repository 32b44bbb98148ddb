"""Checkpoint compatibility with the reference.

* ``ModelSaver`` / ``E2E_TrainingRestorer``: the saver and the resume file of src/utils/load_save.py:43-68,245-312 --
  ``model_step_N.pt`` is a plain CPU state dict with the REFERENCE's key layout (our modules use the same names and the
  same logical OIHW conv shapes, so a file written here loads into the reference and vice versa);
  ``model_step_N_train_state.pt`` / ``restore.pt`` carry the optimizer state of clipbert_amd.optim.FusedAdamW.
* ``load_detectron2_backbone``: what DetectionCheckpointer(self.feature).resume_or_load does for ClipBERT's use of it
  (src/modeling/grid_feat.py:72-80): a detectron2 ``.pth`` (``{"model": {...}}``, keys ``backbone.*``, RPN / ROI heads
  ignored), a detectron2 model-zoo ``.pkl`` (numpy arrays, same keys or the Caffe2-era names are NOT handled), or a
  torchvision ResNet-50 state dict (renamed as convert_torchvision_ckpt_to_detectron2 does, load_save.py:315-363).

Host code only; weights land in the fp32 masters of the parameter bank and the bf16 compute copies are refreshed by the
model's load_state_dict hook.

Ranks (the reference: restore on EVERY rank, then saver / restorer replaced by NoOp on ranks != 0, run_video_retrieval.py:
329-346): build ``ModelSaver`` and ``E2E_TrainingRestorer`` on every rank; both write on rank 0 only, the restorer reads on
all ranks.  File formats: ``model_step_N.pt`` is interchangeable with the reference; ``*_train_state.pt`` / ``restore.pt``
hold clipbert_amd.optim.FusedAdamW's own state dict (moments keyed by parameter NAME) plus the dropout counters of the
runtime -- not loadable by the reference's AdamW, and vice versa."""
import os
import pickle
from typing import Any, Dict, Optional

import numpy as np
import torch

# torchvision -> detectron2 module names (load_save.py:335-345)
TORCHVISION_TO_DETECTRON2 = (("layer1", "res2"), ("layer2", "res3"), ("layer3", "res4"), ("layer4", "res5"),
                             ("bn1", "conv1.norm"), ("bn2", "conv2.norm"), ("bn3", "conv3.norm"),
                             ("downsample.0", "shortcut"), ("downsample.1", "shortcut.norm"))


def convert_torchvision_to_detectron2(state_dict: Dict[str, Any]) -> Dict[str, Any]:
    """Renames a torchvision ResNet state dict to detectron2 backbone names (``stem.conv1.weight``, ``res2.0.conv1.norm.*``
    ...); ``fc.*`` and ``num_batches_tracked`` entries are dropped (detectron2's matching heuristics ignore them)."""
    out = {}
    for name, v in state_dict.items():
        if name.startswith("fc.") or name.endswith("num_batches_tracked"):
            continue
        for old, new in TORCHVISION_TO_DETECTRON2:
            name = name.replace(old, new)
        if not name.startswith("res"):
            name = "stem." + name
        out[name] = v
    return out


def _read_any(path: str) -> Dict[str, Any]:
    if path.endswith(".pkl"):
        with open(path, "rb") as fh:
            obj = pickle.load(fh, encoding="latin1")
    else:
        obj = torch.load(path, map_location="cpu")
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        obj = obj["model"]
    return obj


def detectron2_backbone_state(sd: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """Any of the accepted layouts -> {"backbone.<detectron2 name>": tensor} (the key layout of GridFeatBackbone.feature)."""
    keys = list(sd.keys())
    if any(k.startswith("layer1.") for k in keys):                         # torchvision
        sd = convert_torchvision_to_detectron2(sd)
        keys = list(sd.keys())
    out = {}
    for k in keys:
        v = sd[k]
        name = k
        for pre in ("cnn.feature.", "feature.", "module."):
            if name.startswith(pre):
                name = name[len(pre):]
        if name.startswith(("stem.", "res2.", "res3.", "res4.", "res5.")):
            name = "backbone." + name
        if not name.startswith("backbone."):
            continue                                                    # proposal_generator.*, roi_heads.*, pixel_mean/std
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        out[name] = torch.as_tensor(v)
    return out


def load_detectron2_backbone(cnn, path_or_state) -> int:
    """Loads backbone weights into ``cnn.feature`` (cnn = GridFeatBackbone); returns the number of tensors loaded."""
    sd = _read_any(path_or_state) if isinstance(path_or_state, (str, bytes, os.PathLike)) else path_or_state
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    bb = detectron2_backbone_state(sd)
    own = cnn.feature.state_dict()
    ok = {k: v.reshape(own[k].shape) if v.numel() == own[k].numel() and v.dim() != own[k].dim() else v
          for k, v in bb.items() if k in own and v.numel() == own[k].numel()}
    cnn.feature.load_state_dict(ok, strict=False)
    return len(ok)


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def _cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


class ModelSaver:
    """src/utils/load_save.py:43-68: ``{prefix}_step_{step}.pt`` (+ ``_train_state.pt`` with the optimizer)."""
    def __init__(self, output_dir: str, rank: Optional[int] = None):
        self.output_dir = output_dir
        self.rank = _rank() if rank is None else rank          # every rank may hold one: only rank 0 writes
        if self.rank == 0:
            os.makedirs(output_dir, exist_ok=True)

    def save(self, step: int, model, optimizer=None, prefix: str = "model") -> Optional[str]:
        model_path = os.path.join(self.output_dir, f"{prefix}_step_{step}.pt")
        _assert_whole(model, "ModelSaver.save()")
        if self.rank != 0:
            return None
        # contiguous OIHW copies: the file must not depend on our channels_last memory image
        sd = {k: (v.detach().cpu().contiguous() if torch.is_tensor(v) else v) for k, v in model.state_dict().items()}
        torch.save(sd, model_path)
        if optimizer is not None:
            torch.save({"step": step, "optimizer": _cpu(optimizer.state_dict())},
                       os.path.join(self.output_dir, f"{prefix}_step_{step}_train_state.pt"))
        return model_path


def _assert_whole(model, what):
    rt = getattr(model, "rt", None)
    bank = getattr(rt, "bank", None)
    if bank is not None:
        bank.assert_whole(what)


class E2E_TrainingRestorer:
    """src/utils/load_save.py:245-312: ``restore.pt`` (+ ``restore_backup.pt``) with global step, model and optimizer;
    resumes if one exists.  ``opts`` needs output_dir, num_train_steps, save_steps_ratio."""
    def __init__(self, opts, model, optimizer, rank: Optional[int] = None):
        get = (lambda k, d=None: opts.get(k, d)) if isinstance(opts, dict) else (lambda k, d=None: getattr(opts, k, d))
        out = get("output_dir")
        self.rank = _rank() if rank is None else rank          # restore() runs on every rank, save() on rank 0 only
        if self.rank == 0:
            os.makedirs(out, exist_ok=True)
        self.save_path = os.path.join(out, "restore.pt")
        self.backup_path = os.path.join(out, "restore_backup.pt")
        self.model, self.optimizer = model, optimizer
        self.save_steps = max(1, int(get("save_steps_ratio", 0.01) * get("num_train_steps")))
        self.global_step = 0
        if os.path.exists(self.save_path) or os.path.exists(self.backup_path):
            self.restore()

    def step(self):
        self.global_step += 1
        if self.global_step % self.save_steps == 0:
            self.save()

    def save(self):
        _assert_whole(self.model, "E2E_TrainingRestorer.save()")
        if self.rank != 0:
            return
        ckpt = {"global_step": self.global_step, "model_state_dict": _cpu(self.model.state_dict()),
                "optim_state_dict": _cpu(self.optimizer.state_dict())}
        rt = getattr(self.model, "rt", None)
        if rt is not None:                      # dropout masks are a function of these two counters: a resumed run continues the sequence
            ckpt["dropout_state"] = {"forward_count": int(rt.forward_count),
                                     "seed_dev": int(rt.seed_dev.item()) if rt.seed_dev is not None else 0}
        if os.path.exists(self.save_path):
            os.replace(self.save_path, self.backup_path)
        torch.save(ckpt, self.save_path)

    def restore(self):
        try:
            ckpt = torch.load(self.save_path, map_location="cpu")
        except Exception:
            ckpt = torch.load(self.backup_path, map_location="cpu")
        self.global_step = ckpt["global_step"]
        self.model.load_state_dict(ckpt["model_state_dict"])
        self.optimizer.load_state_dict(ckpt["optim_state_dict"])
        rt, ds = getattr(self.model, "rt", None), ckpt.get("dropout_state")
        if rt is not None and ds is not None:
            rt.forward_count = int(ds["forward_count"])
            if rt.seed_dev is not None:
                rt.seed_dev.fill_(int(ds["seed_dev"]))
