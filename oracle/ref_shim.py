"""Import the reference's OWN src/modeling/{transformers,modeling}.py, verbatim, in this container.

TEST INFRASTRUCTURE ONLY (see oracle/clipbert_oracle.py header).  Works only where
/root/reference exists (this build container, not the GPU box).  Nothing is copied: the reference
files are imported from where they lie; this module only provides the missing third-party
symbols they import:

* ``apex.normalization.fused_layer_norm.FusedLayerNorm`` -> ``torch.nn.LayerNorm`` (same math:
  LN over the last dim, affine, eps; apex only fuses it)              [transformers.py:32, modeling.py:12]
* ``transformers.configuration_bert.BertConfig``, ``transformers.activations.{gelu,gelu_new,swish}``,
  ``transformers.file_utils.add_start_docstrings*``, ``transformers.modeling_utils.{PreTrainedModel,
  prune_linear_layer}`` as they existed in transformers==2.11.0 (docker/requirements.txt:9)
                                                                         [transformers.py:27-31]
The three HF-2.11 ``PreTrainedModel`` behaviours the reference relies on are restated on a minimal
base class: ``get_extended_attention_mask`` = (1 - mask[:,None,None,:]) * -10000,
``get_head_mask(None, n)`` = [None]*n, ``init_weights()`` = apply(_init_weights) + tie the output
embedding to the input embedding.  These are [3P] semantics with no reference test pinning them.
"""
import importlib
import os
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("CLIPBERT_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "modeling", "modeling.py"))


class _Config:
    """Attribute bag standing in for transformers==2.11 BertConfig."""
    def __init__(self, **kw):
        self.output_attentions = False
        self.output_hidden_states = False
        self.is_decoder = False
        self.pruned_heads = {}
        for k, v in kw.items():
            setattr(self, k, v)


class _PreTrainedModel(nn.Module):
    config_class = _Config
    base_model_prefix = ""

    def __init__(self, config, *a, **kw):
        super().__init__()
        self.config = config

    @property
    def base_model(self):
        return getattr(self, self.base_model_prefix, self)

    def get_input_embeddings(self):
        base = getattr(self, self.base_model_prefix, self)
        if base is not self:
            return base.get_input_embeddings()
        raise NotImplementedError

    def get_output_embeddings(self):
        return None

    def tie_weights(self):
        out = self.get_output_embeddings()
        if out is not None:
            out.weight = self.get_input_embeddings().weight

    def init_weights(self):
        self.apply(self._init_weights)
        self.tie_weights()

    def get_extended_attention_mask(self, attention_mask, input_shape, device):
        ext = attention_mask[:, None, None, :].to(dtype=next(self.parameters()).dtype)
        return (1.0 - ext) * -10000.0

    def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
        assert head_mask is None
        return [None] * num_hidden_layers


def _install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "apex" not in sys.modules:
        apex = mod("apex")
        norm = mod("apex.normalization")
        fln = mod("apex.normalization.fused_layer_norm")
        fln.FusedLayerNorm = nn.LayerNorm
        apex.normalization = norm
        norm.fused_layer_norm = fln

    import transformers  # the installed (5.x) package; only sub-module paths are patched

    def sub(name, **attrs):
        full = "transformers." + name
        m = sys.modules.get(full)
        if m is None:
            try:
                m = importlib.import_module(full)
            except Exception:
                m = mod(full)
        for k, v in attrs.items():
            if not hasattr(m, k):
                setattr(m, k, v)
        sys.modules[full] = m
        return m

    def _swish(x):
        return x * torch.sigmoid(x)

    def _gelu_new(x):
        return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))

    def _passthrough_decorator(*a, **kw):
        def deco(fn):
            return fn
        return deco

    acts = sub("activations")
    acts.gelu = torch.nn.functional.gelu            # HF-2.11 ``gelu`` = F.gelu (exact erf) [3P]
    acts.gelu_new = _gelu_new
    acts.swish = _swish
    cb = mod("transformers.configuration_bert")
    cb.BertConfig = _Config
    fu = sub("file_utils")
    fu.add_start_docstrings = _passthrough_decorator
    fu.add_start_docstrings_to_callable = _passthrough_decorator
    mu = mod("transformers.modeling_utils_ref211")
    mu.PreTrainedModel = _PreTrainedModel
    mu.prune_linear_layer = lambda *a, **kw: (_ for _ in ()).throw(NotImplementedError())
    sys.modules["transformers.modeling_utils"] = mu


_CACHE = {}


def load_reference_modeling():
    """Returns (modeling_module, transformers_module) of the reference."""
    if "m" in _CACHE:
        return _CACHE["m"]
    if not available():
        raise RuntimeError(f"reference not present under {REFERENCE_ROOT}")
    saved_mu = sys.modules.get("transformers.modeling_utils")
    _install_stubs()
    # import as a private package so that `from .transformers import ...` resolves without
    # touching the reference's src/__init__ chain (which pulls horovod / detectron2).
    pkg_name = "_clipbert_ref_modeling"
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "src", "modeling")]
    sys.modules[pkg_name] = pkg
    sys.dont_write_bytecode = True
    try:
        tr = importlib.import_module(pkg_name + ".transformers")
        mo = importlib.import_module(pkg_name + ".modeling")
    finally:
        if saved_mu is not None:
            sys.modules["transformers.modeling_utils"] = saved_mu
        else:
            sys.modules.pop("transformers.modeling_utils", None)
    _CACHE["m"] = (mo, tr)
    return mo, tr


def make_config(cfg: dict):
    return _Config(**cfg)
