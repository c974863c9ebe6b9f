"""Model files -> runtime.  Keeps the reference's file-name table and search semantics
[ref: src/utils/model_registry.py:34-57, src/utils/constants.py] without the network download
(no egress; `download_weight` is out of scope).  Checkpoints are read with safetensors only."""

from __future__ import annotations

import os
from typing import Dict, Optional

import torch

DIT_MODELS = {
    "seedvr2_ema_3b_fp16.safetensors": "3b", "seedvr2_ema_3b_fp8_e4m3fn.safetensors": "3b",
    "seedvr2_ema_7b_fp16.safetensors": "7b", "seedvr2_ema_7b_sharp_fp16.safetensors": "7b",
}
DEFAULT_DIT = "seedvr2_ema_3b_fp16.safetensors"
DEFAULT_VAE = "ema_vae_fp16.safetensors"


def find_model_file(name: str, model_dir: Optional[str] = None) -> Optional[str]:
    for d in filter(None, (model_dir, os.environ.get("SEEDVR2_MODEL_DIR"), "models/SEEDVR2", ".")):
        p = os.path.join(d, name)
        if os.path.isfile(p):
            return p
    return None


def load_text_embedding(name: str = "pos_emb.pt", model_dir: Optional[str] = None) -> torch.Tensor:
    """The positive prompt embedding the reference ships next to its scripts and loads from its script directory
    [ref: src/core/generation_utils.py:528-533]: searched in `model_dir`, $SEEDVR2_MODEL_DIR, models/SEEDVR2, the current
    directory, then this package's directory and its parent (a ComfyUI custom-node checkout keeps it at the repository
    root).  Loaded with weights_only=True (a plain tensor; nothing from the file is executed)."""
    here = os.path.dirname(os.path.abspath(__file__))
    tried = []
    for d in filter(None, (model_dir, os.environ.get("SEEDVR2_MODEL_DIR"), "models/SEEDVR2", ".", here, os.path.dirname(here))):
        p = os.path.join(d, name)
        tried.append(p)
        if os.path.isfile(p):
            t = torch.load(p, map_location="cpu", weights_only=True)
            if isinstance(t, (list, tuple)):
                t = t[0]
            if not torch.is_tensor(t) or t.dim() not in (2, 3):
                raise ValueError(f"{p}: expected a [tokens, dim] tensor")
            return t.reshape(-1, t.shape[-1])
    raise FileNotFoundError(f"{name} not found (copy the reference's {name} into one of: {', '.join(tried)})")


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    if not path.endswith(".safetensors"):
        raise ValueError("only .safetensors checkpoints are loaded (no pickle execution)")
    from safetensors.torch import load_file
    sd = load_file(path)
    # fp8 (e4m3fn / e5m2) checkpoints: the reference keeps the fp8 storage and upcasts operands to the compute dtype around
    # every call [ref: src/optimization/compatibility.py:261-265, 628-650]; every fp8 value is exactly representable in
    # bf16, so converting once at load feeds the GEMMs the same weight values
    fp8 = tuple(t for t in (getattr(torch, "float8_e4m3fn", None), getattr(torch, "float8_e5m2", None)) if t is not None)
    return {k: (v.to(torch.bfloat16) if v.dtype in fp8 else v) for k, v in sd.items()}


def config_for(dit_name: str):
    """configs_3b vs configs_7b by file name [ref: src/core/model_configuration.py:715-717]."""
    return "7b" if "7b" in dit_name else "3b"


_DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def normalize_vae_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Checkpoint key layout -> the layout VideoVAEHip consumes (the reference's module tree):
      * old diffusers attention names (query/key/value/proj_attn) -> to_q/to_k/to_v/to_out.0, the
        conversion diffusers applies in `_convert_deprecated_attention_blocks`, which the reference calls
        before loading [ref: src/models/video_vae_v3/modules/attn_video_vae.py:1639-1649];
      * 1x1 conv-style attention projections [C,C,1,1] -> [C,C];
      * conv weights must already be 5-D: the shipped config uses inflation_mode 'pad', for which the
        reference's 2-D -> 3-D inflation asserts [ref: causal_inflation_lib.py:447,316-335]."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts and len(parts) >= 2 and parts[-2] in _DEPRECATED_ATTN:
            parts[-2] = _DEPRECATED_ATTN[parts[-2]]
            k = ".".join(parts)
        if "attentions" in parts and k.endswith(".weight") and v.dim() == 4 and v.shape[2:] == (1, 1):
            v = v[:, :, 0, 0]
        if k.endswith(".weight") and v.dim() == 4 and "attentions" not in parts:
            raise ValueError(f"{k}: 2-D conv weight in a causal 3-D VAE checkpoint (inflation is not defined for "
                             "inflation_mode 'pad')")
        out[k] = v
    return out
