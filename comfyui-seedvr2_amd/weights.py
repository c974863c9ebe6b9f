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


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    if not path.endswith(".safetensors"):
        raise ValueError("only .safetensors checkpoints are loaded (no pickle execution)")
    from safetensors.torch import load_file
    return load_file(path)


def config_for(dit_name: str):
    """configs_3b vs configs_7b by file name [ref: src/core/model_configuration.py:715-717]."""
    return "7b" if "7b" in dit_name else "3b"
