"""The four ComfyUI V3 nodes of the reference, same ids / inputs / outputs, backed by the HIP pipeline
[ref: src/interfaces/video_upscaler.py:37-571, dit_model_loader.py, vae_model_loader.py,
 torch_compile_settings.py, interfaces/__init__.py:14-29].  ComfyUI's ``comfy_api`` is not installed in
the build image, so the classes also work standalone (``execute`` is a plain classmethod)."""

from __future__ import annotations

from typing import Any, Dict

import torch

from . import dit as _dit
from . import hip
from . import pipeline as _pipeline
from . import vae as _vae
from . import weights as _weights

try:  # pragma: no cover - only inside ComfyUI
    from comfy_api.latest import ComfyExtension, io  # type: ignore
    _Base = io.ComfyNode
except Exception:  # standalone
    io = None
    ComfyExtension = object
    _Base = object

UPSCALER_INPUTS = ["image", "dit", "vae", "seed", "resolution", "max_resolution", "batch_size", "uniform_batch_size",
                   "temporal_overlap", "prepend_frames", "color_correction", "input_noise_scale",
                   "latent_noise_scale", "offload_device", "enable_debug"]
DIT_KEYS = ["model", "device", "offload_device", "cache_model", "blocks_to_swap", "swap_io_components",
            "attention_mode", "torch_compile_args", "node_id"]
VAE_KEYS = ["model", "device", "offload_device", "cache_model", "encode_tiled", "encode_tile_size",
            "encode_tile_overlap", "decode_tiled", "decode_tile_size", "decode_tile_overlap", "tile_debug",
            "torch_compile_args", "node_id"]

_PIPELINES: Dict[Any, _pipeline.SeedVR2Pipeline] = {}


def _check_interrupt(done: int = 0, total: int = 0) -> None:
    """Polled once per batch: ComfyUI's cancel button raises from here, like the reference's check_interrupt
    [ref: src/core/generation_utils.py:360-366, 655-658].  A no-op outside ComfyUI."""
    try:
        import comfy.model_management as mm  # type: ignore
    except Exception:
        return
    mm.throw_exception_if_processing_interrupted()


def _build_pipeline(dit_cfg: dict, vae_cfg: dict, text_pos: torch.Tensor) -> _pipeline.SeedVR2Pipeline:
    key = (dit_cfg["model"], vae_cfg["model"], dit_cfg["device"])
    if key not in _PIPELINES:
        dev = torch.device(dit_cfg["device"])
        dpath = _weights.find_model_file(dit_cfg["model"])
        vpath = _weights.find_model_file(vae_cfg["model"])
        if dpath is None or vpath is None:
            raise RuntimeError(f"model files not found: {dit_cfg['model']} / {vae_cfg['model']} "
                               "(no network download on this path; place them under models/SEEDVR2)")
        cfg = _dit.CFG_7B if _weights.config_for(dit_cfg["model"]) == "7b" else _dit.CFG_3B
        dit = _dit.NaDiTHip(cfg, _weights.load_state_dict(dpath), dev)
        vae = _vae.VideoVAEHip(_vae.VAEConfig(), _weights.normalize_vae_state_dict(_weights.load_state_dict(vpath)), dev)
        _PIPELINES[key] = _pipeline.SeedVR2Pipeline(vae, dit, text_pos, dev)
    return _PIPELINES[key]


class SeedVR2LoadDiTModel(_Base):
    @classmethod
    def execute(cls, model: str = _weights.DEFAULT_DIT, device: str = "cuda:0", offload_device: str = "none",
                cache_model: bool = False, blocks_to_swap: int = 0, swap_io_components: bool = False,
                attention_mode: str = "sdpa", torch_compile_args=None, node_id=None):
        if blocks_to_swap:
            raise ValueError("BlockSwap is not needed on 288 GB HBM; set blocks_to_swap=0")
        cfg = dict(zip(DIT_KEYS, (model, device, offload_device, cache_model, 0, False, attention_mode,
                                  torch_compile_args, node_id)))
        return (cfg,)


class SeedVR2LoadVAEModel(_Base):
    @classmethod
    def execute(cls, model: str = _weights.DEFAULT_VAE, device: str = "cuda:0", offload_device: str = "none",
                cache_model: bool = False, encode_tiled: bool = False, encode_tile_size: int = 1024,
                encode_tile_overlap: int = 128, decode_tiled: bool = False, decode_tile_size: int = 1024,
                decode_tile_overlap: int = 128, tile_debug: str = "false", torch_compile_args=None, node_id=None):
        for what, size, ov in (("encode", encode_tile_size, encode_tile_overlap), ("decode", decode_tile_size, decode_tile_overlap)):
            if size < 8 or ov < 0 or ov >= size:
                raise ValueError(f"{what}_tile_size must be >= 8 and {what}_tile_overlap in [0, tile_size)")
        cfg = dict(zip(VAE_KEYS, (model, device, offload_device, cache_model, bool(encode_tiled), encode_tile_size,
                                  encode_tile_overlap, bool(decode_tiled), decode_tile_size, decode_tile_overlap, tile_debug,
                                  torch_compile_args, node_id)))
        return (cfg,)


class SeedVR2TorchCompileSettings(_Base):
    @classmethod
    def execute(cls, **kwargs):
        return (dict(kwargs),)  # accepted for graph compatibility; there is no tracing compiler on this path


class SeedVR2VideoUpscaler(_Base):
    @classmethod
    def execute(cls, image: torch.Tensor, dit: dict, vae: dict, seed: int = 42, resolution: int = 1080,
                max_resolution: int = 0, batch_size: int = 5, uniform_batch_size: bool = False,
                temporal_overlap: int = 0, prepend_frames: int = 0, color_correction: str = "lab",
                input_noise_scale: float = 0.0, latent_noise_scale: float = 0.0, offload_device: str = "cpu",
                enable_debug: bool = False, text_pos: torch.Tensor | None = None, pipeline=None):
        """image [N,H,W,C] float in [0,1], C in {3,4} -> float32 CPU [N',H',W',C]
        [ref: src/interfaces/video_upscaler.py:225-571]."""
        if image.dim() != 4 or image.shape[-1] not in (3, 4):
            raise ValueError("image must be [N,H,W,3|4]")
        if color_correction not in ("none",) + tuple(hip.COLOR_METHODS):
            raise NotImplementedError(f"color_correction '{color_correction}' is not implemented "
                                      f"(available: none, {', '.join(hip.COLOR_METHODS)})")
        if not (0.0 <= input_noise_scale <= 1.0 and 0.0 <= latent_noise_scale <= 1.0):
            raise ValueError("input_noise_scale and latent_noise_scale must lie in [0, 1]")
        if batch_size % 4 != 1:
            raise ValueError("batch_size must follow 4n+1")
        out = cls.run_device(image, dit, vae, seed, resolution, max_resolution, batch_size, uniform_batch_size,
                             temporal_overlap, prepend_frames, color_correction, text_pos=text_pos, pipeline=pipeline,
                             input_noise_scale=input_noise_scale, latent_noise_scale=latent_noise_scale)
        return (out.float().cpu(),)

    @staticmethod
    def prepend_reversed(x: torch.Tensor, count: int) -> torch.Tensor:
        """pad_video_temporal(prepend=True) [ref: src/core/generation_utils.py:593-652]: `count` reversed head frames
        (frame 0 excluded); when count >= T: the last frame repeated count-T+1 times, then frames T-1..1, then the clip."""
        t = x.shape[0]
        if count <= 0:
            return x
        if count >= t:
            rep = x[-1:].repeat(count - t + 1, 1, 1, 1)
            rev = x[1:].flip(0) if t > 1 else x[:0]
            return torch.cat((rep, rev, x), dim=0)
        return torch.cat((x[1:count + 1].flip(0), x), dim=0)

    @classmethod
    def run_device(cls, image: torch.Tensor, dit: dict, vae: dict, seed: int = 42, resolution: int = 1080,
                   max_resolution: int = 0, batch_size: int = 5, uniform_batch_size: bool = False,
                   temporal_overlap: int = 0, prepend_frames: int = 0, color_correction: str = "lab",
                   text_pos: torch.Tensor | None = None, pipeline=None, input_noise_scale: float = 0.0,
                   latent_noise_scale: float = 0.0) -> torch.Tensor:
        """The upscaler without the host round trip: image [N,H,W,3] in [0,1] (any device / float dtype; bf16 frames that
        are already on the device are used as they are) -> bf16 [N,H',W',3] in [0,1] on the device."""
        pipe = pipeline or _build_pipeline(dit, vae, text_pos)
        # tiling is a property of the VAE node's settings, not of the cached pipeline
        pipe.set_tiling(**{k: vae[k] for k in ("encode_tiled", "encode_tile_size", "encode_tile_overlap", "decode_tiled",
                                               "decode_tile_size", "decode_tile_overlap") if k in (vae or {})})
        frames = cls.prepend_reversed(image, prepend_frames)
        out = pipe.generate(frames, resolution, batch_size, temporal_overlap, seed, max_resolution, uniform_batch_size,
                            progress=_check_interrupt, color_correction=color_correction,
                            input_noise_scale=input_noise_scale, latent_noise_scale=latent_noise_scale)
        # postprocess_all_batches removes them again [ref: src/core/generation_phases.py:1377-1386]
        if prepend_frames > 0 and prepend_frames < out.shape[0]:
            out = out[prepend_frames:]
        return out


class SeedVR2Extension(ComfyExtension):  # pragma: no cover
    async def get_node_list(self):
        return [SeedVR2LoadDiTModel, SeedVR2LoadVAEModel, SeedVR2TorchCompileSettings, SeedVR2VideoUpscaler]


async def comfy_entrypoint():  # pragma: no cover
    return SeedVR2Extension()
