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


# ComfyUI V3 schemas: same node ids, input names, kinds, defaults, ranges and option lists as the reference
# [ref: src/interfaces/video_upscaler.py:48-222, dit_model_loader.py:30-130, vae_model_loader.py:31-165,
#  torch_compile_settings.py:13-97]; (name, io kind, keyword arguments)
_DEVICES = ["cuda:0", "cuda:1", "cuda:2", "cuda:3", "cuda:4", "cuda:5", "cuda:6", "cuda:7"]
_OFFLOAD = ["none", "cpu"] + _DEVICES
_SCHEMAS = {
    "SeedVR2VideoUpscaler": dict(
        display_name="SeedVR2 Video Upscaler (MI355X)",
        description="One-step SeedVR2 upscaling of an image batch (RGB or RGBA) on the MI355X-native path.",
        inputs=[
            ("image", "Image", dict(tooltip="Frames [N,H,W,3|4] in [0,1]; the output keeps the channel count.")),
            ("dit", "Custom:SEEDVR2_DIT", dict(tooltip="From SeedVR2 (Down)Load DiT Model.")),
            ("vae", "Custom:SEEDVR2_VAE", dict(tooltip="From SeedVR2 (Down)Load VAE Model.")),
            ("seed", "Int", dict(default=42, min=0, max=2 ** 32 - 1, step=1, tooltip="Noise seed; same seed, same output.")),
            ("resolution", "Int", dict(default=1080, min=16, max=16384, step=2, tooltip="Target length of the shortest edge.")),
            ("max_resolution", "Int", dict(default=0, min=0, max=16384, step=2, tooltip="Cap for the longest edge (0 = none).")),
            ("batch_size", "Int", dict(default=5, min=1, max=16384, step=4, tooltip="Frames per batch, 4n+1.")),
            ("uniform_batch_size", "Boolean", dict(default=False, tooltip="Pad the last batch to batch_size.")),
            ("temporal_overlap", "Int", dict(default=0, min=0, max=16, step=1, tooltip="Frames cross-faded between batches.")),
            ("prepend_frames", "Int", dict(default=0, min=0, max=32, step=1, tooltip="Reversed head frames as warm-up context.")),
            ("color_correction", "Combo", dict(options=["lab", "wavelet", "wavelet_adaptive", "hsv", "adain", "none"], default="lab",
                                              tooltip="Phase-4 colour matching against the input.")),
            ("input_noise_scale", "Float", dict(default=0.0, min=0.0, max=1.0, step=0.001, tooltip="Noise blended into the model input.")),
            ("latent_noise_scale", "Float", dict(default=0.0, min=0.0, max=1.0, step=0.001, tooltip="Noise blended into the condition latent.")),
            ("offload_device", "Combo", dict(options=_OFFLOAD, default="cpu", tooltip="Accepted for compatibility; nothing is offloaded.")),
            ("enable_debug", "Boolean", dict(default=False, tooltip="Accepted for compatibility.")),
        ],
        outputs=[("Image", dict(tooltip="Upscaled frames, float32 [N,H',W',C] in [0,1]."))]),
    "SeedVR2LoadDiTModel": dict(
        display_name="SeedVR2 (Down)Load DiT Model",
        description="Selects the NaDiT checkpoint (3B / 7B .safetensors under models/SEEDVR2) and its device.",
        inputs=[
            ("model", "Combo", dict(options=list(_weights.DIT_MODELS), default=_weights.DEFAULT_DIT, tooltip="Checkpoint file name.")),
            ("device", "Combo", dict(options=_DEVICES, default=_DEVICES[0], tooltip="GPU that runs the model.")),
            ("blocks_to_swap", "Int", dict(default=0, min=0, max=36, step=1, optional=True, tooltip="Must stay 0: 288 GB of HBM need no BlockSwap.")),
            ("swap_io_components", "Boolean", dict(default=False, optional=True, tooltip="Must stay off.")),
            ("offload_device", "Combo", dict(options=_OFFLOAD, default="none", optional=True, tooltip="Accepted; nothing is offloaded.")),
            ("cache_model", "Boolean", dict(default=False, optional=True, tooltip="Models stay resident between runs either way.")),
            ("attention_mode", "Combo", dict(options=["sdpa", "flash_attn"], default="sdpa", optional=True,
                                            tooltip="Accepted; the window attention always runs the HIP flash kernel.")),
            ("torch_compile_args", "Custom:TORCH_COMPILE_ARGS", dict(optional=True, tooltip="Accepted; there is no tracing compiler here.")),
        ],
        outputs=[("Custom:SEEDVR2_DIT", dict(tooltip="DiT configuration."))]),
    "SeedVR2LoadVAEModel": dict(
        display_name="SeedVR2 (Down)Load VAE Model",
        description="Selects the VAE checkpoint, its device and the optional spatial tiling.",
        inputs=[
            ("model", "Combo", dict(options=[_weights.DEFAULT_VAE], default=_weights.DEFAULT_VAE, tooltip="Checkpoint file name.")),
            ("device", "Combo", dict(options=_DEVICES, default=_DEVICES[0], tooltip="GPU that runs the model.")),
            ("encode_tiled", "Boolean", dict(default=False, optional=True, tooltip="Encode in blended spatial tiles.")),
            ("encode_tile_size", "Int", dict(default=1024, min=64, step=32, optional=True, tooltip="Tile edge in pixels.")),
            ("encode_tile_overlap", "Int", dict(default=128, min=0, step=32, optional=True, tooltip="Blend width in pixels.")),
            ("decode_tiled", "Boolean", dict(default=False, optional=True, tooltip="Decode in blended spatial tiles.")),
            ("decode_tile_size", "Int", dict(default=1024, min=64, step=32, optional=True, tooltip="Tile edge in output pixels.")),
            ("decode_tile_overlap", "Int", dict(default=128, min=0, step=32, optional=True, tooltip="Blend width in output pixels.")),
            ("tile_debug", "Combo", dict(options=["false", "encode", "decode"], default="false", optional=True, tooltip="Must stay 'false'.")),
            ("offload_device", "Combo", dict(options=_OFFLOAD, default="none", optional=True, tooltip="Accepted; nothing is offloaded.")),
            ("cache_model", "Boolean", dict(default=False, optional=True, tooltip="Models stay resident between runs either way.")),
            ("torch_compile_args", "Custom:TORCH_COMPILE_ARGS", dict(optional=True, tooltip="Accepted; there is no tracing compiler here.")),
        ],
        outputs=[("Custom:SEEDVR2_VAE", dict(tooltip="VAE configuration."))]),
    "SeedVR2TorchCompileSettings": dict(
        display_name="SeedVR2 Torch Compile Settings",
        description="Kept so that existing graphs load; the settings are carried along and ignored.",
        inputs=[
            ("backend", "Combo", dict(options=["inductor", "cudagraphs"], default="inductor", tooltip="Ignored.")),
            ("mode", "Combo", dict(options=["default", "reduce-overhead", "max-autotune", "max-autotune-no-cudagraphs"], default="default",
                                  tooltip="Ignored.")),
            ("fullgraph", "Boolean", dict(default=False, tooltip="Ignored.")),
            ("dynamic", "Boolean", dict(default=False, tooltip="Ignored.")),
            ("dynamo_cache_size_limit", "Int", dict(default=64, min=0, max=1024, step=1, tooltip="Ignored.")),
            ("dynamo_recompile_limit", "Int", dict(default=128, min=0, max=1024, step=1, tooltip="Ignored.")),
        ],
        outputs=[("Custom:TORCH_COMPILE_ARGS", dict(tooltip="Settings dictionary."))]),
}


def _io_kind(kind: str):
    return io.Custom(kind.split(":", 1)[1]) if kind.startswith("Custom:") else getattr(io, kind)


def _schema(node_id: str):
    spec = _SCHEMAS[node_id]
    return io.Schema(node_id=node_id, display_name=spec["display_name"], category="SEEDVR2", description=spec["description"],
                     inputs=[_io_kind(k).Input(n, **kw) for n, k, kw in spec["inputs"]],
                     outputs=[_io_kind(k).Output(**kw) for k, kw in spec["outputs"]])


def _out(value):
    """Node result: io.NodeOutput inside ComfyUI [ref: dit_model_loader.py:183], a 1-tuple standalone."""
    return io.NodeOutput(value) if io is not None else (value,)


def _build_pipeline(dit_cfg: dict, vae_cfg: dict, text_pos: torch.Tensor | None) -> _pipeline.SeedVR2Pipeline:
    key = (dit_cfg["model"], vae_cfg["model"], dit_cfg["device"])
    if key not in _PIPELINES:
        if text_pos is None:   # inside ComfyUI execute() gets the schema inputs only [ref: generation_utils.py:528-533]
            text_pos = _weights.load_text_embedding("pos_emb.pt")
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
    def define_schema(cls):
        return _schema("SeedVR2LoadDiTModel")

    @classmethod
    def execute(cls, model: str = _weights.DEFAULT_DIT, device: str = "cuda:0", offload_device: str = "none",
                cache_model: bool = False, blocks_to_swap: int = 0, swap_io_components: bool = False,
                attention_mode: str = "sdpa", torch_compile_args=None, node_id=None):
        if blocks_to_swap:
            raise ValueError("BlockSwap is not needed on 288 GB HBM; set blocks_to_swap=0")
        cfg = dict(zip(DIT_KEYS, (model, device, offload_device, cache_model, 0, False, attention_mode,
                                  torch_compile_args, node_id)))
        return _out(cfg)


class SeedVR2LoadVAEModel(_Base):
    @classmethod
    def define_schema(cls):
        return _schema("SeedVR2LoadVAEModel")

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
        return _out(cfg)


class SeedVR2TorchCompileSettings(_Base):
    @classmethod
    def define_schema(cls):
        return _schema("SeedVR2TorchCompileSettings")

    @classmethod
    def execute(cls, **kwargs):
        return _out(dict(kwargs))  # accepted for graph compatibility; there is no tracing compiler on this path


class SeedVR2VideoUpscaler(_Base):
    @classmethod
    def define_schema(cls):
        return _schema("SeedVR2VideoUpscaler")

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
        if batch_size < 1:
            raise ValueError("batch_size must be positive")   # any size: batches are padded to 4n+1 like the reference
        out = cls.run_device(image, dit, vae, seed, resolution, max_resolution, batch_size, uniform_batch_size,
                             temporal_overlap, prepend_frames, color_correction, text_pos=text_pos, pipeline=pipeline,
                             input_noise_scale=input_noise_scale, latent_noise_scale=latent_noise_scale)
        return _out(out.float().cpu())

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
                   latent_noise_scale: float = 0.0, keep_prepended: bool = False) -> torch.Tensor:
        """The upscaler without the host round trip: image [N,H,W,3] in [0,1] (any device / float dtype; bf16 frames that
        are already on the device are used as they are) -> bf16 [N,H',W',3] in [0,1] on the device.
        keep_prepended: leave the outputs of the `prepend_frames` warm-up frames in the result -- what the reference's
        single-GPU CLI does (it passes prepend_frames=0 to postprocess_all_batches, inference_cli.py:920, and only its
        multi-GPU parent strips them, :1200-1204); the node path strips them (generation_phases.py:1377-1386)."""
        pipe = pipeline or _build_pipeline(dit, vae, text_pos)
        # tiling is a property of the VAE node's settings, not of the cached pipeline
        pipe.set_tiling(**{k: vae[k] for k in ("encode_tiled", "encode_tile_size", "encode_tile_overlap", "decode_tiled",
                                               "decode_tile_size", "decode_tile_overlap") if k in (vae or {})})
        frames = cls.prepend_reversed(image, prepend_frames)
        out = pipe.generate(frames, resolution, batch_size, temporal_overlap, seed, max_resolution, uniform_batch_size,
                            progress=_check_interrupt, color_correction=color_correction,
                            input_noise_scale=input_noise_scale, latent_noise_scale=latent_noise_scale)
        # postprocess_all_batches removes them again [ref: src/core/generation_phases.py:1377-1386]
        if prepend_frames > 0 and prepend_frames < out.shape[0] and not keep_prepended:
            out = out[prepend_frames:]
        return out


class SeedVR2Extension(ComfyExtension):  # pragma: no cover
    async def get_node_list(self):
        return [SeedVR2LoadDiTModel, SeedVR2LoadVAEModel, SeedVR2TorchCompileSettings, SeedVR2VideoUpscaler]


async def comfy_entrypoint():  # pragma: no cover
    return SeedVR2Extension()
