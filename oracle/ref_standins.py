"""Stand-ins for third-party packages the reference imports but this image lacks (TEST INFRASTRUCTURE ONLY).

Used by ``oracle/pin_against_reference.py`` (build container only) so that the reference's own model
files -- ``src/models/video_vae_v3/modules/attn_video_vae.py``, ``src/models/dit_3b/nadit.py``,
``src/models/dit_7b/nadit.py`` and everything they import from the reference -- can be imported and
RUN AS THEY ARE.  Nothing here is shipped, imported by the product, or copied from the reference.

Two kinds of stand-in, kept apart on purpose:

* STRUCTURAL (no arithmetic): the ``diffusers`` base classes the reference's 3-D classes inherit from
  (``ResnetBlock2D``, ``Upsample2D``, ``Downsample2D``, ``DownEncoderBlock2D``, ``UpDecoderBlock2D``,
  ``AutoencoderKL``).  Their constructors only create the attributes / sub-modules the reference's
  subclasses read or replace (names = the state-dict keys of the shipped checkpoints, which the reference's
  ``load_state_dict`` relies on).  Every ``forward`` that runs is the reference's own.
* ARITHMETIC restated from the libraries' published algorithms and served from the oracle's functions
  (these stay "parity unpinned" -- the libraries are absent, nothing of theirs can run here):
  ``diffusers...Attention`` (the VAE mid-block attention), ``DiagonalGaussianDistribution``,
  ``diffusers.models.embeddings.get_timestep_embedding``, ``rotary_embedding_torch.RotaryEmbedding``
  (``freqs``, ``get_axial_freqs``) and ``apply_rotary_emb``.  What the pins gain even here: the reference's
  USE of these functions (window-local coordinates, text offset, slicing of the axial table, dtype casts,
  per-frame folding of the attention) is the reference's code, not a reading of it.
"""

from __future__ import annotations

import math
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

from seedvr2_ref import dit as _odit
from seedvr2_ref import rope as _orope


# ------------------------------------------------------------------------------------------ diffusers: structural
class _Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv",
                 kernel_size=None, padding=1, norm_type=None, eps=None, elementwise_affine=None, bias=True,
                 interpolate=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv, self.use_conv_transpose, self.name, self.interpolate = use_conv, use_conv_transpose, name, interpolate
        self.norm = None
        conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=padding, bias=bias) if use_conv else None
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv


class _Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", kernel_size=3,
                 norm_type=None, eps=None, elementwise_affine=None, bias=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv, self.padding, self.name = use_conv, padding, name
        self.norm = None
        conv = (nn.Conv2d(self.channels, self.out_channels, kernel_size, stride=2, padding=padding, bias=bias)
                if use_conv else nn.AvgPool2d(kernel_size=2, stride=2))
        if name == "conv":
            self.Conv2d_0 = conv
        self.conv = conv


class _ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None,
                 up=False, down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.pre_norm, self.in_channels, self.out_channels = True, in_channels, out_channels
        self.use_conv_shortcut, self.up, self.down = conv_shortcut, up, down
        self.output_scale_factor, self.time_embedding_norm, self.skip_time_act = output_scale_factor, time_embedding_norm, skip_time_act
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        assert temb_channels is None, "the VAE builds its resnets without a time embedding"
        self.time_emb_proj = None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        c2 = conv_2d_out_channels or out_channels
        self.conv2 = nn.Conv2d(out_channels, c2, 3, 1, 1)
        assert non_linearity in ("swish", "silu")
        self.nonlinearity = nn.SiLU()
        self.upsample = self.downsample = None
        self.use_in_shortcut = in_channels != c2 if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = nn.Conv2d(in_channels, c2, 1, bias=conv_shortcut_bias) if self.use_in_shortcut else None


class _Block2D(nn.Module):
    """DownEncoderBlock2D / UpDecoderBlock2D: the reference's 3-D subclasses rebuild every sub-module themselves."""

    def __init__(self, **kwargs):
        super().__init__()


class _AutoencoderKL(nn.Module):
    """diffusers.AutoencoderKL as used by the reference: use_slicing switch + device/dtype of the parameters."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.use_slicing = False
        self.config = types.SimpleNamespace(**kwargs)

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


# ------------------------------------------------------------------------------------------ diffusers: arithmetic (unpinned)
class _Attention(nn.Module):
    """diffusers Attention(heads=1, GroupNorm, residual, bias) with the default SDPA processor, on (b c h w).
    Arithmetic = the oracle's restatement (seedvr2_ref.vae.mid_attention); parameter names = diffusers' (to_q ...)."""

    def __init__(self, query_dim, heads=1, dim_head=64, rescale_output_factor=1.0, eps=1e-5, norm_num_groups=None,
                 spatial_norm_dim=None, residual_connection=False, bias=False, upcast_softmax=False,
                 _from_deprecated_attn_block=False, **kw):
        super().__init__()
        assert heads == 1 and dim_head == query_dim and residual_connection and spatial_norm_dim is None
        assert rescale_output_factor in (1, 1.0)
        self.groups, self.eps = norm_num_groups, eps
        self.group_norm = nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
        self.to_q, self.to_k, self.to_v = (nn.Linear(query_dim, query_dim, bias=bias) for _ in range(3))
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states, temb=None, **kw):
        from seedvr2_ref import vae as _ovae
        p = {"a." + k: v for k, v in self.state_dict().items()}
        cfg = _ovae.VAEConfig(norm_groups=self.groups, eps=self.eps)
        # (b f) c h w  ->  the oracle's [C, T, H, W]
        return _ovae.mid_attention(p, "a", hidden_states.permute(1, 0, 2, 3), cfg).permute(1, 0, 2, 3)


class _DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)

    def mode(self):
        return self.mean


def _get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1, scale=1,
                            max_period=10000):
    assert not flip_sin_to_cos and downscale_freq_shift == 0 and scale == 1 and max_period == 10000
    return torch.cat([_odit.timestep_embedding(float(t), embedding_dim) for t in timesteps.float().tolist()], dim=0)


# ------------------------------------------------------------------------------------------ rotary_embedding_torch (unpinned)
class _RotaryEmbedding(nn.Module):
    def __init__(self, dim, freqs_for="lang", theta=10000, max_freq=10, learned_freq=False, **kw):
        super().__init__()
        assert not learned_freq
        self.freqs_for = freqs_for
        fr = _orope.lang_freqs(dim, float(theta)) if freqs_for == "lang" else _orope.pixel_freqs(dim, float(max_freq))
        self.freqs = nn.Parameter(fr, requires_grad=False)

    def get_axial_freqs(self, *dims):
        n = len(dims)
        per_axis = []
        for ax, d in enumerate(dims):
            pos = torch.linspace(-1, 1, steps=d) if self.freqs_for == "pixel" else torch.arange(d)
            a = _orope.axis_angles(pos, self.freqs)                              # [d, 2*nfreq]
            shape = [1] * n + [a.shape[-1]]
            shape[ax] = d
            per_axis.append(a.reshape(shape).expand(*dims, a.shape[-1]))
        return torch.cat(per_axis, dim=-1)


def _apply_rotary_emb(freqs, t, start_index=0, scale=1.0, seq_dim=-2):
    assert start_index == 0 and scale == 1.0
    if t.ndim == 3:
        freqs = freqs[-t.shape[seq_dim]:]
    return _orope.rotate_interleaved_cs(t, freqs.cos(), freqs.sin()).type(t.dtype)


# ------------------------------------------------------------------------------------------ installation
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Put the stand-ins into sys.modules (idempotent).  Call before importing reference modules."""
    class _Name:  # placeholder type, never instantiated on the paths that run
        pass

    _mod("omegaconf", DictConfig=type("DictConfig", (dict,), {}), ListConfig=type("ListConfig", (list,), {}),
         OmegaConf=_Name)
    d = _mod("diffusers", AutoencoderKL=_AutoencoderKL)
    d.models = _mod("diffusers.models")
    _mod("diffusers.models.normalization", RMSNorm=type("RMSNorm", (nn.Module,), {}))
    _mod("diffusers.models.attention_processor", Attention=_Attention, SpatialNorm=_Name)
    _mod("diffusers.models.autoencoders")
    _mod("diffusers.models.autoencoders.vae", DecoderOutput=_Out, DiagonalGaussianDistribution=_DiagonalGaussianDistribution)
    _mod("diffusers.models.downsampling", Downsample2D=_Downsample2D)
    _mod("diffusers.models.lora", LoRACompatibleConv=type("LoRACompatibleConv", (nn.Conv2d,), {}))
    _mod("diffusers.models.modeling_outputs", AutoencoderKLOutput=_Out)
    _mod("diffusers.models.resnet", ResnetBlock2D=_ResnetBlock2D)
    _mod("diffusers.models.unets")
    _mod("diffusers.models.unets.unet_2d_blocks", DownEncoderBlock2D=type("DownEncoderBlock2D", (_Block2D,), {}),
         UpDecoderBlock2D=type("UpDecoderBlock2D", (_Block2D,), {}))
    _mod("diffusers.models.upsampling", Upsample2D=_Upsample2D)
    _mod("diffusers.models.embeddings", get_timestep_embedding=_get_timestep_embedding)
    _mod("diffusers.utils", is_torch_version=lambda op, v: True)
    _mod("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)
    _mod("rotary_embedding_torch", RotaryEmbedding=_RotaryEmbedding, apply_rotary_emb=_apply_rotary_emb)
    # color_fix.py imports two converter NAMES from torchvision (used only by its PIL entry points)
    _mod("torchvision")
    _mod("torchvision.transforms", ToTensor=_Name, ToPILImage=_Name)
