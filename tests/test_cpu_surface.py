"""Drop-in surface: CLI flags and node inputs are the reference's (names extracted from the reference into
tests/golden/surface.json); unsupported low-VRAM options are rejected loudly, never silently ignored."""

import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def surface():
    with open(os.path.join(ROOT, "tests", "golden", "surface.json")) as fh:
        return json.load(fh)


def _cli(pkg):
    import importlib.util
    spec = importlib.util.spec_from_file_location("svr2_cli", os.path.join(ROOT, "comfyui-seedvr2_amd", "inference_cli.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cli_flags_match_reference(pkg, surface):
    cli = _cli(pkg)
    a = cli.parse_arguments(["in.png"])
    have = {("input" if k == "input" else "--" + k) for k in vars(a)}
    assert have == set(surface["cli_flags"])
    assert (a.resolution, a.batch_size, a.seed, a.temporal_overlap, a.color_correction) == (1080, 5, 42, 0, "lab")
    with pytest.raises(SystemExit):
        cli.parse_arguments(["in.png", "--blocks_to_swap", "4"])
    assert cli.parse_arguments(["in.png", "--batch_size", "4"]).batch_size == 4   # padded to 4n+1 per batch, like the reference


def test_node_inputs_and_loader_dicts(pkg, surface):
    N = pkg.nodes
    assert N.UPSCALER_INPUTS == surface["upscaler_inputs"]
    import inspect
    params = list(inspect.signature(N.SeedVR2VideoUpscaler.execute).parameters)[: len(N.UPSCALER_INPUTS)]
    assert params == N.UPSCALER_INPUTS
    dit_cfg, = N.SeedVR2LoadDiTModel.execute()
    vae_cfg, = N.SeedVR2LoadVAEModel.execute()
    assert list(dit_cfg) == N.DIT_KEYS and list(vae_cfg) == N.VAE_KEYS
    tiled, = N.SeedVR2LoadVAEModel.execute(decode_tiled=True, decode_tile_size=768, decode_tile_overlap=64)
    assert (tiled["decode_tiled"], tiled["decode_tile_size"], tiled["decode_tile_overlap"], tiled["encode_tiled"]) == (True, 768, 64, False)
    with pytest.raises(ValueError):
        N.SeedVR2LoadVAEModel.execute(encode_tiled=True, encode_tile_size=64, encode_tile_overlap=64)
    with pytest.raises(NotImplementedError):
        N.SeedVR2VideoUpscaler.execute(torch.zeros(1, 8, 8, 3), dit_cfg, vae_cfg, color_correction="bogus", pipeline=object())


def test_runner_condition_matches_reference_semantics(pkg):
    r = pkg.infer.VideoDiffusionInfer()
    lat = torch.randn(2, 3, 4, 16)
    cond = r.get_condition(torch.zeros_like(lat), lat, "sr")
    assert cond.shape == (2, 3, 4, 17) and torch.equal(cond[..., :16], lat) and (cond[..., 16] == 1).all()
    t = pkg.infer.VideoDiffusionInfer(encode_tiled=True, encode_tile_size=(256, 256), encode_tile_overlap=(32, 32))
    assert t.encode_tiled and not t.decode_tiled and t.decode_tile_size == (512, 512)
    with pytest.raises(NotImplementedError):
        pkg.infer.VideoDiffusionInfer(tile_debug="encode")


def test_vae_checkpoint_key_normalisation(pkg):
    """Deprecated diffusers attention names are converted like the reference does before loading; the
    synthetic state dict (reference key layout) round-trips; 2-D conv weights are rejected."""
    W = pkg.weights
    sd = {"decoder.mid_block.attentions.0.query.weight": torch.zeros(8, 8, 1, 1),
          "decoder.mid_block.attentions.0.proj_attn.bias": torch.zeros(8),
          "decoder.conv_in.weight": torch.zeros(8, 4, 3, 3, 3)}
    out = W.normalize_vae_state_dict(sd)
    assert set(out) == {"decoder.mid_block.attentions.0.to_q.weight", "decoder.mid_block.attentions.0.to_out.0.bias",
                        "decoder.conv_in.weight"}
    assert out["decoder.mid_block.attentions.0.to_q.weight"].shape == (8, 8)
    with pytest.raises(ValueError):
        W.normalize_vae_state_dict({"encoder.conv_in.weight": torch.zeros(8, 3, 3, 3)})
    from seedvr2_ref import vae as ovae
    ch = (32, 32, 64, 64)
    keys = set(ovae.param_shapes(ovae.VAEConfig(block_out_channels=ch)))
    syn = pkg.vae.synthetic_state_dict(pkg.vae.VAEConfig(block_out_channels=ch), "cpu", dtype=torch.float32)
    assert set(syn) == keys == set(W.normalize_vae_state_dict(syn))
    from seedvr2_ref import dit as odit
    assert set(pkg.dit.synthetic_state_dict(pkg.dit.DiTConfig(vid_dim=64, heads=2, head_dim=32, txt_in_dim=32, num_layers=12, mlp_multiple_of=32), "cpu")) == \
        set(odit.param_shapes(odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, txt_in_dim=32, num_layers=12, mlp_multiple_of=32)))
    assert set(pkg.dit.synthetic_state_dict(pkg.dit.DiTConfig(vid_dim=64, heads=2, head_dim=32, txt_in_dim=32, num_layers=2, mm_layers=2,
                                                               mlp_type="normal", out_norm=False, last_vid_only=False), "cpu")) == \
        set(odit.param_shapes(odit.DiTConfig(vid_dim=64, heads=2, head_dim=32, txt_in_dim=32, num_layers=2, mm_layers=2,
                                             mlp_type="normal", out_norm=False, last_vid_only=False)))


def test_comfy_v3_schemas_mirror_the_reference_inputs(pkg, surface, monkeypatch):
    """define_schema() (only used inside ComfyUI) builds io.Schema objects with the reference's node ids, input names and
    defaults; checked here with a recording stand-in for comfy_api's `io`."""
    N = pkg.nodes

    class _Kind:
        def __init__(self, name):
            self.name = name

        def Input(self, n, **kw):
            return ("in", self.name, n, kw)

        def Output(self, **kw):
            return ("out", self.name, kw)

    class _IO:
        Image, Int, Float, Boolean, Combo = (_Kind(k) for k in ("Image", "Int", "Float", "Boolean", "Combo"))

        @staticmethod
        def Custom(tag):
            return _Kind("Custom:" + tag)

        @staticmethod
        def Schema(**kw):
            return kw

        class NodeOutput(tuple):
            def __new__(cls, *a):
                return tuple.__new__(cls, a)

    monkeypatch.setattr(N, "io", _IO)
    up = N.SeedVR2VideoUpscaler.define_schema()
    assert up["node_id"] == "SeedVR2VideoUpscaler" and up["category"] == "SEEDVR2"
    assert [i[2] for i in up["inputs"]] == surface["upscaler_inputs"]
    by = {i[2]: i for i in up["inputs"]}
    assert by["batch_size"][3]["default"] == 5 and by["batch_size"][3]["step"] == 4 and by["color_correction"][3]["default"] == "lab"
    assert by["color_correction"][3]["options"] == ["lab", "wavelet", "wavelet_adaptive", "hsv", "adain", "none"]
    assert by["dit"][1] == "Custom:SEEDVR2_DIT" and up["outputs"][0][1] == "Image"
    dit = N.SeedVR2LoadDiTModel.define_schema()
    assert [i[2] for i in dit["inputs"]] == [k for k in N.DIT_KEYS if k != "node_id"][:2] + ["blocks_to_swap", "swap_io_components",
                                                                                         "offload_device", "cache_model",
                                                                                         "attention_mode", "torch_compile_args"]
    assert dit["outputs"][0][1] == "Custom:SEEDVR2_DIT"
    vae = N.SeedVR2LoadVAEModel.define_schema()
    assert {i[2] for i in vae["inputs"]} == set(N.VAE_KEYS) - {"node_id"} and vae["outputs"][0][1] == "Custom:SEEDVR2_VAE"
    tc = N.SeedVR2TorchCompileSettings.define_schema()
    assert [i[2] for i in tc["inputs"]] == ["backend", "mode", "fullgraph", "dynamic", "dynamo_cache_size_limit", "dynamo_recompile_limit"]
    # inside ComfyUI the nodes answer with io.NodeOutput
    out = N.SeedVR2LoadVAEModel.execute(decode_tiled=True)
    assert isinstance(out, _IO.NodeOutput) and out[0]["decode_tiled"] is True


def test_cli_defaults_and_node_specs_match_reference_literals(pkg, surface):
    """Defaults / ranges read from the reference's own add_argument(...) and io.<Kind>.Input(...) calls (oracle/extract_surface.py)."""
    cli = _cli(pkg)
    a = vars(cli.parse_arguments(["in.png"]))
    skip = {"--dit_model", "--output_format"}        # product default names the same file explicitly; extra container formats
    for flag, want in surface["cli_defaults"].items():
        if "#" in flag or flag in skip:
            continue
        assert a[flag.lstrip("-")] == want, (flag, a[flag.lstrip("-")], want)
    assert a["dit_model"] == pkg.weights.DEFAULT_DIT
    assert set(surface["cli_defaults"]["--color_correction#choices"]) == {"lab", "wavelet", "wavelet_adaptive", "hsv", "adain", "none"}
    specs = {n: {s[0]: s for s in pkg.nodes._SCHEMAS[n]["inputs"]} for n in surface["node_inputs"]}
    for node, inputs in surface["node_inputs"].items():
        assert [s["name"] for s in inputs] == [s[0] for s in pkg.nodes._SCHEMAS[node]["inputs"]], node
        for s in inputs:
            mine = specs[node][s["name"]]
            kind = s["kind"].replace("io.", "")
            if kind.startswith("Custom"):
                assert mine[1] == "Custom:" + kind.split('"')[1] if '"' in kind else kind.split("'")[1], (node, s)
            else:
                assert mine[1] == kind, (node, s["name"], mine[1], kind)
            for key in ("default", "min", "max", "step"):
                if key in s and not (node.startswith("SeedVR2Load") and s["name"] == "model"):
                    assert mine[2].get(key) == s[key], (node, s["name"], key, mine[2].get(key), s[key])
            if "options" in s and s["name"] in ("color_correction", "attention_mode", "tile_debug", "backend", "mode"):
                assert mine[2]["options"] == s["options"], (node, s["name"])
