#!/usr/bin/env python3
"""Extract the drop-in SURFACE of the reference (names and defaults only, no code) into tests/golden/surface.json.

Build container only.  Nothing is imported from the reference: the CLI's `parse_arguments` and the four ComfyUI nodes'
`define_schema` are read as syntax trees and the literal arguments of every `add_argument(...)` / `io.<Kind>.Input(...)`
call are collected:
  cli_flags / cli_defaults            /root/reference/inference_cli.py:1216-1401
  upscaler_inputs / node_inputs       /root/reference/src/interfaces/{video_upscaler,dit_model_loader,vae_model_loader,
                                      torch_compile_settings}.py
usage: python oracle/extract_surface.py
"""
import ast
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "surface.json")


def lit(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return None


def cli_surface():
    tree = ast.parse(open(os.path.join(REF, "inference_cli.py")).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "parse_arguments")
    flags, defaults = [], {}
    calls = [n for n in ast.walk(fn) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == "add_argument"]
    for c in sorted(calls, key=lambda n: (n.lineno, n.col_offset)):
        name = lit(c.args[0])
        if not isinstance(name, str):
            continue
        flags.append(name)
        kw = {k.arg: lit(k.value) for k in c.keywords}
        if kw.get("action") == "store_true":
            defaults[name] = False
        elif "default" in kw:
            defaults[name] = kw["default"]
        if kw.get("choices") is not None:
            defaults[name + "#choices"] = kw["choices"]
    return flags, defaults


def node_inputs(path, cls):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    c = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "define_schema")
    out = []
    calls = [n for n in ast.walk(fn) if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and n.func.attr == "Input"]
    for call in sorted(calls, key=lambda n: (n.lineno, n.col_offset)):
        name = lit(call.args[0]) if call.args else None
        if not isinstance(name, str):
            continue
        kind = ast.unparse(call.func.value)          # io.Int / io.Combo / io.Custom("SEEDVR2_DIT") ...
        spec = {"name": name, "kind": kind}
        for k in call.keywords:
            if k.arg in ("default", "min", "max", "step", "options", "optional"):
                v = lit(k.value)
                if v is not None:
                    spec[k.arg] = v
        out.append(spec)
    return out


def main():
    flags, defaults = cli_surface()
    nodes = {
        "SeedVR2VideoUpscaler": node_inputs("src/interfaces/video_upscaler.py", "SeedVR2VideoUpscaler"),
        "SeedVR2LoadDiTModel": node_inputs("src/interfaces/dit_model_loader.py", "SeedVR2LoadDiTModel"),
        "SeedVR2LoadVAEModel": node_inputs("src/interfaces/vae_model_loader.py", "SeedVR2LoadVAEModel"),
        "SeedVR2TorchCompileSettings": node_inputs("src/interfaces/torch_compile_settings.py", "SeedVR2TorchCompileSettings"),
    }
    surface = {"cli_flags": flags, "upscaler_inputs": [s["name"] for s in nodes["SeedVR2VideoUpscaler"]],
               "cli_defaults": defaults, "node_inputs": nodes}
    with open(OUT, "w") as fh:
        json.dump(surface, fh, indent=1)
    print(f"{len(flags)} CLI flags, " + ", ".join(f"{k}: {len(v)} inputs" for k, v in nodes.items()), "->", OUT)


if __name__ == "__main__":
    main()
