"""CPU oracle for the SeedVR2 hot path (VAE encode -> one-step NaDiT -> VAE decode).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package (``comfyui-seedvr2_amd/``); only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

This is a plain-PyTorch (CPU) restatement of the reference algorithm, written by
reading /root/reference (disk02/ComfyUI-SeedVR2 v2.5.18); each function cites the
reference file:line it follows.

Parity status (see DESIGN.md "Oracle"):
  * pinned against the importable reference modules by ``oracle/pin_against_reference.py``
    (window.py, na.py, attention.py, modulation.py, mlp.py, mm.py, patch_v1.py,
    normalization.py, causal_inflation_lib.py, diffusion/*) -> ``tests/golden/*``.
  * "parity unpinned" for arithmetic living in dependencies that are absent here
    (rotary_embedding_torch, diffusers VAE blocks / timestep embedding,
    torchvision resize): restated from the libraries' published algorithms.
"""

from . import windows, rope, dit, vae, pipeline  # noqa: F401
