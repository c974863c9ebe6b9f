"""seedvr2_amd - MI355X-native SeedVR2 upscaler hot path behind the reference's plugin surface.

Directory name follows the ComfyUI custom-node convention (hyphenated); import it through
``__graft_entry__.load_package()`` or ComfyUI's own loader.  Submodules:
  hip      ctypes binding of libseedvr2_hip.so (the C ABI, include/seedvr2_hip.h)
  windows  host-side window geometry / index tables of the swin attention
  dit      NaDiT runner on the HIP kernels
  vae      causal 3-D video VAE runner on the HIP kernels
  pipeline transform -> encode -> one-step DiT -> decode -> blend (the four generation phases)
"""

from . import hip  # noqa: F401
from . import windows  # noqa: F401
from . import dit  # noqa: F401
from . import vae  # noqa: F401
from . import pipeline  # noqa: F401
from . import parallel  # noqa: F401
from . import weights  # noqa: F401
from . import infer  # noqa: F401
from . import jpeg  # noqa: F401
from . import video_io  # noqa: F401
from . import nodes  # noqa: F401
from .nodes import comfy_entrypoint, SeedVR2Extension  # noqa: F401  (ComfyUI V3 entry point)
