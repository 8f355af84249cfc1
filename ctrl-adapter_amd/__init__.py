"""MI355X-native (gfx950) Ctrl-Adapter denoising hot path: ControlNet + Ctrl-Adapter (+ router) forward behind the
reference's own module interface.  All arithmetic runs in libctrlhip.so (hand-written HIP); see DESIGN.md."""
from . import _lib  # noqa: F401
from .controlnet import ControlNetModel, ControlNetOutput, MultiControlNetModel, pool_latents  # noqa: F401
from .ctrl_adapter import ControlNetAdapter  # noqa: F401
from .ctrl_router import ControlNetRouter  # noqa: F401
from .fused import controlled_step  # noqa: F401
from .image_prep import prepare_images  # noqa: F401
