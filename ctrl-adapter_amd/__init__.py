"""MI355X-native (gfx950) Ctrl-Adapter denoising hot path: ControlNet + Ctrl-Adapter forward behind the
reference's own module interface.  All arithmetic runs in libctrlhip.so (hand-written HIP); see DESIGN.md."""
from . import _lib  # noqa: F401
