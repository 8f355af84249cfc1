"""The C-ABI library loads and exports every symbol include/ctrl_hip.h declares (no compute calls: CPU box)."""
import ctypes
import os
import re

from helpers import ROOT


def test_library_exports_every_declared_symbol():
    import ctrl_adapter_amd  # noqa: F401
    from ctrl_adapter_amd import _lib
    lib = _lib.lib()
    header = open(os.path.join(ROOT, "include", "ctrl_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)        # drop comments
    declared = set(re.findall(r"\b(ctrl_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), "libctrlhip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.ctrl_abi_version() == _lib.ABI_VERSION == 3


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (sizes computed from the header's field order by gcc)."""
    import subprocess
    import tempfile
    from ctrl_adapter_amd import _lib
    src = '#include "ctrl_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ctrl_igemm_seg), sizeof(ctrl_igemm_desc), sizeof(ctrl_attn_desc), sizeof(ctrl_tattn_desc), sizeof(ctrl_tensor_ref), sizeof(ctrl_controlnet_config), sizeof(ctrl_adapter_config), sizeof(ctrl_clip_comm));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    got = [ctypes.sizeof(c) for c in (_lib.IGemmSeg, _lib.IGemmDesc, _lib.AttnDesc, _lib.TAttnDesc, _lib.TensorRef,
                                     _lib.ControlNetConfig, _lib.AdapterConfig, _lib.ClipComm)]
    assert [int(v) for v in out] == got, (out, got)
