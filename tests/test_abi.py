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
    assert lib.ctrl_abi_version() == _lib.ABI_VERSION == 7


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (sizes computed from the header's field order by gcc)."""
    import subprocess
    import tempfile
    from ctrl_adapter_amd import _lib
    src = '#include "ctrl_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ctrl_igemm_seg), sizeof(ctrl_igemm_desc), sizeof(ctrl_attn_desc), sizeof(ctrl_tattn_desc), sizeof(ctrl_tensor_ref), sizeof(ctrl_controlnet_config), sizeof(ctrl_adapter_config), sizeof(ctrl_clip_comm), sizeof(ctrl_ffn_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    got = [ctypes.sizeof(c) for c in (_lib.IGemmSeg, _lib.IGemmDesc, _lib.AttnDesc, _lib.TAttnDesc, _lib.TensorRef,
                                     _lib.ControlNetConfig, _lib.AdapterConfig, _lib.ClipComm, _lib.FfnDesc)]
    assert [int(v) for v in out] == got, (out, got)


def test_every_tile_walk_order_is_a_bijection():
    """csrc/tile_order.h decides which output tile a workgroup of the implicit GEMM computes; any order is correct as long as
    it is a bijection of the grid, so that is what is proved here (host integer code, no GPU): every mode x group width over
    the grid shapes of the path (incl. the ragged ones that fall back to the legacy walk), and the locality property the
    grouped walk exists for."""
    from ctrl_adapter_amd import _lib
    lib = _lib.lib()
    tm, tn = ctypes.c_int(), ctypes.c_int()

    def walk(ntm, ntn, mode, group):
        out = []
        for bid in range(ntm * ntn):
            assert lib.ctrl_igemm_tile_of(bid, ntm, ntn, mode, group, ctypes.byref(tm), ctypes.byref(tn)) == 0
            out.append((tm.value, tn.value))
        return out

    grids = [(512, 32), (128, 16), (32, 20), (8, 40), (8, 8), (64, 4), (512, 1), (1, 1), (3, 5), (7, 16), (24, 3), (40, 13)]
    for ntm, ntn in grids:
        want = {(m, n) for m in range(ntm) for n in range(ntn)}
        for mode in (0, 1, 2):
            for group in (0, 1, 2, 3, 5, 8, 12, 16, 64):
                got = walk(ntm, ntn, mode, group)
                assert len(set(got)) == len(got) == ntm * ntn and set(got) == want, (ntm, ntn, mode, group)
    # legacy == the M-major list cut into 8 contiguous ranges, one per XCD (workgroup b runs on XCD b % 8)
    got = walk(16, 4, 0, 0)
    assert [got[b] for b in range(0, 64, 8)] == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3)]
    assert got[1] == (2, 0) and got[7] == (14, 0)
    # xcd_m, groups of 2: an XCD sweeps its own 2 activation panels against weight panels {0, 1}, then against {2, 3}
    got = walk(16, 4, 1, 2)
    assert [got[b] for b in range(0, 64, 8)] == [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (0, 3), (1, 2), (1, 3)]
    assert {got[b][0] for b in range(3, 128 // 2, 8)} == {6, 7}
    # xcd_n: XCD x owns weight panels [x * ntn / 8, (x + 1) * ntn / 8) and every activation panel
    got = walk(4, 16, 2, 0)
    assert {got[b][1] for b in range(5, 64, 8)} == {10, 11} and {got[b][0] for b in range(5, 64, 8)} == {0, 1, 2, 3}
    assert lib.ctrl_igemm_tile_of(64, 8, 8, 0, 0, ctypes.byref(tm), ctypes.byref(tn)) == 1      # out of range: refused
    for spec, rc in ((b"legacy", 0), (b"m,16", 0), (b"n,0", 0), (b"sideways", 1), (b"auto", 0)):
        assert lib.ctrl_igemm_set_order(spec) == rc, spec
