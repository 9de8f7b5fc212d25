"""CPU: the C-ABI library loads and exports every symbol include/mil_hip.h declares (no compute
calls without a GPU), and the host-side bag/tile bookkeeping is right."""
import ctypes
import os

import numpy as np
import pytest

from mil_amd import _lib
from mil_amd.bags import build_tile_map


def test_library_exports_header_symbols():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    names = _lib.header_symbols()
    assert "mil_gate_scores_fwd" in names and "mil_attn_pool_fwd" in names
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/mil_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table out of sync with the header"
    handle.mil_abi_version.restype = ctypes.c_int
    assert handle.mil_abi_version() == _lib.ABI_VERSION


def test_missing_library_is_a_hard_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmil_hip.so")
    with pytest.raises(_lib.MilHipError):
        _lib.lib()


def test_ops_refuse_cpu_tensors():
    import torch
    from mil_amd import ops
    with pytest.raises(_lib.MilHipError):
        ops.gate_scores_fwd(torch.zeros(4, 512), *[torch.zeros(1)] * 6)


def test_tile_map_ragged():
    lengths = [1, 63, 64, 130, 0, 257]
    tm, bto, bo = build_tile_map(lengths)
    assert bo.tolist() == [0, 1, 64, 128, 258, 258, 515]
    assert bto.tolist() == [0, 1, 3, 5, 10, 10, 19]
    covered = np.zeros(515, dtype=int)
    for bag, row0, n, _ in tm:
        assert 1 <= n <= 32 and bo[bag] <= row0 and row0 + n <= bo[bag + 1]
        covered[row0:row0 + n] += 1
    assert (covered == 1).all()
    for b in range(len(lengths)):
        assert (tm[bto[b]:bto[b + 1], 0] == b).all()


@pytest.mark.parametrize("cname,pyname", [("mil_image_only_step", "ImageOnlyStep"), ("mil_small_dw_desc", "SmallDwDesc"),
                                          ("mil_cohort_feed_desc", "CohortFeedDesc")])
def test_struct_layouts_match_the_header(tmp_path, cname, pyname):
    """The ctypes mirrors of the C structs (the one-call step, the grouped weight-gradient descriptor) against the C compiler's
    view of include/mil_hip.h."""
    import ctypes
    import subprocess
    from mil_amd import _lib
    st = getattr(_lib, pyname)
    fields = [f[0] for f in st._fields_]
    src = tmp_path / "layout.c"
    prints = "\n".join(f'    printf("{f} %zu\\n", offsetof({cname}, {f}));' for f in fields)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{_lib.HEADER_PATH}"\nint main(void) {{\n'
                   f'    printf("sizeof %zu\\n", sizeof({cname}));\n{prints}\n    return 0;\n}}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(out["sizeof"]) == ctypes.sizeof(st)
    for f in fields:
        assert int(out[f]) == getattr(st, f).offset, f


def test_torch_cpp_extension_shim_loads_and_binds_the_same_abi():
    """csrc/torch_shim.cpp (built by __graft_entry__.build()): the torch cpp_extension binding north_star names, over the same
    extern "C" entries.  Loads without a GPU and reports the library's ABI version; no compute here."""
    import os
    import pytest
    from mil_amd import _lib
    if not os.path.exists(os.path.join(_lib.SHIM_DIR, "mil_torch_shim.so")):
        pytest.skip("shim not built (run __graft_entry__.build())")
    os.environ["MIL_TORCH_SHIM"] = "1"               # opt-in (it measured no faster than ctypes)
    _lib._shim = False
    try:
        sh = _lib.shim()
    finally:
        os.environ.pop("MIL_TORCH_SHIM", None)
        _lib._shim = False
    assert sh is not None and sh.abi_version() == _lib.ABI_VERSION
    for name in ("linear_small_fwd", "linear_small_bwd", "layernorm_fwd", "layernorm_bwd_res", "absorb_query",
                 "absorb_query_bwd", "value_proj_bwd"):
        assert callable(getattr(sh, name))
