"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "vfmseg_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|const char\*)\s+(vfm_\w+)\s*\(", txt)))


def test_header_declares_entry_points():
    names = _declared()
    assert "vfm_gemm" in names and "vfm_attn_fwd" in names and "vfm_upsample_ce" in names and len(names) >= 35


@pytest.mark.parametrize("which,kind", [("LIB", 0), ("LIB_F16", 1)])
def test_library_exports_every_declared_symbol(which, kind):
    """Both builds of the sources (bf16 and the fp16 twin, csrc/common.h) export the whole header and say which they are."""
    from vfmseg_amd.csrc import build
    lib_path = getattr(build, which)
    if not os.path.exists(lib_path):
        build.build(verbose=False, only="f16" if kind else "bf16")
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.vfm_abi_version.restype = ctypes.c_int
    from vfmseg_amd.lib import ABI_VERSION
    assert lib.vfm_abi_version() == ABI_VERSION   # (the binding refuses an older in-tree library: descriptor layouts changed)
    lib.vfm_half_kind.restype = ctypes.c_int
    assert lib.vfm_half_kind() == kind


def test_python_binding_matches_header():
    from vfmseg_amd import lib as L
    declared = set(_declared()) - {"vfm_last_error", "vfm_abi_version", "vfm_half_kind"}
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    # argument counts agree with the header prototypes
    txt = open(os.path.join(ROOT, "include", "vfmseg_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name, args in L.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", txt, flags=re.S)
        assert m, name
        n = len([a for a in m.group(1).split(",") if a.strip()])
        assert n == len(args), (name, n, len(args))


def test_plan_op_layout_matches_the_header(tmp_path):
    """vfm_plan_op (launch plans) is a tagged union the Python side fills field by field: its ctypes mirror must have the C layout.
    A C program compiled against the header prints sizeof / offsetof; no GPU needed."""
    import subprocess
    from vfmseg_amd import lib as L
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vfmseg_hip.h"\nint main(void) {\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vfm_plan_op), offsetof(vfm_plan_op, flops), offsetof(vfm_plan_op, u), '
                   'sizeof(vfm_gemm_desc), sizeof(vfm_attn_desc), offsetof(vfm_plan_op, u.ln_drop.offset), offsetof(vfm_plan_op, u.ln_drop.rows), '
                   'offsetof(vfm_plan_op, u.ln_bwd.t_scale), offsetof(vfm_plan_op, u.cast.colscale), offsetof(vfm_plan_op, u.copy.ds), '
                   'offsetof(vfm_plan_op, u.copy.accumulate), offsetof(vfm_plan_op, u.ln_fwd.stats));\nreturn 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    P, U = L.PlanOp, L.PlanOp.u.offset
    want = [ctypes.sizeof(P), P.flops.offset, U, ctypes.sizeof(L.GemmDesc), ctypes.sizeof(L.AttnDesc), U + L._LnDrop.offset.offset,
            U + L._LnDrop.rows.offset, U + L._LnBwd.t_scale.offset, U + L._CastOp.colscale.offset, U + L._CopyOp.ds.offset,
            U + L._CopyOp.accumulate.offset, U + L._LnFwd.stats.offset]
    assert got == want, (got, want)


def test_ops_refuse_cpu_tensors():
    import torch
    from vfmseg_amd import lib as L, ops
    with pytest.raises(L.HipError):
        ops.cast(torch.zeros(4, 4), torch.zeros(4, 4))
