"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/aid_hip.h declares, the ctypes structs mirror the header, and the Python surface keeps the
reference's names / attributes / error behaviour.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

import aid_amd
from aid_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "aid_hip.h")).read()


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = set(re.findall(r"^\s*(?:int|size_t|const char\*)\s+(aid_\w+)\s*\(", HEADER, flags=re.M))
    assert declared == set(_lib.ABI_SYMBOLS), declared ^ set(_lib.ABI_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.aid_abi_version() == int(re.search(r"#define AID_ABI_VERSION (\d+)", HEADER).group(1))


def _c_struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HEADER, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)", decl)
        ctype, ptr, names = m.group(2), m.group(3), m.group(4)
        for nm in names.split(","):
            nm = nm.strip()
            isptr = bool(ptr) or nm.startswith("*")
            fields.append((nm.lstrip("* "), "ptr" if isptr else ctype))
    return fields


@pytest.mark.parametrize("name,cls", [("AidGemmProblem", _lib.AidGemmProblem), ("AidAttnArgs", _lib.AidAttnArgs),
                                      ("AidProcessorArgs", _lib.AidProcessorArgs)])
def test_ctypes_structs_mirror_header(name, cls):
    want = _c_struct_fields(name)
    got = []
    for fname, ftype in cls._fields_:
        kind = {ctypes.c_void_p: "ptr", ctypes.c_int32: "int32_t", ctypes.c_int64: "int64_t",
                ctypes.c_float: "float", ctypes.c_size_t: "size_t"}[ftype]
        got.append((fname, kind))
    assert got == want


def test_error_codes_have_text():
    lib = _lib.load()
    for code in (0, -1, -2, -3, -4, -5, -6):
        assert lib.aid_strerror(code)
    assert b"unknown" in lib.aid_strerror(-99)


def test_null_and_bad_arguments_return_codes_without_a_gpu():
    lib = _lib.load()
    assert lib.aid_attn_fwd(None, None) == -1
    assert lib.aid_processor_fwd(None, None) == -1
    assert lib.aid_gemm_nt(None, 1, 0, None) == -1
    a = _lib.AidProcessorArgs()
    assert lib.aid_processor_workspace_bytes(ctypes.byref(a)) == 0
    p = (_lib.AidGemmProblem * 1)()
    assert lib.aid_gemm_nt(p, 1, 7, None) == -2          # bad dtype
    assert lib.aid_gemm_nt(p, 9, 0, None) == -1          # too many problems


def test_workspace_size_query_is_pure_host_code():
    lib = _lib.load()
    a = _lib.AidProcessorArgs()
    for f in ("x", "wq", "wk", "wv", "wo", "y"):
        setattr(a, f, 0x1000)
    a.n_frames, a.s, a.c, a.heads, a.mode, a.dtype = 7, 4096, 640, 10, 0, 1
    n = lib.aid_processor_workspace_bytes(ctypes.byref(a))
    # q + k + vt + o, each [7, 4096, 640] bf16
    assert n == 4 * 7 * 4096 * 640 * 2
    a.heads = 7                                          # c % heads != 0
    assert lib.aid_processor_workspace_bytes(ctypes.byref(a)) == 0
    a.heads, a.c = 4, 512                                # head dim 128 unsupported
    assert lib.aid_processor_workspace_bytes(ctypes.byref(a)) == 0


# ---- Python surface (reference names / attributes / errors) -----------------------------------
def test_processor_state_surface_matches_reference():
    p = aid_amd.OuterInterpolatedAttnProcessor(t=0.3, is_fused=True)
    assert p.size == 3 and p.is_fused and p.activated and torch.allclose(p.coef, torch.tensor([0, 0.3, 1.0]))
    p.deactivate()
    assert not p.activated
    p.activate(0.6)
    assert p.activated and abs(float(p.coef[1]) - 0.6) < 1e-7
    with pytest.raises(AssertionError):
        p.activate(1.0)
    with pytest.raises(AssertionError):
        aid_amd.InnerInterpolatedAttnProcessor(t=0.0)
    q = aid_amd.InnerInterpolatedAttnProcessor(size=7, alpha=3, beta=3)
    assert q.size == 7 and q.coef[0] == 0 and q.coef[-1] == 1 and q.coef.dtype == torch.float32
    assert isinstance(q, torch.nn.Module) and q.original_attn is None
    for cls in (aid_amd.OuterInterpolatedIPAttnProcessor, aid_amd.InnerInterpolatedIPAttnProcessor,
                aid_amd.ScaleControlIPAttnProcessor):
        ipa = aid_amd.IPAdapterShim(80, 48, num_tokens=4, scale=0.5)
        r = cls(t=0.5, is_fused=True, ip_attn=ipa)
        assert r.num_tokens == (4,) and r.scale == [0.5] and r.ip_attn is ipa and r.size == 3


def test_no_cpu_fallback_fails_loudly():
    attn = aid_amd.AttnShim(80, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        aid_amd.OuterInterpolatedAttnProcessor(t=0.5)(attn, torch.randn(3, 8, 80))
    with pytest.raises(RuntimeError, match="no CPU path"):
        aid_amd.HipAttnProcessor()(attn, torch.randn(3, 8, 80))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_load_aid_wraps_every_attention_layer_and_toggles():
    unet = aid_amd.AttnStackUNet("sd15", dtype=torch.float32, scale_down=64, channel_div=8)
    assert len(unet.attn_processors) == 32
    aid_amd.load_aid(unet, t=0.5, is_fused=True, atype="fused_inner")
    procs = unet.attn_processors
    assert all(isinstance(p, aid_amd.InnerInterpolatedAttnProcessor) for p in procs.values())
    assert all(isinstance(p.original_attn, aid_amd.HipAttnProcessor) for p in procs.values())
    assert any(k.endswith("attn1.processor") for k in procs) and any(k.endswith("attn2.processor") for k in procs)
    aid_amd.deactivate_aid(unet)
    assert not any(p.activated for p in unet.attn_processors.values())
    aid_amd.activate_aid(unet, 0.25)
    assert all(p.activated and abs(float(p.coef[1]) - 0.25) < 1e-7 for p in unet.attn_processors.values())
    with pytest.raises(ValueError):
        unet.set_attn_processor({"a": None})
    sdxl = aid_amd.AttnStackUNet("sdxl", dtype=torch.float32, scale_down=64, channel_div=8)
    assert len(sdxl.attn_processors) == 140


def test_interp_helpers_match_goldens():
    import cases as C
    import numpy as np
    misc = C.load_fixture("misc_goldens.npz")
    v0, v1 = torch.from_numpy(misc["slerp_v0"]), torch.from_numpy(misc["slerp_v1"])
    for t in (0.0, 0.25, 0.5, 1.0):
        np.testing.assert_allclose(aid_amd.slerp(v0, v1, t).numpy(), misc[f"slerp_t{t}"], atol=2e-6)
    e0, e1 = torch.from_numpy(misc["emb0"]), torch.from_numpy(misc["emb1"])
    np.testing.assert_allclose(aid_amd.linear_interpolation(e0, e1, size=5).numpy(), misc["linear_size5"], atol=1e-6)
    np.testing.assert_allclose(aid_amd.linear_interpolation(e0, e1, ts=torch.tensor([0.1, 0.6])).numpy(),
                               misc["linear_ts"], atol=1e-6)
    np.testing.assert_allclose(aid_amd.spherical_interpolation(v0, v1, 4).numpy(), misc["spherical_size4"], atol=2e-6)
    for n, a, b in ((7, 3, 3), (16, 50, 50), (5, 25, 25)):
        np.testing.assert_array_equal(aid_amd.generate_beta_tensor(n, a, b).numpy(), misc[f"beta_{n}_{a}_{b}"])


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """sizeof / offsetof of every struct of include/aid_hip.h as gcc lays it out == the ctypes mirror in _lib.py."""
    import ctypes, subprocess
    structs = {"AidGemmProblem": _lib.AidGemmProblem, "AidAttnArgs": _lib.AidAttnArgs,
               "AidProcessorArgs": _lib.AidProcessorArgs, "AidProfileEntry": _lib.AidProfileEntry}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "aid_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(out[name]) == ctypes.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert int(out[f"{name}.{fname}"]) == getattr(cls, fname).offset, f"{name}.{fname}"


def test_integration_doc_stub_matches_the_header():
    """The ctypes stub a maintainer would copy from INTEGRATION.md lists the AidProcessorArgs fields of the header, in order."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class AidProcessorArgs(C.Structure)"):doc.index("lib.aid_processor_workspace_bytes.restype")]
    names = re.findall(r'\("(\w+)", C\.', stub)
    assert names == [f for f, _ in _lib.AidProcessorArgs._fields_]


# ---- no setting may change results ----------------------------------------------------
KNOB_MAX = {"GEMM_VARIANT": 31, "GEMM_PP": 3, "GEMM_TRI": 1, "ATTN_NW": 8, "ATTN_QB": 2, "ATTN_PIPE": 1, "ATTN_RES": 1,
            "ATTN_RES_CHUNKS": 64, "ATTN_ORDER": 1, "ATTN_V2": 1, "CU_SHARE": 8, "GEMM_RS": 1, "ATTN_TX": 1, "ATTN_TX_TILES": 64, "GEMM_LS": 1}


def test_tuning_knobs_refuse_values_outside_their_range():
    """The timing ablations of development builds were addressed as GEMM_PP >= 8 / ATTN_RES_CHUNKS > 100: the product library refuses
    such values (and has no code behind them)."""
    from aid_amd import ops
    lib = _lib.load()
    for name, top in KNOB_MAX.items():
        assert lib.aid_set_tuning(name.encode(), top + 1) != 0, name
        assert ops.get_tuning(name) == -1, name
        ops.set_tuning(name, top)
        assert ops.get_tuning(name) == top
        ops.set_tuning(name, -1)
        assert ops.get_tuning(name) == -1
    assert lib.aid_set_tuning(b"GEMM_PP", 9) != 0 and lib.aid_set_tuning(b"ATTN_RES_CHUNKS", 116) != 0
    assert lib.aid_set_tuning(b"NO_SUCH_KNOB", 1) != 0


def test_environment_cannot_select_an_ablation():
    """AID_GEMM_PP=9 / AID_ATTN_RES_CHUNKS=116 in the environment (the old ablation addresses) are ignored at library load."""
    import subprocess, sys
    env = dict(os.environ, AID_GEMM_PP="9", AID_ATTN_RES_CHUNKS="116", AID_ATTN_NW="8")
    code = ("import sys; sys.path.insert(0, %r); import aid_amd; from aid_amd import ops; "
            "print(ops.get_tuning('GEMM_PP'), ops.get_tuning('ATTN_RES_CHUNKS'), ops.get_tuning('ATTN_NW'))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True, text=True).stdout.split()
    assert out == ["-1", "-1", "8"]


def test_product_library_has_no_ablation_code():
    """No kernel of libaid_hip.so takes an ablation word, and the development build does not ship next to it."""
    import subprocess
    pkg = os.path.dirname(_lib.LIB_PATH)
    assert not os.path.exists(os.path.join(pkg, "libaid_abl.so"))
    txt = subprocess.run(["strings", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "ablation" not in txt.lower() and "AID_ABLATIONS" not in txt
    for src in ("aid_gemm.hip", "aid_attn_pp.hip", "aid_attn.hip"):
        body = open(os.path.join(_lib.CSRC_DIR, src)).read()
        out, depth = [], 0                                  # every mention of `abl` sits inside #ifdef AID_ABLATIONS
        for line in body.splitlines():
            t = line.strip()
            if t.startswith("#ifdef AID_ABLATIONS"):
                depth += 1
            elif depth and t.startswith("#if"):
                depth += 1
            elif depth and t.startswith("#endif"):
                depth -= 1
            elif depth == 0 and re.search(r"\babl_?\b", line.split("//")[0]):
                out.append(line)
        assert not out, (src, out[:3])
