"""Register budgets of the built gfx950 kernels (CPU test: reads the tables the build leaves next to the objects).

The launch heuristics in csrc/aid_attn.hip / aid_gemm.hip assume a number of waves per SIMD for every kernel variant
(VGPRs <= 168 -> 3, <= 256 -> 2, anything that needs AGPR copies -> 1).  A refactor that moves a variant across one of
those lines costs 20 - 40 % of its speed without failing any parity test — it happened in round 2 (OUTER d64 nw4:
256 -> 256 + 32 AGPRs, S = 1024 launch 179 -> 227 us) and was only found by bisecting against an older library."""
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "attention-interpolation-diffusion_amd", "csrc")


def _table(name):
    path = os.path.join(CSRC, name + ".resources.txt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not there: build the library first (python -c 'import __graft_entry__ as g; g.build()')")
    out = {}
    for blk in open(path).read().split("Name: ")[1:]:
        sym = blk.split()[0]
        num = lambda key: int(re.search(re.escape(key) + r": (\d+)", blk).group(1))   # noqa: E731
        out[sym] = dict(vgpr=num("VGPRs"), agpr=num("AGPRs"), scratch=num("ScratchSize [bytes/lane]"),
                        occ=num("Occupancy [waves/SIMD]"), spill=num("VGPRs Spill"), sgpr_spill=num("SGPRs Spill"))
    return out


def _attn(tab):
    """(dtype, d, mode, nw, qb, pipe, res) -> resources"""
    out = {}
    for sym, r in tab.items():
        m = re.search(r"aid_attn_kernelIDF16(b?)_?Li(\d+)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)", sym)
        if m:
            out[("bf16" if m.group(1) else "f16", int(m.group(2)), "pio"[int(m.group(3))], int(m.group(4)), int(m.group(5)),
                 bool(int(m.group(6))), bool(int(m.group(7))))] = r
    return out


def test_no_kernel_spills_or_uses_scratch():
    for obj in ("aid_attn", "aid_attn_pp", "aid_attn_tx", "aid_gemm", "aid_gemm_rs", "aid_f32", "aid_norm"):
        for sym, r in _table(obj).items():
            assert r["scratch"] == 0 and r["spill"] == 0, (obj, sym, r)


def test_sgpr_spills_are_bounded():
    """SGPR spills go to lanes of a spare VGPR (v_writelane / v_readlane), not to memory — cheap, but each one is VALU issue
    slots in kernels that are VALU-issue bound, and a jump in the count means a refactor pushed scalar state (segment
    pointers, descriptors) out of the 102 SGPRs.  The GEMM and LayerNorm kernels spill none; the attention kernel's
    three-segment variants spill up to 30 (measured at this commit; they sit outside the tile loop: pointers of the
    segments not being walked); the ping-pong kernel keeps two item records (the one it computes, the one it planned) and spills
    38 (PLAIN) / 57 - 115 (INNER, OUTER) of them in its per-mode instantiations of round 4 (82 in the one-kernel-for-all-modes
    version of round 3) — all in the prologue / item planning / item boundary, at most three v_readlane in a PLAIN V slot (counted
    per barrier interval in the ISA; an earlier version with spills in the loop was 9 % slower); the text-key kernel (aid_attn_tx) keeps the whole
    argument record live across its tile loop and spills 20 (PLAIN) / 63 (INNER / OUTER: three segments' roles and sources).
    VERDICT r2 weak #10."""
    for obj, bound in (("aid_gemm", 0), ("aid_gemm_rs", 0), ("aid_norm", 0), ("aid_attn_pp", 120), ("aid_attn_tx", 72), ("aid_attn", 32)):
        for sym, r in _table(obj).items():
            assert r["sgpr_spill"] <= bound, (obj, sym, r)


def test_attention_variants_keep_the_waves_per_simd_the_launcher_assumes():
    a = _attn(_table("aid_attn"))
    assert len(a) >= 60
    for dt in ("f16", "bf16"):
        # the variants the two bench stacks launch (launch_nw / attn_nw / attn_qb / attn_pipe / attn_res in aid_attn.hip)
        want = {
            (dt, 64, "p", 4, 1, False, False): 3,      # SDXL PLAIN: three waves per SIMD
            (dt, 64, "o", 8, 1, False, False): 2,      # SDXL OUTER, L >= 2048
            (dt, 64, "o", 4, 1, False, False): 2,      # SDXL OUTER, S = 1024 and the 77-key launches
            (dt, 40, "i", 4, 1, False, False): 3,      # SD1.5 INNER: three waves per SIMD (the pipelined variant has two)
            (dt, 40, "p", 4, 2, False, False): 2,      # SD1.5 PLAIN, 64 rows per wave
            (dt, 40, "i", 4, 1, False, True): 3,       # SD1.5 77-key launches: resident segments
            (dt, 40, "p", 4, 1, False, True): 3,
            (dt, 80, "i", 4, 1, False, False): 2,
            (dt, 80, "p", 4, 1, False, False): 2,
            (dt, 160, "i", 4, 1, False, False): 2,
            (dt, 160, "p", 4, 1, False, False): 2,
        }
        for key, occ in want.items():
            assert key in a, key
            assert a[key]["occ"] >= occ and (a[key]["agpr"] == 0 or occ == 1), (key, a[key])
    # every built variant except the known one-wave ones (d160 OUTER, OUTER with the pipelined loop) gets two waves
    one_wave = [k for k, r in a.items() if r["occ"] < 2]
    assert all((k[1] == 160 and k[2] == "o") or (k[2] == "o" and k[5]) for k in one_wave), one_wave


def test_pingpong_attention_keeps_two_waves_per_simd():
    """The ping-pong kernel's premise is one wave of each group per SIMD (8 waves per CU): <= 256 registers per wave with the parked
    state of a two-sided frame and the -m block in them (round 4, one instantiation per mode: PLAIN / INNER 215 - 220, OUTER 246 - 250;
    the park / swap code inside the unrolled tile loop had it at 256 + 14 spills, a select between two by-value argument fields
    had put 416 B per lane into scratch, and staging OUTER's output through LDS needed 256 + 8 spills — it stores directly).
    """
    syms = _table("aid_attn_pp")
    assert sum("aid_attn_pp_kernel" in s for s in syms) == 6
    for sym, r in syms.items():
        if "aid_attn_pp_kernel" in sym:
            assert r["vgpr"] + r["agpr"] <= 256 and r["occ"] >= 2 and r["scratch"] == 0 and r["spill"] == 0, (sym, r)


def test_text_key_attention_keeps_the_waves_per_simd_its_latency_hiding_needs():
    """aid_attn_tx hides the latency of its Q rows behind OTHER waves: the PLAIN instantiation has to stay at four waves per SIMD
    (<= 128 VGPRs: 16 score registers + two 16-register output blocks + two Q tiles), the OUTER one at two with three output blocks
    and no scratch (profiles/r05_attn_tx_notes.txt)."""
    syms = _table("aid_attn_tx")
    assert len(syms) == 4
    for sym, r in syms.items():
        plain = "Li1EEE" in sym
        assert r["occ"] >= (4 if plain else 2) and r["scratch"] == 0 and r["spill"] == 0, (sym, r)


def test_gemm_engines_fit_their_workgroups_per_cu():
    g = _table("aid_gemm")
    for sym, r in g.items():
        if "aid_gemm_nt_pp_kernel" in sym:
            assert r["vgpr"] + r["agpr"] <= 256, (sym, r)         # 8 waves per CU = 2 per SIMD
        if "aid_gemm_nt_pipe_kernel" in sym:
            assert r["vgpr"] + r["agpr"] <= 128, (sym, r)         # 2 workgroups of 8 waves per CU = 4 per SIMD


def test_row_stationary_gemm_keeps_its_rows_in_registers_without_spilling():
    """csrc/aid_gemm_rs.hip: a wave holds 32 activation rows x the whole K in VGPRs (K = 640: 160, K = 320: 80) next to four accumulator
    blocks and the fragment window — two waves per SIMD (K = 640, one workgroup of eight waves per CU; K = 320: two workgroups of four),
    so at most 256 registers, and NOTHING in scratch: a spill reload is a scratch_load, whose vmcnt(0) would drain the LDS-DMA ring
    in every slice step (it did while the kernel was being written: 20 - 56 spilled registers at three points)."""
    g = _table("aid_gemm_rs")
    kernels = {s: r for s, r in g.items() if "aid_gemm_rs_kernel" in s}
    assert len(kernels) == 4, list(g)
    for sym, r in kernels.items():
        assert r["vgpr"] + r["agpr"] <= 256 and r["occ"] >= 2 and r["scratch"] == 0 and r["spill"] == 0 and r["sgpr_spill"] == 0, (sym, r)


def test_row_stationary_lds_layout_is_consistent_and_conflict_free():
    """The DMA image of a weight slice against the fragment read addresses of the kernel (tools/dev/rs_layout_check.py restates both):
    every lane reads the chunk its MFMA operand needs, no ds_read_b128 lane group has a bank conflict."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rs_layout_check", os.path.join(root, "tools", "dev", "rs_layout_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(640, 8) == 1 and mod.check(320, 4) == 1
