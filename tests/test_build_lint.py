"""CPU: compile every kernel of the library to gfx950 ISA (same flags as the Makefile) and run the spill lint
(tools/lint_exec_spills.py) on it: a full-wave value moved to an AGPR while EXEC is narrowed and read back after the
region is the miscompile that crashed two build variants of the f64 fill kernel this round.  No GPU needed."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "malstroem_amd" / "csrc"
sys.path.insert(0, str(ROOT / "tools"))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only"]
EXTRA = {"fill.hip": ["-fno-honor-nans"]}


def _isa(src, out):
    r = subprocess.run(["hipcc"] + FLAGS + EXTRA.get(src.name, []) + [str(src), "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


@pytest.mark.timeout(900)
def test_no_full_wave_value_is_spilled_under_narrowed_exec(tmp_path):
    import lint_exec_spills
    srcs = sorted(CSRC.glob("*.hip"))
    assert len(srcs) >= 8
    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(lambda s: _isa(s, tmp_path / (s.stem + ".s")), srcs))
    nkern, bad, scratch, geo, flood = 0, [], {}, {}, {}
    for o in outs:
        nkern += sum(1 for _ in lint_exec_spills.kernels(str(o)))
        bad += lint_exec_spills.lint(str(o))
        if o.stem == "fill":
            scratch = lint_exec_spills.private_segments(str(o))
        if o.stem == "noflat_geo":
            geo = lint_exec_spills.private_segments(str(o))
        if o.stem == "pflood":
            flood = lint_exec_spills.private_segments(str(o))
    assert nkern >= 40            # the lint really saw the kernels
    assert not bad, bad
    # the fill kernels hold their 64 x 64 windows in registers by design: any private segment means the window spilled
    # (the build that faulted on 3 Oct had 100 bytes of it); every shipped fill kernel must have none
    assert sum("fill_round_kernel" in k for k in scratch) >= 5, scratch
    assert all(v == 0 for v in scratch.values()), scratch
    # so do the geodesic no-flats kernels since the windows of one and two classes relax on bit masks (end of round 3: 64 registers of
    # distances + 13 mask words; the full adjacency words only for windows of three or more classes)
    hot = {k: v for k, v in geo.items() if "ng_round_kernel" in k or "ng_first_kernel" in k}
    assert len(hot) == 2 and all(v == 0 for v in hot.values()), geo
    # the flood's kernels: the seed-graph solve kept a dozen per-thread address words in scratch until round 4 (hoisted out of the
    # visit loop) and every level load of a visit waited for a scratch reload first: 0.2 ms of the fill stage
    assert len(flood) >= 10 and all(v == 0 for v in flood.values()), flood


def test_lint_catches_the_pattern(tmp_path):
    """the miscompile in miniature: save under narrowed exec, read back after the region"""
    import lint_exec_spills
    asm = """_Zkernel:
	v_mov_b32_e32 v5, 0
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_2
	v_accvgpr_write_b32 a7, v5
.LBB0_2:
	s_or_b64 exec, exec, s[2:3]
	v_mov_b32_e32 v5, v9
	v_accvgpr_read_b32 v5, a7
	global_load_dword v1, v5, s[4:5]
	s_endpgm
_Zbenign:
	s_and_saveexec_b64 s[2:3], vcc
	v_accvgpr_write_b32 a7, v5
	global_store_dword v0, a7, s[6:7]
	s_or_b64 exec, exec, s[2:3]
	v_accvgpr_write_b32 a7, v1
	v_accvgpr_read_b32 v2, a7
	s_endpgm
"""
    p = tmp_path / "t.s"
    p.write_text(asm)
    res = lint_exec_spills.lint(str(p))
    assert len(res) == 1 and res[0][0] == "_Zkernel"


def test_lint_catches_the_scratch_form(tmp_path):
    """the same hazard through a scratch spill slot; a slot that a full-EXEC store filled first is a per-lane PHI (benign)"""
    import lint_exec_spills
    asm = """_Zbad:
	s_and_saveexec_b64 s[2:3], vcc
	scratch_store_dwordx2 off, v[2:3], off offset:8 ; 8-byte Folded Spill
	v_mov_b32_e32 v2, v9
	s_or_b64 exec, exec, s[2:3]
	scratch_load_dword v5, off, off offset:12 ; 4-byte Folded Reload
	s_endpgm
_Zphi:
	scratch_store_dwordx2 off, v[4:5], off  ; 8-byte Folded Spill
	s_and_saveexec_b64 s[2:3], vcc
	scratch_store_dwordx2 off, v[2:3], off  ; 8-byte Folded Spill
	s_or_b64 exec, exec, s[2:3]
	scratch_load_dwordx2 v[22:23], off, off ; 8-byte Folded Reload
	s_endpgm
_Zinside:
	s_and_saveexec_b64 s[2:3], vcc
	scratch_store_dword off, v2, off offset:4 ; 4-byte Folded Spill
	scratch_load_dword v2, off, off offset:4 ; 4-byte Folded Reload
	s_or_b64 exec, exec, s[2:3]
	s_endpgm
"""
    p = tmp_path / "t.s"
    p.write_text(asm)
    res = lint_exec_spills.lint(str(p))
    assert [r[0] for r in res] == ["_Zbad"]


def _residency(asm_text):
    """kernel name -> (VGPRs, LDS bytes, threads, resident workgroups per CU) from the code object's metadata (MI355X: 512 VGPRs per
    SIMD lane in granules of 8, 160 KB of LDS, at most 8 wavefronts per SIMD)"""
    import re
    out = {}
    pat = (r"\.group_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.max_flat_workgroup_size:\s+(\d+)\n\s+\.name:\s+(\S+)\n(?:.*\n)*?"
           r"\s+\.vgpr_count:\s+(\d+)")
    for m in re.finditer(pat, asm_text):
        lds, wg, name, vg = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
        waves = wg // 64
        by_v = min(8, 512 // max(8, (vg + 7) // 8 * 8)) * 4 // waves
        by_l = (160 * 1024) // lds if lds else 99
        out[name] = (vg, lds, wg, min(by_v, by_l, 32 // waves))
    return out


@pytest.mark.timeout(600)
def test_tile_kernels_keep_their_resident_workgroups(tmp_path):
    """The tile kernels of the tail wait on LDS round trips and barriers: workgroups per CU is what their speed hangs on, and a few
    registers or kilobytes too many cost a quarter of it without a test failing (end of round 3: the labelling's tile union-find ran
    5 of 8 workgroups, the statistics 6, the watersheds' tile pass 4 after the pour-point candidates were added)."""
    want = {"ccl.hip": {"ccl_tile_kernelIf": 8}, "label_ops.hip": {"stats_kernelILb1ELi1E": 8, "stats_kernelILb1ELi2E": 8}, "watershed.hip": {"ws_tile_kernel": 8},
            "accum.hip": {"accum_tile_kernelILb0ELb0ELb1E": 4, "accum_final_walk_kernel": 6}}
    for fn, kernels in want.items():
        res = _residency(_isa(CSRC / fn, tmp_path / (fn + ".s")).read_text())
        for frag, n in kernels.items():
            hit = [v for k, v in res.items() if frag in k]
            assert hit, (fn, frag, sorted(res)[:5])
            assert all(v[3] >= n for v in hit), (fn, frag, hit)


def _dpp_hazards(asm_text, kernel_frag):
    """DPP instructions of a kernel with a source VGPR written by one of the two instructions in front (gfx9: a VALU write needs two wait
    states before a DPP read; the compiler pads its own code with s_nop but does not look inside inline-asm text)"""
    import re
    bad, inside, hist = [], False, []          # hist: (wait states this line provides, registers it writes)
    for line in asm_text.splitlines():
        if re.match(r"^_Z\w+:", line):
            inside, hist = kernel_frag in line, []
            continue
        if not inside:
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        if t.startswith("s_endpgm"):
            inside = False
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(";")[0].split(",")]
        if op == "s_nop":
            hist.append((int(ops[0], 0) + 1, set()))
            continue
        regs = lambda o: (set(range(int(m.group(1)), int(m.group(2)) + 1)) if (m := re.match(r"v\[(\d+):(\d+)\]", o)) else
                          ({int(o[1:])} if re.fullmatch(r"v\d+", o) else set()))
        if "_dpp" in op or " wave_sh" in t or " row_sh" in t or "quad_perm" in t:
            srcs = set().union(*[regs(o.split(" ")[0]) for o in ops[1:]]) if len(ops) > 1 else set()
            need, k = 2, len(hist) - 1
            while need > 0 and k >= 0:
                ws, wr = hist[k]
                if wr & srcs:
                    bad.append(t)
                    break
                need -= ws
                k -= 1
        writes = regs(ops[0]) if op.startswith("v_") and ops else set()
        hist.append((1, writes))
        hist = hist[-6:]
    return bad


@pytest.mark.timeout(600)
def test_no_dpp_read_within_two_slots_of_the_write_in_the_hand_written_rows(tmp_path):
    """The no-flats passes are inline-asm text (NG_ROW, NG_ROW2, NG_ROW_OPEN in noflat_geo.hip) with the DPP hazard kept by instruction
    order: the row behind is written by the last-but-one instruction of a row and read through DPP by the second of the next."""
    text = _isa(CSRC / "noflat_geo.hip", tmp_path / "noflat_geo.s").read_text()
    assert text.count("_dpp") > 500
    assert _dpp_hazards(text, "ng_round_kernel") == []
    assert _dpp_hazards(text, "ng_first_kernel") == []
    # the checker sees the pattern
    demo = "_Zdemo_kernel:\n\tv_bfi_b32 v5, v1, v2, v3\n\tv_add_u32 v6, v5, v7\n\tv_add_u32_dpp v8, v5, v9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_endpgm\n"
    assert len(_dpp_hazards(demo, "demo_kernel")) == 1
    demo_ok = demo.replace("\tv_add_u32 v6, v5, v7\n", "\tv_add_u32 v6, v5, v7\n\ts_nop 0\n")
    assert _dpp_hazards(demo_ok, "demo_kernel") == []
