"""The build-time guard of the kernels that issue loads from inline assembly (tools/check_inflight.py, run by
kmertools_amd/csrc/Makefile on the device assembly of kt_bulk.hip): it must refuse a copy of a register between the asm load
into it and the asm wait behind it - the fault round 4 found in its own build of record - and accept the forms the kernels
use."""
import importlib.util
import pathlib

ROOT = pathlib.Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("check_inflight", ROOT / "tools" / "check_inflight.py")
ci = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ci)

HEAD = "_ZN1x17part2_swwc_kernelIjEEv:\n"
LOAD = "\t;;#ASMSTART\n\tbuffer_load_dword v4, v21, s[48:51], 0 offen\n\tbuffer_load_dwordx2 v[6:7], v21, s[48:51], 0 offen\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(8)\n\tv_mov_b32 v12, v4\n\tv_mov_b64 v[14:15], v[6:7]\n\t;;#ASMEND\n"


def run(tmp_path, body, pattern="part2_swwc_kernel"):
    f = tmp_path / "k.s"
    f.write_text(HEAD + body + "\ts_endpgm\n")
    return ci.check(str(f), pattern)


def test_clean_kernel_is_accepted(tmp_path):
    bad, n_loads, n_waits = run(tmp_path, LOAD + "\tv_add_u32_e32 v30, v31, v32\n\tds_write_b32 v33, v30\n" + WAIT + "\tv_add_u32_e32 v4, v12, v12\n")
    assert bad == [] and n_loads == 2 and n_waits == 1


def test_copy_in_front_of_the_wait_is_refused(tmp_path):
    # what the register allocator did to the in-place wait: the copy it belongs behind the wait sits in front of it
    bad, _, _ = run(tmp_path, LOAD + "\tv_mov_b32_e32 v12, v4\n" + WAIT)
    assert len(bad) == 1 and bad[0][3] == [4]
    bad, _, _ = run(tmp_path, LOAD + "\tv_mov_b64_e32 v[14:15], v[6:7]\n" + WAIT)
    assert len(bad) == 1 and bad[0][3] == [6, 7]


def test_reuse_of_a_register_in_flight_is_refused(tmp_path):
    bad, _, _ = run(tmp_path, LOAD + "\tds_write_b32 v4, v108\n" + WAIT)   # (a live-range split under register pressure)
    assert len(bad) == 1


def test_compiler_wait_for_everything_lands_the_loads(tmp_path):
    bad, _, _ = run(tmp_path, LOAD + "\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v12, v4\n" + WAIT)
    assert bad == []


def test_dead_high_half_of_an_addend_pair_is_accepted_a_live_one_is_not(tmp_path):
    # lo32(a * b) + x written as v_mad_u64_u32 with the pair {x, whatever lies beside it}: fine while the result's high half dies
    mad = "\tv_mad_u64_u32 v[84:85], s[36:37], v56, s35, v[18:19]\n"
    load19 = "\t;;#ASMSTART\n\tbuffer_load_dword v19, v21, s[48:51], 0 offen\n\t;;#ASMEND\n"
    wait19 = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v40, v19\n\t;;#ASMEND\n"
    bad, _, _ = run(tmp_path, load19 + mad + "\tv_lshrrev_b32_e32 v60, 22, v84\n\tv_mov_b32_e32 v85, 0\n" + wait19)
    assert bad == []
    bad, _, _ = run(tmp_path, load19 + mad + "\tv_add_u32_e32 v60, v85, v84\n" + wait19)
    assert len(bad) == 1


def test_other_kernels_are_not_looked_at(tmp_path):
    f = tmp_path / "k.s"
    f.write_text("_ZN1x12build_kernelImEEv:\n" + LOAD + "\tv_mov_b32_e32 v12, v4\n\ts_endpgm\n")
    bad, n_loads, _ = ci.check(str(f), "part2_swwc_kernel")
    assert bad == [] and n_loads == 0
