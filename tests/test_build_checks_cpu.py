"""CPU: the ISA-level build checks (tools/check_kloop_isa.py, run by tools/check_spills.py at the end of every `make`) on hand-made listings:
a scratch access between a kernel's first and last MFMA must be reported, one outside that span must not; a loop that moves its
accumulators between AGPRs and VGPRs around its MFMAs (round 4: every MX build of the generic GEMM tile did, 192 moves per 24 MFMAs) must be
reported, a tile loop that reads each accumulator once in a long epilogue must not."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_kloop_isa", os.path.join(ROOT, "tools", "check_kloop_isa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _listing(tmp_path, name, lines):
    p = tmp_path / (name + ".s")
    p.write_text("_Z%dfoov:\n" % len(name) + "\n".join("\t" + l if not l.endswith(":") else l for l in lines) + "\n.Lfunc_end0:\n")
    return str(p)


MFMA = "v_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]"


def test_scratch_inside_the_matrix_phase_is_reported(tmp_path):
    t = _tool()
    inside = _listing(tmp_path, "inside", ["s_load_dword s0, s[0:1], 0x0", MFMA, "scratch_load_dword v1, off, s0", MFMA, "s_endpgm"])
    outside = _listing(tmp_path, "outside", ["scratch_store_dword off, v1, s0", MFMA, MFMA, "scratch_load_dword v1, off, s0", "s_endpgm"])
    assert t.check(inside, quiet=True) == ["_Z6foov"]
    assert t.check(outside, quiet=True) == []


def test_accumulator_shuffle_in_a_k_loop_is_reported(tmp_path):
    t = _tool()
    shuffled = [".LBB0_1:"] + ["v_accvgpr_read_b32 v%d, a%d" % (i, i) for i in range(64)] + ["v_accvgpr_write_b32 a%d, v%d" % (i, i) for i in range(64)] \
        + [MFMA] * 24 + ["s_cbranch_scc1 .LBB0_1", "s_endpgm"]
    clean = [".LBB0_1:"] + [MFMA] * 24 + ["s_add_i32 s0, s0, 1", "s_cbranch_scc1 .LBB0_1"] + ["v_accvgpr_read_b32 v%d, a%d" % (i, i) for i in range(64)] + ["s_endpgm"]
    # a persistent tile loop: K loop inside, a long epilogue that reads every accumulator once - not a shuffle
    tile_loop = [".LBB0_1:", ".LBB0_2:"] + [MFMA] * 24 + ["s_cbranch_scc1 .LBB0_2"] + ["v_accvgpr_read_b32 v%d, a%d" % (i % 200, i % 64) for i in range(64)] \
        + ["v_add_f32 v1, v1, v2"] * 1000 + ["s_cbranch_scc1 .LBB0_1", "s_endpgm"]
    # not a loop: a block laid out behind s_endpgm that jumps back to a label in front of the (clean) K loop; the span holds the split-K store
    # path's accumulator reads, but nothing in it repeats
    outlined = [".LBB0_1:", ".LBB0_2:"] + [MFMA] * 16 + ["s_cbranch_scc1 .LBB0_2"] + ["v_accvgpr_read_b32 v%d, a%d" % (i, i) for i in range(64)] \
        + ["global_store_dword v[0:1], v2, off", "s_endpgm", ".LBB0_3:", "s_mov_b32 s0, 0", "s_branch .LBB0_1"]
    assert t.acc_shuffles(_listing(tmp_path, "outlined", outlined)) == []
    got = t.acc_shuffles(_listing(tmp_path, "shuffled", shuffled))
    assert len(got) == 1 and got[0][1:] == (24, 128)
    assert t.acc_shuffles(_listing(tmp_path, "clean", clean)) == []
    assert t.acc_shuffles(_listing(tmp_path, "tileloop", tile_loop)) == []


def test_fast_epilogue_key_lists_agree():
    """The device dispatch (gemm_kernels.h direct_epilogue_any: PB_FAST_CASE list) and the host-side PB_EPI_REPORT diagnostic (gemm.hip) name the
    launch kinds with a straight-line epilogue copy twice; a key added to one and not the other would make the report lie."""
    import re
    src = open(os.path.join(ROOT, "prisma_amd", "csrc", "gemm_kernels.h")).read()
    acts = dict(re.findall(r"(ACT_\w+) = (\d+)", open(os.path.join(ROOT, "prisma_amd", "csrc", "gemm.h")).read()))
    ev = lambda e: eval(e, {}, {k: int(v) for k, v in acts.items()})
    sw = src[src.index("switch (key) {"):src.index("#undef PB_FAST_CASE")]
    dev = sorted(ev(e) for e in re.findall(r"PB_FAST_CASE\(([^)]*)\)", sw))
    host_src = open(os.path.join(ROOT, "prisma_amd", "csrc", "gemm.hip")).read()
    lst = host_src[host_src.index("fast_keys[] = {") + len("fast_keys[] = {"):]
    lst = lst[:lst.index("};")]
    host = sorted(ev(e.strip()) for e in lst.replace("\n", " ").split(",") if e.strip())
    assert dev == host and len(dev) == 13, (dev, host)


def test_every_environment_switch_of_the_library_is_documented():
    """INTEGRATION.md's table of environment switches names every variable the HIP sources read (pb_env_int / getenv)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "prisma_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "prisma_amd", "csrc", "*.h")):
        names.update(re.findall(r'(?:pb_env_int|getenv)\("([A-Z][A-Z0-9_]*)"', open(f).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert len(names) > 20 and not missing, missing
