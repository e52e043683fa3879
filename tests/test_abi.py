"""CPU-side checks of the boundary: the product library loads and exports every symbol include/brc.h declares;
without a GPU it refuses to create an engine (no CPU fallback)."""
import ctypes as C
import os
import sys
import re
import subprocess

import pytest

from bam_readcount_amd import capi
from conftest import ROOT


def declared_symbols():
    h = open(os.path.join(ROOT, "include", "brc.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(brc_[a-z_0-9]+)\s*\(", h)))


def test_header_symbols_listed_in_binding():
    assert set(declared_symbols()) == set(capi.EXPORTS)


def test_product_library_builds_and_exports_abi():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bam_readcount_amd", "csrc")])
    lib = capi.load_product()
    for s in declared_symbols():
        assert hasattr(lib.lib, s), s
    assert lib.kind() == "hip-gfx950"
    names = lib.kernel_names()
    assert "k_pileup" in names and "k_annotate" in names


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    lib = capi.load_product()
    with pytest.raises(capi.BrcError) as ei:
        capi.Engine(lib)
    assert "-2" in str(ei.value) or "no HIP device" in str(ei.value)


def test_oracle_is_not_linked_into_product():
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.PRODUCT_LIB]).decode()
    assert "oracle" not in out.lower()
    ldd = subprocess.check_output(["ldd", capi.PRODUCT_LIB]).decode()
    assert "oracle" not in ldd and "brc_sim" not in ldd


def test_product_contains_no_ablation_knobs():
    """The timing-only ablations of the two big kernels (BRC_PILEUP_VARIANT / BRC_ANN_VARIANT: parts of the kernels switched
    off, wrong results) and the other profiling knobs exist only in experiment builds (-DBRC_EXP_KNOBS, tools/build_variant.sh):
    the shipped library does not even contain their names, so no inherited environment variable can change what it computes
    (the GPU half: tests/test_gpu_parity.py::test_product_ignores_ablation_environment)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bam_readcount_amd", "csrc")])
    blob = open(capi.PRODUCT_LIB, "rb").read()
    for knob in (b"BRC_PILEUP_VARIANT", b"BRC_ANN_VARIANT", b"BRC_PILEUP_LDS_PAD", b"BRC_INDEL_OVERLAP"):
        assert knob not in blob, knob
    # ... and the TEST knobs (result-neutral, but they choose the device path: an inherited BRC_NO_TABLE=1 would turn the shipped kernel
    # into its slow path without a word) exist only in libbrc_hip_testknobs.so — the same objects linked with brc_knobs.cpp -DBRC_TEST_KNOBS
    knobs = (b"BRC_NO_TABLE", b"BRC_FLUSH_K", b"BRC_PACK_LIM", b"BRC_FORCE_DOM", b"BRC_IBUCKET_SHIFT", b"BRC_XEV_CAP", b"BRC_DEVICE_TEXT_LIMIT", b"BRC_FORMAT_THREADS", b"BRC_FORMAT_CHUNK")
    for knob in knobs:
        assert knob not in blob, knob
    tk = open(os.path.join(os.path.dirname(capi.PRODUCT_LIB), "libbrc_hip_testknobs.so"), "rb").read()
    for knob in knobs:
        assert knob in tk, knob
    for name in ("brc_engine.hip", "brc_host.cpp"):
        text = open(os.path.join(ROOT, "bam_readcount_amd", "csrc", name)).read()
        for knob in knobs:
            assert ('getenv("%s")' % knob.decode()) not in text, (name, knob)
    src = open(os.path.join(ROOT, "bam_readcount_amd", "csrc", "brc_engine.hip")).read()
    assert "c.variant ==" not in src.replace("#define BRC_PVAR(n) (c.variant == (n))", "")
    assert "c.ann_variant ==" not in src.replace("#define BRC_AVAR(n) (c.ann_variant == (n))", "")


def test_kernel_register_budget():
    """k_pileup2's design point is read off the built library (no GPU): at most 72 VGPRs = 7 waves per SIMD and no scratch."""
    import re, struct
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("no llvm-readelf")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bam_readcount_amd", "csrc")])
    data = open(capi.PRODUCT_LIB, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0
    n = struct.unpack_from("<Q", data, i + 24)[0]; o = i + 32; blob = None
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, o); o += 24
        triple = data[o:o + tl].decode(); o += tl
        if "gfx950" in triple:
            blob = data[i + off:i + off + size]
    assert blob is not None, "no gfx950 code object in the product library"
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(blob); f.flush()
        notes = subprocess.check_output([readelf, "--notes", f.name]).decode()
    kern = {}
    for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", notes, re.S):
        body = m.group(2)
        kern[m.group(1)] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1)) for k in ("private_segment_fixed_size", "sgpr_count", "vgpr_count")}
    piles = [v for k, v in kern.items() if "k_pileup2" in k]           # every instantiation
    anns = [v for k, v in kern.items() if "k_annotate_groups" in k]
    assert len(piles) == 4 and len(anns) == 4, sorted(kern)      # both kernels: (with / without windows, one stream / per library) x (16- / 12-bit narrow packed fields: brc_core.h choose_pack)
    # k_pileup2: 7 waves per SIMD; a few values may be spilled around its rare paths (the drain of queued third-allele / huge-integer
    # entries between half-batches), never in a step: tools/check_isa.py counts the scratch instructions from the piece loop on
    for pile in piles:
        assert pile["vgpr_count"] <= 72 and pile["private_segment_fixed_size"] <= 16, pile
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    for m in re.finditer(r"from the piece loop on: \d+ VALU, \d+ SALU, (\d+) scratch", r.stdout.decode()):
        assert int(m.group(1)) <= 8, r.stdout.decode()
    # K1: 6 waves per SIMD (80 VGPRs); three values that live from its first to its last phase are spilled once around the
    # per-base pass (no scratch instruction inside a loop: checked on the assembly below)
    # (the per-library instantiation stays at the allocator's five waves, without scratch)
    # (the budget is that of the instantiations every short-read region runs, Li16E; the 12 + 20-bit layout of long-read regions has
    # the shifts of its packed fields as other immediates and may be allocated differently: bounded, not pinned)
    ann_by = {("ILb1E" in k): v for k, v in kern.items() if "k_annotate_groups" in k and "Li16E" in k}
    assert len(ann_by) == 2, sorted(kern)
    assert ann_by[True]["private_segment_fixed_size"] <= 64 and ann_by[True]["vgpr_count"] <= 80, ann_by[True]
    assert ann_by[False]["private_segment_fixed_size"] == 0 and ann_by[False]["vgpr_count"] <= 96, ann_by[False]
    for k, v in kern.items():
        if "k_annotate_groups" in k and "Li12E" in k:
            assert v["vgpr_count"] <= 96 and v["private_segment_fixed_size"] <= 128, (k, v)
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                          os.path.join(ROOT, "bam_readcount_amd", "csrc", "brc_engine.hip"), "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode().split("\n")
    starts = [i for i, l in enumerate(asm) if re.match(r"^_ZN3brc17k_annotate_groups\S*Li16E\S*:", l)]
    assert len(starts) == 2
    for st in starts:
        in_loop = False; n_scratch = 0
        for l in asm[st:next(i for i in range(st, len(asm)) if asm[i].startswith(".Lfunc_end"))]:
            if re.match(r"^\.LBB|^; %bb", l):
                in_loop = "in Loop:" in l or "Loop Header" in l
            if l.strip().startswith("scratch_"):
                n_scratch += 1
                assert not in_loop, "K1 spills inside a loop: " + l
        assert n_scratch <= 12


@pytest.mark.parametrize("waves", [7, 6])
def test_early_scalar_loads_are_sound_in_the_machine_code(waves):
    """k_pileup2 issues the scalar loads of a piece record two pieces before the inline-assembly wait that hands the
    registers over (brc_engine.hip: BRC_LD_REC / BRC_WAIT_REC, fixed scalar registers).  tools/check_isa.py walks the
    control-flow graph of the compiled kernel from every such load to the first s_waitcnt lgkmcnt(0) on every path and
    fails if an instruction on the way reads or writes a register in flight.  The product's build (7 waves per SIMD; the
    Makefile runs the same check before it compiles the object) and a 6-wave build must pass."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), "--max-vgpr", {7: "72", 6: "84"}[waves], "--max-scratch", "16", "-DBRC_WAVES_PER_EU=%d" % waves],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    assert b"no instruction touches their registers before the wait on any path" in r.stdout


def test_an_unsound_build_is_refused():
    """At 8 waves per SIMD (64 VGPRs) this compiler copies and spills the record registers while their loads are in flight
    — the build whose planes differed in round 2.  The checker must say so (and the Makefile then refuses to build the
    object); if a future compiler gets it right, the check simply passes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py"), "-DBRC_WAVES_PER_EU=8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert (r.returncode == 1 and b"UNSOUND early scalar loads" in r.stdout) or (r.returncode == 0 and b"no instruction touches" in r.stdout), r.stdout.decode()
    if r.returncode == 1:
        m = subprocess.run(["make", "-s", "-B", "-n", "-C", os.path.join(ROOT, "bam_readcount_amd", "csrc"), "brc_engine.o", "EXTRA=-DBRC_WAVES_PER_EU=8"], stdout=subprocess.PIPE)
        assert b"check_isa.py" in m.stdout          # the gate is part of the object's recipe


def test_isa_checker_follows_every_path():
    """The checker itself, on hand-written instruction lists: a register of an early load touched on the fall-through path, on
    a branch target only, inside a loop body, and not at all."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa

    def errors(text):
        ins, labels = check_isa.parse(text.strip("\n").split("\n"))
        return check_isa.check_loads(ins, labels)

    sound = """
	;;#ASMSTART
	s_load_dwordx8 s[56:63], s[4:5], 0x0
	s_load_dwordx2 s[64:65], s[4:5], 0x20
	;;#ASMEND
	s_add_u32 s4, s4, 48
	v_add_u32_e32 v1, s70, v2
	s_cbranch_scc1 .LBB9_2
	v_mov_b32_e32 v3, s71
.LBB9_2:
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_add_u32_e32 v1, s56, v2
	s_endpgm
"""
    n, errs = errors(sound)
    assert n == 2 and errs == []
    # a copy of a register in flight on the fall-through path
    n, errs = errors(sound.replace("v_mov_b32_e32 v3, s71", "s_mov_b64 s[36:37], s[58:59]"))
    assert len(errs) == 1 and "s_mov_b64" in errs[0] and "[58, 59]" in errs[0]
    # ... only behind a taken branch (an out-of-line block that rejoins before the wait)
    far = sound.replace("s_cbranch_scc1 .LBB9_2", "s_cbranch_scc1 .LBB9_7").replace("	s_endpgm", "	s_endpgm\n.LBB9_7:\n	v_writelane_b32 v62, s64, 3\n	s_branch .LBB9_2")
    n, errs = errors(far)
    assert len(errs) == 1 and "v_writelane_b32" in errs[0]
    # ... a second load into the same set before the first one's wait
    n, errs = errors(sound.replace("s_add_u32 s4, s4, 48", ";;#ASMSTART\n	s_load_dwordx8 s[56:63], s[4:5], 0x30\n	;;#ASMEND"))
    assert any("s_load_dwordx8 s[56:63], s[4:5], 0x30" in e for e in errs)
    # ... a path that never waits
    n, errs = errors(sound.replace("	;;#ASMSTART\n	s_waitcnt lgkmcnt(0)\n	;;#ASMEND\n", ""))
    assert any("s_endpgm" in e or "touched" in e for e in errs)


def test_committed_pmc_passes_belong_to_the_built_kernels():
    """bench.py reports the counters of profiles/r06_traffic.json / r06_pmc_summary.json (separate rocprofv3 --pmc passes) only when they
    were measured on the kernel object the loaded library carries (capi.kernel_object_hash: sha256 of its .hip_fatbin section).  A kernel
    edit without new passes would make the bench line drop its traffic fields: this test says so at commit time."""
    import json
    from bam_readcount_amd import capi
    kobj = capi.kernel_object_hash()
    assert kobj and len(kobj) == 16
    assert capi.kernel_object_hash(os.path.join(ROOT, "bam_readcount_amd", "csrc", "libbrc_hip_testknobs.so")) == kobj      # the same objects
    assert capi.kernel_object_hash(os.path.join(ROOT, "oracle", "libbrc_oracle.so")) is None
    for name in ("r06_traffic.json", "r06_pmc_summary.json"):
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        for cfg in ("wgs30x", "tumor200x"):
            assert j[cfg]["kernel_object_sha256_16"] == kobj, "%s [%s] was measured on kernel object %s, the build is %s: repeat tools/gpu_r6_profile.sh" % (name, cfg, j[cfg]["kernel_object_sha256_16"], kobj)
