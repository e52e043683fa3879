#!/usr/bin/env python3
"""Static checks of k_pileup2's machine code (no GPU needed), run by bam_readcount_amd/csrc/Makefile after every build of
the engine and by tests/test_abi.py for builds at 6, 7 and 8 waves per SIMD.

1. Register budget: VGPRs, SGPRs, scratch bytes; scratch and SGPR-spill (v_writelane / v_readlane) operations from the
   piece loop on (printed; --max-vgpr / --max-scratch turn them into failures).

2. SOUNDNESS OF THE EARLY SCALAR LOADS.  k_pileup2 issues the scalar loads of a piece record two pieces ahead of its use
   (brc_engine.hip, BRC_LD_REC) and waits for them in a later inline-assembly statement (BRC_WAIT_REC); the record sets live
   in fixed scalar registers named in the constraints of both statements.  The compiler knows nothing about the time in
   between, so this script checks it on the code the compiler actually produced: it builds the control-flow graph of the
   kernel and walks it from every inline-assembly s_load_dword* to the first `s_waitcnt lgkmcnt(0)` on EVERY path; an
   instruction on the way that reads or writes one of the destination registers (a copy, a spill to a VGPR lane, a reuse
   as a temporary, another load into them) is an error.  Exit status 1 and a listing of the offending paths.

3. VALU-WRITTEN SGPR READ BY AN INLINE-ASSEMBLY VMEM INSTRUCTION.  On gfx9 a vector memory instruction that reads an SGPR which a
   VALU instruction wrote (v_readlane / v_readfirstlane — e.g. the restore of a spilled scalar pair — or a v_cmp into a pair) needs
   five wait states in between.  The compiler inserts them for its own instructions; it does not look into an assembly statement.
   k_pileup2 stores its planes and loads wide words with scalar-base VMEM instructions written in assembly: this walks the control-flow
   graph BACKWARDS from each of them and fails if a VALU write of one of its scalar operands can be fewer than five wait states away
   (s_nop N counts N + 1).  Found in round 5 on the bounds-checked build, whose checked plane bases come out of v_readfirstlane: the
   stores went to half-updated addresses.

    python tools/check_isa.py [--quiet] [--max-vgpr N] [--max-scratch BYTES] [extra hipcc flags, e.g. -DBRC_WAVES_PER_EU=8]
    python tools/check_isa.py --asm FILE.s [--arch gfx950]     the assembly a build left behind (hipcc -save-temps=obj: the
                                                               Makefile checks the very compile whose object it ships)
Without --asm the script compiles the source itself with $HIPCC (default /opt/rocm/bin/hipcc), --arch and the product's flags.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bam_readcount_amd", "csrc", "brc_engine.hip")


def sregs(text):
    """scalar registers named in an operand string"""
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def parse(body):
    """instructions [(text, in_asm)], label -> instruction index"""
    ins = []; labels = {}; in_asm = False
    for l in body:
        s = l.split(";")[0].strip() if not l.strip().startswith(";;#") else l.strip()
        if s.startswith(";;#ASMSTART"): in_asm = True; continue
        if s.startswith(";;#ASMEND"): in_asm = False; continue
        if not s or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m: labels[m.group(1)] = len(ins)
            continue
        if s.endswith(":"):
            continue
        ins.append((s, in_asm))
    return ins, labels


def successors(ins, labels, i):
    s = ins[i][0]; op = s.split()[0]
    if op in ("s_endpgm",):
        return []
    if op == "s_branch":
        return [labels[s.split()[1]]]
    if op.startswith("s_cbranch"):
        return [labels[s.split()[1]]] + ([i + 1] if i + 1 < len(ins) else [])
    return [i + 1] if i + 1 < len(ins) else []


def waits_lgkm0(s):
    return s.startswith("s_waitcnt") and re.search(r"lgkmcnt\(0\)", s) is not None


def check_loads(ins, labels):
    errors = []; nloads = 0
    for i, (s, in_asm) in enumerate(ins):
        if not (in_asm and s.startswith("s_load_dword")):
            continue
        nloads += 1
        dest = sregs(s.split(",")[0])
        # the statement's own companion loads (same asm block) write other registers; start behind this instruction
        seen = set(); stack = [(j, (i,)) for j in successors(ins, labels, i)]
        while stack:
            j, path = stack.pop()
            if j in seen:
                continue
            seen.add(j)
            t = ins[j][0]
            if waits_lgkm0(t):
                continue
            ops = t.split(None, 1)[1] if " " in t else ""
            if t.split()[0] == "s_endpgm":
                errors.append("load at #%d `%s`: a path reaches s_endpgm without s_waitcnt lgkmcnt(0)" % (i, s)); continue
            hit = dest & sregs(ops)
            if hit and not t.startswith(("s_branch", "s_cbranch")):
                errors.append("load at #%d `%s`: register(s) %s touched before the wait by #%d `%s`" % (i, s, sorted(hit), j, t))
                continue
            for k in successors(ins, labels, j):
                stack.append((k, path))
    return nloads, errors


VALU_SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32", "v_cmp", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32")


def check_vmem_sgpr_hazard(ins, labels):
    """inline-assembly global_* / buffer_* / flat_* / scratch_* instructions whose scalar operands a VALU instruction may have written
    fewer than five wait states earlier, on any path"""
    preds = {}
    for i in range(len(ins)):
        for j in successors(ins, labels, i):
            preds.setdefault(j, []).append(i)
    errors = []; n = 0
    for i, (s, in_asm) in enumerate(ins):
        if not (in_asm and s.split()[0].startswith(("global_", "buffer_", "flat_", "scratch_"))):
            continue
        n += 1
        used = sregs(s.split(None, 1)[1] if " " in s else "")
        if not used:
            continue
        # wait states already provided by s_nop instructions of the same statement in front of it are counted by the walk
        stack = [(j, 0) for j in preds.get(i, [])]; seen = {}
        while stack:
            j, w = stack.pop()
            if w >= 5 or seen.get(j, 99) <= w:
                continue
            seen[j] = w
            t = ins[j][0]; op = t.split()[0]
            m = re.match(r"s_nop (\d+)", t)
            if m:
                w2 = w + int(m.group(1)) + 1
            else:
                if op.startswith(VALU_SGPR_WRITERS):
                    dest = sregs(t.split(None, 1)[1].split(",")[0]) if " " in t else set()
                    if op.startswith(("v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")):
                        dest = sregs(t.split(None, 1)[1].split(",")[1]) if t.count(",") >= 1 else set()      # (the carry-out pair is the second operand)
                    if dest & used:
                        errors.append("`%s` (#%d) reads %s, written by `%s` (#%d) only %d wait state(s) before" % (s, i, sorted(dest & used), t, j, w))
                        continue
                w2 = w + 1
            for k in preds.get(j, []):
                stack.append((k, w2))
    return n, errors


def main():
    args = sys.argv[1:]; quiet = False; max_vgpr = None; max_scratch = None; extra = []; asm_file = None; arch = "gfx950"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    while args:
        a = args.pop(0)
        if a == "--quiet": quiet = True
        elif a == "--max-vgpr": max_vgpr = int(args.pop(0))
        elif a == "--max-scratch": max_scratch = int(args.pop(0))
        elif a == "--asm": asm_file = args.pop(0)
        elif a == "--arch": arch = args.pop(0)
        elif a == "--hipcc": hipcc = args.pop(0)
        else: extra.append(a)
    if asm_file:
        asm = open(asm_file).read()
        m = re.search(r'\.amdgcn_target\s+"[^"]*--(gfx[0-9a-f]+)', asm)
        if not m or m.group(1) != arch:
            print("check_isa: %s was built for %s, not for %s" % (asm_file, m.group(1) if m else "an unknown target", arch)); sys.exit(1)
    else:
        cmd = [hipcc, "--offload-arch=" + arch, "--cuda-device-only", "-S", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"] + extra + [SRC, "-o", "-"]
        asm = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    lines = asm.split("\n")
    # every instantiation of the kernel (k_pileup2<false>: the common one; k_pileup2<true>: with brc_region_windows in force)
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3brc9k_pileup2\S*:", l)]
    if not starts:
        print("check_isa: no k_pileup2 in the assembly"); sys.exit(1)
    rc = 0
    for start in starts:
        sym = re.match(r"^(_ZN3brc9k_pileup2[^\s:]*):", lines[start]).group(1)
        tag = "k_pileup2<true>" if "ILb1E" in sym else "k_pileup2<false>" if "ILb0E" in sym else "k_pileup2"
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[start + 1:end]
        meta = {k: int(re.search(r"%s\.%s, (\d+)" % (re.escape(sym), k), asm).group(1)) for k in ("num_vgpr", "numbered_sgpr", "private_seg_size")}
        loop = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
        tail = body[loop[-1]:] if loop else body
        count = lambda pred: sum(1 for l in tail if pred(l.strip()))
        if not quiet:
            print("%s: %d VGPRs, %d SGPRs, %d bytes of scratch per lane" % (tag, meta["num_vgpr"], meta["numbered_sgpr"], meta["private_seg_size"]))
            print("from the piece loop on: %d VALU, %d SALU, %d scratch, %d v_writelane / v_readlane instructions (static counts)" % (
                count(lambda l: l.startswith("v_")), count(lambda l: l.startswith("s_")), count(lambda l: l.startswith("scratch_")),
                count(lambda l: l.startswith("v_writelane") or l.startswith("v_readlane"))))
        ins, labels = parse(body)
        nloads, errors = check_loads(ins, labels)
        if nloads < 6:
            print("check_isa: %s: expected the read loop's inline-assembly scalar loads, found %d" % (tag, nloads)); rc = 1
        if errors:
            print("check_isa: UNSOUND early scalar loads in %s (%d):" % (tag, len(errors)))
            for e in errors[:20]:
                print("  " + e)
            rc = 1
        elif not quiet:
            print("early scalar loads: %d inline-assembly loads, no instruction touches their registers before the wait on any path" % nloads)
        nvm, herr = check_vmem_sgpr_hazard(ins, labels)
        if herr:
            print("check_isa: %s: VALU-written SGPR read by an inline-assembly VMEM instruction without five wait states (%d):" % (tag, len(herr)))
            for e in herr[:20]:
                print("  " + e)
            rc = 1
        elif not quiet:
            print("inline-assembly VMEM instructions: %d, none reads an SGPR a VALU instruction wrote within five wait states" % nvm)
        if max_vgpr is not None and meta["num_vgpr"] > max_vgpr:
            print("check_isa: %s: %d VGPRs > %d" % (tag, meta["num_vgpr"], max_vgpr)); rc = 1
        if max_scratch is not None and meta["private_seg_size"] > max_scratch:
            print("check_isa: %s: %d bytes of scratch > %d" % (tag, meta["private_seg_size"], max_scratch)); rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
