#!/usr/bin/env python3
"""Static instruction census of one kernel by source line (no GPU needed): compiles brc_engine.hip with line tables and
counts VALU / SALU / LDS / VMEM / SMEM instructions per (file, line) the instruction is attributed to (inlined code is
attributed to its innermost source line).  Usage: python tools/isa_lines.py k_annotate_groups [--top 40] [hipcc flags]"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bam_readcount_amd", "csrc", "brc_engine.hip")


def main():
    kern = sys.argv[1]; top = 40; extra = []
    a = sys.argv[2:]
    while a:
        if a[0] == "--top": top = int(a[1]); a = a[2:]
        else: extra.append(a.pop(0))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-ffp-contract=off", "-gline-tables-only"] + extra + [SRC, "-o", "-"]
    asm = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().split("\n")
    files = {}
    for l in asm:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m: files[int(m.group(1))] = os.path.basename(m.group(2))
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"\s*$', l)
        if m: files[int(m.group(1))] = os.path.basename(m.group(2))
    start = next(i for i, l in enumerate(asm) if re.match(r"^_ZN3brc\d+%s.*:" % kern, l))
    end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
    cur = ("?", 0); cnt = collections.defaultdict(lambda: collections.Counter())
    tot = collections.Counter()
    for l in asm[start:end]:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
        if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"): continue
        op = s.split()[0]
        if op.startswith("v_"): k = "VALU"
        elif op.startswith(("s_load", "s_buffer_load")): k = "SMEM"
        elif op.startswith("s_"): k = "SALU"
        elif op.startswith("ds_"): k = "LDS"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): k = "VMEM"
        else: k = "other"
        cnt[cur][k] += 1; tot[k] += 1
    print("%s: %s" % (kern, dict(tot)))
    rows = sorted(cnt.items(), key=lambda kv: -kv[1]["VALU"])[:top]
    for (f, ln), c in rows:
        print("%-22s %5d  VALU %4d  SALU %4d  LDS %3d  VMEM %3d  SMEM %3d" % (f, ln, c["VALU"], c["SALU"], c["LDS"], c["VMEM"], c["SMEM"]))


if __name__ == "__main__":
    main()
