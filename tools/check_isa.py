#!/usr/bin/env python3
"""Register budget of k_pileup2 from the compiler's assembly (no GPU needed): VGPRs, SGPRs, scratch bytes, and how many
scratch and SGPR-spill (v_writelane / v_readlane) operations sit from the piece loop on.  The kernel's scalar-load scheme
(brc_engine.hip, BRC_LD_REC / BRC_WAIT_REC) and its 7 waves per SIMD depend on that budget: 72 VGPRs, 88 SGPRs, scratch
only outside the loop.  Whether a build is also CORRECT is for the -m gpu parity tests to say (tools/experiments/README.md
has a build that fits 64 registers and is wrong).

    python tools/check_isa.py [extra hipcc flags, e.g. -DBRC_WAVES_PER_EU=8]
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bam_readcount_amd", "csrc", "brc_engine.hip")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-ffp-contract=off"] + sys.argv[1:] + [SRC, "-o", "-"]
    asm = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3brc9k_pileup2.*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    meta = {k: int(re.search(r"k_pileup2\S*\.%s, (\d+)" % k, asm).group(1)) for k in ("num_vgpr", "numbered_sgpr", "private_seg_size")}
    loop = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    tail = body[loop[-1]:] if loop else body
    count = lambda pred: sum(1 for l in tail if pred(l.strip()))
    print("k_pileup2: %(num_vgpr)d VGPRs, %(numbered_sgpr)d SGPRs, %(private_seg_size)d bytes of scratch per lane" % meta)
    print("from the piece loop on: %d VALU, %d SALU, %d scratch, %d v_writelane / v_readlane instructions (static counts)" % (
        count(lambda l: l.startswith("v_")), count(lambda l: l.startswith("s_")), count(lambda l: l.startswith("scratch_")),
        count(lambda l: l.startswith("v_writelane") or l.startswith("v_readlane"))))


if __name__ == "__main__":
    main()
