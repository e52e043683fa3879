#!/usr/bin/env python3
"""isa_diff.py a.s b.s [--only PREFIX] — compares the device assembly of two builds kernel by kernel (comments, directives and
labels' addresses aside) and prints which kernels differ.  Used to prove that an instrumentation macro (BRC_CK outside the
checked build) or a new template parameter leaves the shipped instantiations' machine code unchanged:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S brc_engine.hip -o new.s
Exit code 1 when a compared kernel differs."""
import re
import sys


def kernels(path):
    out = {}; cur = None
    for l in open(path):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur = m.group(1); out[cur] = []; continue
        if cur is not None:
            if l.startswith('.Lfunc_end'):
                cur = None; continue
            t = l.strip()
            if not t or t.startswith(';') or t.startswith('.'):
                continue
            # (basic-block labels carry the function's ordinal in the file: a kernel added in front renumbers them)
            out[cur].append(re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r'\s*;.*$', '', t)))
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
    bad = 0
    for k in sorted(set(a) | set(b)):
        if only and only not in k:
            continue
        if k not in a or k not in b:
            print("ONLY-IN-%s %s" % ("A" if k in a else "B", k)); continue
        same = a[k] == b[k]
        print("%s %s (%d / %d instructions)" % ("same" if same else "DIFF", k, len(a[k]), len(b[k])))
        bad += 0 if same else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
