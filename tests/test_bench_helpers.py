"""bench.py's helpers, on the CPU: the -l site layout of --mode sites (the planner's virtual axis) gives, line by line, what
one oracle region per site gives; the usable-CPU count honours a cgroup quota."""
import os
import sys

import numpy as np

from bam_readcount_amd import capi
from conftest import ROOT

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_site_batch_lines_equal_one_region_per_site(sim_lib, oracle_lib):
    import bench
    import synthgen
    L = 60000
    ref, arrs = synthgen.generate(L, "wgs30x", seed=5, n_chunks=4)
    sites = np.sort(np.random.default_rng(3).integers(200, L - 200, 150))
    sub, vref, events, vbeg0 = bench.site_batch(np, capi, arrs, ref, sites)
    opts = dict(min_mapq=20, min_bq=13)
    eng = capi.Engine(sim_lib, text_only=True, **opts)
    eng.begin_region(0, 0, len(vref), vref); eng.push_reads(sub)
    eng.region_windows(vbeg0.astype(np.int32), (vbeg0 + 1).astype(np.int32))       # as bench.py does: only the lines' tiles are piled up
    eng.end_region()
    ends = capi.read_ends(arrs)
    oe = capi.Engine(oracle_lib, **opts)
    n_ev = 0
    for i, sp in enumerate(int(s) for s in sites):
        oe.begin_region(0, sp - 1, sp, ref); oe.push_reads(capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, sp - 2, sp)))
        oe.end_region(); want = oe.format_region("chrS"); oe.clear_indel_queue(); n_ev += oe.counts()[0]
        d = int(vbeg0[i]) + 1 - sp
        assert eng.format_window("chrS", sp - 1 + d, sp + d, d) == want, (i, sp)
        assert want.startswith(b"chrS\t%d\t" % sp)
    assert n_ev == events
    eng.close(); oe.close()


def test_effective_cpus_is_positive_and_bounded():
    import bench
    n = bench.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
