"""Shared comparison helpers: run the same region through two libraries exporting the brc C-ABI and compare
text (byte-exact), integer planes (bit-exact) and float planes (bit-exact; the 1e-6 relative tolerance of the
north_star is the fallback assertion message only — order-preserving fp32 sums make them identical)."""
import numpy as np

from bam_readcount_amd import capi


def run_engine(lib, arrs, regions, tid=0, chrom="chrS", ref=None, lib_names=(), clear_queue=True, **opts):
    eng = capi.Engine(lib, lib_names=lib_names, **opts)
    try:
        return capi.run_regions(eng, arrs, regions, tid, chrom, ref, clear_queue=clear_queue)
    finally:
        eng.close()


def assert_results_equal(ra, rb, what=""):
    assert (ra.pos0, ra.n_pos, ra.n_lib) == (rb.pos0, rb.n_pos, rb.n_lib), what
    if ra.unavail is not None or rb.unavail is not None:
        np.testing.assert_array_equal(ra.unavail, rb.unavail, err_msg=what + " unavail")
    np.testing.assert_array_equal(ra.ncol, rb.ncol, err_msg=what + " ncol")
    np.testing.assert_array_equal(ra.depth, rb.depth, err_msg=what + " depth")
    np.testing.assert_array_equal(ra.istat, rb.istat, err_msg=what + " istat")
    fa, fb = ra.fstat.view(np.uint32), rb.fstat.view(np.uint32)
    if not np.array_equal(fa, fb):
        bad = np.argwhere(fa != fb)
        rel = np.abs(ra.fstat - rb.fstat) / np.maximum(np.abs(rb.fstat), 1e-30)
        raise AssertionError("%s fstat differs at %d entries (max rel %.3g), first %s: %r vs %r" % (
            what, len(bad), float(np.nanmax(rel)), bad[0], ra.fstat[tuple(bad[0])], rb.fstat[tuple(bad[0])]))
    assert ra.refbase == rb.refbase, what + " refbase"
    assert len(ra.indels) == len(rb.indels), what + " indel count %d vs %d" % (len(ra.indels), len(rb.indels))
    for x, y in zip(ra.indels, rb.indels):
        assert (x["pos"], x["lib"], x["len"], x["allele"]) == (y["pos"], y["lib"], y["len"], y["allele"]), what
        np.testing.assert_array_equal(x["i"], y["i"], err_msg=what + " indel ints %r" % (x["allele"],))
        np.testing.assert_array_equal(x["f"].view(np.uint32), y["f"].view(np.uint32), err_msg=what + " indel floats")
    assert ra.n_events == rb.n_events, what + " n_events"


def compare_libs(lib_a, lib_b, arrs, regions, check_warn=True, **kw):
    ta, resa = run_engine(lib_a, arrs, regions, **kw)
    tb, resb = run_engine(lib_b, arrs, regions, **kw)
    for i, (x, y) in enumerate(zip(resa, resb)):
        assert_results_equal(x, y, "region %d %r" % (i, regions[i]))
        if check_warn:
            assert x.warn[:3] == y.warn[:3], "warn counts region %d: %r vs %r" % (i, x.warn, y.warn)
    assert ta == tb, "text differs"
    return ta, resa


def slice_result(r, lo, hi):
    """planes of the positions [lo, hi) of a RegionResult (positions outside its planes: no reads, zeros), indel list likewise"""
    n = hi - lo
    def cut(a):
        out = np.zeros(a.shape[:-1] + (n,), a.dtype)
        a0, a1 = max(lo, r.pos0), min(hi, r.pos0 + r.n_pos)
        if a1 > a0:
            out[..., a0 - lo:a1 - lo] = a[..., a0 - r.pos0:a1 - r.pos0]
        return out
    ind = [(x["pos"], x["lib"], x["len"], x["allele"], x["i"].tobytes(), x["f"].tobytes()) for x in r.indels if lo <= x["pos"] < hi]
    return cut(r.ncol), cut(r.depth), cut(r.istat), cut(r.fstat).view(np.uint32), ind
