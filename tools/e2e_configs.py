#!/usr/bin/env python3
"""e2e_configs.py — BASELINE configs 4 and 5 end to end through the drop-in command line, on real multi-contig BAM + BAI files.

Bench / test infrastructure.  Two legs, each printing one JSON object (bench.py embeds them as `e2e_sites` / `e2e_tumor`):

  --leg sites   config 4: a synthetic 30x genome of --contigs contigs (the scale against BASELINE's 24 contigs / 3.1 Gbp is stated
                in the output), one BAM + BAI + FASTA; a -l file of single-base sites over ALL contigs at BASELINE's spacing
                (3.1 Gbp / 100 000 = one site per 31 kb), in file order, with a few duplicate lines, lines repeated out of order,
                and multi-base lines overlapping their neighbours.  `bam-readcount -w0 -q20 -b13 -f g.fa -l sites g.bam > /dev/null`
                is timed (BRC_CLI_TIMING=1 stage account), run once more into a file, and checked:
                  * >= --check-lines site-list lines (blocks of consecutive lines spread over the whole list) through the
                    reference's OWN main() (oracle/_ref/bam-readcount-ref, bamreadcount.cpp:574-607 — one samfetch + pileup per
                    line) and through the drop-in: byte-identical; and the drop-in's lines for that sub-list are found, in order,
                    in its output for the full list;
                  * the line count of the full run == the covered positions of every line (what the oracle prints a line for).
  --leg tumor   config 5: 200x, 4 libraries / 8 read groups (two per library), 10 % indel reads, `-p -i`, one region of
                --contig-mbp (BASELINE: 50 Mbp over 8 GPUs = 6.25 Mbp per GPU) inside a multi-contig BAM.  Timed to /dev/null; the
                first --check-mbp of the region are cut into pieces, each piece run through the reference's own main() on all
                cores, and the concatenation must equal the drop-in's text for the same region byte for byte; full line count as above.

The generator (tools/synth_gen.c) is SURVEY.md 8d's data model; every contig has its own seed."""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import synthgen  # noqa: E402

CLI = os.path.join(ROOT, "bam_readcount_amd", "csrc", "bam-readcount")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "bam-readcount-ref")
GENOME_BP, GENOME_SITES = 3.1e9, 100000          # BASELINE config 4


def read_ends(a):
    """bam_endpos of every read (M D N = X advance the reference)"""
    cig = a["cigar"]; n = len(a["pos"])
    op = cig & 15; ln = (cig >> 4).astype(np.int64)
    adv = np.where((op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8), ln, 0)
    cs = np.concatenate([[0], np.cumsum(adv)])
    off = a["cigar_off"].astype(np.int64); nc = a["n_cigar"].astype(np.int64)
    span = cs[off + nc] - cs[off]
    return a["pos"].astype(np.int64) + np.where(nc > 0, span, 1)[:n]


def coverage(a, length):
    """reads covering every position (all reads of the generator are mapped, primary, unfiltered at push)"""
    pos = a["pos"].astype(np.int64).clip(0, length); end = read_ends(a).clip(0, length)
    d = np.bincount(pos, minlength=length + 1).astype(np.int64) - np.bincount(end, minlength=length + 1).astype(np.int64)
    return np.cumsum(d)[:length].astype(np.int32)


def make_genome(d, n_contigs, contig_len, config, seed0, want_cov):
    """g.bam + g.bam.bai + g.fa + g.fa.fai in d; returns contigs and, per contig, what want_cov(tid, coverage array) returns"""
    cfg = synthgen.CONFIGS[config]
    n_libs = cfg["n_libs"]; rgs = 2 if n_libs > 1 else 1
    contigs = [("chr%d" % (i + 1), int(contig_len)) for i in range(n_contigs)]
    w = synthgen.BamWriter(os.path.join(d, "g.bam"), contigs, n_libs=n_libs, rgs_per_lib=rgs)
    refs = []; out = []; n_reads = 0
    for t, (nm, ln) in enumerate(contigs):
        ref, a = synthgen.generate(ln, config, seed=seed0 + 17 * t)
        w.add(t, a); refs.append((nm, ref)); n_reads += len(a["pos"])
        out.append(want_cov(t, coverage(a, ln)))
        del a
    w.close()
    synthgen.write_fasta(os.path.join(d, "g.fa"), refs)
    return contigs, out, n_reads, n_libs, rgs


REF_ENV = None      # environment of the reference-compiled main(): see --ref-reader


def run(cmd, cwd, stdout, env=None, timeout=3600):
    t0 = time.perf_counter()
    if env is None and cmd and os.path.basename(cmd[0]) == os.path.basename(REF_CLI):
        env = REF_ENV
    p = subprocess.run(cmd, cwd=cwd, stdout=stdout, stderr=subprocess.PIPE, env=env, timeout=timeout)
    return time.perf_counter() - t0, p.returncode, p.stderr.decode(errors="replace")


def timed_to_devnull(cmd, cwd, reps):
    """best of `reps` runs to /dev/null with the CLI's own stage account; returns (seconds, stage lines of the best run)"""
    env = dict(os.environ, BRC_CLI_TIMING="1")
    best = None; stages = None
    for _ in range(reps):
        with open(os.devnull, "wb") as dn:
            t, rc, err = run(cmd, cwd, dn, env)
        if rc != 0:
            raise SystemExit("command failed (%d): %s\n%s" % (rc, " ".join(cmd), err[-2000:]))
        if best is None or t < best:
            best, stages = t, [l for l in err.splitlines() if l.startswith(("startup:", "timing:", "sites:", "engine timing", "device buffers"))]
    # once more with the engine's own account (BRC_ENGINE_TIMING=1 takes the orderly exit path: not the timed run)
    with open(os.devnull, "wb") as dn:
        _, rc, err = run(cmd, cwd, dn, dict(env, BRC_ENGINE_TIMING="1"))
    if rc == 0:
        stages = stages + [l for l in err.splitlines() if l.startswith(("engine timing", "device buffers"))]
    return best, stages


def digest_of_stdout(cmd, cwd, env=None):
    """xxh3-128 digest, byte and line count of a command's stdout, streamed (config 5's text is gigabytes)"""
    import xxhash
    h = xxhash.xxh3_128(); nb = 0; nl = 0
    p = subprocess.Popen(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, bufsize=0)
    import threading
    err = []
    t = threading.Thread(target=lambda: err.append(p.stderr.read())); t.start()
    while True:
        b = p.stdout.read(1 << 24)
        if not b:
            break
        h.update(b); nb += len(b); nl += b.count(b"\n")
    p.wait(); t.join()
    return p.returncode, h.hexdigest(), nb, nl, (err[0] if err else b"").decode(errors="replace")


def sharded(args, d, cmd, events, one_seconds):
    """The same command as N processes, one per GPU (--brc-ranks N: contiguous event-weighted slices of the work list in file order,
    text in rank order): best of --reps to /dev/null with the coordinator's account of every rank, and the whole output — every rank's
    slice — compared with the one-process output byte for byte (digests, streamed)."""
    if args.ranks < 2:
        return None
    env = dict(os.environ, BRC_CLI_TIMING="1")
    if args.rank_devices:
        env["BRC_DEVICES"] = args.rank_devices
    rcmd = [cmd[0], "--brc-ranks", str(args.ranks)] + cmd[1:]
    best = None; acct = None
    for _ in range(args.reps):
        with open(os.devnull, "wb") as dn:
            t, rc, err = run(rcmd, d, dn, env)
        if rc != 0:
            raise SystemExit("command failed (%d): %s\n%s" % (rc, " ".join(rcmd), err[-2000:]))
        if best is None or t < best:
            best = t; acct = [l for l in err.splitlines() if l.startswith(("ranks:", "rank "))]
    rc1, h1, nb1, nl1, e1 = digest_of_stdout(cmd, d)
    rcn, hn, nbn, nln, en = digest_of_stdout(rcmd, d, env)
    assert rc1 == 0 and rcn == 0, (e1[-1000:], en[-1000:])
    assert (h1, nb1, nl1) == (hn, nbn, nln), "the output of %d ranks differs from the one-process output (%d vs %d bytes, %d vs %d lines)" % (args.ranks, nbn, nb1, nln, nl1)
    per_rank = []
    for l in acct:
        if l.startswith("ranks:"):
            import re
            per_rank = [{"rank": int(a), "seconds": float(b), "text_out_after_s": float(c)} for a, b, c in re.findall(r"rank (\d+): ([0-9.]+) s \(its text was out after ([0-9.]+) s\)", l)]
    shares = {}
    for l in acct:
        if l.startswith("rank ") and " of the estimated work" in l:
            import re
            m = re.match(r"rank (\d+) of \d+: atoms \[(\d+), (\d+)\) of (\d+), ([0-9.]+) of the estimated work \(([^)]*)\)", l)
            if m:
                shares[int(m.group(1))] = {"atoms": [int(m.group(2)), int(m.group(3))], "of": int(m.group(4)), "share_of_estimated_work": float(m.group(5)), "weights": m.group(6)}
    for pr in per_rank:
        pr.update(shares.get(pr["rank"], {}))
    return {"ranks": args.ranks, "devices": args.rank_devices or ",".join(str(i) for i in range(args.ranks)), "seconds": round(best, 3), "value": round(events / best, 1), "unit": "pileup base-events/s",
            "one_process_seconds": round(one_seconds, 3), "speedup_vs_one_process": round(one_seconds / best, 3), "efficiency": round(one_seconds / best / args.ranks, 3),
            "per_rank": per_rank, "whole_output_byte_identical_to_one_process": True, "output_bytes": nbn, "output_lines": nln, "output_xxh3_128": hn,
            "what": "the same command as %d processes (--brc-ranks %d, GPUs %s): one rank per GPU started before HIP initialisation, contiguous slices of the work list in file order weighted by the "
                    "index's file offsets, text written in rank order; strong scaling end to end (decode, PCIe, kernels, text) against the one-process run above"
                    % (args.ranks, args.ranks, args.rank_devices or "0..%d" % (args.ranks - 1))}


def is_subsequence(sub_lines, full_lines):
    it = iter(full_lines)
    return all(any(x == y for y in it) for x in sub_lines)


def leg_sites(args, d):
    rng = np.random.default_rng(3)
    L = int(args.contig_mbp * 1e6); total = L * args.contigs
    n_sites = args.sites if args.sites > 0 else max(1, int(round(total / (GENOME_BP / GENOME_SITES))))
    # sites U over the genome, sorted per contig, as 1-based positions
    g = np.sort(rng.integers(0, total, n_sites))
    tid = g // L; p1 = (g % L).clip(300, L - 300) + 1
    lines = []                                     # (tid, beg1, end1)
    for k, (t, p) in enumerate(zip(tid.tolist(), p1.tolist())):
        lines.append((t, p, p))
        if k % 199 == 50: lines.append((t, p, p))                                   # the same line twice
        elif k % 199 == 120: lines.append((t, max(1, p - 7), p + 12))               # a 20-base line over the site just printed
        elif k % 331 == 200 and len(lines) > 40: lines.append(lines[-40])           # an earlier line again, out of order
    t0 = time.time()
    def at_sites(t, cov):
        mine = [(b, e) for (tt, b, e) in lines if tt == t]
        nl = sum(int(np.count_nonzero(cov[b - 1:e])) for b, e in mine)
        ev = sum(int(cov[b - 1:e].sum()) for b, e in mine)
        return nl, ev
    contigs, per, n_reads, _, _ = make_genome(d, args.contigs, L, "wgs30x", 100, at_sites)
    want_lines = sum(x[0] for x in per); events = sum(x[1] for x in per)
    t_gen = time.time() - t0
    ref_reader = args.set_ref_env(os.path.getsize(os.path.join(d, "g.bam")))
    with open(os.path.join(d, "sites"), "w") as f:
        f.write("".join("%s\t%d\t%d\n" % (contigs[t][0], b, e) for t, b, e in lines))
    base = [args.cli, "-w", "0", "-q", "20", "-b", "13", "-f", "g.fa", "-l"]
    best, stages = timed_to_devnull(base + ["sites", "g.bam"] + args.cli_extra, d, args.reps)
    # ---- once more into a file: line count, and the text the sub-list's lines must be found in
    with open(os.path.join(d, "full.out"), "wb") as f:
        _, rc, err = run(base + ["sites", "g.bam"] + args.cli_extra, d, f)
    assert rc == 0, err
    full = open(os.path.join(d, "full.out"), "rb").read().split(b"\n")[:-1]
    assert len(full) == want_lines, "the full run printed %d lines, the covered positions of the list are %d" % (len(full), want_lines)
    # ---- the reference's own main() on blocks of consecutive lines spread over the list
    blk = 25; nblk = max(1, -(-args.check_lines // blk)); starts = np.unique(np.linspace(0, max(len(lines) - blk, 0), nblk).astype(int))
    sub = [lines[i] for s in starts for i in range(s, min(s + blk, len(lines)))]
    if nblk * blk >= len(lines):
        sub = list(lines)                      # (a list shorter than what is asked for: all of it — overlapping blocks would repeat lines out of order)
    nproc = max(1, min(args.procs, len(sub) // 20 + 1)); parts = [sub[i * len(sub) // nproc:(i + 1) * len(sub) // nproc] for i in range(nproc)]
    for i, part in enumerate(parts):
        open(os.path.join(d, "sub%d" % i), "w").write("".join("%s\t%d\t%d\n" % (contigs[t][0], b, e) for t, b, e in part))
    open(os.path.join(d, "sub"), "w").write("".join("%s\t%d\t%d\n" % (contigs[t][0], b, e) for t, b, e in sub))
    t0 = time.perf_counter()
    def ref_part(i):
        with open(os.path.join(d, "ref%d.out" % i), "wb") as f:
            _, rc, err = run([args.ref_cli, "-w", "0", "-q", "20", "-b", "13", "-f", "g.fa", "-l", "sub%d" % i, "g.bam"], d, f)
        assert rc == 0, err
    with ThreadPoolExecutor(nproc) as ex:
        list(ex.map(ref_part, range(nproc)))
    t_ref = time.perf_counter() - t0
    want = b"".join(open(os.path.join(d, "ref%d.out" % i), "rb").read() for i in range(nproc))
    with open(os.path.join(d, "sub.out"), "wb") as f:
        _, rc, err = run(base + ["sub", "g.bam"] + args.cli_extra, d, f)
    assert rc == 0, err
    got = open(os.path.join(d, "sub.out"), "rb").read()
    assert got == want, "drop-in and the reference's own main() differ on the sub-list (%d vs %d bytes)" % (len(got), len(want))
    sub_lines = got.split(b"\n")[:-1]
    assert is_subsequence(sub_lines, full), "the sub-list's lines are not found in order in the full run's output"
    ref_events = events * len(sub) / max(len(lines), 1)
    sh = sharded(args, d, base + ["sites", "g.bam"] + args.cli_extra, events, best)
    return {"sharded": sh, "what": "config 4 through the drop-in CLI: bam-readcount -w0 -q20 -b13 -f g.fa -l sites g.bam > /dev/null; %d contigs x %.1f Mbp = %.0f Mbp at 30x (%d reads; genome scaled 1:%.1f "
                    "against BASELINE's 24 contigs / 3.1 Gbp), %d site-list lines at BASELINE's spacing of one site per %.0f kb over all contigs in file order "
                    "(every 199th line twice, every 199th followed by a 20-base line over it, every 331st followed by an earlier line again, out of order)"
                    % (args.contigs, args.contig_mbp, total / 1e6, n_reads, GENOME_BP / total, len(lines), total / max(n_sites, 1) / 1e3),
            "seconds": round(best, 3), "sites_per_s": round(len(lines) / best, 1), "lines_per_s": round(want_lines / best, 1),
            "value": round(events / best, 1), "unit": "pileup base-events/s", "events": int(events), "site_lines": len(lines), "printed_lines": int(want_lines),
            "stages": stages, "bam_bytes": os.path.getsize(os.path.join(d, "g.bam")), "generate_seconds": round(t_gen, 1),
            "validated": {"lines_vs_reference_main": len(sub), "printed_lines_checked": len(sub_lines), "byte_exact_vs_reference_main": True, "found_in_order_in_full_output": True,
                          "full_line_count_equals_covered_positions": True, "full_output_md5": hashlib.md5(b"\n".join(full) + b"\n").hexdigest()},
            "cpu_reference_main": {"seconds": round(t_ref, 2), "processes": nproc, "site_lines": len(sub), "sites_per_s_per_process": round(len(sub) / max(t_ref, 1e-9) / nproc, 1),
                                   "events_per_s_per_process": round(ref_events / max(t_ref, 1e-9) / nproc, 1),
                                   "what": "oracle/_ref/bam-readcount-ref (the reference's own main(), one samfetch + pileup per line) on the checked sub-list, %d processes side by side; BAM access: %s" % (nproc, ref_reader)}}


def leg_tumor(args, d):
    L = int(args.contig_mbp * 1e6)
    pad = 200_000                                          # small neighbours: the region is one contig of a multi-contig file
    t0 = time.time()
    cfg = synthgen.CONFIGS["tumor200x"]; n_libs = cfg["n_libs"]; rgs = 2
    contigs = [("chr1", pad), ("chr2", L), ("chr3", pad)]
    w = synthgen.BamWriter(os.path.join(d, "g.bam"), contigs, n_libs=n_libs, rgs_per_lib=rgs)
    refs = []; n_reads = 0; cov2 = None
    for t, (nm, ln) in enumerate(contigs):
        ref, a = synthgen.generate(ln, "tumor200x", seed=200 + 17 * t)
        w.add(t, a); refs.append((nm, ref)); n_reads += len(a["pos"])
        if t == 1: cov2 = coverage(a, ln)
        del a
    w.close(); synthgen.write_fasta(os.path.join(d, "g.fa"), refs)
    t_gen = time.time() - t0
    ref_reader = args.set_ref_env(os.path.getsize(os.path.join(d, "g.bam")))
    events = int(cov2.sum(dtype=np.int64)); want_lines = int(np.count_nonzero(cov2))
    base = [args.cli, "-w", "0", "-p", "-i", "-f", "g.fa", "g.bam"]
    best, stages = timed_to_devnull(base + ["chr2"] + args.cli_extra, d, args.reps)
    # ---- full line count (second run, piped into a counter)
    p1 = subprocess.Popen(base + ["chr2"] + args.cli_extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    p2 = subprocess.run(["wc", "-l"], stdin=p1.stdout, stdout=subprocess.PIPE); p1.wait()
    got_lines = int(p2.stdout.split()[0])
    assert p1.returncode == 0 and got_lines == want_lines, "the full run printed %d lines, the region has %d covered positions" % (got_lines, want_lines)
    # ---- the first --check-mbp through the reference's own main(), in pieces on all cores
    cb = int(min(args.check_mbp * 1e6, L)); npieces = max(args.procs * 2, 1)
    cuts = [(1 + cb * i // npieces, cb * (i + 1) // npieces) for i in range(npieces)]
    t0 = time.perf_counter()
    def ref_piece(i):
        with open(os.path.join(d, "ref%d.out" % i), "wb") as f:
            _, rc, err = run([args.ref_cli, "-w", "0", "-p", "-i", "-f", "g.fa", "g.bam", "chr2:%d-%d" % cuts[i]], d, f)
        assert rc == 0, err
    with ThreadPoolExecutor(args.procs) as ex:
        list(ex.map(ref_piece, range(npieces)))
    t_ref = time.perf_counter() - t0
    with open(os.path.join(d, "chk.out"), "wb") as f:
        _, rc, err = run(base + ["chr2:1-%d" % cb] + args.cli_extra, d, f)
    assert rc == 0, err
    # compare piece by piece, streaming (the text of 1 Mbp with four libraries is 1.5 GB)
    nbytes = 0
    with open(os.path.join(d, "chk.out"), "rb") as g:
        for i in range(npieces):
            want = open(os.path.join(d, "ref%d.out" % i), "rb").read()
            got = g.read(len(want))
            assert got == want, "drop-in and the reference's own main() differ in piece %d (chr2:%d-%d)" % (i, cuts[i][0], cuts[i][1])
            nbytes += len(want)
        assert g.read(1) == b"", "the drop-in printed more than the reference for chr2:1-%d" % cb
    ref_events = int(cov2[:cb].sum(dtype=np.int64))
    sh = sharded(args, d, base + ["chr2"] + args.cli_extra, events, best)
    return {"sharded": sh, "what": "config 5 through the drop-in CLI: bam-readcount -w0 -p -i -f g.fa g.bam chr2 > /dev/null; chr2 = %.2f Mbp at 200x, 4 libraries / 8 read groups, 10 %% indel reads "
                    "(BASELINE: 50 Mbp over 8 GPUs = 6.25 Mbp per GPU), inside a 3-contig BAM of %d reads" % (args.contig_mbp, n_reads),
            "seconds": round(best, 3), "value": round(events / best, 1), "unit": "pileup base-events/s", "events": events, "positions_per_s": round(want_lines / best, 1),
            "printed_lines": want_lines, "stages": stages, "bam_bytes": os.path.getsize(os.path.join(d, "g.bam")), "generate_seconds": round(t_gen, 1),
            "validated": {"region_vs_reference_main_mbp": cb / 1e6, "text_bytes_checked": nbytes, "byte_exact_vs_reference_main": True, "full_line_count_equals_covered_positions": True},
            "cpu_reference_main": {"seconds": round(t_ref, 2), "processes": args.procs, "events": ref_events, "events_per_s_per_process": round(ref_events / max(t_ref, 1e-9) / args.procs, 1),
                                   "what": "oracle/_ref/bam-readcount-ref (the reference's own main()) on chr2:1-%d cut into %d regions, %d processes side by side; BAM access: %s" % (cb, npieces, args.procs, ref_reader)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", required=True, choices=["sites", "tumor"])
    ap.add_argument("--contigs", type=int, default=8)
    ap.add_argument("--contig-mbp", type=float, default=None, help="sites: every contig (default 12.5); tumor: the region's contig (default 6.25)")
    ap.add_argument("--sites", type=int, default=0, help="site-list lines before the extra ones (0: BASELINE's spacing, one per 31 kb)")
    ap.add_argument("--check-lines", type=int, default=1000)
    ap.add_argument("--check-mbp", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--procs", type=int, default=max(1, min(len(os.sched_getaffinity(0)), 16)))
    ap.add_argument("--cli", default=CLI); ap.add_argument("--ref-cli", default=REF_CLI)
    ap.add_argument("--ref-reader", default="auto", choices=["auto", "indexed", "independent"],
                    help="how the reference-compiled main() reads the BAM (oracle/ref_shim/shim_hts.cpp): 'independent' = the shim's own reader, which loads every record of the file "
                         "and ignores the index (a second BAM decoder, fine for megabytes); 'indexed' = the repository's BGZF/BAM/BAI reader under the reference's samfetch "
                         "(BRC_SHIM_PRODUCT_READER=1); auto: independent below 200 MB of BAM")
    ap.add_argument("--ranks", type=int, default=0, help="also run the leg's command as this many processes, one per GPU (--brc-ranks), timed and compared byte for byte with the one-process output ('sharded' in the result)")
    ap.add_argument("--rank-devices", default=None, help="BRC_DEVICES of the ranks run (e.g. 0,0: two ranks on one GPU); default: GPUs 0..ranks-1")
    ap.add_argument("--keep", default=None, help="work in this directory and keep the files")
    ap.add_argument("cli_extra", nargs="*", help="extra arguments for the drop-in (after --)")
    args = ap.parse_args()
    args.cli, args.ref_cli = os.path.abspath(args.cli), os.path.abspath(args.ref_cli)
    if args.contig_mbp is None:
        args.contig_mbp = 12.5 if args.leg == "sites" else 6.25
    synthgen.build()
    global REF_ENV
    def set_ref_env(bam_bytes):
        global REF_ENV
        indexed = args.ref_reader == "indexed" or (args.ref_reader == "auto" and bam_bytes >= 200e6)
        REF_ENV = dict(os.environ, BRC_SHIM_PRODUCT_READER="1") if indexed else dict(os.environ, BRC_SHIM_PRODUCT_READER="0")
        return "indexed (this repository's BGZF/BAM/BAI reader under the reference's samfetch)" if indexed else "independent (the shim's own whole-file BAM decoder)"
    args.set_ref_env = set_ref_env
    d = args.keep or tempfile.mkdtemp(prefix="brc_e2e_%s_" % args.leg)
    os.makedirs(d, exist_ok=True)
    try:
        res = leg_sites(args, d) if args.leg == "sites" else leg_tumor(args, d)
    finally:
        if not args.keep:
            shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
