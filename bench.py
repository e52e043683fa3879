#!/usr/bin/env python3
"""bench.py — pileup base-events/s of the MI355X readcount engine on BASELINE.json's timed configurations.

A "step" = one pass of the whole device pipeline (k_annotate -> scans -> k_tiles -> k_pileup2 -> indel path) over one
synthetic region whose inputs are already resident in HBM (uploaded before the timed region); the results (compact
slot planes + third-allele list + indel buckets) are written to HBM.  value = events of all ranks per step /
max-over-ranks step time.

  --mode weak    (default) BASELINE config 3 per GPU: synthetic 30x WGS, 150 bp, one 50-Mbp contig, -q20 -b13; with N ranks
                 every rank owns a contig of the same size (independent genomic intervals, no data-path collective)
  --mode strong  BASELINE config 5: synthetic 200x tumour, 4 libraries, -p -i, one 50-Mbp contig cut into N intervals,
                 one per rank (fixed total work)
  --mode sites   BASELINE config 4: a -l list of 100 000 single-base sites in file order, cut into N contiguous slices;
                 each rank lays its sites side by side on one virtual axis (the CLI's site-list planner) and runs the
                 pipeline once per step.  The genome is scaled: one 50-Mbp contig per rank instead of 3.1 Gbp.

`--gpus N` with N > 1 re-executes itself under torch.distributed.run (one rank per GPU, RCCL) unless it already runs
under a launcher.  Extra objects on the JSON line:
  roofline      SURVEY 8d's algorithmic bytes (B_in + B_ref + B_out, dense 312 B per position and library) over the WHOLE device step
                (HIP events on the engine's stream), against the 8 TB/s HBM peak; per_kernel: each of the two big kernels against its OWN
                bytes and against its PMC traffic (profiles/r06_traffic.json, reported only when stamped with the loaded kernel object)
  cpu_baseline  the C oracle (CPU restatement of the reference semantics): 1 thread like the reference on a prefix, the reference's own
                sources compiled over a shim (oracle/_ref) on a smaller prefix, and all cores = the whole-region validation below
  e2e           the drop-in command line on a generated BAM + BAI -> /dev/null (BAM decode, PCIe, formatting included)
  e2e_sites     BASELINE config 4 end to end: `bam-readcount -l sites` over a generated 8-contig 30x BAM + BAI, sites at BASELINE's
  e2e_tumor     spacing in file order; config 5 end to end: `-p -i` on a 200x / 4-library / 8-read-group BAM.  Both validated against
                the reference's own main() (oracle/_ref/bam-readcount-ref) — tools/e2e_configs.py
  e2e_sharded   the same two commands as N processes, one rank per GPU (`bam-readcount --brc-ranks N`: contiguous event-weighted slices
                of the work list in file order, text in rank order): seconds, aggregate events/s, every rank's own seconds and share,
                speed-up and efficiency against the one-process run, the whole output byte-identical to it.  --gpus 1: two ranks on GPU 0
  validated     full_contig: the result of the TIMED region — every position of it — equals the oracle's: the region is cut into
                windows, the oracle computes each as a region of its own on all usable cores (tools/fullcheck.py), the HIP side
                reads the same windows back with brc_fetch_window; planes bit for bit, indel lists, text byte for byte (digests).
                Besides: planes / text / device-side text of a prefix, and event count == sum of the reads' in-window spans
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_COPY_GBS = 6290.0    # measured streaming-copy rate of the same guide's chip table
CLI = os.path.join(ROOT, "bam_readcount_amd", "csrc", "bam-readcount")


def effective_cpus():
    """CPUs this process may really use at once: the affinity mask capped by a cgroup CPU quota (cpu.max / cfs_quota_us)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(-(-int(q) // int(per)))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return n


def site_batch(np, capi, arrs, ref, sites, window=384, lead=170):
    """The CLI planner's layout for single-base -l lines: site i's reads (those samfetch would return for [s-2, s)) and its
    reference slice are translated to virtual positions [i * window, (i + 1) * window)."""
    pos = arrs["pos"].astype(np.int64); ends = capi.read_ends(arrs)
    first = np.searchsorted(pos, sites - lead, side="left"); last = np.searchsorted(pos, sites, side="left")   # pos in [s - lead, s)
    cnt = last - first
    ridx = np.repeat(first, cnt) + (np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    sidx = np.repeat(np.arange(len(sites)), cnt)
    keep = ends[ridx] > sites[sidx] - 2                                 # overlaps [s - 2, s)
    ridx, sidx = ridx[keep], sidx[keep]
    sub = capi.select_reads(arrs, ridx)
    delta = sidx * window + lead - sites[sidx]
    sub["pos"] = (sub["pos"].astype(np.int64) + delta).astype(np.int32)
    vref = np.full(len(sites) * window, ord("N"), np.uint8)
    src = (sites[:, None] - lead + np.arange(window)[None, :]).clip(0, len(ref) - 1)
    vref[:] = ref[src].reshape(-1)
    # unit of work: reads covering the reported position s - 1 itself (the fetch also returns reads that only
    # cover the lead position s - 2)
    events = int(((pos[ridx] <= sites[sidx] - 1) & (ends[ridx] > sites[sidx] - 1)).sum())
    vbeg0 = np.arange(len(sites), dtype=np.int64) * window + lead - 1          # the lines themselves on the virtual axis: [vbeg0, vbeg0 + 1)
    return sub, vref, events, vbeg0


def eb_positions_padded(n):
    """plane stride of the engine: positions rounded up to 64 (every plane store of a wave is one aligned 256-byte segment)"""
    return (int(n) + 63) & ~63


def fullcheck_window_reads(capi, arrs, ends, pos64, a, b):
    """reads samfetch would return for [a - 1, b) (tools/fullcheck.py: window_reads)"""
    import fullcheck
    return fullcheck.window_reads(capi, arrs, ends, pos64, int((ends - pos64).max()) if len(pos64) else 0, a, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (300 x ~7 ms: a timed region above 2 s)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="weak", choices=["weak", "strong", "sites"])
    ap.add_argument("--config", default=None, choices=["wgs30x", "tumor200x", "wgs30x_mixed", "novaseq", "long10k", "ont", "ont_ul", "hifi_eqx"], help="data model (default: wgs30x; tumor200x for --mode strong; wgs30x_mixed: config 3 with 30 %% of the reads trimmed to U[100,149] and 10 %% 250 bases long; long10k: 10-kb reads at 30x; ont: 3-10-kb reads with an insertion or deletion every ~15 bases at 30x (use --contig-mbp 20); ont_ul: the same with 30-100-kb reads — functional and throughput points outside BASELINE's configurations)")
    ap.add_argument("--contig-mbp", type=float, default=50.0, help="weak/sites: contig per GPU; strong: the whole contig")
    ap.add_argument("--sites", type=int, default=100000, help="--mode sites: lines of the site list (all ranks together)")
    ap.add_argument("--cpu-sample-mbp", type=float, default=8.0, help="prefix timed with the 1-thread CPU oracle and used for validation (0 = skip)")
    ap.add_argument("--allow-large", action="store_true", help="--mode strong: accept more than 12.5 Mbp of 200x data per rank")
    ap.add_argument("--cpu-ref-mbp", type=float, default=0.5, help="prefix timed with the reference-compiled library oracle/_ref (0 = skip)")
    ap.add_argument("--cpu-all-cores", type=int, default=-1, help="oracle processes of the whole-region validation = the all-cores CPU baseline (-1: min(usable CPUs, 64); 0 = skip both)")
    ap.add_argument("--full-check", type=int, default=-1, help="validate the WHOLE timed region against the oracle on all cores, window by window (brc_fetch_window): -1 = on a 1-GPU run "
                    "of --mode weak / strong with a CPU sample, 2 = planes only (no text: config 5 whole prints 72 GB), 0 = off")
    ap.add_argument("--e2e-mbp", type=float, default=30.0, help="contig of the end-to-end command-line run (0 = skip)")
    ap.add_argument("--e2e-configs", type=int, default=-1, help="BASELINE configs 4 and 5 END TO END through the drop-in command line on generated multi-contig BAM + BAI files "
                    "(tools/e2e_configs.py; e2e_sites / e2e_tumor on the line): -1 = on the default single-GPU config-3 run, 1 = yes, 0 = no")
    ap.add_argument("--e2e-sites-mbp", type=float, default=25.0, help="e2e_sites: length of each of its 8 contigs")
    ap.add_argument("--abi-mbp", type=float, default=10.0, help="prefix run through the C-ABI from host batches to host text (abi_roundtrip; 0 = skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r06_traffic.json"), help="PMC passes of the big kernels (FETCH_SIZE / WRITE_SIZE per launch), stamped with the kernel object they ran on; <same name with pmc_summary> holds the SQ counters")
    ap.add_argument("--other-configs", type=int, default=-1, help="1: also run BASELINE config 4 (--mode sites) and the per-GPU shape of config 5 (--mode strong --contig-mbp 6.25) "
                    "in sub-processes and put their lines under other_configs (-1: yes on the default single-GPU config-3 run, no otherwise)")
    # test infrastructure (tests/test_bench_multirank.py): the rank arithmetic of this script — who owns which interval / slice of
    # the site list, the reductions of the metrics line — executed on CPUs: gloo instead of RCCL, and the C-ABI served by the CPU
    # lane simulator named here.  The line it prints says so ("dry_run") and carries no throughput.
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 branch even with one rank: init_process_group(nccl = RCCL), the barriers, both all-reduces, the all-gathers and the "
                    "destroy — so that the first time RCCL sees this code is not an 8-GPU run (tests/test_bench_dist_gpu.py)")
    ap.add_argument("--rank-check-mbp", type=float, default=1.5, help="every rank validates this prefix of ITS OWN interval against the oracle after the timed region (an N-rank run, or --force-dist; "
                    "sites mode: 100 of the rank's own sites); 0 = skip")
    ap.add_argument("--e2e-tumor-mbp", type=float, default=0.0, help="e2e_tumor: the region's contig (0: 6.25 Mbp per rank, at most 25)")
    ap.add_argument("--dry-run-lib", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--as-rank", type=int, default=None, help=argparse.SUPPRESS)     # with --as-world: one rank's share, without a launcher
    ap.add_argument("--as-world", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    config = args.config or ("tumor200x" if args.mode == "strong" else "wgs30x")

    # ---- N > 1 without a launcher: become `python -m torch.distributed.run ... bench.py <same arguments>`
    dry = args.dry_run_lib is not None
    if args.gpus > 1 and "RANK" not in os.environ and args.as_rank is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import numpy as np
    import torch
    import synthgen
    from bam_readcount_amd import capi

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    emulated = args.as_rank is not None
    if emulated:                                     # (dry run only) the share of rank --as-rank of --as-world, on its own
        if not dry:
            raise SystemExit("--as-rank is part of the dry run (--dry-run-lib)")
        rank, world = args.as_rank, int(args.as_world or args.gpus)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ncpu = os.cpu_count() or 1
    ncpu_eff = effective_cpus()
    nworkers = (min(ncpu_eff, 64) if args.cpu_all_cores < 0 else args.cpu_all_cores) if (rank == 0 and world == 1 and args.cpu_sample_mbp > 0) else 0

    per_lib = config == "tumor200x"
    names = ["lib%d" % i for i in range(synthgen.CONFIGS[config]["n_libs"])] if per_lib else ()
    opts = dict(min_mapq=20, min_bq=13) if not per_lib else dict(min_mapq=0, min_bq=0, per_lib=True, insertion_centric=True)

    total_len = int(args.contig_mbp * 1e6)
    contig_len = total_len // world if args.mode == "strong" else total_len
    if args.mode == "strong" and contig_len > 50_000_000 and not args.allow_large:
        # (BASELINE config 5 itself — 50 Mbp at 200x, 67 M reads, 21 GB of pinned staging, 80 GB of HBM — runs on one GPU in
        # 13 s all told, 36 ms per step; anything larger wants --allow-large)
        raise SystemExit("--mode strong with %d rank(s) puts %.1f Mbp of 200x data (%.0f M reads) on one GPU: use --gpus 4 / 8, a smaller --contig-mbp "
                         "(6.25 = the per-GPU shape of BASELINE config 5), or --allow-large" % (world, contig_len / 1e6, contig_len * 200 / 150 / 1e6))
    t0 = time.time()
    ref, arrs = synthgen.generate_dense(contig_len, config, seed=1 + 1000 * rank) if config in synthgen.DENSE else synthgen.generate(contig_len, config, seed=1 + 1000 * rank)
    t_gen = time.time() - t0
    # The oracle processes of the whole-region validation are forked HERE — before this process touches the GPU (a process
    # that has initialised the HIP runtime must not fork); they inherit the reads and sleep until the timed region is over.
    pool = None
    full_check = args.full_check if args.full_check >= 0 else (1 if (world == 1 and not dry and args.mode != "sites" and args.cpu_sample_mbp > 0 and nworkers > 0) else 0)
    if full_check and rank == 0 and world == 1 and not dry and args.mode != "sites":
        import fullcheck
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        pool = fullcheck.OraclePool(max(nworkers, 1), ref, arrs, names, opts, capi)

    # test infrastructure (tests/test_dist_gpu.py): BRC_BENCH_SHARE_GPU0=1 puts EVERY rank's engine on GPU 0 and the process group on gloo —
    # a real N > 1 run of this script (both all-reduces, the gathers, every rank's own validation, the e2e_sharded legs with one process
    # per rank) on a box with one GPU; RCCL itself refuses two ranks on one device, and is covered at world 1 (--force-dist).  The line says so.
    share_gpu0 = os.environ.get("BRC_BENCH_SHARE_GPU0") == "1" and not dry
    if share_gpu0:
        local_rank = 0
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
        torch.cuda.set_device(local_rank)
    dist = None
    tdev = "cpu" if (dry or share_gpu0) else "cuda"
    if (world > 1 or args.force_dist) and not emulated:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:                  # --force-dist without a launcher: a group of one
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1])); s.close()
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        if dry or share_gpu0:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm

    hip = capi.Library(os.path.abspath(args.dry_run_lib)) if dry else capi.load_product()
    # (sites mode: the result is only ever cut into lines — no dense planes on the host)
    eng = capi.Engine(hip, lib_names=names, device=local_rank, text_only=(args.mode == "sites"), **opts)
    site_events = 0
    if args.mode == "sites":
        n_mine = args.sites // world + (1 if rank < args.sites % world else 0)
        sites = np.sort(np.random.default_rng(3 + rank).integers(200, contig_len - 200, n_mine))      # 1-based site == 0-based end
        sub, vref, site_events, site_vbeg0 = site_batch(np, capi, arrs, ref, sites)
        region_len, region_ref, region_reads = len(vref), vref, sub
    else:
        region_len, region_ref, region_reads = contig_len, ref, arrs
    t0 = time.time()
    eng.begin_region(0, 0, region_len, region_ref)
    eng.push_reads(region_reads)
    if args.mode == "sites" and not os.environ.get("BRC_BENCH_NO_WINDOWS"):
        # the planner announces the lines it is going to cut out (brc_region_windows): only their tiles are piled up
        eng.region_windows(site_vbeg0.astype(np.int32), (site_vbeg0 + 1).astype(np.int32))
    t_push = time.time() - t0
    t0 = time.time()
    eng.upload()                                  # inputs resident in HBM from here on
    t_up = time.time() - t0

    def sync_all():
        if not dry: torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not dry: torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.compute()
    kernel_names = hip.kernel_names() if not dry else ["k_pileup"]
    k_pile = kernel_names.index("k_pileup")
    kms = np.zeros(len(kernel_names))
    sync_all()
    t0 = time.perf_counter()
    # exactly --steps passes of the pipeline, queued back to back on the engine's stream with one wait at the end
    # (brc_compute_n: the device never waits for the host between two steps); per-kernel times from HIP events of every pass
    ms, _tot = eng.compute_n(args.steps)
    if not dry: kms += np.array(ms)
    sync_all()
    dt = time.perf_counter() - t0
    n_events, n_positions = eng.counts()
    # (regions of reads with many CIGAR segments: the tile ranges are compacted before the pileup — what they held, what was walked)
    ps = eng.piece_steps() if hasattr(eng.L.lib, "brc_region_piece_steps") else (0, 0)
    piece_steps = {"in_tile_ranges": ps[0], "walked": ps[1], "dead_share_without_compaction": round(1.0 - ps[1] / ps[0], 4)} if ps[0] else None
    if args.mode == "sites":
        n_events, n_positions = site_events, len(sites)      # the unit of work counts the requested sites only

    tmax, ev_total, pos_total = dt, n_events, n_positions
    # device memory this rank's engine holds (inputs, intermediates, result planes): what an 8-GPU run must fit per GPU
    hbm_used = 0
    if not dry:
        free_b, total_b = torch.cuda.mem_get_info(local_rank); hbm_used = int(total_b - free_b)
    per_rank = [{"rank": rank, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4), "events": int(n_events), "positions": int(n_positions), "hbm_bytes": hbm_used}]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([n_events, n_positions], dtype=torch.int64, device=tdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)   # the only collectives: 16 bytes of counters for the metrics line ...
        tmax, ev_total, pos_total = float(t.item()), int(c[0].item()), int(c[1].item())
        # ... and every rank's own clock, share and memory (32 bytes each), so that the line of an N-GPU run describes itself
        mine = torch.tensor([dt, float(n_events), float(n_positions), float(hbm_used)], dtype=torch.float64, device=tdev)
        each = [torch.zeros(4, dtype=torch.float64, device=tdev) for _ in range(world)]
        dist.all_gather(each, mine)
        per_rank = [{"rank": r, "ms_per_step": round(float(x[0]) / max(args.steps, 1) * 1e3, 4), "events": int(x[1]), "positions": int(x[2]), "hbm_bytes": int(x[3])} for r, x in enumerate(each)]

    # ---- every rank validates a prefix of ITS OWN interval (its own sites) against the oracle, one thread, after the timed region:
    # a wrong result on rank 5 of an 8-GPU run must not print a healthy line.  The verdicts are MIN-reduced.
    rank_check = None
    if (dist is not None or emulated) and args.rank_check_mbp > 0:
        rank_check = {"ok": 1, "events": 0, "what": "", "error": None}
        try:
            import parity
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            oracle_r = capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so"))
            ends_r = capi.read_ends(arrs)
            oe = capi.Engine(oracle_r, lib_names=names, **opts)
            if args.mode == "sites":
                eng.fetch_result()
                pick = np.unique(np.linspace(0, len(sites) - 1, min(100, len(sites))).astype(np.int64)); pos64 = arrs["pos"].astype(np.int64)
                for i in pick:
                    sp = int(sites[i])
                    lo_i = int(np.searchsorted(pos64, sp - 2 - 1000, side="left")); hi_i = int(np.searchsorted(pos64, sp, side="left"))
                    idx = lo_i + np.nonzero(ends_r[lo_i:hi_i] > max(sp - 2, 0))[0]
                    oe.begin_region(0, sp - 1, sp, ref); oe.push_reads(capi.select_reads(arrs, idx)); oe.end_region()
                    want = oe.format_region("chrS"); oe.clear_indel_queue(); rank_check["events"] += oe.counts()[0]
                    d = int(site_vbeg0[i]) + 1 - sp
                    if eng.format_window("chrS", sp - 1 + d, sp + d, d) != want:
                        raise AssertionError("site %d (position %d): the planner's line differs from the oracle's" % (i, sp))
                rank_check["what"] = "%d of this rank's %d sites, line by line" % (len(pick), len(sites))
            else:
                vlen = int(min(args.rank_check_mbp * 1e6 * (0.15 if per_lib else 1.0), contig_len))
                oe.begin_region(0, 0, vlen, ref); oe.push_reads(capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends_r, -1, vlen)))
                ores = oe.end_region(); otext = oe.format_region_np("chrS").copy()
                eng.clear_indel_queue()
                hres = eng.fetch_window(0, vlen); htext = eng.format_region_np("chrS"); eng.clear_indel_queue()      # the TIMED region's own result
                parity.assert_results_equal(hres, ores, "rank %d prefix" % rank)
                if not (len(htext) == len(otext) and np.array_equal(htext, otext)):
                    raise AssertionError("text of the HIP engine and of the oracle differ")
                rank_check["events"] = int(ores.n_events)
                rank_check["what"] = "planes bit for bit + text of [0, %d) of this rank's interval, read back from the timed region (brc_fetch_window)" % vlen
            oe.close()
        except Exception as ex:                                  # noqa: BLE001 — reduced and reported; raising here would hang the other ranks
            rank_check["ok"] = 0; rank_check["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
        per_rank_checks = [rank_check]
        if dist is not None:
            okt = torch.tensor([rank_check["ok"]], dtype=torch.int64, device=tdev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            mine_v = torch.tensor([rank_check["ok"], rank_check["events"]], dtype=torch.int64, device=tdev)
            each_v = [torch.zeros(2, dtype=torch.int64, device=tdev) for _ in range(world)]
            dist.all_gather(each_v, mine_v)
            for r, x in enumerate(each_v):
                per_rank[r]["validated_ok"] = bool(int(x[0])); per_rank[r]["validated_events"] = int(x[1])
            all_ok = bool(int(okt.item()))
        else:
            per_rank[0]["validated_ok"] = bool(rank_check["ok"]); per_rank[0]["validated_events"] = rank_check["events"]
            all_ok = bool(rank_check["ok"])
        rank_check["all_ranks_ok"] = all_ok
        if not rank_check["ok"]:
            sys.stderr.write("bench.py: rank %d: validation of its own interval FAILED: %s\n" % (rank, rank_check["error"]))

    if dry:
        # the dry run ends here: what every rank (or the emulated one) owned, and — on rank 0 — the reduced totals
        if rank == 0 or emulated:
            print(json.dumps({"dry_run": "rank arithmetic only (gloo, %s)" % hip.kind(), "value": None, "n_gpus": world, "rank": rank, "mode": args.mode,
                              "events_per_step": int(ev_total), "positions_per_step": int(pos_total), "own_events": int(n_events), "own_positions": int(n_positions), "per_rank": per_rank,
                              "reads_per_gpu": int(len(region_reads["pos"])),
                              "validated": None if rank_check is None else {"all_ranks_ok": rank_check["all_ranks_ok"], "own": rank_check["what"], "own_error": rank_check["error"]}}))
        eng.close()
        if dist is not None:
            dist.destroy_process_group()
        if rank_check is not None and not rank_check["all_ranks_ok"]:
            raise SystemExit(3)
        return
    if rank == 0:
        ms_per_step = tmax / args.steps * 1e3
        value = ev_total * args.steps / tmax
        res_libs = len(names) if per_lib else 1
        # ---- roofline of the dominant kernel
        eng_events, eng_positions = eng.counts()
        if args.mode == "sites":
            eng_positions = n_positions          # SURVEY 8d's B_out counts the lines printed: the requested sites, not what their tiles hold besides
        b_in, b_ref, b_out = synthgen.algorithmic_bytes(region_reads, eng_positions, res_libs, 0, ref_positions=region_len)
        alg = b_in + b_ref + b_out
        step_s = ms_per_step * 1e-3
        # ---- what each of the two big kernels must move by ITS OWN account (this design's traffic, not SURVEY 8d's):
        #   K1 (k_annotate_groups + k_refcode + k_annotate_wave): the reads once (8d's B_in), the reference once; out: one event byte per base
        #       (rows padded to 16), a 48-byte record + 8 bytes of (start, reach) per read segment, a 64-byte record per read
        #   k_pileup2: every event byte once, a 48-byte segment record per (segment, 64-position tile it touches), one reference code per
        #       position, a tile range per (tile, library); out: the compact result, 116 bytes per (position, library)
        # (read segments from the CIGARs: one per M / = / X operator of a read; -i's one-base segments and the single segment of a read that
        # cannot count are not modelled: an estimate, within a percent on these data models)
        Lq = region_reads["l_qseq"].astype(np.int64); cig = region_reads["cigar"]; opc = cig & 15
        n_seg = int(np.count_nonzero((opc == 0) | (opc == 7) | (opc == 8)))
        rends = capi.read_ends(region_reads); rpos = region_reads["pos"].astype(np.int64)
        seg_tiles = int((((rends - 1) >> 6) - (rpos >> 6) + 1).clip(min=0).sum()) + max(n_seg - len(rpos), 0)      # (a read's segments share its tiles; every further segment: one more)
        eb_bytes = int(((Lq + 15) & ~15).sum())
        own = {"k_annotate": {"in": int(b_in + b_ref), "out": int(eb_bytes + 56 * n_seg + 64 * len(rpos))},
               "k_pileup": {"in": int(Lq.sum() + 48 * seg_tiles + region_len + 8 * ((region_len + 63) // 64) * res_libs), "out": 116 * int(eb_positions_padded(region_len)) * res_libs}}
        if args.mode == "sites" and not os.environ.get("BRC_BENCH_NO_WINDOWS"):
            # k_pileup2 piles up only the tiles the announced windows [site - 1, site] touch: per such tile the records of the reads over it
            # (48 bytes each) and their event bytes inside the tile (at most 64), its reference codes and range; out: the tile's compact result
            wt = np.unique(np.concatenate([(site_vbeg0.astype(np.int64) - 1).clip(min=0) >> 6, site_vbeg0.astype(np.int64) >> 6]))
            over = np.searchsorted(rpos, (wt + 1) << 6, "left") - np.searchsorted(np.sort(rends), wt << 6, "right")
            own["k_pileup"] = {"in": int((48 + 64) * int(over.clip(min=0).sum()) + (64 + 8 * res_libs) * len(wt)), "out": 116 * 64 * len(wt) * res_libs}
        traffic_json = args.traffic_json if os.path.exists(args.traffic_json) else None
        tj = None; traffic_src = None; kobj = capi.kernel_object_hash(hip.path)
        if traffic_json and args.mode != "sites" and world == 1:
            tj = json.load(open(traffic_json)).get(config)
            if tj and abs(float(tj.get("contig_mbp", 50.0)) - contig_len / 1e6) >= 1e-6:
                tj = None                                                # (the passes measured another workload)
            if tj and tj.get("kernel_object_sha256_16") != kobj:
                traffic_src = "%s holds PMC passes of kernel object %s; the loaded library's is %s: counters of other kernels are not reported (repeat tools/gpu_r6_profile.sh)" % (
                    os.path.relpath(traffic_json, ROOT), tj.get("kernel_object_sha256_16"), kobj)
                tj = None
            elif tj:
                traffic_src = "%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on kernel object %s (= the loaded library's), not measured in this run" % (os.path.relpath(traffic_json, ROOT), kobj)
        def kernel_entry(slot, name):
            ms = float(kms[kernel_names.index(slot)]); ob = own[slot]; tot = ob["in"] + ob["out"]
            tr = (tj or {}).get(name, {}).get("hbm_bytes_per_launch") if tj else None
            return {"ms": round(ms, 4), "own_bytes_in": ob["in"], "own_bytes_out": ob["out"],
                    "achieved_GBs_by_own_bytes": round(tot / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                    "frac_by_own_bytes": round(tot / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                    "traffic": tr, "traffic_GBs": round(tr / (ms * 1e-3) / 1e9, 1) if (tr and ms > 0) else None,
                    "frac_by_traffic": round(tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (tr and ms > 0) else None,
                    "traffic_over_own_bytes": round(tr / tot, 3) if tr else None}
        per_kernel = {"k_annotate_groups": kernel_entry("k_annotate", "k_annotate_groups"), "k_pileup2": kernel_entry("k_pileup", "k_pileup2")}
        step_traffic = (tj or {}).get("step_hbm_bytes") if tj else None
        # The roofline that really binds the two kernels: vector-instruction issue.  From the committed PMC passes of this workload at the
        # loaded kernels (not measured in this run): a wave64 VALU instruction occupies its SIMD for 4 cycles, so SQ_INSTS_VALU x 4 / 1024
        # SIMDs = the cycles every SIMD spends issuing vector instructions, against the kernel's own busy cycles (SQ_BUSY_CYCLES is summed
        # over the chip's 32 shader engines; its quotient by the duration is the clock the kernel really ran at, below the 2.4 GHz peak).
        issue = None
        pmc_json = os.path.join(os.path.dirname(traffic_json), os.path.basename(traffic_json).replace("traffic", "pmc_summary")) if traffic_json else None
        if tj is not None and pmc_json and os.path.exists(pmc_json):
            pm = json.load(open(pmc_json)).get(config, {})
            def issue_of(name):
                k = pm.get(name)
                if not k or not k.get("SQ_BUSY_CYCLES"):
                    return None
                busy = k["SQ_BUSY_CYCLES"] / 32.0
                return {"valu_insts_per_launch": int(k["SQ_INSTS_VALU"]), "salu_insts_per_launch": int(k["SQ_INSTS_SALU"]),
                        "valu_issue_cycles_per_simd": round(k["SQ_INSTS_VALU"] * 4 / 1024.0), "busy_cycles": round(busy),
                        "valu_issue_frac": round(k["SQ_INSTS_VALU"] * 4 / 1024.0 / busy, 4), "salu_issue_frac": round(k["SQ_INSTS_SALU"] * 4 / 1024.0 / busy, 4),
                        "lane_utilisation": round(k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_INSTS_VALU"]), 4) if k.get("SQ_THREAD_CYCLES_VALU") else None}
            issue = {"k_pileup2": issue_of("k_pileup2"), "k_annotate_groups": issue_of("k_annotate_groups"),
                     "source": "%s: separate rocprofv3 --pmc passes of this workload on kernel object %s, not measured in this run" % (os.path.relpath(pmc_json, ROOT), kobj)}
        # HEADLINE: SURVEY 8d's algorithmic bytes (the reads once, the reference once, the DENSE 312-byte result per position and library — what
        # any implementation of the path must move) over the WHOLE device step — annotation, scans, pileup, third-allele fold, indel path —,
        # HIP events on the engine's stream.  No single kernel is charged with bytes another kernel moves.
        roof = {"bound": "hbm", "kernel": "the whole device step (k_annotate_groups -> scans -> k_pileup2 -> third-allele fold -> indel path); k_pileup2 is the longest of its kernels",
                "achieved": round(alg / step_s / 1e9, 1) if world == 1 else round(alg / (float(sum(kms)) * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg / step_s / 1e9 / HBM_PEAK_GBS, 4) if world == 1 else round(alg / (float(sum(kms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy_rate": round(alg / step_s / 1e9 / HBM_COPY_GBS, 4) if world == 1 else None,
                "traffic": step_traffic, "traffic_source": traffic_src, "kernel_object_sha256_16": kobj,
                "traffic_GBs": round(step_traffic / step_s / 1e9, 1) if (step_traffic and world == 1) else None,
                "frac_of_peak_by_traffic": round(step_traffic / step_s / 1e9 / HBM_PEAK_GBS, 4) if (step_traffic and world == 1) else None,
                "algorithmic_bytes_per_step": alg, "bytes_per_event": round(alg / max(eng_events, 1), 3),
                "step_ms": round(ms_per_step, 4) if world == 1 else round(float(sum(kms)), 4),
                "per_kernel": per_kernel,
                "instruction_issue": issue,
                "in_step_since_round_6": "the fold of third-allele events into their buckets (k_xev_scatter / k_xev_fold: the last accumulation the host did)",
                "kernel_ms": {k: round(float(v), 4) for k, v in zip(kernel_names, kms) if k}}

        # ---- validation + 1-thread CPU baseline on a prefix of the timed contig
        cpu = None; validated = None
        ends = capi.read_ends(arrs)
        if args.mode != "sites":
            span = int((np.minimum(ends, contig_len) - np.maximum(arrs["pos"].astype(np.int64), 0)).clip(min=0).sum())
            dropped = (arrs["flag"] & 4) != 0
            validated = {"events_equal_sum_of_spans": bool(eng_events == span and not dropped.any())}
        if args.mode == "sites" and args.cpu_sample_mbp > 0:
            # a sample of the site list against the oracle, site by site: the reference's own -l loop (one fetch + pileup per line,
            # bamreadcount.cpp:574-607) vs the line cut out of the shared virtual axis
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            oracle = capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so"))
            eng.fetch_result()
            oe = capi.Engine(oracle, lib_names=names, **opts)
            pick = np.unique(np.linspace(0, len(sites) - 1, 400).astype(np.int64)); tcpu = 0.0; oev = 0
            pos64 = arrs["pos"].astype(np.int64)
            for i in pick:
                sp = int(sites[i])
                lo_i = int(np.searchsorted(pos64, sp - 2 - 1000, side="left")); hi_i = int(np.searchsorted(pos64, sp, side="left"))     # (reads are at most 1 kb long here)
                idx = lo_i + np.nonzero(ends[lo_i:hi_i] > max(sp - 2, 0))[0]
                sel = capi.select_reads(arrs, idx)
                t0c = time.perf_counter()                                     # the oracle's own work: staging, pileup, text (not the numpy selection)
                oe.begin_region(0, sp - 1, sp, ref); oe.push_reads(sel)
                oe.end_region(); want = oe.format_region("chrS"); oe.clear_indel_queue(); tcpu += time.perf_counter() - t0c; oev += oe.counts()[0]
                d = int(site_vbeg0[i]) + 1 - sp
                got = eng.format_window("chrS", sp - 1 + d, sp + d, d)
                assert got == want, "site %d (position %d): the planner's line differs from the oracle's" % (i, sp)
            oe.close()
            validated = {"sites_checked": int(len(pick)), "site_lines_byte_exact": True}
            cpu = {"value": round(oev / max(tcpu, 1e-9), 1), "unit": "pileup base-events/s", "cores": 1, "kind": "port", "sites_per_s": round(len(pick) / max(tcpu, 1e-9), 1),
                   "sample": "%d of the %d sites, one oracle region per site (staging, pileup and text; the reads handed over already decoded)" % (len(pick), len(sites))}
        elif args.cpu_sample_mbp > 0:
            import parity
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            oracle = capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so"))
            send = int(min(args.cpu_sample_mbp * 1e6 * (0.15 if per_lib else 1.0), contig_len))
            sub = capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, -1, send))
            oe = capi.Engine(oracle, lib_names=names, **opts)
            oe.begin_region(0, 0, send, ref); oe.push_reads(sub)
            t0 = time.perf_counter(); oe.upload(); oe.compute(); tc = time.perf_counter() - t0
            ores = oe.fetch_result(); otext = oe.format_region_np("chrS"); oev, _ = oe.counts()
            he = capi.Engine(hip, lib_names=names, device=local_rank, **opts)
            he.begin_region(0, 0, send, ref); he.push_reads(sub)
            hres = he.end_region(); htext = he.format_region_np("chrS")
            parity.assert_results_equal(hres, ores, "bench prefix")       # raises on the first differing plane element
            assert len(htext) == len(otext) and np.array_equal(htext, otext), "text of the HIP engine and of the oracle differ"
            text_bytes = int(len(htext)); he.close()
            # the same prefix through the device-side text path (what the command line uses): lines written by k_text_write
            # (in pieces, like the command line: a region's text is addressed with 32-bit offsets on the device)
            hd = capi.Engine(hip, lib_names=names, device=local_rank, device_text="chrS", **opts)
            sends = capi.read_ends(sub); at = 0; piece = 2_000_000; dev_pieces = 0
            for a in range(0, send, piece):
                b = min(a + piece, send)
                hd.begin_region(0, a, b, ref); hd.push_reads(capi.select_reads(sub, capi.fetch_overlapping(sub, sends, a - 1, b)))
                r = hd.end_region()
                dev_pieces += 0 if r.ncol.any() else 1      # (deep indel-rich regions are routed to the host formatter by the engine)
                if a > 0: hd.clear_indel_queue()          # an internal piece boundary ("Large regions")
                dtext = hd.format_region_np("chrS")
                assert np.array_equal(dtext, otext[at:at + len(dtext)]), "device-side text and the oracle's text differ in piece %d" % (a // piece)
                at += len(dtext)
            assert at == len(otext), "device-side text is shorter than the oracle's"
            hd.close(); oe.close()
            validated = dict(validated or {}, prefix_mbp=send / 1e6, planes_bit_exact=True, text_byte_exact=True, device_text_byte_exact=True, device_text_pieces=dev_pieces, text_bytes=text_bytes)
            cpu = {"value": round(oev / tc, 1), "unit": "pileup base-events/s", "cores": 1, "kind": "port",
                   "sample": "first %.2f Mbp of the same contig (%d events, %.1f s), C oracle incl. its text formatting, 1 thread of %d host cores"
                             % (send / 1e6, oev, tc, ncpu)}
            # the reference's OWN code (fetch_func / pileup_func / BasicStat / IndelQueue compiled unmodified over the htslib shim,
            # oracle/_ref — prebuilt, it travels with the snapshot) on a smaller prefix: std::map + stringstream per position and a
            # text-encoded tag parsed per event make it several times slower than the C restatement
            ref_so = os.path.join(ROOT, "oracle", "_ref", "libbamrc_ref.so")
            if os.path.exists(ref_so) and args.cpu_ref_mbp > 0:
                rsend = int(min(args.cpu_ref_mbp * 1e6 * (0.15 if per_lib else 1.0), send))
                rsub = capi.select_reads(arrs, capi.fetch_overlapping(arrs, ends, -1, rsend))
                o2 = capi.Engine(oracle, lib_names=names, **opts)
                o2.begin_region(0, 0, rsend, ref); o2.push_reads(rsub); o2.end_region(); want_t = o2.format_region_np("chrS").copy(); rev, _ = o2.counts(); o2.close()      # (a view of the engine's buffer: copy before closing)
                rl = capi.Library(ref_so)
                re_ = capi.Engine(rl, lib_names=names, **opts)
                re_.begin_region(0, 0, rsend, ref); re_.push_reads(rsub)
                t0 = time.perf_counter(); re_.upload(); re_.compute(); tr = time.perf_counter() - t0
                re_.fetch_result(); got_t = re_.format_region_np("chrS")
                assert len(got_t) == len(want_t) and np.array_equal(got_t, want_t), "the reference-compiled library and the oracle print different text"
                re_.close()
                cpu["reference_compiled"] = {"value": round(rev / tr, 1), "unit": "pileup base-events/s", "cores": 1, "kind": "reference",
                                             "sample": "first %.2f Mbp of the same contig (%d events, %.1f s): bam-readcount's own fetch_func / pileup_func / BasicStat / IndelQueue sources "
                                                       "compiled unmodified over the samtools/htslib shim of oracle/ref_shim, text identical to the oracle's" % (rsend / 1e6, rev, tr)}
            if pool is not None:
                # ---- the WHOLE timed region against the oracle on all usable cores: the oracle computes every window as a region of
                # its own (reads fetched the reference's way), the HIP side reads the same windows back from the result of the timed
                # region (brc_fetch_window, no second computation); digests of every plane, of the indel list and of the text
                import fullcheck
                depth = 200.0 / 30.0 if per_lib else 1.0
                nwin = max(3 * len(pool.procs), int(contig_len * depth / 1.0e6))
                full = fullcheck.check_region(pool, eng, parity, fullcheck.windows_of(0, contig_len, nwin), want_events=eng_events, with_text=(full_check != 2))
                validated = dict(validated or {}, **full)
                oc = full["oracle_cpu_seconds"]; npr = full["oracle_processes"]
                cpu["all_cores"] = {"value": round(full["events"] / max(oc["pileup"] / npr, 1e-9), 1), "cores": npr,
                                    "value_incl_text": round(full["events"] / max((oc["pileup"] + oc["text"]) / npr, 1e-9), 1),
                                    "sample": "the whole timed region (%d events) as %d abutting oracle regions on %d processes at once (the CPUs this container may use: %d hardware threads, "
                                              "cgroup quota %d): events / (summed oracle pileup seconds / processes); the same run is the whole-region validation"
                                              % (full["events"], full["windows"], npr, ncpu, ncpu_eff)}
        # ---- end to end through the drop-in command line (BAM decode + PCIe + text)
        e2e = None
        if args.e2e_mbp > 0 and world == 1 and os.path.exists(CLI):
            import tempfile
            n = int(args.e2e_mbp * 1e6)
            d = tempfile.mkdtemp(prefix="brc_e2e_")
            r2, a2 = synthgen.generate(n, "wgs30x", seed=3)
            synthgen.write_bam(os.path.join(d, "syn.bam"), "chrS", n, a2)           # BGZF level 1 + .bai (tools/bam_write.c)
            rows = (n + 59) // 60
            pad = np.full(rows * 60, 10, np.uint8); pad[:n] = r2
            with open(os.path.join(d, "syn.fa"), "wb") as f:
                f.write(b">chrS\n" + np.concatenate([pad.reshape(rows, 60), np.full((rows, 1), 10, np.uint8)], axis=1).tobytes())
            open(os.path.join(d, "syn.fa.fai"), "w").write("chrS\t%d\t6\t60\t61\n" % n)
            e2 = capi.read_ends(a2)
            ev2 = int((np.minimum(e2, n) - a2["pos"].astype(np.int64)).clip(min=0).sum())

            def timed(region):
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    p = subprocess.run([CLI, "-w", "0", "-q", "20", "-b", "13", "-f", "syn.fa", "syn.bam", region], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    t1 = time.perf_counter() - t0
                    if p.returncode == 0 and (best is None or t1 < best):
                        best = t1
                return best
            best, start = timed("chrS"), timed("chrS:1-1000")
            if best:
                e2e = {"value": round(ev2 / best, 1), "unit": "pileup base-events/s", "seconds": round(best, 3), "events": ev2,
                       "startup_seconds": round(start, 3) if start else None,
                       "what": "bam-readcount (this repository's drop-in CLI) -w0 -q20 -b13 -f syn.fa syn.bam chrS > /dev/null on a %.0f-Mbp 30x synthetic BAM: process start and HIP "
                               "initialisation (startup_seconds: the same command on a 1-kb region), BGZF/BAM decode, H2D, device pipeline, D2H, text formatting (about %.1f GB of text)"
                               % (args.e2e_mbp, 0.3586 * args.e2e_mbp)}
            import shutil
            shutil.rmtree(d, ignore_errors=True)

        # ---- the engine apart from the BAM decoder: decoded batches in host memory -> text in host memory through the C-ABI
        # (begin -> push -> upload -> compute -> device-side text -> download -> the host's line patches), piece by piece on ONE
        # engine, nothing overlapped (the command line overlaps decode, engine and formatting of consecutive pieces on threads)
        abi = None
        if args.abi_mbp > 0 and world == 1 and args.mode == "weak" and not per_lib:
            piece = 2_000_000; alen = int(min(args.abi_mbp * 1e6, contig_len))
            cuts = [(a, min(a + piece, alen)) for a in range(0, alen, piece)]
            pos64 = arrs["pos"].astype(np.int64)
            batches = [capi.select_reads(arrs, fullcheck_window_reads(capi, arrs, ends, pos64, a, b)) for a, b in cuts]      # (untimed: the caller's decoded reads)
            ha = capi.Engine(hip, device=local_rank, device_text="chrS", **opts)
            def abi_pass():
                tb = 0; ev = 0
                for (a, b), sub in zip(cuts, batches):
                    ha.begin_region(0, a, b, ref); ha.push_reads(sub); ha.end_region()
                    if a > 0: ha.clear_indel_queue()
                    tb += len(ha.format_region_np("chrS")); ev += ha.counts()[0]
                return tb, ev
            abi_pass()                                                  # (first pass: pinned buffers are allocated)
            t0a = time.perf_counter(); tb, eva = abi_pass(); ta = time.perf_counter() - t0a
            ha.close()
            in_bytes = int(sum(synthgen.algorithmic_bytes(bt, 0, 1, 0, ref_positions=0)[0] for bt in batches))
            abi = {"value": round(eva / ta, 1), "unit": "pileup base-events/s", "seconds": round(ta, 3), "events": int(eva), "pieces": len(cuts),
                   "h2d_bytes": in_bytes, "d2h_text_bytes": int(tb), "pcie_GBs": round((in_bytes + tb) / ta / 1e9, 2),
                   "what": "first %.0f Mbp of the timed contig through the C-ABI from decoded host batches to host text: brc_begin_region, brc_push_reads (staging copy into pinned "
                           "memory), brc_end_region (H2D, kernels, device-side text, D2H), brc_format_region (the host's line patches), %d pieces of 2 Mbp one after the other on one "
                           "engine, nothing overlapped, second pass timed" % (alen / 1e6, len(cuts))}

        # ---- the other timed BASELINE configurations, each a sub-process of this script with a short timed region; their
        # own validation (config 4: 400 sites against one oracle region per site; config-5 shape: planes + text of a prefix
        # against the oracle) runs inside them
        other = None
        want_other = args.other_configs == 1 or (args.other_configs < 0 and world == 1 and args.mode == "weak" and config == "wgs30x" and abs(args.contig_mbp - 50.0) < 1e-9 and args.cpu_sample_mbp > 0)
        if want_other:
            eng.close(); eng = None                              # (the sub-processes get the GPU and the host memory to themselves)
            other = {}
            subs = {"config4_sites": ["--mode", "sites"],
                    "config5_per_gpu_shape": ["--mode", "strong", "--contig-mbp", "6.25"],
                    # config 3's data model with mixed read lengths (30 % trimmed to U[100,149], 10 % 250 bases): what the one-modal-length
                    # fast path of k_pileup2 costs on reads of other lengths
                    "mixed_lengths": ["--mode", "weak", "--config", "wgs30x_mixed"],
                    # NovaSeq-like records: 151 bases, binned qualities, a quarter of the reads adapter-trimmed, 10 % soft clips, duplicates,
                    # secondary / supplementary records, MAPQ-0 multimappers (tools/synthgen.py: "novaseq")
                    "novaseq": ["--mode", "weak", "--config", "novaseq"],
                    # config 5 whole on ONE GPU (50 Mbp, 200x, 4 libraries: 67 M reads, 10 G events per step); every position of it
                    # against the oracle on all cores too — planes only: its text would be 72 GB
                    "config5_full_one_gpu": ["--mode", "strong", "--contig-mbp", "50", "--full-check", "2", "--steps", "10", "--warmup", "2"]}
            for key, extra in subs.items():
                cmd = [sys.executable, os.path.abspath(__file__), "--steps", "60", "--warmup", "5", "--cpu-ref-mbp", "0", "--e2e-mbp", "0", "--abi-mbp", "0", "--other-configs", "0"] + extra
                try:
                    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
                    sub = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])
                    other[key] = {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "steps": sub["steps"], "positions_per_s": sub["positions_per_s"],
                                  "workload": sub["config"]["workload"], "events_per_step": sub["config"]["events_per_step"], "positions_per_step": sub["config"]["positions_per_step"],
                                  "kernel_ms": sub["roofline"]["kernel_ms"], "frac": sub["roofline"]["frac"], "achieved_GBs": sub["roofline"]["achieved"],
                                  "per_kernel": sub["roofline"]["per_kernel"], "validated": sub["validated"], "cpu_baseline": sub["cpu_baseline"]}
                except Exception as ex:                                      # noqa: BLE001 — reported, never hidden
                    other[key] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        # ---- BASELINE configs 4 and 5 through the drop-in CLI itself: multi-contig BAM + BAI, the reference's own -l loop and -p -i
        # lookups (tools/e2e_configs.py validates against oracle/_ref/bam-readcount-ref, the reference's own main())
        e2e_sites = e2e_tumor = None; e2e_sharded = None
        dist_backend = dist.get_backend() if dist is not None else None
        want_e2e = (args.e2e_configs == 1 or (args.e2e_configs < 0 and (want_other or (world > 1 and args.mode == "weak" and config == "wgs30x")))) and os.path.exists(CLI) and not args.force_dist
        if want_e2e and world > 1:
            # an N-rank run: the other ranks are done (their engines close, their processes end) — the drop-in's own one-rank-per-GPU
            # launcher gets the node's GPUs to itself; nothing below is a collective
            if eng is not None:
                eng.close(); eng = None
            dist.destroy_process_group(); dist = None
        if want_e2e:
            tool = os.path.join(ROOT, "tools", "e2e_configs.py")
            # the drop-in as N processes, one per GPU (cli.cpp: --brc-ranks): on a one-GPU run two ranks share GPU 0 (what separate
            # processes do to the host-bound command line); on an N-GPU run rank r owns GPU r — strong scaling of configs 4 and 5 end to end
            rk = ["--ranks", str(world), "--rank-devices", ",".join("0" if share_gpu0 else str(i) for i in range(world))] if world > 1 else ["--ranks", "2", "--rank-devices", "0,0"]
            def e2e_leg(extra):
                try:
                    out = subprocess.run([sys.executable, tool] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
                    if out.returncode != 0:
                        return {"error": out.stderr.decode(errors="replace")[-600:]}
                    return json.loads(out.stdout.decode().strip().splitlines()[-1])
                except Exception as ex:                                  # noqa: BLE001 — reported, never hidden
                    return {"error": "%s: %s" % (type(ex).__name__, ex)}
            e2e_sites = e2e_leg(["--leg", "sites", "--contigs", "8", "--contig-mbp", str(args.e2e_sites_mbp), "--check-lines", "1000"] + rk)
            e2e_tumor = e2e_leg(["--leg", "tumor", "--contig-mbp", str(args.e2e_tumor_mbp if args.e2e_tumor_mbp > 0 else 6.25 * min(world, 4)), "--check-mbp", str(min(1.0, args.e2e_tumor_mbp / 4) if args.e2e_tumor_mbp > 0 else 1.0)] + rk)
            e2e_sharded = {"sites": e2e_sites.pop("sharded", None) if isinstance(e2e_sites, dict) else None, "tumor": e2e_tumor.pop("sharded", None) if isinstance(e2e_tumor, dict) else None}
        what = {"weak": "synthetic 30x WGS, 150bp reads, 1 contig %.0f Mbp per GPU, -q20 -b13" % (contig_len / 1e6),
                "strong": "synthetic 200x tumor 4 libraries, 150bp reads, -p -i, 1 contig %.0f Mbp cut into %d intervals" % (total_len / 1e6, world),
                "sites": "-l site list of %d single-base sites in file order over %d synthetic 30x contigs of %.0f Mbp (genome scaled from 3.1 Gbp), -q20 -b13, cut into %d slices"
                         % (args.sites, world, contig_len / 1e6, world)}[args.mode]
        if config == "wgs30x_mixed":
            what = "synthetic 30x WGS, MIXED read lengths (60 %% 150 bp, 30 %% trimmed to U[100,149], 10 %% 250 bp), 1 contig %.0f Mbp per GPU, -q20 -b13" % (contig_len / 1e6)
        if config == "novaseq":
            what = "synthetic 30x, NovaSeq-like reads (151 bp; quality bins 2 / 12 / 23 / 37; 25 %% adapter-trimmed to U[35,150]; 10 %% soft-clipped; 8 %% duplicates, 1 %% secondary, 1 %% supplementary; 5 %% MAPQ 0), 1 contig %.0f Mbp per GPU, -q20 -b13 — not one of BASELINE's configurations" % (contig_len / 1e6)
        if config == "long10k":
            what = "synthetic 30x, 10-kb reads (30 %% with an insertion, 30 %% with a deletion), 1 contig %.0f Mbp per GPU, -q20 -b13 — not one of BASELINE's configurations" % (contig_len / 1e6)
        if config == "hifi_eqx":
            what = "synthetic 30x, 10-20-kb reads aligned with --eqx (match runs as = and X, no M operator; an insertion or a deletion every ~150 bases, a substitution every ~300: %.0f CIGAR operators per read; HiFi / pbmm2-like), 1 contig %.0f Mbp per GPU, -q20 -b13 — not one of BASELINE's configurations" % (float(arrs["n_cigar"].mean()), contig_len / 1e6)
        if config in ("ont", "ont_ul"):
            what = "synthetic 30x, %s reads with an insertion or a deletion every ~15 bases (%.0f CIGAR operators per read; ONT / CLR-like), 1 contig %.0f Mbp per GPU, -q20 -b13 — not one of BASELINE's configurations" % ("3-10-kb" if config == "ont" else "30-100-kb", float(arrs["n_cigar"].mean()), contig_len / 1e6)
        if config == "tumor200x" and args.mode == "weak":
            what = "synthetic 200x tumor 4 libraries, 150bp reads, -p -i, 1 contig %.2f Mbp per GPU" % (contig_len / 1e6)
        if rank_check is not None:
            validated = dict(validated or {}, all_ranks_ok=rank_check["all_ranks_ok"], per_rank_check=rank_check["what"], rank0_error=rank_check["error"],
                             distributed_backend=dist_backend)
        line = {
            "metric": "pileup base-events/sec", "value": round(value, 1), "unit": "events/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if args.mode == "weak" else "strong", "vs_baseline": None, "dtype": "u32+f32", "data": "synthetic" + (" (TEST RUN: every rank on GPU 0, process group on gloo — BRC_BENCH_SHARE_GPU0)" if share_gpu0 else ""),
            "config": {"workload": what, "mode": args.mode, "reads_per_gpu": int(len(region_reads["pos"])), "piece_steps": piece_steps,
                       "events_per_step": int(ev_total), "positions_per_step": int(pos_total), "parallelism": "interval-shard x%d" % world},
            "positions_per_s": round(pos_total * args.steps / tmax, 1),
            "per_rank": per_rank, "per_gpu_value": round(value / world, 1),
            "roofline": roof, "cpu_baseline": cpu, "e2e": e2e, "e2e_sites": e2e_sites, "e2e_tumor": e2e_tumor, "e2e_sharded": e2e_sharded, "abi_roundtrip": abi, "validated": validated, "other_configs": other,
            "host": {"gen_s": round(t_gen, 2), "push_s": round(t_push, 2), "upload_s": round(t_up, 2), "timed_s": round(tmax, 3)},
        }
        print(json.dumps(line))
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank_check is not None and not rank_check["all_ranks_ok"]:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
