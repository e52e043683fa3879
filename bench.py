#!/usr/bin/env python3
"""bench.py — pileup base-events/s of the MI355X readcount engine on BASELINE.json's timed configuration.

A "step" = one pass of the whole device pipeline (k_annotate -> scans -> k_tiles -> k_pileup -> indel path) over one
synthetic contig whose inputs are already resident in HBM (uploaded before the timed region); outputs (the
position-major BasicStat planes) are written to HBM.  Workload = BASELINE config 3 (synthetic 30x WGS, 150 bp reads,
-q20 -b13, one contig; SURVEY.md 8d generator).  With N ranks every rank processes its own contig of the same size
(independent genomic intervals -> no data-path collective; "weak" scaling).  value = total events of all ranks per step
/ max-over-ranks step time.

Extra objects on the JSON line:
  roofline      k_pileup (dominant kernel): algorithmic bytes per launch (B_in + B_ref + B_out) / its average
                duration measured with HIP events on the engine's stream, against the 8 TB/s HBM peak.
  cpu_baseline  the C oracle (CPU restatement of the reference semantics, 1 thread like the reference) timed on a bounded
                prefix of the same contig on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--contig-mbp", type=float, default=50.0, help="contig length per GPU (BASELINE config 3: 50)")
    ap.add_argument("--config", default="wgs30x", choices=["wgs30x", "tumor200x"])
    ap.add_argument("--cpu-sample-mbp", type=float, default=8.0, help="prefix timed with the CPU oracle (0 = skip); 8 Mbp ~ 240 M events ~ 10-20 s")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r01_traffic.json"),
                    help="JSON with the PMC-derived HBM bytes per k_pileup launch (separate rocprofv3 --pmc passes of this command)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import synthgen
    from bam_readcount_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)

    hip = capi.load_product()
    contig_len = int(args.contig_mbp * 1e6)
    per_lib = args.config == "tumor200x"
    names = ["lib%d" % i for i in range(synthgen.CONFIGS[args.config]["n_libs"])] if per_lib else ()
    opts = dict(min_mapq=20, min_bq=13) if not per_lib else dict(min_mapq=0, min_bq=0, per_lib=True, insertion_centric=True)

    t0 = time.time()
    ref, arrs = synthgen.generate(contig_len, args.config, seed=1 + 1000 * rank)
    t_gen = time.time() - t0
    eng = capi.Engine(hip, lib_names=names, device=local_rank, **opts)
    t0 = time.time()
    eng.begin_region(0, 0, contig_len, ref)
    eng.push_reads(arrs)
    t_push = time.time() - t0
    t0 = time.time()
    eng.upload()                                  # inputs resident in HBM from here on
    t_up = time.time() - t0

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.compute()
    kernel_names = hip.kernel_names()
    k_pile = kernel_names.index("k_pileup")
    kms = np.zeros(len(kernel_names))
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ms, _tot = eng.compute()                  # launches the pipeline on the engine stream and waits for it
        kms += np.array(ms)
    sync_all()
    dt = time.perf_counter() - t0
    kms /= max(args.steps, 1)
    n_events, n_positions = eng.counts()

    tmax, ev_total, pos_total = dt, n_events, n_positions
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c = torch.tensor([n_events, n_positions], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)   # the only collective: 16 bytes of counters for the metrics line
        tmax, ev_total, pos_total = float(t.item()), int(c[0].item()), int(c[1].item())

    if rank == 0:
        ms_per_step = tmax / args.steps * 1e3
        value = ev_total * args.steps / tmax
        # roofline of the dominant kernel
        res_libs = len(names) if per_lib else 1
        b_in, b_ref, b_out = synthgen.algorithmic_bytes(arrs, n_positions, res_libs, 0, ref_positions=contig_len)
        alg = b_in + b_ref + b_out
        achieved = alg / (kms[k_pile] * 1e-3) / 1e9 if kms[k_pile] > 0 else 0.0
        traffic = None
        if args.traffic_json and os.path.exists(args.traffic_json) and args.config == "wgs30x" and abs(args.contig_mbp - 50.0) < 1e-9:
            traffic = json.load(open(args.traffic_json)).get("k_pileup_hbm_bytes_per_launch")   # measured on this exact workload
        roof = {"bound": "hbm", "kernel": "k_pileup", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg, "bytes_per_event": round(alg / max(n_events, 1), 3),
                "kernel_ms": {k: round(float(v), 4) for k, v in zip(kernel_names, kms) if k}}
        # CPU baseline: the oracle on a bounded prefix, 1 thread
        cpu = None
        if args.cpu_sample_mbp > 0:
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            oracle = capi.Library(os.path.join(ROOT, "oracle", "libbrc_oracle.so"))
            send = int(min(args.cpu_sample_mbp * 1e6, contig_len))
            ends = capi.read_ends(arrs)
            idx = capi.fetch_overlapping(arrs, ends, -1, send)
            sub = capi.select_reads(arrs, idx)
            oe = capi.Engine(oracle, lib_names=names, **opts)
            oe.begin_region(0, 0, send, ref)
            oe.push_reads(sub)
            t0 = time.perf_counter()
            oe.upload(); oe.compute()
            tc = time.perf_counter() - t0
            oev, _ = oe.counts()
            oe.close()
            cpu = {"value": round(oev / tc, 1), "unit": "pileup base-events/s", "cores": 1, "kind": "port",
                   "sample": "first %.2f Mbp of the same contig (%d events, %.1f s), C oracle incl. its text formatting, 1 thread of %d host cores"
                             % (send / 1e6, oev, tc, os.cpu_count())}
        line = {
            "metric": "pileup base-events/sec", "value": round(value, 1), "unit": "events/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32+f32", "data": "synthetic",
            "config": {"workload": "synthetic %s, 150bp reads, 1 contig %.0f Mbp per GPU, %s" % (
                "30x WGS" if not per_lib else "200x tumor 4 libraries", contig_len / 1e6,
                "-q20 -b13" if not per_lib else "-p -i"), "reads_per_gpu": int(len(arrs["pos"])),
                "events_per_step": int(ev_total), "positions_per_step": int(pos_total), "parallelism": "interval-shard x%d" % world},
            "positions_per_s": round(pos_total * args.steps / tmax, 1),
            "roofline": roof, "cpu_baseline": cpu,
            "host": {"gen_s": round(t_gen, 2), "push_s": round(t_push, 2), "upload_s": round(t_up, 2)},
        }
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
