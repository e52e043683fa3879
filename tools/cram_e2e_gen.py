"""Inputs of tools/gpu_r5_cram_e2e.sh: <Mbp> of the wgs30x model as syn.bam, syn.fa and (tools/cramio.py, ~110 us per read) syn.cram + .crai, in the current directory."""
import sys, os, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, synthgen, cramio
n = int(float(sys.argv[1]) * 1e6)
ref, a = synthgen.generate(n, "wgs30x", seed=3)
synthgen.write_bam("syn.bam", "chrS", n, a)
synthgen.write_fasta("syn.fa", [("chrS", ref)])
t0 = time.time()
refa = np.frombuffer(ref, np.uint8) if isinstance(ref, (bytes, bytearray)) else np.asarray(ref, np.uint8)
rgs = None
cramio.write_cram("syn.cram", [("chrS", n)], a, np.zeros(len(a["pos"]), int), [refa], per_container=10000, methods=(1,), write_crai=True)
print("cram written in %.0f s, %d reads, %.1f MB (bam %.1f MB)" % (time.time() - t0, len(a["pos"]), os.path.getsize("syn.cram") / 1e6, os.path.getsize("syn.bam") / 1e6))
