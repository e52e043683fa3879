#!/usr/bin/env python3
"""The scenarios of tools/fuzz/extreme.py at the level of the COMMAND LINE (test infrastructure): long-read, spliced and mixed-length
data written as BAM + BAI, the reference's own main() (oracle/_ref/bam-readcount-ref) against the drop-in linked to the CPU lane
simulator — or its sanitizer build (CLI=/tmp/brc_asan/bam-readcount-asan, tools/fuzz/build_asan.sh) — with small --brc-chunk pieces
so that reads of tens of kilobases and introns of 100 kb cross piece boundaries.  stdout, stderr and exit code must be identical.

    FIRST=400000 COUNT=25 [CLI=<binary>] python tools/fuzz/extreme_cli.py"""
import sys,os,time,subprocess,tempfile,shutil
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('tools/fuzz','tests','tools','.'): sys.path.insert(0,os.path.join(ROOT,p))
sys.argv=sys.argv[:1]
import extreme, numpy as np, bamio
from test_cli import SIM_CLI, _write_fasta
from test_ref_compiled import REF_CLI
first=int(os.environ.get("FIRST","400000")); want_n=int(os.environ.get("COUNT","25"))
bad=0; done=0; t0=time.time(); seed=first
while done+bad<want_n:
    rng0=np.random.default_rng(seed)
    kind0=str(rng0.choice(["long", "deep", "libs", "thresholds", "tiny", "mixed_len", "dense_indel", "spliced", "spliced"]))
    if kind0 not in ("spliced","long","mixed_len"): seed+=1; continue
    kind, style, ref, arrs, regions, kw, clear = extreme.scenario(seed); seed+=1
    if len(ref) > 1_200_000: continue
    d=tempfile.mkdtemp(prefix="xcli_")
    try:
        rng=np.random.default_rng(seed+9)
        nl=len(kw.get("lib_names",[])) or 1
        ids=["rg%d"%i for i in range(max(nl, int(np.max(arrs["lib"]))+1))]
        lines=["@RG\tID:%s\tLB:lib%d\tSM:s"%(ids[i],i) for i in range(len(ids))]
        rgs=[ids[int(l)] if l>=0 else None for l in arrs["lib"]]
        bamio.write_bam(os.path.join(d,"x.bam"), [("chrA", len(ref))], arrs, np.zeros(len(arrs["pos"]),int), rg_of_read=rgs, rg_lines=lines)
        _write_fasta(os.path.join(d,"r.fa"), [("chrA", ref)])
        o=["-w","3","-f","r.fa","-q",str(kw["min_mapq"]),"-b",str(kw["min_bq"])]
        if kw.get("per_lib"): o.append("-p")
        if kw.get("insertion_centric"): o.append("-i")
        if "max_cnt" in kw: o+=["-d",str(kw["max_cnt"])]
        regs=["chrA:%d-%d"%(a+1,max(b,a+1)) for a,b in regions]
        extra=["--brc-chunk",str(int(rng.choice([64,333,5000,100000])))]
        if rng.random()<0.5: extra+=["--brc-gpus",str(int(rng.choice([2,3])))]          # (several engines take the pieces in turn)
        if rng.random()<0.3: extra+=["--brc-plan",str(int(rng.choice([0,3])))]
        if rng.random()<0.5:
            open(os.path.join(d,"s.txt"),"w").write("".join("chrA\t%d\t%d\n"%(a+1,max(b,a+1)) for a,b in regions)); args=["-l","s.txt","x.bam"]
        else: args=["x.bam"]+regs
        a=subprocess.run([REF_CLI]+o+args,cwd=d,stdout=subprocess.PIPE,stderr=subprocess.PIPE,timeout=900)
        b=subprocess.run([os.environ.get("CLI",SIM_CLI)]+o+extra+args,cwd=d,stdout=subprocess.PIPE,stderr=subprocess.PIPE,timeout=1800,env=dict(os.environ,ASAN_OPTIONS="detect_leaks=0"))
        ok=(a.returncode==b.returncode and a.stdout==b.stdout and a.stderr==b.stderr)
        if ok: done+=1
        else:
            bad+=1; print("FAIL seed",seed-1,kind,o,extra,args,"rc",a.returncode,b.returncode,"stdout eq",a.stdout==b.stdout,"stderr eq",a.stderr==b.stderr,len(a.stdout),len(b.stdout),b.stderr[-300:],flush=True)
    except Exception as ex:
        bad+=1; print("EXC seed",seed-1,kind,type(ex).__name__,str(ex)[:300],flush=True)
    finally:
        shutil.rmtree(d,ignore_errors=True)
print("cli extreme:",done,"ok",bad,"failed",round(time.time()-t0,1),"s")
