#!/usr/bin/env python3
"""The scenarios of tools/fuzz/extreme.py written twice (test infrastructure) — as BAM (tools/bamio.py, CIGARs and NM normalised the way a
CRAM round trip leaves them) and as CRAM 3.0 (tools/cramio.py: multi-container, every block method and integer codec at random, with and
without .crai) — and run through the drop-in command line (CPU lane simulator): the two inputs must print the same.  Both sides use the
product's own readers, so this checks the CRAM reader against the BAM reader, not against htslib.

    FIRST=500000 COUNT=12 python tools/fuzz/extreme_cram.py"""
import sys,os,time,subprocess,tempfile,shutil
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('tools/fuzz','tests','tools','.'): sys.path.insert(0,os.path.join(ROOT,p))
sys.argv=sys.argv[:1]
import extreme, numpy as np, bamio, cramio
from test_cli import SIM_CLI, _write_fasta, _cram_nm
first=int(os.environ.get("FIRST","500000")); want_n=int(os.environ.get("COUNT","12"))
bad=0; done=0; seed=first; t0=time.time()
while done+bad<want_n:
    rng0=np.random.default_rng(seed)
    kind0=str(rng0.choice(["long", "deep", "libs", "thresholds", "tiny", "mixed_len", "dense_indel", "spliced", "spliced"]))
    if kind0 not in ("spliced","mixed_len","dense_indel"): seed+=1; continue
    kind, style, ref, arrs, regions, kw, clear = extreme.scenario(seed); seed+=1
    if len(ref) > 700_000 or len(arrs["pos"]) > 1500: continue
    d=tempfile.mkdtemp(prefix="xcram_")
    try:
        nl=len(kw.get("lib_names",[])) or 1
        ids=["rg%d"%i for i in range(max(nl, int(np.max(arrs["lib"]))+1))]
        lines=["@RG\tID:%s\tLB:lib%03d\tSM:s"%(ids[i],i) for i in range(len(ids))]
        rgs=[ids[int(l)] if l>=0 else None for l in arrs["lib"]]
        tids=np.zeros(len(arrs["pos"]),int)
        norm=dict(arrs); cig=[]; ncs=[]
        for i in range(len(arrs["pos"])):
            ops=[]
            for c in arrs["cigar"][int(arrs["cigar_off"][i]):int(arrs["cigar_off"][i])+int(arrs["n_cigar"][i])]:
                op,ln=int(c)&15,int(c)>>4
                if op in (7,8): op=0
                if ops and ops[-1][0]==op: ops[-1][1]+=ln
                else: ops.append([op,ln])
            cig+=[(ln<<4)|op for op,ln in ops]; ncs.append(len(ops))
        norm["cigar"]=np.array(cig,np.uint32); norm["n_cigar"]=np.array(ncs,np.uint32)
        norm["cigar_off"]=np.concatenate([[0],np.cumsum(ncs)[:-1]]).astype(np.uint64)
        norm["tags"]=arrs["tags"].copy(); norm["nm"]=arrs["nm"].copy()
        for i in range(len(arrs["pos"])):
            if (int(arrs["tags"][i])&1) or (int(arrs["flag"][i])&4): continue
            norm["tags"][i]|=1; norm["nm"][i]=_cram_nm(norm,i,ref)
        contigs=[("chrA",len(ref))]
        bamio.write_bam(os.path.join(d,"m.bam"),contigs,norm,tids,rg_of_read=rgs,rg_lines=lines)
        meth=(0,1) if seed%2 else (4,5,0,1,2,3,5)
        cramio.write_cram(os.path.join(d,"x.cram"),contigs,arrs,tids,[ref],rg_of_read=rgs,rg_lines=lines,per_container=int(rng0.integers(50,400)),methods=meth,int_codecs=bool(seed%3==0),write_crai=bool(seed%2))
        _write_fasta(os.path.join(d,"r.fa"),[("chrA",ref)])
        o=["-w","3","-f","r.fa","-q",str(kw["min_mapq"]),"-b",str(kw["min_bq"])]
        if kw.get("per_lib"): o.append("-p")
        if kw.get("insertion_centric"): o.append("-i")
        regs=["chrA:%d-%d"%(a+1,max(b,a+1)) for a,b in regions]
        a=subprocess.run([SIM_CLI]+o+["m.bam"]+regs,cwd=d,stdout=subprocess.PIPE,stderr=subprocess.PIPE,timeout=900)
        b=subprocess.run([SIM_CLI]+o+["x.cram"]+regs,cwd=d,stdout=subprocess.PIPE,stderr=subprocess.PIPE,timeout=900)
        ok=(a.returncode==b.returncode and a.stdout==b.stdout and a.stderr==b.stderr)
        if ok: done+=1
        else:
            bad+=1; print("FAIL seed",seed-1,kind,o,regs,"rc",a.returncode,b.returncode,"stdout eq",a.stdout==b.stdout,"stderr eq",a.stderr==b.stderr,len(a.stdout),len(b.stdout),b.stderr[-300:],flush=True)
    except Exception as ex:
        bad+=1; print("EXC seed",seed-1,kind,type(ex).__name__,str(ex)[:300],flush=True)
    finally:
        shutil.rmtree(d,ignore_errors=True)
print("cram extreme:",done,"ok",bad,"failed",round(time.time()-t0,1),"s")
