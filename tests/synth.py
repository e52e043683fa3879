"""Seeded synthetic read batches for parity tests (numpy; small sizes).  Produces brc_read_batch arrays."""
import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
CODE = {c: i for i, c in enumerate(NT16)}


def make_ref(rng, length, weird=0.0):
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=length)
    if weird > 0:
        m = rng.random(length) < weird
        ref[m] = rng.choice(np.frombuffer(b"acgtNnRYMKswBDHV", np.uint8), size=int(m.sum()))
    return ref


def dense_cigar(rng, L):
    """Long-read style: match runs of about a dozen bases separated by 1-3 base insertions / deletions, now and then a
    reference skip; consumes exactly L query bases."""
    ops = []; rem = L
    if rng.random() < 0.3 and rem > 30:
        s = int(rng.integers(1, 25)); ops.append((4, s)); rem -= s
    while rem > 0:
        m = min(rem, int(rng.geometric(0.08))); ops.append((0, m)); rem -= m
        if rem <= 1:
            if rem == 1: ops.append((0, 1)); rem = 0
            break
        k = rng.random()
        if k < 0.45:
            i = min(rem - 1, int(rng.integers(1, 4))); ops.append((1, i)); rem -= i
        elif k < 0.9:
            ops.append((2, int(rng.integers(1, 4))))
        else:
            ops.append((3, int(rng.integers(5, 200))))
    out = []
    for o, l in ops:
        if out and out[-1][0] == o: out[-1] = (o, out[-1][1] + l)
        else: out.append((o, l))
    if out[-1][0] != 0:                     # end on a match: take the base from an earlier run
        out.append((0, 1))
        for i, (o, l) in enumerate(out[:-1]):
            if o == 0 and l > 1:
                out[i] = (o, l - 1); break
    assert sum(l for o, l in out if o in (0, 1, 4)) == L
    return out


def many_cigar(rng, L):
    """Reads with MANY operators of every regular kind, for the wave-form annotator (k_annotate_wave): match runs of a few bases (now and
    then one base, now and then a hundred) between insertions, deletions, skips, a deletion followed by an insertion and the reverse;
    hard / soft clips at either end, an insertion or a deletion as the first aligned operator.  Consumes exactly L query bases."""
    ops = []; rem = L
    if rng.random() < 0.2: ops.append((5, int(rng.integers(1, 9))))
    if rng.random() < 0.4 and rem > 40:
        s = int(rng.integers(1, 30)); ops.append((4, s)); rem -= s
    tail = 0
    if rng.random() < 0.4 and rem > 40:
        tail = int(rng.integers(1, 30)); rem -= tail
    r = rng.random()
    if r < 0.1 and rem > 10:
        i = int(rng.integers(1, 4)); ops.append((1, i)); rem -= i                  # insertion before the first match
    elif r < 0.2:
        ops.append((2, int(rng.integers(1, 5))))                                   # deletion before the first match: a column-only piece
    mean = float(rng.choice([2.0, 6.0, 15.0]))
    while rem > 0:
        m = int(rng.geometric(1.0 / mean)) if rng.random() > 0.03 else int(rng.integers(60, 200))
        m = max(1, min(rem, m)); ops.append((0, m)); rem -= m
        if rem <= 1:
            if rem == 1: ops.append((0, 1)); rem = 0
            break
        k = rng.random()
        if k < 0.35:
            i = min(rem - 1, int(rng.integers(1, 4))); ops.append((1, i)); rem -= i
        elif k < 0.7:
            ops.append((2, int(rng.integers(1, 4))))
        elif k < 0.78:
            ops.append((3, int(rng.integers(5, 120))))
        elif k < 0.9:
            ops.append((2, int(rng.integers(1, 3)))); i = min(rem - 1, int(rng.integers(1, 3))); ops.append((1, i)); rem -= i   # D then I
        else:
            i = min(rem - 1, int(rng.integers(1, 3))); ops.append((1, i)); rem -= i; ops.append((2, int(rng.integers(1, 3))))   # I then D
    out = []
    for o, l in ops:
        if out and out[-1][0] == o: out[-1] = (o, out[-1][1] + l)
        else: out.append((o, l))
    if out[-1][0] != 0:                     # end on a match: take the base from an earlier run
        out.append((0, 1))
        for i, (o, l) in enumerate(out[:-1]):
            if o == 0 and l > 1:
                out[i] = (o, l - 1); break
        else:
            out.pop()
            while out and out[-1][0] != 0: o, l = out.pop(); tail += l if o == 1 else 0
    if tail: out.append((4, tail))
    if rng.random() < 0.2: out.append((5, int(rng.integers(1, 9))))
    assert sum(l for o, l in out if o in (0, 1, 4)) == L, (out, L)
    return out


def random_cigar(rng, L, style):
    """Return list of (op,len) consuming exactly L query bases.  style: 'simple' | 'indel' | 'wild' | 'dense' | 'many'."""
    if style == "dense" and L >= 8:
        return dense_cigar(rng, L)
    if style == "many" and L >= 60:
        return many_cigar(rng, L)
    if style == "simple" or L < 8:
        return [(0, L)]
    ops = []
    rem = L
    if style == "wild" and rng.random() < 0.15:
        ops.append((5, int(rng.integers(1, 10))))             # H
    if rng.random() < (0.3 if style == "wild" else 0.08):
        s = int(rng.integers(1, min(20, rem - 4))); ops.append((4, s)); rem -= s
    tail_s = 0
    if rng.random() < (0.3 if style == "wild" else 0.08) and rem > 8:
        tail_s = int(rng.integers(1, min(20, rem - 4))); rem -= tail_s
    nseg = int(rng.integers(1, 5 if style == "wild" else 3))
    first = True
    while rem > 0:
        if nseg <= 1 or rem < 4:
            ops.append((int(rng.choice([0, 0, 0, 7, 8])) if style == "wild" else 0, rem)); rem = 0
            break
        m = int(rng.integers(1, rem - 1))
        ops.append((int(rng.choice([0, 0, 0, 7, 8])) if style == "wild" else 0, m)); rem -= m
        kind = rng.random()
        if kind < 0.4:
            k = int(rng.integers(1, min(6, rem))); ops.append((1, k)); rem -= k            # I
        elif kind < 0.8:
            ops.append((2, int(rng.integers(1, 6))))                                        # D
        elif style == "wild":
            r = rng.random()
            if r < 0.4:
                ops.append((3, int(rng.integers(1, 30))))                                   # N
            elif r < 0.7 and rem > 2:
                ops.append((6, int(rng.integers(1, 3))))                                    # P then I
                k = int(rng.integers(1, min(4, rem))); ops.append((1, k)); rem -= k
            else:
                ops.append((2, int(rng.integers(1, 4))))
                if rem > 2 and rng.random() < 0.5:
                    k = int(rng.integers(1, min(3, rem))); ops.append((1, k)); rem -= k     # D then I
        else:
            ops.append((2, int(rng.integers(1, 4))))
        nseg -= 1
        first = False
    if ops[-1][0] not in (0, 7, 8):      # must end on a match segment before the optional clip
        ops.append((0, 1)); # steal one base from a previous M if possible
        for i, (o, l) in enumerate(ops[:-1]):
            if o in (0, 7, 8) and l > 1:
                ops[i] = (o, l - 1); break
        else:
            ops.pop()
    if tail_s:
        ops.append((4, tail_s))
    if style == "wild" and rng.random() < 0.1:
        ops.append((5, int(rng.integers(1, 10))))
    # merge adjacent identical ops
    out = []
    for o, l in ops:
        if out and out[-1][0] == o:
            out[-1] = (o, out[-1][1] + l)
        else:
            out.append((o, l))
    q = sum(l for o, l in out if o in (0, 1, 4, 7, 8))
    assert q == L, (out, q, L)
    return out


def make_batch(seed, ref, n_reads, read_len=(30, 120), style="indel", n_libs=1, p_nolib=0.0, p_flagdrop=0.02,
               p_nonm=0.1, p_sm=0.5, p_q2tail=0.3, mismatch=0.02, region=None, p_iupac_read=0.005):
    """Coordinate-sorted batch over `ref` (uint8 array).  Returns dict of arrays (see capi.BATCH_DTYPES)."""
    rng = np.random.default_rng(seed)
    RL = len(ref)
    lo, hi = region if region else (0, RL)
    starts = np.sort(rng.integers(max(lo - 60, 0), max(hi - 1, lo + 1), size=n_reads))
    pos, flag, mapq, lib, lq, ncig, nm, sm, tags = [], [], [], [], [], [], [], [], []
    cig_all, seq_all, qual_all, cig_off, seq_off, qual_off = [], [], [], [], [], []
    co = so = qo = 0
    for s in starts:
        L = int(rng.integers(read_len[0], read_len[1] + 1))
        st = style if style != "mixed" else str(rng.choice(["simple", "simple", "indel", "wild"]))
        cg = random_cigar(rng, L, st)
        span = sum(l for o, l in cg if o in (0, 2, 3, 7, 8))
        # build the read sequence from the reference with mismatches
        seq = np.empty(L, np.uint8)
        qp = 0; rp = int(s); nmv = 0
        for o, l in cg:
            if o in (0, 7, 8):
                for j in range(l):
                    rc = chr(ref[rp + j]).upper() if rp + j < RL else "N"
                    b = CODE.get(rc, 15)
                    if rng.random() < mismatch:
                        b = int(rng.choice([1, 2, 4, 8])); nmv += int(NT16[b] != rc)
                    seq[qp + j] = b
                qp += l; rp += l
            elif o in (1, 4):
                seq[qp:qp + l] = rng.choice([1, 2, 4, 8], size=l); qp += l
                if o == 1:
                    nmv += l
            elif o in (2, 3):
                rp += l
                if o == 2:
                    nmv += l
        iu = rng.random(L) < p_iupac_read
        seq[iu] = rng.choice([0, 3, 5, 15, 15], size=int(iu.sum()))
        q = np.clip(np.rint(rng.normal(30, 8, L)), 3, 41).astype(np.uint8)
        if rng.random() < p_q2tail:
            k = int(rng.integers(1, max(2, L // 3)))
            if rng.random() < 0.5:
                q[-k:] = 2
            else:
                q[:k] = 2
        if rng.random() < 0.02:
            q[:] = 2
        f = int(rng.choice([99, 147, 83, 163, 65, 129, 121, 0, 16, 97, 145]))
        if rng.random() < p_flagdrop:
            f |= int(rng.choice([4, 256, 512, 1024]))
        if rng.random() < 0.01:
            f |= 2048
        pos.append(int(s)); flag.append(f)
        mapq.append(int(rng.choice([60, 60, 60, 47, 29, 13, 0, 255, int(rng.integers(0, 60))])))
        lib.append(-1 if rng.random() < p_nolib else int(rng.integers(0, max(1, n_libs))))
        lq.append(L); ncig.append(len(cg))
        t = 0
        if rng.random() >= p_nonm:
            t |= 1
        if rng.random() < p_sm:
            t |= 2
        tags.append(t); nm.append(nmv if (t & 1) else 0); sm.append(int(rng.integers(0, 61)) if (t & 2) else 0)
        cig_off.append(co); seq_off.append(so); qual_off.append(qo)
        cig_all.extend([(l << 4) | o for o, l in cg]); co += len(cg)
        if L & 1:
            seq = np.append(seq, 0)
        seq_all.append(((seq[0::2] << 4) | seq[1::2]).astype(np.uint8)); so += (L + 1) // 2
        qual_all.append(q); qo += L
    return dict(pos=np.array(pos, np.int32), flag=np.array(flag, np.uint16), mapq=np.array(mapq, np.uint8),
                lib=np.array(lib, np.int16), l_qseq=np.array(lq, np.int32), n_cigar=np.array(ncig, np.uint32),
                cigar_off=np.array(cig_off, np.uint64), seq_off=np.array(seq_off, np.uint64), qual_off=np.array(qual_off, np.uint64),
                nm=np.array(nm, np.int32), sm=np.array(sm, np.int32), tags=np.array(tags, np.uint8),
                cigar=np.array(cig_all, np.uint32), seq4=np.concatenate(seq_all) if seq_all else np.zeros(0, np.uint8),
                qual=np.concatenate(qual_all) if qual_all else np.zeros(0, np.uint8))


def pile_indels(arrs, x, seed=0, frac=0.9):
    """Rewrite the CIGAR of (most of) the single-operator reads that span reference position x so that they carry the SAME
    deletion (3 bases) or an insertion (2 bases) right after x: one (position, library) indel key with hundreds of events."""
    rng = np.random.default_rng(seed)
    cig, off, ncs = [], [], []
    for i in range(len(arrs["pos"])):
        c = [int(v) for v in arrs["cigar"][int(arrs["cigar_off"][i]):int(arrs["cigar_off"][i]) + int(arrs["n_cigar"][i])]]
        p, L = int(arrs["pos"][i]), int(arrs["l_qseq"][i])
        a = x - p + 1                                                # bases up to and including x
        if len(c) == 1 and (c[0] & 15) == 0 and 4 <= a <= L - 6 and rng.random() < frac:
            c = [(a << 4) | 0, (3 << 4) | 2, ((L - a) << 4) | 0] if rng.random() < 0.6 else [(a << 4) | 0, (2 << 4) | 1, ((L - a - 2) << 4) | 0]
        off.append(len(cig)); ncs.append(len(c)); cig += c
    out = dict(arrs)
    out["cigar"] = np.array(cig, np.uint32); out["n_cigar"] = np.array(ncs, np.uint32); out["cigar_off"] = np.array(off, np.uint64)
    return out


def add_sequenceless_secondary(arrs, pos, span=50):
    """Insert a SECONDARY read with SEQ '*' (l_qseq 0) and CIGAR <span>M at `pos`: it sits in the pileup columns (it makes its
    library and positions print) but is never counted (bamreadcount.cpp:295-310)."""
    a = dict(arrs)
    i = int(np.searchsorted(a["pos"], pos))
    for k, v in (("pos", pos), ("flag", 256 | 16), ("mapq", 60), ("lib", 0), ("l_qseq", 0), ("n_cigar", 1), ("nm", 0), ("sm", 0), ("tags", 0),
                 ("cigar_off", len(a["cigar"])), ("seq_off", 0), ("qual_off", 0)):
        a[k] = np.insert(a[k], i, v).astype(a[k].dtype)
    a["cigar"] = np.append(a["cigar"], np.uint32((span << 4) | 0)).astype(np.uint32)
    return a


# reads with MANY operators (many_cigar): the GPU suite runs them through the wave-form annotator (k_annotate_wave)
MANY_OPS = [
    dict(seed=31, n=120, opts=dict()),
    dict(seed=32, n=120, opts=dict(min_mapq=20, min_bq=13)),
    dict(seed=33, n=150, opts=dict(insertion_centric=True)),
    dict(seed=34, n=150, opts=dict(per_lib=True, insertion_centric=True, min_bq=5), n_libs=3, p_nolib=0.05),
    dict(seed=35, n=120, opts=dict(min_bq=30), weird=0.1),
    dict(seed=36, n=100, opts=dict(per_lib=True), n_libs=2, nul=True),     # NUL characters inside the reference text (no FASTA has them: wave form against K1's serial walk)
]


def many_ops_inputs(case):
    """inputs of one MANY_OPS case: reference (IUPAC / lower-case / NUL characters on request), a batch of
    100-1200-base reads in the many_cigar style, library names, regions, whether library-less reads are in it"""
    rng = np.random.default_rng(case["seed"])
    ref = make_ref(rng, 4000, weird=case.get("weird", 0.0))
    if case.get("nul"):
        ref = ref.copy(); ref[rng.integers(0, len(ref), 12)] = 0
    n_libs = case.get("n_libs", 1)
    arrs = make_batch(case["seed"] + 100, ref, case["n"], style="many", read_len=(100, 1200), n_libs=n_libs, p_nolib=case.get("p_nolib", 0.0), p_iupac_read=0.01)
    assert int(arrs["n_cigar"].max()) > 128                      # more operators than lanes: the passes carry their cursors
    names = ["lib%c" % (65 + i) for i in range(n_libs)] if case["opts"].get("per_lib") else ()
    return ref, arrs, names, [(0, 4000), (100, 101), (700, 1500), (3990, 4300)], case.get("p_nolib", 0.0) > 0


def operator_limit_reads():
    """Hand-made reads around the wave form's limits: 1024 M operators (the four-wave instantiation's last), 1025 / 1500 / 5120 (one
    wave per workgroup), 5121 / 13500 (one wave per CU), 13501 (K1's serial walk), a P operator, = and X operators, and a plain
    seven-operator read."""
    rng = np.random.default_rng(41)
    ref = make_ref(rng, 40000)
    def read(pos, ops):
        L = sum(l for o, l in ops if o in (0, 1, 4, 7, 8))
        return pos, ops, rng.choice([1, 2, 4, 8], size=L).astype(np.uint8), rng.integers(3, 41, L).astype(np.uint8)
    def alternating(n_m, gap_op):                                 # n_m one- or two-base matches separated by one-base operators
        ops = []
        for k in range(n_m):
            ops.append((0, 1 + (k & 1)))
            if k + 1 < n_m: ops.append((gap_op if k % 3 else 2, 1))
        return ops
    rows = [read(10, alternating(1024, 1)), read(20, alternating(1025, 1)), read(30, alternating(1500, 2)), read(35, alternating(5120, 1)), read(37, alternating(5121, 2)), read(38, alternating(13500, 1)), read(39, alternating(13501, 1)),
            read(40, [(0, 5), (6, 1), (1, 2), (0, 7), (2, 1), (0, 9), (1, 1), (0, 4)]),          # P
            read(50, [(7, 5), (1, 2), (0, 7), (2, 1), (8, 3), (0, 9), (1, 1), (0, 4)]),          # = and X
            read(70, [(0, 5), (1, 2), (0, 7), (2, 1), (0, 9), (1, 1), (0, 4)])]
    a = dict(pos=[], flag=[], mapq=[], lib=[], l_qseq=[], n_cigar=[], cigar_off=[], seq_off=[], qual_off=[], nm=[], sm=[], tags=[])
    cig, seqs, quals = [], [], []; so = qo = 0
    for pos, ops, seq, q in rows:
        L = len(seq)
        a["pos"].append(pos); a["flag"].append(0); a["mapq"].append(60); a["lib"].append(0); a["l_qseq"].append(L); a["n_cigar"].append(len(ops))
        a["cigar_off"].append(len(cig)); a["seq_off"].append(so); a["qual_off"].append(qo); a["nm"].append(3); a["sm"].append(0); a["tags"].append(1)
        cig += [(l << 4) | o for o, l in ops]
        s4 = np.append(seq, 0) if L & 1 else seq
        seqs.append(((s4[0::2] << 4) | s4[1::2]).astype(np.uint8)); so += (L + 1) // 2
        quals.append(q); qo += L
    dt = dict(pos=np.int32, flag=np.uint16, mapq=np.uint8, lib=np.int16, l_qseq=np.int32, n_cigar=np.uint32, cigar_off=np.uint64, seq_off=np.uint64, qual_off=np.uint64, nm=np.int32, sm=np.int32, tags=np.uint8)
    a = {k: np.array(v, dt[k]) for k, v in a.items()}
    a["cigar"] = np.array(cig, np.uint32); a["seq4"] = np.concatenate(seqs); a["qual"] = np.concatenate(quals)
    return ref, a


def cursor_columns(pos, ops):
    """htslib's resolve_cigar2 as the iterator runs it (a cursor per read, moved by ONE reference-consuming operator when the column has left
    the current one): per column of [pos, bam_endpos) None (not in the column), ("M", qpos) or ("D", qpos).  ops: [(op, len)].  Test-side
    restatement of brc_core.h: cursor_resolve, used to keep generated CIGARs inside the reads they belong to."""
    n = len(ops); k = -1; x = pos; y = 0; out = []
    refop = (0, 2, 3, 7, 8); mop = (0, 7, 8)
    def skip(kk, yy):
        while kk < n:
            o, l = ops[kk]
            if o in refop: break
            if o in (1, 4): yy += l
            kk += 1
        return kk, yy
    rlen = sum(l for o, l in ops if o in refop)
    for col in range(pos, pos + rlen):
        if k == -1:
            if n == 1:
                if ops[0][0] in mop: k, x, y = 0, pos, 0
            else:
                x, y = pos, 0; k, y = skip(0, 0)
            if k < 0 or k >= n: out.append(None); continue
        else:
            l = ops[k][1]
            if col - x >= l:
                if k + 1 >= n: out.append(None); continue
                if ops[k][0] in mop: y += l
                x += l
                k, y = skip(k + 1, y)
                if k >= n: out.append(None); continue
        o, l = ops[k]
        out.append(("M", y + (col - x)) if o in mop else ("D", y))
    return out


def inject_empty_mops(arrs, seed=0, frac=0.5):
    """Round 6: a copy of a batch in which `frac` of the mapped reads with a CIGAR get one to three M / = / X operators of LENGTH ZERO at
    random places (in front of a deletion, behind one, between two matches, as the first operator, several in a row, ...) — htslib's cursor
    steps onto such an operator for one column (brc_core.h: cursor_resolve).  Only sets of empty operators whose columns stay inside the
    read's bases are kept (an empty operator reported at or past query offset l_qseq makes the reference read past the read's qualities:
    the engine refuses that case, tests/test_sim_parity.py)."""
    rng = np.random.default_rng(seed)
    n = len(arrs["pos"])
    new_cig = []; new_off = np.zeros(n, np.uint64); new_nc = np.zeros(n, np.uint32)
    for i in range(n):
        nc = int(arrs["n_cigar"][i]); off = int(arrs["cigar_off"][i])
        ops = [int(c) for c in arrs["cigar"][off:off + nc]]
        if nc > 0 and not (int(arrs["flag"][i]) & 4) and int(arrs["l_qseq"][i]) > 0 and rng.random() < frac:
            for _ in range(4):                                   # (a few tries per read)
                cand = list(ops)
                for _ in range(int(rng.integers(1, 4))):
                    cand.insert(int(rng.integers(0, len(cand) + 1)), int(rng.choice([0, 0, 7, 8])))      # length zero: the operator code alone
                cols = cursor_columns(int(arrs["pos"][i]), [(c & 15, c >> 4) for c in cand])
                if all(c is None or c[1] < int(arrs["l_qseq"][i]) for c in cols):
                    ops = cand; break
        new_off[i] = len(new_cig); new_nc[i] = len(ops); new_cig.extend(ops)
    out = {k: v.copy() for k, v in arrs.items()}
    out["cigar"] = np.array(new_cig, np.uint32); out["cigar_off"] = new_off; out["n_cigar"] = new_nc
    return out


def eqx_cigars(arrs, seed=0, frac=0.8, keep_m=0.3):
    """Round 6: a copy of a batch in which `frac` of the reads have their M operators cut into runs of = and X (pbmm2 / minimap2 --eqx write
    such CIGARs) — with probability keep_m a piece stays an M, so that = / X stand BESIDE M operators: fetch_func moves neither of its cursors
    on = / X (bamreadcount.cpp:133-197) and compares the M operators behind them at lagging offsets, the iterator treats all three alike."""
    rng = np.random.default_rng(seed)
    n = len(arrs["pos"])
    new_cig = []; new_off = np.zeros(n, np.uint64); new_nc = np.zeros(n, np.uint32)
    for i in range(n):
        nc = int(arrs["n_cigar"][i]); off = int(arrs["cigar_off"][i])
        ops = [int(c) for c in arrs["cigar"][off:off + nc]]
        out = []
        if rng.random() < frac:
            for c in ops:
                op, ln = c & 15, c >> 4
                if op != 0 or ln < 1:
                    out.append(c); continue
                rem = ln
                while rem > 0:
                    run = min(rem, int(rng.geometric(0.08)))
                    o = 0 if rng.random() < keep_m else (7 if rng.random() < 0.85 else 8)
                    if out and (out[-1] & 15) == o and o != 0: out[-1] += run << 4
                    else: out.append((run << 4) | o)
                    rem -= run
        else:
            out = ops
        new_off[i] = len(new_cig); new_nc[i] = len(out); new_cig.extend(out)
    res = {k: v.copy() for k, v in arrs.items()}
    res["cigar"] = np.array(new_cig, np.uint32); res["cigar_off"] = new_off; res["n_cigar"] = new_nc
    return res


def one_special_byte_per_read(arrs, rng):
    """Three reads in four get ONE special byte (in place): a quality next to the event byte's range (0, 1, 62, 63, 64, 127, 128, 255), a
    base code 0..15, or both at the same offset — any offset, the last base of the read one time in seven.  All other qualities are
    clipped into 3..45.  Returns a lower bound of the reads that now hold an escape base."""
    q = arrs["qual"].copy(); s4 = arrs["seq4"].copy()
    q[:] = np.clip(q, 3, 45)
    n_wide = 0
    for r in range(len(arrs["pos"])):
        L = int(arrs["l_qseq"][r]); kind = r % 4
        if L == 0 or kind == 0:
            continue
        at = int(rng.integers(0, L)) if r % 7 else L - 1
        if kind in (1, 3):
            v = int(rng.choice([0, 1, 62, 63, 64, 127, 128, 255])); q[int(arrs["qual_off"][r]) + at] = v
            n_wide += v == 0 or v >= 63
        if kind in (2, 3):
            code = int(rng.integers(0, 16)); b = int(arrs["seq_off"][r]) + at // 2
            s4[b] = (s4[b] & 0x0f) | (code << 4) if at % 2 == 0 else (s4[b] & 0xf0) | code
            n_wide += code not in (1, 2, 4, 8) and kind == 2
    arrs["qual"] = q; arrs["seq4"] = s4
    return n_wide
