"""
Seeded synthetic KIR index and read-pair generator.

There is no ``example_index`` and no aligner in the build or bench
environment, so tests and ``bench.py`` type synthetic samples:

* ``makeIndex``  -- G backbones with SNP / deletion / insertion variants at
  sorted sites, allele membership per variant, exon blocks; can be written as
  HISAT2-format ``.snp/.link/.locus`` text (the files the hot path reads,
  ``graphkir/hisat2.py:121-180``).
* ``makeSample`` -- read pairs (150 bp mates, ~400 bp fragments, substitution
  errors) drawn from the sample's true alleles and expressed the way HISAT2
  reports them: 0-based start, CIGAR, mismatch list, inserted strings, NM, NH.
  The sample is held as flat event arrays from which both SAM text lines
  (small cases, reference/oracle input) and packed 128-byte mate records
  (device input, any size) are derived, so both routes see the same
  alignments.

Nothing here is taken from the reference; its simulator needs ART and real
IPD-KIR sequences (``research/kg_create_data.py``).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .msa2hisat import Variant

GENE_NAMES = [
    "KIR2DL1S1", "KIR2DL2", "KIR2DL3", "KIR2DL4", "KIR2DL5", "KIR2DP1", "KIR2DS2",
    "KIR2DS3", "KIR2DS4", "KIR2DS5", "KIR3DL1", "KIR3DL2", "KIR3DL3", "KIR3DP1", "KIR3DS1",
]
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)

EV_SINGLE, EV_INS, EV_DEL = 0, 1, 2


@dataclass
class SynthIndex:
    """Synthetic index: variants (sorted, with alleles/exon flags) + backbones."""

    genes: list[str]
    backbone: dict[str, np.ndarray]          # ASCII uint8
    variants: list[Variant]
    exons: dict[str, list[tuple[int, int]]]  # 0-based (start-1, end-1) like readExons
    alleles: dict[str, list[str]]            # gene -> allele names

    def write(self, prefix: str) -> None:
        """Write ``{prefix}.snp/.link/.locus`` in the reference's on-disk format."""
        with open(prefix + ".snp", "w") as f:
            for v in self.variants:
                f.write(f"{v.id}\t{v.typ}\t{v.ref}\t{v.pos}\t{v.val}\n")
        with open(prefix + ".link", "w") as f:
            for v in self.variants:
                f.write(f"{v.id}\t{' '.join(v.allele)}\n")
        with open(prefix + ".locus", "w") as f:
            for g in self.genes:
                n = len(self.backbone[g])
                ex = " ".join(f"{s + 1}-{e + 1}" for s, e in self.exons[g])
                f.write(f"{g}\t{g}\t0\t{n}\t{n}\t{ex}\t+\n")


def makeIndex(seed: int = 2022, n_genes: int = 15, len_range=(4200, 17000),
              var_range=(1000, 4000), allele_range=(30, 250),
              frac_del: float = 0.09, frac_ins: float = 0.03) -> SynthIndex:
    """Seeded synthetic index with the shape of the real KIR index."""
    rng = np.random.default_rng(seed)
    genes = [f"{n}*BACKBONE" for n in (GENE_NAMES * ((n_genes + 14) // 15))[:n_genes]]
    if len(set(genes)) != len(genes):
        genes = [f"KIRX{i:02d}*BACKBONE" for i in range(n_genes)]
    genes = sorted(genes)
    backbone, exons, alleles = {}, {}, {}
    variants: list[Variant] = []
    for g in genes:
        L = int(rng.integers(len_range[0], len_range[1] + 1))
        n_site = int(rng.integers(var_range[0], var_range[1] + 1))
        n_site = min(n_site, (L - 200) // 3)
        n_al = int(rng.integers(allele_range[0], allele_range[1] + 1))
        seq = BASES[rng.integers(0, 4, L)]
        backbone[g] = seq
        names = [f"{g.split('*')[0]}*{i + 1:03d}{int(rng.integers(0, 100)):02d}{int(rng.integers(0, 100)):02d}"
                 for i in range(n_al)]
        alleles[g] = names
        # exon blocks: 9 blocks covering ~20 % of the backbone
        cuts = np.sort(rng.choice(np.arange(100, L - 100), 18, replace=False))
        ex = []
        tot = 0
        for s, e in zip(cuts[0::2], cuts[1::2]):
            e = min(int(e), int(s) + int(0.2 * L / 9) + 1)
            if e > s + 5:
                ex.append((int(s), e))
                tot += e - s
        exons[g] = ex
        # variant sites, at least 2 apart
        sites = np.sort(rng.choice(np.arange(50, (L - 50) // 2), n_site, replace=False)) * 2
        kinds = rng.random(n_site)
        sizes = [1, 1, 1, 2, 3, 5, 10, max(1, n_al // 4), max(1, n_al // 2)]
        for i, p in enumerate(sites):
            p = int(p)
            gap = int(sites[i + 1]) - p if i + 1 < n_site else 40
            k = int(sizes[int(rng.integers(0, len(sizes)))])
            members = rng.choice(n_al, min(k, n_al), replace=False)
            if kinds[i] < frac_del and gap >= 3:
                dl = int(rng.integers(1, min(gap - 1, 25) + 1))
                variants.append(Variant(pos=p, typ="deletion", ref=g, val=dl,
                                        allele=[names[j] for j in members]))
            elif kinds[i] < frac_del + frac_ins:
                il = int(rng.integers(1, 4))
                ins = BASES[rng.integers(0, 4, il)].tobytes().decode()
                variants.append(Variant(pos=p, typ="insertion", ref=g, val=ins,
                                        allele=[names[j] for j in members]))
            else:
                refb = int(seq[p])
                alts = [int(b) for b in BASES if int(b) != refb]
                rng.shuffle(alts)
                variants.append(Variant(pos=p, typ="single", ref=g, val=chr(alts[0]),
                                        allele=[names[j] for j in members]))
                if rng.random() < 0.05:  # second alt base at the same site, disjoint alleles
                    rest = np.setdiff1d(np.arange(n_al), members)
                    if len(rest):
                        m2 = rng.choice(rest, min(k, len(rest)), replace=False)
                        variants.append(Variant(pos=p, typ="single", ref=g, val=chr(alts[1]),
                                                allele=[names[j] for j in m2]))
    variants.sort()
    for i, v in enumerate(variants):
        v.id = f"hv{i}"
        v.allele = sorted(v.allele)
        ex = exons[v.ref]
        v.in_exon = any(s <= v.pos < e or (v.typ == "deletion" and v.pos < s and v.pos + int(v.val) >= s)
                        for s, e in ex)
    return SynthIndex(genes=genes, backbone=backbone, variants=variants, exons=exons, alleles=alleles)


# --------------------------------------------------------------------------
# samples
# --------------------------------------------------------------------------
@dataclass
class SynthSample:
    """Read pairs as flat event arrays (mate m = 2*pair + {0: READ1, 1: READ2})."""

    index: SynthIndex
    gene_cn: dict[str, int]
    truth: dict[str, list[str]]
    pair_gene: np.ndarray    # int32 [n]  index into index.genes
    pair_nh: np.ndarray      # uint8 [n]
    pair_secondary: np.ndarray  # bool [n]  flag 256 pairs
    pair_qname: np.ndarray   # int64 [n]  read-name ordinal (pairs arrive name-sorted)
    pos0: np.ndarray         # int32 [2n] 0-based leftmost reference position
    span: np.ndarray         # int32 [2n] reference bases covered
    flag: np.ndarray         # uint16 [2n]
    nm: np.ndarray           # int32 [2n] edit distance (-1 = NM tag absent)
    clip: np.ndarray         # int32 [2n, 2] soft-clip head/tail
    ev_off: np.ndarray       # int64 [2n+1]
    ev_pos: np.ndarray       # int32 [E] reference position of event
    ev_kind: np.ndarray      # uint8 [E]  EV_SINGLE / EV_INS / EV_DEL
    ev_val: np.ndarray       # int32 [E]  read base ASCII / index into ins_strings / length
    ins_strings: list[str] = field(default_factory=list)
    read_len: int = 150

    @property
    def n_pairs(self) -> int:
        return len(self.pair_gene)


def _haplotypeEvents(index: SynthIndex, gene: str, allele: str):
    """Sorted variant arrays of one allele."""
    vs = [v for v in index.variants if v.ref == gene and allele in v.allele]
    return vs


def makeSample(index: SynthIndex, seed: int = 1031, n_pairs: int = 2000,
               err_rate: float = 0.001, gene_cn: dict[str, int] | None = None,
               frac_multi: float = 0.03, frac_improper: float = 0.01,
               frac_clip: float = 0.01, read_len: int = 150,
               frag_mean: float = 400.0, frag_sd: float = 10.0,
               variants_by_gene: dict | None = None) -> SynthSample:
    """Draw ``n_pairs`` read pairs from a random genotype of ``index``."""
    rng = np.random.default_rng(seed)
    genes = index.genes
    if gene_cn is None:
        gene_cn = {}
        for g in genes:
            gene_cn[g] = 2 if "3DL3" in g else int(rng.choice([0, 1, 2, 2, 2, 3]))
    truth: dict[str, list[str]] = {}
    haps: list[tuple[int, str]] = []
    for gi, g in enumerate(genes):
        cn = gene_cn.get(g, 0)
        names = index.alleles[g]
        picks = []
        for c in range(cn):
            if c and rng.random() < 0.2:
                picks.append(picks[0])
            else:
                picks.append(names[int(rng.integers(0, len(names)))])
        truth[g] = picks
        haps.extend((gi, a) for a in picks)
    if not haps:
        raise ValueError("sample has no gene copies")
    if variants_by_gene is None:
        variants_by_gene = {}
        for v in index.variants:
            variants_by_gene.setdefault(v.ref, []).append(v)

    weights = np.array([len(index.backbone[genes[gi]]) for gi, _ in haps], dtype=np.float64)
    hap_of_pair = rng.choice(len(haps), size=n_pairs, p=weights / weights.sum())

    ins_strings: list[str] = []
    ins_lookup: dict[str, int] = {}

    pos0 = np.zeros(2 * n_pairs, dtype=np.int32)
    span = np.zeros(2 * n_pairs, dtype=np.int32)
    ev_m: list[np.ndarray] = []
    ev_p: list[np.ndarray] = []
    ev_k: list[np.ndarray] = []
    ev_v: list[np.ndarray] = []
    pair_gene = np.zeros(n_pairs, dtype=np.int32)

    for h, (gi, allele) in enumerate(haps):
        pidx = np.nonzero(hap_of_pair == h)[0]
        if not len(pidx):
            continue
        g = genes[gi]
        pair_gene[pidx] = gi
        bb = index.backbone[g]
        L = len(bb)
        hv = [v for v in variants_by_gene.get(g, []) if allele in v.allele]
        hpos = np.array([v.pos for v in hv], dtype=np.int64)
        hkind = np.array([{"single": EV_SINGLE, "insertion": EV_INS, "deletion": EV_DEL}[v.typ]
                          for v in hv], dtype=np.uint8)
        hval = np.zeros(len(hv), dtype=np.int32)
        for i, v in enumerate(hv):
            if v.typ == "single":
                hval[i] = ord(str(v.val))
            elif v.typ == "deletion":
                hval[i] = int(v.val)
            else:
                s = str(v.val)
                if s not in ins_lookup:
                    ins_lookup[s] = len(ins_strings)
                    ins_strings.append(s)
                hval[i] = ins_lookup[s]
        # allele coordinate -> reference position (-1 for inserted bases)
        keep = np.ones(L, dtype=bool)
        for v in hv:
            if v.typ == "deletion":
                keep[v.pos:v.pos + int(v.val)] = False
        ref_of = np.nonzero(keep)[0].astype(np.int64)
        ins_v = [v for v in hv if v.typ == "insertion"]
        if ins_v:
            at = np.searchsorted(ref_of, [v.pos for v in ins_v])
            at = np.repeat(at, [len(str(v.val)) for v in ins_v])
            ref_of = np.insert(ref_of, at, -1)
        La = len(ref_of)
        allele_base = bb.copy()
        for v in hv:
            if v.typ == "single":
                allele_base[v.pos] = ord(str(v.val))

        n = len(pidx)
        frag = np.clip(np.rint(rng.normal(frag_mean, frag_sd, n)).astype(np.int64),
                       2 * read_len // 2 + 10, La - 2)
        frag = np.maximum(frag, read_len)
        u = (rng.random(n) * (La - frag)).astype(np.int64)
        starts = np.stack([u, u + frag - read_len], axis=1)  # [n, 2] allele coords
        # both ends of each mate must be reference-aligned bases
        for _ in range(8):
            bad = (ref_of[starts] < 0) | (ref_of[starts + read_len - 1] < 0)
            if not bad.any():
                break
            starts = np.where(bad, np.minimum(starts + 1, La - read_len), starts)
        bad = (ref_of[starts] < 0) | (ref_of[starts + read_len - 1] < 0)
        starts = np.where(bad, 0, starts)  # allele coordinate 0 is always aligned enough
        m_ids = (2 * pidx[:, None] + np.arange(2)[None, :]).reshape(-1)
        st = starts.reshape(-1)
        p0 = ref_of[st]
        pe = ref_of[st + read_len - 1] + 1
        bad0 = (p0 < 0) | (pe <= 0)
        if bad0.any():  # pathological allele start inside an insertion: fall back to a clean window
            st = np.where(bad0, np.argmax(ref_of >= 0), st)
            p0 = ref_of[st]
            pe = ref_of[st + read_len - 1] + 1
        pos0[m_ids] = p0
        span[m_ids] = pe - p0

        # allele-carried variants inside each mate
        lo = np.searchsorted(hpos, p0, side="left")
        hi = np.searchsorted(hpos, pe, side="left")
        cnt = hi - lo
        tot = int(cnt.sum())
        if tot:
            rep = np.repeat(np.arange(len(m_ids)), cnt)
            offs = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            vi = np.repeat(lo, cnt) + offs
            k = hkind[vi]
            p = hpos[vi]
            ok = (k == EV_SINGLE) | (p > p0[rep])
            ev_m.append(m_ids[rep][ok]); ev_p.append(p[ok].astype(np.int32))
            ev_k.append(k[ok]); ev_v.append(hval[vi][ok])
        # substitution errors on aligned bases
        n_err = rng.binomial(read_len, err_rate, len(m_ids))
        te = int(n_err.sum())
        if te:
            rep = np.repeat(np.arange(len(m_ids)), n_err)
            off = rng.integers(0, read_len, te)
            rp = ref_of[st[rep] + off]
            ok = rp >= 0
            rep, rp = rep[ok], rp[ok]
            cur = allele_base[rp]
            shift = rng.integers(1, 4, len(rp))
            code = (np.searchsorted(BASES, cur) + shift) % 4
            eb = BASES[code]
            # priority 1 events: override allele base; dropped later if equal to backbone
            ev_m.append(m_ids[rep]); ev_p.append(rp.astype(np.int32))
            ev_k.append(np.full(len(rp), 8 + EV_SINGLE, dtype=np.uint8)); ev_v.append(eb.astype(np.int32))

    if ev_m:
        m = np.concatenate(ev_m); p = np.concatenate(ev_p)
        k = np.concatenate(ev_k); v = np.concatenate(ev_v)
    else:
        m = np.zeros(0, np.int64); p = np.zeros(0, np.int32)
        k = np.zeros(0, np.uint8); v = np.zeros(0, np.int32)
    # order: mate, position, (ins < single < del like the walk), errors after allele singles
    is_err = (k >= 8)
    kk = (k & 7).astype(np.int64)
    rank = np.where(kk == EV_INS, 0, np.where(kk == EV_SINGLE, 1, 2))
    order = np.lexsort((is_err, rank, p, m))
    m, p, k, v, kk, is_err = m[order], p[order], k[order], v[order], kk[order], is_err[order]
    # collapse same (mate, pos) singles: the last one (error) wins
    single = kk == EV_SINGLE
    nxt_same = np.zeros(len(m), dtype=bool)
    if len(m) > 1:
        nxt_same[:-1] = single[:-1] & single[1:] & (m[:-1] == m[1:]) & (p[:-1] == p[1:])
    keep = ~nxt_same
    # singles equal to the backbone are no mismatch at all
    gname_of_m = pair_gene[m // 2]
    refbase = np.zeros(len(m), dtype=np.int32)
    for gi, g in enumerate(genes):
        sel = gname_of_m == gi
        if sel.any():
            refbase[sel] = index.backbone[g][np.minimum(p[sel], len(index.backbone[g]) - 1)]
    keep &= ~(single & (v == refbase))
    m, p, kk, v, is_err = m[keep], p[keep], kk[keep], v[keep], is_err[keep]

    n_m = 2 * n_pairs
    ev_cnt = np.bincount(m, minlength=n_m)
    ev_off = np.zeros(n_m + 1, dtype=np.int64)
    np.cumsum(ev_cnt, out=ev_off[1:])
    # NM: HISAT2 counts only edits that are NOT variants of the graph (the reference's own example,
    # hisat2.py:294, is a read with two graph deletions that must pass the NM <= 4 filter), so
    # allele-carried variants cost nothing and substitution errors cost 1 unless they hit a graph SNP.
    known = np.array(sorted((gi << 40) | (vv.pos << 8) | ord(str(vv.val))
                            for gi, g in enumerate(genes) for vv in variants_by_gene.get(g, [])
                            if vv.typ == "single"), dtype=np.int64)
    ekey = (pair_gene[m // 2].astype(np.int64) << 40) | (p.astype(np.int64) << 8) | v.astype(np.int64)
    novel = is_err & ~np.isin(ekey, known)
    nm = np.bincount(m[novel], minlength=n_m).astype(np.int32)

    # flags / NH / secondary / clipping
    strand = rng.random(n_pairs) < 0.5
    flag = np.zeros(n_m, dtype=np.uint16)
    flag[0::2] = np.where(strand, 99, 83)
    flag[1::2] = np.where(strand, 147, 163)
    improper = rng.random(n_pairs) < frac_improper
    flag[0::2] = np.where(improper, flag[0::2] & ~np.uint16(2), flag[0::2])
    flag[1::2] = np.where(improper, flag[1::2] & ~np.uint16(2), flag[1::2])
    # mates that would not fit the packed device record are reported as not properly paired (both
    # the SAM and the packed route then drop the pair in filterRead)
    n_single = np.bincount(m[kk == EV_SINGLE], minlength=n_m)
    n_indel = np.bincount(m[kk != EV_SINGLE], minlength=n_m)
    n_ins = np.bincount(m[kk == EV_INS], minlength=n_m)
    too_big = (n_single > 16) | (n_ins > 6) | (2 * n_indel + 1 > 14) | (ev_cnt > 22)
    big_pair = too_big[0::2] | too_big[1::2]
    flag[0::2] = np.where(big_pair, flag[0::2] & ~np.uint16(2), flag[0::2])
    flag[1::2] = np.where(big_pair, flag[1::2] & ~np.uint16(2), flag[1::2])
    pair_nh = np.where(rng.random(n_pairs) < frac_multi, 2, 1).astype(np.uint8)
    clip = np.zeros((n_m, 2), dtype=np.int32)
    clipped = np.nonzero(rng.random(n_m) < frac_clip)[0]
    for mm in clipped:  # only event-free mates are clipped (keeps MD/CIGAR trivially valid)
        if ev_cnt[mm] == 0:
            c = int(rng.integers(1, 12))
            side = int(rng.integers(0, 2))
            clip[mm, side] = c
            if side == 0:
                pos0[mm] += c
            span[mm] -= c

    return SynthSample(index=index, gene_cn=gene_cn, truth=truth, pair_gene=pair_gene,
                       pair_nh=pair_nh, pair_secondary=np.zeros(n_pairs, dtype=bool),
                       pair_qname=np.arange(n_pairs, dtype=np.int64),
                       pos0=pos0, span=span, flag=flag, nm=nm, clip=clip, ev_off=ev_off,
                       ev_pos=p.astype(np.int32), ev_kind=kk.astype(np.uint8),
                       ev_val=v.astype(np.int32), ins_strings=ins_strings, read_len=read_len)


# --------------------------------------------------------------------------
# SAM text (small cases)
# --------------------------------------------------------------------------
def mateCigarMd(sample: SynthSample, m: int, known: dict | None = None, rng=None):
    """CIGAR, MD, SEQ and Zs of mate ``m`` rebuilt from its events."""
    idx = sample.index
    g = idx.genes[int(sample.pair_gene[m // 2])]
    bb = idx.backbone[g]
    p0 = int(sample.pos0[m]); sp = int(sample.span[m])
    head, tail = int(sample.clip[m, 0]), int(sample.clip[m, 1])
    b, e = int(sample.ev_off[m]), int(sample.ev_off[m + 1])
    cigar = []
    md = []
    seq = []
    zs = []
    run = 0          # current MD match run
    cur = p0         # reference cursor
    read_i = head    # read cursor (includes head clip)
    zs_end = 0       # read offset of end of previous Zs entry
    mlen = 0         # current M op length

    def flush_m():
        nonlocal mlen
        if mlen:
            cigar.append(f"{mlen}M")
            mlen = 0

    if head:
        cigar.append(f"{head}S")
        seq.append("N" * head)
    for i in range(b, e):
        ep = int(sample.ev_pos[i]); k = int(sample.ev_kind[i]); val = int(sample.ev_val[i])
        # matched stretch before the event
        if ep > cur:
            seq.append(bb[cur:ep].tobytes().decode())
            run += ep - cur; mlen += ep - cur; read_i += ep - cur
            cur = ep
        kid = None
        if known is not None:
            kid = known.get((g, ep, k, val if k != EV_INS else sample.ins_strings[val]))
        if k == EV_SINGLE:
            md.append(str(run)); md.append(chr(int(bb[ep]))); run = 0
            seq.append(chr(val)); mlen += 1
            if kid is not None and (rng is None or rng.random() < 0.7):
                zs.append(f"{read_i - zs_end}|S|{kid}"); zs_end = read_i + 1
            read_i += 1; cur += 1
        elif k == EV_INS:
            flush_m()
            s = sample.ins_strings[val]
            cigar.append(f"{len(s)}I"); seq.append(s)
            # no Zs entry for insertions: the reference's Zs bookkeeping only lines up for an
            # insertion when no MD match run is pending (hisat2.py:360-373), see DESIGN.md
            read_i += len(s)
        else:
            flush_m()
            cigar.append(f"{val}D")
            md.append(str(run)); md.append("^" + bb[ep:ep + val].tobytes().decode()); run = 0
            if kid is not None and (rng is None or rng.random() < 0.7):
                zs.append(f"{read_i - zs_end}|D|{kid}"); zs_end = read_i
            cur += val
    end = p0 + sp
    if end > cur:
        seq.append(bb[cur:end].tobytes().decode())
        run += end - cur; mlen += end - cur
    flush_m()
    md.append(str(run))
    if tail:
        cigar.append(f"{tail}S")
        seq.append("N" * tail)
    return "".join(cigar), "".join(md), "".join(seq), ",".join(zs)


def toSamLines(sample: SynthSample, zs_seed: int = 7, with_zs: bool = True) -> list[str]:
    """Name-collated SAM records (READ1 line then READ2 line of each pair)."""
    idx = sample.index
    known = None
    rng = None
    if with_zs:
        known = {(v.ref, v.pos, {"single": EV_SINGLE, "insertion": EV_INS, "deletion": EV_DEL}[v.typ],
                  (ord(str(v.val)) if v.typ == "single" else v.val)): v.id for v in idx.variants}
        rng = np.random.default_rng(zs_seed)
    lines = []
    for r in range(sample.n_pairs):
        g = idx.genes[int(sample.pair_gene[r])]
        recs = []
        for t in range(2):
            m = 2 * r + t
            cigar, md, seq, zs = mateCigarMd(sample, m, known, rng)
            recs.append((cigar, md, seq, zs))
        for t in range(2):
            m = 2 * r + t
            o = 2 * r + (1 - t)
            cigar, md, seq, zs = recs[t]
            tags = ["AS:i:0", "ZS:i:0", "XN:i:0"]
            if sample.nm[m] >= 0:
                tags.append(f"NM:i:{int(sample.nm[m])}")
            tags.append(f"MD:Z:{md}")
            tags += ["YS:i:0", "YT:Z:CP"]
            if zs:
                tags.append(f"Zs:Z:{zs}")
            tags.append(f"NH:i:{int(sample.pair_nh[r])}")
            fl = int(sample.flag[m]) | (256 if sample.pair_secondary[r] else 0)
            tlen = 0
            lines.append("\t".join([
                f"r{int(sample.pair_qname[r]):09d}", str(fl), g, str(int(sample.pos0[m]) + 1), "60",
                cigar, "=", str(int(sample.pos0[o]) + 1), str(tlen), seq, "I" * len(seq), *tags]))
    return lines


def withManyMismatches(lines: list[str], index: "SynthIndex", pairs: list[int], rng, n_mm=(17, 60)) -> list[str]:
    """SAM lines with the mates of ``pairs`` (pure-match mates only) rewritten to carry many substitutions against the
    backbone -- MD to match, no Zs, NM:i:0 so the mate still passes the filter: more mismatches than a ``gk_mate``
    holds, i.e. test input for the wide record format.  Returns a new list."""
    out = list(lines)
    for p in pairs:
        for m in (2 * p, 2 * p + 1):
            f = out[m].split("\t")
            if not f[5].endswith("M") or not f[5][:-1].isdigit():
                continue
            n = int(f[5][:-1])
            bb = index.backbone[f[2]]
            bb = bb if isinstance(bb, str) else bytes(bytearray(bb)).decode()
            pos0 = int(f[3]) - 1
            ref = bb[pos0:pos0 + n]
            if len(ref) != n or n != len(f[9]):
                continue
            k = int(rng.integers(n_mm[0], min(n_mm[1], n) + 1))
            at = sorted(rng.choice(n, size=k, replace=False).tolist())
            seq, md, last = list(ref), "", 0
            for q in at:
                seq[q] = "ACGT"[("ACGT".index(ref[q]) + 1 + int(rng.integers(3))) % 4] if ref[q] in "ACGT" else "A"
                md += f"{q - last}{ref[q]}"
                last = q + 1
            md += str(n - last)
            f[9] = "".join(seq)
            f = [c for c in f if not c.startswith("Zs:Z:")]
            f = ["MD:Z:" + md if c.startswith("MD:Z:") else "NM:i:0" if c.startswith("NM:i:") else c for c in f]
            out[m] = "\t".join(f)
    return out
