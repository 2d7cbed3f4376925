"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of ``graphkir/pileup.py``
(``getPileupBaseRatio`` 57-81) on SAM text.

The reference shells out to ``samtools mpileup -a`` and parses its base column; samtools is not in this
image, so the pileup itself is restated from the defaults documented in samtools-mpileup(1) -- skip
UNMAP / SECONDARY / QCFAIL / DUP records and paired reads outside a proper pair, drop bases of quality
< 13, overlapping mates: equal bases count once, unequal ones keep the better base at 0.8 of its quality,
deleted positions count as '*', no BAQ without a reference -- **parity unpinned for these counts**.  The
parser of the mpileup base column (``basesOfColumn`` <- parsePileupBase 13-37), the ratio dictionary
(``ratiosOfColumns`` <- getPileupBaseRatio 57-81) and everything downstream (``hisat2.errorCorrection``,
restated in ``oracle/tabulate.pileupCorrect``) follow the reference line by line and ARE pinned: fixture
``tests/golden/t11_pileup.json.gz`` holds the reference's own outputs on hand-written mpileup columns.
"""
from __future__ import annotations

import re
from collections import Counter, defaultdict

_CIGAR = re.compile(r"(\d+)([MIDNSHP=X])")
_NUM = re.compile(r"\d+")


def basesOfColumn(column: str) -> str:
    """Bases of one mpileup base column (parsePileupBase, pileup.py:13-37): ``$`` dropped, ``^x`` (read start +
    mapping quality) dropped, ``+n...`` / ``-n...`` indel annotations skipped, ``*`` and every other
    character kept."""
    out, i = [], 0
    while i < len(column):
        c = column[i]
        if c == "$":
            i += 1
        elif c in "+-":
            n = _NUM.search(column, i + 1)
            assert n
            i += 1 + len(n.group()) + int(n.group())
        elif c == "^":
            i += 2
        else:
            out.append(c)
            i += 1
    return "".join(out)


def ratiosOfColumns(rows) -> dict:
    """``rows`` = (ref, 0-based pos, depth, base column) like ``readPileup`` yields (pileup.py:40-54);
    depth-0 rows are skipped, bases are upper-cased, shares are count / total (getPileupBaseRatio 57-81)."""
    stat = {}
    for ref, pos, depth, column in rows:
        if depth == 0:
            continue
        bases = basesOfColumn(column)
        assert depth == len(bases)
        count = Counter(bases.upper())
        s = sum(count.values())
        stat[(ref, pos)] = {k: v / s for k, v in count.items()}
        stat[(ref, pos)]["all"] = s
    return stat


def _cover(line: str):
    """[(pos0, base or '*', quality)] of one alignment line."""
    f = line.split("\t")
    pos, seq, qual = int(f[3]) - 1, f[9], f[10]
    out, ri = [], 0
    for n, op in _CIGAR.findall(f[5]):
        n = int(n)
        if op in "M=X":
            for k in range(n):
                q = 255 if qual == "*" else ord(qual[ri]) - 33
                out.append((pos, seq[ri].upper(), q))
                pos += 1
                ri += 1
        elif op == "D":
            q = 255 if (qual == "*" or ri == 0) else ord(qual[ri - 1]) - 33
            for k in range(n):
                out.append((pos, "*", q))
                pos += 1
        elif op in "IS":
            ri += n
        elif op == "N":
            pos += n
    return f[2], out


def pileupOfLines(lines) -> dict:
    """``{(ref, pos): {base: share, ..., "all": depth}}`` like getPileupBaseRatio (pileup.py:70-81)."""
    recs = []
    for i, line in enumerate(lines):
        if not line or line.startswith("@"):
            continue
        flag = int(line.split("\t", 2)[1])
        if flag & (4 | 256 | 512 | 1024) or ((flag & 1) and not (flag & 2)):
            continue
        recs.append((i, line, flag))
    by_name = defaultdict(list)
    for k, (_, line, flag) in enumerate(recs):
        if flag & 1:
            by_name[line.split("\t", 1)[0]].append(k)
    mate = {}
    for name, ks in by_name.items():
        for a, b in zip(ks[0::2], ks[1::2]):
            mate[a], mate[b] = b, a
    covers = [_cover(line) for _, line, _ in recs]
    count = defaultdict(Counter)
    for k, (i, line, flag) in enumerate(recs):
        ref, mine = covers[k]
        other = {}
        first = True
        if k in mate and covers[mate[k]][0] == ref:
            other = {p: (b, q) for p, b, q in covers[mate[k]][1]}
            my_start, its_start = int(line.split("\t")[3]), int(recs[mate[k]][1].split("\t")[3])
            first = my_start < its_start or (my_start == its_start and i < recs[mate[k]][0])
        for p, b, q in mine:
            if p in other and b != "*" and other[p][0] != "*":
                ob, oq = other[p]
                if ob == b:
                    q = min(200, q + oq) if first else 0
                else:
                    wins = q >= oq if first else q > oq
                    q = int(0.8 * q) if wins else 0
            if q >= 13:
                count[(ref, p)][b if b in "ACGT*" else "N"] += 1
    stat = {}
    for key, c in count.items():
        s = sum(c.values())
        stat[key] = {b: v / s for b, v in c.items()}
        stat[key]["all"] = s
    return stat
