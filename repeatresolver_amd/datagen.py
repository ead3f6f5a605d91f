"""Seeded synthetic-input generator for the PW_ReAligner hot path.

Restates the *parameters and distributions* of the reference's (unseeded, Python-2)
``DataSimulator.py`` so that BASELINE.json's configurations can be produced reproducibly
without Python 2:

* random ``acgt`` repeat template                      (DataSimulator.py:119-123)
* copy families ``Tree`` / ``Distributed``             (DataSimulator.py:93-115, :29-49)
* 10 kb random flanks on both sides of every copy      (DataSimulator.py:222-225)
* read lengths from the Drosophila histogram, uniform
  starts, until the per-copy repeat coverage is reached (DataSimulator.py:126-160)
* PacBio error model: keep 95.2 %, substitute 1.4 %,
  delete 3.4 %, geometric insertions p = 0.103139      (DataSimulator.py:12-27)

The reference then runs ``ReadCutter`` + ``InitialAligner`` to obtain the MSA that
``PW_ReAligner`` reads.  Those tools are out of scope; this module keeps the
*true* read-to-template alignment while simulating and stacks it into an MSA with the same
layout rule as ``InitialAligner.c:553-663`` (per template slot: the longest insertion run of
any read, left-aligned insertions padded with '-', then the template column).  The text format
is exactly what ``PW_ReAligner`` consumes: equal-width rows over ``acgt-``, one ``\\n`` per row.

It can also emit the reads / template as FASTA so that, inside the build container, the
reference's own ``InitialAligner`` can be used to make fixture inputs (see oracle/gen_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

ASCII = np.frombuffer(b"acgt", dtype=np.uint8)
GAP = ord("-")

# DataSimulator.py:126-127 -- read-length histogram, 1 kb bins
LENGTHS_HISTO = np.array(
    [0, 323, 427, 411, 355, 353, 358, 321, 293, 321, 281, 275, 241, 239, 226, 185, 177, 162, 126,
     117, 126, 108, 88, 83, 61, 52, 51, 29, 16, 7, 3, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.float64)

P_KEEP = 0.837 + 0.115           # DataSimulator.py:16
P_SUB = P_KEEP + 0.014           # DataSimulator.py:18
P_INS = 0.103139                 # DataSimulator.py:24
FLANK = 10000                    # DataSimulator.py:223-224


@dataclass
class SimConfig:
    """Mirror of DataSimulator.py's flags (-c -n -d -l -t) plus the knobs we need for small cases."""
    kind: str = "Tree"           # -t  Tree | Distributed
    copies: int = 100            # -n
    coverage: float = 40.0       # -c
    difference: float = 0.01     # -d given in percent on the reference's command line (1 -> 0.01)
    repeat_len: int = 30000      # -l
    flank: int = FLANK
    length_scale: float = 1.0    # shrink read lengths for toy cases (1.0 = reference histogram)
    min_aligned: int = 1000      # rows with fewer template-aligned bases are dropped
    seed: int = 1

    @property
    def name(self) -> str:
        perc = ("%g" % (self.difference * 100.0)).replace(".", "")
        return f"{self.kind}_{perc}perc_{self.repeat_len}kb"   # DataSimulator.py:199 naming


@dataclass
class SimData:
    template: np.ndarray                       # uint8 codes 0..3
    reads: list = field(default_factory=list)  # list of uint8 code arrays (repeat part only)
    tpos: list = field(default_factory=list)   # list of int32 arrays: template position or -1
    copy_of: list = field(default_factory=list)
    start_of: list = field(default_factory=list)


def _other_base(rng, base):
    # NotBase[...] pick (DataSimulator.py:11, :19): a uniformly chosen different base
    return (base + 1 + rng.integers(0, 3, size=np.shape(base), dtype=np.int64)) % 4


def _apply_edit(rng, bases, tpos, position, errortype, rand3):
    """One Sub/Del/Ins edit, each with probability 1/3 (DataSimulator.py:104-111)."""
    if errortype <= 1.0 / 3.0:
        bases = bases.copy()
        # NotBase[b][rand] with rand in 0..2
        bases[position] = (bases[position] + 1 + rand3) % 4
        return bases, tpos
    if errortype <= 2.0 / 3.0:
        return np.delete(bases, position), np.delete(tpos, position)
    newbase = rng.integers(0, 4)
    return np.insert(bases, position, newbase), np.insert(tpos, position, -1)


def tree_copies(rng, seq, copynumber, difference):
    """DataSimulator.py:93-115 -- binary tree of depth floor(log2 n)+1, d/2*len edits per edge."""
    snp = int(difference / 2.0 * len(seq))
    level = [(seq.copy(), np.arange(len(seq), dtype=np.int32))]
    for _ in range(int(math.log(copynumber, 2)) + 1):
        nxt = []
        for ob, ot in level:
            for _c in range(2):
                b, t = ob, ot
                for _tt in range(snp):
                    position = int(rng.random() * (len(ob) - snp))
                    b, t = _apply_edit(rng, b, t, position, rng.random(), int(rng.random() * 3))
                nxt.append((b, t))
        level = nxt
    return level[:copynumber]


def distributed_copies(rng, seq, copynumber, difference):
    """DataSimulator.py:29-49 -- len*d*3 variant sites, each applied to a random-size random subset."""
    snp = int(len(seq) * difference * 3)
    copies = [(seq.copy(), np.arange(len(seq), dtype=np.int32)) for _ in range(copynumber)]
    positions = np.sort(10 + (rng.random(snp) * (len(seq) - 20)).astype(np.int64))
    for t in range(snp):
        position = int(positions[-t - 1])
        order = rng.permutation(copynumber)
        rand = int(rng.random() * copynumber)
        errortype = rng.random()
        newbase = rng.integers(0, 4)
        for x in order[:rand]:
            b, tp = copies[x]
            if position >= len(b):
                continue
            if errortype <= 1.0 / 3.0:
                b = b.copy()
                b[position] = (b[position] + 1 + rand % 3) % 4
            elif errortype <= 2.0 / 3.0:
                b, tp = np.delete(b, position), np.delete(tp, position)
            else:
                b, tp = np.insert(b, position, newbase), np.insert(tp, position, -1)
            copies[x] = (b, tp)
    return copies


def equidistant_copies(rng, seq, copynumber, difference):
    """DataSimulator.py:72-90 -- every copy gets d/2*len edits of its own (any two copies differ by d)."""
    snp = int(difference / 2.0 * len(seq))
    out = []
    for _c in range(copynumber):
        b, t = seq.copy(), np.arange(len(seq), dtype=np.int32)
        for _t in range(snp):
            position = min(int(rng.random() * len(seq)), len(b) - 1)      # (the reference indexes with the ORIGINAL length)
            b, t = _apply_edit(rng, b, t, position, rng.random(), int(rng.random() * 3))
        out.append((b, t))
    return out


def pacbio_error(rng, bases, tpos):
    """Vectorised DataSimulator.py:12-27: per base keep / substitute / delete, then a geometric
    number of random insertions after it."""
    n = len(bases)
    r = rng.random(n)
    keep = r < P_SUB                       # kept or substituted
    sub = keep & (r >= P_KEEP)
    out_b = bases.copy()
    if sub.any():
        out_b[sub] = _other_base(rng, bases[sub].astype(np.int64)).astype(np.uint8)
    nins = rng.geometric(1.0 - P_INS, size=n) - 1          # P(k >= 1) = 0.103139
    per = keep.astype(np.int64) + nins
    total = int(per.sum())
    res_b = np.empty(total, dtype=np.uint8)
    res_t = np.full(total, -1, dtype=np.int32)
    ends = np.cumsum(per)
    starts = ends - per
    idx = starts[keep]
    res_b[idx] = out_b[keep]
    res_t[idx] = tpos[keep]
    ins_mask = np.ones(total, dtype=bool)
    ins_mask[idx] = False
    res_b[ins_mask] = rng.integers(0, 4, size=int(ins_mask.sum()), dtype=np.uint8)
    return res_b, res_t


def simulate(cfg: SimConfig) -> SimData:
    rng = np.random.default_rng(cfg.seed)
    seq = rng.integers(0, 4, size=cfg.repeat_len, dtype=np.uint8)
    if cfg.kind == "Tree":
        copies = tree_copies(rng, seq, cfg.copies, cfg.difference)
    elif cfg.kind == "Distributed":
        copies = distributed_copies(rng, seq, cfg.copies, cfg.difference)
    elif cfg.kind == "EquiDistant":
        copies = equidistant_copies(rng, seq, cfg.copies, cfg.difference)
    else:
        raise ValueError("kind must be Tree, Distributed or EquiDistant (DataSimulator.py:186)")
    prob = LENGTHS_HISTO / LENGTHS_HISTO.sum()
    data = SimData(template=seq)
    fl = cfg.flank
    for c, (cb, ct) in enumerate(copies):
        glen = len(cb) + 2 * fl
        current = 0.0
        covsum = 0
        while current < cfg.coverage:                       # DataSimulator.py:136-152
            length = int(rng.choice(len(prob), p=prob)) * 1000 + int(rng.random() * 1000)
            length = max(1, int(length * cfg.length_scale))
            length = min(length, glen - 1)
            start = int(rng.random() * (glen - length))
            lo, hi = max(start, fl), min(glen - fl, start + length)
            covsum += hi - lo
            current = covsum / float(glen - 2 * fl)
            if hi - lo <= 0:
                continue                                     # read entirely in a flank: no repeat part
            rb, rt = pacbio_error(rng, cb[lo - fl:hi - fl], ct[lo - fl:hi - fl])
            if int((rt >= 0).sum()) < cfg.min_aligned:
                continue
            data.reads.append(rb)
            data.tpos.append(rt)
            data.copy_of.append(c)
            data.start_of.append(start)
    return data


FLANK_MARK = -2          # tpos of a base that comes from a flank (simulate_dataset)


def simulate_dataset(cfg: SimConfig):
    """The whole of DataSimulator.py:204-262 including what simulate() leaves out: the random flanks, the FULL reads
    (flank bases included, as `<ds>.fasta` holds them) and the ground truth (`_ReadPlacements`, `_ReadCopynumbers`).
    Returns (template, full_reads, starts, copy_ids, cut_reads, cut_tpos): cut_reads[i] is the part of read i that the
    pipeline's ReadCutter keeps -- the stretch sampled from the repeat copy -- or None when the read lies in a flank.
    (Its own random stream: simulate()'s seeded workloads do not change.)"""
    rng = np.random.default_rng([cfg.seed, 0x5EED])
    seq = rng.integers(0, 4, size=cfg.repeat_len, dtype=np.uint8)
    if cfg.kind == "Tree":
        copies = tree_copies(rng, seq, cfg.copies, cfg.difference)
    elif cfg.kind == "Distributed":
        copies = distributed_copies(rng, seq, cfg.copies, cfg.difference)
    elif cfg.kind == "EquiDistant":
        copies = equidistant_copies(rng, seq, cfg.copies, cfg.difference)
    else:
        raise ValueError("kind must be Tree, Distributed or EquiDistant (DataSimulator.py:186)")
    prob = LENGTHS_HISTO / LENGTHS_HISTO.sum()
    fl = cfg.flank
    full, starts, cids, cut_b, cut_t = [], [], [], [], []
    for c, (cb, ct) in enumerate(copies):
        left = rng.integers(0, 4, size=fl, dtype=np.uint8)                # DataSimulator.py:222-225
        right = rng.integers(0, 4, size=fl, dtype=np.uint8)
        gb = np.concatenate((left, cb, right))
        gt = np.concatenate((np.full(fl, FLANK_MARK, np.int32), ct, np.full(fl, FLANK_MARK, np.int32)))
        glen = len(gb)
        covsum, current = 0, 0.0
        while current < cfg.coverage:                                      # DataSimulator.py:136-152
            length = int(rng.choice(len(prob), p=prob)) * 1000 + int(rng.random() * 1000)
            length = min(max(1, int(length * cfg.length_scale)), glen - 1)
            start = int(rng.random() * (glen - length))
            covsum += min(glen - fl, start + length) - max(start, fl)
            current = covsum / float(glen - 2 * fl)
            rb, rt = pacbio_error(rng, gb[start:start + length], gt[start:start + length])
            # inserted bases (tpos -1) inside the repeat stretch belong to it: mark origin by the nearest kept base on the left
            origin = rt.copy()
            kept = rt != -1
            if kept.any():
                idx = np.maximum.accumulate(np.where(kept, np.arange(len(rt)), -1))
                origin = np.where(idx >= 0, rt[np.maximum(idx, 0)], rt[kept][0])
            rep = np.nonzero(origin != FLANK_MARK)[0]
            full.append(rb)
            starts.append(start)
            cids.append(c)
            if len(rep):
                a, b = int(rep[0]), int(rep[-1]) + 1
                cut_b.append(rb[a:b])
                cut_t.append(np.where(rt[a:b] == FLANK_MARK, -1, rt[a:b]).astype(np.int32))
            else:
                cut_b.append(None)
                cut_t.append(None)
    return seq, full, starts, cids, cut_b, cut_t


def write_dataset(prefix: str, cfg: SimConfig) -> dict:
    """Writes what DataSimulator.py:241-262 writes -- `<prefix>.fasta` (full reads), `<prefix>_ReadPlacements`,
    `<prefix>_ReadCopynumbers`, `<prefix>_Template.fasta` -- plus `<prefix>Seq.fasta`, the reads cut to their repeat
    part: the file ReadCutter hands to InitialAligner (ReadCutter.c:953-970; simulated reads hold at most one repeat
    copy, so cutting at the true boundaries is what it amounts to).  Returns the counts."""
    seq, full, starts, cids, cut_b, _ = simulate_dataset(cfg)
    write_fasta(prefix + ".fasta", full)
    with open(prefix + "_ReadPlacements", "w") as f:
        f.writelines("%d\n" % s_ for s_ in starts)
    with open(prefix + "_ReadCopynumbers", "w") as f:
        f.writelines("%d\n" % c_ for c_ in cids)
    with open(prefix + "_Template.fasta", "wb") as f:
        f.write(b">\n" + ASCII[seq].tobytes() + b"\n")                   # one line, DataSimulator.py:259-262
    cut = [c_ for c_ in cut_b if c_ is not None]
    write_fasta(prefix + "Seq.fasta", cut)
    return {"reads": len(full), "cut_reads": len(cut), "template": len(seq)}


def build_msa(data: SimData) -> np.ndarray:
    """Stack the true alignments into an MSA exactly like InitialAligner.c:553-663 does with its
    computed ones.  Returns a (rows, width) uint8 array of ASCII bytes over ``acgt-``."""
    tl = len(data.template)
    gapcount = np.zeros(tl + 1, dtype=np.int64)
    slots, within = [], []
    for rt in data.tpos:
        n = len(rt)
        aligned = rt >= 0
        idx = np.arange(n)
        # slot of an inserted base = (template position of the last aligned base before it) + 1,
        # leading insertions go to the slot of the first aligned base (InitialAligner.c:579-596)
        last_al = np.maximum.accumulate(np.where(aligned, idx, -1))
        first_tp = rt[aligned][0]
        slot = np.where(last_al >= 0, rt[np.maximum(last_al, 0)] + 1, first_tp).astype(np.int64)
        slot[aligned] = rt[aligned]
        run_start = np.where(last_al >= 0, last_al + 1, 0)
        w = (idx - run_start).astype(np.int64)
        ins = ~aligned
        if ins.any():
            np.maximum.at(gapcount, slot[ins], w[ins] + 1)
        slots.append(slot)
        within.append(w)
    colstart = np.concatenate(([0], np.cumsum(gapcount + 1)))
    width = int(colstart[-1])
    msa = np.full((len(data.reads), width), GAP, dtype=np.uint8)
    for j, (rb, rt) in enumerate(zip(data.reads, data.tpos)):
        aligned = rt >= 0
        slot, w = slots[j], within[j]
        col = np.where(aligned, colstart[slot] + gapcount[slot], colstart[slot] + w)
        msa[j, col] = ASCII[rb]
    return msa


def write_msa(path, msa: np.ndarray) -> None:
    out = np.empty((msa.shape[0], msa.shape[1] + 1), dtype=np.uint8)
    out[:, :-1] = msa
    out[:, -1] = ord("\n")
    out.tofile(path)


def write_fasta(path, seqs) -> None:
    """FASTA as DataSimulator.py:243-262 writes it ('>' header line, 100 bases per line)."""
    with open(path, "wb") as f:
        for s in seqs:
            f.write(b">\n")
            a = ASCII[s].tobytes()
            for t in range(0, len(a), 100):
                f.write(a[t:t + 100] + b"\n")


# Named configurations of BASELINE.json ("configs"), plus toy shapes used by the test-suite.
CONFIGS = {
    # configs[0]/[1]/[3]: DataSimulator default  -t Tree -n 100 -c 40 -d 1 -l 30000
    "tree_default": SimConfig(kind="Tree", copies=100, coverage=40, difference=0.01, repeat_len=30000),
    # configs[2]: -t Distributed -n 200 -c 60 -l 30000
    "distributed_stress": SimConfig(kind="Distributed", copies=200, coverage=60, difference=0.01,
                                    repeat_len=30000),
    # toy shapes (same generator, shrunk): fast CI inputs
    "toy_a": SimConfig(kind="Tree", copies=4, coverage=8, difference=0.01, repeat_len=1500, flank=500,
                       length_scale=0.08, min_aligned=100, seed=11),
    "toy_b": SimConfig(kind="Tree", copies=10, coverage=12, difference=0.01, repeat_len=4000, flank=1500,
                       length_scale=0.25, min_aligned=200, seed=12),
    "toy_c": SimConfig(kind="Tree", copies=40, coverage=25, difference=0.01, repeat_len=1000, flank=300,
                       length_scale=0.05, min_aligned=100, seed=13),
    # mid-size shape for profiling runs (about 1/10 of the default in every dimension)
    "tree_medium": SimConfig(kind="Tree", copies=24, coverage=30, difference=0.01, repeat_len=10000, flank=3500,
                             length_scale=0.35, min_aligned=350, seed=15),
    # transposon-sized stand-in for configs[4] (real Drosophila files are not available offline)
    "transposon_like": SimConfig(kind="Distributed", copies=30, coverage=20, difference=0.02,
                                 repeat_len=5000, flank=2000, length_scale=0.3, min_aligned=300, seed=14),
}


def make_msa(name_or_cfg, seed=None) -> np.ndarray:
    cfg = CONFIGS[name_or_cfg] if isinstance(name_or_cfg, str) else name_or_cfg
    if seed is not None:
        cfg = SimConfig(**{**cfg.__dict__, "seed": seed})
    return build_msa(simulate(cfg))
