"""CPU checks of the seeded generator: the three copy families, the ground-truth files of
DataSimulator.py:241-262 and the cut reads, and that the seeded workloads the fixtures rest on do not drift."""
import hashlib
import os

import numpy as np

from repeatresolver_amd import datagen as dg


def _read_fasta(path):
    seqs, cur = [], None
    for line in open(path, "rb").read().split(b"\n"):
        if line.startswith(b">"):
            if cur is not None:
                seqs.append(b"".join(cur))
            cur = []
        elif line:
            assert len(line) <= 100 or path.endswith("_Template.fasta")
            cur.append(line)
    if cur is not None:
        seqs.append(b"".join(cur))
    return seqs


def test_dataset_files_and_ground_truth(tmp_path):
    for kind in ("Tree", "Distributed", "EquiDistant"):
        cfg = dg.SimConfig(kind=kind, copies=5, coverage=4, difference=0.02, repeat_len=900, flank=300, length_scale=0.06, seed=7)
        prefix = str(tmp_path / cfg.name)
        n = dg.write_dataset(prefix, cfg)
        reads = _read_fasta(prefix + ".fasta")
        places = [int(v) for v in open(prefix + "_ReadPlacements").read().split()]
        copies = [int(v) for v in open(prefix + "_ReadCopynumbers").read().split()]
        tmpl = _read_fasta(prefix + "_Template.fasta")
        cut = _read_fasta(prefix + "Seq.fasta")
        assert len(reads) == len(places) == len(copies) == n["reads"] > 10
        assert len(tmpl) == 1 and len(tmpl[0]) == 900 and set(tmpl[0]) <= set(b"acgt")
        assert max(copies) == 4 and min(copies) == 0 and copies == sorted(copies)          # copy after copy, DataSimulator.py:228-232
        assert all(0 <= s <= 900 * 2 + 600 for s in places)
        assert len(cut) == n["cut_reads"] <= len(reads) and all(set(c) <= set(b"acgt") for c in reads + cut)
        # every cut read is a contiguous piece of its full read
        seq, full, starts, cids, cut_b, cut_t = dg.simulate_dataset(cfg)
        j = 0
        for rb, cb in zip(full, cut_b):
            if cb is None:
                continue
            assert bytes(dg.ASCII[cb]) in bytes(dg.ASCII[rb]) and bytes(dg.ASCII[cb]) == cut[j]
            j += 1
        # a read's template positions increase where defined
        for ct in cut_t:
            if ct is not None:
                t = ct[ct >= 0]
                assert (np.diff(t) > 0).all()


def test_equidistant_pairwise_distance_scale():
    rng = np.random.default_rng(3)
    seq = rng.integers(0, 4, size=4000, dtype=np.uint8)
    copies = dg.equidistant_copies(rng, seq, 6, 0.02)
    # each copy carries d/2 * len edits of its own (DataSimulator.py:73-74)
    for b, t in copies:
        changed = int((t == -1).sum()) + (4000 - int((t >= 0).sum())) + int((b[t >= 0] != seq[t[t >= 0]]).sum())
        assert 20 <= changed <= 45


def test_seeded_workloads_do_not_drift():
    """Fixtures and the bench rest on these byte streams: the committed fixture inputs ARE generator outputs."""
    from conftest import golden_input
    for name, fx in (("toy_a", "toy_a_b1000"), ("toy_b", "toy_b_b1000")):
        m = dg.make_msa(name)
        out = np.empty((m.shape[0], m.shape[1] + 1), dtype=np.uint8)
        out[:, :-1] = m
        out[:, -1] = 10
        assert out.tobytes() == golden_input(fx), name
    assert dg.make_msa("tree_medium").shape == (2320, 37301)
