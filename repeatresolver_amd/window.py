"""Section split of an MSA: integer-exact restatement of the reference's Window.py (Python 2) and
the column slicer / merger the multi-GPU path uses.

Window.py:41-60: coverage (rows that are not ' ') sampled at every 100th column, the integer mean
of those samples, trim both ends while the sample is below `coverage * mean`, then `parts`
equally spaced boundaries with Python-2 integer division."""
from __future__ import annotations


def window_boundaries(rows, coverage: float = 0.90, parts: int = 6):
    """rows: list of equal-length bytes.  Returns parts+1 column boundaries (Window.py:56-60)."""
    width = len(rows[0])
    cols = range(0, width, 100)                                              # Window.py:41
    cov = [sum(1 for r in rows if r[c] != 0x20) for c in cols]
    average = sum(cov) // len(cov)                                           # Window.py:43 (py2 int division)
    start = 0
    while cov[start] < coverage * average:                                   # Window.py:46-48
        start += 1
    start *= 100
    ende = len(cov) - 1
    while cov[ende] < coverage * average:                                    # Window.py:50-53
        ende -= 1
    ende *= 100
    return [start] + [start + (p + 1) * (ende - start) // parts for p in range(parts)]   # Window.py:56-59


def slice_sections(rows, boundaries):
    """All rows, columns [b_p, b_{p+1}) for every section p.  The reference never writes sliced MSAs
    (RepeatResolver.c:330-334 slices in memory); this layout is ours: every section keeps all T
    rows so that row indices stay aligned across sections."""
    return [[r[boundaries[p]:boundaries[p + 1]] for r in rows] for p in range(len(boundaries) - 1)]


def merge_sections(sections):
    """Concatenate realigned sections column-wise (each section: list of T equal-length rows)."""
    T = len(sections[0])
    return [b"".join(sec[r] for sec in sections) for r in range(T)]
