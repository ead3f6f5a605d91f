"""Dev aid (CPU): what is gained by speculating on rows that are DISJOINT from every uncommitted row before them instead of on
the next rows in order?  The row intervals of a workload's MSA (first / last base -+ half a bandwidth), a window of jobs per
batch, the first row always commits, a row picked ahead commits when it is disjoint from all uncommitted rows before it, a row in
order that overlaps commits with probability p; a batch costs as long as its longest fill (the plan of k_fill_v3: own rows + 300
of warm-up at 0.3 us) + 100 us of tail.  usage: plan_sim.py [workload]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from repeatresolver_amd import datagen as dg
m = dg.make_msa(sys.argv[1] if len(sys.argv) > 1 else "tree_default")
T, W = m.shape
isb = (m != ord('-')) & (m != ord(' '))
L = isb.sum(axis=1)
first = isb.argmax(axis=1); last = W - 1 - isb[:, ::-1].argmax(axis=1)
H, B = 500, 1000
lo = np.maximum(0, first - H - 1); hi = np.minimum(W - 1, last - H + B - 1)


def chain(l):
    S = max(1, min((l + 80) // 160, 64, l // 128))
    return l / S + (300 if S > 1 else 0)


def sim(window, smart, p_overlap, spec_len=6, free_len=1000, slack=2):
    rng = np.random.default_rng(0)
    done = np.zeros(T, bool); k = 0; batches = commits = 0; t = 0.0
    while k < T:
        if done[k]:
            k += 1; continue
        batches += 1
        chosen = [k]
        lim = L[k] + L[k] * spec_len // 100 + 64
        if smart:
            ulo, uhi = [lo[k]], [hi[k]]
            for j in range(k + 1, min(T, k + 64)):
                if done[j]:
                    continue
                if len(chosen) < window and L[j] <= L[k] * (100 + free_len) // 100 + 64 and all(hi[j] + slack < a or b + slack < lo[j] for a, b in zip(ulo, uhi)):
                    chosen.append(j)
                ulo.append(lo[j]); uhi.append(hi[j])
        j = k + 1
        while len(chosen) < window and j < min(T, k + 64):
            if not done[j] and j not in chosen:
                if L[j] > lim:
                    break
                chosen.append(j)
            j += 1
        chosen.sort()
        committed = {k}
        for j in chosen[1:]:
            earlier = [i for i in range(k, j) if not done[i]]
            if all(hi[j] + 2 < lo[i] or hi[i] + 2 < lo[j] for i in earlier):
                committed.add(j)
            elif all(i in committed for i in earlier) and rng.random() < p_overlap:
                committed.add(j)
        for j in committed:
            done[j] = True
        commits += len(committed)
        t += (102 + 0.30 * max(chain(L[j]) for j in chosen)) * 1.09        # (9 % of the fills are repeated after a failed check)
    return commits / batches, t / 1e6


for p in (0.05, 0.1):
    for w in (3, 4):
        a = sim(w, False, p); b = sim(w, True, p, free_len=30); c = sim(w, True, p); d = sim(w, True, p, slack=2050)
        print("p_overlap %.2f window %d: in order %.3f commits/batch %.2f s/round | picked ahead, rows <= 1.3 x the first %.3f %.2f s | any length %.3f %.2f s | any length, gap 2050 %.3f %.2f s" % ((p, w) + a + b + c + d))
