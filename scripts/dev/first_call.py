import sys, time
sys.path.insert(0, ".")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
rows = [bytes(r) for r in dg.make_msa("tree_default")]
g = PWReAligner(rows, bandwidth=1000)
t0 = time.time(); g.trim_ends(); t1 = time.time(); s = g.total_score(); t2 = time.time()
print("trim %.2f s, first total_score (upload) %.2f s" % (t1 - t0, t2 - t1))
for k in range(4):
    t0 = time.time(); g.realign_row(k); print("realign_row %d: %.3f s" % (k, time.time() - t0))


g.close()
