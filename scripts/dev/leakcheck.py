"""dev (GPU box): does creating and destroying contexts leak process resources (file descriptors, mappings, threads)?
A second HIP runtime in the process (torch brings its own) failed to initialise after ~250 contexts had come and gone."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import golden_input, split_rows
from repeatresolver_amd.realigner import PWReAligner
rows = split_rows(golden_input("toy_a_b1000"))
def snap(tag):
    fds = os.listdir("/proc/self/fd")
    kinds = {}
    for f in fds:
        try: t = os.readlink("/proc/self/fd/" + f)
        except OSError: t = "?"
        k = t.split(":")[0] if not t.startswith("/") else t
        kinds[k] = kinds.get(k, 0) + 1
    maps = sum(1 for _ in open("/proc/self/maps"))
    thr = len(os.listdir("/proc/self/task"))
    top = sorted(kinds.items(), key=lambda kv: -kv[1])[:6]
    print(tag, "fds", len(fds), "maps", maps, "threads", thr, top, flush=True)
snap("start")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(n):
    g = PWReAligner(rows, bandwidth=1000, window=(1, 3, 16)[i % 3])
    g.trim_ends(); g.realign_rows(0, 4); g.total_score(); g.export_rows()
    g.close()
    if i % 50 == 49: snap("after %d contexts" % (i + 1))
import resource
print("RLIMIT_NOFILE", resource.getrlimit(resource.RLIMIT_NOFILE))
import torch
try:
    torch.zeros(4, device="cuda"); print("torch.cuda init ok")
except Exception as e:
    print("torch.cuda init FAILED:", e)
snap("end")
