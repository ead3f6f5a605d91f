"""dev (GPU box): one test function of the suite over and over in ONE process, every failure with its message.
usage: repeat_test.py <tests/file.py> <function> <times> [parametrize args as python literals ...]"""
import ast, importlib.util, os, sys, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import Oracle
path, fn, times = sys.argv[1], sys.argv[2], int(sys.argv[3])
args = [ast.literal_eval(a) for a in sys.argv[4:]]
spec = importlib.util.spec_from_file_location("t", path); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
f = getattr(mod, fn)
f = getattr(f, "__wrapped__", f)
oracle = Oracle()
bad = 0
for i in range(times):
    try:
        f(*args, oracle)
        print("run", i, "ok", flush=True)
    except Exception:
        bad += 1
        print("run", i, "FAILED", flush=True)
        traceback.print_exc()
print(times, "runs,", bad, "failures")
