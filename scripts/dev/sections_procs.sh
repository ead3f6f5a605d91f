# dev (GPU box): the six sections as six PROCESSES on the one GPU (is the threads' aggregate held back by the host's launch path?)
for p in 0 1 2 3 4 5; do
  python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --sections 6 --only-section $p "$@" > gpurun_out/secp_$p.json 2> gpurun_out/secp_$p.err &
done
wait
python3 - <<'PY'
import json
tot = 0; tmax = 0
for p in range(6):
    d = json.load(open(f"gpurun_out/secp_{p}.json"))
    cells = d["value"] * d["ms_per_step"] * d["steps"] / 1e3
    tot += cells; tmax = max(tmax, d["ms_per_step"] * d["steps"] / 1e3)
    print(p, "ms/step %.1f value %.3e" % (d["ms_per_step"], d["value"]))
print("six processes: cells %.3e in %.2f s (slowest) = %.3e cells/s" % (tot, tmax, tot / tmax))
PY
