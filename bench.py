#!/usr/bin/env python3
"""bench.py -- sum-of-pairs DP cells/s of the MI355X PW_ReAligner hot path.

A "step" is one full realignment round (every row of the MSA once, PW_ReAligner.c:1695-1737) over
a synthetic MSA that is resident in HBM when the timed region starts.  At N=1 the workload is
BASELINE.json configs[1]: the DataSimulator-default Tree_1perc_30000kb MSA (100 copies, 40x, 30 kb).
With N>1 every rank realigns its own, independent MSA of the same shape (weak scaling; the path
shards by MSA / section with no data-path collective).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL = 4.0           # algorithmic bytes per DP cell (one 32-bit score per cell)


def measured_traffic_per_cell():
    """HBM bytes per computed DP cell of the fill kernel, from the committed rocprofv3 PMC passes
    (profiles/r01_traffic_model.json: 2 x FETCH_SIZE + WRITE_SIZE over the cells of the same run)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic_model.json")) as f:
            return float(json.load(f)["hbm_bytes_per_cell"])
    except Exception:
        return None


def cpu_baseline(rows, bandwidth, budget_s=15.0):
    """oracle/ is the checker; here it is only timed (kind "port", 1 core) on a bounded sample:
    the first rows of round 1 of the same MSA, until budget_s of CPU time is used."""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "libpworacle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", odir, "port"], check=True, stdout=subprocess.DEVNULL)
    o = ctypes.CDLL(so)
    o.pwo_create.restype = ctypes.c_void_p
    o.pwo_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    o.pwo_cells.restype = ctypes.c_uint64
    o.pwo_cells.argtypes = [ctypes.c_void_p]
    o.pwo_trim.argtypes = [ctypes.c_void_p]
    o.pwo_destroy.argtypes = [ctypes.c_void_p]
    o.pwo_realign_row.argtypes = [ctypes.c_void_p, ctypes.c_int]
    h = o.pwo_create(len(rows), len(rows[0]), b"".join(rows), bandwidth)
    o.pwo_trim(h)
    t0 = time.process_time()
    k = 0
    while k < len(rows) and time.process_time() - t0 < budget_s:
        o.pwo_realign_row(h, k)
        k += 1
    dt = time.process_time() - t0
    cells = o.pwo_cells(h)
    o.pwo_destroy(h)
    return {"value": cells / dt, "unit": "DP cells/s", "cores": 1, "kind": "port",
            "sample": f"first {k} row realignments of round 1 of the same MSA ({cells} cells, {dt:.1f} s CPU, "
                      f"oracle/pw_oracle.c -O2, array-based restatement pinned to the reference by tests/golden)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="tree_default")
    ap.add_argument("--bandwidth", type=int, default=1000)
    ap.add_argument("--window", type=int, default=None)
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--fill", type=int, default=None, help="DP fill kernel: 4 k_fill_v3 one work-group per pipeline wave (default), 3 k_fill_v2 one work-group per DP, 1 polled wave pipeline, 0 LDS-staged rows")
    ap.add_argument("--waves", type=int, default=None, help="waves per DP of the wave-pipeline fills: 9 (default), 8, 5, 4 or 3")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend for --gpus > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal on a 1-GPU box: every rank uses device 0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = 0 if args.one_device else local_rank
    if world > 1:
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)
    red_dev = "cuda" if (world > 1 and args.backend == "nccl") else "cpu"

    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    cfg = dg.CONFIGS[args.workload]
    cfg = dg.SimConfig(**{**cfg.__dict__, "seed": args.seed + rank})
    t0 = time.time()
    msa = dg.build_msa(dg.simulate(cfg))
    rows = [bytes(r) for r in msa]
    T, W0 = msa.shape
    del msa
    gen_s = time.time() - t0
    note(f"generated {T} rows x {W0} columns in {gen_s:.1f} s")

    g = PWReAligner(rows, bandwidth=args.bandwidth, device=dev, window=args.window, profile=True, threads=args.threads, fill=args.fill, waves=args.waves)
    g.trim_ends()
    score0 = g.total_score()            # first device call: uploads the MSA into HBM
    note(f"resident in HBM, score {score0}")

    def run_round(tag):
        """One realignment round with a heartbeat on stderr (a full-size round runs for minutes)."""
        import threading
        done = threading.Event()

        def beat():
            t_start = time.time()
            while not done.wait(60.0):
                note(f"{tag}: still running, {time.time() - t_start:.0f} s")
        th = threading.Thread(target=beat, daemon=True)
        th.start()
        try:
            g.realign_round()
        finally:
            done.set()
            th.join()

    for i in range(args.warmup):
        run_round(f"warm-up round {i + 1}")
        note(f"warm-up round {i + 1} done")
    g.reset_stats()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_round(f"timed round {i + 1}")
        note(f"timed round {i + 1} done")
    fence()
    dt = time.perf_counter() - t0
    st = g.stats()
    clk_mhz, fill_us = g.debug_fill_clock()
    dbg = g.debug_last_job(cap=4)
    print("debug: last job L=%d fill_us=%.1f clk=%.0f MHz rounds=%d" % (dbg["L"], fill_us, clk_mhz, dbg.get("rounds", -1)), file=sys.stderr)
    score1 = g.total_score()
    _, W1 = g.dims()

    cells = float(st["cells_reference"])
    tmax, csum = dt, cells
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cs = torch.tensor([cells], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
        tmax, csum = float(t.item()), float(cs.item())

    if rank == 0:
        fill_s = st["fill_ms"] / 1e3
        launches = max(1, st["fill_launches"])
        timed = max(1, st["fill_launches_timed"])
        # all launches are timed unless there were more than 65536 of them; scale the cells accordingly
        cells_timed = st["cells_computed"] * (timed / launches)
        achieved = (cells_timed * BYTES_PER_CELL / fill_s / 1e9) if fill_s > 0 else 0.0
        out = {
            "metric": "sum-of-pairs DP cells/sec",
            "value": csum / tmax,
            "unit": "DP cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{cfg.name} ({args.workload}: {cfg.kind}, {cfg.copies} copies, {cfg.coverage:g}x, "
                                   f"{cfg.repeat_len} bp) -> {T} rows x {W0} columns per GPU, bandwidth {args.bandwidth}, "
                                   f"one step = one realignment round",
                       "rows": T, "columns_in": W0, "columns_now": W1, "bandwidth": args.bandwidth,
                       "window": args.window, "score_before": score0, "score_after": score1,
                       "rows_committed": st["rows_committed"], "rows_recomputed": st["rows_recomputed"], "batches": st["batches"], "rows_changed": st["rows_changed"], "reject_reason": st["reject_reason"],
                       "fill_threads": args.threads, "shader_clock_mhz_last_fill": round(clk_mhz),
                       "generate_s": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (None if measured_traffic_per_cell() is None
                                     else measured_traffic_per_cell() * st["cells_computed"] / launches),
                         "traffic_unit": "HBM bytes per launch = PMC-measured bytes per cell (profiles/r01_traffic_model.json) "
                                         "x cells of the average launch",
                         "algorithmic_bytes_per_launch": BYTES_PER_CELL * st["cells_computed"] / launches,
                         "kernel": {None: "k_fill_v3", 4: "k_fill_v3", 3: "k_fill_v2", 1: "k_fill_wp", 0: "k_fill"}[args.fill], "launches": st["fill_launches"],
                         "avg_launch_ms": st["fill_ms"] / timed,
                         "cells_per_launch": st["cells_computed"] / launches,
                         "note": "achieved = cells computed by k_fill x 4 B / sum of HIP-event launch durations"},
        }
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU baseline on a bounded sample")
            out["cpu_baseline"] = cpu_baseline(rows, args.bandwidth)
        print(json.dumps(out), flush=True)
    g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
