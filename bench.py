#!/usr/bin/env python3
"""bench.py -- sum-of-pairs DP cells/s of the MI355X PW_ReAligner hot path.

The workload is BASELINE.json configs[1]: the DataSimulator-default Tree_1perc_30000kb data set (100 copies, 40x,
30 kb), its reads aligned into the template by the InitialAligner (on the GPU, include/pia.h) and stacked into the MSA
exactly as the reference pipeline does before it calls PW_ReAligner (--input truth: the true alignments stacked
instead, rounds 1-2's workload); the MSA is resident in HBM when the timed region starts.  One full realignment
round of it (PW_ReAligner.c:1695-1737) takes tens of seconds, so a "step" is a SLAB of that round: the next
T/8 rows in input order (pwr_realign_rows, a partial k loop of PW:1695).  Slabs follow each other through the
round and on into the next rounds, exactly as the reference's loop would visit the rows, so K steps are K/8
rounds of the real computation.  `value` = DP cells the reference would have filled for the rows realigned in
the timed steps / wall time.

N > 1 (one process per GPU, torch.distributed): BASELINE.json configs[3] -- ONE MSA of the same shape is cut
into its Window.py sections (repeatresolver_amd/window.py), the sections are dealt to the ranks (no data-path
collective: a section is an independent MSA), a step realigns the next slab of every section, and the
totals are reduced over RCCL.  Fixed total work => "scaling": "strong".

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

T_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL = 4.0           # algorithmic bytes per DP cell (one 32-bit score per cell)
FILL_KERNELS = {4: "k_fill_v3", 3: "k_fill_v2"}


def measured_traffic():
    """HBM bytes per fill launch from a rocprofv3 --pmc run of THIS command (scripts/pmc_bench.sh writes
    profiles/bench_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes).  None when no such
    measurement is committed -- it is never modelled."""
    try:
        with open(os.path.join(ROOT, "profiles", "bench_traffic.json")) as f:
            d = json.load(f)
        return float(d["hbm_bytes_per_launch"]), d.get("command")
    except Exception:
        return None, None


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


def opt_or_none(ctx, key):
    """an option's value, None for a library that does not know it (scripts/dev/ab.sh runs older builds)"""
    try:
        return ctx.get_option(key)
    except Exception:
        return None


def slab_bounds(T, slabs, i):
    """Rows [k0, k1) of step i: slab i mod slabs of round i // slabs."""
    s = i % slabs
    return (s * T) // slabs, ((s + 1) * T) // slabs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="tree_default")
    ap.add_argument("--input", default="pipeline", choices=["pipeline", "truth"],
                    help="pipeline: the reads aligned into the template by the GPU InitialAligner and stacked by Building_MSA (what the "
                         "reference feeds PW_ReAligner); truth: the TRUE read-to-template alignments stacked with the same layout rule")
    ap.add_argument("--bandwidth", type=int, default=1000)
    ap.add_argument("--slabs", type=int, default=8, help="steps per realignment round (a step = T/slabs consecutive rows)")
    ap.add_argument("--sections", type=int, default=None,
                    help="Window.py parts the MSA is cut into (N > 1: default 6 = configs[3]; more parts than ranks are dealt by bases, longest first).  "
                         "With --gpus 1 the parts run as that many contexts side by side on the ONE GPU, each on its own stream: config 4's aggregate rate per GPU")
    ap.add_argument("--only-section", type=int, default=None, help="--gpus 1 --sections P: run only this section (what ONE section's chain reaches with the GPU to itself)")
    ap.add_argument("--window", type=int, default=None)
    ap.add_argument("--split", default="sections", choices=["sections", "rows"],
                    help="N > 1: sections = configs[3], the MSA cut into Window.py sections dealt to the ranks (default); rows = the WHOLE MSA on every "
                         "rank, every speculative batch of the round's k loop split over the ranks, placements all-gathered (intra_round.py)")
    ap.add_argument("--fill", type=int, default=None, help="DP fill kernel (see include/pwr.h)")
    ap.add_argument("--waves", type=int, default=None)
    ap.add_argument("--spec-len", type=int, default=None, help="percent a speculative row may be longer than its batch's first row")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="any other knob of pwr_set_option (include/pwr.h), e.g. seg_rows=512")
    ap.add_argument("--time-every", type=int, default=8, help="HIP events around every n-th fill launch (an event record costs ~6 us of stream time; 1 = every launch)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--deadline-s", type=float, default=500.0,
                    help="N = 1: once the process has run this long, the line is printed for the steps finished so far (no further step is started)")
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
        ndev = torch.cuda.device_count()
        if not args.one_device and ndev < world:
            raise SystemExit(f"bench.py --gpus {world}: one process per GPU needs {world} visible devices, this node shows {ndev} "
                             "(a rehearsal on fewer devices: --one-device --backend gloo)")
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)
    red_dev = "cuda" if (world > 1 and args.backend == "nccl") else "cpu"

    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.realigner import PWReAligner
    from repeatresolver_amd.window import slice_sections, window_boundaries

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')} +{time.time() - T_START:.0f}s] {msg}", file=sys.stderr, flush=True)

    cfg = dg.CONFIGS[args.workload]
    cfg = dg.SimConfig(**{**cfg.__dict__, "seed": args.seed})     # every rank generates the SAME MSA
    t0 = time.time()
    if args.input == "pipeline":
        from repeatresolver_amd.pipeline import initial_msa
        rows, ia_info = initial_msa(cfg, device=dev)
        T, W0 = len(rows), len(rows[0])
        note(f"InitialAligner: {ia_info}")
    else:
        msa = dg.build_msa(dg.simulate(cfg))
        rows = [bytes(r) for r in msa]
        T, W0 = msa.shape
        ia_info = None
        del msa
    gen_s = time.time() - t0
    note(f"generated {T} rows x {W0} columns in {gen_s:.1f} s")

    # ---- the units this rank owns
    split_rows_mode = world > 1 and args.split == "rows"
    if world > 1 and args.sections is None:
        args.sections = 6
    if (world == 1 and not args.sections) or split_rows_mode:
        units = [("whole MSA", rows)]
        bounds = None
        owned = [["whole MSA (replica)"] for _ in range(world)]
        load = [sum(len(r) - r.count(b"-") - r.count(b" ") for r in rows)] * world if split_rows_mode else [0]
        if split_rows_mode and args.window is None:
            args.window = max(3, world)                 # a batch has a job for every rank
    else:
        bounds = window_boundaries(rows, parts=args.sections)                 # Window.py:41-60
        secs = slice_sections(rows, bounds)
        # longest-processing-time deal by bases per section: identical on every rank
        cost = [sum(len(r) - r.count(b"-") - r.count(b" ") for r in s) for s in secs]
        load = [0] * world
        owner = [0] * len(secs)
        for p in sorted(range(len(secs)), key=lambda p: -cost[p]):
            r = min(range(world), key=lambda r: (load[r], r))
            owner[p] = r
            load[r] += cost[p]
        units = [(f"section {p} columns [{bounds[p]},{bounds[p + 1]})", secs[p]) for p in range(len(secs)) if owner[p] == rank and (args.only_section is None or p == args.only_section)]
        owned = [[p for p in range(len(secs)) if owner[p] == r] for r in range(world)]
        if args.window is None:
            # inside a section every row overlaps every other (0.99 commits per batch whatever the window): a third job per batch is
            # work for nothing, and with several sections side by side on a GPU it takes the SIMDs of the others' first jobs
            # (six sections on one GPU: window 1 428, 2 425, 3 464 ms per step, profiles/r04_option_sweeps.txt)
            args.window = 2
        note(f"sections {bounds} dealt as {owner}: " + ", ".join(f"rank {r} {owned[r] or 'IDLE'} ({load[r]} bases)" for r in range(world)))
    ctxs = []
    score0 = 0
    for _, urows in units:
        g = PWReAligner(urows, bandwidth=args.bandwidth, device=dev, window=args.window, profile=args.time_every, fill=args.fill, waves=args.waves,
                        **{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.opt})
        if args.spec_len is not None:
            g.set_option("spec_len", args.spec_len)
        g.trim_ends()
        score0 += g.total_score()            # first device call: uploads the MSA into HBM
        ctxs.append(g)
    note(f"resident in HBM, score {score0}")

    ctx_seconds = [0.0] * len(ctxs)
    splitters = []
    if split_rows_mode:
        from repeatresolver_amd.intra_round import SplitRound
        splitters = [SplitRound(ctxs[0], device=dev)]

    def run_step(i):
        k0, k1 = slab_bounds(T, args.slabs, i)
        if splitters:
            splitters[0].realign_rows(k0, k1 - k0)
        elif len(ctxs) == 1:
            ctxs[0].realign_rows(k0, k1 - k0)
        elif ctxs:
            def one(ci, g):
                t_ = time.perf_counter()
                g.realign_rows(k0, k1 - k0)
                ctx_seconds[ci] += time.perf_counter() - t_
            ths = [threading.Thread(target=one, args=(ci, g)) for ci, g in enumerate(ctxs)]   # the C calls release the GIL
            for th in ths:
                th.start()
            for th in ths:
                th.join()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def stats_sum():
        tot = {}
        for g in ctxs:
            st = g.stats()
            for k_, v in st.items():
                if isinstance(v, list):
                    tot[k_] = [a + b for a, b in zip(tot.get(k_, [0] * len(v)), v)]
                else:
                    tot[k_] = tot.get(k_, 0) + v
        return tot

    def report(steps_done, dt, final):
        """Builds and prints THE json line (rank 0)."""
        st = stats_sum() if ctxs else {"cells_reference": 0, "cells_computed": 0, "fill_ms": 0.0, "fill_launches": 0, "fill_launches_timed": 0,
                                       "rows_committed": 0, "rows_recomputed": 0, "batches": 0, "rows_changed": 0, "reject_reason": [0, 0, 0, 0],
                                       "stalls": 0, "rows_wide": 0, "rows_ahead": 0, "seg_jobs": 0, "segs": 0, "seg_fails": 0, "rows_jumped": 0}
        cells = float(st["cells_reference"])
        tmax, csum, tmin = dt, cells, dt
        per_rank = None
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            ts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(ts, t)
            cs = torch.tensor([cells], dtype=torch.float64, device=red_dev)
            css = [torch.zeros_like(cs) for _ in range(world)]
            dist.all_gather(css, cs)
            per_rank = [{"rank": r, "seconds": float(ts[r].item()), "cells": float(css[r].item())} for r in range(world)]
            tmax = max(p["seconds"] for p in per_rank)
            csum = per_rank[0]["cells"] if split_rows_mode else sum(p["cells"] for p in per_rank)   # (replicas: every rank commits every row)
            for p_ in per_rank:                    # a scaling curve must explain itself: 6 sections on 8 ranks leave 2 ranks idle
                p_["sections"] = owned[p_["rank"]]
                p_["bases"] = load[p_["rank"]]
                p_["idle_s"] = tmax - p_["seconds"] if owned[p_["rank"]] else tmax
        if rank != 0:
            return
        fill_s = st["fill_ms"] / 1e3
        launches = max(1, st["fill_launches"])
        timed = max(1, st["fill_launches_timed"])
        # the contract's cells are the REFERENCE's (executions of PW:1503-1510 for the rows realigned) -- warm-up rows and
        # speculative fills that were thrown away are not throughput.  Every launches/timed-th launch is bracketed by HIP events.
        achieved = (st["cells_reference"] * (timed / launches) * BYTES_PER_CELL / fill_s / 1e9) if fill_s > 0 else 0.0
        achieved_all = (st["cells_computed"] * (timed / launches) * BYTES_PER_CELL / fill_s / 1e9) if fill_s > 0 else 0.0
        traffic, traffic_cmd = measured_traffic()
        score1 = sum(g.total_score() for g in ctxs)
        W1 = sum(g.dims()[1] for g in ctxs)
        made = ("the reads cut to their repeat part, aligned into the template by the InitialAligner (GPU, placements identical to the reference's) "
                "and stacked by Building_MSA: the pipeline's real input" if args.input == "pipeline" else
                "the MSA stacks the TRUE read-to-template alignments with InitialAligner's layout rule -- it is not an InitialAligner product")
        if world == 1 and args.sections:
            wl = (f"{cfg.name} ({args.workload}; {made}) -> {T} rows x {W0} columns cut into {args.sections} Window.py sections {bounds} (BASELINE.json configs[3]), "
                  f"ALL of them on this one GPU as {len(ctxs)} contexts side by side, each on its own stream; one step = the next {T // args.slabs} rows of "
                  f"every section; value = the sections' reference cells together / wall time")
        elif world == 1:
            wl = (f"{cfg.name} ({args.workload}: {cfg.kind}, {cfg.copies} copies, {cfg.coverage:g}x, {cfg.repeat_len} bp; reads simulated with "
                  f"DataSimulator.py's distributions, seed {cfg.seed}; {made}) -> {T} rows x {W0} columns, bandwidth {args.bandwidth}; one step = "
                  f"{args.slabs}th of a realignment round = {T // args.slabs} consecutive rows, steps continue through successive rounds")
        elif split_rows_mode:
            wl = (f"{cfg.name} as above ({T} rows x {W0} columns; {made}), the WHOLE MSA as a replica on each of {world} ranks; every speculative batch "
                  f"of the k loop (window {args.window}) is split over the ranks -- rank r fills and traces the jobs j % {world} == r --, the new placements "
                  f"are all-gathered ({splitters[0].bytes_gathered / max(1, splitters[0].batches):.0f} B per batch and rank, {splitters[0].batches} batches) and "
                  f"every rank commits all of them; one step = {T // args.slabs} consecutive rows")
        else:
            wl = (f"{cfg.name} as above ({T} rows x {W0} columns; {made}) cut into {args.sections} Window.py sections {bounds}, sections dealt to "
                  f"{world} ranks by bases; one step = the next {T // args.slabs} rows of every section")
        out = {
            "metric": "sum-of-pairs DP cells/sec",
            "value": csum / tmax if tmax > 0 else 0.0,
            "unit": "DP cells/s",
            "n_gpus": world,
            "steps": steps_done,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * tmax / max(1, steps_done),
            "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": wl, "rows": T, "columns_in": W0, "columns_now": W1, "bandwidth": args.bandwidth,
                       "slabs_per_round": args.slabs, "rounds_timed": steps_done / args.slabs,
                       "score_before": score0, "score_after": score1,
                       "rows_committed": st["rows_committed"], "rows_recomputed": st["rows_recomputed"], "batches": st["batches"],
                       "rows_changed": st["rows_changed"], "reject_reason": st["reject_reason"],
                       # self-audit: a time-out repeated by the stand-in kernel, a 64-bit fill or a failed segment check would show here
                       "stalls": st["stalls"], "rows_wide": st["rows_wide"], "rows_ahead": st["rows_ahead"], "rows_jumped": st.get("rows_jumped", 0),
                       "useful_frac": (st["cells_reference"] / st["cells_computed"]) if st["cells_computed"] else None,
                       "commits_per_batch": (st["rows_committed"] / st["batches"]) if st["batches"] else None,
                       "seg_jobs": st["seg_jobs"], "segs": st["segs"], "seg_fails": st["seg_fails"],
                       "options": {k_: opt_or_none(ctxs[0], k_) for k_ in ("window", "fill", "waves", "spec_len", "seg_rows", "seg_max", "warm_pct", "src_start", "warm_adapt", "warm_min_pct", "warm_down_pm", "warm_up_pm", "warm_now", "plan_ahead", "plan_slack", "plan_evrate_x100", "evrate_x100", "fail_stops", "hard_rows", "hard_up_pm", "hard_down_pm", "hard_marked", "hard_fills", "hard_refail", "host_enqueue_us", "host_wait_us")} if ctxs else None,
                       "generate_s": round(gen_s, 1), "input": args.input, "initial_aligner": ia_info, "complete": bool(final)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_all_computed_cells": achieved_all / HBM_PEAK_GBS,
                         "frac_end_to_end": (csum * BYTES_PER_CELL / tmax / 1e9 / HBM_PEAK_GBS / max(1, world)) if tmax > 0 else None,
                         "traffic": traffic,
                         "traffic_note": (f"HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of `{traffic_cmd}` (profiles/bench_traffic.json)"
                                          + ("" if traffic_cmd and traffic_cmd.strip() == f"bench.py --steps {args.steps} --warmup {args.warmup}" and not args.opt and args.window is None
                                             else "; NOT this run's command: a committed measurement of another invocation")
                                          if traffic is not None else "no PMC pass of this command is committed"),
                         "algorithmic_bytes_per_launch": BYTES_PER_CELL * st["cells_reference"] / launches,
                         "kernel": FILL_KERNELS.get(ctxs[0].get_option("fill") if ctxs else 4, "k_fill"),
                         "launches": st["fill_launches"], "launches_timed": st["fill_launches_timed"],
                         "avg_launch_ms": st["fill_ms"] / timed,
                         "cells_per_launch": st["cells_reference"] / launches, "cells_computed_per_launch": st["cells_computed"] / launches,
                         "cells_reference": st["cells_reference"], "cells_computed": st["cells_computed"],
                         "note": "achieved = the REFERENCE's DP cells (PW:1503-1510 executions for the rows committed in the timed steps, SURVEY 8d) x 4 B / sum of "
                                 "HIP-event durations of the fill kernel's launches on the context's stream (rank 0's contexts; every launches/launches_timed-th "
                                 "launch is bracketed, an event record costs ~6 us of stream time).  frac_all_computed_cells counts every cell the kernel "
                                 "touched (warm-up rows of its segments, speculative fills thrown away: 1/config.useful_frac as many); frac_end_to_end = "
                                 "value x 4 B / peak, all kernels and gaps included"},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world == 1 and args.sections:
            out["per_section"] = [{"section": units[ci][0], "seconds_in_calls": round(ctx_seconds[ci], 3), "cells": g.stats()["cells_reference"],
                                   "stalls": g.stats()["stalls"], "batches": g.stats()["batches"]} for ci, g in enumerate(ctxs)]
        if world == 1 and not args.no_cpu_baseline and final:
            note("timing the CPU baseline on a bounded sample")
            out["cpu_baseline"] = cpu_baseline(rows, args.bandwidth)
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    for i in range(args.warmup):
        run_step(i)
        note(f"warm-up step {i + 1}/{args.warmup} done")
    for g in ctxs:
        g.reset_stats()
    ctx_seconds[:] = [0.0] * len(ctxs)
    fence()
    t0 = time.perf_counter()
    done = 0
    for i in range(args.steps):
        run_step(args.warmup + i)
        done = i + 1
        note(f"timed step {done}/{args.steps} done, {time.perf_counter() - t0:.1f} s")
        # A kill at the driver's limit must not leave the run without its line: past the deadline, report what is there.
        if world == 1 and done < args.steps and time.time() - T_START > args.deadline_s:
            note(f"deadline: reporting {done} finished steps")
            break
    fence()
    dt = time.perf_counter() - t0
    report(done, dt, final=(done == args.steps))
    for g in ctxs:
        g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
