#!/usr/bin/env python3
"""bench.py -- self-play throughput of the MI355X engine on BASELINE.json's headline configuration.

Workloads (--workload):
  c2 (default, the headline; BASELINE configs[1]): Connect4 7x6, 800 simulations/move, 4096 concurrent self-play games
     per GPU, network R4/F16/D16 random-init (weight seed 0), c_puct 0.85, temp 1, Beta prior noise alpha 0.2 / eps 0.3
     (always on in the reference, NetworkFactory.py:176-180), float32 results (by default formed on the bf16 matrix pipe
     from exactly split float32 operands, blackbird_amd/csrc/net_x3.hip.h; BB_NET_X3=0: float32 MFMA; `roofline.peak` is
     the ceiling of the form that ran).
  c5 (configs[4] on one GPU): the same game with the 20-block x 256-filter network (one step is 3.3 M evaluations of
     1.98 GFLOP: use --steps 1 --warmup 0 --prefill 1).
  dc (configs[3]): DragonChess, 400 simulations/move, 1024 concurrent games, R4/F16/D16 on 17 planes, 4032-wide policy
     head, ply cap 512 (the reference has no draw rule: documented deviation).

A "step" is `sims` tree visits per concurrent game, handed out by the persistent self-play kernel from one pool per launch
(or `sims` x (tree kernel + network kernel) + one move kernel in the launch-per-round modes).  A visit completes at least
one simulation (more when it meets terminal leaves whose value is already known), so a step is about one ply of every
game; `plies_per_step` says how many it was.  Finished games hand their slot to a fresh game, so the batch stays full;
`value` is games completed inside the timed region / wall time (whole job, all ranks).  The timed region = the K steps +
the extraction to the host of the example records of as many finished games as it produced (SURVEY.md 8d counts example
extraction in the metric).  Before warm-up the games are de-synchronised by an untimed prefill at few simulations per
move (otherwise all games would start and finish in lock-step and a short timed window would see no completions);
`games_per_sec_steady` = plies/s / mean plies of the games that finished in the timed region is the renewal-rate
estimate that does not depend on where the window falls.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), disjoint game-id/RNG streams per rank, no
collective in the data path; the (s, pi, z) examples are all-gathered once after timing, device to device
(blackbird_amd/dist.py; the epoch-end exchange of SURVEY.md 8e).  Under torch.distributed.run the ranks are what the
launcher made them; `python bench.py --gpus N` without a launcher starts that same command itself.

usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c5|dc]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from blackbird_amd import _lib, weights as W  # noqa: E402

FLOPS_PER_EVAL_C2 = 1_588_700  # SURVEY.md 8d / BASELINE.md: C2 network, conv + heads
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_16x16x32_bf16, 16 cycles per SIMD)
X3_PRODUCTS = 6                # bf16 MFMA products per float32 product of the split-operand tower (net_x3.hip.h)
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec (6290 measured copy)

WORKLOADS = {
    # game, H*W, planes, actions, blocks, filters, sims, slots, max_plies, prefill (plies, sims/move), steps, warmup, BASELINE configs index
    "c2": dict(game=_lib.GAME_CONNECT4, name="Connect4 7x6", hw=42, C=3, A=7, blocks=4, filters=16, sims=800, slots=4096,
               max_plies=42, prefill=(64, 32), steps=32, warmup=8, cfg=1, min_game_plies=7),
    "c5": dict(game=_lib.GAME_CONNECT4, name="Connect4 7x6", hw=42, C=3, A=7, blocks=20, filters=256, sims=800, slots=4096,
               max_plies=42, prefill=(64, 32), steps=32, warmup=8, cfg=4, min_game_plies=7),
    "dc": dict(game=_lib.GAME_DRAGONCHESS, name="DragonChess 8x8", hw=64, C=17, A=4032, blocks=4, filters=16, sims=400,
               slots=1024, max_plies=512, prefill=(96, 8), steps=32, warmup=4, cfg=3, min_game_plies=3),
}


def flops_per_eval(w):
    """F_eval = 2 HW 9 C F + 4 R HW 9 F^2 + 6 HW F + 4 A + 4 D  (D = 16)."""
    hw, C, F, R, A = w["hw"], w["C"], w["filters"], w["blocks"], w["A"]
    return 2 * hw * 9 * C * F + 4 * R * hw * 9 * F * F + 6 * hw * F + 4 * A + 4 * 16


def tree_bytes_per_sim(w, mean_depth, mean_children):
    """SURVEY.md 8d B_sim: 16 A_c d (select reads) + 16 (d+1) (backup) + 16 A_c + 2 S (expand) + HW C (leaf planes) +
    4 (A+1) (network outputs read back); A_c = child slots per stored node (7 dense for Connect4, the mean legal count
    for DragonChess), S = state bytes (24 / 104 as the survey counts them)."""
    S = 104 if w["game"] == _lib.GAME_DRAGONCHESS else 24
    return 16 * mean_children * mean_depth + 16 * (mean_depth + 1) + 16 * mean_children + 2 * S + w["hw"] * w["C"] + 4 * (w["A"] + 1)


def usable_cpus():
    """CPUs this process can actually run on at once: the affinity mask, cut down to the cgroup's CPU quota when there is
    one (a GPU box hands a 1-GPU job a share of the host, e.g. 16 of 256 hardware threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(round(int(parts[0]) / int(parts[1])))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(round(quota / int(f.read().split()[0])))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def host_cpu():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def cpu_baseline(key, w, flat, gpu_mean_plies, budget_s=20.0):
    """The oracle (a compiled "port" of the reference's serial algorithm: one game, one simulation, one batch-1 forward at
    a time, float32) on EVERY host core, one independent game per thread, same network / noise as the GPU run.
    Bounded to about `budget_s` seconds: c2 plays one full game per thread; for dc and c5 a full game would take minutes
    to hours, so every thread plays the first plies of a game and games/s is extrapolated from the measured plies/s
    (simulations/s for c5) with the mean game length the GPU run observed."""
    from oracle import orc
    og = {"c2": orc.C4, "c5": orc.C4, "dc": orc.DC}[key]
    H, Wd = (8, 8) if key == "dc" else (6, 7)
    ow = orc.NetWeights(H, Wd, w["C"], w["filters"], w["blocks"], 16, w["A"], flat)
    cfg = orc.make_cfg(og, evaluator=orc.EVAL_NET, net=ow, noise_on=True, alpha=0.2, eps=0.3, seed=1234)
    nproc, model = host_cpu()
    cores = usable_cpus()   # every core this job may use (nproc is what the host has)
    if key == "c2":
        sims, plies_cap, reps, what = w["sims"], 42, 3, "full Connect4 games at 800 sims/move"
    elif key == "dc":
        sims, plies_cap, reps, what = w["sims"], 100, 1, "the first 100 plies of a DragonChess game at 400 sims/move"
    else:
        sims, plies_cap, reps, what = 48, 1, 1, "one Connect4 ply of 48 simulations with the 20x256 network (~2 GFLOP per evaluation)"
    res = [None] * (cores * reps)

    def work(i):
        for r in range(reps):
            res[i * reps + r] = orc.selfplay_game(cfg, 10_000_000 + i * reps + r, 1.0, sims, plies_cap)

    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    tot_sims = sum(r["stats"].sims for r in res)
    plies = sum(r["n"] - 1 for r in res)
    out = {"unit": "games/s", "cores": cores, "nproc": nproc, "cpu_model": model, "kind": "port",
           "cores_note": "threads used = CPUs usable by this job (affinity mask / cgroup quota); nproc = host hardware threads",
           "port_note": "oracle/ C restatement of the reference's serial search; ONE network evaluation per simulation "
                        "(value + priors of the expanded node together) where the reference runs 1 + b batch-1 sess.run "
                        "calls per simulation (SURVEY.md 3.2), so this baseline is faster than the reference itself",
           "sims_per_s": tot_sims / dt, "plies_per_s": plies / dt, "wall_s": dt}
    if key == "c2":
        out["value"] = len(res) / dt
        out["sample"] = f"{len(res)} {what}, {reps} per host thread on {cores} threads, {plies} plies, {dt:.1f} s wall"
    else:
        per_game = max(gpu_mean_plies, 1.0)
        rate = (plies / dt) if key == "dc" else (tot_sims / dt / w["sims"])   # plies per second
        out["value"] = rate / per_game
        out["sample"] = (f"{cores} threads x {what} ({plies} plies, {tot_sims} simulations, {dt:.1f} s wall); games/s "
                         f"extrapolated = plies/s / {per_game:.1f} plies per game (the GPU run's mean finished-game length"
                         + ("; plies/s = simulations/s / 800)" if key == "c5" else ")"))
    return out


def pmc_traffic_bytes_per_second(key):
    """HBM bytes/s of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
    separate runs; profiles/README.md).  FETCH_SIZE is used as reported: the tree side reads scattered 4-32 B fields, not
    the wide streams for which MI355X_MICROARCH.md gives the x2 correction, so the read side is a lower bound."""
    for name in {"c2": ("r02_queue_pmc_summary.json", "r01_v6_queue_pmc_summary.json"),
                 "dc": ("r02_dc_pmc_summary.json",), "c5": ()}[key]:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                p = json.load(f)
            return (p["hbm_read_GBps_raw"] + p["hbm_write_GBps"]) * 1e9, name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--slots", type=int, default=None)
    ap.add_argument("--sims", type=int, default=None)
    ap.add_argument("--prefill", type=int, default=None, help="untimed de-synchronisation plies at few sims/move")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 = BASELINE configs[1] (default, the headline); c5 = configs[4]'s 20x256 network on one GPU "
                         "(use --steps 1 --warmup 0 --prefill 1); dc = configs[3] DragonChess 1024 games x 400 sims")
    args = ap.parse_args()
    w = dict(WORKLOADS[args.workload])
    K = w["steps"] if args.steps is None else args.steps
    Wm = w["warmup"] if args.warmup is None else args.warmup
    slots = w["slots"] if args.slots is None else args.slots
    sims = w["sims"] if args.sims is None else args.sims
    prefill = w["prefill"][0] if args.prefill is None else args.prefill
    w["sims"] = sims

    if args.gpus > 1 and "RANK" not in os.environ:
        # started by hand for several GPUs: become the launcher (a child process, before anything touches a GPU) --
        # the same command line the driver uses, one rank per GPU on this node
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # rehearsal hooks (1-GPU box): BB_BENCH_ONE_GPU=1 maps every rank to cuda:0, BB_BENCH_BACKEND=gloo avoids RCCL's
        # one-rank-per-device rule; the driver's real multi-GPU runs use neither
        if os.environ.get("BB_BENCH_ONE_GPU") == "1":
            local = 0
        backend = os.environ.get("BB_BENCH_BACKEND", "nccl")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    game = w["game"]
    flat = W.flatten(W.init_weights(w["C"], w["filters"], w["blocks"], 16, w["A"], seed=0))
    fpe = flops_per_eval(w)
    assert args.workload != "c2" or fpe == FLOPS_PER_EVAL_C2
    total_plies = prefill + Wm + K + 2
    # every slot finishes at most one game per `min_game_plies` plies (a visit can complete several simulations, so a
    # step can be more than one ply: x2 head-room)
    max_games = slots * (2 * total_plies // w["min_game_plies"] + 2)
    if args.workload == "dc":
        max_games = slots * 8  # DragonChess games last tens to hundreds of plies (mean ~80 under weak play)
    from blackbird_amd import dist as bdist
    first_id, seed = bdist.shard(rank, 1234)
    eng = _lib.Engine(game, n_slots=slots, sims_per_move=sims, evaluator=_lib.EVAL_NET, c_puct=0.85,
                      seed=seed, first_game_id=first_id, noise_on=True, alpha=0.2, epsilon=0.3,
                      device=local, max_games=max_games, max_plies=w["max_plies"])
    eng.load_weights(flat)
    eng.selfplay_begin(max_games, 1.0)
    if prefill > 0:
        eng.set_sims_per_move(min(w["prefill"][1], sims))
        eng.selfplay_step(prefill)
        eng.set_sims_per_move(sims)
    if Wm > 0:
        eng.selfplay_step(Wm)
    eng.synchronize()
    eng.reset_counters()
    eng.timing_enable(97)  # HIP events (engine stream) around every persistent launch / every 97th network launch

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    eng.selfplay_step(K)
    eng.synchronize()
    # example extraction belongs to the metric ("all kernels + host orchestration + example extraction")
    # -- as many finished games as the timed region produced (the oldest ids: complete records) come back to the host
    tf = time.perf_counter()
    cnt = eng.counters()
    rec_t, _offs_t, _win_t = eng.fetch_examples(0, max(int(cnt["games_finished"]), 1))
    fetch_s = time.perf_counter() - tf
    barrier()
    dt = time.perf_counter() - t0

    net_ms, net_min_ms, net_n = eng.timing_read()
    games, nsims, plies = cnt["games_finished"], cnt["sims"], cnt["plies"]
    # game lengths of what finished inside the timed region (game ids are dealt in order, so "finished" is not a prefix:
    # read every header once)
    rec_all, offs_all, win_all = eng.fetch_examples(0, max_games)
    lens = np.diff(offs_all)
    fin = np.nonzero(lens > 0)[0]
    # records of games that were finished when the timed region began carry nothing new: the timed region's games are
    # those whose LAST record's total-visit count belongs to a full-strength search or that finished after the snapshot;
    # the length statistic simply uses every finished game of the run that started at full strength
    first_tot = rec_all["total"][offs_all[fin]] if len(fin) else np.zeros(0)
    full = fin[first_tot >= sims - 1] if len(fin) else fin
    mean_plies_all = float((lens[fin] - 1).mean()) if len(fin) else 0.0
    # games that started at full strength AND finished inside a short run are the short ones: only trust their mean when
    # the run is long enough for typical games to be among them
    mean_plies_full = float((lens[full] - 1).mean()) if (len(full) >= 200 and total_plies - prefill >= w["max_plies"]) else 0.0
    if dist is not None:
        dev = f"cuda:{local}" if backend == "nccl" else None
        (games, nsims, plies), (dt,) = bdist.reduce_totals([games, nsims, plies], [dt], device=dev)
        # epoch-end exchange: all-gather the finished games' (s, pi, z) records
        import torch
        tg = time.perf_counter()
        if backend == "nccl":   # device-resident: engine store -> compaction on the GPU -> RCCL -> GPU
            allrec, _counts = bdist.allgather_engine_examples(eng, f"cuda:{local}")
            torch.cuda.synchronize()
            n_examples_all = int(allrec.shape[0])
        else:
            n_examples_all = int(len(bdist.allgather_records(rec_all)))
        allgather_s = time.perf_counter() - tg
    else:
        allgather_s = None
        n_examples_all = None

    rc = 0
    if rank == 0:
        mode = eng.selfplay_mode()
        gname = {"c2": "Connect4", "c5": "Connect4", "dc": "DragonChess"}[args.workload]
        form = eng.net_form()
        if mode == 3 and form == 2:
            kernel = f"k_selfplay_queue<{gname},8,x3> (8 network + 4 tree waves per CU, LDS work queue; tower on the bf16 matrix pipe, float32 by 3-way operand split)"
        elif mode == 3:
            kernel = f"k_selfplay_queue<{gname},8> (8 network + 4 tree waves per CU, LDS work queue)"
        elif mode == 5:
            kernel = "k_dc_selfplay_fused (one wave per game: tree step, network and move in the same wave)"
        elif w["filters"] != 16:
            kernel = ("k_gnet_conv_x3<%s,4,4> (bf16 matrix pipe, float32 by 3-way operand split)" if form == 3 else "k_gnet_conv<%s,false,4,2>") % gname + " x %d conv layers + first conv + heads per evaluation batch" % (2 * w["blocks"])
        else:
            kernel = (f"k_net_x3<{gname}>" if form == 2 else f"k_net_compact<{gname},4>" if mode == 1 else f"k_net_fused16<{gname}>")
        if mode >= 2:
            # persistent kernel: a launch covers up to 16 steps; its FLOPs are the evaluations it performed
            launches = max(net_n, 1)
            flops_per_launch = fpe * cnt["evals"] / launches
            sims_per_launch = cnt["sims"] / launches
        else:
            flops_per_launch = fpe * cnt["evals"] / max(K * sims, 1)
            sims_per_launch = cnt["sims"] / max(K * sims, 1)
        achieved = flops_per_launch / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
        # The roof of the arithmetic actually issued.  float32 MFMA forms: the f32 MFMA peak.  Split-operand form: every
        # float32 product is six bf16 MFMA products, so the ceiling for float32-equivalent FLOP/s is the bf16 peak / 6
        # (416.7 TFLOP/s); `achieved` stays the ALGORITHMIC float32 FLOPs of SURVEY.md 8d either way.
        x3 = form in (2, 3)
        peak = PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS if x3 else PEAK_F32_MFMA_TFLOPS
        mean_depth = cnt["sum_depth"] / max(cnt["sims"], 1)
        a_c = 7.0 if gname == "Connect4" else 14.6   # DragonChess: mean legal count of SURVEY.md 6 [probe]
        b_sim = tree_bytes_per_sim(w, mean_depth, a_c)
        tree_gbps = b_sim * sims_per_launch / (net_ms * 1e-3) / 1e9 if net_ms > 0 else 0.0
        traffic_rate, traffic_src = pmc_traffic_bytes_per_second(args.workload)
        # mean game length for the steady-state estimate and the CPU extrapolation: observed, or (no game finished in this
        # run) the value default runs of that workload observe
        mean_len = mean_plies_full or mean_plies_all or {"c2": 28.0, "c5": 28.0, "dc": 60.0}[args.workload]
        out = {
            "metric": "selfplay_games_per_sec", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": dt / max(K, 1) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (float32 operands split exactly into 3 bf16 values, 6 bf16 MFMA products each, f32 accumulate)" if x3 else "f32",
            "data": "synthetic (self-play from the initial position, random-init weights seed 0)",
            "config": {"workload": "%s, DynamicMCTS %d sims/move, %d concurrent games per GPU, net R%d/F%d/D16 fp32, "
                                   "noise alpha 0.2 eps 0.3 (BASELINE configs[%d])"
                                   % (w["name"], sims, slots, w["blocks"], w["filters"], w["cfg"]),
                       "game": gname, "sims_per_move": sims, "concurrent_games_per_gpu": slots,
                       "blocks": w["blocks"], "filters": w["filters"], "max_plies": w["max_plies"],
                       "step": "%d tree visits of every game (>= one ply each; see plies_per_step)" % sims,
                       "launch_structure": ["lockstep", "async-rounds", "retired", "persistent-queue", "retired", "wave-per-game"][mode],
                       "prefill": "%d plies at %d sims/move (untimed)" % (prefill, min(w["prefill"][1], sims)),
                       "parallelism": f"games sharded over {world} GPU(s), no data-path collective"},
            "node_evals_per_sec": nsims / dt, "net_evals_per_sec_rank0": cnt["evals"] / dt, "plies_per_sec": plies / dt,
            "plies_per_step": plies / max(K, 1) / (slots * world),
            "games_finished": games, "mean_plies_finished_games": mean_plies_all,
            "mean_plies_games_started_at_full_sims": mean_plies_full,
            "games_per_sec_steady": (plies / dt) / mean_len if mean_len > 0 else None, "mean_plies_used": mean_len,
            "examples_fetched": int(len(rec_t)), "examples_fetch_s": fetch_s,
            "terminal_leaf_fraction": cnt["terminal_leaves"] / max(cnt["sims"], 1),
            "mean_leaf_depth": mean_depth, "overflow": cnt["overflow"],
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "peak_note": ("bf16 MFMA dense peak 2500 / 6 products per float32 product; against the float32 MFMA "
                                       "peak (157.3) the same algorithmic rate is %.3f" % (achieved / PEAK_F32_MFMA_TFLOPS))
                                      if x3 else "float32 MFMA dense peak",
                         "bf16_tflops_issued": (achieved * X3_PRODUCTS * 48.0 / 42.0) if (form == 2 and gname == "Connect4") else None,
                         "traffic": (traffic_rate * net_ms * 1e-3) if (mode >= 2 and traffic_rate) else None,
                         "traffic_note": ("HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) rate from profiles/%s x this "
                                          "launch's duration" % traffic_src) if traffic_src else "no PMC pass committed for this workload",
                         "launch_ms_mean": net_ms, "launch_ms_min": net_min_ms,
                         "launches_timed": net_n, "flops_per_launch": flops_per_launch, "flops_per_eval": fpe,
                         "tree_side": {"bound": "hbm", "bytes_per_sim": b_sim, "achieved": tree_gbps, "peak": PEAK_HBM_GBPS,
                                       "unit": "GB/s", "frac": tree_gbps / PEAK_HBM_GBPS,
                                       "note": "SURVEY.md 8d B_sim x simulations of the launch / launch time: latency-bound "
                                               "pointer chasing, reported for completeness"}},
        }
        if allgather_s is not None:
            out["examples_allgather_s"] = allgather_s
            out["examples_gathered"] = n_examples_all
            out["examples_allgather_path"] = "device (engine store -> RCCL)" if backend == "nccl" else "host (gloo rehearsal)"
        if cnt["overflow"]:
            # a pool ran out or a persistent launch aborted: the games are not the specified workload any more
            out["value"] = None
            out["error"] = "overflow counter is %d: result invalid" % cnt["overflow"]
            rc = 1
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, w, flat, mean_len)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
