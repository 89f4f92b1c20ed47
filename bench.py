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

A "step" is `sims` tree visits per concurrent game, handed out by the persistent self-play kernel from the launch's visit
pools (or `sims` x (tree kernel + network kernel) + one move kernel in the launch-per-round modes).  A visit completes at
least one simulation (more when it meets terminal leaves whose value is already known), so a step is about one ply of every
game; `plies_per_step` says how many it was.  Finished games hand their slot to a fresh game, so the batch stays full.

Untimed, before the W warm-up steps: a PREFILL at few simulations per move that de-synchronises the games (otherwise all
games would start and finish in lock-step) and a SETTLE phase of max_plies steps at full strength -- after it no game that was
begun during the prefill is left, so every game that finishes inside the timed region was played at `sims` simulations
per move from its first move to its last (`full_strength_fraction` = 1) and the window is a steady-state sample.
Timed: the K steps as >= 3 back-to-back launches (HIP events on the engine's stream around each: `roofline.launch_ms_*`) + the
extraction to the host of the example records of exactly the games that finished inside the region (found by the change
of their `done` words).  `value` = those games / wall time (whole job, all ranks); `games_per_sec_steady` = plies/s / their
mean length is the renewal-rate estimate of the same thing.  `api_games_per_sec` is what a caller of the drop-in
`Blackbird.GenerateTrainingSamples(model, 8192, 1.0)` gets end to end: protobuf blobs + one PutGames per game into sqlite,
host work overlapped with the GPU, the batch's start and tail included (Blackbird.py:219-268).

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
               max_plies=42, prefill=(64, 32), settle=44, steps=32, warmup=8, cfg=1, min_game_plies=7),
    "c5": dict(game=_lib.GAME_CONNECT4, name="Connect4 7x6", hw=42, C=3, A=7, blocks=20, filters=256, sims=800, slots=4096,
               max_plies=42, prefill=(64, 32), settle=0, steps=32, warmup=8, cfg=4, min_game_plies=7),
    "dc": dict(game=_lib.GAME_DRAGONCHESS, name="DragonChess 8x8", hw=64, C=17, A=4032, blocks=4, filters=16, sims=400,
               slots=1024, max_plies=512, prefill=(96, 8), settle=0, steps=32, warmup=4, cfg=3, min_game_plies=3),
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


def profile_figures(key):
    """Per-evaluation figures of the dominant kernel from the rocprofv3 PMC passes committed under profiles/ (separate passes,
    counters only: tools/profile_round.sh): bf16 MFMA operations issued (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 FLOP) and HBM bytes
    (FETCH_SIZE as reported + WRITE_SIZE: the tree side reads scattered 4-32 B fields, not the wide streams for which
    MI355X_MICROARCH.md gives the x2 correction, so the read side is a lower bound).  They are measurements of ANOTHER run of
    the same kernel: the line labels them `from_profile` with the commit they were taken at."""
    for name in {"c2": ("r03_queue_pmc_summary.json",), "dc": ("r03_dc_pmc_summary.json",), "c5": ()}[key]:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                p = json.load(f)
            return {"file": "profiles/" + name, "commit": p.get("commit"),
                    "mfma_bf16_flop_per_eval": p["mfma_bf16_instructions_per_eval"] * 16384.0,
                    "hbm_bytes_per_eval": p["hbm_bytes_per_eval"]}
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None


def api_games_per_sec(n_games=8192, sims=800):
    """Blackbird.GenerateTrainingSamples(model, n_games, 1.0) end to end on the drop-in API (Blackbird.py:219-268): engine +
    example extraction + state.proto blobs + one Conn.PutGames per game into a fresh sqlite database."""
    import tempfile
    from blackbird_amd import Blackbird, Connect4
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="bb_bench_api_"))
    try:
        net = {"blocks": 4, "filters": 16, "eval": {"dense": 16}, "hasTeacher": False,
               "policy": {"dirichlet": {"alpha": 0.2, "epsilon": 0.3}}, "training": {"optimizer": "adam"}}
        model = Blackbird.Model(Connect4.BoardState, "bench", {"explorationRate": 0.85, "playLimit": sims}, net)
        model._selfplay_engine(n_games)                      # engine creation + weight upload: one-time, untimed
        Blackbird.GenerateTrainingSamples(model, 64, 1.0)    # warm the code paths
        t0 = time.perf_counter()
        Blackbird.GenerateTrainingSamples(model, n_games, 1.0)
        dt = time.perf_counter() - t0
        stored = len(model.Conn.GetGames(model.Name, model.Version))
        # the same batch on the engine alone (no blobs, no sqlite): a finite batch cannot finish before its longest slot has
        # played its games one after another -- n_games / slots games of ~30 plies at ~25 ms per ply -- whatever the host does
        eng = model._batch_engine
        eng.set_rng_stream(12345, 0)
        t1 = time.perf_counter()
        eng.selfplay_begin(n_games, 1.0)
        while not eng.selfplay_done()[0]:
            eng.selfplay_step(4)
        dt_eng = time.perf_counter() - t1
        eng.close()
        model._batch_engine = None
        return {"value": n_games / dt, "unit": "games/s", "games": n_games, "wall_s": dt, "examples_stored": stored,
                "engine_only_same_batch_games_per_sec": n_games / dt_eng, "engine_only_same_batch_wall_s": dt_eng,
                "api_over_engine_same_batch": dt_eng / dt,
                "what": "Blackbird.GenerateTrainingSamples(model, %d, 1.0): %d concurrent games with slot refill, blobs + PutGames "
                        "into sqlite, host sink overlapped with the GPU; start-up and tail of the batch included.  "
                        "engine_only_same_batch = the same finite batch without the host sink: its games/s is below the "
                        "steady-state `value` because a batch ends with its longest chain of games (the tail), not because of host work"
                        % (n_games, min(n_games, 4096))}
    finally:
        os.chdir(cwd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--slots", type=int, default=None)
    ap.add_argument("--sims", type=int, default=None)
    ap.add_argument("--prefill", type=int, default=None, help="untimed de-synchronisation plies at few sims/move")
    ap.add_argument("--settle", type=int, default=None, help="untimed full-strength steps after the prefill (default: max_plies + 2 for c2)")
    ap.add_argument("--launches", type=int, default=3, help="launches the K timed steps are split into (>= 3 by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the GenerateTrainingSamples end-to-end leg (c2, one GPU)")
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
    settle = w["settle"] if args.settle is None else args.settle
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
    total_plies = prefill + settle + Wm + K + 2
    # every slot finishes at most one game per `min_game_plies` plies (a visit can complete several simulations, so a
    # step can be more than one ply: x2 head-room)
    max_games = slots * (2 * total_plies // w["min_game_plies"] + 2)
    if args.workload == "dc":
        max_games = slots * (8 + settle // 32)  # DragonChess games last tens to hundreds of plies (mean ~80 under weak play)
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
    if settle > 0:
        eng.selfplay_step(settle)   # untimed, full strength: afterwards every game in flight was begun at `sims` simulations per move
    if Wm > 0:
        eng.selfplay_step(Wm)
    eng.synchronize()
    eng.reset_counters()
    hdr0 = eng.selfplay_headers(0, max_games)
    eng.timing_enable(97)  # HIP events (engine stream) around every persistent launch / every 97th network launch

    def barrier():
        if dist is not None:
            dist.barrier()

    n_launch = max(1, min(args.launches, K))
    parts = [K // n_launch + (1 if i < K % n_launch else 0) for i in range(n_launch)]
    barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    for part in parts:          # exactly K steps, as back-to-back launches on the engine's stream
        eng.selfplay_step(part)
    eng.synchronize()
    # example extraction belongs to the metric ("all kernels + host orchestration + example extraction"):
    # the records of exactly the games that finished inside the timed region come back to the host
    tf = time.perf_counter()
    cnt = eng.counters()
    hdr1 = eng.selfplay_headers(0, max_games)
    new_ids = np.nonzero((hdr1[:, 3] != 0) & (hdr0[:, 3] == 0))[0]
    if len(new_ids):
        rec_t, offs_t, _win_t = eng.fetch_games(new_ids, int(hdr1[new_ids, 0].sum()))
    else:
        rec_t, offs_t = np.zeros(0, dtype=_lib.example_dtype(game)), np.zeros(1, dtype=np.int32)
    fetch_s = time.perf_counter() - tf
    barrier()
    dt = time.perf_counter() - t0

    net_ms, net_min_ms, net_n = eng.timing_read()
    games, nsims, plies = cnt["games_finished"], cnt["sims"], cnt["plies"]
    assert games == len(new_ids), (games, len(new_ids))   # the counter and the headers tell the same story
    # the games of the timed region: their lengths, and whether they were played at full strength from the first move (the first
    # record's visit total is playLimit - 1 on a fresh root)
    lens_new = hdr1[new_ids, 2].astype(np.float64) if len(new_ids) else np.zeros(0)
    first_tot = rec_t["total"][offs_t[:-1]] if len(new_ids) else np.zeros(0)
    full = first_tot >= sims - 1
    mean_plies_all = float(lens_new.mean()) if len(new_ids) else 0.0
    mean_plies_full = float(lens_new[full].mean()) if full.sum() >= 100 else 0.0
    full_fraction = float(full.mean()) if len(new_ids) else 0.0
    overflow_any = cnt["overflow"]
    if dist is not None:
        dev = f"cuda:{local}" if backend == "nccl" else None
        (games, nsims, plies, overflow_any), (dt,) = bdist.reduce_totals([games, nsims, plies, cnt["overflow"]], [dt], device=dev)
        # a pool ran out or a launch aborted on SOME rank: no rank's games are the specified workload any more

        # epoch-end exchange: all-gather the finished games' (s, pi, z) records
        import torch
        tg = time.perf_counter()
        if backend == "nccl":   # device-resident: engine store -> compaction on the GPU -> RCCL -> GPU
            allrec, _counts = bdist.allgather_engine_examples(eng, f"cuda:{local}")
            torch.cuda.synchronize()
            n_examples_all = int(allrec.shape[0])
        else:
            rec_all, _o, _w = eng.fetch_examples(0, max_games)
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
        prof = profile_figures(args.workload)
        # mean game length for the steady-state estimate and the CPU extrapolation: observed, or (no game finished in this
        # run) the value default runs of that workload observe
        mean_len = mean_plies_full or mean_plies_all or {"c2": 28.0, "c5": 28.0, "dc": 60.0}[args.workload]
        dc = args.workload == "dc"
        games_rate = games / dt
        out = {
            # DragonChess games last up to the 512-ply cap: a window of tens of steps finishes mostly games that were begun in the
            # weak-play prefill, so the headline there is plies/s (node_evals_per_sec beside it) and games/s is only quoted from
            # games played at full strength throughout (--settle 512), else null
            "metric": "selfplay_plies_per_sec" if dc else "selfplay_games_per_sec",
            "value": (plies / dt) if dc else games_rate, "unit": "plies/s" if dc else "games/s",
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
                       "settle": "%d steps at %d sims/move (untimed): no game from the prefill is left in flight" % (settle, sims),
                       "timed_launches": parts,
                       "parallelism": f"games sharded over {world} GPU(s), no data-path collective"},
            "node_evals_per_sec": nsims / dt, "net_evals_per_sec_rank0": cnt["evals"] / dt, "plies_per_sec": plies / dt,
            "evals_timed_rank0": cnt["evals"], "sims_timed": nsims,
            "plies_per_step": plies / max(K, 1) / (slots * world),
            "games_finished": games, "games_per_sec_window": games_rate, "mean_plies_finished_games": mean_plies_all,
            "mean_plies_games_started_at_full_sims": mean_plies_full, "full_strength_fraction": full_fraction,
            "games_per_sec_full_strength": ((plies / dt) / mean_plies_full) if mean_plies_full > 0 else None,
            "games_per_sec_steady": (plies / dt) / mean_len if mean_len > 0 else None, "mean_plies_used": mean_len,
            "examples_fetched": int(len(rec_t)), "examples_fetch_s": fetch_s,
            "examples_fetch": "records of the games whose `done` word turned on inside the timed region (bb_examples_fetch_games)",
            "terminal_leaf_fraction": cnt["terminal_leaves"] / max(cnt["sims"], 1),
            "mean_leaf_depth": mean_depth, "overflow": cnt["overflow"],
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "peak_note": ("bf16 MFMA dense peak 2500 / 6 products per float32 product; against the float32 MFMA "
                                       "peak (157.3) the same algorithmic rate is %.3f" % (achieved / PEAK_F32_MFMA_TFLOPS))
                                      if x3 else "float32 MFMA dense peak",
                         # counters of ANOTHER run of this kernel (rocprofv3 --pmc passes committed under profiles/), per
                         # evaluation, scaled by this launch's evaluations -- never a measurement of the timed run itself
                         "bf16_tflops_issued": (prof["mfma_bf16_flop_per_eval"] * flops_per_launch / fpe / (net_ms * 1e-3) / 1e12)
                                               if (prof and x3 and mode >= 2 and net_ms > 0) else None,
                         "traffic": (prof["hbm_bytes_per_eval"] * flops_per_launch / fpe) if (prof and mode >= 2) else None,
                         "traffic_source": ("from_profile: %s (commit %s): SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 FLOP and (FETCH_SIZE + "
                                            "WRITE_SIZE) bytes per evaluation x this launch's evaluations" % (prof["file"], prof["commit"]))
                                           if prof else "no PMC pass committed for this workload",
                         "launch_ms_mean": net_ms, "launch_ms_min": net_min_ms, "launch_ms_total": net_ms * net_n,
                         "launches_timed": net_n, "flops_per_launch": flops_per_launch, "flops_per_eval": fpe,
                         "tree_side": {"bound": "hbm", "bytes_per_sim": b_sim, "achieved": tree_gbps, "peak": PEAK_HBM_GBPS,
                                       "unit": "GB/s", "frac": tree_gbps / PEAK_HBM_GBPS,
                                       "note": "SURVEY.md 8d B_sim x simulations of the launch / launch time: latency-bound "
                                               "pointer chasing, reported for completeness"}},
        }
        if allgather_s is not None:
            out["rccl_ranks"] = dist.get_world_size()
            out["collective_backend"] = backend + (" (RCCL over xGMI)" if backend == "nccl" else "")
            out["examples_allgather_s"] = allgather_s
            out["examples_gathered"] = n_examples_all
            out["examples_allgather_path"] = "device (engine store -> RCCL)" if backend == "nccl" else "host (gloo rehearsal)"
        if overflow_any:
            # a pool ran out or a persistent launch aborted (on any rank): the games are not the specified workload any more
            out["value"] = None
            out["error"] = "overflow counter is %d (summed over ranks): result invalid" % overflow_any
            rc = 1
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, w, flat, mean_len)
            gps = games_rate if not dc else (out["games_per_sec_full_strength"] or out["games_per_sec_steady"])
            out["gpu_over_cpu"] = gps / out["cpu_baseline"]["value"] if (gps and out["cpu_baseline"]["value"]) else None
    eng.close()
    if rank == 0:
        if rc == 0 and world == 1 and args.workload == "c2" and not args.no_api:
            api = api_games_per_sec(sims=sims)   # (the engine above is closed: the drop-in model creates its own)
            out["api_games_per_sec"] = api["value"]
            out["api"] = api
            out["api_over_engine"] = api["value"] / games_rate if games_rate else None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
