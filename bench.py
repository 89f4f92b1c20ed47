#!/usr/bin/env python3
"""bench.py -- self-play throughput of the MI355X engine on BASELINE.json's headline configuration.

Workload (configs[1], "C2"): Connect4 7x6, 800 simulations/move, 4096 concurrent self-play games
per GPU, network R4/F16/D16 random-init (weight seed 0), c_puct 0.85, temp 1, Beta prior noise
alpha 0.2 / eps 0.3 (always on in the reference, NetworkFactory.py:176-180), float32.

A "step" is one ply of every concurrent game: 800 visits of every game by the persistent self-play kernel (or
800 x (tree kernel + network kernel) + one move kernel in the launch-per-round modes).  Finished games hand their
slot to a fresh game, so the batch stays full; `value` is games completed inside the timed region / wall time (whole
job, all ranks).  The timed region = the K steps + the extraction to the host of the example records of as many
finished games as it produced (SURVEY.md 8d counts example extraction in the metric).  Before warm-up the
games are de-synchronised by an untimed prefill at 32 simulations/move (otherwise all 4096 games
would start and finish in lock-step and a short timed window would see no completions).

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), disjoint game-id/RNG
streams per rank, no collective in the data path; the (s, pi, z) examples of the timed region are
all-gathered once after timing (the epoch-end exchange of SURVEY.md 8e).  Under torch.distributed.run the ranks are
what the launcher made them; `python bench.py --gpus N` without a launcher starts that same command itself.

usage: python bench.py [--gpus N] [--steps K] [--warmup W]
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

FLOPS_PER_EVAL = 1_588_700  # SURVEY.md 8d / BASELINE.md: C2 network, conv + heads
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak


def cpu_baseline(flat, seconds_budget=20.0, sims=800):
    """Oracle ("port" of the reference's serial algorithm) on the host cores: one independent
    game per thread, same network/seed/noise as the GPU run.  Bounded: one game per thread."""
    from oracle import orc
    ow = orc.NetWeights(6, 7, 3, 16, 4, 16, 7, flat)
    cfg = orc.make_cfg(orc.C4, evaluator=orc.EVAL_NET, net=ow, noise_on=True, alpha=0.2, eps=0.3, seed=1234)
    cores = min(os.cpu_count() or 1, 16)
    res = [None] * cores

    def work(i):
        res[i] = orc.selfplay_game(cfg, 10_000_000 + i, 1.0, sims, 42)

    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    tot_sims = sum(r["stats"].sims for r in res)
    return {"value": cores / dt, "unit": "games/s", "cores": cores, "kind": "port",
            "sims_per_s": tot_sims / dt,
            "sample": f"{cores} full Connect4 games at {sims} sims/move, one per host thread, "
                      f"{sum(r['n'] - 1 for r in res)} plies, {dt:.1f} s wall"}


def pmc_traffic_bytes_per_second():
    """HBM bytes/s of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected in separate runs, profiles/r01_v6_queue_pmc_summary.json).  FETCH_SIZE is used as reported: the
    tree kernels read scattered 4-16 B fields, not the wide streams for which MI355X_MICROARCH.md gives the x2
    correction, so the read side is a lower bound."""
    path = os.path.join(ROOT, "profiles", "r01_v6_queue_pmc_summary.json")
    try:
        with open(path) as f:
            p = json.load(f)
        return (p["hbm_read_GBps_raw"] + p["hbm_write_GBps"]) * 1e9
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--slots", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--prefill", type=int, default=64, help="untimed de-synchronisation plies at 32 sims/move")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 = BASELINE configs[1] (default, the headline); c5 = the same game with the 20-block x "
                         "256-filter network of configs[4] (one ply is 3.3 M evaluations: use --steps 1 --warmup 0 --prefill 1)")
    args = ap.parse_args()

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

    game = _lib.GAME_CONNECT4
    K, Wm = args.steps, args.warmup
    blocks, filters = (20, 256) if args.workload == "c5" else (4, 16)
    flat = W.flatten(W.init_weights(3, filters, blocks, 16, 7, seed=0))
    # F_eval = 2 HW 9 C F + 4 R HW 9 F^2 + 6 HW F + 4 A + 4 D
    flops_per_eval = 2 * 42 * 9 * 3 * filters + 4 * blocks * 42 * 9 * filters * filters + 6 * 42 * filters + 4 * 7 + 4 * 16
    assert args.workload != "c2" or flops_per_eval == FLOPS_PER_EVAL
    total_plies = args.prefill + Wm + K + 2
    # every slot finishes at most one game per 7 plies (shortest Connect4 game)
    max_games = args.slots * (total_plies // 7 + 2)
    from blackbird_amd import dist as bdist
    first_id, seed = bdist.shard(rank, 1234)
    eng = _lib.Engine(game, n_slots=args.slots, sims_per_move=args.sims, evaluator=_lib.EVAL_NET, c_puct=0.85,
                      seed=seed, first_game_id=first_id, noise_on=True, alpha=0.2, epsilon=0.3,
                      device=local, max_games=max_games)
    eng.load_weights(flat)
    eng.selfplay_begin(max_games, 1.0)
    if args.prefill > 0:
        eng.set_sims_per_move(32)
        eng.selfplay_step(args.prefill)
        eng.set_sims_per_move(args.sims)
    if Wm > 0:
        eng.selfplay_step(Wm)
    eng.synchronize()
    eng.reset_counters()
    eng.timing_enable(97)  # HIP events around every 97th network launch of the timed region

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
    games, sims, plies = cnt["games_finished"], cnt["sims"], cnt["plies"]
    tot = np.array([games, sims, plies, dt], dtype=np.float64)
    if dist is not None:
        import torch
        t = torch.tensor(tot, device="cuda")
        tmax = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        games, sims, plies = [float(x) for x in t[:3].tolist()]
        dt = float(tmax[3])
        # epoch-end exchange: all-gather the finished games' (s, pi, z) records over RCCL
        from blackbird_amd import dist as bdist
        rec, offs, win = eng.fetch_examples(0, max_games)
        tg = time.perf_counter()
        allrec = bdist.allgather_records(rec, device=f"cuda:{local}")
        torch.cuda.synchronize()
        allgather_s = time.perf_counter() - tg
        n_examples_all = int(len(allrec))
    else:
        allgather_s = None
        n_examples_all = None

    if rank == 0:
        mode = eng.selfplay_mode()
        if mode >= 2:
            # persistent kernel: a launch covers up to 16 plies; its FLOPs are the evaluations it performed
            kernel = {2: "k_selfplay_mega<Connect4> (4 network + 4 tree waves per CU, lock-step phases)",
                      3: "k_selfplay_queue<Connect4,8> (8 network + 4 tree waves per CU, LDS work queue)",
                      4: "k_selfplay_team<Connect4,2> (2 teams of 3 network waves + 6 tree waves per CU)"}[mode]
            launches = max(net_n, 1)
            flops_per_launch = flops_per_eval * cnt["evals"] / launches
        else:
            kernel = "k_net_compact<Connect4,4>" if mode == 1 else "k_net_fused16<Connect4,4>"
            if filters != 16:
                kernel = "k_gnet_conv<Connect4,false,4,2> x %d conv layers + first conv + heads per evaluation batch" % (2 * blocks)
            flops_per_launch = flops_per_eval * cnt["evals"] / (K * args.sims)
        achieved = flops_per_launch / (net_ms * 1e-3) / 1e12 if net_ms > 0 else 0.0
        out = {
            "metric": "selfplay_games_per_sec", "value": games / dt, "unit": "games/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (self-play from the initial position, random-init weights seed 0)",
            "config": {"workload": "Connect4 7x6, DynamicMCTS 800 sims/move, %d concurrent games per GPU, "
                                   "net R%d/F%d/D16 fp32, noise alpha 0.2 eps 0.3 (BASELINE configs[%d])"
                                   % (args.slots, blocks, filters, 4 if args.workload == "c5" else 1),
                       "game": "Connect4", "sims_per_move": args.sims, "concurrent_games_per_gpu": args.slots,
                       "blocks": blocks, "filters": filters, "step": "800 tree+network rounds over all games (one ply in lock-step terms)",
                       "launch_structure": ["lockstep", "async-rounds", "persistent-phases", "persistent-queue", "persistent-teams", "wave-per-game"][mode],
                       "parallelism": f"games sharded over {world} GPU(s), no data-path collective"},
            "node_evals_per_sec": sims / dt, "net_evals_per_sec_rank0": cnt["evals"] / dt, "plies_per_sec": plies / dt,
            "games_finished": games, "examples_fetched": int(len(rec_t)), "examples_fetch_s": fetch_s, "terminal_leaf_fraction": cnt["terminal_leaves"] / max(cnt["sims"], 1),
            "mean_leaf_depth": cnt["sum_depth"] / max(cnt["sims"], 1), "overflow": cnt["overflow"],
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": (pmc_traffic_bytes_per_second() * net_ms * 1e-3) if (mode >= 2 and pmc_traffic_bytes_per_second()) else None,
                         "traffic_note": "HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) rate from profiles/r01_v6_queue_pmc_summary.json x this launch's duration",
                         "launch_ms_mean": net_ms, "launch_ms_min": net_min_ms,
                         "launches_timed": net_n, "flops_per_launch": flops_per_launch},
        }
        if allgather_s is not None:
            out["examples_allgather_s"] = allgather_s
            out["examples_gathered"] = n_examples_all
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            out["cpu_baseline"] = cpu_baseline(flat)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
