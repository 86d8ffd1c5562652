#!/usr/bin/env python3
"""Headline benchmark: molecules/sec of 1000-step DDPM sampling at batch 256 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one reverse-diffusion step (one score evaluation + posterior update) of one batch of
256 synthetic MOSES-sized molecules per GPU; value = molecules finished per second for 1000-step
chains = (batch * n_gpus) / (1000 * seconds_per_step).  With the default K = 1000 the timed region
IS one complete chain per GPU (plus, for N > 1, the RCCL gather of the generated molecules).
Inputs are resident in HBM before the timed region; noise is generated on the device (Philox)
inside it; per-step trajectories are written to HBM inside it (their D2H copy is reported
separately as `traj_d2h_ms`, never part of `value`).

Before the W warm-up steps the flags ask for, an untimed chain of `--clock-warmup` steps (default 200, ~0.1 s; reported as
`clock_warmup_steps`) brings the device to its loaded clocks: measured on MI355X, the first ~40 reverse steps after an idle
period run 7-9 % slower (0.54 -> 0.50 ms/step at B = 256), which a 5-step warm-up in front of a 20-step
timed region would otherwise charge to every step of the metric's 1000-step chain.  `--clock-warmup 0` switches it off.

Also on the JSON line: `roofline` of the dominant kernel -- per-launch time from the begin/end
timestamps of the kernel dispatches themselves (hipExtLaunchKernelGGL start/stop events on the launch
stream, the quantity rocprofv3's kernel trace reports), measured live in a short pass after the timed
region -- and `cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample; rank 0,
N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import yaml  # noqa: E402
from shapemol_amd import ScorePosNet3D, synth  # noqa: E402
from shapemol_amd.dist import GatherPlan, gather_molecules  # noqa: E402
from shapemol_amd.runtime import ChainRunner  # noqa: E402

TRAIN_YML = os.path.join(ROOT, "config", "training",
                         "dgcnn_signeddist_512_attention_residue_uniform_pos0_10_pos1.e-7_0.01_6_v001.yml")
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0      # dense f16 / bf16 matrix peak (same guide)
HBM_PEAK_GBS = 8000.0              # HBM3E peak (same guide)
PROFILE_DIR = "r04"          # profiles/<dir>/pmc_traffic*.json hold the PMC traffic of the kernels timed here
CHAIN_STEPS = 1000                 # the metric is quoted for 1000-step chains
# what the arithmetic is: fp32 inputs, outputs, accumulators and vector work; every matrix operand is split EXACTLY into three
# bf16 pieces (8 + 8 + 8 = the 24 significand bits of fp32, exact for every value of magnitude >= 2^-110) and a product is the six piece products of
# total order <= 2 with fp32 accumulation: each dropped term is below 2^-24 |x w| (mid x lo, lo x mid; lo x lo below 2^-32).  This is the library's default mode and what
# `value` is measured on; `f16x2_mode` on the JSON line times the same chain with two-piece f16 operands (22-23 bits, three
# products: the round-2/3 kernels), `f16_features_mode` with single f16 pieces (11 bits) -- optional modes, never `value`.
DTYPE = "f32 (matrix operands split exactly into 3 x bf16 pieces = 24 bits, six piece products, fp32 accumulate)"
PIECE_PRODUCTS = 6                 # matrix instructions per fp32 product in the headline mode


def executed_flops_per_atom_step(H, L, k, G=20, heads=16, C=15, S=32):
    """FLOPs the kernels execute (factorised first layers), MAC = 2."""
    edge_x2h = 2 * (G * H + H * H) * 2
    edge_h2x = ((G * H + H * H) + (G * H + H * heads)) * 2
    node = (4 * H * H + 2 * H * H) * 2 * 2 + (2 * H * H + H * H) * 2 + (1 + heads + 0) * heads * 3 * 2 * 2
    per_layer = k * (edge_x2h + edge_h2x) + node
    return L * per_layer + k * (G * H + H) * 2 + (H * H + H * C) * 2, edge_x2h


def reference_flops_per_atom_step(k, L=8):
    """reference formulation (full 308-wide first layers)."""
    return L * (k * 417792 + 238784) + k * 5376 + 42496


CPU_THREADS_DEFAULT = 16           # tools/cpu_baseline_sweep.py (profiles/r03/cpu_baseline_sweep.json): the fastest of 8 / 16 / 32 / 64 on the GPU box


def cpu_baseline(cfg, batch, n_steps, threads=0):
    """The CPU oracle on the same workload, timed on this host (bounded sample)."""
    from oracle import shapemol_oracle as O
    # the box's CPU share (16 cores per GPU) is smaller than os.cpu_count(): oversubscribing OpenMP stalls the run
    torch.set_num_threads(threads if threads > 0 else max(1, min(torch.get_num_threads(), len(os.sched_getaffinity(0)), CPU_THREADS_DEFAULT)))
    sd = O.state_dict_from_numpy(synth.synthetic_state_dict(cfg, seed=7))
    dm = O.Dims(cfg)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    n = len(batch["batch"])
    args = (T(batch["init_pos"]), T(batch["init_v"]), T(batch["batch"]), T(batch["shape"]))
    noise = lambda s: synth.step_noise(n, dm.C, s, seed=1)  # noqa: E731
    O.sample_chain(sd, dm, *args, 1, noise, keep_traj=False)          # warm-up
    t0 = time.perf_counter()
    O.sample_chain(sd, dm, *args, n_steps, noise, keep_traj=True)
    dt = (time.perf_counter() - t0) / n_steps
    return dt, torch.get_num_threads()


def lib_sha16():
    """First 16 hex digits of the SHA-1 of the loaded library: PMC summaries are keyed to the build they were measured on."""
    import hashlib
    from shapemol_amd import _lib
    try:
        return hashlib.sha1(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--clock-warmup", type=int, default=200, help="untimed reverse steps before the warm-up, to reach the loaded clocks (0 = off)")
    ap.add_argument("--batch", type=int, default=256, help="molecules per GPU (BASELINE config 2: 256)")
    ap.add_argument("--atoms", type=str, default="", help="lo,hi: uniform atom counts instead of the MOSES prior (configs[4]: 40,80)")
    ap.add_argument("--knn", type=int, default=0, help="override the model's k (configs[4]: 32)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: still initialise a 1-rank process group of --backend and run the final gather of the molecules "
                         "through its packing / all_gather_into_tensor / unpacking path (exercises the RCCL branch on a one-GPU box)")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--cpu-steps", type=int, default=20, help="reverse steps of the CPU oracle to time (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU oracle (0 = the measured best, see CPU_THREADS_DEFAULT)")
    ap.add_argument("--exact-steps", "--mode-steps", dest="exact_steps", type=int, default=-1,
                    help="reverse steps of the optional-mode chains (extra fields `f16x2_mode`, `f16_features_mode`; -1 = as --steps, 0 = skip)")
    ap.add_argument("--no-traj", action="store_true", help="do not keep per-step trajectories")
    ap.add_argument("--eager", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--concurrent", type=int, default=2,
                    help="also report the throughput of this many independent batch-256 chains run concurrently on the GPU "
                         "(extra field, never `value`; 0/1 = skip)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="context option for kernel experiments (shapemol_set_option), e.g. --opt x2h_chain=0; recorded in config")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()          # rehearsal: ranks share the GPUs that exist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    cfg = yaml.safe_load(open(TRAIN_YML))["model"]
    if args.knn:
        cfg["knn"] = args.knn
    atoms_range = tuple(int(x) for x in args.atoms.split(",")) if args.atoms else None
    model = ScorePosNet3D(cfg, 15)
    sdn = synth.synthetic_state_dict(cfg, seed=7)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
    model = model.to(dev)
    for kv in args.opt:
        name, val = kv.split("=")
        model.set_option(name, int(val))
    steps, warm = min(args.steps, CHAIN_STEPS), max(0, min(args.warmup, CHAIN_STEPS))
    clock_warm = max(0, min(args.clock_warmup, CHAIN_STEPS))
    if args.exact_steps < 0:
        args.exact_steps = steps

    bb = synth.synthetic_batch(args.batch, seed=2021 + rank, atoms_range=atoms_range)     # every rank owns a different batch
    n_atoms = len(bb["batch"])
    runner = ChainRunner(model, n_atoms, args.batch, max(steps, warm, clock_warm, 1), keep_traj=not args.no_traj, device=dev)
    runner.load_batch(bb["init_pos"], bb["init_v"], bb["batch"], bb["shape"])
    counts = torch.from_numpy(bb["counts"]).to(dev)
    # job set-up: the ranks exchange their (atoms, molecules) sizes once -- atom counts are drawn before a sampling job's chains
    # start -- so that the gather behind the chain is a single collective
    plan = GatherPlan(n_atoms, args.batch, dev) if dist is not None else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not args.eager
    log(f"rank {rank}/{world}: {args.batch} molecules, {n_atoms} atoms; warmup {warm} steps")
    if warm:
        runner.run(warm, seed=11, use_graph=use_graph)
        runner.synchronize()
        if dist is not None:
            gather_molecules(runner.out_pos, runner.out_v, counts, _single_rank_too=True, plan=plan)
    # The clock warm-up comes LAST, behind a barrier that aligns the ranks (and behind everything that initialises lazily: the
    # process group's first collectives, the first launches of the gather's kernels): a device that idles >= 5 ms in front of the
    # timed region falls back to low clocks and runs its first steps up to 10 % slower (tools/idle_gap_probe.py: 20 steps after
    # 0-2 / 5 / 20 / 100 ms of idling take 10.0 / 10.2 / 11.0 / 11.1 ms).  After it every rank reaches the barrier before t0 at
    # loaded clocks and within microseconds of the others.
    barrier()
    if clock_warm:
        runner.run(clock_warm, seed=10, use_graph=use_graph)
        runner.synchronize()
    log(f"timing {steps} steps")
    barrier()
    t0 = time.perf_counter()
    runner.run(steps, seed=12, use_graph=use_graph)
    t_enq = time.perf_counter() - t0
    runner.synchronize()
    t_chain = time.perf_counter() - t0
    if dist is not None:
        g_pos, g_v, g_counts = gather_molecules(runner.out_pos, runner.out_v, counts, _single_rank_too=True, plan=plan)
    t_gather = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    log(f"timed region: enqueue {t_enq * 1e3:.3f} ms, chain done {t_chain * 1e3:.3f}, gather enqueued {t_gather * 1e3:.3f}, barrier passed {elapsed * 1e3:.3f}")
    if dist is not None and world == 1:       # --force-collective: the gathered molecules are this rank's own, bit for bit
        assert torch.equal(g_pos, runner.out_pos) and torch.equal(g_v, runner.out_v) and torch.equal(g_counts, counts)
    local_elapsed = elapsed
    if dist is not None:
        cdev = dev if args.backend == "nccl" else torch.device("cpu")
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        natoms_all = torch.tensor([n_atoms], dtype=torch.int64, device=cdev)
        dist.all_reduce(natoms_all)
        total_atoms = int(natoms_all.item())
    else:
        total_atoms = n_atoms
    sec_per_step = elapsed / steps
    value = args.batch * world / (CHAIN_STEPS * sec_per_step)
    rank_ms = None
    if dist is not None:       # per-rank step time (the spread shows stragglers; `value` uses the max)
        mine = torch.tensor([local_elapsed / steps * 1e3], dtype=torch.float64, device=cdev)
        allr = torch.empty(world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine)
        rank_ms = [round(float(x), 4) for x in allr.cpu()]

    out = {
        "metric": f"molecules/sec (1000-step DDPM sample, batch {args.batch})", "value": round(value, 3),
        "unit": "molecules/s", "n_gpus": world, "steps": steps, "warmup": warm, "clock_warmup_steps": clock_warm,
        "ms_per_step": round(sec_per_step * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": (("BASELINE configs[1]: batch 256" if args.batch == 256 else
                                 ("BASELINE configs[2]/[3] size: batch 1024" if args.batch == 1024 else f"batch {args.batch}")) +
                                (f" molecules of {atoms_range[0]}-{atoms_range[1]} atoms" if atoms_range else " MOSES-prior molecules (9-27 atoms)") +
                                f" per GPU, 1000-step chain, fp32 results, k={cfg['knn']}, synthetic shapes + hash-filled weights" +
                                (f"; besides the {warm} warm-up steps an untimed {clock_warm}-step chain runs first to reach the loaded clocks" if clock_warm else "")),
                   "batch_per_gpu": args.batch, "atoms_per_gpu": n_atoms, "total_atoms": total_atoms,
                   "noise": "device Philox", "trajectories": "kept in HBM" if not args.no_traj else "off",
                   "launch": "hipGraph replay" if use_graph else "eager", "parallelism": f"dp{world} (whole batches per rank)",
                   "ranks_seen": (dist.get_world_size() if dist is not None else 1), **({"collective": f"forced 1-rank {args.backend} gather"} if (dist is not None and world == 1) else {}), "ms_per_step_by_rank": rank_ms, **({"options": args.opt} if args.opt else {}),
                   "matrix_products": "edge and node MLPs: every operand split exactly into three bf16 pieces (24 significand bits), six "
                                      "products per term on v_mfma_f32_16x16x32_bf16, fp32 accumulate; everything else fp32 (library defaults "
                                      "edge_bf16 = 2, node_f16 = 0)"},
    }

    log(f"timed region done: {elapsed:.3f} s, {value:.2f} molecules/s")
    if rank == 0:
        # ---- roofline of the dominant kernel, from a short event-timed eager pass -------------
        prof = runner.profile(max(1, args.profile_steps), seed=13)
        runner.synchronize()
        dm = model.dims
        tot_ms = sum(v[0] for v in prof.values())
        dom = max(prof, key=lambda k: prof[k][0])
        dom_ms, dom_n = prof[dom]
        avg_s = dom_ms * 1e-3 / dom_n
        f_exec_total, f_edge_x2h = executed_flops_per_atom_step(dm.H, dm.L, dm.k, dm.G, dm.heads, dm.C, dm.S)
        HH = dm.H * dm.H
        per_launch = {"edge_x2h": f_edge_x2h * dm.k * n_atoms,
                      "edge_h2x": ((dm.G * dm.H + HH) + (dm.G * dm.H + dm.H * dm.heads)) * 2 * dm.k * n_atoms,
                      "node_chain": 14 * HH * n_atoms,          # out MLP (2H->H->H) + two follow-up MLPs (H->H->H)
                      "node_pre": 16 * HH * n_atoms}            # 8H x H paired products (the first launch does 4H)
        per_launch["edge_x2h_chain"] = per_launch["edge_x2h"] + per_launch["node_chain"]      # x2h_chain16_kernel: both in one launch
        # larger batches run a class's work of a layer as several launches (slices): FLOPs per ACTUAL launch
        slices = max(1.0, dom_n / (8.0 * max(1, args.profile_steps)))
        flops = per_launch.get(dom, 0.0) / slices
        ach = flops / avg_s / 1e12 if flops else 0.0
        # HBM bytes per launch of that kernel from the committed PMC passes of the same workload (rocprofv3 cannot run inside this
        # process).  The summary records the SHA-1 of the library it was measured on: another build -> null, not a stale number
        traffic, traffic_src = None, None
        try:
            name = "pmc_traffic.json" if args.batch == 256 else f"pmc_traffic_b{args.batch}.json"
            pmc = json.load(open(os.path.join(ROOT, "profiles", PROFILE_DIR, name)))
            if pmc.get("lib_sha16") and pmc.get("lib_sha16") == lib_sha16() and not args.opt and not args.knn:
                traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{PROFILE_DIR}/{name} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, library {pmc['lib_sha16']})"
        except Exception:
            pass
        # matrix instructions the dominant kernel really issues (bf16 piece products, K padding of the RBF block included), against
        # the 2.5 PFLOP/s dense bf16 peak
        PP = PIECE_PRODUCTS
        pieces = {"edge_x2h": PP * 2 * (2 * HH + 2 * 32 * dm.H) * dm.k * n_atoms,
                  "edge_h2x": PP * 2 * ((HH + 32 * dm.H) + (16 * dm.H + 32 * dm.H)) * dm.k * n_atoms,
                  "node_chain": PP * 14 * HH * n_atoms, "node_pre": PP * 16 * HH * n_atoms}
        pieces["edge_x2h_chain"] = pieces["edge_x2h"] + pieces["node_chain"]
        pieces = pieces.get(dom)
        if pieces:
            pieces /= slices
        step_exec = f_exec_total * n_atoms / sec_per_step / 1e12
        # ceiling of the formulation the kernel runs: every fp32 product is six bf16 piece products on the matrix cores, so the
        # algorithm's fp32 FLOPs are bounded by the dense bf16 peak / 6
        peak_equiv = F16_MFMA_PEAK_TFLOPS / PP
        out["roofline"] = {
            "bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": round(peak_equiv, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak_equiv, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "hbm_frac": (round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
            "matrix_pipe_frac": (round(pieces / avg_s / 1e12 / F16_MFMA_PEAK_TFLOPS, 4) if pieces else None),
            "flops_per_launch_executed": flops, "avg_launch_us": round(avg_s * 1e6, 2), "launches": dom_n,
            "share_of_step": round(dom_ms / tot_ms, 3),
            "timing": "begin/end timestamps of the kernel dispatches (hipExtLaunchKernelGGL start/stop events), eager pass "
                      f"of {max(1, args.profile_steps)} steps right after the timed region; rocprofv3 --kernel-trace --stats of the "
                      f"same command: profiles/{PROFILE_DIR}/kernel_stats.csv",
            "note": "achieved: fp32 FLOPs of the kernel's algorithm (factorised first Linears) per second; peak: the dense bf16 MFMA peak "
                    "(2.5 PFLOP/s) / 6, because each fp32 product is evaluated exactly as six bf16 piece products (24-bit operands, fp32 "
                    "accumulate).  matrix_pipe_frac = bf16 piece-product FLOPs actually issued (K padding included) / 2.5 PFLOP/s; "
                    "hbm_frac = PMC HBM bytes per launch / launch time / 8 TB/s (Infinity-Cache resident re-reads, not a binding roof)",
            "step_tflops_executed": round(step_exec, 3),
            "step_tflops_ref_equiv": round(reference_flops_per_atom_step(dm.k, dm.L) * n_atoms / sec_per_step / 1e12, 3),
            "breakdown_ms_per_step": {k: round(v[0] / max(1, args.profile_steps), 4) for k, v in prof.items()},
        }
        log("profile pass done: " + ", ".join(f"{k}={v[0] / max(1, args.profile_steps):.3f}ms" for k, v in prof.items()))
        # ---- throughput mode: independent chains side by side (extra field; `value` stays the single-chain number) ----
        if world == 1 and args.concurrent > 1 and use_graph:
            runners = [runner]
            for j in range(1, args.concurrent):
                mj = ScorePosNet3D(cfg, 15)
                mj.load_state_dict({k: torch.from_numpy(v) for k, v in sdn.items()}, strict=True)
                mj = mj.to(dev)
                bj = synth.synthetic_batch(args.batch, seed=3021 + j, atoms_range=atoms_range)
                rj = ChainRunner(mj, len(bj["batch"]), args.batch, max(steps, warm, 1), keep_traj=not args.no_traj, device=dev)
                rj.load_batch(bj["init_pos"], bj["init_v"], bj["batch"], bj["shape"])
                runners.append(rj)
            for r_ in runners:
                r_.run(max(warm, 5, min(clock_warm, r_.max_steps)), seed=21, use_graph=True)
            for r_ in runners:
                r_.synchronize()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for r_ in runners:
                r_.run(steps, seed=22, use_graph=True)
            for r_ in runners:
                r_.synchronize()
            dtc = time.perf_counter() - t1
            out["concurrent_chains"] = {"chains": len(runners), "value": round(len(runners) * args.batch / (CHAIN_STEPS * dtc / steps), 3),
                                        "unit": "molecules/s", "ms_per_step_all_chains": round(dtc / steps * 1e3, 4),
                                        "note": "independent batch-256 chains (own batch-norm statistics each) on separate streams of one GPU"}
            log(f"concurrent chains: {out['concurrent_chains']}")
            del runners
        # ---- optional modes, driver-timed like `value`, never `value`: two-piece f16 operands (22-23 bits, three products) and
        #      "f16 features" (single f16 pieces, 11 bits; BASELINE configs[2] names a reduced-precision feature mode) ----
        def timed_mode(options, restore, seed):
            es = min(args.exact_steps, runner.max_steps)
            for k_, v_ in options:
                model.set_option(k_, v_)
            try:
                runner.run(max(1, min(warm, es), min(clock_warm, runner.max_steps)), seed=seed, use_graph=True)   # re-captures the step graph with these kernels; loaded clocks
                runner.synchronize()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                runner.run(es, seed=seed + 1, use_graph=True)
                runner.synchronize()
                dte = (time.perf_counter() - t1) / es
            finally:
                for k_, v_ in restore:
                    model.set_option(k_, v_)
            return {"value": round(args.batch / (CHAIN_STEPS * dte), 3), "unit": "molecules/s", "ms_per_step": round(dte * 1e3, 4), "steps": es,
                    "options": [f"{k_}={v_}" for k_, v_ in options]}

        if world == 1 and args.exact_steps > 0 and use_graph and not args.opt:
            out["f16x2_mode"] = dict(timed_mode([("edge_bf16", 3), ("node_f16", 1)], [("node_f16", 0), ("edge_bf16", 2)], 31),
                                     dtype="f32 (matrix operands as 2 x f16 pieces: error <= 2^-22 |x| per operand, absolute floor 2^-25 for |x| < 2^-3; "
                                           "three products, fp32 accumulate)",
                                     note="the round-2/3 kernels; passes the same parity gates, NOT reference operand precision")
            log(f"f16x2 mode: {out['f16x2_mode']}")
            out["f16_features_mode"] = dict(timed_mode([("edge_bf16", 3), ("node_f16", 1), ("feat_f16", 1)], [("feat_f16", 0), ("node_f16", 0), ("edge_bf16", 2)], 41),
                                            dtype="f16 matrix operands (11 bits, one product per term), fp32 accumulate / LayerNorm / softmax / coordinates",
                                            note="reduced precision, outside the parity gates (forward error ~1e-3)")
            log(f"f16 features mode: {out['f16_features_mode']}")
        # trajectory D2H cost (reported, never part of value)
        if not args.no_traj:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            _ = [runner.bufs[k][:steps].cpu() for k in ("pos_traj", "v_traj", "v0_traj", "vt_traj")]
            out["traj_d2h_ms"] = round((time.perf_counter() - t1) * 1e3, 2)
        # ---- CPU baseline: the oracle on this host, bounded sample ---------------------------
        if world == 1 and args.cpu_steps > 0:
            log(f"CPU oracle baseline: {args.cpu_steps} steps")
            dt, cores = cpu_baseline(cfg, bb, args.cpu_steps, args.cpu_threads)
            out["cpu_baseline"] = {"value": round(args.batch / (CHAIN_STEPS * dt), 4), "unit": "molecules/s", "cores": cores,
                                   "os_cpu_count": os.cpu_count(), "cpu_affinity": len(os.sched_getaffinity(0)),
                                   "kind": "port", "sample": f"{args.cpu_steps} reverse steps of the same B={args.batch} batch "
                                   f"({dt:.3f} s/step, torch-CPU oracle), extrapolated to 1000 steps"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
