"""Headline benchmark: env-steps/s (whole job) and PPO-iters/s of the on-device rollout + PPO update.

One "step" = one PPO iteration: T control steps for each of N envs (policy + critic inference and the
fused env step on device), GAE, then `epochs` passes of minibatch updates over the N*T samples --
exactly the per-iteration body of the reference's PPO.train (reference rl/algos/ppo.py:459-566).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA peak (= f32 vector peak)
MFMA_F16_PEAK_TFLOPS = 2500.0  # fp16 / bf16 MFMA dense peak (MI355X_MICROARCH.md: ~2.5 PF dense; 2:1 sparsity figures are not used)
FP64_VALU_PEAK_TFLOPS = 78.6


def committed_pmc(env_name, kernel_substr="humanoid_rollout"):
    """Counter digest of the RESIDENT rollout kernel from the rocprofv3 --pmc passes COMMITTED under profiles/ this round
    (profiles/r06_<env>_rollout_pmc.csv, written by scripts/gpu_pmc.sh traffic <env>: one dispatch = one rollout of N envs x T = 400
    control steps; FETCH_SIZE / WRITE_SIZE in their own passes).  Per env-step: executed fp64 FLOPs ((2 FMA + MUL + ADD)
    wave-instructions x active lanes per VALU instruction -- SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU, calibrated on fully active kernels:
    profiles/r05_pmc_lane_calibration.txt), VALU instructions, HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KB counters; the factor 2 is the
    gfx950 correction of MI355X_MICROARCH.md for coalesced reads).  NOT measured by this run: the counters need their own rocprofv3 runs."""
    import csv
    path = os.path.join(ROOT, "profiles", f"r06_{env_name}_rollout_pmc.csv")
    if not os.path.exists(path):
        return None
    n_envs = 8192 if env_name in ("h1", "h1_walk") else 4096      # what scripts/gpu_pmc.sh runs
    env_steps = n_envs * 400
    v = {}
    for row in csv.reader(open(path)):
        if len(row) >= 4 and kernel_substr in row[0]:
            try:
                v[row[1]] = float(row[3])
            except ValueError:
                pass
    need = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE")
    if not all(k in v for k in need):
        return None
    lanes = v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"]
    flops = (2 * v["SQ_INSTS_VALU_FMA_F64"] + v["SQ_INSTS_VALU_MUL_F64"] + v["SQ_INSTS_VALU_ADD_F64"]) * lanes
    envs_per_wave = 1 if env_name == "jvrc_step" else 2
    out = dict(executed_flops_per_env_step=flops / env_steps, active_lanes_per_valu_instruction=lanes,
               valu_instructions_per_env_substep=v["SQ_INSTS_VALU"] / env_steps / 25.0,
               hbm_bytes_per_env_step=(2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / env_steps,
               source=f"profiles/r06_{env_name}_rollout_pmc.csv ({env_name} @ {n_envs}, T = 400; per-dispatch means / {env_steps} env-steps; "
                      f"{envs_per_wave} env(s) per wavefront: a wave's instruction counts once per env it serves)")
    if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_VALU" in v and "SQ_WAIT_ANY" in v:
        out["wave_cycles_issuing_valu"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
        out["wave_cycles_waiting"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    return out


def update_flops_per_sample_epoch(D, H, A, mirror):
    """Multiply-add FLOPs the update performs per sample and epoch: actor forward + backward on the row (and on its mirrored twin),
    critic forward + backward.  forward = 2 (D H + H H + H O); backward = the weight gradients (the same count) + the activation
    gradients of the two upper layers.  No old-policy forward: the behaviour log-probabilities are stored by the rollout."""
    def net(O):
        fwd = 2 * (D * H + H * H + H * O)
        return fwd + fwd + 2 * (H * H + H * O)
    return (2 if mirror else 1) * net(A) + net(1)


def cpu_baseline_worker(a):
    """Oracle env + batch-1 torch actor/critic forward per step on one host core (the reference worker's
    per-step work, rl/workers/rollout_worker.py:142-146).  Returns env-steps done and seconds."""
    env_name, steps, seed = a
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from oracle import make_oracle_env
    env, obs_dim, act_dim = make_oracle_env(env_name, seed)
    obs = env.reset()
    W = [torch.randn(256, obs_dim) * 0.1, torch.zeros(256), torch.randn(256, 256) * 0.05, torch.zeros(256)]
    Wa, Wc = torch.randn(act_dim, 256) * 0.01, torch.randn(1, 256) * 0.05
    t0 = time.time()
    with torch.no_grad():
        for i in range(steps):
            x = torch.as_tensor(obs, dtype=torch.float32)
            h = torch.relu(torch.relu(x @ W[0].T + W[1]) @ W[2].T + W[3])
            a = (h @ Wa.T + 0.223 * torch.randn(act_dim)).numpy()
            _v = h @ Wc.T
            obs, r, flags, _, _ = env.step_auto(a)
    return steps, time.time() - t0


def run_cpu_baseline(env_name, target_seconds=15.0):
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 16)
    probe, dt = cpu_baseline_worker((env_name, 50, 0))
    per_step = dt / probe
    steps = max(50, int(target_seconds / per_step))
    ctx = mp.get_context("spawn")
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        res = pool.map(cpu_baseline_worker, [(env_name, steps, 100 + i) for i in range(cores)])
    wall = time.time() - t0
    total = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return dict(value=total / busy, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} processes x {steps} control steps of the float64 CPU oracle ({env_name}, stand-in for the "
                       f"reference's Ray workers: oracle physics + numpy task logic + batch-1 torch actor/critic forward); "
                       f"sampling only, no PPO update; {busy:.1f}s busy / {wall:.1f}s wall")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--env", default=os.environ.get("LHW_BENCH_ENV", "auto"))
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--traj-len", type=int, default=400, help="control steps per env per iteration (max_traj_len)")
    ap.add_argument("--minibatch-size", type=int, default=32768, help="per-GPU minibatch rows")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mirror", action="store_true")
    ap.add_argument("--infer-fp16", action="store_true", help="rollout inference with fp16 operands; update stays f32")
    ap.add_argument("--fp16", action="store_true", help="BASELINE config 5: fp16 actor / critic (inference and every GEMM of the update with fp16 operands, f32 accumulation / master weights / Adam)")
    ap.add_argument("--task-hook", choices=["none", "walking", "walking-own-done"], default="none",
                    help="measurement of the BaseTask seam (task_hook.py; walking envs): the reference's WalkingTask as a PLUG-IN instead of "
                         "the fused task -- 'walking': reward-only (resident rollout + one batched evaluation), 'walking-own-done': the task "
                         "also decides terminations (launch-per-step, host-side resets)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as a plain command: become N ranks (rank 0 prints the one JSON line)
        from learninghumanoidwalking_amd.dist_utils import relaunch_under_torchrun
        raise SystemExit(relaunch_under_torchrun(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # LHW_SHARE_GPU=1 (tests only): put every rank on GPU 0 and use gloo, to exercise the N>1 host path on a 1-GPU box
    share = os.environ.get("LHW_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # LHW_FORCE_DIST=1 (tests only): initialise the process group even for one rank, so that the RCCL branch below -- otherwise
    # reached only on a multi-GPU node -- executes on a 1-GPU box
    use_dist = world > 1 or os.environ.get("LHW_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from learninghumanoidwalking_amd import envs as lenvs
    from learninghumanoidwalking_amd.ppo import PPO
    env_name = args.env
    if env_name == "auto":
        env_name = "jvrc_walk" if hasattr(lenvs, "JvrcWalkSpec") else "cartpole"
    spec_cls = lenvs.ENVIRONMENTS[env_name]
    ppo_args = SimpleNamespace(
        gamma=0.99, lam=0.95, lr=3e-4, eps=1e-5, entropy_coeff=0.0, clip=0.2, minibatch_size=args.minibatch_size,
        epochs=args.epochs, max_traj_len=args.traj_len, num_procs=args.num_envs, num_envs=args.num_envs, max_grad_norm=0.5,
        mirror_coeff=0.4, eval_freq=10**9, recurrent=False, imitate=None, imitate_coeff=0.3, learn_std=False,
        std_dev=0.223, no_mirror=args.no_mirror, infer_fp16=args.infer_fp16, fp16=args.fp16, continued=None, logdir=os.path.join("/tmp", f"lhw_bench_{os.getpid()}"),
        device_index=local_rank)
    hook = None
    if args.task_hook != "none":
        from learninghumanoidwalking_amd.task_hook import VectorWalkingTask
        own = args.task_hook == "walking-own-done"
        hook = (lambda spec, dev: VectorWalkingTask(spec, dev, height_limits=(0.6, 1.4000001))) if own else (lambda spec, dev: VectorWalkingTask(spec, dev))
    algo = PPO(spec_cls, ppo_args, seed=0, task=hook)
    if algo.obs_rms is not None:  # cartpole path: frozen running normalisation after a short warm-up (ppo.py:442-457)
        b = algo.sample_parallel_with_workers()
        algo.obs_rms.update(b.states.cpu().numpy())
        algo.kernels.set_obs_norm(algo.obs_rms.mean, algo.obs_rms.std)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # per-launch timing of the env-step kernel with HIP events on the launch stream (rank 0 only)
    env = algo.env
    step_events = []
    orig_step = env.step
    timing = {"on": False}

    def timed_step(*a, **k):
        if not timing["on"]:
            return orig_step(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_step(*a, **k)
        e1.record()
        step_events.append((e0, e1))
        return r

    env.step = timed_step
    launch_envs = {"n": args.num_envs}
    if hasattr(env, "step_range"):
        orig_range = env.step_range

        def timed_range(first, count, *a, **k):
            if not timing["on"]:
                return orig_range(first, count, *a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                      # on the group's stream (the current stream inside Rollout.collect)
            r = orig_range(first, count, *a, **k)
            e1.record()
            step_events.append((e0, e1))
            launch_envs["n"] = count
            return r

        env.step_range = timed_range
    rollout_events = []
    if hasattr(env, "rollout"):
        orig_rollout = env.rollout

        def timed_rollout(*a, **k):
            if not timing["on"]:
                return orig_rollout(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_rollout(*a, **k)
            e1.record()
            if r:
                rollout_events.append((e0, e1))
            return r

        env.rollout = timed_rollout
    for i in range(args.warmup):
        algo.iterate(i)
    barrier()
    if hasattr(env, "pop_fault_stats"):
        env.pop_fault_stats()
    if hasattr(env, "pop_rerun_count"):
        env.pop_rerun_count()
    timing["on"] = rank == 0
    from learninghumanoidwalking_amd import dist_utils
    if use_dist and rank == 0:
        dist_utils.allreduce_events = []
    t0 = time.time()
    sample_t = opt_t = 0.0
    iter_s = []
    for i in range(args.steps):
        _, st, ot = algo.iterate(args.warmup + i)
        sample_t += st
        opt_t += ot
        iter_s.append(st + ot)
    barrier()
    elapsed = time.time() - t0
    faults = env.pop_fault_stats() if hasattr(env, "pop_fault_stats") else (0, 0)
    reruns = env.pop_rerun_count() if hasattr(env, "pop_rerun_count") else 0
    # calibration of the dominant kernel outside the timed region (rank 0): whole-batch launches, one at a time, nothing
    # else on the GPU -- the duration rocprofv3 --kernel-trace reports for an isolated launch
    isolated_ms = None
    timing["on"] = False
    if rank == 0 and hasattr(env, "step_range"):
        ro = algo.rollout
        evs = []
        for t in range(min(20, ro.T)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_step(ro.act[t])
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        isolated_ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax[0])

    if rank == 0:
        N, T, K = args.num_envs, args.traj_len, args.steps
        total_env_steps = N * T * K * world
        value = total_env_steps / elapsed
        step_ms = [a.elapsed_time(b) for a, b in step_events]
        resident_ms = [a.elapsed_time(b) for a, b in rollout_events]
        resident = bool(resident_ms)
        # resident rollout: ONE launch advances all N envs by T control steps; its span / T is the per-control-step figure the
        # launch-per-step pipeline reports per launch
        avg_step_ms = float(np.mean(resident_ms)) / T if resident else (float(np.mean(step_ms)) if step_ms else float("nan"))
        spec = algo.spec
        bytes_per_env_step = spec.algorithmic_bytes_per_env_step()
        flops_per_env_step = spec.algorithmic_flops_per_env_step()
        NL = N if resident else launch_envs["n"]   # envs per launch (N / rollout groups)
        groups = max(1, N // NL)
        achieved_gbs = bytes_per_env_step * NL / (avg_step_ms * 1e-3) / 1e9
        overlapped_tf = flops_per_env_step * NL / (avg_step_ms * 1e-3) / 1e12
        wall_step_ms = sample_t / K / T * 1e3     # wall time per control step of all N envs, policy inference included
        # The fused control-step code touches each env's state once per control step (3 KB): by design it is not HBM-bound
        #  but bound by fp64 vector issue + on-chip latency, so the primary roofline is the fp64 VALU one.
        # `achieved` = SURVEY.md 8(d)'s ALGORITHMIC FLOPs of one launch / that launch's duration, measured live with HIP events.
        iso_tf = None if isolated_ms is None else flops_per_env_step * N / (isolated_ms * 1e-3) / 1e12
        mode = getattr(algo.rollout, "last_mode", "steps")
        roofline = dict(
            bound="valu_fp64", kernel=spec.step_kernel_name, achieved=iso_tf if iso_tf is not None else overlapped_tf,
            peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s",
            frac=(iso_tf if iso_tf is not None else overlapped_tf) / FP64_VALU_PEAK_TFLOPS, traffic=None,
            traffic_note="HBM bytes are not measured by this run (PMC counters need separate rocprofv3 passes)",
            step_kernel_isolated=dict(kernel=spec.step_kernel_name, avg_launch_ms=isolated_ms, envs_per_launch=N,
                                      algorithmic_fp64_tflops=iso_tf, algorithmic_fp64_frac=None if iso_tf is None else iso_tf / FP64_VALU_PEAK_TFLOPS,
                                      note="one control step of the whole batch as ONE launch of the launch-per-step kernel, median of 20 issued one at a "
                                           "time after the timed region (HIP events on the launch stream)"),
            algorithmic_flops_per_launch=flops_per_env_step * N, algorithmic_bytes_per_launch=bytes_per_env_step * N,
            algorithmic_flops_per_env_step=flops_per_env_step, algorithmic_bytes_per_env_step=bytes_per_env_step,
            avg_launch_ms=isolated_ms if isolated_ms is not None else avg_step_ms, envs_per_launch=N if isolated_ms is not None else NL,
            launch_note="median of 20 whole-batch control-step launches issued one at a time after the timed region, HIP events on the "
                        "launch stream (no overlap): the duration rocprofv3 --kernel-trace reports for an isolated dispatch",
            rollout_mode=mode,
            overlapped=(None if resident else dict(avg_launch_ms=avg_step_ms, launches=len(step_ms), envs_per_launch=NL, concurrent_launches=groups,
                            fp64_tflops=overlapped_tf, fp64_frac=overlapped_tf / FP64_VALU_PEAK_TFLOPS,
                            note="HIP events on the launch stream around lhw_env_step_range over the timed region: the two-envs-per-wave "
                                 "kernel plus the (normally empty) one-env-per-wave re-run launch behind it; with concurrent_launches > 1 "
                                 "the groups' kernels overlap, so a launch's span includes the share of the GPU it cedes to the other group")),
            aggregate=dict(wall_ms_per_control_step=wall_step_ms,
                           fp64_tflops=flops_per_env_step * N / (wall_step_ms * 1e-3) / 1e12,
                           fp64_frac=flops_per_env_step * N / (wall_step_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                           note="whole batch's algorithmic FLOPs / wall time per control step of the rollout (policy inference included)"),
            hbm=dict(bound="hbm", achieved=achieved_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved_gbs / HBM_PEAK_GBS,
                     note="secondary: algorithmic bytes / launch span; ~3e-4 of peak by construction"))
        if resident:
            # The dominant kernel of the resident mode is the rollout kernel itself: ONE launch = T control steps of all N envs,
            # policy steps included.  achieved / frac: SURVEY.md 8(d)'s 75 kFLOP per env-sub-step x 25 x N x T over the launch's
            # duration (HIP events on its stream, this run).  What the hardware EXECUTED -- more, because every lane in EXEC counts --
            # and the HBM traffic come from the committed counter passes of this round and are labelled as such.
            launch_ms = float(np.mean(resident_ms))
            alg_tf = flops_per_env_step * N * T / (launch_ms * 1e-3) / 1e12
            queued = bool(env.last_rollout_queued()) if hasattr(env, "last_rollout_queued") else False
            roofline["kernel"] = spec.step_kernel_name.replace("humanoid_kernel<0, ", "humanoid_rollout_kernel<").replace(">", ", true>" if queued else ", false>")
            roofline["achieved"], roofline["frac"] = alg_tf, alg_tf / FP64_VALU_PEAK_TFLOPS
            roofline["achieved_note"] = "ALGORITHMIC fp64 FLOPs of one launch (SURVEY.md 8(d): 75 kFLOP per env-sub-step x 25 sub-steps x N x T) / its measured duration"
            roofline["avg_launch_ms"] = launch_ms
            roofline["launches_timed"] = len(resident_ms)
            roofline["envs_per_launch"] = N
            roofline["control_steps_per_launch"] = T
            roofline["launch_note"] = ("mean span of lhw_env_rollout (one launch = T control steps of all N envs, policy steps included) over the timed "
                                       "region, HIP events on its stream: the duration rocprofv3 --kernel-trace reports for humanoid_rollout_kernel "
                                       "(profiles/r06_jvrc_walk_kernel_stats.csv)")
            roofline["algorithmic_flops_per_launch"] = flops_per_env_step * N * T
            roofline["algorithmic_bytes_per_launch"] = bytes_per_env_step * N * T
            pmc = committed_pmc(env_name)
            if pmc is not None:
                ex_tf = pmc["executed_flops_per_env_step"] * N * T / (launch_ms * 1e-3) / 1e12
                roofline["executed"] = dict(tflops=ex_tf, frac_of_fp64_peak=ex_tf / FP64_VALU_PEAK_TFLOPS, **pmc,
                                            note="fp64 FLOPs the hardware executed per env-step by the COMMITTED SQ counter pass x this run's N x T, over this "
                                                 "run's launch duration; counts every lane in EXEC, useful or not -- context for `frac`, not a replacement")
                roofline["traffic"] = pmc["hbm_bytes_per_env_step"] * N * T
                roofline["traffic_note"] = ("HBM bytes per launch from the COMMITTED FETCH_SIZE / WRITE_SIZE passes of this round (2 x FETCH + WRITE per env-step "
                                            "x this run's N x T), not collected by this run; compare algorithmic_bytes_per_launch")
        L = algo.last_losses
        kk = algo.kernels
        use_mirror = bool(getattr(kk, "use_mirror", False))
        n_upd = int(L.get("n_updates") or 0)
        upd_flops = update_flops_per_sample_epoch(kk.obs_dim, kk.hidden, kk.act_dim, use_mirror) * float(n_upd * min(args.minibatch_size, N * T)) if not getattr(kk, "recurrent", False) else None
        if upd_flops:
            upd_tf = upd_flops / (opt_t / K) / 1e12
            upd_peak = MFMA_F16_PEAK_TFLOPS if args.fp16 else MFMA_F32_PEAK_TFLOPS
            roofline["update"] = dict(bound="mfma", achieved=upd_tf, peak=upd_peak, unit="TFLOP/s", frac=upd_tf / upd_peak,
                                      flops_per_iteration=upd_flops, flops_per_sample_epoch=update_flops_per_sample_epoch(kk.obs_dim, kk.hidden, kk.act_dim, use_mirror),
                                      seconds_per_iteration=opt_t / K,
                                      note="whole update phase (gather, forward, loss, backward, ordered reductions, clip + Adam of every optimiser step) "
                                           "over its wall time; " + ("fp16 MFMA dense peak (v_mfma_f32_32x32x16_f16; activations stored as fp16 in HBM, float32 master weights): "
                                                                     "at this size the phase is bound by HBM streaming and launch latency of its small GEMMs, not by the matrix cores; "
                                                                     f"against the f32-MFMA peak it would read {upd_tf / MFMA_F32_PEAK_TFLOPS:.3f}" if args.fp16 else "f32-input MFMA peak"))
        out = dict(
            metric="env-steps/s (whole job): on-device rollout + GAE + PPO update", value=value, unit="env-steps/s",
            n_gpus=world, steps=K, warmup=args.warmup, ms_per_step=elapsed / K * 1e3, higher_is_better=True, scaling="weak",
            vs_baseline=None, dtype="f64 physics / " + ("fp16-operand networks, f32 accumulate + master weights" if args.fp16 else "f32 networks" + (" (fp16-operand rollout inference)" if args.infer_fp16 else "")), data="synthetic",
            config=dict(workload=f"{env_name} @ {N} envs/GPU, T={T} control steps/iter, {args.epochs} epochs, "
                                 f"minibatch {args.minibatch_size}/GPU" + (" (JVRC stand-in model)" if env_name.startswith("jvrc") else " (H1 stand-in model)" if env_name.startswith("h1") else "")
                                 + (", flat ground (the reference's uneven-terrain hook is dead code: tasks/walking_task.py:173-179)" if env_name.startswith("h1") else "")
                                 + (f", task plug-in: {args.task_hook}" if args.task_hook != "none" else ""),
                        envs_per_gpu=N, traj_len=T, epochs=args.epochs, minibatch_per_gpu=args.minibatch_size,
                        mirror=not args.no_mirror and spec.mirror_tables() is not None,
                        frame_skip=spec.frame_skip, sim_dt=spec.sim_dt, control_dt=spec.control_dt),
            ppo_iters_per_s=K / elapsed, sample_s_per_iter=sample_t / K, optimize_s_per_iter=opt_t / K,
            median_iter_s=float(np.median(iter_s)), iter_s=[round(x, 4) for x in iter_s],
            optimizer_steps_per_iter=L.get("n_updates"), roofline=roofline,
            stepper_counters=dict(contact_overflow_steps=int(faults[0]), diverged_env_steps=int(faults[1]), one_env_per_wave_reruns=int(reruns),
                                  env_steps=int(N * T * K),
                                  note="rank 0, timed iterations: control steps that dropped contacts beyond the 16-contact layout / whose state "
                                       "went non-finite / that the two-envs-per-wave kernel handed to the one-env-per-wave kernel (> 8 contacts)"))
        if use_dist:
            ar = [a.elapsed_time(b) for a, b in (dist_utils.allreduce_events or [])]
            out["data_parallel"] = dict(n_gpus=world, envs_per_rank=[N] * world, backend="gloo (LHW_SHARE_GPU test mode)" if share else "nccl (RCCL)",
                                        gradient_floats=int(kk.grad.numel()) if hasattr(kk, "grad") else None,
                                        allreduce_calls_per_iter=len(ar) / K if K else 0, allreduce_ms_per_step=float(np.mean(ar)) if ar else None,
                                        allreduce_ms_per_iter=float(np.sum(ar)) / K if ar else None,
                                        note="rank 0, timed iterations: one sum-all-reduce of the flat gradient per optimiser step (events on the update's stream, "
                                             "so a call's span includes waiting for the slowest rank's minibatch); envs are sharded by global env id, no other data-path collective")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = run_cpu_baseline(env_name)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
