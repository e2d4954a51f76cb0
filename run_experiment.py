"""Entry point with the reference's `train` command line (reference run_experiment.py:153-208), running the
on-device rollout + PPO of this repository.

    python run_experiment.py train --env jvrc_walk --logdir /tmp/logs --num-envs 4096 --n-itr 100 --seed 0
    python run_experiment.py train --env jvrc_walk --gpus 8 ...        (re-executes itself as 8 ranks under torch.distributed.run)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_experiment.py train --env jvrc_walk ...

Differences from the reference, by design: no Ray (`--num-procs` is the number of on-device envs per GPU unless
`--num-envs` is given); with `--recurrent` (LSTM actor / critic) `--minibatch-size` counts env columns = trajectories, as it
counts trajectories in the reference; the `eval` sub-command (GL viewer on CPU MuJoCo) is not part of the path built here;
`--imitate` needs an env description with `imitation_projector()` (as in the reference).
"""
import argparse
import os
import pickle
import shutil
import sys
from datetime import datetime
from functools import partial
from pathlib import Path


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--env", required=True, type=str)
    p.add_argument("--logdir", default=Path("/tmp/logs"), type=Path)
    p.add_argument("--input-norm-steps", type=int, default=100000)
    p.add_argument("--n-itr", type=int, default=20000)
    p.add_argument("--lr", type=float, default=3e-4)
    p.add_argument("--eps", type=float, default=1e-5)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--lam", type=float, default=0.95)
    p.add_argument("--std-dev", type=float, default=0.223)
    p.add_argument("--learn-std", action="store_true")
    p.add_argument("--entropy-coeff", type=float, default=0.0)
    p.add_argument("--clip", type=float, default=0.2)
    p.add_argument("--minibatch-size", type=int, default=64)
    p.add_argument("--epochs", type=int, default=3)
    p.add_argument("--num-procs", type=int, default=12)
    p.add_argument("--num-envs", type=int, default=None, help="on-device environments per GPU (default: --num-procs)")
    p.add_argument("--max-grad-norm", type=float, default=0.5)
    p.add_argument("--max-traj-len", type=int, default=400)
    p.add_argument("--no-mirror", action="store_true")
    p.add_argument("--mirror-coeff", default=0.4, type=float)
    p.add_argument("--eval-freq", default=100, type=int)
    p.add_argument("--continued", type=Path)
    p.add_argument("--recurrent", action="store_true")
    p.add_argument("--imitate", type=str, default=None)
    p.add_argument("--infer-fp16", action="store_true", help="rollout inference with fp16 operands (update stays float32)")
    p.add_argument("--fp16", action="store_true", help="fp16 actor / critic: inference and all GEMMs of the update with fp16 operands, float32 accumulation / master weights / Adam")
    p.add_argument("--imitate-coeff", type=float, default=0.3)
    p.add_argument("--yaml", type=str, default=None)
    p.add_argument("--device", type=str, default="auto", choices=["auto", "cpu", "cuda"])
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--gpus", type=int, default=1, help="GPUs of this node to train on (data parallel: envs sharded, gradients all-reduced over RCCL); "
                                                        "N > 1 outside a torch.distributed.run launcher re-executes this command as N ranks")
    return p


def run_experiment(args):
    import torch
    import torch.distributed as dist
    from learninghumanoidwalking_amd.envs import ENVIRONMENTS
    from learninghumanoidwalking_amd.ppo import PPO

    if args.device == "cpu" or not torch.cuda.is_available():
        raise SystemExit("this trainer runs on MI355X only: there is no CPU path (use the reference for --device cpu)")
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    share = os.environ.get("LHW_SHARE_GPU") == "1"     # (tests only: every rank on GPU 0 over gloo, to run the N > 1 path on a 1-GPU box)
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rank = dist.get_rank() if world > 1 else 0
    if args.env not in ENVIRONMENTS:
        raise SystemExit(f"unknown --env {args.env!r}; available: {sorted(ENVIRONMENTS)}")
    stamp = [datetime.now().strftime("%y-%m-%d-%H-%M-%S-%f")[:-3]]
    if world > 1:
        dist.broadcast_object_list(stamp, src=0)     # one log directory per run, named by rank 0's clock
    args.logdir = Path(args.logdir) / f"{stamp[0]}_{args.env}"
    args.device_index = local_rank
    Spec = ENVIRONMENTS[args.env]
    env_fn = partial(Spec, yaml_path=args.yaml) if (args.yaml and args.env != "cartpole") else Spec
    if rank == 0:
        Path.mkdir(args.logdir, parents=True, exist_ok=True)
        with open(Path(args.logdir, "experiment.pkl"), "wb") as f:   # run_experiment.py:136-139
            pickle.dump(args, f)
        if args.yaml:
            shutil.copyfile(args.yaml, Path(args.logdir, "config.yaml"))
    if args.seed is not None:
        torch.manual_seed(args.seed)
    algo = PPO(env_fn, args, seed=args.seed)
    algo.train(env_fn, args.n_itr)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("train", "eval"):
        raise SystemExit("usage: run_experiment.py train --env <name> [...]")
    if sys.argv[1] == "eval":
        raise SystemExit("`eval` (GL viewer / video on CPU MuJoCo) is outside the hot path of this repository; "
                         "use the reference's run_experiment.py eval")
    sys.argv.remove("train")
    args = build_parser().parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from learninghumanoidwalking_amd.dist_utils import relaunch_under_torchrun
        raise SystemExit(relaunch_under_torchrun(args.gpus, os.path.abspath(__file__), ["train"] + sys.argv[1:]))
    run_experiment(args)
