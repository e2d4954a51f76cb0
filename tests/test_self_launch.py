"""bench.py / run_experiment.py fan themselves out: `--gpus N` with N > 1 and no launcher environment re-executes the command as
N ranks under torch.distributed.run (dist_utils.relaunch_under_torchrun).  CPU check of the command line that is built; the
end-to-end run is tests/test_entry_gpu.py::test_bench_gpus_2_as_a_plain_command_launches_its_own_ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_relaunch_builds_a_one_node_torchrun_command(monkeypatch):
    from learninghumanoidwalking_amd import dist_utils
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    rc = dist_utils.relaunch_under_torchrun(4, "/x/bench.py", ["--gpus", "4", "--steps", "2"])
    cmd = seen["cmd"]
    assert rc == 7
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"      # the launcher picks the rendezvous port itself
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_relaunches_only_outside_a_launcher(monkeypatch):
    """inside a torch.distributed.run worker (WORLD_SIZE set) bench.py must NOT fan out again"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.gpus > 1 and "WORLD_SIZE" not in os.environ' in src
    src = open(os.path.join(ROOT, "run_experiment.py")).read()
    assert 'args.gpus > 1 and "WORLD_SIZE" not in os.environ' in src
