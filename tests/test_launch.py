"""hessgpu_amd/launch.py on the CPU: the ranks of a one-node job as child processes of a parent that never touches
the GPU -- rendezvous environment, rank 0's line relayed, worst exit code, the whole job ended when a rank dies or
the time is up (reference pattern: TestWin/MultiThreadSIFT.cpp:231-244, ServerSiftGPU.cpp:156-194)."""
import io
import json
import os
import subprocess
import sys
import time

import pytest

from hessgpu_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text("import os, sys, time, json\nrank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])\n" + body)
    return [sys.executable, str(p)]


def test_ranks_get_the_rendezvous_environment_and_rank0_is_relayed(tmp_path):
    cmd = _script(tmp_path, "print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}), flush=True)\n")
    out, err = io.StringIO(), open(tmp_path / "err.txt", "w+")
    assert launch.run_ranks(cmd, 3, timeout_s=60, out=out, err=err) == 0
    lines = out.getvalue().splitlines()
    assert len(lines) == 1                      # rank 0's stdout only
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and int(d["MASTER_PORT"]) > 0
    err.seek(0)
    others = sorted(json.loads(l)["RANK"] for l in err.read().splitlines() if l.startswith("{"))
    assert others == ["1", "2"]                 # the other ranks' stdout went to stderr


def test_two_ranks_rendezvous_over_gloo(tmp_path):
    cmd = _script(tmp_path, "import torch, torch.distributed as dist\n"
                            "dist.init_process_group('gloo')\n"
                            "t = torch.tensor([rank + 1]); dist.all_reduce(t)\n"
                            "if rank == 0: print(int(t), flush=True)\n"
                            "dist.destroy_process_group()\n")
    out = io.StringIO()
    assert launch.run_ranks(cmd, 2, timeout_s=120, out=out) == 0
    assert out.getvalue().strip().splitlines()[-1] == "3"    # (gloo prints a banner on stdout first: bench.py moves its stdout away for that reason)


def test_a_rank_that_dies_ends_the_job_with_its_code(tmp_path):
    # rank 1 leaves with 17 while rank 0 waits for it "in a collective" (here: sleeps for an hour)
    cmd = _script(tmp_path, "if rank == 1:\n    time.sleep(0.3); os._exit(17)\ntime.sleep(3600)\n")
    t0 = time.monotonic()
    rc = launch.run_ranks(cmd, 2, timeout_s=120, grace_s=0.5, kill_after_s=1.0, out=io.StringIO())
    assert rc == 17 and time.monotonic() - t0 < 20


def test_a_rank_killed_by_a_signal_counts_as_128_plus_signal(tmp_path):
    cmd = _script(tmp_path, "import signal\nif rank == 0:\n    os.kill(os.getpid(), signal.SIGABRT)\ntime.sleep(3600)\n")
    assert launch.run_ranks(cmd, 2, timeout_s=120, grace_s=0.2, kill_after_s=1.0, out=io.StringIO()) == 128 + 6


def test_timeout_ends_every_rank_even_one_that_ignores_sigterm(tmp_path):
    pidfile = tmp_path / "pids"
    cmd = _script(tmp_path, "import signal\nsignal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
                            f"open(r'{pidfile}', 'a').write(str(os.getpid()) + '\\n')\ntime.sleep(3600)\n")
    t0 = time.monotonic()
    assert launch.run_ranks(cmd, 2, timeout_s=1.0, kill_after_s=0.5, out=io.StringIO()) == launch.EXIT_TIMEOUT
    assert time.monotonic() - t0 < 20
    for pid in pidfile.read_text().split():
        with pytest.raises(ProcessLookupError):
            os.kill(int(pid), 0)


def test_worst_exit_code_of_ranks_that_all_finish(tmp_path):
    cmd = _script(tmp_path, "sys.exit([0, 3, 2][rank])\n")
    assert launch.run_ranks(cmd, 3, timeout_s=60, out=io.StringIO()) == 3


def test_bench_without_a_launcher_starts_its_own_ranks_and_says_so_without_a_gpu():
    """`python3 bench.py --gpus 2` with no WORLD_SIZE: the parent spawns two ranks (it used to print "launch with
    torch.distributed.run" and exit 2).  Without a GPU both ranks end with bench.py's "no GPU" code 3, which the
    parent hands on -- and the parent itself never imports torch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""      # (also on a GPU box: this test is about the launch, not the device)
    env["ROCR_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--launch-timeout", "300"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "no GPU visible" in r.stderr and "launch with torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""
