"""The single-node launcher (`dfd-clip_amd/launch.py`, the path's `accelerate launch`: reference
scripts/cross-manipulation-train.sh:6) with stub workers on CPU: every rank gets its own rendezvous variables, the
ranks can actually form a gloo group from them, a failing rank takes the job down with a non-zero exit code, and
`bench.py --gpus N` turns into that launcher before it imports anything that could touch the GPU."""
import json
import os
import subprocess
import sys
import textwrap
import time

from dfd_clip_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / "worker.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_ranks_get_their_environment(tmp_path):
    w = _script(tmp_path, """
        import json, os, sys
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]
        json.dump({k: os.environ.get(k) for k in keys}, open(os.path.join(sys.argv[1], "r%s.json" % os.environ["RANK"]), "w"))
        if os.environ["RANK"] == "0":
            print("LINE-FROM-RANK-0")
        else:
            print("noise from another rank")
    """)
    out = subprocess.run([sys.executable, "-c",
                          f"import sys; sys.path.insert(0, {ROOT!r}); from dfd_clip_amd import launch; "
                          f"sys.exit(launch.spawn_ranks(3, [{w!r}, {str(tmp_path)!r}]))"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "LINE-FROM-RANK-0"  # only rank 0 owns stdout: ONE result line
    envs = [json.load(open(tmp_path / f"r{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in envs} == {"0"}


def test_ranks_form_a_process_group(tmp_path):
    w = _script(tmp_path, """
        import os, sys, torch, torch.distributed as dist
        dist.init_process_group("gloo")
        one = torch.ones(1)
        dist.all_reduce(one)
        assert int(one.item()) == int(os.environ["WORLD_SIZE"]) == 2
        open(os.path.join(sys.argv[1], "seen%d" % dist.get_rank()), "w").write(str(int(one.item())))
        dist.destroy_process_group()
    """)
    assert launch.spawn_ranks(2, [w, str(tmp_path)]) == 0
    assert (tmp_path / "seen0").read_text() == "2" and (tmp_path / "seen1").read_text() == "2"


def test_a_failing_rank_ends_the_job(tmp_path):
    w = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)  # a rank blocked in a collective its peer will never join
    """)
    t0 = time.time()
    assert launch.spawn_ranks(2, [w], grace_s=5.0) == 7
    assert time.time() - t0 < 60


def test_under_launcher_detection():
    assert not launch.under_launcher({})
    assert launch.under_launcher({"WORLD_SIZE": "2", "RANK": "0"})
    env = launch.rank_env(1, 4, 1234, base={})
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "4" and env["MASTER_PORT"] == "1234"


def test_bench_gpus_flag_becomes_the_launcher(tmp_path):
    """`python bench.py --gpus 2` must start two ranks itself.  Here (no GPU) each rank stops at bench.py's own
    "needs a GPU" assertion — AFTER the launcher has given it WORLD_SIZE = 2; a recording sitecustomize shows the two
    child interpreters and their rendezvous variables, and the parent never imports torch."""
    (tmp_path / "sitecustomize.py").write_text(textwrap.dedent("""
        import os, sys
        if os.environ.get("RANK") is not None:
            open(os.path.join(os.environ["DFD_TEST_DIR"], "child%s" % os.environ["RANK"]), "w").write(os.environ["WORLD_SIZE"])
        else:
            import atexit
            atexit.register(lambda: open(os.path.join(os.environ["DFD_TEST_DIR"], "parent"), "w").write(str("torch" in sys.modules)))
    """))
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""), DFD_TEST_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0  # no GPU here: both ranks refuse to run
    assert "needs a GPU" in out.stderr
    assert (tmp_path / "child0").read_text() == "2" and (tmp_path / "child1").read_text() == "2"
    assert (tmp_path / "parent").read_text() == "False"
