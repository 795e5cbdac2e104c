"""CPU tier: bench.py's launch logic -- which frames each rank owns (weak default, BASELINE
config 4's single batch of 64 sharded 8 per GPU), the command `--gpus N` starts by itself, and
that the parent decides all of this without importing torch (it must not touch a GPU before it
spawns the ranks).  Nothing here needs a device: `--dry-run` prints the plan and exits."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def dry(*args, env=None):
    out = subprocess.run([sys.executable, BENCH, "--dry-run", *args], capture_output=True, text=True,
                         env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_single_gpu_plan_is_the_plain_command():
    plan = dry()
    assert plan["gpus"] == 1 and plan["scaling"] == "weak"
    assert [(r["rank"], r["frames"]) for r in plan["ranks"]] == [(0, [0, 64])]
    # 64 frames per encode call fill the device: the read-once encoder
    assert plan["ranks"][0]["frames_per_call"] == 64 and "read-once" in plan["ranks"][0]["encoder"]
    # ... and the default call is the one that samples during that encoder's pass
    assert plan["ranks"][0]["one_pass"].startswith("strip walker")
    assert dry("--one-pass", "off")["ranks"][0]["one_pass"] is False
    assert dry("--source", "yuv420p")["ranks"][0]["one_pass"].startswith("strip walker")
    # (the strip walker's form switched off: the band writer's takes the call)
    assert dry("--opt", "fuse.walk=0")["ranks"][0]["one_pass"].startswith("band writer")
    assert dry("--opt", "fuse.walk=0", "--opt", "fuse.band=0")["ranks"][0]["one_pass"] == \
        "two calls inside the library"
    # one frame per call: the reference's pair unless the one-pass call is asked for
    assert dry("--frames-per-call", "1")["ranks"][0]["one_pass"] is False
    assert dry("--frames-per-call", "1", "--one-pass", "on")["ranks"][0]["one_pass"] == \
        "two calls inside the library"   # (a single frame is faster as the two calls)
    assert dry("--frames-per-call", "1", "--one-pass", "on", "--opt",
               "fuse.band=2")["ranks"][0]["one_pass"].startswith("band writer")
    assert plan["launch"][1:] == [BENCH]          # no launcher around N = 1


def test_global_batch_64_over_8_gpus_is_8_frames_each():
    plan = dry("--gpus", "8", "--global-batch", "64", "--steps", "5", "--warmup", "2")
    assert plan["scaling"] == "strong" and plan["frames_total"] == 64
    assert [r["frames"] for r in plan["ranks"]] == [[8 * k, 8 * k + 8] for k in range(8)]
    # 8 frames per rank = 240 strips: too few for the strip walker, the three kernels take them
    assert all(r["frames_per_call"] == 8 and r["encoder"].startswith("three kernels")
               and r["one_pass"].startswith("band writer") for r in plan["ranks"])
    cmd = plan["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    # the ranks get the user's arguments unchanged (and not --dry-run)
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--global-batch", "64", "--steps", "5", "--warmup", "2"]


def test_weak_default_gives_every_rank_its_own_batch():
    plan = dry("--gpus", "4", "--batch", "16")
    assert plan["scaling"] == "weak" and plan["frames_total"] == 64
    assert [r["frames"] for r in plan["ranks"]] == [[0, 16], [16, 32], [32, 48], [48, 64]]
    # uneven single batch: blocks differ by at most one frame and cover it exactly once
    plan = dry("--gpus", "3", "--global-batch", "64")
    sizes = [b - a for a, b in (r["frames"] for r in plan["ranks"])]
    assert sorted(sizes) == [21, 21, 22] and plan["ranks"][0]["frames"][0] == 0
    assert plan["ranks"][-1]["frames"][1] == 64


def test_rank_count_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True,
                         env=env, timeout=120)
    assert out.returncode == 2 and "must agree" in out.stderr
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--global-batch", "4"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "without a frame" in out.stderr


def test_parent_plans_without_importing_torch():
    code = ("import runpy, sys; sys.argv = ['bench.py', '--gpus', '2', '--dry-run']; "
            "runpy.run_path(%r, run_name='__main__'); "
            "assert 'torch' not in sys.modules, 'the launching parent imported torch'" % BENCH)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr


def test_send_frame_loop_plan_maps_client_c_to_gpu_c():
    """Config 5's 8-GPU form (client c <-> GPU c, src/video_server.cc:62-66), plan-checked without
    a GPU: examples/send_frame_loop_synth ... plan prints the device every client's thread opens."""
    ex = os.path.join(REPO, "examples")
    out = subprocess.run(["make", "-C", ex, "send_frame_loop_synth"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    exe = os.path.join(ex, "send_frame_loop_synth")

    def plan(*a):
        o = subprocess.run([exe, *a], capture_output=True, text=True, timeout=60)
        assert o.returncode == 0, o.stderr
        return json.loads(o.stdout)
    p = plan("8", "60", "120", "7680", "3840", "", "8", "rgb0", "rgb0", "plan")
    assert p["client_device"] == list(range(8)) and p["clients_per_device"] == [1] * 8
    assert p["reduced"] == [4272, 2144] and p["upload_bytes_per_frame"] == 7680 * 3840 * 4
    p = plan("8", "60", "120", "7680", "3840", "", "3", "yuv420p", "yuv420p", "plan")
    assert p["client_device"] == [0, 1, 2, 0, 1, 2, 0, 1] and p["clients_per_device"] == [3, 3, 2]
    assert p["upload_bytes_per_frame"] == 7680 * 3840 * 3 // 2
    o = subprocess.run([exe, "8", "60", "120", "7680", "3840", "", "0", "rgb0", "rgb0", "plan"],
                       capture_output=True, text=True, timeout=60)
    assert o.returncode != 0  # the plan needs an explicit GPU count
