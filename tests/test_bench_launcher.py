"""bench.py's N > 1 plumbing on CPU with the stub workload (no GPU, no model): (i) the built-in launcher -- `python
bench.py --gpus 2` with WORLD_SIZE unset starts 2 children, rendezvous over pipes, no process group; (ii) the driver's launch
-- torch.distributed.run, 2 ranks, gloo barrier + scalar exchange; (iii) a failing replica ends the run with a non-zero
exit; (iv) --gpus must agree with WORLD_SIZE."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LL_BENCH_CHILD")}
    env.update(kw)
    return env


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check(rec, n, steps):
    assert rec["n_gpus"] == n and rec["steps"] == steps and rec["scaling"] == "weak" and rec["unit"] == "frames/s"
    per = rec["config"]["per_replica_fps"]
    assert len(per) == n and rec["config"]["replicas"] == n
    # stub: replica r sleeps 0.05 (r + 1) s for 12 * steps frames; whole job = all frames / slowest replica
    total = 12 * steps * n
    assert abs(rec["value"] * rec["ms_per_step"] * steps / 1e3 - total) < 1e-6 * total
    assert rec["ms_per_step"] * steps >= 1e3 * 0.05 * n * 0.98
    assert rec["value"] <= total / (0.05 * n) * 1.02


def test_builtin_launcher_two_replicas():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "0", "--stub-workload"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    _check(rec, 2, 3)
    assert rec["config"]["per_replica_fps"][0] > rec["config"]["per_replica_fps"][1]       # replica 1 sleeps twice as long


def test_torchrun_two_ranks_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--stub-workload"]
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                                                     # rank 0 only
    _check(json.loads(lines[0]), 2, 2)


def test_world_size_must_match_gpus():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--stub-workload"], env=_env(WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_failing_replica_fails_the_run():
    # without the stub the children need a GPU: here they exit with "needs an MI355X" -> the launcher must fail, not hang
    env = _env(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c",
                          "import sys, torch; torch.cuda.device_count = lambda: 2; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0'];"
                          f"sys.path.insert(0, {ROOT!r}); import bench; bench.main()"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "exited" in out.stderr and "{\"metric\"" not in out.stdout


def test_launcher_translates_ranks_through_the_inherited_device_mask():
    """with an allotment like HIP_VISIBLE_DEVICES=2,3 the children must run on entries 2 and 3 of the physical
    numbering (the r-th entry of the inherited mask), not on the bare ranks 0 and 1; ROCR_VISIBLE_DEVICES is passed through and an
    inherited CUDA_VISIBLE_DEVICES cannot compose with the new value."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.child_visibility({"HIP_VISIBLE_DEVICES": "2,3"}, 1, 2) == {"HIP_VISIBLE_DEVICES": "3", "CUDA_VISIBLE_DEVICES": "3"}
    assert bench.child_visibility({"CUDA_VISIBLE_DEVICES": "5, 7"}, 0, 2)["HIP_VISIBLE_DEVICES"] == "5"
    assert bench.child_visibility({}, 3, 8)["HIP_VISIBLE_DEVICES"] == "3"
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-workload"],
                         env=_env(HIP_VISIBLE_DEVICES="2,3", CUDA_VISIBLE_DEVICES="0,1,2,3", ROCR_VISIBLE_DEVICES="0,1,2,3,4,5"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["config"]["per_replica_visible_devices"] == ["2", "3"]
    # one entry in the mask but two replicas asked for: refuse rather than run both on a device outside the allotment
    out = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, %r); import bench; print(bench.child_visibility({'HIP_VISIBLE_DEVICES': '4'}, 1, 2))" % ROOT],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "inherited device mask" in out.stderr
