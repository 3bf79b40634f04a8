"""bench.py's launcher pieces that need no GPU: a child that passes its deadline is killed together with its process group, the phase it wrote is reported."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_run_child_kills_a_hung_process_group_and_names_the_phase():
    import bench
    code = ("import os, sys, time, subprocess\n"
            "open(os.environ['NMFAMD_BENCH_PHASE_FILE'], 'w').write('waiting for a peer that never comes')\n"
            "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(300)'])\n"      # a grandchild: the whole group must go
            "time.sleep(300)\n")
    t0 = time.monotonic()
    rc, out, err, hung = bench.run_child([sys.executable, "-c", code], dict(os.environ), "the test child", 3.0)
    assert rc == 3 and hung == "waiting for a peer that never comes" and time.monotonic() - t0 < 30


def test_run_child_passes_output_and_exit_code_through():
    import bench
    rc, out, err, hung = bench.run_child([sys.executable, "-c", "import sys; print('{\"a\": 1}'); sys.exit(7)"], dict(os.environ), "the test child", 30.0)
    assert rc == 7 and hung is None and out.strip() == '{"a": 1}'


def test_gpus_n_refuses_without_devices():
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
    assert out.returncode == 2 and not out.stdout.strip() and "--gpus 2" in out.stderr


def _torchrun(extra, timeout=240):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", NMFAMD_BENCH_DEADLINE="120", GLOO_SOCKET_IFNAME="lo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *extra]
    t0 = time.monotonic()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return out, time.monotonic() - t0


def test_ranks_under_torchrun_without_devices_agree_and_leave_nonzero():
    """The driver's launch (one rank per GPU under torch.distributed.run) on a box with NO usable device: the control plane (gloo) comes up, rank 0's team
    child refuses, every rank hears about it and leaves non-zero -- no line, no hang.  (world_size 2, gloo, CPU only.)"""
    out, took = _torchrun([])
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert took < 200


def test_rccl_ranks_without_devices_leave_nonzero():
    out, took = _torchrun(["--transport", "rccl"])
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "HIP device" in out.stderr or "RCCL" in out.stderr or "librccl" in out.stderr
    assert took < 200
