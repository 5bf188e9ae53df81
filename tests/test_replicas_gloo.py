"""CPU, world_size 2 over gloo: the replica protocol bench.py uses across GPUs — rank 0 broadcasts
the prompt, every replica decodes independently (no data-path collective), sampled ids are all-gathered and must
agree, timing is the MAX over ranks.  The GPU decode is replaced by a deterministic stand-in here; the collective
logic is the same code path (bench.replica_exchange)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    prompt = np.arange(100, 140, dtype=np.int32) if rank == 0 else np.zeros(40, dtype=np.int32)
    prompt = bench.broadcast_prompt(dist, prompt, rank, device="cpu")
    ids = (prompt[:16] * 7 + 3) % 1000          # stand-in for the replica's greedy decode (deterministic in the prompt)
    if rank == 1 and os.environ.get("PS_TEST_DIVERGE"):
        ids = ids.copy(); ids[5] += 1
    agree, all_ids = bench.gather_ids(dist, ids.astype(np.int32), world, device="cpu")
    tmax = bench.max_over_ranks(dist, [0.5 + rank, 2.0 - rank], device="cpu")
    q.put((rank, prompt.tolist(), agree, [a.tolist() for a in all_ids], tmax))
    dist.destroy_process_group()


def _run(diverge=False):
    if diverge:
        os.environ["PS_TEST_DIVERGE"] = "1"
    else:
        os.environ.pop("PS_TEST_DIVERGE", None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    return out


def test_broadcast_gather_max():
    out = _run()
    for rank, prompt, agree, all_ids, tmax in out:
        assert prompt == list(range(100, 140))           # rank 0's prompt reached every replica
        assert agree and all_ids[0] == all_ids[1]
        assert tmax == [1.5, 2.0]                        # MAX over ranks of each timing


def test_divergent_replica_is_detected():
    out = _run(diverge=True)
    assert all(not agree for _, _, agree, _, _ in out)


def test_self_launch_command_and_world_check(tmp_path):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with one rank per GPU
    on 127.0.0.1 (the driver's command line); started under a launcher whose world size disagrees with --gpus it refuses."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.distributed_command(4, 29511, ["--gpus", "4", "--steps", "3"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_reference_baseline_runs_under_a_time_limit(tmp_path):
    """bench.py's cpu_baseline leg calls the REAL reference's mat-muls through the reference's own ThreadPool, whose spin barrier can stall for good
    on a busy host: the leg runs in a child process that is killed at a time limit, and the bench line then carries the port as its baseline with
    the reason next to it -- the line itself never hangs."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    from oracle import binding
    from powerserve_amd import gguf, synth
    if not binding.have_ref():
        pytest.skip("oracle/_ref/libps_ref.so not built (needs /root/reference)")
    d = str(tmp_path / "m")
    synth.write_model_dir(d, "tiny-llama", gguf.NAME_TYPE["Q4_K"], n_ctx=64, seed=1)
    ok = bench.cpu_reference_guarded(d, 60)
    if "error" in ok:  # (the very stall the limit is for has been seen on a busy host: then this is the answer, within the limit)
        assert "did not finish" in ok["error"]
    else:
        assert ok["kind"] == "reference" and ok["value"] > 0 and ok["pool_size_sweep"][str(ok["cores"])]["median"] == ok["value"]  # (the best median of the swept pool sizes is the headline)
    late = bench.cpu_reference_guarded(d, 0.01)
    assert set(late) == {"error"} and "did not finish" in late["error"]
