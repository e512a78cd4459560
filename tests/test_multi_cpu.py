"""world_size-2 gloo test of the replica-parallel bookkeeping used by bench.py --gpus N (no GPU needed)."""
import os
import socket
import subprocess
import sys
import textwrap

import helpers as H


def test_partition_is_disjoint_and_complete():
    m = H.pkg.multi if hasattr(H.pkg, "multi") else __import__("importlib").import_module("vins-mobile_amd.multi")
    for n, w in ((64, 8), (10, 3), (5, 8)):
        parts = [m.sequences_of_rank(n, r, w) for r in range(w)]
        flat = sorted(s for p in parts for s in p)
        assert flat == list(range(n))
        assert all(s % w == r for r, p in enumerate(parts) for s in p)
    assert m.seed_of_sequence(0) == 42


def test_two_rank_gloo_barrier_and_max(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import importlib, os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        m = importlib.import_module("vins-mobile_amd.multi")
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        mine = m.sequences_of_rank(64, rank, world)
        dist.barrier()
        dt = m.max_over_ranks(dist, 1.0 + rank)           # slowest rank defines the step time
        rate = m.aggregate_rate(dist, len(mine) * 10, 1.0 + rank)
        assert dt == float(world), dt
        assert abs(rate - 64 * 10 / world) < 1e-9, rate
        # result poses of each rank's first sequence, gathered on every rank (bench.py multi_gpu_proof)
        import numpy as np
        poses = m.all_gather_array(dist, np.arange(77, dtype=np.float64) + 1000.0 * mine[0])
        assert len(poses) == world
        for r, p in enumerate(poses):
            assert p.shape == (77,) and p[0] == 1000.0 * m.sequences_of_rank(64, r, world)[0] and p[76] == p[0] + 76
        # both batch sizes of one job (bench.py: headline and configs[3] literal), with every rank's own step time
        rep = m.replica_report(dist, 8, 20, 0.5 * (1 + rank))
        assert rep["n_gpus"] == world and rep["sequences_per_gpu"] == 8
        assert rep["per_rank_ms_per_step"] == [25.0 * (1 + r) for r in range(world)], rep
        assert abs(rep["ms_per_step"] - 25.0 * world) < 1e-9 and abs(rep["value"] - 8 * world * 20 / (0.5 * world)) < 1e-6, rep
        dist.destroy_process_group()
        print("ok", rank)
    """ % H.ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_two_rank_estimator_path_up_to_the_device_call(tmp_path):
    """Two CPU-side ranks of one node walk what a through-the-ABI multi-GPU run does before its first device call: the
    host pool of each process takes its SHARE of the node's CPUs (LOCAL_WORLD_SIZE; eight full-width pools in one cgroup
    bring the quota-throttling stalls back), the sequences are partitioned by id, the estimator object is created (host state only) and its
    device context refuses to start without a device (no CPU fallback) with the same status on both ranks, and the per-rank figures are gathered the way bench.py
    reports them in multi_gpu.ranks[*]."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # what ONE process alone would take on this machine
    base_env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_WORLD_SIZE", "VIO_AMD_HOST_THREADS")}
    alone = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import importlib; "
                            "abi = importlib.import_module('vins-mobile_amd').abi; print(abi.host_pool_width()[0])" % H.ROOT],
                           env=base_env, capture_output=True, text=True, timeout=300)
    assert alone.returncode == 0, alone.stderr
    width_alone = int(alone.stdout.strip().splitlines()[-1])
    script = tmp_path / "worker_est.py"
    script.write_text(textwrap.dedent("""
        import ctypes as C, importlib, os, sys
        import numpy as np
        sys.path.insert(0, %r)
        import torch.distributed as dist
        pkg = importlib.import_module("vins-mobile_amd")
        abi, m = pkg.abi, importlib.import_module("vins-mobile_amd.multi")
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        width, pools = abi.host_pool_width()
        assert width == max(1, %d // world), (width, %d, world)
        mine = m.sequences_of_rank(64, rank, world)
        assert len(mine) == 64 // world
        # the estimator of this rank's sequences: without a device the create call must refuse (VIO_ENODEV), not fall back
        lib = abi.load_product()
        cfg = abi.default_config()
        h = C.c_void_p()
        tic, ric = np.zeros(3), np.eye(3).ravel()
        rc = lib.vio_estimator_create(C.byref(cfg), len(mine), tic.ctypes.data_as(C.POINTER(C.c_double)),
                                      ric.ctypes.data_as(C.POINTER(C.c_double)), C.byref(h))
        assert rc == abi.VIO_OK and h.value, rc          # (a host object: its device contexts come with the first frame)
        hb = C.c_void_p()
        rcb = lib.vio_backend_create(C.byref(cfg), len(mine), C.byref(hb))
        assert rcb == abi.VIO_ENODEV and not hb.value, rcb  # the device call: refused without a device, on every rank alike
        lib.vio_estimator_destroy(h)
        widths = m.per_rank_values(dist, float(width))
        assert widths == [float(width)] * world, widths
        dist.destroy_process_group()
        print("ok", rank, width)
    """ % (H.ROOT, width_alone, width_alone)))
    env = dict(base_env, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)
