"""CPU, world_size 2 on the gloo backend: the multi-GPU harness (moca_video_amd/dist.py) -- strided
prompt sharding (videocrafter_main.py:181), flat-bucketed parameter broadcast (C1), result gather (C2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from moca_video_amd import dist as md
    r, l, w = md.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)
    m = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.LayerNorm(53), torch.nn.Linear(53, 7, bias=False))
    m.register_buffer("sched", torch.randn(11))
    before = [p.clone() for p in m.parameters()]
    sent = md.broadcast_parameters(m, src=0, bucket_bytes=4096)        # small buckets: several messages
    flat = torch.cat([p.reshape(-1) for p in m.parameters()] + [m.sched])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    changed = rank == 0 or any(not torch.equal(a, b) for a, b in zip(before, m.parameters()))
    res = torch.full((2, 3), float(rank))
    outs = md.gather_results(res, dst=0)
    ok_gather = (outs is None) if rank != 0 else all(torch.equal(o, torch.full((2, 3), float(i))) for i, o in enumerate(outs))
    t = md.max_over_ranks(1.0 + rank, torch.device("cpu"))
    md.barrier()
    q.put((rank, same, changed, ok_gather, t, sent, md.shard_indices(10, rank, world)))
    dist.destroy_process_group()


def test_dist_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, changed, ok_gather, t, sent, shard in out:
        assert same and changed and ok_gather
        assert t == 2.0
        assert sent > 0
        assert shard == list(range(10))[rank::world]


def _worker_packed(rank, world, port, q):
    import functools
    import types
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from moca_video_amd import dist as md
    from moca_video_amd import ops
    md.init_from_env(backend="gloo")
    torch.manual_seed(7)
    m = torch.nn.Sequential(torch.nn.Linear(96, 200), torch.nn.LayerNorm(200), torch.nn.Linear(200, 64, bias=False))
    if rank != 0:                                     # placeholders: the receivers never see real fp32 masters
        with torch.no_grad():
            for p in m.parameters():
                p.fill_(float("nan"))
    pk = {id(m[0]): ops.pack_linear(m[0].weight.detach(), m[0].bias.detach(), device="cpu"),
          id(m[1]): (m[1].weight.detach().float().clone(), m[1].bias.detach().float().clone()),
          id(m[2]): ops.finish_lnfold(ops.pack_linear(m[2].weight.detach(), None, device="cpu")),
          "unused": ops.pack_linear(torch.randn(64, 64), None, device="cpu")}       # packed but read by no recorded launch
    m._packed = pk
    step = lambda *a, **k: None
    plan = types.SimpleNamespace(steps=[lambda: None,                                  # (an engine's pre() hook: no operands)
                                        functools.partial(step, None, pk[id(m[0])], None, M=5),
                                        functools.partial(step, None, None, pk[id(m[1])][0], pk[id(m[1])][1]),
                                        functools.partial(step, None, pk[id(m[2])], None, ln=(pk[id(m[1])][0], pk[id(m[1])][1], None, 1e-5)),
                                        functools.partial(step, None, pk[id(m[0])], None, M=9)])   # the same operand twice
    sent, tensors = md.broadcast_packed(m, [plan], src=0, bucket_bytes=20000)
    expect = (256 * 128 * 2 + 256 * 4) + 2 * 200 * 4 + (64 * 256 * 2 + 64 * 4)      # w0 (200 -> 256 rows, 96 -> 128 k) + b0, gamma, beta, w2 + its row sums
    flat = torch.cat([t.reshape(-1).double() for t in tensors])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered) and bool(torch.isfinite(flat).all())
    masters = sum(p.numel() for p in m.parameters())
    q.put((rank, sent, expect, len(tensors), same, masters, getattr(m, "_packed_only", False),
           bool(torch.isnan(pk["unused"].w).any())))
    dist.destroy_process_group()


def test_broadcast_packed_world2_gloo():
    """C1 as the GPU job runs it (dist.broadcast_packed): the operand set the recorded launches read -- fp16 packed weights, fp32
    biases / folded row sums / norm parameters, each once, in first-use order -- is what moves (bytes = the packed sizes, not the
    fp32 masters); the receivers' packed buffers equal rank 0's afterwards, their masters are gone and they refuse new plans"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_packed, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n_master = 96 * 200 + 200 + 2 * 200 + 200 * 64
    for rank, sent, expect, n_t, same, masters, packed_only, unused_nan in out:
        assert sent == expect and n_t == 6 and same
        assert masters == (n_master if rank == 0 else 0) and packed_only == (rank != 0)
        assert not unused_nan


def test_shard_indices_cover():
    from moca_video_amd.dist import shard_indices
    for n in (0, 1, 7, 64):
        for w in (1, 2, 8):
            got = sorted(i for r in range(w) for i in shard_indices(n, r, w))
            assert got == list(range(n))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (the driver's SCALE run uses that
    form) and print ONE JSON line from rank 0 with n_gpus = 2.  `--selftest-cpu` keeps the hot path out (no GPU here):
    rendezvous, C1 broadcast, barrier, max-over-ranks and C2 gather run on gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-cpu"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_gathered"] == [0, 1] and res["weights_equal_after_broadcast"]
    assert res["steps"] == 3
    mg = res["multi_gpu"]                          # C1 proof: ranks as the backend saw them, bytes moved, checksums equal and finite
    # C1 moves the PACKED set: two fp16 [64][64] matrices (the 8-row one zero padded to 64 rows) + two fp32 [64] biases
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "gloo" and mg["broadcast_bytes"] == 2 * (64 * 64 * 2 + 64 * 4)
    assert "packed" in mg["broadcast_what"]
    assert mg["param_checksums_equal"] and mg["broadcast_s"] > 0 and all(v == v for v in mg["param_checksum_rank0"])
    assert res["rows_covered"]                     # the 64 prompt rows of config[4], strided over the ranks, all accounted for


def test_bench_fails_when_the_broadcast_is_skipped():
    """negative test: only rank 0 materialises the weights (the others hold NaN), so a C1 that does not happen must end the job
    with a non-zero exit code and no JSON line -- promptly (the launcher ends the remaining ranks)"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MOCA_BENCH_STUB_BROADCAST"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--selftest-cpu"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "checksums differ" in r.stderr
    assert not any(l.lstrip().startswith("{") for l in r.stdout.splitlines())
    assert time.time() - t0 < 300


def test_launcher_ends_the_job_when_one_rank_dies():
    """a rank that dies during start-up must not leave the others waiting in the rendezvous: launch_ranks polls every child"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys, time\n"
            "sys.argv = ['bench.py']\n"
            "sys.path.insert(0, %r)\n"
            "import bench\n"
            "bench.__file__ = os.path.join(%r, 'tools', '_rank_stub.py')\n"
            "t0 = time.time(); rc = bench.launch_ranks(2, [], timeout_s=120); print('rc', rc, 'dt', round(time.time() - t0, 1))\n") % (root, root)
    stub = os.path.join(root, "tools", "_rank_stub.py")
    with open(stub, "w") as f:
        f.write("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(600)\n")
    try:
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert "rc 3" in r.stdout, (r.stdout, r.stderr[-500:])
        assert time.time() - t0 < 60
    finally:
        os.remove(stub)


def test_bench_under_torchrun_env_does_not_relaunch():
    """under torch.distributed.run (WORLD_SIZE set) bench.py must NOT spawn again: each rank is already a process"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--selftest-cpu"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    assert json.loads(outs[0][0].strip().splitlines()[-1])["n_gpus"] == 2
    assert not any(l.lstrip().startswith("{") for l in outs[1][0].splitlines())      # only rank 0 prints the JSON line


def test_packed_operands_skip_weights_written_by_the_forward():
    """per-statistics-group weights of a folded GroupNorm (ops.groupnorm_fold_weights -> PackedWeight.derived) are OUTPUTS of a launch of
    the forward: they must not enter the C1 broadcast set (uninitialised pool memory: NaN checksums on every rank -- caught by the
    2-rank rehearsal of round 6), while the weights they are derived from must"""
    import functools
    from moca_video_amd import dist as mdist
    from moca_video_amd.ops import PackedWeight
    w, b = torch.zeros(64, 64, dtype=torch.float16), torch.zeros(64)
    src = PackedWeight(w, b, 64, 64, 64)
    der = PackedWeight(torch.full((128, 64), float("nan"), dtype=torch.float16), torch.full((128,), float("nan")), 64, 64, 64)
    der.derived = True
    gamma, beta = torch.ones(64), torch.zeros(64)

    class Plan:
        steps = [functools.partial(lambda *a, **k: None, src, gamma, beta, der.w, der.bias, n_sg=2),
                 functools.partial(lambda *a, **k: None, torch.zeros(4, 64), der, wgroup=(2, 4096))]
    ops_ = mdist.packed_operands({1: (gamma, beta)}, [Plan])
    ptrs = {t.data_ptr() for t in ops_}
    assert w.data_ptr() in ptrs and b.data_ptr() in ptrs and gamma.data_ptr() in ptrs and beta.data_ptr() in ptrs
    assert der.w.data_ptr() not in ptrs and der.bias.data_ptr() not in ptrs
    assert all(torch.isfinite(t.float()).all() for t in ops_)
