"""world_size-2 gloo test of the multi-GPU plumbing: weight broadcast and page
sharding.  The data path has no collective, so this is all the distributed logic there is."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from retto_amd import dist as rdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blobs = [bytes([i]) * (1000 + 37 * i) for i in range(4)] if rank == 0 else None
    got = rdist.broadcast_blobs(blobs, 4, rank, device="cpu")
    sizes = [(960, 960)] * 5 + [(2000, 1400), (640, 640), (1088, 1920)]
    mine = rdist.shard_pages(sizes, world, rank)
    t = torch.tensor([len(mine)], dtype=torch.int64)
    dist.all_reduce(t)
    q.put((rank, rdist.digest(got), mine, int(t.item())))
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(30) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == res[1][1] == rdist.digest([bytes([i]) * (1000 + 37 * i) for i in range(4)])
    a, b = res[0][2], res[1][2]
    assert sorted(a + b) == list(range(8)) and not set(a) & set(b)
    assert res[0][3] == 8
    cost = lambda idx: sum([(960 * 960)] * 5 + [2000 * 1400, 640 * 640, 1088 * 1920][i - 5:i - 4] or [960 * 960] for i in idx)  # noqa
    sizes = [(960, 960)] * 5 + [(2000, 1400), (640, 640), (1088, 1920)]
    la = sum(sizes[i][0] * sizes[i][1] for i in a); lb = sum(sizes[i][0] * sizes[i][1] for i in b)
    assert abs(la - lb) / max(la, lb) < 0.25          # LPT keeps the two shards balanced


def test_sharding_uniform_is_even():
    for world in (1, 2, 4, 8):
        shards = [rdist.shard_pages([(960, 960)] * 32, world, r) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(32))
        assert all(len(s) == 32 // world for s in shards)


# ---- one global batch over the ranks: order-preserving gather, results independent of the world size ------------------
def _fake_process(sizes):
    """Stand-in for rt_run_batch on the CPU: a page's 'result' is a pure function of the page (its id and size), like the
    real pipeline's (a page's boxes / tokens do not depend on what else is in the batch -- checked on the GPU by bench.py)."""
    import hashlib

    def process(ids):
        return [hashlib.sha256(("%d:%dx%d" % (i, sizes[i][0], sizes[i][1])).encode()).hexdigest()[:12] for i in ids]
    return process


def _gb_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [[(640, 640), (960, 960), (720, 1280), (1080, 1920), (1754, 1240), (3508, 2480)][(7 * i) % 6] for i in range(37)]
    calls = []
    proc = _fake_process(sizes)

    def process(ids):
        calls.append(list(ids))
        return proc(ids)
    res = rdist.run_global_batch(sizes, rank, world, process, est_lines=[32] * len(sizes), chunk=8)
    q.put((rank, res, calls))
    dist.destroy_process_group()


def test_global_batch_gather_is_in_input_order_and_world_invariant():
    sizes = [[(640, 640), (960, 960), (720, 1280), (1080, 1920), (1754, 1240), (3508, 2480)][(7 * i) % 6] for i in range(37)]
    # world 1 (no process group needed: the gather is skipped)
    ref = rdist.run_global_batch(sizes, 0, 1, _fake_process(sizes), est_lines=[32] * len(sizes), chunk=8)
    assert ref == _fake_process(sizes)(list(range(37)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gb_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(30) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, got, calls in res:
        assert got == ref                                   # every rank holds all results, in input order, equal to world 1
        assert all(len(c) <= 8 for c in calls) and all(c == sorted(c) for c in calls)
    ids0 = [i for c in res[0][2] for i in c]; ids1 = [i for c in res[1][2] for i in c]
    assert sorted(ids0 + ids1) == list(range(37)) and not set(ids0) & set(ids1)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run: bench.py starts the two ranks itself (fresh child processes,
    RANK / WORLD_SIZE / MASTER_* in their environment), the weights go rank 0 -> rank 1 over the process group (gloo here, RCCL
    on the GPU box), a global page list is sharded and gathered in input order, and rank 0's ONE JSON line is relayed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                        env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] and j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["bcast_ms"] is not None
    dig = j["blob_digest"][:8]
    assert j["gathered"] == ["%d:%s" % (i, dig) for i in range(19)]   # every page once, input order, same weights on both ranks
    # one rank: same line, no process group
    pr1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run"], env=env, capture_output=True, text=True, timeout=600)
    assert pr1.returncode == 0, pr1.stderr[-2000:]
    j1 = json.loads([ln for ln in pr1.stdout.splitlines() if ln.startswith("{")][0])
    assert j1["gathered"] == j["gathered"] and j1["rccl_ranks"] == 1


def test_bench_world_8_dry_run():
    """The driver's 8-GPU launch shape on CPU: `bench.py --gpus 8 --backend gloo --dry-run` starts eight ranks, broadcasts the
    weights rank 0 -> 7, shards ONE page list over the eight ranks (LPT) and gathers it in input order."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run"],
                        env=env, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] and j["n_gpus"] == 8 and j["rccl_ranks"] == 8
    dig = j["blob_digest"][:8]
    assert j["gathered"] == ["%d:%s" % (i, dig) for i in range(19)]


def test_bench_launcher_ends_the_job_when_a_rank_dies():
    """launch_ranks() polls all ranks: a rank that exits non-zero before the rendezvous ends the job at once with ITS exit code
    (its peers would otherwise wait in init_process_group), instead of the launcher sitting on rank 0 until a timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["RT_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                        env=env, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 3, (pr.returncode, pr.stderr[-1000:])
    assert time.time() - t0 < 120
    assert not [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
