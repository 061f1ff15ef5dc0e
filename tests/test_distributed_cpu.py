"""The N > 1 path (window sharding + all-gather + fold) with world_size 2 over gloo on CPU.  The per-window totals come
from the oracle (stand-in for gmsm_window_sums_device, which needs a GPU); sharding, packing, the collective and the
product's host-side fold are the code bench.py runs on N GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, curve, which, c, n, q):
    try:
        import importlib
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        import torch
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gm = importlib.import_module("gnark-crypto_amd")
        sharding = importlib.import_module("gnark-crypto_amd.sharding")
        import oracle
        from conftest import random_scalars, rng_for
        g = (gm.G1Affine if which == "g1" else gm.G2Affine)(curve)
        o = oracle.Oracle(curve, which)
        pts = o.gen_points(n, 77, 3)
        sc = random_scalars(rng_for(31, n), g.curve, n)
        digits = o.partition_scalars(sc, c)
        lastc = c + 1 - (o.nb_chunks(c) * c - g.curve.fr_bits)

        def window_sums(c_, first, stride):
            return np.stack([o.process_chunk(max(c_, lastc), pts, digits[w]) for w in range(first, o.nb_chunks(c_), stride)])

        expected = o.msm_affine(pts, sc, c=c)
        jac = sharding.sharded_multiexp(g, window_sums, c, rank, world, sharding.torch_all_gather(dist, torch.device("cpu")))
        ok = bool((g.jac_to_affine(jac) == expected).all())

        # the path bench.py runs on N GPUs: totals written into the exchange buffer, one all_gather_into_tensor, fold;
        # both decompositions (here the oracle stands in for gmsm_window_sums_enqueue)
        for mode in ("windows", "points"):
            plan = sharding.shard_plan(g, n, rank, world, mode, c=c)
            ex = sharding.Exchange(dist, torch.device("cpu"), plan["rows"], g.xyzz_limbs)

            def enqueue(plan_, local):
                lo, hi = plan_["lo"], plan_["hi"]
                dg = o.partition_scalars(sc[lo:hi], plan_["c"])
                rows = [o.process_chunk(max(plan_["c"], lastc), pts[lo:hi], dg[w])
                        for w in range(plan_["win_first"], plan_["nwin"], plan_["win_stride"])]
                for i, row in enumerate(rows):
                    local[i] = torch.from_numpy(np.ascontiguousarray(row).view(np.int64))

            for _ in range(2):  # the buffers are reused call after call
                jac2 = sharding.sharded_multiexp_exchange(g, plan, enqueue, ex)
                ok = ok and bool((g.jac_to_affine(jac2) == expected).all())
        q.put((rank, ok, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))


def _batch_worker(rank, world, port, k, n, q):
    """Replica mode: k scalar vectors over the same bases, vector j on rank j % world, one all-gather of k results."""
    try:
        import importlib
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        import torch
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gm = importlib.import_module("gnark-crypto_amd")
        sharding = importlib.import_module("gnark-crypto_amd.sharding")
        import oracle
        from conftest import random_scalars, rng_for
        g = gm.G1Jac("bn254")
        o = oracle.Oracle("bn254", "g1")
        pts = o.gen_points(n, 5, 11)
        vectors = [random_scalars(rng_for(77, j), g.curve, n) for j in range(k)]
        computed = []

        def local_batch(mine):
            computed.extend(mine)
            return np.stack([o.multiexp(pts, vectors[j])[1] for j in mine])

        res = sharding.replicated_batch(k, rank, world, g.jac_limbs, local_batch,
                                        sharding.torch_all_gather(dist, torch.device("cpu")))
        ok = computed == list(range(rank, k, world))  # every rank only computed its own vectors
        for j in range(k):
            ok = ok and bool((g.jac_to_affine(res[j]) == o.msm_affine(pts, vectors[j])).all())
        q.put((rank, ok, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize("k", [1, 5])
def test_replicated_batch_gloo_world2(k):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, k, 150, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in results), results


@pytest.mark.parametrize("curve,which,c", [("bn254", "g1", 16), ("bn254", "g1", 11), ("bls12_381", "g2", 13)])
def test_window_sharded_multiexp_gloo_world2(curve, which, c):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve, which, c, 300, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in results), results


def test_shard_plans_cover_everything():
    """Every (point, window) pair is owned by exactly one rank in both decompositions; all ranks agree on c."""
    import importlib
    gm = importlib.import_module("gnark-crypto_amd")
    sharding = importlib.import_module("gnark-crypto_amd.sharding")
    g = gm.G1Jac("bn254")
    for n in (1, 7, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            for mode in ("windows", "points", "auto"):
                plans = [sharding.shard_plan(g, n, r, world, mode) for r in range(world)]
                assert len({(p["c"], p["nwin"], p["rows"], p["mode"]) for p in plans}) == 1
                nwin = plans[0]["nwin"]
                cover = np.zeros((min(n, 64), nwin), dtype=np.int64)  # the first points are enough to see the pattern
                pts_owned = 0
                for p in plans:
                    wins = list(range(p["win_first"], nwin, p["win_stride"]))
                    assert len(wins) <= p["rows"]
                    lo, hi = p["lo"], min(p["hi"], cover.shape[0])
                    if hi > lo:
                        cover[lo:hi][:, wins] += 1
                    pts_owned += (p["hi"] - p["lo"]) * len(wins)
                assert (cover == 1).all()
                assert pts_owned == n * nwin


def test_sharding_layout():
    import importlib
    sharding = importlib.import_module("gnark-crypto_amd.sharding")
    for nwin in (16, 17, 24, 29):
        for world in (1, 2, 4, 8):
            owned = [sharding.owned_windows(nwin, r, world) for r in range(world)]
            assert sorted(w for o in owned for w in o) == list(range(nwin))
            per = sharding.slots_per_rank(nwin, world)
            g = np.zeros((world, per, 4), dtype=np.uint64)
            for r in range(world):
                loc = np.array([[w, w, w, w] for w in owned[r]], dtype=np.uint64).reshape(-1, 4)
                g[r] = sharding.pack_local(loc, nwin, world, 4)
            tot = sharding.unpack_gathered(g, nwin, world, 4)
            assert (tot[:, 0] == np.arange(nwin)).all()


def _wait_worker(rank, world, port, q):
    """bench.py's host_side_wait: rank 0 works while the others wait on a key of the rendezvous store (no collective)."""
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import time
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import bench

        t0 = time.perf_counter()

        def work():
            time.sleep(0.5)
            return {"by": rank}

        out = bench.host_side_wait(dist, rank, "unit_test_key", work)
        waited = time.perf_counter() - t0
        ok = (out == {"by": 0}) if rank == 0 else (out is None and waited >= 0.4)
        q.put((rank, ok, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))


def test_bench_host_side_wait_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_wait_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in results), results


def test_bench_launcher_asks_for_the_devices_it_needs():
    """`python bench.py --gpus N` as a plain command becomes the launcher of N ranks; on a machine with fewer devices it
    says so instead of failing inside a launcher (here: no device at all)."""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0
    assert "needs 2 devices" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_bench_sharded_inputs_are_the_same_on_every_rank(gm):
    """bench.py's N > 1 rows generate their synthetic inputs per rank and per chunk (chunked_scalars): whatever the point
    slices, the ranks must see slices of ONE array - otherwise the closed-form check (every rank contributes the dot product
    of its slice) and the window decomposition (every rank holds all points) would disagree on the MultiExp being computed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_cpu", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = gm.G1Jac("bn254")
    logn = 21  # two chunks of 2^20
    n = 1 << logn
    for tag in (0, 1):
        full = bench.chunked_scalars(g, logn, tag, 0, n)
        assert full.shape == (n, g.fr_limbs)
        for world in (2, 3, 8):
            parts = [bench.chunked_scalars(g, logn, tag, r * n // world, (r + 1) * n // world) for r in range(world)]
            assert (np.concatenate(parts) == full).all(), (tag, world)
        lo, hi = 1048570, 1048590  # a slice across the chunk boundary
        assert (bench.chunked_scalars(g, logn, tag, lo, hi) == full[lo:hi]).all()
    assert not (bench.chunked_scalars(g, logn, 0, 0, 64) == bench.chunked_scalars(g, logn, 1, 0, 64)).all()
    small = bench.chunked_scalars(g, 10, 0, 0, 1 << 10)  # sizes below one chunk
    assert small.shape == (1 << 10, g.fr_limbs) and (bench.chunked_scalars(g, 10, 0, 100, 200) == small[100:200]).all()
    assert bench.max_over_ranks(None, 1, {"a": 1.0}) == {"a": 1.0}
