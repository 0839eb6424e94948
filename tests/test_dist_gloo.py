"""Multi-rank path on CPU: world_size 2 over gloo.  Images are sharded by contiguous blocks, each
rank runs the path on its shard (here with the CPU oracle standing in for the per-rank backend,
since there is no GPU), and the feature lists are gathered to rank 0 exactly as bench.py does
over RCCL.  Rank 0 checks the gathered lists against a single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fixtures
from hessgpu_amd import dist as hdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 5, 8, 63, 64):
        for world in (1, 2, 3, 8):
            spans = [hdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import OracleSession

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, last = hdist.shard_range(n_images, rank, world)
        imgs = np.stack([fixtures.synthetic_blobs(160, 120, i) for i in range(first, last)])
        o = OracleSession(threads=1, keep_levels=False, truncate_method=3, feature_count_threshold=64)
        counts = o.run(imgs)
        keys, desc = hdist.host_feature_tensors(o, counts)
        all_counts, gk, gd = hdist.gather_feature_lists(counts, keys, desc, dst=0)
        if rank == 0:
            flat_counts = [c for r in all_counts for c in r]
            # the other ranks' lists landed in the destination's host buffers (bench.py --gather-dest host): the
            # destination's own block stays where its context delivered it; buffers are reused from step to step
            landing = hdist.HostLanding()
            for _ in range(2):
                hk, hd = landing.land(gk, gd, own_rank=0)
            assert hk[0] is None and hd[0] is None
            for r in range(1, world):
                assert torch.equal(hk[r], gk[r]) and torch.equal(hd[r], gd[r]) and hk[r].data_ptr() != gk[r].data_ptr()
            lk = [gk[0]] + [hk[r] for r in range(1, world)]
            ld = [gd[0]] + [hd[r] for r in range(1, world)]
            np.savez(result_path, counts=np.array(flat_counts),
                     keys=np.concatenate([k.numpy() for k in lk]), desc=np.concatenate([d.numpy() for d in ld]))
        else:
            assert gk is None and gd is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process(tmp_path):
    from oracle_lib import OracleSession

    n_images, world = 4, 2
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(world, _free_port(), n_images, out), nprocs=world, join=True)
    got = np.load(out)
    imgs = np.stack([fixtures.synthetic_blobs(160, 120, i) for i in range(n_images)])
    o = OracleSession(threads=1, keep_levels=False, truncate_method=3, feature_count_threshold=64)
    counts = o.run(imgs)
    assert got["counts"].tolist() == counts and sum(counts) > 0
    ks, ds = [], []
    for b in range(n_images):
        k, d = o.fetch(b)
        ks.append(np.frombuffer(k.tobytes(), np.uint8).reshape(-1, 24))
        ds.append(d)
    assert np.array_equal(got["keys"], np.concatenate(ks))
    assert np.array_equal(got["desc"].view(np.uint32), np.concatenate(ds).view(np.uint32))


def test_gather_handles_ranks_with_no_features(tmp_path):
    """Empty and ragged lists: the destination itself contributes nothing, the others blocks of different
    sizes; every rank sends exactly its own records (SURVEY 8c edge cases)."""
    out = str(tmp_path / "g2.npz")
    mp.spawn(_ragged_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    assert got["counts"].tolist() == [0, 3, 5]
    assert got["sizes"].tolist() == [0, 3, 5]                      # received blocks are not padded to the largest
    assert got["keys"].shape == (8, 24) and got["desc"].shape == (8, 128)
    assert got["keys"][:, 0].tolist() == [1, 1, 1, 2, 2, 2, 2, 2]
    assert np.array_equal(got["desc"][:, 0], np.array([1.0, 2.0, 3.0, 1.0, 2.0, 3.0, 4.0, 5.0], np.float32))


def _ragged_worker(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = [0, 3, 5][rank]
        keys = torch.full((n, 24), rank, dtype=torch.uint8)
        desc = torch.arange(1, n + 1, dtype=torch.float32)[:, None].repeat(1, 128)
        all_counts, gk, gd = hdist.gather_feature_lists([n], keys, desc, dst=0)
        if rank == 0:
            np.savez(result_path, counts=np.array([c for r in all_counts for c in r]), sizes=np.array([len(k) for k in gk]),
                     keys=np.concatenate([k.numpy() for k in gk]), desc=np.concatenate([d.numpy() for d in gd]))
    finally:
        dist.destroy_process_group()


def test_gather_inside_a_subgroup_uses_group_local_ranks(tmp_path):
    """`dst` of gather_feature_lists is a rank INSIDE the group: with the sub-group (1, 2) of a 3-rank job,
    dst=1 is global rank 2 (global rank 1 is nobody's destination)."""
    out = str(tmp_path / "g3.npz")
    mp.spawn(_subgroup_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    assert got["dst_global"] == 2 and got["counts"].tolist() == [2, 4]          # group order: global 1, then global 2
    assert got["keys"][:, 0].tolist() == [1, 1, 2, 2, 2, 2]


def _subgroup_worker(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = dist.new_group(ranks=[1, 2])          # every rank takes part in creating it
        if rank in (1, 2):
            n = {1: 2, 2: 4}[rank]
            keys = torch.full((n, 24), rank, dtype=torch.uint8)
            desc = torch.full((n, 128), float(rank), dtype=torch.float32)
            all_counts, gk, gd = hdist.gather_feature_lists([n], keys, desc, dst=1, group=grp)
            if dist.get_rank(grp) == 1:
                np.savez(result_path, dst_global=rank, counts=np.array([c for r in all_counts for c in r]),
                         keys=np.concatenate([k.numpy() for k in gk]))
            else:
                assert gk is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_helper_thread_exchange_keeps_step_order_and_only_dst_learns_the_counts(tmp_path):
    """bench.py's N > 1 step loop: every step's exchange is posted to a GatherWorker (one helper thread per rank, jobs
    in step order), counts travel to the destination alone.  Three ranks, five steps of different sizes per rank, the
    submitting thread posts all of them before it asks for the first result."""
    out = str(tmp_path / "g4.npz")
    mp.spawn(_helper_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    # step s, rank r contributes r + s records whose first byte is 16 * s + r
    for s in range(5):
        assert got[f"counts{s}"].tolist() == [[0 + s], [1 + s], [2 + s]]
        assert got[f"first{s}"].tolist() == sum(([16 * s + r] * (r + s) for r in range(3)), [])


def _helper_worker(rank, world, port, result_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = hdist.GatherWorker(torch.device("cpu"))

        def step(s):
            n = rank + s
            keys = torch.full((n, 24), 16 * s + rank, dtype=torch.uint8)
            desc = torch.full((n, 128), float(s), dtype=torch.float32)
            return hdist.gather_feature_lists([n], keys, desc, dst=0, counts_to_dst_only=True)

        tickets = [w.post(step, s) for s in range(5)]
        res = [hdist.GatherWorker.result(t, timeout=120) for t in tickets]
        if rank == 0:
            np.savez(result_path, **{f"counts{s}": np.array(r[0]) for s, r in enumerate(res)},
                     **{f"first{s}": np.concatenate([k.numpy()[:, 0] for k in r[1]]) for s, r in enumerate(res)})
        else:
            for s, (allc, gk, gd) in enumerate(res):
                assert gk is None and gd is None
                assert allc[rank] == [rank + s] and all(allc[r] is None for r in range(world) if r != rank)
        # a job that raises is handed to whoever asks, and every later job fails too (the ranks are out of step)
        bad = w.post(lambda: 1 // 0)
        after = w.post(lambda: 5)
        with pytest.raises(ZeroDivisionError):
            hdist.GatherWorker.result(bad, timeout=30)
        with pytest.raises(RuntimeError):
            hdist.GatherWorker.result(after, timeout=30)
        w.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()
