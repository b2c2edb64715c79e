"""N > 1 host logic on CPU: world_size-2 gloo processes.  The per-rank arithmetic is stood in for by
the oracle (no GPU here); what is under test is the product's sharding, all-gather and fold."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from ecgpu import parallel
    from oracle import coracle as CO, ecmodel as M, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = M.K256

    def local_msm(lo, hi):
        s = CO.synth_scalars(0, hi - lo, synth.SEED, lo)
        p = CO.synth_points(0, hi - lo, synth.SEED, lo)
        aff = CO.msm_naive(0, s, p)
        pt = M.IDENTITY if aff[64] else (int.from_bytes(bytes(aff[:32]), "big"), int.from_bytes(bytes(aff[32:64]), "big"), 1)
        return np.frombuffer(M.proj_bytes(c, pt), dtype=np.uint8).copy()

    def add_points(a, b):
        return CO.point_op(0, 0, a.reshape(1, 96), b.reshape(1, 96))[0]

    res = parallel.msm_sharded(local_msm, add_points, n)
    lo, hi = parallel.shard_range(n, rank, world)
    # the tensor form bench.py uses (here over gloo with host tensors): every rank ends up with all the partial sums
    import torch
    part = torch.from_numpy(local_msm(lo, hi).copy())
    allp = torch.zeros((world, 96), dtype=torch.uint8)
    called = []
    parallel.allgather_into(allp, part, "gloo", synchronize=lambda: called.append(1))
    assert called == [1] and bytes(allp[rank].numpy()) == bytes(part.numpy())
    acc = allp[0].numpy()
    for r in range(1, world):
        acc = add_points(acc, allp[r].numpy())
    assert bytes(acc) == bytes(res)
    q.put((rank, lo, hi, bytes(res)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 64])
def test_msm_sharded_two_ranks_gloo(n):
    from oracle import coracle as CO, ecmodel as M, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == n          # disjoint cover
    assert got[0][3] == got[1][3]                                                  # every rank holds the same point
    want = CO.msm_naive(0, CO.synth_scalars(0, n, synth.SEED), CO.synth_points(0, n, synth.SEED))
    X, Y, Z = (int.from_bytes(got[0][3][32 * t:32 * t + 32], "big") for t in range(3))
    assert M.affine_bytes(M.K256, M.to_affine(M.K256, (X, Y, Z))) == bytes(want)


def test_shard_range_covers_everything():
    from ecgpu import parallel
    for n in (0, 1, 7, 8, 1000, 2**24):
        for world in (1, 2, 3, 8):
            edges = [parallel.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
