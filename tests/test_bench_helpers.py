"""bench.py's host-side arithmetic (no GPU): the structured MSM inputs P_i = (a0 + i d) G and the closed form the summed
result is checked against, and the work-accounting tables."""
import numpy as np

import bench
from oracle import coracle as CO
from oracle import ecmodel as M


def test_structured_scalars_and_expected_sum():
    n, first = 1500, (1 << 33) + 12345
    v = bench.structured_point_scalars(first, n)
    assert v.shape == (n, 32) and v.dtype == np.uint8
    for j in (0, 1, 7, n - 1):
        assert int.from_bytes(bytes(v[j]), "big") == bench.MSM_A0 + (first + j) * bench.MSM_D
    ks = CO.synth_scalars(0, n, bench.SEED, first)
    kb = ks.tobytes()
    want = sum(int.from_bytes(kb[32 * j:32 * j + 32], "big") * (bench.MSM_A0 + (first + j) * bench.MSM_D) for j in range(n)) % M.K256.n
    assert bench.msm_expected_scalar(ks, first, M.K256.n) == want
    # the closed form really is the MSM of those points: check on a handful of terms with the big-integer model
    m = 6
    pts = [M.affine_mul(M.K256, int.from_bytes(bytes(v[j]), "big"), (M.K256.gx, M.K256.gy)) for j in range(m)]
    tot = None
    for j in range(m):
        tot = M.affine_add(M.K256, tot, M.affine_mul(M.K256, int.from_bytes(kb[32 * j:32 * j + 32], "big"), pts[j]))
    assert tot == M.affine_mul(M.K256, bench.msm_expected_scalar(ks[:m], first, M.K256.n), (M.K256.gx, M.K256.gy))


def test_work_tables_cover_every_workload():
    for name, wl in bench.WORKLOADS.items():
        key = name if name != "k256_varbase" else "k256_varbase_fast"
        m, s = bench.WORK[key]
        assert m > 0 and s > 0 and wl["curve"] in bench.MAC_CONV and wl["curve"] in bench.MAC_ISSUED
    assert "k256_varbase_ref" in bench.WORK
    assert set(bench.OTHER_CONFIGS) <= set(bench.WORKLOADS)


def test_spread_blocks_end_at_the_batch_end():
    n, sample = 1 << 24, 983040
    st = bench.spread_blocks(n, sample)
    assert len(st) == bench.SPREAD_BLOCKS and st[0] == sample and st[-1] + bench.SPREAD_LEN == n
    assert bench.spread_blocks(4096, 4096) == []
