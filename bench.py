#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): k256 variable-base scalar multiplications per second.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  The driver launches N > 1 through torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE in the environment); called by hand with --gpus N > 1 and no WORLD_SIZE, this script
starts that launcher itself as a child process (before anything touches the GPU) and relays its
output and exit code.

A "step" is one pass of the hot path over one batch: 2^24 independent (scalar, point) pairs per GPU
(BASELINE.json configs[1]), inputs generated on the device from the synthetic-input spec
(oracle/synth.py) and resident in HBM before the timed region.  Independent batches shard across
GPUs with no data-path collective (weak scaling: per-GPU work is fixed); the split MSM
(--workload k256_msm) all-gathers one projective point per rank and folds them on the device.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  roofline      the dominant kernel against its bound.  This path is integer-VALU bound (no MFMA,
                ~0.2 % of HBM): achieved/peak are 32x32-bit multiply-accumulates per second.
                `peak` is the datasheet half-rate figure, `peak_measured` the v_mad_u64_u32 issue rate
                measured on this GPU in this run (tools/ubench/peak.hip).  `traffic` and everything under
                roofline["pmc"] are NOT measured by this run: they are read from the committed rocprofv3
                PMC summary of this command (profiles/pmc_summary_r*_<workload>.json) and roofline["pmc"]
                says which file, which commit it was profiled at and whether the kernel sources of this
                tree still hash to what was profiled; at N > 1 they are left out.  The HBM view of the
                same kernel is under roofline["hbm"].
  cpu_baseline  the C restatement of the reference CPU path (oracle/ecoracle.c, "port": rustc is not
                available) timed on this box's host cores on a bounded sample of the same workload
                (median of 5, all cores and single thread), and used to check the GPU output of that
                sample byte for byte.
  host_io       (N = 1, variable-base workloads) the same batch handed over in HOST memory - what a drop-in caller of the reference's
                trait surface does - through the chunked pipeline of csrc/host_pipe.hpp: PCIe-inclusive rate from page-locked and from
                pageable buffers, each compared byte for byte with the device-resident result.  Never `value`.
  other_configs (N = 1, default workload only) the other BASELINE.json configs - p256 fixed base 2^24,
                k256 MSM 2^23 terms, p384 variable base 2^22 - run for a few steps each after the
                headline, with their own parity checks, so that every config has a driver-run number.
Every run checks parity: N = 1 against the CPU sample, N > 1 every rank's first and last 4 096 units
against the oracle (MSM: the sum over ALL terms of all ranks against the closed form of structured
points) - a mismatch anywhere makes the run exit non-zero.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))

SEED = 0xEC5CA1A5

# ---- algorithmic work per unit ---------------------------------------------------------------
# Work is counted in field multiplications (M) and squarings (S) per unit of the schedule that runs, with the EXPECTED
# number of table additions (a signed 4-bit digit is zero with probability 1/16; the first addition into an empty
# accumulator is a copy).  Two conversions to 32x32-bit multiply-accumulates:
#   mac          SURVEY.md 8d convention: 64 (256-bit) or 144 (384-bit) per M or S - the schoolbook product only
#   mac_issued   what the kernels really issue: k256 M = 64 + 8 fold products, S = 36 cross + 8 squares + 8 folds + 1;
#                p256 M = S = 64 + 24 reduction terms; p384 M = 144 + 24, S = 66 + 12 + 48 (dedicated squaring)
MAC_CONV = {"k256": 64, "p256": 64, "p384": 144}
MAC_ISSUED = {"k256": (72, 53), "p256": (88, 88), "p384": (168, 126)}
# secp256k1 mixed additions compute Y3 = R (V - X3) - Y1 HHH as ONE fused difference of two products (round 3: k256::mul_add2,
# one reduction for both): 8 fold products less per addition in the ISSUED count; the algorithmic count (two multiplications) stays.
# The secp256k1 doubling does the same with Y3 = E (D - X3) - 8 B^2: the squaring B^2 becomes the second product of the fused form (72 issued
# products instead of 53, no fold of its own: + 19 - 8 = + 11 per doubling; FUSED_DBL).
FUSED_PAIRS = {"k256_varbase_fast": 52 * 31 / 32, "k256_msm": 14, "k256_ecdsa_verify": 52 * 31 / 32 + 11}
FUSED_DBL = {"k256_varbase_fast": 125, "k256_ecdsa_verify": 125}

WORK = {
    # reference schedule: 128 doublings (6M+2S) + 80 complete additions (12M) + to_affine (255S + 17M)
    "k256_varbase_ref": (128 * 6 + 80 * 12 + 17, 128 * 2 + 255),
    # throughput schedule (csrc/mulfast_k256.hpp), batch of 32 results per inversion; signed 5-bit windows since round 4 (26 positions per GLV half)
    #   table   : [P .. 16P] as a co-Z chain: doubling with update 2M+4S + 14 co-Z additions (4M+2S, no Z) + rescale of 14 entries
    #             14x(4M+1S) + 16 beta*x + common Z 1M                                                   = 131M +  46S   (8 entries until round 3: 59M + 22S)
    #   loop    : 25 x 5 dbl (3M+4S) + 52 * 31/32 mixed adds (8M+3S)                                     = 778M + 651.1S (4-bit windows: 128 dbl, 66 * 15/16 adds)
    #   output  : 2M (global Z) + batched normalise 6M+1S + (255S+15M)/32                                =   9M +   9S
    "k256_varbase_fast": (131 + 125 * 3 + 52 * 31 / 32 * 8 + 9, 46 + 125 * 4 + 52 * 31 / 32 * 3 + 9),
    # 10 signed 26-bit windows (21.5 GB table), XYZZ accumulator (round 3): the first entry is a copy, the second meets ZZ = ZZZ = 1
    # (4M+2S), 8 mixed additions 8M+2S, normalise 8M + (255S+12M)/64.  (Until then counted as 9 Jacobian additions 8M+3S and 6M+1S.)
    "p256_fixedbase": (4 + 8 * 8 + 8 + 12 / 64, 2 + 8 * 2 + 255 / 64),
    # signed 5-bit windows since round 4 (77 positions): 76 x 5 doublings (4M+4S) + 73.6 mixed additions 8M+3S (77 digits x 31/32, the first one a
    # copy) + table [P .. 16P] as a co-Z chain (doubling with update 2M+4S, 14 co-Z additions 4M+2S, 14M for the denominators = 72M+32S) + table to
    # affine through ONE inverted denominator and the chain's ratios (77M+15S) + (385S+14M)/16 + output normalise 6M+1S + (385S+14M)/16 (16 units per
    # lane and pass share the two inversions).  (Rounds 2-3: 4-bit windows, 96 x 4 doublings, 89 additions, 8-entry tables: 4 200 per unit.)
    # (the TABLE inversion is shared by the 2 units a wave draws at a time since the dynamic scheduling of round 4 - (385S+14M)/2 -, the output inversion by 16)
    "p384_varbase": (1520 + (77 * 31 / 32 - 1) * 8 + 72 + 77 + 14 / 2 + 6 + 14 / 16, 1520 + (77 * 31 / 32 - 1) * 3 + 32 + 15 + 385 / 2 + 1 + 385 / 16),
    # bucket method with GLV halves, 7 windows of 18 / 19 bits at this size: 14 XYZZ mixed additions (8M+2S) per term;
    # per-term share of the endomorphism (1M), of the bucket pieces and of the bucket reduction (1.8 M buckets: XYZZ -> Jacobian
    # and two general additions 12M+4S each; 1.5 M pieces folded) ~ 9M + 3S
    "k256_msm": (14 * 8 + 1 + 9, 14 * 2 + 3),
    # verification = u2 Q (headline kernel) + u1 G (20-bit table at this batch size) + prep / check (57 scalar-field equivalents + 7)
    "k256_ecdsa_verify": (131 + 125 * 3 + 52 * 31 / 32 * 8 + 9 + 4 + 11 * 8 + 8 + 64, 46 + 125 * 4 + 52 * 31 / 32 * 3 + 9 + 2 + 11 * 2 + 4),
    # p256 verification = u2 Q (vb::mul_kernel<CurveP256,16,4>: 256 doublings 4M+4S, 60 general additions 11M+5S, table 4 dbl + 3 add,
    # output normalise 6M+1S + (255S+12M)/8 = 1 740 M + 1 388 S; counted with 8 units per inversion, 16 since round 3: -1 %) + u1 G (20-bit table, XYZZ: 100 M + 28 S) + prep / check (~64 M)
    "p256_ecdsa_verify": (1740 - 28 + 100 + 64, 1388 - 16 + 28),     # - 28 M, - 16 S: the co-Z table chain of round 3
}

WORKLOADS = {
    "k256_varbase": dict(curve="k256", cid=0, log2n=24, fixed=False, msm=False, metric="k256 variable-base scalar-muls/sec", unit="scalar-muls/s",
                         bytes_per_unit=32 + 64 + 65, kernel="k256_mul_fast_kernel<32,4>", pmc_match="k256_mul_fast_kernel",
                         desc="k256 variable-base scalar multiplication, 2^%d independent (scalar, point) pairs per GPU, affine output"),
    "p256_fixedbase": dict(curve="p256", cid=1, log2n=24, fixed=True, msm=False, metric="p256 fixed-base (mul_by_generator) scalar-muls/sec", unit="scalar-muls/s",
                           bytes_per_unit=32 + 65, kernel="fb::mul_wide_kernel<CurveP256,26,64,4>", pmc_match="P256Params>, 26, 64, 4",
                           desc="p256 mul_by_generator, 2^%d independent scalars per GPU, affine output"),
    "p384_varbase": dict(curve="p384", cid=2, log2n=22, fixed=False, msm=False, metric="p384 variable-base scalar-muls/sec", unit="scalar-muls/s",
                         bytes_per_unit=48 + 96 + 97, kernel="vb::mul_kernel<CurveP384,16,4,1,5>", pmc_match="vb::mul_kernel",
                         desc="p384 variable-base scalar multiplication, 2^%d independent (scalar, point) pairs per GPU, affine output"),
    "k256_msm": dict(curve="k256", cid=0, log2n=23, fixed=False, msm=True, metric="k256 MSM points/sec", unit="points/s",
                     bytes_per_unit=32 + 64, kernel="msm pipeline (digits / sort / bucket sums / reduce)", pmc_match="bucket_sum_kernel",
                     desc="k256 multi-scalar multiplication, 2^%d terms per GPU (one sum; ranks exchange one point each), affine output"),
    "k256_ecdsa_verify": dict(curve="k256", cid=0, log2n=22, fixed=False, msm=False, ecdsa=True, metric="k256 ECDSA verifications/sec", unit="verifications/s",
                              bytes_per_unit=32 + 64 + 64 + 1, kernel="verify_prep + fb::mul_wide_kernel + k256_mul_fast_kernel<32,4> + verify_check", pmc_match="k256_mul_fast_kernel",
                              desc="k256 ECDSA verify_prehashed (low-s rule), 2^%d independent (prehash, signature, public key) triples per GPU"),
    "p256_ecdsa_verify": dict(curve="p256", cid=1, log2n=22, fixed=False, msm=False, ecdsa=True, metric="p256 ECDSA verifications/sec", unit="verifications/s",
                              bytes_per_unit=32 + 64 + 64 + 1, kernel="verify_prep + fb::mul_wide_kernel + vb::mul_kernel<CurveP256,16,4> + verify_check", pmc_match="vb::mul_kernel",
                              desc="p256 ECDSA verify_prehashed, 2^%d independent (prehash, signature, public key) triples per GPU"),
}
OTHER_CONFIGS = ("p256_fixedbase", "k256_msm", "p384_varbase")      # BASELINE.json configs 3, 4 (one GPU's share), 5

# v_mad_u64_u32 at half the FP32-FMA rate: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz (datasheet-derived bound)
PEAK_TMACS = 256 * 4 * 16 * 2.4e9 / 1e12
PEAK_HBM_GBS = 8000.0

ECDSA_CORRUPT_EVERY = 7     # every 7th signature of the synthetic ECDSA batch has one bit of s flipped (must be rejected)
CHECK_LEN = 4096            # N > 1: units at each end of a rank's slice compared with the oracle
MSM_A0 = 0x1234567890ABCDEF1234567890ABCDEF0F1E2D3C4B5A6978          # structured MSM points P_i = (A0 + i * D) G
MSM_D = 0xFEDCBA0987654321


# ---------------------------------------------------------------------------------------------------------------------
# structured MSM inputs: P_i = (a0 + i d) G, so sum k_i P_i = (sum k_i (a0 + i d) mod n) G
# ---------------------------------------------------------------------------------------------------------------------
def structured_point_scalars(first, n):
    """(a0 + i d) for i in [first, first + n) as big-endian 32-byte rows; a0 < 2^192, d < 2^64, i < 2^40: no reduction needed."""
    import numpy as np
    i = np.arange(first, first + n, dtype=np.uint64)
    d0, d1 = np.uint64(MSM_D & 0xFFFFFFFF), np.uint64(MSM_D >> 32)
    ilo, ihi = i & np.uint64(0xFFFFFFFF), i >> np.uint64(32)
    limbs = [None] * 8                      # little-endian 32-bit limbs as uint64 columns, carries propagated below
    # i * d = (ilo + 2^32 ihi)(d0 + 2^32 d1)
    t0 = ilo * d0
    t1 = ilo * d1 + ihi * d0                # < 2^65 only if ihi large; i < 2^40 keeps ihi < 2^8, so no overflow
    t2 = ihi * d1
    acc = [t0 & np.uint64(0xFFFFFFFF), (t0 >> np.uint64(32)) + (t1 & np.uint64(0xFFFFFFFF)), (t1 >> np.uint64(32)) + (t2 & np.uint64(0xFFFFFFFF)), t2 >> np.uint64(32)]
    acc += [np.zeros(n, dtype=np.uint64) for _ in range(4)]
    carry = np.zeros(n, dtype=np.uint64)
    for l in range(8):
        v = acc[l] + np.uint64((MSM_A0 >> (32 * l)) & 0xFFFFFFFF) + carry
        limbs[l] = (v & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        carry = v >> np.uint64(32)
    be = np.stack([limbs[7 - j] for j in range(8)], axis=1).astype(">u4")
    return np.ascontiguousarray(be).view(np.uint8).reshape(n, 32)


def msm_expected_scalar(ks, first, order):
    """sum_j k_j * (a0 + (first + j) d) mod order, from the big-endian scalar rows ks: two integer sums, no big-int loop."""
    import numpy as np
    n = ks.shape[0]
    k16 = ks.view(">u2").reshape(n, 16)
    j = np.arange(n, dtype=np.int64)
    jl, jh = j & 0xFFF, j >> 12
    s0 = s1 = 0
    for col in range(16):
        c = k16[:, col].astype(np.int64)
        w = 16 * (15 - col)
        s0 += int(c.sum()) << w
        s1 += (int(np.dot(jl, c)) + (int(np.dot(jh, c)) << 12)) << w
    return ((MSM_A0 + first * MSM_D) * s0 + MSM_D * s1) % order


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (forked children, BEFORE the parent touches the GPU)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_ecdsa_worker(args):
    first, n, cid, reps = args
    import numpy as np
    from oracle import coracle as CO
    d = CO.synth_scalars(cid, n, SEED, first)
    k = CO.synth_scalars(cid, n, SEED + 1, first)
    z = CO.synth_scalars(cid, n, SEED + 2, first)
    low_s = (cid == 0)
    sig, rec, ok = CO.ecdsa_sign_batch(cid, d, k, z, low_s=low_s)
    q = CO.lincomb_batch(cid, d, None, threads=1)[:, :-1].copy()
    idx = np.arange(first, first + n)
    sig[idx % ECDSA_CORRUPT_EVERY == 0, -1] ^= 1
    times, v = [], None
    per = max(1, n // reps)
    for r in range(reps):
        lo, hi = r * per, (n if r == reps - 1 else (r + 1) * per)
        t0 = time.perf_counter()
        part = CO.ecdsa_verify_batch(cid, z[lo:hi], sig[lo:hi], q[lo:hi], low_s=low_s)
        times.append((time.perf_counter() - t0, hi - lo))
        v = part if v is None else np.concatenate([v, part])
    return first, n, times, sig.tobytes() + v.tobytes()


def cpu_baseline_worker(args):
    """C oracle on a slice, timed in `reps` equal chunks (the median chunk rate is what gets reported)."""
    first, n, cid, fixed, kind, reps = args
    if kind == "ecdsa":
        return cpu_ecdsa_worker((first, n, cid, reps))
    import numpy as np
    from oracle import coracle as CO
    s = CO.synth_scalars(cid, n, SEED, first)
    if fixed:
        p = None
    elif kind == "msm":
        p = CO.lincomb_batch(cid, structured_point_scalars(first, n), None, threads=1)[:, :-1].copy()     # untimed: P_i = (a0 + i d) G
    else:
        p = CO.synth_points(cid, n, SEED, first)
    times, out = [], None
    per = max(1, n // reps)
    for r in range(reps):
        lo, hi = r * per, (n if r == reps - 1 else (r + 1) * per)
        if lo >= hi:
            continue
        t0 = time.perf_counter()
        # MSM: the reference has no bucket method; its large-N form is one reference multiplication per term plus an addition
        part = CO.lincomb_batch(cid, s[lo:hi], None if p is None else p[lo:hi], out_proj=(kind == "msm"), threads=1)
        times.append((time.perf_counter() - t0, hi - lo))
        out = part if out is None else np.concatenate([out, part])
    return first, n, times, out.tobytes()


SPREAD_BLOCKS, SPREAD_LEN = 16, 4096     # extra parity blocks spread over the batch (sampled indices + the tail)


def spread_blocks(n, sample):
    if n <= sample + SPREAD_BLOCKS * SPREAD_LEN:
        return []
    span = n - sample - SPREAD_LEN
    return [sample + (span * j) // (SPREAD_BLOCKS - 1) for j in range(SPREAD_BLOCKS)]


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def run_cpu_baseline(sample, procs, cid, fixed, kind, n, reps=5, single=True):
    """Reference CPU path (C port) on `sample` units: all cores (one single-threaded process per core, each slice timed
    in `reps` chunks; per chunk index the rate is units / slowest worker; median over the chunks) and one thread alone
    (median of `reps` runs on a smaller slice).  For the element-wise workloads also SPREAD_BLOCKS blocks across the rest
    of the batch, checked but not timed."""
    from concurrent.futures import ProcessPoolExecutor
    per = (sample + procs - 1) // procs
    jobs = [(i * per, min(per, sample - i * per), cid, fixed, kind, reps) for i in range(procs) if i * per < sample]
    extra = [] if kind in ("msm", "ecdsa") else [(st, SPREAD_LEN, cid, fixed, kind, 1) for st in spread_blocks(n, sample)]
    extra_res = []
    if extra:
        with ProcessPoolExecutor(max_workers=procs) as ex:
            extra_res = [(r[0], r[1], r[3]) for r in ex.map(cpu_baseline_worker, extra)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=procs) as ex:
        res = list(ex.map(cpu_baseline_worker, jobs))
    wall = time.perf_counter() - t0
    nchunks = min(len(r[2]) for r in res)
    rates = []
    for ci in range(nchunks):
        units = sum(r[2][ci][1] for r in res)
        rates.append(units / max(r[2][ci][0] for r in res))
    single_rate = None
    if single:
        sn = max(256, min(per, 4096 if cid == 0 else 1024 if cid == 1 else 512))
        with ProcessPoolExecutor(max_workers=1) as ex:
            r1 = list(ex.map(cpu_baseline_worker, [(0, sn, cid, fixed, kind, reps)]))[0]
        single_rate = median([u / t for t, u in r1[2]])
    return {"wall_s": wall, "rate": median(rates), "rates": rates, "single_rate": single_rate, "out": b"".join(r[3] for r in sorted(res)), "procs": len(jobs),
            "parts": [(r[0], r[1], r[3]) for r in sorted(res)], "extra": extra_res, "reps": reps}


# ---------------------------------------------------------------------------------------------------------------------
def maybe_launch_ranks(args):
    """--gpus N > 1 without a launcher: become the parent of `python -m torch.distributed.run ... bench.py ...`.
    Runs before torch / ecgpu are imported, so this process never initialises the GPU (an exec after that is forbidden)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def load_peak_lib():
    import ctypes
    path = os.path.join(ROOT, "tools", "ubench", "libecpeak.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.ecpeak_measure.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    return lib


def measure_peak(device):
    """(mad-only TMAC/s, mad+addc TMAC/s) measured now on this GPU, or (None, None) when the tool library is not built."""
    import ctypes
    lib = load_peak_lib()
    if lib is None:
        return None, None
    a, b = ctypes.c_double(), ctypes.c_double()
    if lib.ecpeak_measure(device, ctypes.byref(a), ctypes.byref(b)) != 0:
        return None, None
    return a.value, b.value


def csrc_sha16(workload=None):
    """Hash of the kernel sources a workload runs (csrc/*; the MSM's own files count for the MSM only): ties a committed PMC
    summary to the code it was profiled on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "rustcrypto-elliptic-curves_amd", "csrc", "*"))):
        if os.path.basename(path).startswith("msm") and workload is not None and not WORKLOADS[workload]["msm"]:
            continue
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def pmc_summary(workload, log2n, default_size):
    """The committed rocprofv3 PMC summary of this workload (profiles/pmc_summary_r*_<workload>.json, written by
    tools/profile_workload.sh), at the config's own size only; the newest round wins.  -> (summary, file name)"""
    if log2n != default_size:
        return {}, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_summary_r*_%s.json" % workload)), reverse=True):
        try:
            with open(path) as f:
                js = json.load(f)
        except Exception:
            continue
        if js.get("workload") == workload:
            return js, os.path.relpath(path, ROOT)
    return {}, None


# ---------------------------------------------------------------------------------------------------------------------
def run_workload(env, name, log2n, steps, warmup, schedule, cpu):
    """Generate inputs on the device, time `steps` passes, check parity.  env: torch, ecgpu, ctx, dev, rank, world, dist, backend."""
    import numpy as np
    torch, ecgpu, ctx, dev = env["torch"], env["ecgpu"], env["ctx"], env["dev"]
    rank, world, dist, backend = env["rank"], env["world"], env["dist"], env["backend"]
    from oracle import coracle as CO
    from oracle import ecmodel as M
    from ecgpu import parallel
    wl = WORKLOADS[name]
    n = 1 << log2n
    cv = ctx.curve(wl["curve"])
    nb = cv.nb
    first = rank * n                         # disjoint slices of one global batch
    d_s = torch.empty((n, nb), dtype=torch.uint8, device=dev)
    d_p = None if wl["fixed"] else torch.empty((n, 2 * nb), dtype=torch.uint8, device=dev)
    d_o = torch.empty((1 if wl["msm"] else n, 2 * nb), dtype=torch.uint8, device=dev)
    d_i = torch.empty((n,), dtype=torch.uint8, device=dev)
    cv.synth_scalars_device(d_s, n, SEED, first)
    if wl["msm"]:
        # structured points P_i = (a0 + i d) G, made on the device by the fixed-base kernel (untimed)
        d_v = torch.from_numpy(structured_point_scalars(first, n)).to(dev)
        cv.mul_device(d_v, None, d_p, n)
        ctx.synchronize()
        del d_v
    elif d_p is not None:
        cv.synth_points_device(d_p, n, SEED, first)
    if wl.get("ecdsa"):
        # d_s = secret keys; nonces and prehashes from two more seeded streams; public keys and signatures are made on
        # the device (fixed-base kernel, sign pipeline) before the timed region; d_p holds the public keys
        d_k = torch.empty((n, nb), dtype=torch.uint8, device=dev)
        d_z = torch.empty((n, nb), dtype=torch.uint8, device=dev)
        d_sig = torch.empty((n, 2 * nb), dtype=torch.uint8, device=dev)
        cv.synth_scalars_device(d_k, n, SEED + 1, first)
        cv.synth_scalars_device(d_z, n, SEED + 2, first)
        cv.mul_device(d_s, None, d_p, n)
        cv.ecdsa_sign_device(d_s, d_k, d_z, d_sig, None, d_i, n, flags=cv.default_ecdsa_flags() | ecgpu.PUBLIC_SCALARS)
        ctx.synchronize()
        bad = (torch.arange(first, first + n, device=dev) % ECDSA_CORRUPT_EVERY) == 0
        d_sig[bad, -1] ^= 1
    torch.cuda.synchronize()
    table = None
    if wl["fixed"]:
        # the generator table is built (and its memory taken) by the first call of a size class: outside the timed region,
        # but on the record - first-call time, steady-state time of the same call, and what the context holds afterwards
        t0 = time.perf_counter()
        cv.mul_device(d_s, None, d_o, n, d_out_inf=d_i)
        ctx.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        cv.mul_device(d_s, None, d_o, n, d_out_inf=d_i)
        ctx.synchronize()
        again_ms = (time.perf_counter() - t0) * 1e3
        tb, tw = ctx.fb_table_bytes(wl["curve"])
        table = {"table_bytes": tb, "widest_window_bits": tw, "first_call_ms": first_ms, "table_build_ms": max(0.0, first_ms - again_ms),
                 "note": "built by the first call of this size class, outside the timed region; kept by the context"}
    d_part = torch.empty((3 * nb,), dtype=torch.uint8, device=dev)
    d_all = torch.empty((max(world, 1), 3 * nb), dtype=torch.uint8, device=dev)
    d_fold = torch.empty((max(world, 1), 3 * nb), dtype=torch.uint8, device=dev)       # scratch of the fold of the gathered points
    d_fold_inf = torch.empty((1,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        if wl.get("ecdsa"):
            cv.ecdsa_verify_device(d_z, d_sig, d_p, d_i, n)
        elif wl["msm"]:
            if dist is None:
                cv.msm_device(d_s, d_p, n, d_o)
            else:
                # every rank sums its slice; one projective point per rank is all-gathered (96 bytes each; elliptic-curve
                # addition is not an RCCL reduction operator) and the world's points are folded on the device
                cv.msm_device(d_s, d_p, n, d_part, out_format=ecgpu.PROJECTIVE)
                parallel.allgather_into(d_all, d_part, backend, synchronize=ctx.synchronize)
                if backend != "nccl":
                    torch.cuda.synchronize()
                parallel.fold_points_device(cv, d_all, world, d_fold, d_o, d_fold_inf)
        else:
            cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i, flags=(ecgpu.EXACT_REFERENCE if schedule == "ref" else 0))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()                        # HIP events on the launch stream bracket the same region
    for _ in range(steps):
        step()
    kernel_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=(dev if backend == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity ---------------------------------------------------------------------------------------------------------
    parity, checked = True, []
    threads = max(1, min(8, (os.cpu_count() or 1) // max(1, min(world, 8))))
    if wl["msm"]:
        # the sum over ALL terms of all ranks against the closed form (O(N) scalar-field arithmetic on the host)
        ks = CO.synth_scalars(0, n, SEED, first)
        t_local = msm_expected_scalar(ks, first, M.K256.n)
        del ks
        if dist is not None:
            objs = [None] * world
            dist.all_gather_object(objs, t_local)
            t_total = sum(objs) % M.K256.n
        else:
            t_total = t_local
        want = M.affine_mul(M.K256, t_total, (M.K256.gx, M.K256.gy))
        got = bytes(d_o.cpu().numpy().reshape(-1))
        parity &= (got == (bytes(64) if want is None else M.i2b(M.K256, want[0]) + M.i2b(M.K256, want[1])))
        checked.append("sum over all %d terms equals (sum k_i (a0 + i d) mod n) G for the structured points P_i = (a0 + i d) G" % (world * n))
        if cpu is not None:
            # and the first m terms against the oracle's term-by-term products folded with its complete addition
            m = cpu["sample"]
            parts = np.frombuffer(cpu["out"], dtype=np.uint8).reshape(m, 3 * nb)
            ident = np.frombuffer(b"".join([bytes(nb), (1).to_bytes(nb, "big"), bytes(nb)]), dtype=np.uint8)[None, :]
            while parts.shape[0] > 1:
                if parts.shape[0] % 2:
                    parts = np.concatenate([parts, ident])
                parts = CO.point_op(wl["cid"], 0, parts[0::2].copy(), parts[1::2].copy())
            d_chk = torch.empty((3 * nb,), dtype=torch.uint8, device=dev)
            cv.msm_device(d_s, d_p, m, d_chk, out_format=ecgpu.PROJECTIVE)
            ctx.synchronize()
            g = d_chk.cpu().numpy().reshape(1, -1)

            def aff(b):
                X, Y, Z = (int.from_bytes(bytes(b[0][nb * t:nb * (t + 1)]), "big") for t in range(3))
                return M.to_affine(M.K256, (X, Y, Z))

            parity &= (aff(parts) == aff(g))
            checked.append("first %d terms against the C oracle" % m)
    elif wl.get("ecdsa"):
        g_ok = d_i.cpu().numpy()
        want_ok = (np.arange(first, first + n) % ECDSA_CORRUPT_EVERY != 0)
        parity &= bool((g_ok.astype(bool) == want_ok).all())
        checked.append("accept / reject flags of all units (every %dth signature corrupted)" % ECDSA_CORRUPT_EVERY)
        if cpu is not None:
            m = cpu["sample"]
            g_sig = d_sig[:m].cpu().numpy()
            for lo, cnt, blob in cpu["parts"]:
                parity &= (g_sig[lo:lo + cnt].tobytes() == blob[:cnt * 2 * nb]) and (g_ok[lo:lo + cnt].tobytes() == blob[cnt * 2 * nb:])
            checked.append("signatures and flags of the first %d units against the C oracle" % m)
    else:
        def rows(lo, cnt):
            return torch.cat([d_o[lo:lo + cnt], d_i[lo:lo + cnt, None]], dim=1).cpu().numpy().tobytes()
        if cpu is not None:
            m = cpu["sample"]
            parity &= (rows(0, m) == cpu["out"])
            for lo, cnt, blob in cpu["extra"]:       # blocks spread over the rest of the batch, the last one at its end
                parity &= (rows(lo, cnt) == blob)
            checked.append("first %d units and %d blocks of %d spread to the end of the batch against the C oracle" % (m, len(cpu["extra"]), SPREAD_LEN))
        else:
            cl = min(CHECK_LEN if wl["curve"] != "p384" else CHECK_LEN // 4, n // 2)
            for lo in (0, n - cl):
                s = d_s[lo:lo + cl].cpu().numpy()
                p = None if d_p is None else d_p[lo:lo + cl].cpu().numpy()
                parity &= (rows(lo, cl) == CO.lincomb_batch(wl["cid"], s, p, threads=threads).tobytes())
            checked.append("first and last %d units of every rank's slice against the C oracle" % cl)
    if dist is not None:
        t = torch.tensor([1 if parity else 0], dtype=torch.int32, device=(dev if backend == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        parity = bool(t.item())
    host_io = None
    if env.get("host_io") and not wl["msm"] and not wl.get("ecdsa") and not wl["fixed"] and schedule == "fast":
        host_io = measure_host_io(env, cv, d_s, d_p, d_o, d_i, n, wl)
        parity &= host_io["pinned"]["parity_ok"] and host_io["pageable"]["parity_ok"]
    del d_s, d_p, d_o, d_i
    torch.cuda.empty_cache()
    return {"name": name, "n": n, "log2n": log2n, "elapsed": elapsed, "kernel_ms": kernel_ms / steps, "steps": steps, "parity": bool(parity), "checked": checked,
            "value": world * n * steps / elapsed, "table": table, "host_io": host_io}


def measure_host_io(env, cv, d_s, d_p, d_o, d_i, n, wl, reps=3):
    """The same batch through the C ABI with HOST buffers (ECGPU_MEM_HOST: what a caller on the reference's side of the boundary hands
    over, k256/src/arithmetic/mul.rs:442-481 - values in host memory): upload, kernels and download, end to end, wall clock around
    the blocking call.  Page-locked buffers (ecgpu_host_alloc) and ordinary pageable numpy arrays (already touched: a caller's
    buffers are not fresh from calloc); outputs compared byte for byte with the device-resident result d_o / d_i.
    Never `value`: that is the device-resident rate."""
    import numpy as np
    ctx = env["ctx"]
    nb = cv.nb
    ctx.synchronize()
    want_o, want_i = d_o.cpu().numpy().tobytes(), d_i.cpu().numpy().tobytes()
    hs, hp = d_s.cpu().numpy(), d_p.cpu().numpy()
    unit_bytes = wl["bytes_per_unit"]
    res = {"units": n, "bytes_per_unit_over_pcie": unit_bytes, "chunks": env["ecgpu"].host_chunk_schedule(n, (1 << 23) if wl["curve"] == "k256" else (1 << 22)),     # one pass: 32 (k256) / 16 results per resident lane
           "note": "PCIe-inclusive: wall clock of one blocking ecgpu_mul_batch call with ECGPU_MEM_HOST; best and median of %d calls; never `value`" % reps}
    out, inf = np.ones((n, 2 * nb), dtype=np.uint8), np.ones(n, dtype=np.uint8)
    bufs = {"pageable": (hs, hp, out, inf)}
    ps, pp = ctx.pinned_array((n, nb)), ctx.pinned_array((n, 2 * nb))
    po, pi = ctx.pinned_array((n, 2 * nb)), ctx.pinned_array((n,))
    ps[:] = hs
    pp[:] = hp
    bufs["pinned"] = (ps, pp, po, pi)
    for kind in ("pinned", "pageable"):
        a, b, o, i = bufs[kind]
        times = []
        for _ in range(reps):
            o[:1] = 0xA5
            t0 = time.perf_counter()
            cv.mul(a, b, out=o, out_inf=i)
            times.append(time.perf_counter() - t0)
        ok = (o.tobytes() == want_o) and (i.tobytes() == want_i)
        best, med = min(times), median(times)
        res[kind] = {"rate": n / med, "rate_best": n / best, "unit": wl["unit"], "ms": med * 1e3, "ms_best": best * 1e3,
                     "pcie_gb_per_s": n * unit_bytes / med / 1e9, "parity_ok": bool(ok)}
    return res


def roofline_for(name, res, schedule, peak_meas, pair_meas, world=1):
    wl = WORKLOADS[name]
    key = name if name != "k256_varbase" else "k256_varbase_" + schedule
    m_cnt, s_cnt = WORK[key]
    modmul = m_cnt + s_cnt
    mac_conv = modmul * MAC_CONV[wl["curve"]]
    iss_m, iss_s = MAC_ISSUED[wl["curve"]]
    mac_issued = m_cnt * iss_m + s_cnt * iss_s - 8 * FUSED_PAIRS.get(key, 0) + 11 * FUSED_DBL.get(key, 0)
    kernel_s = res["kernel_ms"] / 1e3
    n = res["n"]
    achieved = n * mac_conv / kernel_s / 1e12
    issued = n * mac_issued / kernel_s / 1e12
    pmc, pmc_file = pmc_summary(name, res["log2n"], wl["log2n"]) if (schedule == "fast" and world == 1) else ({}, None)
    ctr = pmc.get("counters_per_launch", {})
    # SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU to the last digit on this stack (it counts instructions, not busy cycles),
    # so instructions x 4 / SIMD cycles is the share of 4-cycle ISSUE SLOTS the VALU instructions would fill - not a busy
    # measurement (moves and plain adds issue faster than one per 4 cycles: the MSM reads 105 %).  What the counters do
    # say about stalls is SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, the share of wave-cycles spent waiting to issue.
    issue_slots = None
    if "SQ_INSTS_VALU" in ctr and "GRBM_GUI_ACTIVE" in ctr:
        issue_slots = 100.0 * ctr["SQ_INSTS_VALU"] * 4 / 1024 / (ctr["GRBM_GUI_ACTIVE"] / 8)
    valu_cpi = (ctr["GRBM_GUI_ACTIVE"] / 8 * 1024 / ctr["SQ_INSTS_VALU"]) if ("SQ_INSTS_VALU" in ctr and "GRBM_GUI_ACTIVE" in ctr) else None
    stall = (ctr["SQ_WAIT_INST_ANY"] / ctr["SQ_WAVE_CYCLES"]) if ("SQ_WAIT_INST_ANY" in ctr and ctr.get("SQ_WAVE_CYCLES")) else None
    pmc_obj = None
    if pmc:
        kt = pmc.get("kernel_trace") or {}
        pmc_obj = {"source": pmc_file, "profiled_at_commit": pmc.get("commit"), "profiled_csrc_sha16": pmc.get("csrc_sha16"),
                   "csrc_sha16_of_this_tree": csrc_sha16(name), "kernel_sources_unchanged_since_profile": pmc.get("csrc_sha16") == csrc_sha16(name),
                   "kernel_match": pmc.get("kernel_match"), "profiled_kernel_ms": (kt.get("avg_ns") / 1e6 if kt.get("avg_ns") else None),
                   "hbm_bytes_per_launch": pmc.get("hbm_bytes_per_launch"), "traffic_over_algorithmic": (pmc["hbm_bytes_per_launch"] / (n * wl["bytes_per_unit"])
                                                                                                      if pmc.get("hbm_bytes_per_launch") else None),
                   # the same counters under the guide's rule for wide streaming reads (FETCH_SIZE x 2): the upper reading.  hbm_bytes_per_launch
                   # follows the calibration on this kernel's known read volume (fetch_calibration: "raw" for lane-divergent 64-byte blocks)
                   "hbm_bytes_per_launch_x2_streaming_rule": pmc.get("hbm_bytes_per_launch_x2_streaming_rule"),
                   "traffic_over_algorithmic_x2_streaming_rule": (pmc["hbm_bytes_per_launch_x2_streaming_rule"] / (n * wl["bytes_per_unit"])
                                                                  if pmc.get("hbm_bytes_per_launch_x2_streaming_rule") else None),
                   "fetch_calibration": pmc.get("fetch_calibration"),
                   "l2_hit_rate": pmc.get("l2_hit_rate"), "valu_insts_per_launch": ctr.get("SQ_INSTS_VALU"),
                   "valu_issue_slots_pct": issue_slots, "valu_cycles_per_inst": valu_cpi, "issue_stall_frac": stall,
                   # the clock the chip held inside the profiled kernel (GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's own duration): the kernels
                   # that gather from beyond L2 hold ~2.1 GHz, the register-resident ones ~2.36 (profiles/r03_clock_probe.txt)
                   "effective_clock_ghz": pmc.get("effective_clock_ghz"),
                   "note": "read from the committed summary, not measured by this run"}
    alg_bytes = n * wl["bytes_per_unit"]
    r = {
        "bound": "valu", "achieved": achieved, "peak": PEAK_TMACS, "unit": "TMAC/s (32x32+64 v_mad_u64_u32)", "frac": achieved / PEAK_TMACS,
        "traffic": pmc.get("hbm_bytes_per_launch"),
        "kernel": ("lincomb_ref_kernel<CurveK256,1>" if (name == "k256_varbase" and schedule == "ref") else wl["kernel"]), "kernel_ms": res["kernel_ms"],
        "modmul_per_unit": round(modmul, 1), "mac_per_unit": round(mac_conv), "mac_issued_per_unit": round(mac_issued),
        "achieved_issued": issued, "frac_issued": issued / PEAK_TMACS,
        "peak_measured": peak_meas, "frac_of_peak_measured": (achieved / peak_meas if peak_meas else None),
        "mac_pair_peak_measured": pair_meas, "frac_of_mac_pair_peak": (issued / pair_meas if pair_meas else None),
        "pmc": pmc_obj,
        "hbm": {"bound": "hbm", "achieved": alg_bytes / kernel_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": alg_bytes / kernel_s / 1e9 / PEAK_HBM_GBS, "bytes_per_unit": wl["bytes_per_unit"]},
    }
    return r


def run_inproc(args):
    """--inproc: all N GPUs from ONE process through the library's device group (include/ecgpu.h "device groups", csrc/group.hip) -
    what a caller on the reference's side of the boundary gets from one call: per-device contexts, host threads and streams,
    contiguous index ranges, no collective for the independent batches; per-device bucket method, RCCL all-gather of one point per
    device and a fold on the first device for the split sum.  Inputs resident in each device's HBM before the timed region, as in the
    one-process-per-GPU path; the same metric, the same parity rules (every member's first and last units against the C oracle,
    the split sum over ALL terms against the closed form).  --same-device puts all members on device 0 (rehearsal on a one-GPU box)."""
    import numpy as np
    import torch
    import ecgpu
    from oracle import coracle as CO
    from oracle import ecmodel as M
    wl = WORKLOADS[args.workload]
    if wl.get("ecdsa"):
        raise SystemExit("--inproc runs the BASELINE configs (variable base, fixed base, MSM)")
    k = args.gpus
    n = 1 << (args.log2n or wl["log2n"])
    devices = [0] * k if args.same_device else list(range(k))
    g = ecgpu.Group(devices)
    nb = ecgpu.FIELD_BYTES[wl["cid"]]
    d_s, d_p, d_o, d_i = [], [], [], []
    for i, dv in enumerate(devices):
        dev = torch.device("cuda", dv)
        cv = g.context(i).curve(wl["curve"])
        first = i * n
        s = torch.empty((n, nb), dtype=torch.uint8, device=dev)
        cv.synth_scalars_device(s, n, SEED, first)
        p = None
        if wl["msm"]:
            p = torch.empty((n, 2 * nb), dtype=torch.uint8, device=dev)
            v = torch.from_numpy(structured_point_scalars(first, n)).to(dev)
            cv.mul_device(v, None, p, n)
            g.context(i).synchronize()
            del v
        elif not wl["fixed"]:
            p = torch.empty((n, 2 * nb), dtype=torch.uint8, device=dev)
            cv.synth_points_device(p, n, SEED, first)
        d_s.append(s); d_p.append(p)
        d_o.append(torch.empty((n, 2 * nb), dtype=torch.uint8, device=dev))
        d_i.append(torch.empty((n,), dtype=torch.uint8, device=dev))
    g.synchronize()
    counts = [n] * k
    result = [None]

    def step():
        if wl["msm"]:
            result[0] = g.msm_sharded(wl["curve"], d_s, d_p, counts)
        else:
            g.lincomb_sharded(wl["curve"], d_s, None if wl["fixed"] else d_p, d_o, counts, d_out_inf=d_i)

    for _ in range(max(1, args.warmup)):             # the first call builds per-device tables / workspaces (and the RCCL communicator)
        step()
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    g.synchronize()
    elapsed = time.perf_counter() - t0
    parity, checked = True, []
    if wl["msm"]:
        t_total = 0
        for i in range(k):
            t_total += msm_expected_scalar(CO.synth_scalars(0, n, SEED, i * n), i * n, M.K256.n)
        want = M.affine_mul(M.K256, t_total % M.K256.n, (M.K256.gx, M.K256.gy))
        parity &= bytes(result[0]) == (bytes(64) if want is None else M.i2b(M.K256, want[0]) + M.i2b(M.K256, want[1]))
        checked.append("sum over all %d terms equals (sum k_i (a0 + i d) mod n) G for the structured points P_i = (a0 + i d) G" % (k * n))
    else:
        cl = min(CHECK_LEN if wl["curve"] != "p384" else CHECK_LEN // 4, n // 2)
        for i in range(k):
            for lo in (0, n - cl):
                s = d_s[i][lo:lo + cl].cpu().numpy()
                p = None if d_p[i] is None else d_p[i][lo:lo + cl].cpu().numpy()
                got = torch.cat([d_o[i][lo:lo + cl], d_i[i][lo:lo + cl, None]], dim=1).cpu().numpy().tobytes()
                parity &= (got == CO.lincomb_batch(wl["cid"], s, p, threads=8).tobytes())
        checked.append("first and last %d units of every member's range against the C oracle" % cl)
    line = {"metric": wl["metric"], "value": k * n * args.steps / elapsed, "unit": wl["unit"], "n_gpus": k, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl["desc"] % (args.log2n or wl["log2n"]), "units_per_gpu_per_step": n,
                       "parallelism": ("in-process device group over devices %s (ecgpu_group_*): " % devices) +
                                      ("per-device bucket method, gather of one projective point per device (%s), fold on device %d" % (g.gather_path(), devices[0])
                                       if wl["msm"] else "contiguous index ranges, one host thread + context + stream per device, no collective")},
            "parity_ok": bool(parity), "parity_checked": checked}
    g.close()
    print(json.dumps(line), flush=True)
    if not parity:
        raise SystemExit("PARITY FAILURE: GPU output differs from the CPU oracle / closed form")


def cpu_plan(wl, n, cpu_sample, procs):
    per_proc = {"k256": 61440, "p256": 7680, "p384": 3840}[wl["curve"]]
    if wl.get("ecdsa"):
        per_proc //= 2                  # a verification is two scalar multiplications on the CPU
    return cpu_sample or min(n, per_proc * procs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="k256_varbase",
                    help="BASELINE.json config to run (default: configs[1], the headline metric)")
    ap.add_argument("--log2n", type=int, default=0, help="units per GPU per step (0 = the config's size)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="units for the CPU baseline (0 = auto, about 10-20 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="processes of the all-core CPU baseline (0 = all host cores, at most 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs 3, 4, 5 after the headline")
    ap.add_argument("--no-host-io", action="store_true", help="skip the host-buffer (PCIe-inclusive) runs of the variable-base workloads at N = 1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse several ranks on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: initialise a world-size-1 process group all the same, so that the split-MSM step runs its collective branch "
                         "(RCCL all_gather_into_tensor on device tensors + the device fold) on a one-GPU box")
    ap.add_argument("--schedule", choices=["fast", "ref"], default="fast",
                    help="fast = throughput schedule (affine result specified); ref = reference-faithful schedule (exact XYZ)")
    ap.add_argument("--inproc", action="store_true",
                    help="drive all --gpus N devices from THIS process through the library's device group (ecgpu_group_*) instead of one process per GPU")
    ap.add_argument("--same-device", action="store_true", help="--inproc: put all group members on device 0 (rehearsal on a one-GPU box)")
    args = ap.parse_args()
    if args.inproc:
        return run_inproc(args)
    maybe_launch_ranks(args)
    # Rank 0 prints ONE JSON line on stdout.  Native libraries print there too (RCCL's version banner, gloo's connection
    # lines), so file descriptor 1 points at stderr for the duration of the run and is restored for the line itself.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    wl = WORKLOADS[args.workload]
    if not args.log2n:
        args.log2n = wl["log2n"]
    n = 1 << args.log2n
    others = list(OTHER_CONFIGS) if (world == 1 and args.workload == "k256_varbase" and args.schedule == "fast" and args.log2n == wl["log2n"]
                                     and not args.no_other_configs) else []

    # ---- CPU baseline first (rank 0, N = 1 only), before this process initialises the GPU ----------
    cpu, cpu_others = None, {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        procs = args.cpu_threads or max(1, min(os.cpu_count() or 1, 16))
        kind = "ecdsa" if wl.get("ecdsa") else ("msm" if wl["msm"] else "mul")
        sample = cpu_plan(wl, n, args.cpu_sample, procs)
        cpu = run_cpu_baseline(sample, procs, wl["cid"], wl["fixed"], kind, n)
        cpu["sample"] = sample
        for o in others:            # small untimed-grade samples: parity of the other configs (and a rough rate)
            w2 = WORKLOADS[o]
            s2 = min(1 << w2["log2n"], {"k256": 16384, "p256": 2048, "p384": 1024}[w2["curve"]] * procs)
            cpu_others[o] = run_cpu_baseline(s2, procs, w2["cid"], w2["fixed"], "msm" if w2["msm"] else "mul", 1 << w2["log2n"], reps=5, single=True)
            cpu_others[o]["sample"] = s2

    import torch
    import ecgpu

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if args.backend == "nccl":
            raise SystemExit("rank %d has no GPU of its own (%d visible): one process per GPU" % (local_rank, ndev))
        local_rank %= ndev                  # gloo rehearsal: ranks share the card
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world > 1:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        else:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            dist.init_process_group(backend=args.backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)

    ctx = ecgpu.Context(local_rank)
    # One explicit stream for everything (RCCL orders a collective against torch's CURRENT stream, the library launches on
    # the stream it was given): all device work of this script runs under `with torch.cuda.stream(stream)`.
    stream = torch.cuda.Stream(device=local_rank)
    ctx.set_stream(stream.cuda_stream)
    env = {"torch": torch, "ecgpu": ecgpu, "ctx": ctx, "dev": torch.device("cuda", local_rank), "rank": rank, "world": world, "dist": dist,
           "backend": args.backend, "host_io": (world == 1 and rank == 0 and not args.no_host_io and args.log2n >= 21)}

    with torch.cuda.stream(stream):
        res = run_workload(env, args.workload, args.log2n, args.steps, args.warmup, args.schedule, cpu)
        other_res = []
        for o in others:
            other_res.append(run_workload(env, o, WORKLOADS[o]["log2n"], max(2, min(args.steps, 5)), 1, "fast", cpu_others.get(o)))
    torch.cuda.synchronize()
    peak_meas, pair_meas = measure_peak(local_rank) if rank == 0 else (None, None)

    ok = res["parity"] and all(r["parity"] for r in other_res)
    if rank == 0:
        line = {
            "metric": wl["metric"],
            "value": res["value"],
            "unit": wl["unit"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": wl["desc"] % args.log2n,
                       "units_per_gpu_per_step": n,
                       "parallelism": (("one sum split over %d GPU(s): per-rank bucket method, all-gather of one projective point per rank%s, fold on the device" % (
                                            world, (" (%s, world %d)" % (args.backend, world)) if dist is not None else " (no process group at N = 1)"))
                                       if wl["msm"] else ("independent batches, %d GPU(s), no collective" % world)),
                       "schedule": ("reference-faithful (GLV + signed radix-16, RCB complete formulas, constant-time table scans, per-point inversion)" if args.schedule == "ref"
                                    else "throughput (GLV + signed 5-bit windows, Jacobian, common-Z table of 16 entries, batched inversion)")
                       if args.workload == "k256_varbase" else "throughput schedule of this workload (DESIGN.md section 4)"},
            "parity_ok": res["parity"],
            "parity_checked": res["checked"],
            "roofline": roofline_for(args.workload, res, args.schedule, peak_meas, pair_meas, world),
        }
        if res.get("table"):
            line["generator_table"] = res["table"]
        if res.get("host_io"):
            line["host_io"] = res["host_io"]
        if cpu is not None:
            line["cpu_baseline"] = {
                "value": cpu["rate"], "unit": wl["unit"], "cores": cpu["procs"], "host_nproc": os.cpu_count(), "kind": "port",
                "sample": "first %d units of the same seeded batch, C restatement of the reference path (oracle/ecoracle.c), %d single-threaded processes, each slice timed in %d chunks: median chunk rate" % (
                    cpu["sample"], cpu["procs"], cpu["reps"]),
                "single_thread": {"value": cpu["single_rate"], "cores": 1, "median_of": cpu["reps"]},
                "wall_s": cpu["wall_s"], "parity_ok": res["parity"],
            }
        if other_res:
            line["other_configs"] = []
            for r in other_res:
                w2 = WORKLOADS[r["name"]]
                ent = {"metric": w2["metric"], "value": r["value"], "unit": w2["unit"], "config": w2["desc"] % r["log2n"], "steps": r["steps"],
                       "ms_per_step": r["elapsed"] / r["steps"] * 1e3, "parity_ok": r["parity"], "parity_checked": r["checked"],
                       "roofline": roofline_for(r["name"], r, "fast", peak_meas, pair_meas)}
                c2 = cpu_others.get(r["name"])
                if c2 is not None:
                    ent["cpu_baseline"] = {"value": c2["rate"], "unit": w2["unit"], "cores": c2["procs"], "host_nproc": os.cpu_count(), "kind": "port",
                                           "sample": "first %d units of the same seeded batch, %d single-threaded processes, each slice timed in %d chunks: median chunk rate" % (
                                               c2["sample"], c2["procs"], c2["reps"]),
                                           "single_thread": {"value": c2["single_rate"], "cores": 1, "median_of": c2["reps"]}, "wall_s": c2["wall_s"]}
                if r.get("table"):
                    ent["generator_table"] = r["table"]
                if r.get("host_io"):
                    ent["host_io"] = r["host_io"]
                line["other_configs"].append(ent)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()
    if not ok:
        raise SystemExit("PARITY FAILURE: GPU output differs from the CPU oracle / closed form (see parity_ok in the line above)")


if __name__ == "__main__":
    main()
