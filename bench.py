#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): k256 variable-base scalar multiplications per second.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE).  A "step" is one pass
of the hot path over one batch: 2^24 independent (scalar, point) pairs per GPU (BASELINE.json
configs[1]), inputs generated on the device from the synthetic-input spec (oracle/synth.py) and
resident in HBM before the timed region.  Independent batches shard across GPUs with no
data-path collective (weak scaling: per-GPU work is fixed); the only communication is the
barrier and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  roofline     the dominant kernel against its bound.  This path is integer-VALU bound (no MFMA,
               ~0.2 % of HBM): `bound` is "valu", achieved/peak are 32x32-bit multiply-accumulates
               per second (v_mad_u64_u32 issue rate); the HBM view of the same kernel is under
               roofline["hbm"] (algorithmic bytes, GB/s, fraction of 8 TB/s, PMC traffic if a
               profiles/ summary is present).
  cpu_baseline the C restatement of the reference CPU path (oracle/ecoracle.c, "port": rustc is not
               available) timed on this box's host cores on a bounded sample of the same workload,
               and used to check the GPU output of that sample byte for byte.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))

SEED = 0xEC5CA1A5

# ---- algorithmic work per unit (DESIGN.md section "Kernels and rooflines") ---------------------
# unit = one k256 variable-base scalar multiplication with affine output.
# 32x32-bit multiply-accumulates of the schoolbook products, 64 per 256-bit modular multiplication
# or squaring; reduction and additions are not counted.
MODMUL_PER_UNIT = {
    # reference schedule: 128 doublings (6M+2S) + 80 additions (12M) + to_affine (255S + 17M)
    "k256_varbase_ref": 128 * 8 + 80 * 12 + 272,
    # throughput schedule (csrc/mulfast_k256.hpp): Jacobian, common-Z table, batched inversion (batch 32)
    #   table   : 1 dbl (3M+4S) + 6 mixed adds (8M+3S) + rescale 7M + 7x(3M+1S) + 8 beta*x  =  87M + 29S
    #   loop    : 128 dbl (3M+4S) + 66 mixed adds (8M+3S)                                   = 912M + 710S
    #   output  : 2M (global Z) + batched normalise 6M+1S + (255S+15M)/32                   =   9M +   9S
    "k256_varbase_fast": (87 + 912 + 9) + (29 + 710 + 9),
}
MAC_PER_MODMUL = 64
BYTES_PER_UNIT = 32 + 64 + 65          # scalar + affine point in, x||y||inf out

# Other BASELINE.json configs, selectable with --workload (the driver's default run is configs[1]).
#   modmul: field multiplications/squarings per unit of the schedule that runs
#   mac   : 32x32 multiply-accumulates of one schoolbook product (64 for 256-bit, 144 for 384-bit)
WORKLOADS = {
    "k256_varbase":   dict(curve="k256", cid=0, log2n=24, fixed=False, msm=False, metric="k256 variable-base scalar-muls/sec", unit="scalar-muls/s",
                           modmul=None, mac=64, bytes_per_unit=32 + 64 + 65, kernel=None,
                           desc="k256 variable-base scalar multiplication, 2^%d independent (scalar, point) pairs per GPU, affine output"),
    "p256_fixedbase": dict(curve="p256", cid=1, log2n=24, fixed=True, msm=False, metric="p256 fixed-base (mul_by_generator) scalar-muls/sec", unit="scalar-muls/s",
                           # batches >= 2^21: 13 signed 20-bit windows -> 13 mixed additions (8M+3S) + batched normalise 6M+1S + (256S+128M)/64
                           modmul=13 * 11 + 7 + 6, mac=64, bytes_per_unit=32 + 65, kernel="fb::mul_wide_kernel<CurveP256,20,64,4>",
                           desc="p256 mul_by_generator, 2^%d independent scalars per GPU, affine output"),
    "p384_varbase":   dict(curve="p384", cid=2, log2n=22, fixed=False, msm=False, metric="p384 variable-base scalar-muls/sec", unit="scalar-muls/s",
                           # 96 windows x (4 doublings (4M+4S) + 15/16 mixed additions (8M+3S)) + table (4 dbl + 3 add = 80) + its share of the
                           # affine conversion (8 x 7 + 430 / 8) + output normalise 7 + 430 / 8
                           modmul=96 * 32 + 90 * 11 + 80 + 110 + 61, mac=144, bytes_per_unit=48 + 96 + 97, kernel="vb::mul_kernel<CurveP384,8,4>",
                           desc="p384 variable-base scalar multiplication, 2^%d independent (scalar, point) pairs per GPU, affine output"),
    "k256_msm":       dict(curve="k256", cid=0, log2n=23, fixed=False, msm=True, metric="k256 MSM points/sec", unit="points/s",
                           # 16 signed 16-bit windows: one mixed addition (8M+3S) per term per window; bucket reduction amortised
                           modmul=16 * 11, mac=64, bytes_per_unit=32 + 64, kernel="msm::bucket_sum_kernel (+ digits/scan/scatter/reduce)",
                           desc="k256 multi-scalar multiplication, 2^%d terms per GPU (one sum; ranks exchange one point each), affine output"),
    # ECDSA verification: prep (scalar field) -> u1 G (16-bit fixed-base table) -> u2 Q (the headline
    # kernel) -> inversion-free check.  Field modmuls: 1756 + 156 + 7; scalar-field work (27 dense Montgomery products of
    # 136 MACs per signature) is folded in as 57 modmul equivalents.
    "k256_ecdsa_verify": dict(curve="k256", cid=0, log2n=22, fixed=False, msm=False, ecdsa=True, metric="k256 ECDSA verifications/sec", unit="verifications/s",
                           modmul=1756 + 156 + 7 + 57, mac=64, bytes_per_unit=32 + 64 + 64 + 1, kernel="verify_prep + fb::mul_wide_kernel + k256_mul_fast_kernel<32,4> + verify_check",
                           desc="k256 ECDSA verify_prehashed (low-s rule), 2^%d independent (prehash, signature, public key) triples per GPU"),
    "p256_ecdsa_verify": dict(curve="p256", cid=1, log2n=22, fixed=False, msm=False, ecdsa=True, metric="p256 ECDSA verifications/sec", unit="verifications/s",
                           modmul=3160 + 156 + 7 + 57, mac=64, bytes_per_unit=32 + 64 + 64 + 1, kernel="verify_prep + fb::mul_wide_kernel + vb::mul_kernel<CurveP256,8,4> + verify_check",
                           desc="p256 ECDSA verify_prehashed, 2^%d independent (prehash, signature, public key) triples per GPU"),
}
# v_mad_u64_u32 issues at half the FP32-FMA rate on gfx950 (measured, tools/ubench/valu_rates.hip):
# 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz
PEAK_TMACS = 256 * 4 * 16 * 2.4e9 / 1e12
# one exact multiply-accumulate is the pair v_mad_u64_u32 + v_addc_co_u32: 9.54 cycles per wave per SIMD at the
# kernels' occupancy (profiles/r01_mac_mix_microbench.txt) -> the ceiling any column-accumulating schedule can reach
MAC_PAIR_PEAK_TMACS = 256 * 4 * 64 * 2.4e9 / 9.54 / 1e12
PEAK_HBM_GBS = 8000.0


ECDSA_CORRUPT_EVERY = 7     # every 7th signature of the synthetic ECDSA batch has one bit of s flipped (must be rejected)


def cpu_ecdsa_worker(args):
    """ECDSA workload in a forked child: inputs made with the C oracle (untimed), verification timed."""
    first, n, cid = args
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
    t0 = time.perf_counter()
    v = CO.ecdsa_verify_batch(cid, z, sig, q, low_s=low_s)
    dt = time.perf_counter() - t0
    return first, n, dt, sig.tobytes() + v.tobytes()


def cpu_baseline_worker(args):
    """Runs in a forked child BEFORE the parent touches the GPU: C oracle on a slice."""
    first, n, cid, fixed, msm = args
    if msm == "ecdsa":
        return cpu_ecdsa_worker((first, n, cid))
    from oracle import coracle as CO
    s = CO.synth_scalars(cid, n, SEED, first)
    p = None if fixed else CO.synth_points(cid, n, SEED, first)
    t0 = time.perf_counter()
    if msm:
        # the reference has no Pippenger: its large-N form is one reference multiplication per term plus an addition
        out = CO.lincomb_batch(cid, s, p, out_proj=True, threads=1)
    else:
        out = CO.lincomb_batch(cid, s, p, threads=1)
    dt = time.perf_counter() - t0
    return first, n, dt, out.tobytes()


SPREAD_BLOCKS, SPREAD_LEN = 16, 4096     # extra parity blocks spread over the batch (sampled indices + the tail)


def spread_blocks(n, sample):
    """Start indices of the extra checked blocks: evenly spaced behind the timed sample, the last one ending the batch."""
    if n <= sample + SPREAD_BLOCKS * SPREAD_LEN:
        return []
    span = n - sample - SPREAD_LEN
    return [sample + (span * j) // (SPREAD_BLOCKS - 1) for j in range(SPREAD_BLOCKS)]


def run_cpu_baseline(sample, procs, cid=0, fixed=False, msm=False, n=0):
    """Reference CPU path (C port) on `sample` units spread over `procs` single-threaded processes; for the
    element-wise workloads also SPREAD_BLOCKS blocks of SPREAD_LEN units across the rest of the batch (checked, not
    part of the timed sample)."""
    from concurrent.futures import ProcessPoolExecutor
    per = (sample + procs - 1) // procs
    jobs = [(i * per, min(per, sample - i * per), cid, fixed, msm) for i in range(procs) if i * per < sample]
    extra = [] if msm else [(st, SPREAD_LEN, cid, fixed, msm) for st in spread_blocks(n, sample)]
    if extra:
        with ProcessPoolExecutor(max_workers=procs) as ex:
            extra_res = [(r[0], r[1], r[3]) for r in ex.map(cpu_baseline_worker, extra)]
    else:
        extra_res = []
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=procs) as ex:
        res = list(ex.map(cpu_baseline_worker, jobs))
    wall = time.perf_counter() - t0
    busy = max(r[2] for r in res)      # slowest worker, excludes process start-up and input synthesis
    outs = b"".join(r[3] for r in sorted(res))
    return {"wall_s": wall, "busy_s": busy, "out": outs, "procs": len(jobs), "parts": [(r[0], r[1], r[3]) for r in sorted(res)],
            "extra": extra_res}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="k256_varbase",
                    help="BASELINE.json config to run (default: configs[1], the headline metric)")
    ap.add_argument("--log2n", type=int, default=0, help="units per GPU per step (0 = the config's size)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="units for the CPU baseline (0 = auto, about 10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo to rehearse several ranks on one GPU)")
    ap.add_argument("--schedule", choices=["fast", "ref"], default="fast",
                    help="fast = throughput schedule (affine result specified); ref = reference-faithful schedule (exact XYZ)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    wl = WORKLOADS[args.workload]
    if not args.log2n:
        args.log2n = wl["log2n"]
    n = 1 << args.log2n

    # ---- CPU baseline first (rank 0, N = 1 only), before this process initialises the GPU ----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        procs = max(1, min(os.cpu_count() or 1, 16))
        # the C port runs ~16 k units/s/core: 64 k units per process is ~4 s of work each (about a minute of CPU
        # time on 16 cores); --cpu-sample overrides
        per_proc = {"k256": 65536, "p256": 8192, "p384": 4096}[wl["curve"]] // (1 if not wl["fixed"] or wl["curve"] != "k256" else 1)
        sample = args.cpu_sample or min(n, per_proc * procs)
        if wl.get("ecdsa"):
            per_proc //= 2                  # a verification is two scalar multiplications on the CPU
            sample = args.cpu_sample or min(n, per_proc * procs)
        cpu = run_cpu_baseline(sample, procs, wl["cid"], wl["fixed"], "ecdsa" if wl.get("ecdsa") else wl["msm"], n)
        cpu["sample"] = sample

    import numpy as np
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
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    ctx = ecgpu.Context(local_rank)
    cv = ctx.curve(wl["curve"])
    nb = cv.nb
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    dev = torch.device("cuda", local_rank)
    d_s = torch.empty((n, nb), dtype=torch.uint8, device=dev)
    d_p = None if wl["fixed"] else torch.empty((n, 2 * nb), dtype=torch.uint8, device=dev)
    d_o = torch.empty((1 if wl["msm"] else n, 2 * nb), dtype=torch.uint8, device=dev)
    d_i = torch.empty((n,), dtype=torch.uint8, device=dev)
    first = rank * n                         # disjoint slices of one global batch
    cv.synth_scalars_device(d_s, n, SEED, first)
    if d_p is not None:
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
        cv.ecdsa_sign_device(d_s, d_k, d_z, d_sig, None, d_i, n)
        ctx.synchronize()
        bad = (torch.arange(first, first + n, device=dev) % ECDSA_CORRUPT_EVERY) == 0
        d_sig[bad, -1] ^= 1
    torch.cuda.synchronize()
    msm_result = {}

    def step():
        if wl.get("ecdsa"):
            cv.ecdsa_verify_device(d_z, d_sig, d_p, d_i, n)
        elif wl["msm"]:
            if dist is None:
                cv.msm_device(d_s, d_p, n, d_o)
            else:
                # every rank sums its slice; one projective point per rank is all-gathered and folded locally
                from ecgpu import parallel
                d_part = torch.empty((3 * nb,), dtype=torch.uint8, device=dev)

                def local_msm(lo, hi):
                    cv.msm_device(d_s, d_p, n, d_part, out_format=ecgpu.PROJECTIVE)
                    ctx.synchronize()
                    return d_part.cpu().numpy()

                tot = parallel.msm_sharded(local_msm, lambda a, b: cv.add(a.reshape(1, -1), b.reshape(1, -1))[0], world * n)
                msm_result["xyz"] = tot
        else:
            cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i, flags=(ecgpu.EXACT_REFERENCE if args.schedule == "ref" else 0))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()                        # HIP events on the launch stream bracket the same region
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=(dev if args.backend == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity of the sample against the CPU oracle (same seeded inputs) ----------------------------
    parity = None
    if cpu is not None:
        m = cpu["sample"]
        if wl.get("ecdsa"):
            # signatures made on the device and accept/reject flags, against the C oracle's for the same indices
            g_sig, g_ok = d_sig[:m].cpu().numpy(), d_i[:m].cpu().numpy()
            parity = True
            for lo, cnt, blob in cpu["parts"]:
                parity &= (g_sig[lo:lo + cnt].tobytes() == blob[:cnt * 2 * nb]) and (g_ok[lo:lo + cnt].tobytes() == blob[cnt * 2 * nb:])
            want_ok = (np.arange(m) % ECDSA_CORRUPT_EVERY != 0)
            parity &= bool((g_ok.astype(bool) == want_ok).all())
        elif wl["msm"]:
            # the oracle computed k_i * P_i for the first m terms (projective); fold them with the oracle's
            # complete addition and compare with the GPU MSM over the same m terms
            from oracle import coracle as CO
            parts = np.frombuffer(cpu["out"], dtype=np.uint8).reshape(m, 3 * nb)
            while parts.shape[0] > 1:
                if parts.shape[0] % 2:
                    parts = np.concatenate([parts, np.frombuffer(b"".join([bytes(nb), (1).to_bytes(nb, "big"), bytes(nb)]), dtype=np.uint8)[None, :]])
                parts = CO.point_op(wl["cid"], 0, parts[0::2].copy(), parts[1::2].copy())
            d_chk = torch.empty((3 * nb,), dtype=torch.uint8, device=dev)
            cv.msm_device(d_s, d_p, m, d_chk, out_format=ecgpu.PROJECTIVE)
            ctx.synchronize()
            g = d_chk.cpu().numpy().reshape(1, -1)
            # compare as group elements: normalise both through the library-independent route (x = X/Z)
            from oracle import ecmodel as M
            c_ = M.CURVES[wl["curve"]]

            def aff(b):
                X, Y, Z = (int.from_bytes(bytes(b[0][nb * t:nb * (t + 1)]), "big") for t in range(3))
                return M.to_affine(c_, (X, Y, Z))

            parity = aff(parts) == aff(g)
        else:
            got = torch.cat([d_o[:m], d_i[:m, None]], dim=1).cpu().numpy().tobytes()
            parity = (got == cpu["out"])
            for lo, cnt, blob in cpu["extra"]:       # blocks spread over the rest of the batch, the last one at its end
                parity &= (torch.cat([d_o[lo:lo + cnt], d_i[lo:lo + cnt, None]], dim=1).cpu().numpy().tobytes() == blob)
        if not parity:
            raise SystemExit("PARITY FAILURE: GPU output differs from the CPU oracle on the sampled units")

    if rank == 0:
        units = world * n * args.steps
        value = units / elapsed
        kernel_s = kernel_ms / 1e3 / args.steps            # average launch duration, HIP events
        modmul = wl["modmul"] or MODMUL_PER_UNIT["k256_varbase_" + args.schedule]
        macs_per_launch = n * modmul * wl["mac"]
        achieved_tmacs = macs_per_launch / kernel_s / 1e12
        alg_bytes = n * wl["bytes_per_unit"]
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and args.workload == "k256_varbase" and args.schedule == "fast" and args.log2n == 24:
            try:
                with open(pmc) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": wl["metric"],
            "value": value,
            "unit": wl["unit"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": wl["desc"] % args.log2n,
                       "units_per_gpu_per_step": n, "parallelism": (("one sum split over %d GPU(s): per-rank Pippenger, all-gather of one projective point per rank, local complete additions" % world)
                                       if wl["msm"] else ("independent batches, %d GPU(s), no collective" % world)),
                       "schedule": ("reference-faithful (GLV + signed radix-16, RCB complete formulas, per-point inversion)" if args.schedule == "ref"
                                    else "throughput (GLV + signed radix-16, Jacobian, common-Z table, batched inversion)")
                       if args.workload == "k256_varbase" else "throughput schedule of this workload (DESIGN.md section 4)"},
            "roofline": {
                "bound": "valu", "achieved": achieved_tmacs, "peak": PEAK_TMACS, "unit": "TMAC/s (32x32+64 v_mad_u64_u32)",
                "frac": achieved_tmacs / PEAK_TMACS, "traffic": traffic,
                "kernel": wl["kernel"] or ("lincomb_ref_kernel<CurveK256,1>" if args.schedule == "ref" else "k256_mul_fast_kernel<32,4>"), "kernel_ms": kernel_s * 1e3,
                "modmul_per_unit": modmul, "mac_per_unit": modmul * wl["mac"],
                "mac_pair_peak": MAC_PAIR_PEAK_TMACS, "frac_of_mac_pair_peak": achieved_tmacs / MAC_PAIR_PEAK_TMACS,
                "hbm": {"bound": "hbm", "achieved": alg_bytes / kernel_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": alg_bytes / kernel_s / 1e9 / PEAK_HBM_GBS, "bytes_per_unit": BYTES_PER_UNIT},
            },
        }
        if cpu is not None:
            line["cpu_baseline"] = {
                "value": cpu["sample"] / cpu["busy_s"], "unit": wl["unit"], "cores": cpu["procs"], "kind": "port",
                "sample": "first %d units of the same seeded batch, C restatement of the reference path (oracle/ecoracle.c), %d single-threaded processes; GPU output of the sample%s verified byte-identical" % (
                    cpu["sample"], cpu["procs"], (" and of %d more blocks of %d units spread to the end of the batch" % (len(cpu["extra"]), SPREAD_LEN)) if cpu.get("extra") else ""),
                "wall_s": cpu["wall_s"], "parity_ok": bool(parity),
            }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
