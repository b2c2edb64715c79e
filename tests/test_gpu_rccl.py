"""The RCCL branch of the split MSM (ecgpu/parallel.py::allgather_into with backend "nccl" = RCCL on ROCm) on the hardware
a one-GPU box has: a world-size-1 NCCL process group.  `all_gather_into_tensor` runs on device tensors on torch's current
stream, the library folds the gathered points on the same stream, no host round trip in between - exactly the calls every
rank of the 8-GPU run makes (bench.py --workload k256_msm), with one rank.  No scaling number is claimed from this."""
import socket

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import synth

pytestmark = pytest.mark.gpu


def test_rccl_world1_allgather_and_device_fold():
    import torch
    import torch.distributed as dist
    import ecgpu
    from ecgpu import parallel
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        ctx = ecgpu.Context(0)
        st = torch.cuda.Stream()
        ctx.set_stream(st.cuda_stream)
        cv = ctx.curve("k256")
        n = (1 << 17) + 5
        s = CO.synth_scalars(0, n, synth.SEED, 4242)
        p = CO.synth_points(0, n, synth.SEED, 4242)
        with torch.cuda.stream(st):
            d_s, d_p = torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()
            d_part = torch.empty((96,), dtype=torch.uint8, device="cuda")
            d_all = torch.zeros((1, 96), dtype=torch.uint8, device="cuda")
            d_scr = torch.empty((1, 96), dtype=torch.uint8, device="cuda")
            d_out = torch.empty((64,), dtype=torch.uint8, device="cuda")
            d_inf = torch.empty((1,), dtype=torch.uint8, device="cuda")
            d_direct = torch.empty((64,), dtype=torch.uint8, device="cuda")
            for _ in range(3):                                   # no synchronisation between the three stages or the rounds
                cv.msm_device(d_s, d_p, n, d_part, out_format=ecgpu.PROJECTIVE)
                parallel.allgather_into(d_all, d_part, "nccl")
                parallel.fold_points_device(cv, d_all, 1, d_scr, d_out, d_inf)
            cv.msm_device(d_s, d_p, n, d_direct)
        st.synchronize()
        want = CO.msm_naive(0, s[:4096], p[:4096])               # the oracle on a prefix pins the path itself ...
        with torch.cuda.stream(st):
            d_chk = torch.empty((64,), dtype=torch.uint8, device="cuda")
            cv.msm_device(d_s[:4096], d_p[:4096], 4096, d_chk)
        st.synchronize()
        assert bytes(d_chk.cpu().numpy()) == bytes(want[:64])
        assert torch.equal(d_out, d_direct) and int(d_inf.cpu()[0]) == 0      # ... and the gathered-and-folded sum equals the direct one
        assert torch.equal(d_all[0], d_part)
        ctx.close()
    finally:
        dist.destroy_process_group()
