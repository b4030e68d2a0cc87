"""The *_device entry points are eager launches (include/pcdhip.h, Conventions): a stream that is capturing a HIP graph
is refused with PCD_ERR_UNSUPPORTED before anything is allocated, synchronised or launched -- round 3's replay of a
step captured with torch.cuda.graph faulted because the captured kernels kept scratch addresses that later eager calls
had freed.  After the refused capture the handles work as before."""
import numpy as np
import pytest

from pcdhip import synth

pytestmark = pytest.mark.gpu


def test_device_entry_points_refuse_a_capturing_stream(gpu):
    import torch
    xyz, nrm = synth.cloud_planes(200000, seed=3, patches=16)
    Q = 100000                                   # > 65536: the grid path (scratch, bookkeeping, brick + fallback kernels)
    q = synth.queries(xyz, Q, seed=4)
    c = gpu.Cloud(xyz, nrm, raw_lidar_frame=False)
    scene = synth.ba_scene(6, 300, seed=5)
    ba = gpu.BA(**scene)
    dq = torch.from_numpy(q).cuda()
    mr = torch.full((Q,), 1.5, dtype=torch.float64, device="cuda")
    keys = torch.empty(Q, dtype=torch.int64, device="cuda")
    out = dict(type=torch.empty(Q, dtype=torch.uint8, device="cuda"), abcd=torch.empty((Q, 4), dtype=torch.float64, device="cuda"))
    cost = torch.zeros(1, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s = side.cuda_stream
        c.nn_device(dq, Q, keys, stream=s)                                   # eager on the side stream: fine
        side.synchronize()
        ref = keys.clone()
        g = torch.cuda.CUDAGraph()
        g.capture_begin(capture_error_mode="relaxed")
        try:
            calls = [lambda: c.nn_device(dq, Q, keys, stream=s),
                     lambda: c.nn_refine_device(dq, Q, keys, stream=s),
                     lambda: c.associate_device(dq, Q, mr, Q, gpu.GATE_MAPPER_LOCAL, out, stream=s),
                     lambda: ba.evaluate_device(dict(cost=cost), stream=s)]
            for call in calls:
                with pytest.raises(gpu.PcdError) as e:
                    call()
                assert e.value.status == gpu.PCD_ERR_UNSUPPORTED, str(e.value)
                assert "capturing" in str(e.value)
        finally:
            g.capture_end()
        # the refused capture left no trace: the same eager call gives the same keys
        keys.zero_()
        c.nn_device(dq, Q, keys, stream=s)
        side.synchronize()
        assert torch.equal(keys, ref)
        ba.evaluate_device(dict(cost=cost), stream=s)
        side.synchronize()
        assert np.isfinite(cost.item()) and cost.item() > 0
    c.close()
    ba.close()
