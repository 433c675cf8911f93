"""Stream handling of the "_dev" entry points: a caller on HIP's legacy default stream (handle 0, what
torch.cuda.current_stream().cuda_stream is by default) hands that handle to fs_set_stream and may then consume
the outputs from the default stream without any further synchronisation."""
import numpy as np
import pytest

from conftest import DeviceArrays

pytestmark = pytest.mark.gpu


def test_outputs_consumed_from_the_default_stream():
    from imagesegmentation_amd._lib import Context
    from imagesegmentation_amd.synth import make_batch
    H, W, n = 96, 128, 3
    d, c = make_batch([31, 32, 33], H, W)
    dev = DeviceArrays()
    try:
        with Context(H, W, max_batch=n) as ctx:
            ref = ctx.find_surfaces_batch(d, c)                       # host-pointer entry point, synchronous
            own = ctx.stream()
            ctx.set_stream(None)                                      # legacy default stream
            assert not ctx.stream()
            d_depth, d_rgb = dev.upload(d), dev.upload(c)
            outs = {"patches": (np.int64, ()), "depth_edges": (np.int32, ()), "surfaces": (np.int64, ()),
                    "angles": (np.float32, (2,))}
            ptr = {k: dev.alloc(n * H * W * int(np.prod(t, dtype=np.int64)) * np.dtype(dt).itemsize) for k, (dt, t) in outs.items()}
            ns = ctx.find_surfaces_batch_dev(n, d_depth, d_rgb, d_angles=ptr["angles"], d_patches=ptr["patches"],
                                             d_edges=ptr["depth_edges"], d_surfaces=ptr["surfaces"])
            # hipMemcpy is ordered after the default stream's work: no explicit synchronisation here
            for k, (dt, t) in outs.items():
                got = dev.download(ptr[k], np.empty((n, H, W) + t, dt))
                assert np.array_equal(got, ref[k]), k
            assert np.array_equal(ns, ref["num_surfaces"])
            ctx.use_own_stream()
            assert ctx.stream() == own
    finally:
        dev.free()
