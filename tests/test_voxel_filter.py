"""Next row N3: voxel-grid down-sampling, Homework1 voxel_filter.py:17-52 (centroid mode).
CPU: the oracle against the golden produced by the reference function itself; GPU: the HIP path against both."""
import numpy as np
import pytest

CASES = [("kitti", 0.5), ("kitti", 2.0), ("synth", 0.5), ("synth", 2.0)]


def tag(name, leaf):
    return f"{name}_leaf{str(leaf).replace('.', 'p')}"


@pytest.mark.parametrize("name,leaf", CASES)
def test_oracle_voxel_filter_matches_reference_output(orc, golden, name, leaf):
    g = golden("voxel_filter_hw1.npz")
    want = g[f"out_{tag(name, leaf)}"]
    got = orc.voxel_filter_f32(np.ascontiguousarray(g[f"in_{name}"].T), leaf)
    assert got.shape == (3, want.shape[0])
    assert np.array_equal(got.T.astype(np.float64), want)          # the reference stores f32-valued centroids as f64


def test_oracle_voxel_filter_drops_last_voxel_and_handles_tiny_inputs(orc):
    # two voxels -> only the first is emitted (voxel_filter.py:41-50 flushes on change only); one voxel -> nothing
    pts = np.array([[0.1, 0.2, 5.1, 5.2], [0, 0, 0, 0], [0, 0, 0, 0]], np.float32)
    out = orc.voxel_filter_f32(pts, 1.0)
    assert out.shape == (3, 1) and np.allclose(out[:, 0], [0.15, 0, 0], atol=1e-6)
    assert orc.voxel_filter_f32(pts[:, :2].copy(), 1.0).shape == (3, 0)
    assert orc.voxel_filter_f32(np.zeros((3, 0), np.float32), 1.0).shape == (3, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,leaf", CASES)
def test_gpu_voxel_filter_matches_reference_output(pcr, golden, name, leaf):
    g = golden("voxel_filter_hw1.npz")
    want = g[f"out_{tag(name, leaf)}"]
    ctx = pcr.Context(0)
    try:
        c = ctx.cloud(np.ascontiguousarray(g[f"in_{name}"].T))
        f = ctx.voxel_filter(c, leaf)
        got = f.numpy()
        assert got.shape == (3, want.shape[0])
        assert np.array_equal(got.T.astype(np.float64), want)
        # the filtered cloud is a first-class device cloud: it can serve as an ICP target right away
        idx, d2 = ctx.nn1(f, f)
        assert (d2 == 0).all()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_gpu_voxel_filter_full_scan_and_edges(pcr, orc, synth):
    ctx = pcr.Context(0)
    try:
        scan = synth.kitti_like_scan(120000)
        c = ctx.cloud(scan)
        for leaf in (0.3, 1.0):                                      # 0.3 = hw9's readBinaryAndVoxelDown leaf (main.cpp:30)
            got = ctx.voxel_filter(c, leaf).numpy()
            want = orc.voxel_filter_f32(scan, leaf)
            assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
            assert 1000 < got.shape[1] < 120000
        pts = np.array([[0.1, 0.2, 5.1, 5.2], [0, 0, 0, 0], [0, 0, 0, 0]], np.float32)
        assert ctx.voxel_filter(ctx.cloud(pts), 1.0).numpy().shape == (3, 1)
        assert ctx.voxel_filter(ctx.cloud(pts[:, :2].copy()), 1.0).numpy().shape == (3, 0)
        assert ctx.voxel_filter(ctx.cloud(np.zeros((3, 0), np.float32)), 1.0).numpy().shape == (3, 0)
        with pytest.raises(pcr.PcrError):
            ctx.voxel_filter(c, 0.0)
    finally:
        ctx.close()
