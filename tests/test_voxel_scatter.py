"""§8f-4 dynamic scatter-reduce: oracle self-checks on CPU; GPU kernels vs the oracle (parity unpinned: the reference
op is CUDA-only; see oracle/voxel_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import voxel_oracle as vo


def _cloud(n, seed, grid=(40, 40, 2), c=9, neg_frac=0.03, dup=True):
    rng = np.random.default_rng(seed)
    coors = np.stack([rng.integers(0, g, n) for g in grid], -1).astype(np.int32)
    coors[rng.random(n) < neg_frac, rng.integers(0, 3)] = -1          # out-of-range points
    feats = rng.normal(0, 1, (n, c)).astype(np.float32)
    if dup and n > 10:
        feats[n // 2] = feats[n // 3]                                     # ties for max
        coors[n // 2] = coors[n // 3]
    return coors, feats


def test_oracle_index_and_reduce_properties():
    coors, feats = _cloud(3000, 0)
    uniq, pmap, cnt = vo.scatter_index(coors)
    assert (uniq >= 0).all() and cnt.sum() == (pmap >= 0).sum() and (np.diff(uniq.view([('', uniq.dtype)] * 3).ravel().argsort()) == 1).all()
    assert ((coors < 0).any(-1) == (pmap < 0)).all()
    np.testing.assert_array_equal(uniq[pmap[pmap >= 0]], coors[pmap >= 0])
    for red in ('sum', 'mean', 'max'):
        out, _ = vo.scatter_reduce(feats, pmap, cnt, red)
        v = 17
        rows = feats[pmap == v].astype(np.float64)
        want = {'sum': rows.sum(0), 'mean': rows.mean(0), 'max': rows.max(0)}[red]
        np.testing.assert_allclose(out[v], want, rtol=1e-12)
        # backward == autograd of a torch restatement
        gv = np.random.default_rng(1).normal(0, 1, out.shape)
        g = vo.scatter_backward(gv, feats, pmap, cnt, red)
        tf = torch.from_numpy(feats).double().requires_grad_(True)
        idx = torch.from_numpy(pmap[pmap >= 0].astype(np.int64))
        src = tf[torch.from_numpy(np.nonzero(pmap >= 0)[0])]
        if red == 'max':
            tout = torch.full(out.shape, -np.inf, dtype=torch.float64).scatter_reduce(0, idx[:, None].expand(-1, feats.shape[1]), src, 'amax')
            # torch splits ties evenly; compare only where the max is unique
            continue
        tout = torch.zeros(out.shape, dtype=torch.float64).index_add(0, idx, src)
        if red == 'mean':
            tout = tout / torch.from_numpy(cnt).double()[:, None]
        (tout * torch.from_numpy(gv)).sum().backward()
        np.testing.assert_allclose(g, tf.grad.numpy(), rtol=1e-12, atol=1e-14)


def test_host_scatter_index_matches_oracle_cpu():
    """The ATen statement of the index (what the product runs on the DEVICE beyond the kernels' limits, and what the GPU
    test compares vox_index_build with) against the numpy oracle, on CPU tensors; the public entry has no CPU path."""
    from mmdet3d_gaussian_amd.scatter import _index_and_grouping_torch, group_points, scatter_index
    coors, _ = _cloud(2000, 3)
    with pytest.raises(RuntimeError):
        scatter_index(torch.from_numpy(coors))
    u, m, c = _index_and_grouping_torch(torch.from_numpy(coors))[:3]
    uo, mo, co = vo.scatter_index(coors)
    np.testing.assert_array_equal(u.numpy(), uo); np.testing.assert_array_equal(m.numpy(), mo); np.testing.assert_array_equal(c.numpy(), co)
    assert m.dtype == torch.int32 and c.dtype == torch.int32
    order, seg = group_points(m, c)
    o = order.numpy(); s = seg.numpy()
    assert s[0] == (mo < 0).sum() and s[-1] == len(mo)
    for v in (0, 5, len(co) - 1):
        pts = o[s[v]:s[v + 1]]
        assert (mo[pts] == v).all() and (np.diff(pts) > 0).all()          # grouped, ascending point id
    e = scatter_index(torch.zeros(0, 3, dtype=torch.int32))
    assert e[0].shape == (0, 3) and e[1].numel() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('n,ndim,dtype', [(1, 3, torch.int32), (255, 3, torch.int32), (70_001, 4, torch.int32), (2_000_000, 4, torch.int32),
                                          (5000, 4, torch.int64), (3000, 2, torch.int32), (4096, 8, torch.int32)])
def test_gpu_index_build_equals_the_aten_statement_and_the_oracle(n, ndim, dtype):
    """vox_index_build (round 3: one library call instead of ~20 ATen launches) against the ATen statement on the same
    device tensors — every output, bit for bit, incl. the grouping — and against the numpy oracle; dropped points
    (negative coordinates), an all-dropped batch, a single voxel and a batch column."""
    from mmdet3d_gaussian_amd.scatter import _index_and_grouping, _index_and_grouping_torch, scatter_index
    rng = np.random.default_rng(n + ndim)
    hi = [4, 6, 60, 70, 3, 3, 3, 3][:ndim] if ndim >= 4 else [40, 50, 30][:ndim]
    coors = np.stack([rng.integers(-1 if d == ndim - 1 else 0, h, n) for d, h in enumerate(hi)], -1)
    cases = [coors]
    if n >= 255:
        allneg = coors.copy(); allneg[:, 0] = -1
        one = np.zeros_like(coors); one[:, -1] = 7
        cases += [allneg, one]
    for cc in cases:
        t = torch.from_numpy(cc).to(dtype).cuda()
        got = _index_and_grouping(t)
        want = _index_and_grouping_torch(t)
        assert got[0].dtype == t.dtype and got[1].dtype == torch.int32 and got[2].dtype == torch.int32
        for a, b in zip(got[:3], want[:3]):
            assert torch.equal(a, b)
        assert torch.equal(got[3][0], want[3][0]) and torch.equal(got[3][1], want[3][1])
        if n <= 70_001:
            uo, mo, co = vo.scatter_index(cc)
            np.testing.assert_array_equal(got[0].cpu().numpy(), uo)
            np.testing.assert_array_equal(got[1].cpu().numpy(), mo)
            np.testing.assert_array_equal(got[2].cpu().numpy(), co)
    t0 = torch.from_numpy(coors).to(dtype).cuda()
    u, m, c = scatter_index(t0)
    w0 = _index_and_grouping_torch(t0)
    assert torch.equal(u, w0[0]) and torch.equal(m, w0[1]) and torch.equal(c, w0[2])


@pytest.mark.gpu
@pytest.mark.parametrize('n,c', [(1, 4), (257, 9), (5000, 10), (20000, 64), (3000, 130), (9000, 16), (7000, 128), (4000, 12), (60000, 3), (30000, 33),
                                 (5000, 127), (2000, 129), (1, 3)])
@pytest.mark.parametrize('red', ['sum', 'mean', 'max'])
def test_gpu_scatter_reduce_forward_backward(host_glue, n, c, red):
    import mmdet3d_gaussian_amd as amd  # noqa: F401
    from mmdet3d_gaussian_amd.scatter import Scatter
    coors, feats = _cloud(n, n + c, c=c)
    uo, mo, co = vo.scatter_index(coors)
    sc = Scatter(torch.from_numpy(coors).cuda())
    np.testing.assert_array_equal(sc.voxel_coors.cpu().numpy(), uo)
    np.testing.assert_array_equal(sc.pts_voxel_maps.cpu().numpy(), mo)
    np.testing.assert_array_equal(sc.voxel_pts_counts.cpu().numpy(), co)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    out, vc = sc.reduce(f, red)
    want, _ = vo.scatter_reduce(feats, mo, co, red)
    if len(co) == 0:
        assert out.shape[0] == 0
        return
    if red == 'max':
        np.testing.assert_array_equal(out.detach().cpu().numpy(), want.astype(np.float32))     # max is exact
    else:
        np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=2e-6, atol=2e-6)
    gv = np.random.default_rng(7).normal(0, 1, want.shape).astype(np.float32)
    out.backward(torch.from_numpy(gv).cuda())
    gw = vo.scatter_backward(gv, feats, mo, co, red)
    np.testing.assert_allclose(f.grad.cpu().numpy(), gw, rtol=2e-6, atol=1e-7)
    # mapback / reduce_mapback as the pillar encoders use them (voxel_encoders/utils.py:56)
    back = sc.reduce_mapback(f.detach(), red, default_feat=0)
    ref_back = np.where((mo >= 0)[:, None], want[np.clip(mo, 0, None)], 0)
    np.testing.assert_allclose(back.cpu().numpy(), ref_back, rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_gpu_scatter_batched_coors_and_determinism(host_glue):
    from mmdet3d_gaussian_amd.scatter import Scatter
    rng = np.random.default_rng(0)
    n = 8000
    coors3, feats = _cloud(n, 11, c=16)
    b = rng.integers(0, 3, n).astype(np.int32)
    coors4 = np.concatenate([b[:, None], coors3], 1)
    sc = Scatter(torch.from_numpy(coors4).cuda())
    assert sc.batch_size == 3 and sc.voxel_coors.shape[1] == 4
    f = torch.from_numpy(feats).cuda()
    o1, _ = sc.reduce(f, 'sum'); o2, _ = sc.reduce(f, 'sum')
    assert torch.equal(o1, o2)                                     # fixed summation order
    # per-sample voxels: every voxel row equals the sum over its batch-restricted points
    vc = sc.voxel_coors.cpu().numpy(); m = sc.pts_voxel_maps.cpu().numpy()
    for v in (0, len(vc) // 2, len(vc) - 1):
        sel = (m == v)
        assert (coors4[sel] == vc[v]).all()
        np.testing.assert_allclose(o1[v].cpu().numpy(), feats[sel].astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('c', [3, 4, 9, 10, 16, 33, 64, 100, 127])
@pytest.mark.parametrize('red', [0, 1, 2])
def test_gpu_backward_voxel_order_equals_map_order(c, red):
    """vox_scatter_backward_grouped (each voxel row read once, streamed to its points) must produce the same bits as the
    map-ordered gather, incl. zero rows for the points outside every voxel (3 % of the cloud) and untouched arg-max ties."""
    import mmdet3d_gaussian_amd as amd
    from mmdet3d_gaussian_amd.scatter import Scatter, group_points
    lib = amd.load_library()
    n = 30000
    coors, feats = _cloud(n, 100 + c, c=c, neg_frac=0.03)
    sc = Scatter(torch.from_numpy(coors).cuda())
    order, seg = group_points(sc.pts_voxel_maps, sc.voxel_pts_counts)
    assert int(seg[0]) > 0                                  # there ARE invalid points in this cloud
    v = sc.voxel_coors.shape[0]
    f = torch.from_numpy(feats).cuda()
    out = torch.empty(v, c, device='cuda'); arg = torch.empty(v, c, dtype=torch.int32, device='cuda')
    assert lib.vox_scatter_reduce(f.data_ptr(), order.data_ptr(), seg.data_ptr(), n, c, v, red, out.data_ptr(),
                                  arg.data_ptr() if red == 2 else None, None) == 0
    gv = torch.randn(v, c, device='cuda')
    a = torch.full((n, c), float('nan'), device='cuda'); b = torch.full((n, c), float('nan'), device='cuda')
    am = arg.data_ptr() if red == 2 else None
    assert lib.vox_scatter_backward(gv.data_ptr(), sc.pts_voxel_maps.data_ptr(), sc.voxel_pts_counts.data_ptr(), am, n, c, v,
                                    red, a.data_ptr(), None) == 0
    assert lib.vox_scatter_backward_grouped(gv.data_ptr(), order.data_ptr(), seg.data_ptr(), am, n, c, v, red, b.data_ptr(),
                                            None) == 0
    torch.cuda.synchronize()
    assert not torch.isnan(b).any() and torch.equal(a, b)
    assert (b[sc.pts_voxel_maps < 0] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('red', ['sum', 'max'])
def test_gpu_scatter_feature_offsets_beyond_2_31(red):
    """36 M points x 64 channels = 2.3e9 floats: element offsets beyond int32 in the point-feature array.  The 64-channel
    result equals the two 32-channel halves reduced on their own (every channel is summed in ascending point order whatever
    the channel count), forward and backward, and the rows of the LAST points carry the right voxel's gradient."""
    from mmdet3d_gaussian_amd.scatter import Scatter
    n, c = 36_000_000, 64
    free, _ = torch.cuda.mem_get_info()
    if free < (48 << 30):
        pytest.skip('needs 48 GB of free HBM')
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(3)
    coors = torch.stack([torch.randint(0, 10, (n,), generator=g, device=dev), torch.randint(0, 400, (n,), generator=g, device=dev),
                         torch.randint(0, 400, (n,), generator=g, device=dev)], 1).int()
    coors[::97, 1] = -1                                     # points outside every voxel
    feats = torch.empty(n, c, device=dev)
    for s in range(0, n, 4_000_000):
        feats[s:s + 4_000_000] = torch.randn(4_000_000, c, generator=g, device=dev)
    assert feats.numel() > 2 ** 31
    sc = Scatter(coors)
    f = feats.requires_grad_(True)
    out, _ = sc.reduce(f, red)
    halves = [sc.reduce(feats.detach()[:, k:k + 32].contiguous(), red)[0] for k in (0, 32)]
    assert torch.equal(out.detach()[:, :32], halves[0]) and torch.equal(out.detach()[:, 32:], halves[1])
    gv = torch.randn(out.shape, generator=g, device=dev)
    out.backward(gv)
    m = sc.pts_voxel_maps
    tail = slice(n - 1000, n)                               # rows whose offsets are beyond 2^31
    mt = m[tail].long()
    if red == 'sum':
        want = torch.where((mt >= 0)[:, None], gv[mt.clamp(min=0)], torch.zeros((), device=dev))
        assert torch.equal(f.grad[tail], want)
    else:
        hit = out.detach()[mt.clamp(min=0)] == feats.detach()[tail]
        got = f.grad[tail]
        assert bool(((got != 0) <= (hit & (mt >= 0)[:, None])).all())          # gradient only where the point IS the max
        assert torch.equal(got[got != 0], gv[mt.clamp(min=0)][got != 0])
    del f, feats, out, gv
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_gpu_scatter_all_points_outside_and_single_voxel(host_glue):
    """Edge clouds: every point invalid (no voxel at all), and every point in ONE voxel (the longest possible segment)."""
    from mmdet3d_gaussian_amd.scatter import Scatter
    n, c = 5000, 16
    feats = torch.randn(n, c, device='cuda', requires_grad=True)
    none = Scatter(torch.full((n, 3), -1, dtype=torch.int32, device='cuda'))
    assert none.voxel_coors.shape[0] == 0 and bool((none.pts_voxel_maps == -1).all())
    out, vc = none.reduce(feats, 'max')
    assert out.shape == (0, c) and vc.shape[0] == 0
    back = none.reduce_mapback(feats.detach(), 'mean', default_feat=7.0)
    assert back.shape == (n, c) and bool((back == 7.0).all())
    one = Scatter(torch.tensor([[3, 2, 1]], dtype=torch.int32, device='cuda').expand(n, 3).contiguous())
    assert one.voxel_coors.tolist() == [[3, 2, 1]] and one.voxel_pts_counts.tolist() == [n]
    for red in ('sum', 'mean', 'max'):
        feats.grad = None
        o, _ = one.reduce(feats, red)
        f64 = feats.detach().double()
        want = {'sum': f64.sum(0), 'mean': f64.mean(0), 'max': f64.max(0).values}[red]
        # 5000 fp32 terms added in ascending point order: |error| <~ n * eps * max|partial sum| ~ 2e-3 on the sum
        atol = {'sum': 2e-3, 'mean': 2e-3 / n, 'max': 0.0}[red]
        assert torch.allclose(o[0].double(), want, rtol=1e-6, atol=atol)
        o.sum().backward()
        if red == 'max':
            assert int((feats.grad != 0).sum()) == c                  # exactly one winner per channel
        else:
            assert torch.allclose(feats.grad, torch.full_like(feats, 1.0 if red == 'sum' else 1.0 / n))


@pytest.mark.gpu
@pytest.mark.parametrize('c', [3, 10, 16])
@pytest.mark.parametrize('red', ['sum', 'mean', 'max'])
def test_gpu_narrow_rows_long_runs_and_same_bits_as_the_vector_kernels(host_glue, c, red):
    """Rows that are not whole 16-byte vectors, and narrow rows, on a skewed cloud: three voxels hold 40 % of the points (runs of
    thousands of points; in the voxel-ordered backward several LDS tiles per workgroup), the rest are singletons and small
    voxels.  Every kernel visits a voxel's points in ascending point id: the c channels embedded in the first columns of a
    64-channel array (which takes the 16-byte vector kernels) give bit-identical sums / means / maxima."""
    from mmdet3d_gaussian_amd.scatter import Scatter
    rng = np.random.default_rng(c)
    n = 50_000
    coors = np.stack([rng.integers(0, 60, n), rng.integers(0, 60, n), rng.integers(0, 2, n)], -1).astype(np.int32)
    big = rng.random(n) < 0.4
    coors[big] = np.array([[5, 5, 0], [30, 31, 1], [59, 59, 1]], np.int32)[rng.integers(0, 3, big.sum())]
    coors[rng.random(n) < 0.02, 1] = -1
    feats = rng.normal(0, 1, (n, c)).astype(np.float32)
    wide = np.zeros((n, 64), np.float32); wide[:, :c] = feats
    sc = Scatter(torch.from_numpy(coors).cuda())
    assert int(sc.voxel_pts_counts.max()) > 4000
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    out, _ = sc.reduce(f, red)
    ref, _ = sc.reduce(torch.from_numpy(wide).cuda(), red)
    assert torch.equal(out.detach(), ref[:, :c].contiguous())
    uo, mo, co = vo.scatter_index(coors)
    want, _ = vo.scatter_reduce(feats, mo, co, red)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-4, atol=1e-3)   # fp32 running sums over up to 7 000 points
    gv = torch.randn(out.shape, device='cuda')
    out.backward(gv)
    gw = vo.scatter_backward(gv.cpu().numpy(), feats, mo, co, red)
    np.testing.assert_allclose(f.grad.cpu().numpy(), gw, rtol=2e-6, atol=1e-7)
    assert (f.grad[sc.pts_voxel_maps < 0] == 0).all()


@pytest.mark.gpu
def test_gpu_index_reports_key_overflow_and_out_of_range_coordinates():
    """extents whose product passes 2^63 used to wrap the mixed-radix key (merged voxels, fake dropped points); an
    int64 coordinate beyond int32 used to wrap in the cast.  Both are errors now; a large negative int64 is a dropped point."""
    from mmdet3d_gaussian_amd.scatter import scatter_index
    big = torch.tensor([[2 ** 30, 2 ** 30, 2 ** 30, 5], [1, 2, 3, 4]], dtype=torch.int32, device='cuda')   # product ~ 2^90 x 6
    with pytest.raises(RuntimeError, match='63-bit'):
        scatter_index(big)
    ok = torch.tensor([[2 ** 20, 2 ** 20, 2 ** 20], [1, 2, 3], [1, 2, 3]], dtype=torch.int32, device='cuda')        # 2^60: fits
    vc, pm, cnt = scatter_index(ok)
    assert vc.shape[0] == 2 and pm.tolist() == [1, 0, 0] and cnt.tolist() == [2, 1]
    wide = torch.tensor([[1, 2, 3], [2 ** 40, 0, 0]], dtype=torch.int64, device='cuda')
    with pytest.raises(RuntimeError, match='int32'):
        scatter_index(wide)
    neg = torch.tensor([[1, 2, 3], [-2 ** 40, 0, 0], [1, 2, 3]], dtype=torch.int64, device='cuda')
    vc, pm, cnt = scatter_index(neg)
    assert pm.tolist() == [0, -1, 0] and vc.dtype == torch.int64


@pytest.mark.gpu
def test_gpu_backward_with_its_zero_fill_replays_in_a_hipgraph(host_glue):
    """The arg-max backward at narrow rows clears grad_feats and then routes the voxel gradients: the clear is a KERNEL of the
    library, not a hipMemsetAsync — a memset node of a captured graph was found not to be reliably ordered against the kernels
    around it on this ROCm (round 4, rotated NMS).  Five replays on a poisoned output buffer must equal the eager result; the
    degenerate calls whose whole effect is a clear (no voxels; an empty loss call) are captured alongside."""
    import ctypes
    import mmdet3d_gaussian_amd as amd
    from mmdet3d_gaussian_amd.scatter import Scatter, group_points
    lib = amd.load_library()
    n, c, red = 30000, 10, 2
    coors, feats = _cloud(n, 321, c=c, neg_frac=0.03)
    sc = Scatter(torch.from_numpy(coors).cuda())
    order, seg = group_points(sc.pts_voxel_maps, sc.voxel_pts_counts)
    v = sc.voxel_coors.shape[0]
    f = torch.from_numpy(feats).cuda()
    out = torch.empty(v, c, device='cuda'); arg = torch.empty(v, c, dtype=torch.int32, device='cuda')
    assert lib.vox_scatter_reduce(f.data_ptr(), order.data_ptr(), seg.data_ptr(), n, c, v, red, out.data_ptr(), arg.data_ptr(), None) == 0
    gv = torch.randn(v, c, device='cuda')
    want = torch.full((n, c), float('nan'), device='cuda')
    assert lib.vox_scatter_backward(gv.data_ptr(), sc.pts_voxel_maps.data_ptr(), sc.voxel_pts_counts.data_ptr(), arg.data_ptr(), n, c, v,
                                    red, want.data_ptr(), None) == 0
    got = torch.full((n, c), float('nan'), device='cuda')
    none = torch.full((64, c), float('nan'), device='cuda')          # a call with v == 0: its whole effect is the clear
    lsum = torch.full((1,), float('nan'), device='cuda')             # gd3d_loss_reduce with n == 0: writes 0
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        s = ctypes.c_void_p(side.cuda_stream)
        with torch.cuda.graph(graph, stream=side):
            assert lib.vox_scatter_backward(gv.data_ptr(), sc.pts_voxel_maps.data_ptr(), sc.voxel_pts_counts.data_ptr(), arg.data_ptr(),
                                            n, c, v, red, got.data_ptr(), s) == 0
            assert lib.vox_scatter_backward(gv.data_ptr(), sc.pts_voxel_maps.data_ptr(), sc.voxel_pts_counts.data_ptr(), None,
                                            64, c, 0, 0, none.data_ptr(), s) == 0
            assert lib.gd3d_loss_reduce(None, 0, lsum.data_ptr(), s) == 0
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5):
        got.fill_(float('nan')); none.fill_(float('nan')); lsum.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert (none == 0).all() and lsum.item() == 0.0
