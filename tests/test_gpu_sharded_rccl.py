"""The N > 1 product combination on ONE GPU: an RCCL ("nccl") process group of world size 1 around the real HIP GDLoss —
ShardedGDLoss, the asynchronous all_gather of shard losses and a hipGraph replay of the step, as bench.py runs it at
N > 1 (reference counterpart: per-rank loss under DDP, /root/reference/tools/train.py:130-137, dist_train.sh:8-9).
tests/test_sharded_gloo.py covers world_size 2 on the CPU with an oracle-backed stand-in for the local loss; here the
local loss is the product kernel and the collective really goes through RCCL.  With one rank every sharded result must
equal the plain GDLoss result BIT FOR BIT, and the gradient must not be touched by the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _pairs(n, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -3.0], device=dev)
    hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, 3.0], device=dev)
    t = torch.rand(n, 7, generator=g, device=dev) * (hi - lo) + lo
    p = t + torch.randn(n, 7, generator=g, device=dev) * 0.1
    return p.contiguous(), t.contiguous()


@pytest.fixture(scope='module')
def rccl_group():
    assert torch.cuda.is_available()
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    yield dev
    torch.cuda.synchronize()
    dist.destroy_process_group()
    assert not dist.is_initialized()


@pytest.mark.parametrize('lt', ['gwd3d', 'kld3d', 'bd3d'])
@pytest.mark.parametrize('n', [5000, 300_001])
def test_sharded_loss_over_rccl_equals_plain_loss(rccl_group, lt, n):
    import mmdet3d_gaussian_amd as amd
    dev = rccl_group
    pred, tgt = _pairs(n, 3, dev)
    mod = amd.GDLoss(lt, loss_weight=5.0)
    p0 = pred.clone().requires_grad_(True)
    plain = mod(p0, tgt)
    plain.backward()
    sh = amd.sharded.ShardedGDLoss(amd.GDLoss(lt, loss_weight=5.0))
    p1 = pred.clone().requires_grad_(True)
    out = sh(p1, tgt)                       # total_pairs through an all_reduce, the loss through an all_gather
    out.backward()
    assert out.device == plain.device and torch.equal(out, plain)
    assert torch.equal(p1.grad, p0.grad)
    # the asynchronous gather the benchmark uses
    pend = amd.sharded.gather_shard_losses(plain.detach(), async_op=True)
    total, per_rank = pend.result()
    assert per_rank.shape == (1,) and torch.equal(total, plain.detach()) and torch.equal(per_rank[0], plain.detach())


def test_graph_replayed_step_with_async_gather(rccl_group):
    """bench.py's N > 1 step: a hipGraph of (three GDLoss forwards + backward), replayed, followed by the asynchronous
    all_gather of the (3,) shard losses on a private copy; values and gradients equal the eager plain module, replay
    after replay, and a changed input is seen by the next replay."""
    import mmdet3d_gaussian_amd as amd
    from mmdet3d_gaussian_amd import gd_loss as gdl
    dev = rccl_group
    n = 200_000
    pred, tgt = _pairs(n, 5, dev)
    lts = ('gwd3d', 'kld3d', 'bd3d')
    mods = {lt: amd.sharded.ShardedGDLoss(amd.GDLoss(lt, loss_weight=5.0)) for lt in lts}
    preds = {lt: pred.clone().requires_grad_(True) for lt in lts}
    unit = [gdl.unit_grad(dev)] * 3

    def compute():
        ls = []
        for lt in lts:
            preds[lt].grad = None
            ls.append(mods[lt].local_loss(preds[lt], tgt, total_pairs=n))
        torch.autograd.backward(ls, grad_tensors=unit)
        return torch.stack([l.detach() for l in ls])

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            compute()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        gouts = compute()
    ggrads = {lt: preds[lt].grad for lt in lts}

    def reference():
        vals, grads = [], {}
        for lt in lts:
            q = preds[lt].detach().clone().requires_grad_(True)
            v = amd.GDLoss(lt, loss_weight=5.0)(q, tgt)
            v.backward()
            vals.append(v.detach())
            grads[lt] = q.grad
        return torch.stack(vals), grads

    for rep in range(3):
        if rep == 2:   # new inputs in the captured buffers: the replay must see them
            with torch.no_grad():
                for lt in lts:
                    preds[lt].add_(0.05)
        graph.replay()
        pend = amd.sharded.gather_shard_losses(gouts.clone(), async_op=True)
        total, per_rank = pend.result()
        want, wgrads = reference()
        assert per_rank.shape == (1, 3)
        assert torch.equal(total, want), (rep, total, want)
        for lt in lts:
            assert torch.equal(ggrads[lt], wgrads[lt]), (rep, lt)
    torch.cuda.synchronize()


def test_bench_line_proves_its_ranks_over_rccl():
    """bench.py under the driver's launcher with ONE rank over RCCL (a child process; the card allows it next to this one):
    the line's `config` carries what the collective RETURNED — rank ids that travelled in the per-step payload, the per-rank
    loss stack, per-rank fused-kernel means — and the RCCL version; at N ranks the same fields are what proves N ranks ran
    (tests/test_bench_rehearsal.py asserts them for 2 and 3 gloo ranks).  Launcher as /root/reference/tools/dist_train.sh:8-9."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--pairs', '1000000', '--steps', '5', '--warmup', '2',
           '--prewarm', '0.2', '--cpu-sample', '0', '--no-traffic', '--graph']     # --graph: the launch mode of N > 1 ranks
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.lstrip().startswith('{') and '"metric"' in ln]
    assert len(lines) == 1
    d, c = lines[0], lines[0]['config']
    assert d['n_gpus'] == 1 and c['launch'] == 'hipGraph replay' and 'RCCL' in c['collective']
    assert c['ranks_seen'] == [0] and len(c['per_rank_loss']) == 1 and len(c['per_rank_loss'][0]) == 3
    assert all(abs(a - d['loss_values'][k]) <= 1e-5 for a, k in zip(c['per_rank_loss'][0], ('gwd3d', 'kld3d', 'bd3d')))
    assert c['rccl_version'] and c['rccl_version'][0].isdigit() and '.' in c['rccl_version']
    assert c['per_rank_kernel_ms']['max'] == c['per_rank_kernel_ms']['min'] and c['per_rank_kernel_ms']['max']['kld3d'] > 0
    assert c['host_glue'] in ('python', 'cpp') and d['value_form'].startswith('plain')
