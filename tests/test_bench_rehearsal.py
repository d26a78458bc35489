"""Rehearsal of bench.py's N > 1 control flow without GPUs: the same rank loop — shard ranges, per-step
asynchronous gather of the three shard losses, MAX-reduce of the elapsed time, exactly one JSON line from rank 0, non-zero
exit when a rank dies — under `python -m torch.distributed.run` (the driver's launcher) and under bench.py's own self-launch,
with `--backend gloo --device cpu` (CPU tensors through GDLoss's `_cpu` twins).  What an 8-GPU node's first run then has left
to discover is RCCL itself.  Reference counterpart of the launcher: tools/dist_train.sh:8-9."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')
COMMON = ['--backend', 'gloo', '--device', 'cpu', '--steps', '4', '--warmup', '2', '--prewarm', '0', '--cpu-sample', '0', '--plain-steps', '3']


def _port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra)
    return env


def _torchrun(n, args, **env):
    for attempt in range(2):   # (a port found free a moment ago can be taken by the time the launcher binds it: one more try)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
               '--master-port', str(_port()), BENCH, '--gpus', str(n)] + COMMON + args
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env(**env), cwd=ROOT)
        if r.returncode == 0 or not any(m in r.stderr for m in ('Address already in use', 'EADDRINUSE', 'address already in use')):
            break
    return r


def _result_lines(stdout):
    out = []
    for line in stdout.splitlines():
        if line.lstrip().startswith('{'):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if 'metric' in d:
                out.append(d)
    return out


def _expected_losses(world, rows_of_rank):
    """Mean over ranks of the per-rank mean loss x 5 (bench.py seeds rank r's pairs with r), from the fp64 oracle."""
    sys.path.insert(0, ROOT)
    import bench
    vals = np.zeros(3)
    for r in range(world):
        p, t = bench.synthetic_pairs(rows_of_rank(r), r, torch.device('cpu'))
        for k, lt in enumerate(bench.LOSSES):
            n = p.shape[0]
            vals[k] += oracle.gd_loss(p.numpy(), t.numpy(), oracle.make_params(lt, fun='log1p', tau=1.0), scale=5.0 / n)['loss_sum']
    return vals / world


def test_two_ranks_under_the_drivers_launcher_weak_scaling():
    r = _torchrun(2, ['--pairs', '20000'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _result_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['warmup'] == 2 and d['scaling'] == 'weak' and d['roofline'] is None
    assert d['config']['pairs_per_gpu'] == 20000 and d['config']['parallelism'] == 'pair-sharded x2'
    assert 'gloo' in d['config']['collective'] and d['config']['device'].startswith('cpu')
    # whole-job value: 3 losses x 20 000 pairs x 2 ranks x 4 steps over the max-over-ranks time
    assert abs(d['value'] - 3 * 20000 * 2 * 4 / (d['ms_per_step'] * 4e-3) / 1e6) <= 0.02 * d['value']
    # round 5: `value` times the reference caller's step (plain backward); the unit-gradient form is the second region
    assert d['value_form'].startswith('plain (l0 + l1 + l2).backward()')
    assert d['plain_backward_steps'] == 4 and d['value_plain_backward'] == d['value']
    assert d['unit_grad_steps'] == 3 and d['value_unit_grad'] > 0
    assert d['config']['host_glue'] in ('python', 'cpp')
    want = _expected_losses(2, lambda rank: 20000)
    got = np.array([d['loss_values'][k] for k in ('gwd3d', 'kld3d', 'bd3d')])
    assert np.all(np.abs(got - want) <= 1e-5 * (1 + np.abs(want))), (got, want)
    # the line proves its ranks from what the collective RETURNED: the rank ids travel in the per-step
    # payload, the per-rank losses are the last step's (world, 3) stack
    c = d['config']
    assert c['ranks_seen'] == [0, 1] and c['rccl_version'] is None and 'rank id' in c['collective']
    per = np.array(c['per_rank_loss'])
    assert per.shape == (2, 3)
    assert np.allclose(per.mean(0), want, rtol=1e-5) and not np.allclose(per[0], per[1])   # rank r's pairs are seeded with r
    assert c['per_rank_rows'] == [[0, 20000], [20000, 40000]]
    assert set(c['per_rank_kernel_ms']) == {'max', 'min'}


def test_three_ranks_strong_scaling_contiguous_row_ranges():
    r = _torchrun(3, ['--pairs', '30001', '--strong'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _result_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert d['scaling'] == 'strong' and d['n_gpus'] == 3 and d['config']['pairs_per_gpu'] == 30001 * 1 // 3
    assert d['config']['ranks_seen'] == [0, 1, 2] and np.array(d['config']['per_rank_loss']).shape == (3, 3)
    assert d['config']['per_rank_rows'] == [[0, 10000], [10000, 20000], [20000, 30001]]     # contiguous ranges [r N/G, (r+1) N/G)
    assert abs(d['value'] - 3 * 30001 * 4 / (d['ms_per_step'] * 4e-3) / 1e6) <= 0.02 * d['value']


def test_self_launch_starts_the_ranks_and_relays_one_line():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it: bench.py is the launcher (children only, never a re-exec)."""
    cmd = [sys.executable, BENCH, '--gpus', '2', '--pairs', '8000'] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(_result_lines(r.stdout)) == 1 and r.stdout.count('\n') == 1


def test_a_dying_rank_fails_the_job_and_prints_no_result():
    r = _torchrun(2, ['--pairs', '8000'], GD3D_BENCH_FAIL_RANK='1')
    assert r.returncode != 0
    assert _result_lines(r.stdout) == []
    cmd = [sys.executable, BENCH, '--gpus', '2', '--pairs', '8000'] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env(GD3D_BENCH_FAIL_RANK='0'), cwd=ROOT)
    assert r.returncode != 0 and _result_lines(r.stdout) == []


def test_every_launch_path_gets_the_rccl_environment_defaults():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC; RCCL fails without it on these hosts) is set by main() itself before anything
    touches the GPU — not only by self_launch — and an explicit setting of the user's survives."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.RCCL_ENV_DEFAULTS == {'HSA_ENABLE_IPC_MODE_LEGACY': '0'}
    src = open(BENCH).read()
    main_body = src[src.index('def main():'):]
    assert main_body.index('RCCL_ENV_DEFAULTS') < main_body.index('init_process_group') < main_body.index('import mmdet3d_gaussian_amd')
    probe = ("import os, sys; sys.argv = ['bench.py', '--pmc-child', '--pairs', '0']; sys.path.insert(0, %r); import bench\n"
             "try:\n    bench.main()\nexcept BaseException:\n    pass\nprint('IPC=' + os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset'))" % ROOT)
    env = _env(); env.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)
    r = subprocess.run([sys.executable, '-c', probe], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert 'IPC=0' in r.stdout, (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, '-c', probe], capture_output=True, text=True, timeout=120, env=_env(HSA_ENABLE_IPC_MODE_LEGACY='1'), cwd=ROOT)
    assert 'IPC=1' in r.stdout


def test_input_allocation_forms_give_the_same_losses_and_say_which_one_ran():
    """Since round 6 the headline form is one torch allocation per input array (what a caller of the reference has: pred and target
    are separate torch.cat outputs); `--arena` makes the four arrays row ranges of ONE allocation (rounds 4-5's headline: separately
    allocated read streams collide on some draws of their physical placement).  `--separate-inputs` is accepted and
    means the default.  Same values either way; the result line names the form in `config` and in `value_form`."""
    lines = []
    for extra in ([], ['--separate-inputs'], ['--arena']):
        r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--pairs', '20000'] + COMMON + extra, capture_output=True, text=True,
                           timeout=300, env=_env(), cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        got = _result_lines(r.stdout)
        assert len(got) == 1
        lines.append(got[0])
    a, b, c = lines
    assert a['config']['input_allocation'] == b['config']['input_allocation'] == 'one torch allocation per array'
    assert 'one allocation per input array' in a['value_form']
    assert 'row ranges of one allocation' in c['config']['input_allocation'] and 'row ranges of one allocation' in c['value_form']
    assert a['loss_values'] == b['loss_values'] == c['loss_values']
    assert a['config']['gc_disabled'] is True
    assert a['value_separate_inputs'] == a['value'] and c['value_one_allocation'] == c['value']


def test_prewarm_with_collectives_leaves_every_rank_with_the_same_step_count():
    """The pre-warm loop runs for a wall-clock time (and, on the GPU, until step times settle) and every step carries a
    collective: the decision to stop must be taken by all ranks together, or the ranks end up with different collective counts
    and the job hangs in its last gather (latent through round 3: the rehearsals ran with --prewarm 0, the driver does not)."""
    args = [a for a in COMMON]
    args[args.index('--prewarm') + 1] = '0.4'
    r = _torchrun(3, args + ['--pairs', '30000'])
    assert r.returncode == 0, r.stderr[-3000:]
    got = _result_lines(r.stdout)
    assert len(got) == 1 and got[0]['n_gpus'] == 3 and got[0]['config']['prewarm_s'] >= 0.3


# ---- the REAL world size: eight ranks, as the driver's SCALE run starts them -------------------------------------

def _bench_children(marker):
    """Live processes whose command line carries `marker` (a --pairs value no other test uses)."""
    import psutil
    out = []
    for p in psutil.process_iter(['pid', 'cmdline', 'status']):
        try:
            if p.info['status'] != psutil.STATUS_ZOMBIE and marker in ' '.join(p.info['cmdline'] or []):
                out.append(p.info['pid'])
        except (psutil.NoSuchProcess, psutil.AccessDenied):
            pass
    return out


FAST = ['--backend', 'gloo', '--device', 'cpu', '--steps', '2', '--warmup', '1', '--prewarm', '0', '--cpu-sample', '0', '--plain-steps', '1']


def test_eight_ranks_strong_scaling_at_the_benchmark_size_under_the_drivers_launcher():
    """`--strong` at the headline size: 10 M pairs over 8 ranks = 1 250 000 rows each, through
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ...` (the form the driver uses for its
    SCALE run; reference counterpart tools/dist_train.sh:8-9).  The line must prove all eight ranks from what the collective
    returned, the row ranges must tile [0, 10 M) and `value` must count the TOTAL once."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
           '--master-port', str(_port()), BENCH, '--gpus', '8', '--pairs', '10000000', '--strong'] + FAST
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _result_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    c = d['config']
    assert d['n_gpus'] == 8 and d['scaling'] == 'strong' and c['pairs_per_gpu'] == 1_250_000
    assert c['ranks_seen'] == list(range(8)) and np.array(c['per_rank_loss']).shape == (8, 3)
    rows = c['per_rank_rows']
    assert rows[0][0] == 0 and rows[-1][1] == 10_000_000 and all(rows[k][1] == rows[k + 1][0] for k in range(7))
    assert sum(b - a for a, b in rows) == 10_000_000 and all(b - a == 1_250_000 for a, b in rows)
    assert abs(d['value'] - 3 * 10_000_000 * 2 / (d['ms_per_step'] * 2e-3) / 1e6) <= 0.02 * d['value']
    assert 'gloo' in c['collective']


def test_eight_ranks_self_launch_weak_and_a_total_that_does_not_divide():
    """`python bench.py --gpus 8` with no launcher around it starts its eight ranks itself (children, never a re-exec): weak
    scaling keeps --pairs per rank; a strong total of 100 003 pairs splits into the contiguous ranges [r N / 8, (r + 1) N / 8)."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--pairs', '6007'] + FAST, capture_output=True, text=True, timeout=600,
                       env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _result_lines(r.stdout)
    assert len(lines) == 1 and r.stdout.count('\n') == 1
    d = lines[0]
    assert d['n_gpus'] == 8 and d['scaling'] == 'weak' and d['config']['ranks_seen'] == list(range(8))
    assert d['config']['per_rank_rows'] == [[6007 * k, 6007 * (k + 1)] for k in range(8)]
    assert abs(d['value'] - 3 * 6007 * 8 * 2 / (d['ms_per_step'] * 2e-3) / 1e6) <= 0.02 * d['value']
    want = _expected_losses(8, lambda rank: 6007)
    got = np.array([d['loss_values'][k] for k in ('gwd3d', 'kld3d', 'bd3d')])
    assert np.all(np.abs(got - want) <= 1e-5 * (1 + np.abs(want))), (got, want)
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--pairs', '100003', '--strong'] + FAST, capture_output=True, text=True,
                       timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _result_lines(r.stdout)[0]
    rows = d['config']['per_rank_rows']
    assert rows == [[100003 * k // 8, 100003 * (k + 1) // 8] for k in range(8)] and sum(b - a for a, b in rows) == 100003
    assert d['config']['ranks_seen'] == list(range(8))


def test_eight_ranks_one_dies_nonzero_exit_and_every_child_reaped():
    """Rank 5 of 8 dies after its warm-up steps: the job exits non-zero, prints no result line, and leaves no rank process behind
    (seven ranks sit in a collective their peer will never join: the launcher must end them)."""
    marker = '7919'
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--pairs', marker] + FAST, capture_output=True, text=True, timeout=600,
                       env=_env(GD3D_BENCH_FAIL_RANK='5'), cwd=ROOT)
    assert r.returncode != 0 and _result_lines(r.stdout) == []
    import time
    for _ in range(50):
        left = _bench_children('--pairs ' + marker)
        if not left:
            break
        time.sleep(0.2)
    assert left == [], left
