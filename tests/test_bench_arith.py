"""bench.py's rank / shard arithmetic, `value` formula and launcher logic, without a GPU."""
import io
import json
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_strong_scaling_row_ranges_partition_the_total():
    for total in (10_000_000, 10, 7, 1):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_rows(total, r, world, strong=True) for r in range(world)]
            assert spans[0][0] == 0
            for (lo, n), (lo2, _) in zip(spans, spans[1:]):
                assert lo + n == lo2                                  # contiguous, in rank order
            assert spans[-1][0] + spans[-1][1] == total               # covers every row exactly once
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1


def test_weak_scaling_keeps_pairs_per_gpu():
    for world in (1, 2, 4, 8):
        for r in range(world):
            assert bench.shard_rows(10_000_000, r, world, strong=False) == (r * 10_000_000, 10_000_000)


def test_value_is_whole_job_pairs_per_second():
    # 3 losses x 10 M pairs x 20 steps in 9.16 ms at N = 1 -> 65.5 G pairs/s (the r01 driver record)
    v1 = bench.job_value(10_000_000, 1, False, 20, 20 * 0.4583e-3)
    assert abs(v1 - 65459.3) < 1.0
    # weak scaling: 8 ranks, same step time -> 8x; strong scaling: the total stays 10 M pairs
    assert abs(bench.job_value(10_000_000, 8, False, 20, 20 * 0.4583e-3) - 8 * v1) < 1e-6 * v1
    assert abs(bench.job_value(10_000_000, 8, True, 20, 20 * 0.4583e-3) - v1) < 1e-6 * v1


# ---- `python bench.py --gpus N` starts its own ranks (reference: tools/dist_train.sh:8-9) ----
def test_self_launch_refuses_when_fewer_gpus_are_visible():
    out, err = io.StringIO(), io.StringIO()
    rc = bench.self_launch(2, ['--gpus', '2'], visible=1, out=out, err=err)
    assert rc != 0 and out.getvalue() == ''
    assert '2 GPUs requested, 1 visible' in err.getvalue()


def test_launch_command_is_the_drivers_form():
    cmd = bench.launch_command(8, ['--gpus', '8', '--steps', '20', '--warmup', '5'], 29555)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29555'
    assert cmd[-7] == os.path.join(ROOT, 'bench.py') and cmd[-6:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']


def _fake(code):
    return [sys.executable, '-c', code]


def test_self_launch_relays_the_result_line_and_the_exit_status(tmp_path):
    line = json.dumps({'metric': 'M box-pairs/sec', 'value': 1.0, 'n_gpus': 2})
    errf = open(tmp_path / 'err.txt', 'w+')
    out = io.StringIO()
    ok = _fake(f"import sys; print('noise from a rank'); print({line!r}); print('{{not json'); sys.stderr.write('warn\\n')")
    assert bench.self_launch(2, [], visible=2, cmd=ok, out=out, err=errf) == 0
    assert out.getvalue() == line + '\n'                 # exactly the one result line on stdout
    errf.seek(0)
    text = errf.read()
    assert 'noise from a rank' in text and 'warn' in text and '{not json' in text
    # a rank that fails: non-zero, whatever it printed
    out = io.StringIO()
    bad = _fake(f"import sys; print({line!r}); sys.exit(3)")
    assert bench.self_launch(2, [], visible=2, cmd=bad, out=out, err=errf) == 3
    # ranks that exit 0 without a result line are a failure too
    out = io.StringIO()
    assert bench.self_launch(2, [], visible=2, cmd=_fake("print('nothing')"), out=out, err=errf) != 0
    errf.close()


def test_bench_main_becomes_the_launcher_without_touching_a_gpu():
    """On this CPU-only container `python bench.py --gpus 2` must fail from the PARENT with the visible-GPU message
    (before any rank is started and before anything initialises a device)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2, (r.returncode, r.stderr[-500:])
        assert '2 GPUs requested' in r.stderr and r.stdout.strip() == ''
