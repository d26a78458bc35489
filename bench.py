#!/usr/bin/env python3
"""bench.py — throughput of the Gaussian-distance hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): M box-pairs/s, forward+backward, for GWD / KLD / BCD at 10 M pairs.
Workload (BASELINE.json configs[2]): 10 M synthetic anchor x gt pairs per GPU (SURVEY.md §8d recipe,
seed 0), GDLoss(loss_type, fun='log1p', tau=1.0, alpha=1.0, reduction='mean', loss_weight=5.0).
One "step" = gwd3d, kld3d and bd3d, each forward + backward over the whole batch, through the
reference's module surface (GDLoss.forward x 3 -> one autograd backward of the summed losses, as a
training step does with its loss dict).  Inputs are resident in HBM before
the timed region.  value = (3 losses x pairs x steps x ranks) / max-over-ranks wall time.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), pairs sharded by rank with no
data-path collective; once per step the three per-shard loss values are all-gathered (3 fp32 per rank),
asynchronously (the next step's kernels overlap it).  Weak scaling: every rank keeps 10 M pairs
(`--strong`: 10 M pairs in TOTAL, contiguous row ranges per rank — SURVEY.md §8e asks for both).

Launch mode.  N = 1 launches eagerly, and on one MI355X the eager step is the FASTER one: same process, same buffers, round 3
(profiles/r03_step_variants.jsonl): eager 423-427 us per step, hipGraph replay 430 us around 411 us of fused kernels; the
host needs ~100 us per step.  (Round 2's eager step cost 434-445 us: its backward added a ones-fill, two adds and three
early-exit launches.)  Every `--event-every`-th step of the timed region (default 5) carries a HIP event pair on each of
its fused launches (gd3d_loss_fused_timed: hipExtLaunchKernel binds them to the dispatch's begin/end timestamps; no
marker packets): the kernel durations behind `roofline` are measured INSIDE the region, but not on every launch, because a
dispatch with events bound to it costs ~5 us of GPU time more than a plain one (profiles/r03_event_every_ab.txt: the step
runs 28.7 us over the sum of its three fused kernels with events on every launch, 14-16 us over it with every fifth step or
none timed; round 2 timed every launch and so taxed its own `value` by 3.5 %).
N > 1 launches eagerly too since round 5: with the plain-backward step and the per-step collective in the loop the eager step is
the faster one (one rank under torch.distributed.run, RCCL all_gather per step, profiles/r05_rccl_rehearsal_*.json: 439.6 us per
step eager, 443.6 / 444.9 us as a hipGraph replay, against 416.5 us for the N = 1 step without a collective; rounds 3-4, whose
step handed over the unit gradient, measured the replay ahead: 425 vs 432 us).  `--graph` replays a hipGraph of the step
(torch.cuda.CUDAGraph; host ~10 us/step): HIP events cannot be bound to a dispatch inside a captured graph on ROCm, so there the
per-kernel durations come from an eager pass run right after the timed region, same stream, same process (`roofline.timing` says
which), followed by a bracketed run of bare replays (`config.graph_replay_ms_per_step`).
The step (round 5): three GDLoss forwards, then a plain `(l0 + l1 + l2).backward()` — the form every caller of the reference
runs (tools/train.py:213-220 -> mmcv's OptimizerHook: `loss.backward()`): torch adds the losses, fills a ones tensor, and each
loss's node launches one early-exit `grad_finish` (the fused forward launch already wrote the final gradients).  `value` times
THIS step.  Rounds 3-4 headlined the cheaper form beside it (`--unit-grad`, reported as `value_unit_grad`: one
torch.autograd.backward over the three losses whose upstream gradients are the library's unit-gradient constant,
gd_loss.unit_grad, recognised by address: backward launches nothing; ~10 us per step less).
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(python -m torch.distributed.run, 127.0.0.1), relays rank 0's JSON line and exits non-zero if any rank failed.

Rehearsal of the N > 1 control flow without GPUs: `--device cpu --backend gloo` runs the same rank loop (shard ranges,
per-step asynchronous gather of the three shard losses, MAX-reduce of the elapsed time, one JSON line from rank 0) on CPU
tensors through GDLoss's `_cpu` twins (tests/test_bench_rehearsal.py); the line then says "device": "cpu" in `config` and
carries no roofline claim.

The JSON line also carries
  value_form   : which step `value` times (plain backward unless --unit-grad) and how the inputs are allocated.
  value_unit_grad, roofline.frac_step_unit_grad : the step handing the library's unit gradient to torch.autograd.backward
                 (no ones-fill, no early-exit launches), timed over a second, shorter region right after the main one (eager).
  value_plain_backward, roofline.frac_step_plain_backward : the plain-backward figures under their round-4 names (= `value` /
                 `roofline.frac_step` unless --unit-grad swapped the regions).
  value_separate_inputs : the headline step again on inputs that are one torch allocation EACH (third region, N = 1, eager). The
                 main region's four input arrays are row ranges of ONE allocation: two read streams from separate allocations
                 collide in the memory system on some draws of their physical placement (5-8 % of the kernel time).
  roofline     : HBM roofline of the dominant kernel (the fused fwd+grad kernel): algorithmic bytes
                 (88 B/pair) / average launch duration measured with a HIP event pair bound to
                 every fused dispatch, on the stream it is launched on.  `frac` is that kernel alone;
                 `frac_step` prices the WHOLE timed step the same way (3 x 88 B x pairs / ms_per_step / peak:
                 SURVEY.md §8d's protocol, reduce launches and the backward call included).
  cpu_baseline : the fp32 C oracle ("port") timed on this host's cores on a bounded sample; beside it `torch_chain` = the
                 reference's OWN PyTorch-CPU op chain restated op for op (its time is the reference's time on this host; 1 k / 1 M /
                 10 M pairs, 1 thread and all granted cores), `torch_chain_lean` = an entry-form rewrite 2.2-2.6x faster than
                 that (not the reference's shape), `product_cpu` = this package's own CPU path (the `_cpu` twins).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LOSSES = ('gwd3d', 'kld3d', 'bd3d')
BYTES_PER_PAIR = 88          # read pred 28 + target 28, write loss 4 + grad_pred 28
MOVED_BYTES_PER_PAIR = 84    # what the kernel moves under reduction='mean': the per-pair loss is summed, not stored
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def synthetic_pairs(n, seed, device):
    """SURVEY.md §8d generator, produced on the device in chunks (no 560 MB host staging)."""
    g = torch.Generator(device=device).manual_seed(seed)
    lo = torch.tensor([0, -40, -3, 0.5, 0.5, 0.5, -math.pi], device=device)
    hi = torch.tensor([70, 40, 1, 2.5, 4.5, 2.0, math.pi], device=device)
    sigma = torch.tensor([0.3, 0.3, 0.1, 0.1, 0.1, 0.1, 0.1], device=device)
    tgt = torch.rand(n, 7, generator=g, device=device) * (hi - lo) + lo
    pred = tgt + torch.randn(n, 7, generator=g, device=device) * sigma
    return pred.float().contiguous(), tgt.float().contiguous()


def shard_rows(pairs, rank, world, strong):
    """Rows of rank `rank`: weak scaling keeps `pairs` per GPU; strong scaling splits `pairs` in total into the contiguous
    ranges [r N/G, (r+1) N/G).  Returns (first_row, n_rows) in the global numbering."""
    if not strong:
        return rank * pairs, pairs
    lo, hi = pairs * rank // world, pairs * (rank + 1) // world
    return lo, hi - lo


def job_value(pairs, world, strong, steps, elapsed_s, losses=len(LOSSES)):
    """Whole-job throughput in M box-pairs/s: every loss counts its pairs once per step (fwd + bwd together)."""
    total_pairs = losses * (pairs if strong else pairs * world) * steps
    return total_pairs / elapsed_s / 1e6


def usable_cores():
    """Threads worth starting: the scheduler affinity, capped by the cgroup CPU quota when there is one (a GPU box may
    show 256 CPUs and grant a 16-core share; 256 OpenMP threads on such a share run slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:           # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:   # v1
                quota, period = int(f.read()), int(g.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(n, 1)


def cpu_baseline(sample_pairs, seed):
    """fp32 CPU oracle (oracle/gd_oracle.c) on a bounded sample of the same workload, all host cores."""
    import oracle
    cores = usable_cores()
    pred, tgt = synthetic_pairs(sample_pairs, seed, torch.device('cpu'))
    p, t = pred.numpy(), tgt.numpy()
    loss = np.empty(sample_pairs, np.float32)
    gp = np.empty((sample_pairs, 7), np.float32)
    prms = [oracle.make_params(lt, fun='log1p', tau=1.0) for lt in LOSSES]
    for prm in prms:  # untimed pass: page in the buffers, spin up the OpenMP team
        oracle.gd_loss_timed(p[:100000], t[:100000], prm, 5.0 / sample_pairs, loss[:100000], gp[:100000], cores)
    reps, t0 = 0, time.perf_counter()
    while True:  # whole passes over the sample until ~10 s of wall time are spent (bounded: at most 64 passes)
        for prm in prms:
            oracle.gd_loss_timed(p, t, prm, 5.0 / sample_pairs, loss, gp, cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or reps >= 64:
            break
    # the same code on ONE thread, on a 1/16 slice (SURVEY.md §8d asks for the scalar single-thread figure as well)
    n1 = max(sample_pairs // 16, 1)
    t1 = time.perf_counter()
    for prm in prms:
        oracle.gd_loss_timed(p[:n1], t[:n1], prm, 5.0 / sample_pairs, loss[:n1], gp[:n1], 1)
    dt1 = time.perf_counter() - t1
    out = {'value': round(3 * sample_pairs * reps / dt / 1e6, 3), 'unit': 'M box-pairs/s', 'cores': cores, 'kind': 'port',
           'single_thread_value': round(3 * n1 / dt1 / 1e6, 3), 'cpu_model': cpu_model(),
           'sample': f'{reps} pass(es) over {sample_pairs} pairs x 3 losses fwd+grad, fp32 C oracle '
                     f'(oracle/gd_oracle.c), OpenMP {cores} threads, {dt:.2f} s wall; single thread: {n1} pairs x 3 '
                     f'losses in {dt1:.2f} s'}
    # the reference's CPU path as PyTorch runs it: the literal op chain (= the reference's time), then the lean rewrite
    big = [n for n in (10_000_000,) if sample_pairs >= 4_000_000]     # rehearsals with a small --cpu-sample stay small
    out['torch_chain'] = torch_chain_baseline(seed, cores, True, [1000, min(sample_pairs, 1_000_000)] + big)
    out['torch_chain_lean'] = torch_chain_baseline(seed, cores, False, [min(sample_pairs, 1_000_000)], budget_s=10.0)
    out['product_cpu'] = product_cpu_baseline(min(sample_pairs, 4_000_000), seed, cores)
    return out


def product_cpu_baseline(sample_pairs, seed, cores):
    """The PRODUCT's own CPU path on the same workload: GDLoss on CPU tensors -> gd3d_loss_fused_cpu (the fused kernel's
    per-pair math compiled for the host, csrc/gd3d_cpu.cpp; what a machine without a GPU runs), forward + backward through
    the module surface, with all granted cores and with one thread."""
    import mmdet3d_gaussian_amd as amd
    pred, tgt = synthetic_pairs(sample_pairs, seed, torch.device('cpu'))
    mods = [amd.build_loss(dict(type='GDLoss', loss_type=lt, fun='log1p', tau=1.0, alpha=1.0, reduction='mean', loss_weight=5.0))
            for lt in LOSSES]
    res = {'unit': 'M box-pairs/s'}
    keep = torch.get_num_threads()
    try:
        for label, nt, n in ((f'threads_{cores}', cores, sample_pairs), ('threads_1', 1, max(sample_pairs // 8, 1))):
            torch.set_num_threads(nt)
            p = pred[:n].clone().requires_grad_(True)
            for m in mods:      # untimed pass: page in, start the thread team
                m(p[:100_000], tgt[:100_000]).backward()
            reps, t0 = 0, time.perf_counter()
            while True:
                for m in mods:
                    p.grad = None
                    m(p, tgt[:n]).backward()
                reps += 1
                if time.perf_counter() - t0 >= 2.0 or reps >= 16:
                    break
            res[label] = round(3 * n * reps / (time.perf_counter() - t0) / 1e6, 3)
    finally:
        torch.set_num_threads(keep)
    res['sample'] = (f'{sample_pairs} pairs x 3 losses, GDLoss forward + backward on CPU tensors (gd3d_loss_fused_cpu, std::thread '
                     f'team of torch.get_num_threads()); single thread on {max(sample_pairs // 8, 1)} pairs')
    return res


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def torch_chain_baseline(seed, cores, literal, sizes, budget_s=45.0):
    """SURVEY.md §8d CPU baseline (2): the reference's CPU path as an eager PyTorch op chain with autograd, fp32, on this host with
    torch.set_num_threads(1) and (all granted cores), forward + backward of the three losses.

    literal=True : oracle/gd_torch.py `literal_gd_loss` — the reference's OWN chain op for op (stack / diag_embed / bmm on
                   (N,2,2), the weighted_loss wrapper: 103 / 110 / 140 top-level ATen ops forward, the same count as the
                   reference module; reproduces the reference's fp32 golden values bit for bit, tests/test_oracle_torch.py).
                   Its time IS the reference's time on this host.  -> cpu_baseline.torch_chain
    literal=False: `gd_loss` — an entry-form rewrite without (N,2,2) matrices or bmm: 2.2-2.6x faster than the reference's
                   chain, NOT its shape; kept for continuity with earlier rounds.  -> cpu_baseline.torch_chain_lean
    sizes: pairs per pass, ascending (1 k, 1 M, 10 M).  Small sizes repeat until ~0.3 s.  Bounded: a (size,
    threads) cell is skipped (recorded as null with the reason) once the chain's share of the budget is spent."""
    from oracle import gd_torch
    fn = gd_torch.literal_gd_loss if literal else gd_torch.gd_loss
    res = {'unit': 'M box-pairs/s'}
    keep = torch.get_num_threads()
    spent, skipped = 0.0, []
    try:
        for n in sizes:
            pred, tgt = synthetic_pairs(n, seed, torch.device('cpu'))
            cell = {}
            for label, nt in ((f'threads_{cores}', cores), ('threads_1', 1)):
                if spent >= budget_s:
                    cell[label] = None
                    skipped.append(f'{n} pairs {label}')
                    continue
                torch.set_num_threads(nt)
                p = pred.clone().requires_grad_(True)
                m = min(n, 50_000)
                for lt in LOSSES:   # untimed warm pass on a slice
                    fn(p[:m], tgt[:m], lt, fun='log1p', tau=1.0, loss_weight=5.0).backward()
                reps, t0 = 0, time.perf_counter()
                while True:
                    for lt in LOSSES:
                        p.grad = None
                        fn(p, tgt, lt, fun='log1p', tau=1.0, loss_weight=5.0).backward()
                    reps += 1
                    dt = time.perf_counter() - t0
                    if dt >= 0.3 or reps >= 200:
                        break
                spent += dt
                cell[label] = round(3 * n * reps / dt / 1e6, 3)
            res[f'pairs_{n}'] = cell
            del pred, tgt
    finally:
        torch.set_num_threads(keep)
    which = ('the reference\'s own op chain restated op for op (oracle/gd_torch.py literal_gd_loss: (N,2,2) stack / diag_embed / bmm + '
             'weighted_loss wrapper)' if literal else 'entry-form rewrite without bmm (oracle/gd_torch.py gd_loss), 2.2-2.6x faster than '
             'the reference\'s chain')
    res['sample'] = (f'{" / ".join(str(n) for n in sizes)} pairs x 3 losses, forward + autograd backward, fp32, eager PyTorch-CPU: {which}; '
                     f'{spent:.1f} s' + (f'; skipped (time bound): {", ".join(skipped)}' if skipped else ''))
    if literal:
        res['top_level_aten_ops_forward'] = {'gwd3d': 103, 'kld3d': 110, 'bd3d': 140, 'reference_module_same_count': [103, 110, 140],
                                             'survey_8a': [106, 113, 143]}
    return res


def pmc_child(pairs):
    """`--pmc-child`: what the parent runs under `rocprofv3 --pmc ...` to count HBM bytes: three launches of each fused kernel
    at the benchmark size, nothing else worth counting."""
    import mmdet3d_gaussian_amd as amd
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    amd.load_library()
    pred, tgt = synthetic_pairs(pairs, 0, dev)
    pred.requires_grad_(True)
    for lt in LOSSES:
        mod = amd.build_loss(dict(type='GDLoss', loss_type=lt, fun='log1p', tau=1.0, alpha=1.0, reduction='mean', loss_weight=5.0))
        for _ in range(3):
            mod(pred, tgt)
    torch.cuda.synchronize(dev)


def measure_traffic(pairs, timeout_s=120):
    """HBM bytes per launch of every fused kernel, counted in THIS run: two `rocprofv3 --pmc` passes (FETCH_SIZE and
    WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots) over a child process that launches the
    kernels at the benchmark size.  Corrections as that guide's HBM section prescribes: both counters are in units of
    1024 bytes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B per lane) coalesced streaming read.
    Returns ({loss: bytes}, note); ({}, reason) when rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return {}, 'rocprofv3 not found'
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return {}, 'this process already runs under a profiler (no nested rocprofv3)'
    names = dict(zip(('0', '1', '2'), LOSSES))
    raw = {}
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = tempfile.mkdtemp(prefix='gd3d_pmc_', dir='/tmp')
        try:
            # the program itself after `--` (no env / shell hop: the profiler's preloaded library initialises the GPU)
            cmd = [prof, '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__),
                   '--pmc-child', '--pairs', str(pairs)]
            r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s, env=env, cwd='/tmp')
            if r.returncode != 0:
                return {}, f'rocprofv3 --pmc {counter} exited with {r.returncode}: {r.stderr[-200:]}'
            vals = {}
            for path in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        name = row.get('Kernel_Name', '')
                        if 'fused_kernel<' in name and row.get('Counter_Name') == counter:
                            lt = names.get(name.split('fused_kernel<')[1].split(',')[0].strip())
                            if lt:
                                vals.setdefault(lt, []).append(float(row['Counter_Value']))
            if not vals:
                return {}, f'no {counter} rows for the fused kernels'
            raw[counter] = {lt: sum(v) / len(v) for lt, v in vals.items()}
        except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
            return {}, f'{type(e).__name__}: {str(e)[:160]}'
        finally:
            shutil.rmtree(out, ignore_errors=True)
    res = {lt: int(round(2 * raw['FETCH_SIZE'][lt] * 1024 + raw['WRITE_SIZE'][lt] * 1024))
           for lt in LOSSES if lt in raw['FETCH_SIZE'] and lt in raw['WRITE_SIZE']}
    return res, ('measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over a child process '
                 'launching the same kernels at the same size; bytes = 2 x FETCH_SIZE x 1024 (gfx950: wide coalesced reads are '
                 'tallied at half) + WRITE_SIZE x 1024')


# Environment every rank needs BEFORE its first HIP call, whoever launched it (this file's self_launch, the driver's
# `python -m torch.distributed.run ...`, a cluster scheduler).  The reference's launcher sets its rendezvous in
# tools/dist_train.sh:8-9 and nothing else; on these hosts RCCL additionally needs dmabuf IPC (the legacy IPC mode fails with
# `hipIpcGetMemHandle: invalid argument`).
RCCL_ENV_DEFAULTS = {'HSA_ENABLE_IPC_MODE_LEGACY': '0'}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_command(n, argv, port):
    """The command the driver itself uses for N > 1 (one rank per GPU over RCCL, rendezvous on 127.0.0.1)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, visible, cmd=None, out=None, err=None, need_gpus=True):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (this process has not touched
    the GPU and never replaces itself), relay rank 0's JSON line to stdout, everything else the children print to
    stderr, and return their exit status (non-zero if any rank failed, or if no result line appeared).
    Reference counterpart: /root/reference/tools/dist_train.sh:8-9."""
    import subprocess
    out = out or sys.stdout
    err = err or sys.stderr
    if need_gpus and visible < n:
        print(f'bench.py: {n} GPUs requested, {visible} visible', file=err, flush=True)
        return 2
    cmd = cmd or launch_command(n, argv, free_port())
    env = dict(os.environ)
    for k, v in RCCL_ENV_DEFAULTS.items():   # the ranks set them again themselves (main): whichever launcher starts them
        env.setdefault(k, v)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=err, text=True, env=env)
    lines = 0
    for line in proc.stdout:
        is_result = False
        if line.lstrip().startswith('{'):
            try:
                is_result = 'metric' in json.loads(line)
            except ValueError:
                is_result = False
        if is_result:
            lines += 1
            out.write(line)
            out.flush()
        else:
            err.write(line)
            err.flush()
    rc = proc.wait()
    if rc != 0:
        print(f'bench.py: the {n}-rank job exited with status {rc}', file=err, flush=True)
        return rc if 0 < rc < 256 else 1
    if lines != 1:
        print(f'bench.py: expected one result line from rank 0, saw {lines}', file=err, flush=True)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--pairs', type=int, default=10_000_000, help='pairs per GPU (default: BASELINE config 3)')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: --pairs is the TOTAL, split into contiguous row ranges over the ranks '
                         '(SURVEY.md §8e; default is weak scaling, --pairs per GPU)')
    ap.add_argument('--cpu-sample', type=int, default=10_000_000, help='pairs in the CPU baseline sample (0 = skip)')
    ap.add_argument('--prewarm', type=float, default=1.0,
                    help='seconds of untimed steps before the W warmup steps (clock ramp of a cold GPU; 0 = off)')
    ap.add_argument('--graph', action='store_true', help='replay a hipGraph of the step (4-20 us per step slower than eager launches for the plain-backward step, with or without the collective)')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly (the default)')
    ap.add_argument('--separate-inputs', action='store_true',
                    help='(the default since round 6) one torch allocation per input array — target + one prediction leaf per loss — '
                         'as a caller of the reference has them (pred and target are separate torch.cat outputs)')
    ap.add_argument('--arena', action='store_true',
                    help="rounds 4-5's headline form: the four input arrays are row ranges of ONE allocation (never collides in the "
                         'memory system, DESIGN.md 5.3); by default that form is timed in the third region and reported as '
                         '`value_one_allocation`')
    ap.add_argument('--no-standins', action='store_true',
                    help='skip the parity / nms / head keys (bench_standins.py: ~10 s after the timed regions, N = 1 only)')
    ap.add_argument('--unit-grad', action='store_true',
                    help="main region = rounds 3-4's step: one torch.autograd.backward over the three losses with the library's unit "
                         'gradient (gd_loss.unit_grad) instead of the plain (l0 + l1 + l2).backward() every reference caller runs')
    ap.add_argument('--sum-backward', action='store_true', help=argparse.SUPPRESS)   # the default since round 5
    ap.add_argument('--event-every', type=int, default=5,
                    help='eager launches: bind a HIP event pair to the fused dispatches of every N-th step of the timed region '
                         '(1 = every step as in round 2, which costs ~5 us of GPU time per timed launch; 0 = none in the '
                         'region, the kernel durations then come from an eager pass right after it)')
    ap.add_argument('--no-traffic', action='store_true',
                    help='skip the two rocprofv3 --pmc passes that count HBM bytes per launch (roofline.traffic; N = 1 only, ~40 s)')
    ap.add_argument('--device', choices=('cuda', 'cpu'), default='cuda',
                    help="cpu: rehearse the rank loop on CPU tensors through GDLoss's _cpu twins (no roofline claim)")
    ap.add_argument('--backend', choices=('nccl', 'gloo'), default=None, help='torch.distributed backend (default: nccl = RCCL on cuda, gloo on cpu)')
    ap.add_argument('--plain-steps', type=int, default=-1,
                    help='steps of the second region that ends in a plain (l0+l1+l2).backward() (default: half of --steps, at least 10; 0 = skip)')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    for k, v in RCCL_ENV_DEFAULTS.items():   # before anything touches the GPU, on every launch path
        os.environ.setdefault(k, v)
    args.separate_inputs = not args.arena
    on_gpu = args.device == 'cuda'
    backend = args.backend or ('nccl' if on_gpu else 'gloo')

    if args.pmc_child:
        pmc_child(args.pairs)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: be the launcher (device_count() does not initialise the GPU)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], torch.cuda.device_count() if on_gpu else 0, need_gpus=on_gpu))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # stdout carries ONE line: the result.  Native libraries print there too (RCCL's five-line version banner at communicator
    # creation, on every rank): send file descriptor 1 to stderr for the duration of the run and keep the real one for the line.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X (the metric is a GPU figure; `--device cpu --backend gloo` only rehearses the rank loop)')
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    else:
        dev = torch.device('cpu')
        torch.set_num_threads(max(1, usable_cores() // max(world, 1)))
    use_dist = world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ)  # torchrun, even with 1 rank
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if on_gpu:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    use_graph = on_gpu and args.graph and not args.no_graph
    fail_rank = int(os.environ.get('GD3D_BENCH_FAIL_RANK', '-1'))   # tests: this rank dies after the warmup steps

    def device_sync():
        if on_gpu:
            torch.cuda.synchronize(dev)

    import mmdet3d_gaussian_amd as amd
    from mmdet3d_gaussian_amd import _lib as amd_lib, gd_loss as gdl
    amd.load_library()
    host_glue = amd_lib.host_glue()

    _, n = shard_rows(args.pairs, rank, world, args.strong)
    pred0, tgt = synthetic_pairs(n, seed=rank, device=dev)
    # one leaf per loss (same values): the step runs ONE autograd backward over the three losses, as a training step does
    # with its loss dict, and separate leaves keep autograd from adding 2 x 280 MB gradient accumulations that are not
    # part of the metric.  (One backward per loss costs ~60 us of autograd-engine thread hand-off each: 3 x that made
    # the eager step host-bound on boxes with a slow host, measured in round 2.)
    if args.separate_inputs:
        preds = {lt: pred0.clone().requires_grad_(True) for lt in LOSSES}
    else:
        # The four input arrays are row ranges of ONE allocation.  Streams read from separately allocated 280 MB buffers
        # collide in the memory system on some draws of their physical placement (a 2-read-1-write pass over three such
        # buffers: 122-134 us by triple; over three ranges of one allocation: 121-126 us on every draw, whether the
        # allocation is made first, after others, or after others were freed: profiles/r04_placement_scan.txt part 3).
        # `--separate-inputs` restores one torch allocation per array.
        arena = torch.empty((len(LOSSES) + 1) * n, 7, dtype=torch.float32, device=dev)
        arena[:n].copy_(tgt)
        tgt = arena[:n]
        preds = {}
        for k, lt in enumerate(LOSSES):
            view = arena[(k + 1) * n:(k + 2) * n]
            view.copy_(pred0)
            preds[lt] = view.detach().requires_grad_(True)
    del pred0
    cur = {'tgt': tgt}   # the arrays the step reads (the third region swaps in the other allocation form's)
    mods = {lt: amd.build_loss(dict(type='GDLoss', loss_type=lt, fun='log1p', tau=1.0, alpha=1.0,
                                    reduction='mean', loss_weight=5.0)) for lt in LOSSES}
    events = {lt: [] for lt in LOSSES}
    events_alt = {lt: [] for lt in LOSSES}   # fused-dispatch timers of the third region (the other allocation form)
    ev_sink = {'d': events}
    last = {}

    unit = [gdl.unit_grad(dev)] * len(LOSSES)
    rank_id = torch.full((), float(rank), dtype=torch.float32, device=dev)

    main_plain = not args.unit_grad

    def compute(record, plain=None):
        """gwd3d, kld3d, bd3d: GDLoss forward each, then ONE autograd backward over the three losses (every loss's fused
        kernel has already produced its gradient).  `plain` (the main region's form, round 5): end in `(l0 + l1 + l2).backward()` as
        the reference's callers do (tools/train.py:213-220 -> mmcv's OptimizerHook): torch fills a ones tensor and each loss's
        node launches one early-exit `grad_finish`.  Otherwise: torch.autograd.backward with the library's unit gradient as upstream
        gradient, recognised by address — backward launches nothing.  Returns the 3 detached loss scalars (N > 1: stacked into the
        (3,) tensor the per-step collective sends — inside the captured graph)."""
        if plain is None:
            plain = main_plain
        losses_ = []
        for lt in LOSSES:
            gdl.PROFILE_EVENTS = ev_sink['d'][lt] if (record and on_gpu) else None
            preds[lt].grad = None
            losses_.append(mods[lt](preds[lt], cur['tgt']))
        gdl.PROFILE_EVENTS = None
        if plain:
            (losses_[0] + losses_[1] + losses_[2]).backward()
        else:
            torch.autograd.backward(losses_, grad_tensors=unit)
        outs = [l.detach() for l in losses_]
        # N > 1: the per-step payload is (4,): the three shard losses and THIS RANK'S ID — so that what the collective returns
        # names the ranks that took part (`config.ranks_seen` is read from the last step's gather, not from WORLD_SIZE)
        return torch.stack(outs + [rank_id]) if use_dist else outs

    def make_alt():
        """The OTHER allocation form's arrays (third region, N = 1 on the GPU).  Made before the clocks are ramped (an allocation of
        1.1 GB between two timed regions stalls the host for milliseconds, the GPU idles, drops its clocks, and the region that
        follows runs slow) but AFTER the headline form's own first steps: the bench's extra arrays must not sit between the
        headline form's inputs and its gradient buffers in the allocator's sequence (measured: with them in between, every box
        drew the slow placement for the one-allocation-per-array form, 135-140 us per fused kernel against 129-133)."""
        if not (on_gpu and world == 1 and not use_dist and not args.graph):
            return None
        src_p, src_t = preds[LOSSES[0]].detach(), cur['tgt']
        if args.separate_inputs:
            arena2 = torch.empty((len(LOSSES) + 1) * n, 7, dtype=torch.float32, device=dev)
            arena2[:n].copy_(src_t)
            alt_preds = {}
            for k, lt in enumerate(LOSSES):
                view = arena2[(k + 1) * n:(k + 2) * n]
                view.copy_(src_p)
                alt_preds[lt] = view.detach().requires_grad_(True)
            return arena2[:n], alt_preds
        return src_t.clone(), {lt: src_p.clone().requires_grad_(True) for lt in LOSSES}

    graph = None
    graph_note = None
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    compute(False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            # thread_local: calls made by other threads (e.g. the RCCL watchdog) must not invalidate the capture
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                graph_outs = compute(False)
        except Exception as e:  # noqa: BLE001 — a failed capture must not cost the measurement: launch eagerly instead
            graph = None
            graph_note = f'hipGraph capture failed ({type(e).__name__}: {str(e)[:120]}); eager launches'
            torch.cuda.synchronize(dev)
        if use_dist:  # every rank must take the same path (the collective pattern is identical either way, but keep
            ok = torch.tensor([1 if graph is not None else 0], device=dev)   # the launch mode reported by rank 0 true)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() == 0 and graph is not None:
                graph = None
                graph_note = 'hipGraph capture failed on another rank; eager launches'

    def step(record, plain=None):
        if graph is None:
            outs = compute(record, plain)
        else:
            graph.replay()   # nothing else enters the stream: no event brackets (marker packets) in the timed region
            # the collective reads its input on RCCL's stream while the next replay may already run: give it a private
            # copy of the graph's static output (12 bytes) instead of the buffer the next replay writes
            outs = graph_outs.clone() if use_dist else graph_outs
        if use_dist:  # one tiny collective per step: (3,) shard losses -> (world, 3)
            last['pending'] = amd.sharded.gather_shard_losses(outs, async_op=True)
        else:
            last['outs'] = outs

    def sync_all():
        device_sync()
        if use_dist:
            dist.barrier()
            device_sync()

    # a fresh box starts at its idle clocks and a 30 ms benchmark can be over before they have ramped: run untimed
    # steps for a fixed wall time first (same step function; the W warmup steps still follow, then exactly K timed ones)
    # The pre-warm steps are launched exactly like the timed ones (event pairs recorded when the timed region records
    # them, then dropped): the first event pairs created after a few thousand event-free launches cost the host ~100 us
    # each on this ROCm (round 2 measurement), which would make the timed region host-bound.
    # Which steps carry event pairs: every `--event-every`-th one.  A dispatch with events bound to it costs ~5 us of GPU time
    # more than a plain one on this ROCm (round 3: the same eager step ran 13.7 us over the sum of its three fused kernels
    # without events, tools/step_variants.py, and 28.7 us over it with events on every launch), so timing EVERY launch in
    # the region taxes `value` by 3.5 %; every fifth step keeps the durations in-region and the tax under 1 %.
    every = max(args.event_every, 0) if on_gpu else 0
    tick = [0]

    def sampled():
        tick[0] += 1
        return every > 0 and tick[0] % every == 0

    # No cyclic garbage collection between here and the end of the timed regions: a generation-2 pass of the Python collector
    # falling into a region stalls the host for ~0.14 s (measured, round 5: tools/scratch note in profiles/r05_step_variants.jsonl),
    # the GPU idles, its clocks drop, and the steps that follow run 5-20 % slow for tens of steps — round 5's first runs caught one
    # in the second region (0.485 ms per step instead of 0.406).  Collect NOW (before the pre-warm ramps the clocks), then keep the
    # collector off until the regions are done; reference counting still frees every tensor at once.
    import gc
    gc.collect()
    gc.disable()
    if on_gpu:
        # the torch kernels the parity sample takes right after the main region (arange, index_select): first use loads their code
        # objects — tens of ms of host time during which the GPU would idle and drop its clocks in front of the second region
        _w = torch.arange(0, 64, 2, device=dev)
        torch.zeros(64, 7, device=dev).index_select(0, _w)
        del _w
    # Both backward forms once, BEFORE the clocks are ramped: the first torch.autograd.backward(..., grad_tensors=...) of a process
    # spends ~0.14 s of host time in one-time work inside torch (measured, round 5); left to happen in the second region's warm-up
    # steps it idles the GPU, the clocks drop, and that region and the next run 10-20 % slow (0.46-0.49 ms per step instead of 0.41).
    alt = None
    if graph is None:
        for form in (not main_plain, main_plain):
            for _ in range(2):
                step(False, plain=form)
        device_sync()
        alt = make_alt()
        if alt is not None:   # one step on the other allocation form's arrays too: its gradient buffers exist before the clocks ramp
            held0 = (cur['tgt'], dict(preds))   # (AFTER the main form's first steps: the bench's extra arrays must not change what the
            cur['tgt'] = alt[0]                 #  headline form's own allocations draw)
            preds.update(alt[1])
            step(False)
            device_sync()
            cur['tgt'] = held0[0]
            preds.update(held0[1])
        if use_dist and not on_gpu:
            last['pending'].result()
    prewarm_blocks = []
    if args.prewarm > 0:
        # at least `--prewarm` seconds; on the GPU then on until two consecutive 20-step blocks agree to 1.5 % (at most 5 x as
        # long): insurance against a box that is still ramping.  Nothing slow may sit between the warm-up steps and the timed ones:
        # a gc.collect() placed there (tried, against host stalls inside the region) leaves the GPU idle for ~0.1 s, its clocks drop,
        # and the 20 timed steps that follow run 10-15 % slow (fused kernels 139-150 us instead of 130; same memory, same three
        # gradient blocks: profiles/r04_bench_three_boxes.txt).
        t_pre = time.perf_counter()
        while True:
            tb = time.perf_counter()
            for _ in range(20 if on_gpu else 1):
                step(sampled())
            device_sync()
            prewarm_blocks.append(time.perf_counter() - tb)
            for lt in LOSSES:
                events[lt].clear()
            spent = time.perf_counter() - t_pre
            settled = len(prewarm_blocks) >= 2 and abs(prewarm_blocks[-1] - prewarm_blocks[-2]) <= 0.015 * prewarm_blocks[-2]
            more = spent < args.prewarm or (on_gpu and not settled and spent < 5.0 * args.prewarm)
            if use_dist:
                # every step carries a collective: all ranks must leave the loop after the SAME number of blocks (a clock- or
                # settle-based decision taken per rank would leave the ranks with different collective counts and the job hanging
                # in its last gather) — go on while ANY rank wants more
                flag = torch.tensor([1 if more else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                more = bool(flag.item())
            if not more:
                break
    for _ in range(args.warmup):
        step(sampled())
    if rank == fail_rank:
        os._exit(3)
    sync_all()
    for lt in LOSSES:
        events[lt].clear()
    tick[0] = -1 if every > 0 else 0   # the first timed step is a sampled one
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(sampled())
    host_enqueue = time.perf_counter() - t0   # host time to enqueue all steps (GPU-bound iff this < elapsed)
    device_sync()
    if use_dist and not on_gpu:
        last['pending'].result()                # gloo: the last step's gather is part of the step
    elapsed = time.perf_counter() - t0

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
        device_sync()
        return tt.item()
    elapsed = max_over_ranks(elapsed)
    # rows of the gradients the LAST TIMED STEP left in HBM, for the parity key (bench_standins.parity): taken now, before the
    # later regions and the copy probe write those buffers again
    grad_rows = None
    if on_gpu and rank == 0 and world == 1 and not args.no_standins and not args.strong and n > 0:
        sample_idx = torch.arange(0, n, max(1, n // 200_000), device=dev)
        grad_rows = {lt: preds[lt].grad.index_select(0, sample_idx) for lt in LOSSES}
    timing = (f'HIP event pair bound to the fused dispatches of every {every}-th step inside the timed region '
              '(hipExtLaunchKernel start/stop events: begin/end timestamps of the dispatch itself, no marker packets)')
    replay_ms = None
    if on_gpu and graph is None and not any(events[lt] for lt in LOSSES):   # --event-every 0: an eager pass right after the region
        for it in range(25):
            compute(True)
            if it == 4:
                torch.cuda.synchronize(dev)
                for lt in LOSSES:
                    events[lt].clear()
        torch.cuda.synchronize(dev)
        timing = ('HIP event pair bound to every fused dispatch (hipExtLaunchKernel start/stop events), eager pass of 20 steps run '
                  'right after the timed region (which carried no events)')
    if graph is not None:  # events cannot be bound inside a captured graph on ROCm: eager pass right after
        for it in range(25):
            compute(True)
            if it == 4:   # the first passes re-warm the eager path (event creation, allocator); keep the last 20
                torch.cuda.synchronize(dev)
                for lt in LOSSES:
                    events[lt].clear()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(min(args.steps, 10)):   # GPU time of the bare replays (no collective), after the timed region
            graph.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        replay_ms = e0.elapsed_time(e1) / min(args.steps, 10)
        timing = ('HIP event pair bound to every fused dispatch (hipExtLaunchKernel start/stop events), eager pass run '
                  'right after the graph-replayed timed region')

    # dominant-kernel durations from the HIP events recorded inside the timed region
    kern_ms, kern_med, kern_min = {}, {}, {}
    for lt in LOSSES:
        d = sorted(tm.elapsed_ms() for tm in events[lt])
        kern_ms[lt] = sum(d) / max(len(d), 1)
        kern_med[lt] = d[len(d) // 2] if d else 0.0
        kern_min[lt] = d[0] if d else 0.0
    # Second, shorter region: the same step ending in a plain `(l0 + l1 + l2).backward()` — what every reference caller does
    # (mmcv's OptimizerHook, set up from tools/train.py:213-220).  Eager launches only (a captured graph would need its own
    # capture of the other backward); same barrier + synchronize brackets and MAX over ranks as the main region.
    plain_steps = args.plain_steps if args.plain_steps >= 0 else max(10, args.steps // 2)
    plain_elapsed = None      # elapsed time of the second region, which runs the OTHER backward form than the main region
    if graph is None and plain_steps > 0:
        for _ in range(5):
            step(False, plain=not main_plain)
        sync_all()
        tp = time.perf_counter()
        for _ in range(plain_steps):
            step(False, plain=not main_plain)
        device_sync()
        if use_dist and not on_gpu:
            last['pending'].result()
        plain_elapsed = max_over_ranks(time.perf_counter() - tp)

    # Third region (N = 1, eager, GPU): the headline step again on the OTHER input allocation form.  Main region = one torch
    # allocation per array (what a caller of the reference has; its value moves with the physical placement the process draws,
    # DESIGN.md 5.3) -> here the four arrays are row ranges of one allocation (`value_one_allocation`); with --arena the roles swap.
    sep_elapsed = None
    if graph is None and not use_dist and on_gpu and plain_steps > 0 and alt is not None:
        held = (cur['tgt'], dict(preds))
        cur['tgt'] = alt[0]
        for lt in LOSSES:
            preds[lt].grad = None
            preds[lt] = alt[1][lt]
        for _ in range(5):
            step(False)
        sync_all()
        ev_sink['d'] = events_alt      # every `--event-every`-th step of this region carries event pairs too
        tick[0] = -1 if every > 0 else 0
        tp = time.perf_counter()
        for _ in range(plain_steps):
            step(sampled())
        device_sync()
        sep_elapsed = time.perf_counter() - tp
        ev_sink['d'] = events
        for lt in LOSSES:     # back to the arrays of the main region (the copy probe and the parity check below run on them)
            preds[lt].grad = None
            preds[lt] = held[1][lt]
        cur['tgt'] = held[0]
        step(False)
        device_sync()

    kern_ms_alt = {}
    for lt in LOSSES:
        d = [tm.elapsed_ms() for tm in events_alt[lt]]
        if d:
            kern_ms_alt[lt] = sum(d) / len(d)
    first_row, _ = shard_rows(args.pairs, rank, world, args.strong)
    mine = torch.tensor([float(rank), float(first_row), float(n)] + [kern_ms[lt] for lt in LOSSES], dtype=torch.float64, device=dev)
    gc.enable()
    if use_dist:
        total, per_rank = last['pending'].result()   # (4,), (world, 4): the last timed-path step's gather
        per_rank = per_rank.double().cpu()
        vals = (total[:3] / world).tolist()           # mean over ranks of per-rank means (equal shard sizes)
        ranks_seen = sorted(int(round(r)) for r in per_rank[:, 3].tolist())
        per_rank_loss = [[round(v, 6) for v in row[:3]] for row in per_rank.tolist()]
        # one more call of the SAME collective routine, after the timed regions: every rank's id, row range and fused-kernel means
        facts = amd.sharded.gather_shard_losses(mine, async_op=False).result()[1].cpu()
    else:
        vals = [v.item() for v in last['outs']]
        ranks_seen, per_rank_loss, facts = [0], [[round(v, 6) for v in vals]], mine.reshape(1, -1).cpu()
    losses = dict(zip(LOSSES, vals))
    facts = facts[facts[:, 0].argsort()]
    rank_rows = [[int(r[1]), int(r[1]) + int(r[2])] for r in facts.tolist()]      # [first, end) in the global numbering
    kern_by_rank = facts[:, 3:]
    rccl_version = None
    if on_gpu and use_dist and backend == 'nccl':
        try:
            rccl_version = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            rccl_version = f'unavailable ({type(e).__name__})'

    # The box's own ceiling for this access mix, measured in this process right after the timed region: z = x + y with
    # nontemporal 16-byte accesses over the fused kernel's OWN buffers (pred and target read, the gradient written:
    # 2 x 280 MB in, 280 MB out at 10 M pairs), dispatch-bound events as for the fused kernel.
    # (measured on every loss's own buffer triple: which three buffers are combined moves both the kernel and the probe by
    #  up to 5 %, DESIGN.md §5.3, so the ceiling that belongs to the dominant kernel is the one of ITS buffers)
    probe_by_loss = {}
    if rank == 0 and on_gpu and n > 0 and (7 * n) % 4 == 0:
        lib = amd.load_library()
        stream = torch.cuda.current_stream().cuda_stream
        for lt in LOSSES:
            x, y, z = preds[lt].detach(), cur['tgt'], preds[lt].grad
            if z is None:
                continue
            tms = []
            for it in range(25):
                tm = gdl.DispatchTimer()
                rc = lib.gd3d_probe_stream(x.data_ptr(), y.data_ptr(), z.data_ptr(), 7 * n, stream, tm.start, tm.stop)
                assert rc == 0, rc
                if it >= 5:
                    tms.append(tm)
            torch.cuda.synchronize(dev)
            d = sorted(t.elapsed_ms() for t in tms)
            probe_by_loss[lt] = sum(d) / len(d)
    dom_probe = max(LOSSES, key=lambda k: kern_ms[k])
    probe_ms = probe_by_loss.get(dom_probe)

    if rank == 0:
        value = job_value(args.pairs, world, args.strong, args.steps, elapsed)
        dom = max(LOSSES, key=lambda k: kern_ms[k])          # slowest of the three fused kernels
        dom_s = kern_ms[dom] * 1e-3
        achieved = BYTES_PER_PAIR * n / dom_s / 1e9 if dom_s > 0 else 0.0
        step_gbps = BYTES_PER_PAIR * n * len(LOSSES) / (elapsed / args.steps) / 1e9   # per GPU: every rank runs n pairs x 3 losses a step
        moved = MOVED_BYTES_PER_PAIR * n / dom_s / 1e9 if dom_s > 0 else 0.0
        ceiling = MOVED_BYTES_PER_PAIR * n / (probe_ms * 1e-3) / 1e9 if probe_ms else None
        traffic, traffic_by_loss, traffic_note = None, {}, None
        if world == 1 and on_gpu and not args.no_traffic and not args.strong:
            traffic_by_loss, traffic_note = measure_traffic(n)
            traffic = traffic_by_loss.get(dom)
        if traffic is None and on_gpu:   # the counters could not be collected here: fall back to the committed collection, and say so
            why = traffic_note if (world == 1 and on_gpu and not args.no_traffic) else 'not collected in this run (--no-traffic, --device cpu or N > 1)'
            tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.isfile(tpath):
                try:
                    with open(tpath) as f:
                        traffic = json.load(f).get(dom, {}).get('hbm_bytes_per_launch')
                except Exception:  # noqa: BLE001
                    traffic = None
            traffic_note = f'static: PMC passes of an earlier run of this command (profiles/traffic.json); {why}'
        # the two backward forms: the main region ran one (`value`), the second region the other
        other_value = round(job_value(args.pairs, world, args.strong, plain_steps, plain_elapsed), 2) if plain_elapsed else None
        other_ms = round(plain_elapsed / plain_steps * 1e3, 4) if plain_elapsed else None
        other_frac = (round(BYTES_PER_PAIR * n * len(LOSSES) / (plain_elapsed / plain_steps) / 1e9 / HBM_PEAK_GBPS, 4)
                      if plain_elapsed else None)
        main_ms, main_frac = round(elapsed / args.steps * 1e3, 4), round(step_gbps / HBM_PEAK_GBPS, 4)
        other_alloc_value = round(job_value(args.pairs, world, args.strong, plain_steps, sep_elapsed), 2) if sep_elapsed else None
        other_alloc_ms = round(sep_elapsed / plain_steps * 1e3, 4) if sep_elapsed else None
        by_form = {'plain': (round(value, 2), main_ms, main_frac, args.steps), 'unit': (other_value, other_ms, other_frac, plain_steps if plain_elapsed else 0)}
        if not main_plain:
            by_form = {'plain': by_form['unit'], 'unit': by_form['plain']}
        line = {
            'metric': 'M box-pairs/sec (fwd+bwd) for GWD/KLD/BCD @10M pairs',
            # which step `value` times (round 5: the reference caller's form; rounds 3-4 headlined the unit-gradient form)
            'value_form': ('plain (l0 + l1 + l2).backward(), as every reference caller runs it (tools/train.py:213-220 -> mmcv OptimizerHook)'
                           if main_plain else 'torch.autograd.backward([l0, l1, l2], grad_tensors=[gd_loss.unit_grad] * 3)')
                          + ('; inputs are row ranges of one allocation' if not args.separate_inputs else '; one allocation per input array'),
            'value': round(value, 2), 'unit': 'M box-pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            # the same step ending in a plain (l0 + l1 + l2).backward(), second region of `plain_backward_steps` steps
            'value_plain_backward': by_form['plain'][0], 'ms_per_step_plain_backward': by_form['plain'][1],
            'plain_backward_steps': by_form['plain'][3],
            # the same step with the library's unit gradient handed to torch.autograd.backward (backward launches nothing)
            'value_unit_grad': by_form['unit'][0], 'ms_per_step_unit_grad': by_form['unit'][1], 'unit_grad_steps': by_form['unit'][3],
            # the two input allocation forms: one torch allocation per array (the main region unless --arena) and row ranges of ONE
            # allocation (third region, N = 1; with --arena the roles swap)
            'value_separate_inputs': round(value, 2) if args.separate_inputs else other_alloc_value,
            'ms_per_step_separate_inputs': main_ms if args.separate_inputs else other_alloc_ms,
            'value_one_allocation': other_alloc_value if args.separate_inputs else round(value, 2),
            'ms_per_step_one_allocation': other_alloc_ms if args.separate_inputs else main_ms,
            'config': {'workload': (f'{args.pairs} synthetic anchor x gt 7-dof box pairs in total, row ranges of {n} per GPU '
                                    if args.strong else
                                    f'{n} synthetic anchor x gt 7-dof box pairs per GPU ') + '(BASELINE configs[2]); '
                                   'step = gwd3d + kld3d + bd3d, each GDLoss forward + backward over the whole batch '
                                   + ('(one plain backward() of the summed losses' if main_plain else
                                      '(one torch.autograd.backward over the three losses, upstream gradients = gd_loss.unit_grad')
                                   + '; fun=log1p, tau=1, reduction=mean, loss_weight=5)',
                       'pairs_per_gpu': n, 'losses': list(LOSSES), 'parallelism': f'pair-sharded x{world}',
                       # which host glue above the C ABI made the calls (mmdet3d-gaussian_amd/_lib.py: Python autograd.Function +
                       # ctypes, or the optional C++ node; GD3D_HOST=python|cpp)
                       'host_glue': host_glue,
                       'launch': 'hipGraph replay' if graph is not None else (graph_note or 'eager'),
                       'prewarm_s': round(sum(prewarm_blocks), 2),
                       # Python's cyclic collector is OFF across the timed regions (one collect() before the pre-warm; reference
                       # counting still frees every tensor): a generation-2 pass inside a region stalls the host for ~0.14 s
                       'gc_disabled': True,
                       'input_allocation': 'one torch allocation per array' if args.separate_inputs else
                                           'target and the three prediction leaves are row ranges of one allocation',
                       'device': 'MI355X (HIP kernels)' if on_gpu else f'cpu (rehearsal: GDLoss _cpu twins, {torch.get_num_threads()} threads per rank; not the metric)',
                       'collective': (f'all_gather of (4,) = 3 shard losses + the rank id per step over {"RCCL" if backend == "nccl" else backend}, async') if use_dist else None,
                       # evidence that N ranks ran, from what the collective RETURNED (not from WORLD_SIZE): the rank ids and the per-rank
                       # losses of the last step's gather; per-rank fused-kernel means and row ranges from one more gather after the regions
                       'ranks_seen': ranks_seen, 'per_rank_loss': per_rank_loss,
                       'per_rank_kernel_ms': {'max': {lt: round(float(kern_by_rank[:, k].max()), 4) for k, lt in enumerate(LOSSES)},
                                              'min': {lt: round(float(kern_by_rank[:, k].min()), 4) for k, lt in enumerate(LOSSES)}},
                       'per_rank_rows': rank_rows if (args.strong or world > 1) else None,
                       'rccl_version': rccl_version,
                       'host_enqueue_ms_per_step': round(host_enqueue / args.steps * 1e3, 4),
                       'graph_replay_ms_per_step': round(replay_ms, 4) if replay_ms is not None else None},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBPS, 4),
                         # the whole timed step priced like the kernel (reduce launches and backward included)
                         'achieved_step': round(step_gbps, 1), 'frac_step': round(step_gbps / HBM_PEAK_GBPS, 4),
                         'frac_step_plain_backward': by_form['plain'][2], 'frac_step_unit_grad': by_form['unit'][2],
                         # the same step on the two input allocation forms (main region / third region; see value_separate_inputs)
                         'frac_step_separate_inputs': main_frac if args.separate_inputs else (
                             round(BYTES_PER_PAIR * n * len(LOSSES) / (sep_elapsed / plain_steps) / 1e9 / HBM_PEAK_GBPS, 4) if sep_elapsed else None),
                         'frac_step_one_allocation': (round(BYTES_PER_PAIR * n * len(LOSSES) / (sep_elapsed / plain_steps) / 1e9 / HBM_PEAK_GBPS, 4)
                                                      if sep_elapsed else None) if args.separate_inputs else main_frac,
                         # the dominant kernel on the other allocation form (event pairs of the third region), priced like `frac`
                         'kernel_ms_other_allocation': {k: round(v, 4) for k, v in kern_ms_alt.items()} or None,
                         ('frac_one_allocation' if args.separate_inputs else 'frac_separate_inputs'):
                             (round(BYTES_PER_PAIR * n / (max(kern_ms_alt.values()) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if kern_ms_alt else None),
                         # the whole step priced on the bytes the kernels really move (84 B/pair: no per-pair loss store under 'mean')
                         'frac_step_moved_bytes': round(step_gbps / HBM_PEAK_GBPS * MOVED_BYTES_PER_PAIR / BYTES_PER_PAIR, 4),
                         'traffic': traffic,
                         'traffic_source': traffic_note, 'traffic_by_loss': traffic_by_loss or None,
                         'kernel': f'gd3d::fused_kernel<{dom}>', 'bytes_per_pair': BYTES_PER_PAIR,
                         # what the kernel really moves under reduction=mean|sum (no per-pair loss store): 84 B/pair
                         'moved_bytes_per_pair': MOVED_BYTES_PER_PAIR, 'achieved_moved_GBps': round(moved, 1),
                         'frac_actual_bytes': round(moved / HBM_PEAK_GBPS, 4),
                         # z = x + y (nontemporal, 16 B/lane, 64-thread workgroups) over the DOMINANT kernel's own three buffers, same process, after the region
                         'copy_ceiling_GBps': round(ceiling, 1) if ceiling else None,
                         'copy_ceiling_ms': round(probe_ms, 4) if probe_ms else None,
                         'copy_ceiling_ms_by_loss': {k: round(v, 4) for k, v in probe_by_loss.items()},
                         'copy_ceiling_frac_of_peak': round(ceiling / HBM_PEAK_GBPS, 4) if ceiling else None,
                         'frac_of_ceiling': round(moved / ceiling, 4) if ceiling else None,
                         'timing': timing,
                         'kernel_ms': {k: round(v, 4) for k, v in kern_ms.items()},
                         'kernel_ms_median': {k: round(v, 4) for k, v in kern_med.items()},
                         'kernel_ms_min': {k: round(v, 4) for k, v in kern_min.items()},
                         'mpairs_per_s_kernel': {k: round(n / (v * 1e-3) / 1e6, 1) if v > 0 else None
                                                 for k, v in kern_ms.items()}},
            'loss_values': {k: round(v, 6) for k, v in losses.items()},
        }
        if not on_gpu:   # a rehearsal of the rank loop on host memory: no roofline claim
            line['roofline'] = None
        if grad_rows is not None:
            # SURVEY.md §8d "plus max-abs-error vs oracle" and BASELINE.md's NMS boxes/s, and the other BASELINE configs' stand-ins:
            # after the timed regions, outside them (bench_standins.py)
            import bench_standins
            t_st = time.perf_counter()
            for key, fn in (('parity', lambda: bench_standins.parity(amd, LOSSES, preds, cur['tgt'], n, 5.0, sample_idx, grad_rows)),
                            ('nms', lambda: bench_standins.nms(amd, dev)), ('head', lambda: bench_standins.head(amd, dev))):
                try:
                    line[key] = fn()
                except Exception as e:  # noqa: BLE001 — a failed stand-in must not cost the headline line; it is reported as failed
                    line[key] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
            line['standins_s'] = round(time.perf_counter() - t_st, 2)
        if args.cpu_sample > 0 and world == 1:   # reported baseline: rank 0 at N = 1 only
            line['cpu_baseline'] = cpu_baseline(args.cpu_sample, seed=0)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
