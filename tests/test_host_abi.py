"""CPU-side checks of the product package: the C-ABI library loads and exports exactly what
include/gd3d.h declares (no compute calls), and the host logic of GDLoss / registry mirrors the
reference module's constructor and argument handling.  No GPU needed."""
import ctypes
import os
import re

import pytest
import torch

import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(header):
    txt = open(os.path.join(ROOT, 'include', header)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|size_t|int64_t|int32_t)\s+((?:gd3d|rnms|riou|vox|eval|coder|center_infer|center_targets|anchor_infer|anchor_targets)_\w+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    """gd3d.h is the hot path's boundary and libgd3d.so exports exactly it; gd3d_extras.h is the frozen extras'
    (DESIGN_EXTRAS.md), exported by libgd3d_extras.so (round 6: two libraries; the extras are out of the §8 image)."""
    names = _header_functions('gd3d.h')
    assert len(names) >= 10, names
    lib = ctypes.CDLL(amd.lib_path())           # built by __graft_entry__.build() / first import
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/gd3d.h but not exported by libgd3d.so'
    assert sorted(_lib.SYMBOLS) == names, 'ctypes table and header disagree'
    extra = _header_functions('gd3d_extras.h')
    assert len(extra) >= 10 and not set(extra) & set(names)
    xlib = ctypes.CDLL(_lib.extras_path())
    for n in extra:
        assert hasattr(xlib, n), f'{n} declared in include/gd3d_extras.h but not exported by libgd3d_extras.so'
        assert not hasattr(lib, n), f'{n} is an extra: it must not be part of libgd3d.so'
    assert sorted(_lib.EXTRA_SYMBOLS) == extra, 'ctypes table and gd3d_extras.h disagree'
    assert _lib.load_extras() is not None      # binds: every EXTRA_SYMBOLS entry resolves, the §8 library next to it is found


def test_abi_version_and_queries_without_gpu():
    lib = amd.load_library()
    arch = ctypes.c_char_p()
    assert lib.gd3d_abi_version(ctypes.byref(arch)) == 6
    assert arch.value == b'gfx950'
    assert lib.gd3d_loss_workspace_bytes(0) >= 16
    assert lib.gd3d_loss_workspace_bytes(10_000_000) >= 4 * ((10_000_000 + 255) // 256)
    assert lib.gd3d_loss_workspace_bytes(10_000_000) % 16 == 0
    n = 4096
    assert lib.rnms_workspace_bytes(n) >= n * 64 + n * (n // 64) * 8


def test_params_struct_layout_matches_header():
    assert ctypes.sizeof(_lib.Params) == 32
    assert _lib.Params.center_offset.offset == 16 and _lib.Params.flag.offset == 28


def test_gdloss_ctor_mirrors_reference_asserts():
    amd.GDLoss('gwd3d'); amd.GDLoss('kfiou3d', fun='expm1'); amd.GDLoss('kfiou3d', fun='nlog')
    with pytest.raises(AssertionError):
        amd.GDLoss('gwd3d', fun='expm1')          # ref :267-268
    with pytest.raises(AssertionError):
        amd.GDLoss('kfiou3d', fun='log1p')        # ref :269-270
    with pytest.raises(AssertionError):
        amd.GDLoss('nope')
    with pytest.raises(AssertionError):
        amd.GDLoss('gwd3d', reduction='avg')
    m = amd.GDLoss('bd3d', center_offset=(0, 0, 0.5), fun='log1p', tau=1.0, alpha=1.0, reduction='mean',
                   loss_weight=5.0, sqrt=False)
    assert m.kwargs == {'sqrt': False} and m.loss_weight == 5.0


def test_registry_builds_reference_config_dicts():
    """The dict the reference's KITTI configs carry (configs/kitti/*tau1*.py:7-8) builds unchanged."""
    cfg = dict(type='GDLoss', loss_type='kld3d', fun='log1p', tau=1.0, alpha=1.0, loss_weight=5.0)
    m = amd.build_loss(cfg)
    assert isinstance(m, amd.GDLoss) and m.loss_type == 'kld3d' and 'GDLoss' in amd.LOSSES
    with pytest.raises(KeyError):
        amd.build_loss(dict(type='SmoothL1Loss'))
    with pytest.raises(KeyError):
        amd.LOSSES.register_module()(amd.GDLoss)   # duplicate without force


def test_make_params_flags_and_unknown_kwargs():
    p = amd.make_params('gwd3d', 'log1p', 1.0, 2.0, (0.1, 0.2, 0.3), {'normalize': False})
    assert (p.loss_type, p.fun, p.flag) == (0, 1, 0) and abs(p.alpha - 2.0) < 1e-7
    assert [round(c, 3) for c in p.center_offset] == [0.1, 0.2, 0.3]
    assert amd.make_params('kld3d', 'none', 0.0, 1.0, (0, 0, 0.5), {}).flag == 1          # sqrt default True
    assert amd.make_params('kfiou3d', 'expm1', 0.0, 1.0, (0, 0, 0.5), {}).flag == 0       # sqrt default False
    with pytest.raises(TypeError):
        amd.make_params('kld3d', 'none', 0.0, 1.0, (0, 0, 0.5), {'normalize': True})      # ref: unexpected kwarg


def test_cpu_tensors_take_the_cpu_twin_and_gpu_tensors_never_do():
    """SURVEY.md §8b `_cpu` twins: GDLoss follows the device of its tensors like the reference module
    (gaussian_distance_loss.py:280-310).  CPU tensors run gd3d_loss_fused_cpu (the kernel's per-pair math compiled for the
    host, not the oracle); a tensor that says it is on the GPU goes to the HIP entry points and nowhere else — on this
    GPU-less machine that attempt fails loudly instead of being answered by the CPU twin."""
    m = amd.GDLoss('gwd3d')
    out = m(torch.rand(4, 7) + 0.5, torch.rand(4, 7) + 0.5)
    assert out.dim() == 0 and torch.isfinite(out)
    if not torch.cuda.is_available():
        with pytest.raises(Exception) as info:       # the HIP path is taken (and cannot run here); no silent CPU answer
            m(_FakeCuda(4), _FakeCuda(4))
        assert not isinstance(info.value, AssertionError)
    # the rotated-box entry points and the matcher (CPU code in the reference) follow their tensors too; the batched / scored forms
    # and the scatter ops are GPU-only (no CPU form in the reference either) and say so
    assert amd.nms_gpu(torch.rand(4, 5), torch.rand(4), 0.5).device.type == 'cpu'
    assert amd.iou_3d(torch.rand(4, 7) + 0.5, torch.rand(4, 7) + 0.5).shape == (4, 4)
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.nms_gpu_batched(torch.rand(4, 5), torch.rand(1, 4), 0.5)
    assert amd.match_coco(torch.rand(3, 2), torch.tensor([0.5]), torch.zeros(2, dtype=torch.bool), torch.zeros(2, dtype=torch.bool)).device.type == 'cpu'
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.scatter_index(torch.zeros(4, 3, dtype=torch.int32))
    with pytest.raises(ValueError):
        # avg_factor with reduction='sum' is rejected before any device work (mmdet weight_reduce_loss)
        amd.GDLoss('gwd3d', reduction='sum')(_FakeCuda(4), _FakeCuda(4), avg_factor=2.0)
    with pytest.raises(ValueError):
        amd.GDLoss('gwd3d', reduction='sum')(torch.rand(4, 7), torch.rand(4, 7), avg_factor=2.0)


class _FakeCuda(torch.Tensor):
    """A CPU tensor that claims to be on the GPU, to reach host-side argument checks without a device."""
    @staticmethod
    def __new__(cls, n):
        return torch.Tensor._make_subclass(cls, torch.rand(n, 7))

    @property
    def is_cuda(self):
        return True


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no file of the product package may mention it."""
    pkg = os.path.join(ROOT, 'mmdet3d-gaussian_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt and 'oracle/' not in txt, f


def test_xywhr2xyxyr():
    b = torch.tensor([[1.0, 2.0, 4.0, 2.0, 0.3]])
    assert torch.allclose(amd.xywhr2xyxyr(b), torch.tensor([[-1.0, 1.0, 3.0, 3.0, 0.3]]))


def test_head_slices_validate_shapes_before_touching_the_gpu():
    """Position-indexed kernels must never be launched with operands that do not cover the index range."""
    m = amd.GDLoss('kld3d')
    B, A, H, W = 1, 2, 3, 4
    bbox_pred = torch.zeros(B, A * 7, H, W)
    M = B * H * W * A
    good = dict(bbox_targets=torch.zeros(B, M, 7), bbox_weights=torch.ones(B, M, 7), labels=torch.zeros(B, M, dtype=torch.long),
                anchor_list=torch.zeros(M, 7))
    with pytest.raises(RuntimeError, match='shape mismatch'):
        amd.anchor_head_decoded_loss_fused(m, bbox_pred, good['bbox_targets'][:, :-1], good['bbox_weights'], good['labels'],
                                           good['anchor_list'], 3, 1.0, [1.0] * 7)
    with pytest.raises(RuntimeError, match='shape mismatch'):
        amd.anchor_head_decoded_loss_fused(m, bbox_pred, good['bbox_targets'], good['bbox_weights'], good['labels'],
                                           good['anchor_list'][:-1], 3, 1.0, [1.0] * 7)
    with pytest.raises(RuntimeError, match='multiple of the box code size'):
        amd.anchor_head_decoded_loss_fused(m, torch.zeros(1, 13, 3, 4), good['bbox_targets'], good['bbox_weights'],
                                           good['labels'], good['anchor_list'], 3, 1.0, None)
    with pytest.raises(RuntimeError, match='same number of rows'):
        amd.anchor_decoded_gd_loss(m, torch.zeros(5, 7), torch.zeros(4, 7), torch.zeros(4, 7))


def test_bench_core_count_respects_the_cgroup_quota():
    """bench.py's cpu_baseline must start as many threads as the box GRANTS (a GPU box shows 256 CPUs and a 16-core
    quota): never more than the affinity mask, never less than 1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            assert n <= max(1, int(int(quota) / int(period) + 0.5))
    except OSError:
        pass


def test_center_infer_struct_layouts_match_the_header():
    """sizeof / offsetof of center_infer_task and center_infer_desc as gcc sees include/gd3d.h against the ctypes mirrors."""
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "gd3d_extras.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(center_infer_task), offsetof(center_infer_task, sample_stride), offsetof(center_infer_task, classes),'
           'offsetof(center_infer_task, nms_thresh), sizeof(center_infer_desc), offsetof(center_infer_desc, out_size_factor),'
           'offsetof(center_infer_desc, score_threshold), offsetof(center_infer_desc, limit_range), offsetof(center_infer_desc, tasks));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, 'l.c'), os.path.join(d, 'l')
        open(c, 'w').write(src)
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        got = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    T, D = _lib.CenterInferTask, _lib.CenterInferDesc
    want = [ctypes.sizeof(T), T.sample_stride.offset, T.classes.offset, T.nms_thresh.offset, ctypes.sizeof(D),
            D.out_size_factor.offset, D.score_threshold.offset, D.limit_range.offset, D.tasks.offset]
    assert got == want, (got, want)


def test_center_infer_queries_and_argument_checks_without_gpu():
    lib = _lib.load_extras()   # the frozen extras: libgd3d_extras.so
    assert lib.center_infer_max_k() == 4096
    task = _lib.CenterInferTask()
    task.heatmap = 4096          # never dereferenced: the checks below fail (or only size things) before any launch
    task.classes = 2
    d = _lib.CenterInferDesc()
    d.num_tasks, d.batch, d.height, d.width = 1, 2, 128, 128
    d.max_per_img, d.num_channels, d.decode, d.nms_type = 500, 11, 2, 0
    d.pre_max_size, d.post_max_size = 1000, 83
    d.tasks = ctypes.pointer(task)
    assert lib.center_infer_rows_per_task(ctypes.byref(d)) == 83
    ws = lib.center_infer_workspace_bytes(ctypes.byref(d))
    assert ws >= 2 * 500 * (9 + 1 + 1 + 5) * 4 and ws % 256 == 0
    d.post_max_size = -1
    assert lib.center_infer_rows_per_task(ctypes.byref(d)) == 500
    d.pre_max_size = 40
    assert lib.center_infer_rows_per_task(ctypes.byref(d)) == 40
    d.max_per_img = 5000
    assert lib.center_infer_bboxes(ctypes.byref(d), 256, 256, 256, 256, 256, None) == 10002   # GD3D_E_TOOLARGE
    d.max_per_img, d.height, d.width = 500, 16, 16
    assert lib.center_infer_bboxes(ctypes.byref(d), 256, 256, 256, 256, 256, None) == 10001   # GD3D_E_BADARG: K > H*W: torch.topk raises in the reference
    d.height = d.width = 128
    d.num_channels = 8
    assert lib.center_infer_bboxes(ctypes.byref(d), 256, 256, 256, 256, 256, None) == 10001   # yaw decode needs 9 channels
    d.num_channels = 17
    assert lib.center_infer_select(ctypes.byref(d), 256, 256, 256, 256, 256, None) == 10001
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.select_best(torch.zeros(1, 1, 8, 8), torch.zeros(1, 4, 8, 8), 4)
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.CenterPointBBoxCoderRev([0, 0], 4, [0.2, 0.2]).decode(torch.zeros(1, 4, 2), torch.zeros(1, 4, 10))


def test_center_targets_struct_layout_and_argument_checks():
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "gd3d_extras.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(center_targets_desc), offsetof(center_targets_desc, classes), offsetof(center_targets_desc, sample_start),'
           'offsetof(center_targets_desc, pc_range), offsetof(center_targets_desc, out_size_factor),'
           'offsetof(center_targets_desc, gaussian_overlap));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, 'l.c'), os.path.join(d, 'l')
        open(c, 'w').write(src)
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        got = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    D = _lib.CenterTargetsDesc
    assert got == [ctypes.sizeof(D), D.classes.offset, D.sample_start.offset, D.pc_range.offset, D.out_size_factor.offset,
                   D.gaussian_overlap.offset], got
    lib = _lib.load_extras()   # the frozen extras: libgd3d_extras.so
    assert lib.center_targets_max_boxes() == 8192 and lib.center_targets_workspace_bytes(100) % 256 == 0
    d = D()
    d.num_tasks, d.batch, d.height, d.width, d.total, d.box_cols = 1, 1, 8, 8, 9000, 9
    d.classes[0] = 1
    assert lib.center_targets_build(ctypes.byref(d), 256, 256, 256, 256, 256, 256, 256, None) == 10002      # too many boxes
    d.total = 4
    d.sample_start[1] = 3                                                                                       # does not cover the rows
    assert lib.center_targets_build(ctypes.byref(d), 256, 256, 256, 256, 256, 256, 256, None) == 10001
    d.sample_start[1], d.classes[0] = 4, 0
    assert lib.center_targets_build(ctypes.byref(d), 256, 256, 256, 256, 256, 256, 256, None) == 10001


def test_center_task_struct_layout_matches_header():
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "gd3d.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(gd3d_center_task), offsetof(gd3d_center_task, pos_ind), offsetof(gd3d_center_task, n), offsetof(gd3d_center_task, gd_scale),'
           'offsetof(gd3d_center_task, rows_dev), offsetof(gd3d_center_task, avg_dev), offsetof(gd3d_center_task, gd_weight));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, 'l.c'), os.path.join(d, 'l')
        open(c, 'w').write(src)
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        got = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    D = _lib.CenterTask
    assert got == [ctypes.sizeof(D), D.pos_ind.offset, D.n.offset, D.gd_scale.offset, D.rows_dev.offset, D.avg_dev.offset,
                   D.gd_weight.offset], got


def test_anchor_targets_struct_layout_and_argument_checks():
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "gd3d_extras.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
           'sizeof(anchor_targets_desc), offsetof(anchor_targets_desc, num_dir_bins), offsetof(anchor_targets_desc, gt_start),'
           'offsetof(anchor_targets_desc, pos_iou_thr), offsetof(anchor_targets_desc, min_pos_iou),'
           'offsetof(anchor_targets_desc, dir_offset));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, 'l.c'), os.path.join(d, 'l')
        open(c, 'w').write(src)
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        got = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    D = _lib.AnchorTargetsDesc
    assert got == [ctypes.sizeof(D), D.num_dir_bins.offset, D.gt_start.offset, D.pos_iou_thr.offset, D.min_pos_iou.offset, D.dir_offset.offset], got
    lib = _lib.load_extras()   # the frozen extras: libgd3d_extras.so
    assert lib.anchor_targets_max_gt() == 1024 and lib.anchor_targets_workspace_bytes(3, 100) % 256 == 0
    d = D()
    d.batch, d.cells, d.num_sizes, d.num_rots, d.num_classes, d.num_assigners, d.num_dir_bins = 1, 16, 3, 2, 3, 2, 2
    args = [256] * 11
    assert lib.anchor_targets_build(ctypes.byref(d), *args, None) == 10001          # 2 assigners for 3 sizes
    d.num_assigners = 3
    d.gt_start[1] = 2000
    assert lib.anchor_targets_build(ctypes.byref(d), *args, None) == 10002          # too many boxes in a sample
    d.gt_start[1] = -1
    assert lib.anchor_targets_build(ctypes.byref(d), *args, None) == 10001
    with pytest.raises(RuntimeError, match='no CPU path'):
        amd.extras.anchor_head_get_targets(torch.zeros(2, 2, 1, 2, 7), [torch.zeros(1, 7)], [torch.zeros(1, dtype=torch.long)],
                                    [dict(type='MaxIoUAssigner', pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45)], 1)
    with pytest.raises(RuntimeError, match='sampling=True'):
        amd.extras.anchor_head_get_targets(torch.zeros(2, 2, 1, 2, 7), [torch.zeros(1, 7)], [torch.zeros(1, dtype=torch.long)],
                                    dict(type='MaxIoUAssigner', pos_iou_thr=0.6, neg_iou_thr=0.45, min_pos_iou=0.45), 1, sampling=True)


def test_cpu_twins_refuse_a_host_without_avx2_fma_instead_of_sigill():
    """the `_cpu` twins are compiled for x86-64-v3; on a host CPU without AVX2 / FMA the binding replaces every
    `*_cpu` entry by a callable that raises a clear RuntimeError (a fresh process: the check is made once, at bind time)."""
    import subprocess
    import sys
    code = f'''
import sys
sys.path.insert(0, {ROOT!r})
import torch
from mmdet3d_gaussian_amd import _lib
_lib._simd_ok = False                      # what host_simd_ok() finds on such a host
import mmdet3d_gaussian_amd as amd
t = torch.rand(8, 7) + 0.5
try:
    amd.GDLoss('gwd3d', reduction='none')(t, t)
except RuntimeError as e:
    print('REFUSED', e)
try:
    amd.nms_gpu(torch.rand(4, 5), torch.rand(4), 0.5)
except RuntimeError as e:
    print('REFUSED', e)
'''
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.stdout.count('REFUSED') == 2 and 'AVX2' in r.stdout, r.stdout + r.stderr[-2000:]


def test_host_thread_team_reports_a_throwing_worker(tmp_path):
    """a worker body that throws (std::bad_alloc in a scratch vector) must not escape a std::thread (std::terminate):
    parallel_ranges catches it, the other ranges finish, and the entry point returns GD3D_E_HOST."""
    import shutil
    import subprocess
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        pytest.skip('no host C++ compiler')
    src = tmp_path / 't.cpp'
    src.write_text('''
#include <cstdio>
#include <new>
#include "%s/mmdet3d-gaussian_amd/csrc/host_threads.h"
int main() {
  std::atomic<long> done{0};
  const bool ok = gd3d_host::parallel_ranges(1000, 4, [&](int64_t a, int64_t b) {
    if (a == 250) throw std::bad_alloc();
    done += b - a;
  });
  const bool ok1 = gd3d_host::parallel_ranges(10, 1, [&](int64_t, int64_t) { throw 1; });
  const bool ok2 = gd3d_host::parallel_ranges(100, 3, [&](int64_t a, int64_t b) { done += b - a; });
  std::printf("%%d %%d %%d %%ld\\n", (int)ok, (int)ok1, (int)ok2, done.load());
  return 0;
}
''' % ROOT)
    exe = tmp_path / 't'
    subprocess.run([cxx, '-std=c++17', '-O1', '-pthread', str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.split()
    assert out == ['0', '0', '1', '850'], out          # 750 from the three healthy ranges of the first team + 100
    hdr = open(os.path.join(ROOT, 'include', 'gd3d.h')).read()
    assert '#define GD3D_E_HOST 10003' in hdr
