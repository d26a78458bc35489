"""Build recipe of libgd3d.so (hand-written HIP for gfx950 behind the C ABI of include/gd3d.h) and of _gd3d_node.so
(the C++ autograd node GDLoss calls it through).

libgd3d.so: plain ``hipcc --offload-arch=gfx950``: no torch headers, no hipify, no cmake.  _gd3d_node.so: one host
translation unit (csrc/torch_node.cpp) against the torch headers of the running interpreter, no kernel in it.  Both are
built IN-TREE (mmdet3d-gaussian_amd/) so that they travel to the GPU box with the snapshot.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libgd3d.so')
ARCH = 'gfx950'

# per-translation-unit flags:
#  * gd3d_loss.hip: 1-ulp v_rcp/v_sqrt instead of the ~10-instruction correctly-rounded sequences
#    (the losses are graded at 1e-5; the kernel must stay HBM-bound, not VALU-bound);
#  * rbox.hip: correctly-rounded IEEE division/sqrt and NO fma contraction, because the NMS keep
#    indices must be bit-identical to a CPU evaluation of the same fp32 operation sequence.
SOURCES = {   # the SURVEY.md §8 surface: libgd3d.so (include/gd3d.h)
    'gd3d_loss.hip': ['-fno-hip-fp32-correctly-rounded-divide-sqrt', '-ffp-contract=fast'],
    'rbox.hip': ['-ffp-contract=off'],
    'voxel_scatter.hip': [],
    'voxel_index.hip': [],                     # rocPRIM radix sort + scan between hand-written kernels
    'eval_match.hip': ['-ffp-contract=off'],
    'coders.hip': ['-ffp-contract=off'],      # same rounding sequence as the torch elementwise ops it replaces
}
# The frozen round-3 extras OUTSIDE §8 (DESIGN_EXTRAS.md; include/gd3d_extras.h): a library of their own since round 6,
# libgd3d_extras.so, linked against libgd3d.so (the inference slices call its rnms_* entry points) and loaded only by
# mmdet3d_gaussian_amd.extras' modules (_lib.load_extras()).  What ships as libgd3d.so is the §8 translation units only.
EXTRA_SOURCES = {
    'center_infer.hip': ['-ffp-contract=off'],    # decode as coders.hip; the NMS boxes feed bit-exact keep decisions
    'anchor_targets.hip': ['-ffp-contract=off'],   # IoU and thresholds decide as the torch elementwise ops do
    'anchor_cls.hip': [],                       # elementwise focal + direction loss with gradient, graded at 1e-5
    'heat_focal.hip': ['-fno-hip-fp32-correctly-rounded-divide-sqrt', '-ffp-contract=fast'],   # elementwise loss + gradient, graded at 1e-5
    'anchor_infer.hip': ['-ffp-contract=off'],    # delta decode in the reference's operation order; boxes feed the NMS
    'center_targets.hip': ['-ffp-contract=off'],  # gaussian_radius in the reference's fp32 operation order
}
EXTRAS_PATH = os.path.join(PKG_DIR, 'libgd3d_extras.so')
COMMON = ['--offload-arch=' + ARCH, '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']

# host translation units (the `_cpu` twins of include/gd3d.h): compiled by the ROCm toolchain's own clang++ as plain C++
# (csrc/gd3d_device.h uses clang vector extensions), for the x86-64-v3 level (AVX2 + FMA: the per-pair math is written in
# explicit fmaf(), which must be an instruction, not a libm call), same contraction mode as the device build of the loss
HOST_SOURCES = {
    'gd3d_cpu.cpp': ['-march=x86-64-v3', '-ffp-contract=fast', '-pthread'],
    'rbox_cpu.cpp': ['-march=x86-64-v3', '-ffp-contract=off', '-pthread'],   # same single-operation sequence as rbox.hip: bit-identical decisions
}
HOST_COMMON = ['-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']


def hipcc_path():
    return shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def host_cxx_path():
    """The ROCm toolchain's own clang++ (host compiler of the `_cpu` twins and of the optional C++ node), or None."""
    for c in ('/opt/rocm/lib/llvm/bin/clang++', '/opt/rocm/llvm/bin/clang++', shutil.which('amdclang++') or ''):
        if c and os.path.exists(c):
            return c
    return None


NODE_SOURCE = 'torch_node.cpp'
NODE_PATH = os.path.join(PKG_DIR, '_gd3d_node.so')
NODE_HASH_PATH = NODE_PATH + '.srchash'


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files if f != NODE_SOURCE]
    out.append(os.path.join(PKG_DIR, '..', 'include', 'gd3d.h'))
    out.append(os.path.join(PKG_DIR, '..', 'include', 'gd3d_extras.h'))
    out.append(os.path.abspath(__file__))
    return out


HASH_PATH = LIB_PATH + '.srchash'


def source_hash():
    """Content hash of everything the library is built from (mtimes do not survive a snapshot to another box)."""
    h = hashlib.sha256()
    for d in sorted(os.path.normpath(x) for x in _deps()):
        if os.path.exists(d):
            h.update(os.path.basename(d).encode())
            with open(d, 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def is_stale():
    if not os.path.isfile(LIB_PATH) or not os.path.isfile(EXTRAS_PATH) or not os.path.isfile(HASH_PATH):
        return True
    with open(HASH_PATH) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libgd3d.so (the §8 surface) and libgd3d_extras.so (the frozen
    extras, linked against it).  Returns libgd3d.so's path."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = hipcc_path()
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libgd3d.so')
    objdir = os.path.join(PKG_DIR, 'build', str(os.getpid()))  # per process: ranks may build concurrently
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src, flags in list(SOURCES.items()) + list(EXTRA_SOURCES.items()):
        path = os.path.join(CSRC, src)
        if not os.path.isfile(path):
            continue
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        jobs.append((src, obj, [hipcc] + COMMON + flags + ['-c', path, '-o', obj]))
    cxx = host_cxx_path()
    if cxx is None:
        raise RuntimeError('the ROCm clang++ (host compiler of the _cpu twins) was not found next to hipcc: cannot build libgd3d.so')
    for src, flags in HOST_SOURCES.items():
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace('.cpp', '.o'))
        jobs.append((src, obj, [cxx] + HOST_COMMON + flags + ['-c', path, '-o', obj]))

    def compile_one(job):
        src, obj, cmd = job
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'compiling {src} failed:\n{r.stderr[-4000:]}')
        return obj
    # the translation units are independent: one hipcc per core, at most 8 (each takes 5-25 s and ~1 GB)
    workers = max(1, min(8, os.cpu_count() or 1, len(jobs)))
    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
        objs = dict(zip((j[0] for j in jobs), pool.map(compile_one, jobs)))
    main_objs = [objs[s_] for s_ in list(SOURCES) + list(HOST_SOURCES) if s_ in objs]
    extra_objs = [objs[s_] for s_ in EXTRA_SOURCES if s_ in objs]
    tmp = f'{LIB_PATH}.{os.getpid()}.tmp'
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-pthread', '-o', tmp] + main_objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr[-4000:]}')
    os.replace(tmp, LIB_PATH)  # atomic
    # the extras: their own image, resolving rnms_* from libgd3d.so next to it ($ORIGIN: the pair travels together)
    tmp = f'{EXTRAS_PATH}.{os.getpid()}.tmp'
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-pthread', '-o', tmp] + extra_objs + \
          ['-L' + PKG_DIR, '-l:libgd3d.so', '-Wl,-rpath,$ORIGIN']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link of libgd3d_extras.so failed:\n{r.stderr[-4000:]}')
    os.replace(tmp, EXTRAS_PATH)
    with open(HASH_PATH + f'.{os.getpid()}', 'w') as f:
        f.write(source_hash())
    os.replace(HASH_PATH + f'.{os.getpid()}', HASH_PATH)
    shutil.rmtree(objdir, ignore_errors=True)
    return LIB_PATH


def node_source_hash():
    """The node is rebuilt when its source, the C header, the flags or the torch it was compiled against change.  The
    compiler's PATH is not part of it: a box without the compiler can still verify (and load) a matching prebuilt binary."""
    import torch
    h = hashlib.sha256()
    h.update(torch.__version__.encode())
    for d in (os.path.join(CSRC, NODE_SOURCE), os.path.join(PKG_DIR, '..', 'include', 'gd3d.h')):
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(repr(node_command('OUT', cxx='CXX')).encode())
    return h.hexdigest()


def node_command(out, cxx=None):
    import sysconfig
    import torch
    tdir = os.path.dirname(os.path.abspath(torch.__file__))
    tlib = os.path.join(tdir, 'lib')
    return [cxx or host_cxx_path(), '-std=c++17', '-O2', '-fPIC', '-shared', '-Wall', '-Wno-unused-function',
            '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-DTORCH_EXTENSION_NAME=_gd3d_node',
            f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}',
            '-isystem', os.path.join(tdir, 'include'), '-isystem', os.path.join(tdir, 'include', 'torch', 'csrc', 'api', 'include'),
            '-isystem', '/opt/rocm/include', '-isystem', sysconfig.get_paths()['include'],
            os.path.join(CSRC, NODE_SOURCE), '-o', out,
            '-L' + tlib, '-ltorch', '-ltorch_cpu', '-lc10', '-lc10_hip', '-ltorch_hip', '-ltorch_python', '-ldl',
            '-Wl,-rpath,' + tlib]


def node_is_stale():
    if not os.path.isfile(NODE_PATH) or not os.path.isfile(NODE_HASH_PATH):
        return True
    with open(NODE_HASH_PATH) as f:
        return f.read().strip() != node_source_hash()


def build_node(force=False, verbose=False):
    """Compile csrc/torch_node.cpp (host C++, ~25 s) into _gd3d_node.so.  It resolves the C ABI from libgd3d.so at run
    time (`bind`), so the two are built independently."""
    if not force and not node_is_stale():
        return NODE_PATH
    if host_cxx_path() is None:
        raise RuntimeError('the ROCm clang++ was not found: cannot build the optional C++ node (_gd3d_node.so)')
    tmp = f'{NODE_PATH}.{os.getpid()}.tmp'
    cmd = node_command(tmp)
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'compiling {NODE_SOURCE} failed:\n{r.stderr[-4000:]}')
    os.replace(tmp, NODE_PATH)
    with open(NODE_HASH_PATH + f'.{os.getpid()}', 'w') as f:
        f.write(node_source_hash())
    os.replace(NODE_HASH_PATH + f'.{os.getpid()}', NODE_HASH_PATH)
    return NODE_PATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
    print(build_node(force=True, verbose=True))
