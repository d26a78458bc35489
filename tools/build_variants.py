#!/usr/bin/env python3
"""Build experimental variants of libgd3d.so (macro switches in csrc/gd3d_loss.hip) into tools/variants/.
usage: tools/build_variants.py name1="-DGD_X=1 -DGD_Y=2" name2=...   (A/B timing with tools/kernel_time.py)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location('gdbuild', os.path.join(ROOT, 'mmdet3d-gaussian_amd', 'build.py'))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
out = os.path.join(ROOT, 'tools', 'variants'); os.makedirs(out, exist_ok=True)
procs = []
for arg in sys.argv[1:]:
    name, _, flags = arg.partition('=')
    objs = []
    for src, fl in b.SOURCES.items():
        obj = os.path.join(out, f'{name}_{src}.o')
        cmd = [b.hipcc_path()] + b.COMMON + fl + flags.split() + ['-c', os.path.join(b.CSRC, src), '-o', obj]
        procs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        objs.append(obj)
    for src, fl in b.HOST_SOURCES.items():      # the _cpu twins: same object for every variant
        obj = os.path.join(out, f'{name}_{src}.o')
        procs.append((name, subprocess.Popen([b.host_cxx_path()] + b.HOST_COMMON + fl + ['-c', os.path.join(b.CSRC, src), '-o', obj], stderr=subprocess.PIPE, text=True)))
        objs.append(obj)
    procs.append((name, ('link', objs)))
for name, p in procs:
    if isinstance(p, tuple):
        so = os.path.join(out, f'libgd3d_{name}.so')
        r = subprocess.run([b.hipcc_path(), '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', so] + p[1], capture_output=True, text=True)
        print(name, 'OK' if r.returncode == 0 else r.stderr[-2000:])
    else:
        _, err = p.communicate()
        if p.returncode != 0: print(name, 'compile failed:\n', err[-3000:])
