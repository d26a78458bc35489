#!/usr/bin/env python3
"""Compile the stand-alone measurement programs under tools/ for gfx950 (plain hipcc, no torch):
   hbm_probe   measured HBM ceilings for the fused kernel's access mix
   clip_probe  cycle accounting of one polygon-clipping pass of the rotated-NMS predicate
   mask_probe  cycle stamps of one wave of the NMS mask kernel inside a real call
   sort_probe  LDS bitonic sort of (key, ~index) entries in one 1024-thread workgroup: passes, barriers, idle-chip effects
The binaries are built in-tree (git-ignored; they travel to the GPU box with the snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PROBES = {
    'hbm_probe': ['-O3'],
    'hbm_probe2': ['-O3'],
    'clip_probe': ['-O3', '-ffp-contract=off'],
    'mask_probe': ['-O3', '-ffp-contract=off', '-Wno-unused-value'],
    'sort_probe': ['-O3', '-Wno-unused-result'],
}


def build(verbose=False):
    hipcc = '/opt/rocm/bin/hipcc'
    out = []
    for name, flags in PROBES.items():
        src, exe = os.path.join(HERE, name + '.hip'), os.path.join(HERE, name)
        if os.path.isfile(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(src), *(
                os.path.getmtime(os.path.join(HERE, '..', 'mmdet3d-gaussian_amd', 'csrc', f))
                for f in ('rbox.hip', 'rbox_device.h'))):
            out.append(exe)
            continue
        cmd = [hipcc, '--offload-arch=gfx950'] + flags + ['-o', exe, src]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=None if verbose else subprocess.DEVNULL)
        out.append(exe)
    return out


if __name__ == '__main__':
    print('\n'.join(build(verbose='-v' in sys.argv)))
