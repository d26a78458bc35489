#!/usr/bin/env python3
"""Compile the stand-alone measurement programs under tools/ for gfx950 (plain hipcc, no torch):
   hbm_probe   measured HBM ceilings for the fused kernel's access mix
   hbm_probe2  the fused kernel's tile mechanics without its math, flat copies by shape, the occupancy cap
   pair_variants  does the access mode change what a cross-class pair of read streams costs
   launch_floor  what a tiny dependent kernel costs behind a streaming kernel (the floor of the loss sum's second stage)
The binaries are built in-tree (git-ignored; they travel to the GPU box with the snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PROBES = {
    'hbm_probe': ['-O3'],
    'hbm_probe2': ['-O3'],
    'launch_floor': ['-O3'],
    'pair_variants': ['-O3'],
}


def build(verbose=False):
    hipcc = '/opt/rocm/bin/hipcc'
    out = []
    for name, flags in PROBES.items():
        src, exe = os.path.join(HERE, name + '.hip'), os.path.join(HERE, name)
        if os.path.isfile(exe) and os.path.getmtime(exe) >= os.path.getmtime(src):
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
