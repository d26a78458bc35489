#!/usr/bin/env python3
"""Prove that the parity gates bite: run the golden-vector tests of the HIP path against deliberately
damaged builds of libgd3d.so and record that they FAIL, next to the product build PASSING.

  perturb : -DGD_TEST_PERTURB_KL_BWD=1e-4   everything kl_bwd accumulates scaled by (1 + 1e-4)
  naive   : -DGD_TEST_NAIVE_RATIO           rho^2 - 1 formed the textbook way (cancels on near-identical boxes only)

usage (on the GPU box): python tools/gate_bites.py [out_file]   (builds the variants with tools/build_variants.py first)
Each run is a child `pytest` with GD3D_LIB pointing at the variant and GD3D_HOST=python (the glue does not matter; the Python
one needs no rebuild).  Output: per build, passed / failed counts of the selected tests and, for failures, which families failed
with the worst error against its bound."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {'perturb': '-DGD_TEST_PERTURB_KL_BWD=1e-4', 'naive': '-DGD_TEST_NAIVE_RATIO=1'}
SELECT = 'test_pairs_against_reference_golden and (kld3d or jd3d)'   # every loss that goes through kl_bwd


def run(lib, report):
    env = dict(os.environ, GD3D_HOST='python', GD3D_ACCURACY_REPORT=report)   # the child's report must not replace the suite's
    if lib:
        env['GD3D_LIB'] = lib
    else:
        env.pop('GD3D_LIB', None)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_gd_loss.py'), '-m', 'gpu', '-q', '-k', SELECT,
                        '--tb=line', '-p', 'no:cacheprovider'], capture_output=True, text=True, env=env, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
    fails = re.findall(r'AssertionError: (\S+): (\d+) of (\d+) outside tolerance; .*?\|err\|=(\S+) bound=(\S+)', r.stdout)
    return r.returncode, tail, fails


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r05_gate_bites.txt')
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'build_variants.py')] + [f'{k}={v}' for k, v in VARIANTS.items()],
                   check=True)
    lines = ['# tools/gate_bites.py — tests/test_gpu_gd_loss.py::test_pairs_against_reference_golden (the kl_bwd losses: kld3d, jd3d,',
             '# kld3d_symmax, kld3d_symmin; all input families; gate = flat 1e-5 x (1 + scale) vs the reference\'s fp64, tests/gd_golden.py)',
             '# against the product build and two deliberately damaged builds.  The damaged builds MUST fail.']
    ok = True
    for name, lib in [('product', None)] + [(k, os.path.join(ROOT, 'tools', 'variants', f'libgd3d_{k}.so')) for k in VARIANTS]:
        rc, tail, fails = run(lib, f'{out}.{name}.accuracy.txt')
        lines.append(f'{name:8s} ({VARIANTS.get(name, "as shipped")}): pytest rc={rc}: {tail}')
        fam = {}
        for key, nbad, ntot, err, bound in fails:
            f = key.split('.')[2]
            q = key.split('.')[-1].split('[')[0]
            cur = fam.setdefault((f, q), [0, 0.0])
            cur[0] += 1
            cur[1] = max(cur[1], float(err) / float(bound))
        for (f, q), (cnt, worst) in sorted(fam.items()):
            lines.append(f'    first failing comparison of {cnt:3d} case(s) was family {f:6s} quantity {q:4s}: worst |err| / bound = {worst:.1f}')
        ok &= (rc == 0) if name == 'product' else (rc != 0)
    lines.append('RESULT: ' + ('the gates bite (product passes, both damaged builds fail)' if ok else 'UNEXPECTED'))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
