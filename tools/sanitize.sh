#!/bin/bash
# tools/sanitize.sh — the HOST side of the package under sanitizers (CPU only: never on the GPU build, never on a GPU box).
#   tools/sanitize.sh [asan|tsan|all]   -> profiles/r06_sanitizers.txt (summary) + /tmp/gd3d_san/<kind>/ (logs)
# What is instrumented (clang of the ROCm toolchain, -O1 -g):
#   * libgd3d.so's host code: csrc/gd3d_cpu.cpp, csrc/rbox_cpu.cpp (the `_cpu` twins; csrc/host_threads.h's std::thread team) AND
#     the host halves of every .hip translation unit (the C-ABI argument checks and launch wrappers: -fno-gpu-sanitize keeps the
#     device code as it ships) -> one library with every symbol include/gd3d.h declares;
#   * csrc/torch_node.cpp (the optional C++ autograd node: node lifetime, saved variables, the count mailbox);
#   * oracle/gd_oracle.c, oracle/rbox_oracle.c (the checkers).
# What runs against them: the CPU suites that go through that code — tests/test_cpu_gd_loss.py, test_cpu_rbox.py, test_host_abi.py,
# test_autograd_node.py, test_sharded_gloo.py (two gloo ranks) — with the sanitizer runtime preloaded into python.
#   asan : -fsanitize=address,undefined      tsan : -fsanitize=thread (own build, own run: the two cannot be combined)
set -u
KIND=${1:-all}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LLVM=/opt/rocm/lib/llvm/bin
CXX=$LLVM/clang++; CC=$LLVM/clang; HIPCC=/opt/rocm/bin/hipcc
CSRC="$ROOT/mmdet3d-gaussian_amd/csrc"
TESTS="tests/test_cpu_gd_loss.py tests/test_cpu_rbox.py tests/test_host_abi.py tests/test_autograd_node.py tests/test_sharded_gloo.py"
SUMMARY=$ROOT/profiles/r06_sanitizers.txt
one() {
  local kind=$1 san rt opts
  local out=/tmp/gd3d_san/$kind; rm -rf $out; mkdir -p $out
  if [ $kind = asan ]; then
    san="-fsanitize=address,undefined -fno-omit-frame-pointer"; rt=$($CXX -print-file-name=libclang_rt.asan-x86_64.so)
  else
    san="-fsanitize=thread"; rt=$($CXX -print-file-name=libclang_rt.tsan-x86_64.so)
  fi
  echo "[$kind] building into $out" >&2
  # per-translation-unit flags as mmdet3d-gaussian_amd/build.py sets them (contraction modes decide bits), -O1 -g for the reports
  python3 - "$ROOT" "$out" "$san" <<'PY' || return 1
import importlib.util, os, subprocess, sys, concurrent.futures
root, out, san = sys.argv[1], sys.argv[2], sys.argv[3].split()
spec = importlib.util.spec_from_file_location('gdbuild', os.path.join(root, 'mmdet3d-gaussian_amd', 'build.py'))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
jobs = []
for src, fl in b.SOURCES.items():
    obj = os.path.join(out, src + '.o')
    jobs.append((obj, [b.hipcc_path(), '--offload-arch=gfx950', '-O1', '-g', '-fPIC', '-std=c++17', '-Wno-unused-function'] + fl + san + ['-fno-gpu-sanitize', '-c', os.path.join(b.CSRC, src), '-o', obj]))
for src, fl in b.HOST_SOURCES.items():
    obj = os.path.join(out, src + '.o')
    jobs.append((obj, [b.host_cxx_path(), '-O1', '-g', '-fPIC', '-std=c++17', '-Wno-unused-function'] + fl + san + ['-c', os.path.join(b.CSRC, src), '-o', obj]))
def run(j):
    r = subprocess.run(j[1], capture_output=True, text=True)
    if r.returncode: raise SystemExit(' '.join(j[1]) + '\n' + r.stderr[-3000:])
    return j[0]
with concurrent.futures.ThreadPoolExecutor(6) as ex: objs = list(ex.map(run, jobs))
r = subprocess.run([b.hipcc_path(), '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-shared-libsan'] + san + ['-fno-gpu-sanitize', '-o', os.path.join(out, 'libgd3d.so')] + objs, capture_output=True, text=True)
if r.returncode: raise SystemExit('link: ' + r.stderr[-3000:])
cmd = b.node_command(os.path.join(out, '_gd3d_node.so'))
cmd = [c for c in cmd if c != '-O2'] + ['-O1', '-g', '-shared-libsan'] + san
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode: raise SystemExit('node: ' + r.stderr[-3000:])
print('built', out)
PY
  # the oracle with the same clang (its OpenMP runtime is libomp; one sanitizer runtime per process)
  $CC -O1 -g -fPIC -fopenmp -ffp-contract=off $san -shared-libsan -shared -o $out/libgd3d_oracle.so "$ROOT/oracle/gd_oracle.c" "$ROOT/oracle/rbox_oracle.c" -lm \
      -Wl,-rpath,/opt/rocm/lib/llvm/lib || return 1
  if [ $kind = asan ]; then
    export ASAN_OPTIONS="detect_leaks=0:halt_on_error=0:log_path=$out/asan:protect_shadow_gap=0:detect_odr_violation=0"
    export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:log_path=$out/ubsan"
  else
    # libomp and libtorch are not instrumented: their synchronisation is invisible to the tool -> only races BETWEEN instrumented
    # frames count (ignore_noninstrumented_modules), the oracle's OpenMP loops run on one thread
    export TSAN_OPTIONS="halt_on_error=0:log_path=$out/tsan:ignore_noninstrumented_modules=1:report_signal_unsafe=0"
    export OMP_NUM_THREADS=1
  fi
  # canary first: the tool must BITE on this build in this process model (python + preloaded runtime + ctypes) — asan: the twin
  # writes the gradient of 300 rows into a buffer of 299; tsan: two threads let the twin write the same output arrays at once
  ( cd "$ROOT" && mkdir -p $out/canary && ASAN_OPTIONS="${ASAN_OPTIONS:-}:log_path=$out/canary/asan" UBSAN_OPTIONS="${UBSAN_OPTIONS:-}:log_path=$out/canary/asan" \
      TSAN_OPTIONS="${TSAN_OPTIONS:-}:log_path=$out/canary/tsan" \
      LD_PRELOAD=$rt GD3D_LIB=$out/libgd3d.so GD3D_HOST=python python3 tools/sanitize_canary.py $kind > $out/canary/out.log 2>&1 )
  local canary=$(cat $out/canary/asan.* $out/canary/tsan.* 2>/dev/null | grep -cE "heap-buffer-overflow|data race")
  ( cd "$ROOT" && LD_PRELOAD=$rt GD3D_LIB=$out/libgd3d.so GD3D_NODE_LIB=$out/_gd3d_node.so GD3D_ORACLE_LIB=$out/libgd3d_oracle.so \
      GD3D_SANITIZER=$kind python3 -m pytest $TESTS -q -p no:cacheprovider > $out/pytest.log 2>&1 )
  local rc=$?
  unset ASAN_OPTIONS UBSAN_OPTIONS TSAN_OPTIONS OMP_NUM_THREADS
  local reports=$(ls $out | grep -E "^(asan|ubsan|tsan)\." | wc -l)
  {
    echo "== $kind ($san), $(date -u +%Y-%m-%dT%H:%MZ), $($CXX --version | head -1)"
    echo "   pytest: $(tail -1 $out/pytest.log)   (exit status $rc)"
    echo "   canary (a deliberate $( [ $kind = asan ] && echo 'one-row heap overflow through gd3d_loss_fused_cpu' || echo 'two-thread write race on one gradient array' ), tools/sanitize_canary.py): $( [ $canary -gt 0 ] && echo "CAUGHT ($canary report lines)" || echo 'NOT caught — the run below proves nothing' )"
    echo "   sanitizer report files of the test run: $reports"
    for f in $(ls $out | grep -E "^(asan|ubsan|tsan)\." | head -20); do
      echo "   --- $f"; grep -E "ERROR|WARNING|runtime error|SUMMARY" $out/$f | sort | uniq -c | sort -rn | head -12 | sed 's/^/       /'
    done
  } >> $SUMMARY
  return $rc
}
{
  echo "r06_sanitizers.txt — host code of the package under clang's sanitizers (tools/sanitize.sh; CPU container, no GPU minute)"
  echo "instrumented: libgd3d.so's host code (the _cpu twins gd3d_cpu.cpp / rbox_cpu.cpp with host_threads.h, and the host halves of all"
  echo ".hip units), csrc/torch_node.cpp, oracle/*.c.  run: $TESTS"
} > $SUMMARY
rc=0
case $KIND in
  asan|tsan) one $KIND || rc=$? ;;
  *) one asan || rc=$?; one tsan || rc=$? ;;
esac
cat $SUMMARY
exit $rc
