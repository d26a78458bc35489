#!/bin/bash
# A/B of NMS kernel time across library variants: tools/nms_ab.sh n thr clutter lib1 lib2 ...
n=$1; thr=$2; cl=$3; shift 3
export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/kt_ab
  if [ "$lib" = product ]; then unset GD3D_LIB; else export GD3D_LIB=$GRAFT_REPO_ROOT/tools/variants/libgd3d_$lib.so; fi
  (cd /tmp && GD3D_HOST=python rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_ab -- python3 $GRAFT_REPO_ROOT/tools/nms_one.py $n $thr $cl > /dev/null 2>&1)
  echo "== $lib n=$n thr=$thr clutter=$cl"
  python3 $GRAFT_REPO_ROOT/tools/nms_kstats.py /tmp/kt_ab | grep -v rank_place
done
