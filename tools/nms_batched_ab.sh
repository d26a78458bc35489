#!/bin/bash
# Run ON THE GPU BOX: per-kernel times of the configs[4] NMS stand-in, single class vs the batched call (tools/nms_batched_one.py)
# usage: tools/nms_batched_ab.sh [variant ...]      (variants from tools/build_variants.py; "product" = the in-tree library)
export TMPDIR=/tmp
[ $# -eq 0 ] && set -- product
for lib in "$@"; do
  if [ "$lib" = product ]; then unset GD3D_LIB; else export GD3D_LIB=$GRAFT_REPO_ROOT/tools/variants/libgd3d_$lib.so; fi
  for form in ${FORMS:-single batched batched_c}; do
    rm -rf /tmp/kt_nb
    echo "== $lib $form"
    GD3D_HOST=python python3 $GRAFT_REPO_ROOT/tools/nms_batched_one.py $form 2>&1 | grep "us per call"
    (cd /tmp && GD3D_HOST=python rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_nb -- python3 $GRAFT_REPO_ROOT/tools/nms_batched_one.py $form > /dev/null 2>&1)
    python3 $GRAFT_REPO_ROOT/tools/nms_kstats.py /tmp/kt_nb
  done
done
