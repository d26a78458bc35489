#!/bin/bash
# Run ON THE GPU BOX: where the time of the rotated NMS goes at n = 4096 (thr 0.25, clustered): issue counters of the mask
# kernel and the scan workgroup (rocprofv3 --pmc, separate passes), then the scan's own cycle stamps per 64-box block from the
# profiling build (tools/build_variants.py prof="-DSCAN_PROFILE=1" must have been run in the build container).
#   tools/nms_pmc.sh <out-file>
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/nms_pmc.txt}
D=gpurun_out/_pmc_nms
rm -rf $D; mkdir -p $D
i=0
for cnt in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1)); mkdir -p $D/$i
  rocprofv3 --pmc $cnt --output-format csv -d $D/$i -- python3 tools/nms_pmc_driver.py > $D/$i.log 2>&1 || tail -2 $D/$i.log
done
{
  echo "# rotated NMS, n = 4096 clustered boxes, thr 0.25 (BASELINE configs[4] stand-in): rocprofv3 --pmc, mean per launch over 5 rnms_bev calls"
  echo "# (SQ_ACTIVE_* and SQ_WAIT_* are in units of 4 cycles, summed over waves; SQ_BUSY_CYCLES over all SEs)"
  python3 tools/pmc_summary.py "$D/**/*_counter_collection.csv" --kernel rbox::
  echo
  echo "# cycle stamps of the scan workgroup per 64-box block (profiling build, tools/scan_profile.py; cycles of the 100 MHz-independent shader clock counter clock64())"
  if [ -f tools/variants/libgd3d_prof.so ]; then GD3D_LIB=tools/variants/libgd3d_prof.so python3 tools/scan_profile.py 2>&1 | grep -v amdgpu.ids; else echo "(no tools/variants/libgd3d_prof.so)"; fi
} > $OUT
rm -rf $D
cat $OUT
