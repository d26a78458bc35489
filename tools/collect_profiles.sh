#!/bin/bash
# Run ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence for the round into gpurun_out/<round>/.
#   tools/collect_profiles.sh r06
# Hot path only; the frozen extras (DESIGN_EXTRAS.md) are not measured any more.
# kernel-trace/stats and each PMC set are separate runs (gpurun refuses --pmc combined with trace domains,
# and FETCH_SIZE / WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
R=${1:-r06}
export TMPDIR=/tmp
OUT=gpurun_out/$R
mkdir -p $OUT
BENCH="python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-traffic"
SHORT="python3 bench.py --steps 3 --warmup 2 --cpu-sample 0 --no-traffic --plain-steps 0"
# the plain bench line first, on the fresh box: the defaults (100 steps / 20 warmup), then the driver's own N = 1 command
python3 bench.py > $OUT/${R}_bench.json 2> $OUT/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${R}_bench_driver_form.json 2> $OUT/bench_driver_form.err
# rounds 4-5's headline form (the four input arrays are row ranges of one allocation), and the plain step without the LossValue
# (GD3D_UNIT_ROOT=0: torch fills a ones tensor, every node launches its early-exit finish — the step of round 5)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --arena --cpu-sample 0 --no-traffic --no-standins > $OUT/${R}_bench_driver_form_arena.json 2>> $OUT/bench_driver_form.err
GD3D_UNIT_ROOT=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-traffic --no-standins > $OUT/${R}_bench_driver_form_no_loss_value.json 2>> $OUT/bench_driver_form.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH > $OUT/bench_under_kernel_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $SHORT > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $OUT/pmc_sq -- $SHORT > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_sq2 -- $SHORT > $OUT/pmc_sq2.log 2>&1
python3 tools/build_probes.py > /dev/null; [ -x tools/hbm_probe ] && ./tools/hbm_probe > $OUT/${R}_hbm_probe.txt 2>&1
# multi-GPU rehearsal with ONE rank under torch.distributed.run (RCCL backend, hipGraph replay + per-step all_gather): the
# N > 1 code path of bench.py as far as one GPU can exercise it; weak and strong scaling (--graph: one rank alone would launch
# eagerly, N > 1 ranks replay a graph), and the eager form
RUN1="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
$RUN1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-traffic --graph > $OUT/${R}_rccl_rehearsal_weak.json 2> $OUT/rehearsal_weak.err
$RUN1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-traffic --graph --strong > $OUT/${R}_rccl_rehearsal_strong.json 2> $OUT/rehearsal_strong.err
$RUN1 --master-port 29513 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-traffic --no-graph > $OUT/${R}_rccl_rehearsal_weak_eager.json 2> $OUT/rehearsal_weak_eager.err
python3 tests/perf/nms_time.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_nms_time.txt
FORMS="single batched batched_c multi" tools/nms_batched_ab.sh product 2>&1 | grep -v amdgpu.ids > $OUT/${R}_nms_batched_kernels.txt
[ -x tools/launch_floor ] && ./tools/launch_floor > $OUT/${R}_launch_floor.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_nms -- python3 tests/perf/nms_time.py > /dev/null 2>&1
cp $OUT/kt_nms/*/*kernel_stats.csv $OUT/${R}_nms_kernel_stats.csv 2>/dev/null
python3 tests/perf/small_p_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_small_p_latency.jsonl
python3 tests/perf/head_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_head_latency.jsonl
python3 tests/perf/config_standins.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_config_standins.jsonl
python3 tools/step_variants.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_step_variants.jsonl
tools/pmc_issue_mix.sh $OUT/${R}_pmc_issue_mix.txt > /dev/null 2>&1
python3 tests/perf/eval_time.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_eval_time.jsonl
python3 tools/scatter_time.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_scatter_time.jsonl
python3 tools/scatter_kernel_time.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_scatter_kernel_time.txt
# the NMS scan's cycle stamps (profiling build) and the proof that the parity gates bite (damaged builds must fail)
python3 tools/build_variants.py prof="-DSCAN_PROFILE=1" > /dev/null 2>&1
GD3D_LIB=tools/variants/libgd3d_prof.so GD3D_HOST=python python3 tools/scan_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_nms_scan_stamps.txt
python3 tools/gate_bites.py $OUT/${R}_gate_bites.txt > /dev/null 2>&1
# host glue A/B at training sizes: the Python autograd.Function + ctypes layer against the optional C++ node, same box
GD3D_HOST=python python3 tests/perf/small_p_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_small_p_latency_python_glue.jsonl
GD3D_HOST=cpp python3 tests/perf/small_p_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_small_p_latency_cpp_glue.jsonl
# accuracy report of THIS build: the test module deletes any older file of that name before it runs and writes it only from
# the rows it really compared; a failed or empty run leaves no report and is said so loudly
rm -f $OUT/${R}_accuracy_report.txt
GD3D_ACCURACY_REPORT=$PWD/$OUT/${R}_accuracy_report.txt python3 -m pytest tests/test_gpu_gd_loss.py -m gpu -q -k 'pairs_against_reference_golden or config0' > $OUT/accuracy_pytest.log 2>&1
ACC_RC=$?
if [ $ACC_RC -ne 0 ] || [ ! -s $OUT/${R}_accuracy_report.txt ]; then
  echo "ACCURACY REPORT MISSING: pytest rc=$ACC_RC (see $OUT/accuracy_pytest.log)" | tee $OUT/${R}_accuracy_report.FAILED
  rm -f $OUT/${R}_accuracy_report.txt
fi
python3 tools/profile_summary.py $OUT $R
ls -la $OUT
