#!/bin/bash
# round 3: counters of the A2 pipeline kernel (nam_kq_kernel; KP_NAME=nam_kp_kernel with NAM_HIP_KQ=0 for the four-waves-per-stage form; A2-Full, 256 streams) and, for comparison, nam_a1_p4_kernel: one resident launch of 300 steps
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for T in ${PROF_SET:-a2 c2}; do
  if [ $T = a2 ]; then ARGS="--model A2 --streams 256"; else ARGS="--config 2"; fi
  BENCH="python bench.py $ARGS --launch resident --steps 300 --warmup 30 --reps 1 --no-cpu-baseline --no-side-runs --no-other-configs --check 0 --spinup-ms 0"
  D=gpurun_out/prof_kp_$T
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- $BENCH > gpurun_out/prof_kp_bench_$T.json 2> gpurun_out/prof_kp_$T.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D -o pmc_fetch -- $BENCH > /dev/null 2>> gpurun_out/prof_kp_$T.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D -o pmc_write -- $BENCH > /dev/null 2>> gpurun_out/prof_kp_$T.err
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $D -o pmc_sq -- $BENCH > /dev/null 2>> gpurun_out/prof_kp_$T.err
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $D -o pmc_inst -- $BENCH > /dev/null 2>> gpurun_out/prof_kp_$T.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $D -o pmc_inst2 -- $BENCH > /dev/null 2>> gpurun_out/prof_kp_$T.err
  K=$([ $T = a2 ] && echo ${KP_NAME:-nam_kq_kernel} || echo nam_a1_p4_kernel)
  python scripts/resident_counters.py $D $K gpurun_out/counters_${T}_resident.json > /dev/null 2>&1
  python - $D $K <<'PY'
import collections, csv, glob, os, sys
root, kernel = sys.argv[1], sys.argv[2]
out = {}
for path in glob.glob(os.path.join(root, "**", "pmc_*_counter_collection.csv"), recursive=True):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if per:
        big = max(per.values(), key=lambda c: max(c.values()))
        out.update(big)
tr = [r for r in csv.DictReader(open(glob.glob(os.path.join(root, "**", "trace_kernel_trace.csv"), recursive=True)[0])) if kernel in r["Kernel_Name"]]
ns = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr)
print(kernel, "300-step launch:", ns / 1e3, "us =", ns / 300e3, "us per step")
for k in sorted(out):
    print(f"  {k:32s} {out[k] / 300:14.1f} per step")
PY
done 2>&1 | tee gpurun_out/r3_kp_counters.txt
