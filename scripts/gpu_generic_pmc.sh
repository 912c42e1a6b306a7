#!/bin/bash
# developer tool: instruction-mix counters of the generic kernel on wavenet_a2_max (512 streams)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --model ${1:-wavenet_a2_max} --streams ${2:-512} --launch block --steps 100 --warmup 10 --reps 1 --no-cpu-baseline --no-side-runs --spinup-ms 0 --check 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_gen -o pmc1 -- $BENCH > /dev/null 2> gpurun_out/pmc_gen1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d gpurun_out/prof_gen -o pmc2 -- $BENCH > /dev/null 2> gpurun_out/pmc_gen2.err
python - <<'PY'
import csv, glob, collections
for tag in ("pmc1", "pmc2"):
    agg = collections.defaultdict(float); n = set()
    for f in glob.glob(f"gpurun_out/prof_gen/**/{tag}_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "nam_generic" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    print(tag, "dispatches", len(n), {k: round(v / max(len(n), 1)) for k, v in sorted(agg.items())})
PY
