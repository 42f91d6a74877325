#!/bin/bash
# round 6, call B: the GPU suite again (mixed image + segment lists), then what the deferred-leaf kernels do with their time:
# drain statistics of the instrumented builds and SQ / TA counters of the default against defer-p3d16 on the random set
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5
OUT=gpurun_out/r06
RODENT_HIP_LAB=1 timeout 600 python scripts/defer_experiment.py --only 'stats-defer|defer-p3d16|defer-p3d32-' --no-oracle 2>&1 | tee $OUT/defer_experiment_b.txt | tail -12
for V in top defer-p3d16; do
  IDX=$(RODENT_HIP_LAB=1 python -c "from rodent_amd import abi; print(abi.variants(2).index('$V'))")
  for G in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "sq3 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "ta1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
    set -- $G; NAME=$1; shift
    RODENT_HIP_LAB=1 timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${V}_$NAME -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --only random --variant $IDX > $OUT/pmc_${V}_$NAME.log 2>&1 || echo "pass $V $NAME failed"
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r06/pmc_*_*")):
    if not d.endswith(("sq1", "sq3", "ta1")): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            if "k_bvh2_top" in k:
                print(d.split("/")[-1], k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "dispatches", len(next(iter(c.values()))))
PY
