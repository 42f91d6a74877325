#!/bin/bash
# Lab: the library built with other instruction-scheduler options of the AMDGPU backend (the refill loop runs 7 % slower when it is compiled under
# the chunk loop's register budget, profiles/r04_sweep_auto.log: is there a global option that schedules both well?).
#   build (container):  scripts/flags_experiment.sh build
#   run (GPU box):      scripts/flags_experiment.sh run
cd "$(dirname "$0")/.."
SETS=("base:" "maxilp:-mllvm -amdgpu-sched-strategy=max-ilp" "maxclause:-mllvm -amdgpu-sched-strategy=max-memory-clause" "nounclustered:-mllvm -amdgpu-disable-unclustered-high-rp-reschedule"
      "noclustered:-mllvm -amdgpu-disable-clustered-low-occupancy-reschedule" "relaxed:-mllvm -amdgpu-schedule-relaxed-occupancy" "bias0:-mllvm -amdgpu-schedule-metric-bias=0" "trackers:-mllvm -amdgpu-use-amdgpu-trackers")
if [ "$1" = build ]; then
    mkdir -p build/flags
    for s in "${SETS[@]}"; do
        name=${s%%:*}; flags=${s#*:}
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -Iinclude $flags \
            -DRODENT_HIP_SOURCE_DIGEST="\"flags-$name\"" -DRODENT_HIP_LAB -shared rodent_amd/csrc/traversal.hip rodent_amd/csrc/render.hip rodent_amd/csrc/services.hip rodent_amd/host/image.cpp -lz \
            -o rodent_amd/lib/librodent_hip_flags_$name.so 2>&1 | grep -E "error" | head -3 &
    done
    wait; ls -la rodent_amd/lib/
else
    export TMPDIR=/tmp
    for s in "${SETS[@]}"; do
        name=${s%%:*}
        echo "== $name"
        RODENT_HIP_LAB=1 RODENT_HIP_LIB=rodent_amd/lib/librodent_hip_flags_$name.so timeout 300 python scripts/sweep_auto.py --steps 30 --variants top,top-nohint,refill 2>&1 | grep -A3 "^closest"
        RODENT_HIP_LIB=rodent_amd/lib/librodent_hip_flags_$name.so timeout 300 python scripts/frame_rate.py --spp 32 2>&1 | tail -1
    done
fi
