#!/bin/bash
# round 6, call J: the node's child ids fetched for lanes on a node only (exp_idsnodes: -DRODENT_JOINT_IDS_NODES_ONLY=1) against the default build:
# parity of both suites' traversal / renderer cores, traversal ABI A/B, frame-rate A/B.  (The macro was an experiment patch to joint_fetch / joint_fetch_off --
# s_and_b64 exec, exec, node_mask + s_cbranch_execz in front of the dwordx2 load -- that lost and was not committed: LAB_NOTES 12.7.)
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
OUT=gpurun_out/r06
RODENT_HIP_LIB=rodent_amd/lib/exp_idsnodes.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_atrium.py tests/test_gpu_render.py -m gpu -q -x -k "not bench_py and not cli and not through_indices and not up_front" 2>&1 | tail -3
( for rep in 1 2; do for lib in "" rodent_amd/lib/exp_idsnodes.so; do echo "== RODENT_HIP_LIB=$lib"; RODENT_HIP_LIB=$lib timeout 600 python scripts/defer_experiment.py --only 'nothing-matches' --big --steps 40 --no-oracle; done; done ) 2>&1 | grep -v "amdgpu.ids\|^scene\|^variant" | tee $OUT/ids_nodes_only_traversal.txt
( for rep in 1 2; do for lib in "" rodent_amd/lib/exp_idsnodes.so; do echo "== RODENT_HIP_LIB=$lib atrium 3840x2160 x 64 spp"; RODENT_HIP_LIB=$lib timeout 600 python scripts/frame_rate.py --spp 64; done; done
  for lib in "" rodent_amd/lib/exp_idsnodes.so; do echo "== RODENT_HIP_LIB=$lib gallery 16 spp"; RODENT_HIP_LIB=$lib timeout 600 python scripts/frame_rate.py --scene gallery --spp 16; done
  for lib in "" rodent_amd/lib/exp_idsnodes.so; do echo "== RODENT_HIP_LIB=$lib plant 16 spp"; RODENT_HIP_LIB=$lib timeout 900 python scripts/frame_rate.py --scene plant --spp 16; done ) 2>&1 | grep -v "amdgpu.ids\|Missing material" | tee $OUT/ids_nodes_only_render.txt
