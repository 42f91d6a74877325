export TMPDIR=/tmp; mkdir -p gpurun_out/r03 gpurun_out/profiles
C="rodent_amd/bin/rodent --scene tests/golden/cornell_box.obj --bench 2 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4 --target amdgpu-megakernel"
run() { name=$1; shift; timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/profiles/r03m_$name -o rodent -- $C > gpurun_out/profiles/r03m_$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
python scripts/pmc_digest.py gpurun_out/profiles r03m_ k_mega 2>&1 | grep -v "^\[" | tee gpurun_out/r03/mega_pmc.txt
