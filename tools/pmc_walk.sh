for m in "pq 128" "f32 2048"; do
  tag=$(echo $m | tr ' ' '_')
  tools/pmc_run.sh walk_${tag}_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" python3 tools/walk_prof.py 1000000 $m
  tools/pmc_run.sh walk_${tag}_b "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" python3 tools/walk_prof.py 1000000 $m
  tools/pmc_run.sh walk_${tag}_c "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" python3 tools/walk_prof.py 1000000 $m
  tools/pmc_run.sh walk_${tag}_d "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" python3 tools/walk_prof.py 1000000 $m
done
grep -h "hnsw_search" gpurun_out/pmc/walk_*.csv
