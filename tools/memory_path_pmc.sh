#!/bin/bash
# usage: tools/memory_path_pmc.sh TAG  -- counter passes over tools/one_frame.py for the address and L1 units of the vector memory
# path (TA, TCP, TD): is the tree kernel bound by the rate at which a CU turns per-lane addresses into cache-line requests?
# -> gpurun_out/mem_TAG/<scene>_N.txt (sums per kernel)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/mem_$tag
rm -rf "$out"; mkdir -p "$out"
export FRAMES=2 RT_GPU_PARKED_PERCENT=100
for scene in dog_room dogs_and_cat; do
  n=0
  for counters in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
                  "TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
                  "TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    n=$((n+1))
    SCENE=$scene timeout -s KILL 150 rocprofv3 --pmc $counters -d "$out/p" -o pmc --output-format csv -- python3 tools/one_frame.py > "$out/${scene}_$n.log" 2>&1 || { echo "pass $n of $scene failed"; tail -5 "$out/${scene}_$n.log"; rm -rf "$out/p"; continue; }
    f=$(find "$out/p" -name '*counter_collection.csv' | head -1)
    python3 tools/pmc_summary.py "$f" > "$out/${scene}_$n.txt"
    rm -rf "$out/p"
  done
done
echo done
