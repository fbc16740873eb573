#!/bin/bash
# usage: tools/profile_heavy.sh TAG [SCENE]  -- counter passes over tools/one_frame.py (plain frames of the mesh-heavy scene, nothing
# else) -> gpurun_out/heavy_TAG/pmc_N.csv + .txt and counters_SCENE.json (copy into profiles/ with the lines at the end)
tag=$1; scene=${2:-dogs_and_cat}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/heavy_$tag
rm -rf "$out"; mkdir -p "$out"
export SCENE=$scene FRAMES=2
# full lists from the first call: one launch that starts paths per frame (with the default smaller lists the first frame of a
# handle enqueues safety-net passes, whose empty launches would be counted as frames)
export RT_GPU_PARKED_PERCENT=100
n=0
for counters in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
                "FETCH_SIZE" "WRITE_SIZE" \
                "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" \
                "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  n=$((n+1))
  timeout -s KILL 150 rocprofv3 --pmc $counters -d "$out/pmc_$n" -o pmc --output-format csv -- python3 tools/one_frame.py > "$out/pmc_$n.log" 2>&1 || { echo "pmc pass $n ($counters) failed"; tail -5 "$out/pmc_$n.log"; continue; }
  f=$(find "$out/pmc_$n" -name '*counter_collection.csv' | head -1)
  cp "$f" "$out/pmc_$n.csv"
  python3 tools/pmc_summary.py "$f" > "$out/pmc_$n.txt"
  rm -rf "$out/pmc_$n"
done
python3 tools/make_counters_json.py "${tag}_heavy" "$scene 1024x1024 64 spp" "$out"/pmc_[0-9].csv > "$out/counters_$scene.json" || echo "counters json failed"
echo done
# keep: for f in gpurun_out/heavy_TAG/pmc_?.csv; do n=$(basename $f .csv); gzip -9 -c $f > profiles/TAG_heavy_$n.csv.gz; cp gpurun_out/heavy_TAG/$n.txt profiles/TAG_heavy_$n.txt; done
#       cp gpurun_out/heavy_TAG/counters_SCENE.json profiles/
