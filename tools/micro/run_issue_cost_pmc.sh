cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d gpurun_out/micro_pmc -o pmc --output-format csv -- ./tools/micro/issue_cost > gpurun_out/micro_pmc.log 2>&1 || { tail -5 gpurun_out/micro_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/micro_pmc/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"][:28], r["Grid_Size"], r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = {}
for (k, g, d), v in agg.items():
    last[(k, g)] = v
for (k, g), v in last.items():
    if g != "524288": continue
    w = v["SQ_WAVE_CYCLES"]
    print(f'{k:30s} ACTIVE_INST_VALU per instruction {v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"]:.2f} quad-cycles; GUI_ACTIVE {v.get("GRBM_GUI_ACTIVE", 0):.3g} cycles; instructions {v["SQ_INSTS_VALU"]:.3g}; ACTIVE_INST_VALU x4 / (GUI_ACTIVE x 1024 SIMDs) = {4 * v["SQ_ACTIVE_INST_VALU"] / max(1.0, v.get("GRBM_GUI_ACTIVE", 0) * 1024):.2f}')
PY
