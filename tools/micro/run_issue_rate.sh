cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 120 ./tools/micro/issue_rate > gpurun_out/r3c_issue_rate.txt 2>&1 || exit 1
cat gpurun_out/r3c_issue_rate.txt
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/micro_pmc -o pmc --output-format csv -- ./tools/micro/issue_rate > gpurun_out/micro_pmc.log 2>&1 || { tail -5 gpurun_out/micro_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/micro_pmc/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"][:40], r["Grid_Size"], r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for (k, g, d), v in agg.items():
    if (k, g) in seen: continue
    seen.add((k, g))
    w = v["SQ_WAVE_CYCLES"]
    print(k, "grid", g, "INSTS_VALU %.3g" % v["SQ_INSTS_VALU"], "ACTIVE_INST_VALU/WAVE_CYCLES %.3f" % (v["SQ_ACTIVE_INST_VALU"] / w), "WAIT_INST_ANY/WC %.3f" % (v["SQ_WAIT_INST_ANY"] / w), "WAIT_ANY/WC %.3f" % (v["SQ_WAIT_ANY"] / w), "ACTIVE_VALU cycles per inst %.2f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"]), "WAVE_CYCLES/BUSY_CYCLES %.1f" % (w / v["SQ_BUSY_CYCLES"]))
PY
