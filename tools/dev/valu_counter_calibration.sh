cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $GRAFT_REPO_ROOT/tools/ubench/valu_table.hip -o /tmp/valu_table 2>/dev/null
rm -rf /tmp/cal; rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/cal -o cal -- /tmp/valu_table > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/cal/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()):
    g=lambda n: max(v[n]) if v[n] else 0   # the long launch of each kernel
    if g("SQ_BUSY_CU_CYCLES"):
        print("%-18s insts %.3g active_valu/busy_cu %.3f  active_valu per inst %.3f quad-cycles  busy_cu %.3g gui %.3g" % (k, g("SQ_INSTS_VALU"), g("SQ_ACTIVE_INST_VALU")/g("SQ_BUSY_CU_CYCLES"), g("SQ_ACTIVE_INST_VALU")/max(1,g("SQ_INSTS_VALU")), g("SQ_BUSY_CU_CYCLES"), g("GRBM_GUI_ACTIVE")))
PY
