#!/bin/bash
# Run ON THE GPU BOX: SQ counters + kernel trace of the fused Winograd kernel and its ablations on one layer (scripts/wf_bench.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LAYERS=${LAYERS:-conv_4_2} ABLS=${ABLS:-1,2,4,7}
OUT=${OUT:-gpurun_out/wf_pmc}
rm -rf $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/sq -- python3 scripts/wf_bench.py > $OUT.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/wf_bench.py >> $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/sq/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if "wino_fused" in r["Kernel_Name"] or "gemm_planes" in r["Kernel_Name"]:
            agg[r['Kernel_Name'].split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob('$OUT/trace/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if "wino_fused" in r["Kernel_Name"] or "gemm_planes" in r["Kernel_Name"]:
            dur[r['Kernel_Name'].split('(')[0][-60:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = sorted(dur.get(k, [0]))[len(dur.get(k, [0])) // 2]
    clk = m['GRBM_GUI_ACTIVE'] / 8 / d / 1e3 if d else 0
    print(f"{k:60s} dur {d:8.1f} us  clock {clk:5.2f} GHz  mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (m['GRBM_GUI_ACTIVE'] / 8):.3f}  "
          f"wait_any {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f} wait_inst {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} active {m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} lds_conf {m['SQ_LDS_BANK_CONFLICT']:.3g}")
PY
