"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/) per kernel -> profiles/r02_pmc_summary.json.
HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE
reads exactly half the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), so it is doubled."""
import csv, glob, json, collections, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = collections.defaultdict(dict)
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
for tag, suffix in [(t, sfx) for sfx in ('', '_bf16') for t in ('fetch', 'write', 'sq')]:
    files = glob.glob(os.path.join(root, 'gpurun_out', f'pmc_{tag}{suffix}', '*', '*_counter_collection.csv'))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files.sort(key=os.path.getmtime)           # merged scratch dirs keep older passes: take the newest
    for r in csv.DictReader(open(files[-1])):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        if suffix and ('launches_' + tag) in out.get(k, {}) and not out[k].get('from' + suffix):
            continue                    # a kernel of the fp32 headline run keeps the headline run's counters
        if suffix:
            out[k]['from' + suffix] = 1
        for c, v in cs.items():
            out[k][c] = sum(v) / len(v)
            out[k]['launches_' + tag] = len(v)
res = {}
for k, d in out.items():
    e = dict(d)
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        e['hbm_bytes_per_launch'] = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and d.get('GRBM_GUI_ACTIVE'):
        # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
        e['mfma_pipe_util'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (d['GRBM_GUI_ACTIVE'] / 8)
    res[k] = e
json.dump(res, open(os.path.join(root, 'profiles', os.environ.get('PMC_OUT', 'r03_pmc_summary.json')), 'w'), indent=1, sort_keys=True)
for k in sorted(res, key=lambda k: -res[k].get('hbm_bytes_per_launch', 0))[:14]:
    e = res[k]
    print(f"{k[:70]:70s} hbm/launch {e.get('hbm_bytes_per_launch', 0)/1e6:9.2f} MB  mfma_util {e.get('mfma_pipe_util', 0):.3f}  lds_conflict {e.get('SQ_LDS_BANK_CONFLICT', 0):.3g}")
