"""Medians of the MFMA / LDS counters of one kernel from two rocprofv3 --pmc passes:
python scripts/counters_summary.py <mfma_busy+grbm.csv> <mops+lds.csv> <kernel substring> <kernel_stats.csv> <out.json>"""
import csv, json, statistics, sys
from collections import defaultdict

def medians(path, kernel):
    vals = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row['Kernel_Name']:
                vals[row['Counter_Name']].append(float(row['Counter_Value']))
    return {k: statistics.median(v) for k, v in vals.items()}, max(len(v) for v in vals.values())

a, n = medians(sys.argv[1], sys.argv[3])
b, _ = medians(sys.argv[2], sys.argv[3])
avg_us = None
with open(sys.argv[4]) as f:
    for row in csv.DictReader(f):
        if sys.argv[3] in row['Name']:
            avg_us = float(row['AverageNs']) / 1e3
flop = b['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512
out = {'kernel': sys.argv[3], 'avg_us_rocprof': avg_us, 'launches': n, **a, **b, 'mfma_flop_from_counter': flop,
       'MfmaUtil_percent': 100.0 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8 * 1024),
       'lds_conflict_percent': 100.0 * b['SQ_LDS_BANK_CONFLICT'] / b['SQ_LDS_IDX_ACTIVE'],
       'note': 'medians over the dispatches of `python scripts/roofline_kernel.py`; separate --pmc passes (MFMA busy + GRBM, MFMA ops + LDS); '
               'GRBM_GUI_ACTIVE is summed over the 8 XCDs; MfmaUtil = MFMA busy cycles / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); MOPS x 512 = FLOP'}
json.dump(out, open(sys.argv[5], 'w'), indent=1)
print(json.dumps(out))
