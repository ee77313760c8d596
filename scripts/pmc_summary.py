"""Reduce two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass over
`python scripts/roofline_kernel.py`) to the per-launch HBM traffic of the roofline kernel, as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: counters are KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide
coalesced reads at 64 B, so it is doubled; WRITE_SIZE is exact.
Usage: python scripts/pmc_summary.py <fetch.csv> <write.csv> <kernel-name-substring> <out.json>"""
import csv, json, statistics, sys

fetch_csv, write_csv, needle, out = sys.argv[1:5]


def values(path, counter):
    vals = []
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if needle in row['Kernel_Name'] and row['Counter_Name'] == counter:
                vals.append(float(row['Counter_Value']))
    return vals


res = {}
for path, counter in ((fetch_csv, 'FETCH_SIZE'), (write_csv, 'WRITE_SIZE')):
    v = values(path, counter)
    assert v, f'no {counter} rows for a kernel matching {needle!r} in {path}'
    res[counter] = {'dispatches': len(v), 'median_raw_KiB': statistics.median(v), 'min_raw_KiB': min(v), 'max_raw_KiB': max(v)}
rd = res['FETCH_SIZE']['median_raw_KiB'] * 1024 * 2
wr = res['WRITE_SIZE']['median_raw_KiB'] * 1024
res.update(note='rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python scripts/roofline_kernel.py`; '
                'KiB counters; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950); WRITE_SIZE exact',
           kernel=needle, hbm_read_bytes=rd, hbm_write_bytes=wr, traffic_bytes_per_launch=rd + wr,
           algorithmic_bytes_per_launch=2 * 64 * 80 * 80 * 64 * 2 + 64 * 64 * 9 * 2)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
