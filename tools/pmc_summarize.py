"""Per-kernel HBM traffic from two rocprofv3 counter-collection runs of tools/pmc_probe.py (one with
--pmc FETCH_SIZE, one with --pmc WRITE_SIZE):  python tools/pmc_summarize.py <fetch.csv> <write.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB... of 1024 bytes? -- the unit is calibrated, not assumed: the
statistics kernel (colsumsq_kernel) streams exactly 2 * rows * K bytes with 16-byte loads, so the factor that maps
its FETCH_SIZE to that byte count is applied to every kernel's reads (on gfx950 wide streaming reads are tallied
at half their size); writes are taken at face value in the counter's unit of 1024 bytes... and checked the same
way against the transpose kernel, which writes exactly what it reads."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            agg[name][0] += 1
            agg[name][1] += float(row["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    # calibration: colsumsq_kernel reads 16 samples x 2048 rows x 4096 columns of bf16 per launch (probe shape)
    cal_name = next(k for k in fetch if "colsumsq_kernel" in k)
    known = 16 * 2048 * 4096 * 2
    per_launch = fetch[cal_name][1] / fetch[cal_name][0]
    read_unit = known / per_launch            # bytes per counter unit for wide streaming reads
    out = {"calibration": {"kernel": cal_name, "known_bytes": known, "counter_per_launch": per_launch,
                           "bytes_per_fetch_unit": read_unit, "bytes_per_write_unit": 1024.0}, "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        n = fetch.get(name, write.get(name))[0]
        rd = fetch[name][1] / fetch[name][0] * read_unit if name in fetch else None
        wr = write[name][1] / write[name][0] * 1024.0 if name in write else None
        out["kernels"][name] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k[:60]:60s} n={v['launches']:5d} read {0 if v['read_bytes_per_launch'] is None else v['read_bytes_per_launch']/1e6:10.2f} MB  "
              f"write {0 if v['write_bytes_per_launch'] is None else v['write_bytes_per_launch']/1e6:10.2f} MB")
    print("calibration:", out["calibration"])


if __name__ == "__main__":
    main()
