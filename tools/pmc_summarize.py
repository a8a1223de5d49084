"""Per-kernel HBM traffic from two rocprofv3 counter-collection runs of the same single-threaded command (one
with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE):

    python tools/pmc_summarize.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <calib bytes> [commit]

The counters' unit is calibrated, not assumed: the statistics kernel (colsumsq_kernel / colsumsq_tall_kernel) streams exactly 2 bytes per
activation element with 16-byte loads, so <calib bytes> = the bytes it read over the whole run (bench.py, one step:
128 * 2048 * 2 * sum of the 7 linears' input widths = 18 656 788 480) fixes the bytes per FETCH_SIZE unit for wide
streaming reads (on gfx950 these are tallied at half their size, which the calibration absorbs).  WRITE_SIZE is
taken in its documented unit of 1024 bytes.  Output: read / write bytes per launch for every kernel."""
import csv
import json
import os
import sys
from collections import defaultdict


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            agg[name][0] += 1
            agg[name][1] += float(row["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    # (the statistics kernel has two forms, colsumsq_kernel and colsumsq_tall_kernel<n>: same bytes, both count)
    cal = [k for k in fetch if "colsumsq_" in k]
    if not cal:
        raise SystemExit("no launch of the statistics kernel (colsumsq_*) in " + sys.argv[1])
    cal_name = " + ".join(cal)
    cal_launches, cal_counter = sum(fetch[k][0] for k in cal), sum(fetch[k][1] for k in cal)
    # "<bytes>/launch": the average bytes per launch of the calibration kernel instead of the run's total
    known = float(sys.argv[4][:-7]) * cal_launches if sys.argv[4].endswith("/launch") else float(sys.argv[4])
    read_unit = known / cal_counter
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from slim_amd.build import source_fingerprint
    # the library sources the counters were collected on (run this right after the counter passes, on the same tree) and,
    # when given, the commit that holds them: bench.py quotes these figures only while the fingerprint still matches
    out = {"source_fingerprint": source_fingerprint(), "commit": sys.argv[5] if len(sys.argv) > 5 else None,
           "calibration": {"kernel": cal_name, "known_bytes_total": known, "counter_total": cal_counter,
                           "bytes_per_fetch_unit": read_unit, "bytes_per_write_unit": 1024.0}, "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        n = fetch.get(name, write.get(name))[0]
        rd = fetch[name][1] / fetch[name][0] * read_unit if name in fetch else None
        wr = write[name][1] / write[name][0] * 1024.0 if name in write else None
        out["kernels"][name] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        rd, wr = v["read_bytes_per_launch"] or 0.0, v["write_bytes_per_launch"] or 0.0
        print(f"{k[:70]:70s} n={v['launches']:5d} read {rd / 1e6:10.2f} MB  write {wr / 1e6:10.2f} MB")
    print("calibration:", out["calibration"])


if __name__ == "__main__":
    main()
