"""Summarise rocprofv3 PMC passes into per-kernel HBM traffic per launch.

usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json [DEVICE_BATCH_ROOMS]
FETCH_DIR / WRITE_DIR hold the counter_collection.csv of `rocprofv3 --pmc FETCH_SIZE` and
`rocprofv3 --pmc WRITE_SIZE` runs of the same bench.py command (separate passes: the two counters do not
fit the TCC slots together).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE is in KiB-like units of 1 KB? -- no: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes;
on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes, so reads are doubled; WRITE_SIZE is exact.
"""
import csv, glob, json, os, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                a = acc[row["Kernel_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        fk = f / nf if nf else 0.0     # KB per launch as reported
        wk = w / nw if nw else 0.0
        out[k] = {
            "launches_fetch_pass": nf, "launches_write_pass": nw,
            "fetch_kb_reported": fk, "write_kb_reported": wk,
            "read_bytes_corrected": 2.0 * fk * 1024.0,   # gfx950: 128-B requests tallied as 64 B
            "write_bytes": wk * 1024.0,
            "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0,
        }
    out["_meta"] = {"device_batch_rooms": int(sys.argv[4]) if len(sys.argv) > 4 else 32}
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    rows = {k: v for k, v in out.items() if k != "_meta"}
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:25]:
        print("%10.2f MB/launch  (n=%d)  %s" % (v["hbm_bytes_per_launch"] / 1e6, v["launches_fetch_pass"], k[:110]))


if __name__ == "__main__":
    main()
