#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.
gfx950: FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced stream -> doubled here
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  Units: the counters are in KiB."""
import collections, csv, json, sys


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            k = (r["Kernel_Name"], int(r["Grid_Size"]))
            d[k][0] += 1
            d[k][1] += float(r["Counter_Value"])
    return d


def main(fetch_csv, write_csv, out_json):
    fe, wr = agg(fetch_csv), agg(write_csv)
    out = {}
    for (name, grid), (n, f) in fe.items():
        w = wr.get((name, grid), [n, 0.0])[1]
        e = out.setdefault(name, [])
        e.append({"grid": grid, "launches": n, "fetch_raw_bytes_per_launch": f / n * 1024, "fetch_corrected_bytes_per_launch": 2 * f / n * 1024,
                  "write_bytes_per_launch": w / n * 1024})
    for name in out:
        out[name].sort(key=lambda e: -e["launches"] * (e["fetch_corrected_bytes_per_launch"] + e["write_bytes_per_launch"]))
    json.dump(out, open(out_json, "w"), indent=1)
    rows = sorted(((sum(e["launches"] * (e["fetch_corrected_bytes_per_launch"] + e["write_bytes_per_launch"]) for e in v), k) for k, v in out.items()), reverse=True)
    for t, k in rows[:12]:
        e = out[k][0]
        print(f"{k[:100]:100s} grid={e['grid']:9d} n={e['launches']:5d} fetch(x2)={e['fetch_corrected_bytes_per_launch']/1e6:9.1f} MB write={e['write_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
