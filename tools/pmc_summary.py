"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel.
  python tools/pmc_summary.py <dir> <kernel substring>                         one line per file and counter
  python tools/pmc_summary.py --json profiles/roofline_pmc.json <key> <substring> <fetch dir> <write dir>
      adds / replaces entry <key>: HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 reports half the bytes of a wide
      coalesced read: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both KiB per dispatch, from separate passes"""
import collections, csv, glob, json, os, sys


def collect(d, sub):
    out = {}
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(a in r["Kernel_Name"] for a in sub.split("|")):     # `a|b`: either spelling (mangled / demangled)
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = (f, v)
    return out


if sys.argv[1] == "--json":
    path, key, sub, dfetch, dwrite = sys.argv[2:7]
    fe, wr = collect(dfetch, sub)["FETCH_SIZE"][1], collect(dwrite, sub)["WRITE_SIZE"][1]
    fetch, write = sum(fe) / len(fe) * 1024, sum(wr) / len(wr) * 1024
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = {"traffic_bytes": 2 * fetch + write, "fetch_size_bytes_reported": fetch, "fetch_correction": 2.0, "write_size_bytes": write,
                 "dispatches": len(fe), "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python3 bench.py --roofline-only"}
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(key, data[key])
else:
    for k, (f, v) in collect(sys.argv[1], sys.argv[2]).items():
        print(f"{f}: {k}: dispatches {len(v)} mean {sum(v)/len(v):.1f} min {min(v):.1f} max {max(v):.1f}")
