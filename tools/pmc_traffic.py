"""Per-launch HBM traffic of every kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench command.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_hbm_traffic.json

Units and gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
FETCH_SIZE tallies 128-byte read requests of wide coalesced streams at 64 bytes, so read bytes = 2 x FETCH_SIZE x 1024;
WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(root, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {root}")
    for fn in files:
        with open(fn) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    a = acc[row["Kernel_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        f, n = fetch[k]
        w, nw = write.get(k, [0.0, 0])
        rd = 2.0 * f * 1024.0 / max(n, 1)
        wr = w * 1024.0 / max(nw, 1)
        out[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                  "raw_FETCH_SIZE_KiB_per_launch": f / max(n, 1), "raw_WRITE_SIZE_KiB_per_launch": w / max(nw, 1)}
    with open(sys.argv[3], "w") as fh:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-trace --no-fp8-leg",
                   "correction": "read = 2 x FETCH_SIZE x 1024 B (gfx950 128-B requests tallied at 64 B), write = WRITE_SIZE x 1024 B",
                   "kernels": out}, fh, indent=1)
    for k, v in list(out.items())[:8]:
        print(f"{v['launches']:6d}  rd {v['read_bytes_per_launch'] / 1e6:9.1f} MB  wr {v['write_bytes_per_launch'] / 1e6:8.1f} MB  {k[:90]}")


if __name__ == "__main__":
    main()
