"""Turns rocprofv3 output directories into the small summaries kept under profiles/.

    python profiles/summarize.py stats  <dir with *_kernel_stats.csv>        > profiles/rNN_..._kernel_stats.csv
    python profiles/summarize.py pmc    <FETCH_SIZE dir> <WRITE_SIZE dir>     > profiles/rNN_pmc_traffic.json
    python profiles/summarize.py counters <dir of any --pmc pass>             > profiles/rNN_pmc_<what>.json

HBM bytes follow the MI355X counter notes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half the bytes of a wide coalesced streaming read (128-byte requests tallied at
64 B), so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import collections
import csv
import glob
import json
import sqlite3
import sys


def per_kernel(directory, counter):
    """Per-launch counter values by kernel, from rocprofv3's CSV output or its default rocpd database."""
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{directory}/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?",
                                      (counter,)):
            agg[name.split("(")[0].replace("void ", "")].append(float(value))
    return {k: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in agg.items()}


def main():
    mode = sys.argv[1]
    if mode == "stats":
        for f in glob.glob(f"{sys.argv[2]}/**/*kernel_stats.csv", recursive=True):
            sys.stdout.write(open(f).read())
        for f in glob.glob(f"{sys.argv[2]}/**/*.db", recursive=True):
            w = csv.writer(sys.stdout)
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            for row in sqlite3.connect(f).execute(
                    "select name, total_calls, total_duration, average, percentage from top_kernels"):
                w.writerow(row)
    elif mode == "counters":
        out = {"units": "mean counter value per launch, summed over the device by rocprofv3", "kernels": {}}
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{sys.argv[2]}/**/*.db", recursive=True):
            for name, counter, value in sqlite3.connect(f).execute(
                    "select kernel_name, counter_name, value from counters_collection"):
                if "picp" in name:
                    agg[name.split("(")[0].replace("void ", "")][counter].append(float(value))
        for k in sorted(agg):
            out["kernels"][k] = {c: sum(v) / len(v) for c, v in sorted(agg[k].items())}
            out["kernels"][k]["launches"] = len(next(iter(agg[k].values())))
        json.dump(out, sys.stdout, indent=1)
        sys.stdout.write("\n")
    elif mode == "pmc":
        fetch, write = per_kernel(sys.argv[2], "FETCH_SIZE"), per_kernel(sys.argv[3], "WRITE_SIZE")
        out = {"units": "FETCH_SIZE / WRITE_SIZE in KiB per launch; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
               "kernels": {}}
        for k in sorted(fetch):
            if "picp" not in k:
                continue
            w = write.get(k, {"mean": 0.0})
            out["kernels"][k] = {"FETCH_SIZE_KiB": fetch[k]["mean"], "WRITE_SIZE_KiB": w["mean"],
                                 "launches": fetch[k]["launches"],
                                 "hbm_bytes_per_launch": (2.0 * fetch[k]["mean"] + w["mean"]) * 1024.0}
        json.dump(out, sys.stdout, indent=1)
        sys.stdout.write("\n")


if __name__ == "__main__":
    main()
