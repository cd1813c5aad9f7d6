"""Summarise rocprofv3 --pmc output: per kernel and counter, launches / total / per launch.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> > profiles/<name>.csv"""
import collections, csv, glob, os, re, sys

rows = collections.OrderedDict()
for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            key = (name, r["Counter_Name"])
            ent = rows.setdefault(key, {"ids": set(), "total": 0.0})
            ent["ids"].add(r["Dispatch_Id"])
            ent["total"] += float(r["Counter_Value"])
print("kernel,counter,launches,total,per_launch")
for (name, counter), ent in sorted(rows.items()):
    n = len(ent["ids"])
    print("%s,%s,%d,%.3f,%.3f" % (name, counter, n, ent["total"], ent["total"] / n))
