"""Prints this repo's kernels from a rocprofv3 kernel_stats.csv: short name, calls, average microseconds."""
import csv
import glob
import sys

root = sys.argv[1]
for f in glob.glob(root + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sdm::" in r["Name"]:
            name = r["Name"].split("(")[0].replace("void ", "")
            print("%-34s calls %5s  avg %10.1f us  total %10.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                     float(r["TotalDurationNs"]) / 1e3))
