"""Sum rocprofv3 counter_collection CSVs per kernel: python pmc_sum.py <dir> <kernel-substring>"""
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(k, tot[k], "dispatch-rows", n[k])
