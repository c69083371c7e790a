"""Summarise rocprofv3 --pmc CSV output per kernel: mean of every counter over the dispatches of each kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("adypt::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, "dispatches", max(len(v) for v in acc[k].values()))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-36s mean %.6g  sum %.6g" % (c, sum(v) / len(v), sum(v)))
