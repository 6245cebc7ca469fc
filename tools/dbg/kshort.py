"""Short form of a rocprofv3 kernel_stats.csv (tools/kstats.sh): name, calls, average / min / max microseconds.  kshort.py file.csv [substring ...]"""
import csv, sys
pats = sys.argv[2:]
for r in csv.reader(open(sys.argv[1])):
    if not r or r[0] == "Name":
        continue
    n = r[0].replace("void ", "").replace("qh::", "").split("(")[0]
    if pats and not any(p in n for p in pats):
        continue
    try:
        print("%-70s calls %4s avg %9.1f min %9.1f max %9.1f us" % (n[:70], r[1], float(r[3]) / 1e3, float(r[5]) / 1e3, float(r[6]) / 1e3))
    except (ValueError, IndexError):
        pass
