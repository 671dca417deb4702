"""Print calls / average µs / total ms of the kernels whose name contains any of the given substrings, from a rocprofv3
kernel_stats.csv:  python tools/kstat.py stats.csv gemm_p8 rmsnorm"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2:]):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f'{name[:60]:60s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.2f} us {int(r["TotalDurationNs"]) / 1e6:9.2f} ms')
