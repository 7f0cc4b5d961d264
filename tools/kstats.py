"""Print name / calls / average us of a rocprofv3 kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:44]:46s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} us")
