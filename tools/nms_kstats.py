"""Per-kernel averages of the rbox:: kernels from a rocprofv3 --kernel-trace --stats output directory.  usage: tools/nms_kstats.py DIR"""
import csv, glob, sys
for p in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'rbox' in r['Name']:
            print(f"{r['Name'].split('(')[0][-40:]:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:7.2f}  max {float(r['MaxNs'])/1e3:7.2f}")
