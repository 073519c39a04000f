"""Summarise rocprofv3 --pmc output: mean counter value per launch of each kernel matching a substring.
Usage: python tools/pmc_summary.py <dir> <kernel-substring> [...more dirs]"""
import csv, glob, json, os, sys
from collections import defaultdict

def summarise(dirs, needle):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            for row in csv.DictReader(open(f)):
                if needle not in row['Kernel_Name']:
                    continue
                key = (row['Dispatch_Id'], row['Counter_Name'])
                per_dispatch[key] += float(row['Counter_Value'])
                names[row['Dispatch_Id']] = row['Kernel_Name'].split('(')[0][-60:]
            for (disp, ctr), v in per_dispatch.items():
                acc[names[disp]][ctr].append(v)
    return {k: {c: {'mean': sum(v) / len(v), 'launches': len(v)} for c, v in ctrs.items()} for k, ctrs in acc.items()}

if __name__ == '__main__':
    print(json.dumps(summarise(sys.argv[3:] + [sys.argv[1]], sys.argv[2]), indent=1))
