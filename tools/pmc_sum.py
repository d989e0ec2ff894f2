"""Sum a PMC counter per kernel name from rocprofv3 --pmc CSV output: usage pmc_sum.py <dir> [counter]"""
import csv, glob, sys, collections
counter = sys.argv[2] if len(sys.argv) > 2 else "SQ_INSTS_VALU"
tot, cnt = collections.Counter(), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        nm = r["Kernel_Name"].split("(")[0].replace("void vdf::", "").replace("vdf::", "")
        tot[nm] += float(r["Counter_Value"]); cnt[nm] += 1
for nm, v in tot.most_common(16):
    print(f"{nm[:40]:40s} launches {cnt[nm]:4d}  avg {v / cnt[nm] / 1e6:10.2f} M")
