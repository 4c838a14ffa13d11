import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
