import csv,glob,sys
fs=glob.glob("/tmp/prof/**/*kernel_stats.csv",recursive=True)
print(fs)
rows=list(csv.DictReader(open(fs[0])))
for r in rows[:10]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
