import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:24]: print("%-80s %7s calls %9.2f ms %5.1f%% avg %8.1f us" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["Percentage"]), float(r["AverageNs"])/1e3))
