import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# take a window in the middle
n = len(rows)
mid = rows[n // 2: n // 2 + 90]
for r in mid:
    print("%10.1f %10.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:50]))
