"""Aggregate a rocprofv3 rocpd sqlite result (kernels view): python tools/prof_summary.py <db> [top]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1]); top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e3:.2f} ms")
for r in rows[:top]:
    nm = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:64]
    print(f"{nm:64s} n={r[1]:6d} total={r[2]/1e3:8.3f} ms avg={r[3]:8.2f} min={r[4]:7.2f} max={r[5]:8.2f} us {100*r[2]/tot:5.1f}%")
