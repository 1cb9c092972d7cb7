"""Aggregate a rocprofv3 rocpd sqlite result (kernels view): python tools/prof_summary.py <db> [top] [--by-grid]
--by-grid splits every kernel by its grid size (threads): the skinny GEMMs of different matrices are one template instantiation."""
import re, sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
by_grid = "--by-grid" in sys.argv
c = sqlite3.connect(args[0]); top = int(args[1]) if len(args) > 1 else 20
grp = "name, grid_x" if by_grid else "name"
sel = "name, grid_x," if by_grid else "name, 0,"
rows = c.execute(f"select {sel} count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by {grp} order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"total kernel time {tot/1e3:.2f} ms")
for r in rows[:top]:
    nm = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:64]
    g = f" grid={r[1]:<7d}" if by_grid else ""
    print(f"{nm:64s}{g} n={r[2]:6d} total={r[3]/1e3:8.3f} ms avg={r[4]:8.2f} min={r[5]:7.2f} max={r[6]:8.2f} us {100*r[3]/tot:5.1f}%")
