"""Print a slice of the kernel timeline (start/end in us, relative) from a rocpd database."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
lo, hi = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 60
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select d.start, d.end, d.queue_id, d.stream_id, d.grid_size_x, d.workgroup_size_x, s.display_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
rows = [r for r in rows if any(t in r[6] for t in ("scan_kernel", "merge_query", "prep_queries", "tiny_search", "convert_rows", "rescore"))]
rows = rows[-(hi):][: hi - lo] if lo == 0 else rows[lo:hi]
t0 = rows[0][0]
for st, en, q, sid, gx, wx, name in rows:
    nm = re.sub(r"\(.*", "", name)[:28]
    print(f"{(st-t0)/1e3:10.1f} -> {(en-t0)/1e3:10.1f}  dur {(en-st)/1e3:8.1f}  q{q} s{sid}  grid {gx//max(wx,1):4d}  {nm}")
