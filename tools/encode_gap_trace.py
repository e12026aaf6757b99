"""Where does the device idle during an end-to-end corpus encode?  Reads a rocprofv3 --kernel-trace rocpd database of
`python tools/encode_e2e_probe.py 0` and prints, for the LAST encode pass: busy time, idle time, the distribution of idle gaps
between consecutive kernels and the largest ones with the kernels around them.  python tools/encode_gap_trace.py results.db"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select d.start, d.end, s.display_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# the end-to-end pass = the last run of >= 1000 kernels without a gap of > 50 ms
segs, cur_seg = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 50e6:
        segs.append(cur_seg); cur_seg = []
    cur_seg.append(b)
segs.append(cur_seg)
def show(seg, label):
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy = 0; end = seg[0][0]; gaps = []
    for st, en, nm in seg:
        if st > end:
            gaps.append((st - end, end, nm))
        busy += max(0, en - max(st, end)); end = max(end, en)
    print(f"{label}: {len(seg)} kernels, wall {(t1-t0)/1e6:.1f} ms, busy {busy/1e6:.1f} ms, idle {(t1-t0-busy)/1e6:.1f} ms")
    for lo, hi in ((0, 2e3), (2e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e9)):
        g = [x[0] for x in gaps if lo <= x[0] < hi]
        print(f"   gaps {lo/1e3:7.0f}-{hi/1e3:7.0f} us: {len(g):6d}  total {sum(g)/1e6:8.2f} ms")
    for gsz, at, nm in sorted(gaps, reverse=True)[:8]:
        print(f"   gap {gsz/1e3:8.1f} us at {(at-t0)/1e6:8.2f} ms before {re.sub(r'[(<].*', '', nm)[:60]}")
big = sorted(segs, key=len, reverse=True)[:3]
for i, s in enumerate(sorted(big, key=lambda s_: s_[0][0])):
    show(s, f"segment {i}")
