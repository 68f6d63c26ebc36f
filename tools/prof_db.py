"""Per-kernel totals from a rocprofv3 results database (development aid): prof_db.py <dir> [runs to divide by]."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {disp} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit 24"
print(f"{'kernel':72s} {'calls':>7s} {'avg us':>9s} {'ms/run':>8s}")
for r in c.execute(q):
    print(f"{r[0][:72]:72s} {r[1]:7d} {r[2]:9.1f} {r[3] / div:8.2f}")
