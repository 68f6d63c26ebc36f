"""Per-(kernel, grid) totals from a rocprofv3 results database (development aid): prof_db_grid.py <dir> <name substring>."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
q = f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {disp} d join {sym} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' group by 1,2,3,4 order by 7 desc limit 30"
for r in c.execute(q):
    print(f"{r[0][:60]:60s} grid {r[1]}x{r[2]}x{r[3]} calls {r[4]:7d} avg {r[5]:8.2f} us total {r[6]:9.2f} ms")
