cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pl; rocprofv3 --kernel-trace -d /tmp/pl -o p -- python $R/bench.py --mode train --no-cpu-baseline --steps 6 --warmup 2 --traffic off > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/pl/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'info_kernel_symbol' in t][0]
rows = list(con.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# last step: find the last adam_multi and take kernels between the previous adam and it
ad = [i for i,r in enumerate(rows) if 'adam_multi' in r[0]]
lo, hi = ad[-2]+1, ad[-1]+1
step = rows[lo:hi]
print("kernels in step:", len(step), "span us", (step[-1][2]-step[0][1])/1e3)
pw = [(r[0][:60], (r[2]-r[1])/1e3, r[3]//r[6], r[4], r[5]) for r in step if 'pw_conv_kernel' in r[0]]
print("pw_conv launches", len(pw), "total us", sum(x[1] for x in pw))
for i,x in enumerate(pw): print(i, "%-45s %7.1f us  grid %d x %d x %d" % x)
PY
