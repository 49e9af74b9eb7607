export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_fd
SAEV_FIN_DEBUG=1 timeout 240 rocprofv3 --kernel-trace -d /tmp/prof_fd -o run -- python bench.py --steps 40 --warmup 5 --pretrain-steps 1500 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras > /tmp/fd.log 2>&1
echo "rc $?"; tail -2 /tmp/fd.log | cut -c1-200
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/prof_fd/**/*.db', recursive=True)[0])
t = [r[0] for r in db.execute("select name from sqlite_master where type='table'") if r[0].startswith('rocpd_kernel_dispatch')][0]
suf = t[len('rocpd_kernel_dispatch'):]
rows = db.execute(f"select d.grid_size_x, d.end-d.start from {t} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id where s.kernel_name like '%finalize_light%' order by d.start").fetchall()
rows = rows[-80:]
import collections
ev = [dt / 1e3 for i, (g, dt) in enumerate(rows) if i % 2 == 0]
od = [dt / 1e3 for i, (g, dt) in enumerate(rows) if i % 2 == 1]
print("first of each pair mean us %.1f, second %.1f" % (sum(ev) / len(ev), sum(od) / len(od)), [round(x) for x in ev[:6]], [round(x) for x in od[:6]])
PY
