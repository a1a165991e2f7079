"""ordered kernel list of one steady-state step from a rocprofv3 --kernel-trace DB (start offset us, duration us,
queue, grid, name).  usage: steplist.py <dir> out.txt"""
import sys, glob, sqlite3, re
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, queue_id from kernels order by start"))
short = lambda n: re.sub(r'\(anonymous namespace\)::|vlfb::|void |unsigned short|__hip_bfloat16', lambda m: {'unsigned short': 'bf16'}.get(m.group(0), ''), n).split('(')[0][:100]
sgd = [i for i, r in enumerate(rows) if 'sgd_kernel' in r[0]]
lo, hi = sgd[-4] + 1, sgd[-3] + 1
step = rows[lo:hi]
t0 = step[0][1]
qs = sorted(set(r[7] for r in step))
with open(sys.argv[2], 'w') as f:
    for r in step:
        f.write('%9.1f %8.1f q%d %7d %s\n' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, qs.index(r[7]), r[3] * r[4] * r[5] // max(r[6], 1), short(r[0])))
