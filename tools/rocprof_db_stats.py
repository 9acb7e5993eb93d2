import sqlite3, sys, glob
for f in sys.argv[1:]:
    c=sqlite3.connect(f)
    rows=c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc limit 22").fetchall()
    tot=sum(r[3] for r in c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name").fetchall())
    print(f, "total kernel ms", tot/1e6)
    for r in rows: print(f"{r[1]:5d} {r[2]/1e3:10.1f} us  {r[3]/1e6:9.2f} ms  {r[0][:100]}")
