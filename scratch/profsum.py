import sqlite3, re, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
rows=cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot=sum(r[2] for r in rows)
def short(n):
    n=n.replace('(anonymous namespace)::',''); n=re.sub(r'\(.*','',n); n=n.replace('void ','')
    return n[:100]
print("total kernel ms", round(tot/1e6,1), "launches", sum(r[1] for r in rows))
out=["name,calls,total_ms,avg_us,min_us,max_us,pct"]
for r in rows:
    out.append(f"\"{short(r[0])}\",{r[1]},{r[2]/1e6:.3f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.2f},{100*r[2]/tot:.2f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 30]:
    print(f"{short(r[0]):100s} {r[1]:7d} {r[2]/1e6:9.1f} ms avg {r[3]/1e3:8.1f} us {100*r[2]/tot:5.1f}%")
if len(sys.argv)>2 and sys.argv[2]!='-': open(sys.argv[2],'w').write("\n".join(out)+"\n")
