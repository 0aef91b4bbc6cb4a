import sqlite3,sys,re,collections
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
pmc=[t for t in tabs if 'pmc_event' in t][0]; kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]; pi=[t for t in tabs if 'info_pmc' in t][0]
q=f"select s.kernel_name, i.name, sum(p.value), count(distinct d.id) from {pmc} p join {kd} d on p.event_id=d.event_id join {ks} s on d.kernel_id=s.id join {pi} i on p.pmc_id=i.id group by s.kernel_name, i.name"
res=collections.defaultdict(dict)
for k,c,v,n in db.execute(q):
    k=re.sub(r'\(.*','',k)
    res[k][c]=(v,n)
for k in res:
    if any(x in k for x in sys.argv[2:]):
        print(k[:60], {c:(round(v/n),n) for c,(v,n) in res[k].items()})
